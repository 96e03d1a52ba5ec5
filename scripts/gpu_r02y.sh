cd $GRAFT_REPO_ROOT; O=gpurun_out/r02y; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
export LIZARDGPU_CHUNK_MB=8192
for l in 13 16 17; do
( LD_LIBRARY_PATH=$V/prof timeout 300 tests/gpu_quick 16384 $l 1 2>&1 | grep -E "batch|prof raw" | sed "s/^/[prof L$l] /" ) | tee -a $O/summary.txt
done
