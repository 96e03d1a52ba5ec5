# hashChain phase split (instrumented build): chain build (slot 6) vs outer search rounds (0) vs the three searches (1-3) vs container
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02m; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
export LIZARDGPU_CHUNK_MB=8192
for l in 13 15 17 35; do
( LD_LIBRARY_PATH=$V/prof timeout 300 tests/gpu_quick 8192 $l 1 2>&1 | grep -E "batch|prof" | sed "s/^/[prof L$l] /" ) | tee -a $O/summary.txt
done
