# experiments (wrong results on purpose): where do passes 2 and 3 of the chain build wait?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02r; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
export LIZARDGPU_CHUNK_MB=8192
for v in prof hcx_flat3 hcx_noxchg; do
( LD_LIBRARY_PATH=$V/$v timeout 300 tests/gpu_quick 8192 13 1 2>&1 | grep -E "batch|prof raw" | sed "s/^/[$v] /" ) | tee -a $O/summary.txt
done
