cd $GRAFT_REPO_ROOT; O=gpurun_out/r02b; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
export LIZARDGPU_CHUNK_MB=4096
for v in prof_w4 prof_w12 prof_w16; do
  ( LD_LIBRARY_PATH=$V/$v timeout 300 tests/gpu_quick 4096 21 1 2>&1 | grep -E "batch|prof" | sed "s/^/[$v] /" ) | tee -a $O/summary.txt
done
unset LIZARDGPU_CHUNK_MB
( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ) | tee -a $O/summary.txt
tail -25 $O/pytest.log
