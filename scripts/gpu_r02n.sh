# chain build through hash bins + LDS head tables: parity sanity, throughput, phase split
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02n; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
( timeout 120 tests/lds_atomic_order ) 2>&1 | tee -a $O/summary.txt
for l in 13 14 15 16 17 34 35 38; do ( timeout 600 tests/gpu_quick 256 $l 2>&1 | tail -3 ) | tee -a $O/summary.txt; done
for l in 13 15 35; do ( timeout 300 tests/gpu_bench $l 262144 8192 2 2>&1 | tail -1 ) | tee -a $O/summary.txt; done
( timeout 300 tests/gpu_bench 13 4194304 512 2 2>&1 | tail -1 ) | tee -a $O/summary.txt
export LIZARDGPU_CHUNK_MB=8192
for l in 13 15; do
( LD_LIBRARY_PATH=$V/prof timeout 300 tests/gpu_quick 8192 $l 1 2>&1 | grep -E "batch|prof" | sed "s/^/[prof L$l] /" ) | tee -a $O/summary.txt
done
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_random_parity.py -m gpu -x -q 2>&1 | tail -5 ) | tee -a $O/summary.txt
