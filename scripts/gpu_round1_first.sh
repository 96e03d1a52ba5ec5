set -x
mkdir -p gpurun_out
rocm-smi --showproductname 2>&1 | head -8 > gpurun_out/gpu_info.txt; nproc >> gpurun_out/gpu_info.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/gpu_info.txt; free -g | head -2 >> gpurun_out/gpu_info.txt
timeout 900 python -m pytest tests -m gpu -x -q --timeout 400 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
timeout 600 python bench.py --blocks 4096 --steps 2 --warmup 1 2>&1 | tail -3 | tee gpurun_out/bench_4096.json
timeout 900 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench_full.json
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --blocks 16384 --steps 2 --warmup 1 --no-cpu 2>&1 | tail -3
