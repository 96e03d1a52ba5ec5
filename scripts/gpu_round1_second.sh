set -x
mkdir -p gpurun_out
timeout 120 ./tests/gpu_quick 2048 > gpurun_out/quick2048.log 2>&1; echo rc=$? >> gpurun_out/quick2048.log
grep -E "batch|gpu_quick|rc=" gpurun_out/quick2048.log
timeout 500 python bench.py > gpurun_out/bench_L10_full.json 2> gpurun_out/bench_L10_full.err; tail -2 gpurun_out/bench_L10_full.json; tail -3 gpurun_out/bench_L10_full.err
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --blocks 16384 --steps 2 --warmup 1 --no-cpu --verify 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_r1_bench.json 2>&1); tail -2 gpurun_out/prof_r1_bench.json
timeout 700 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
timeout 300 python bench.py --level 30 --blocks 16384 --cpu-seconds 6 > gpurun_out/bench_L30_16k.json 2>/dev/null; tail -1 gpurun_out/bench_L30_16k.json
timeout 300 python bench.py --level 21 --blocks 16384 --cpu-seconds 6 > gpurun_out/bench_L21_16k.json 2>/dev/null; tail -1 gpurun_out/bench_L21_16k.json
ls -R gpurun_out | head -40
