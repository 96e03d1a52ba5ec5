cd $GRAFT_REPO_ROOT; O=gpurun_out/r02h; mkdir -p $O
run () { name=$1; shift; ( timeout 300 tests/gpu_bench "$@" 2>&1 | tail -3 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
timeout 600 tests/gpu_quick 512 > $O/gpu_quick.log 2>&1; tail -2 $O/gpu_quick.log
run base 10 262144 16384 2 50 16 d
run base 30 262144 16384 2 50 16 d
run base 21 262144 16384 2 50 16 d
run base 41 262144 16384 2 50 16 d
run base 10 4194304 1024 2 50 16 d
( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ) | tee -a $O/summary.txt
tail -12 $O/pytest.log
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" ) | tee -a $O/summary.txt
tail -3 $O/bench.err; cut -c1-600 $O/bench.json
