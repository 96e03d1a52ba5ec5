cd $GRAFT_REPO_ROOT; O=gpurun_out/r02j; mkdir -p $O
run () { name=$1; shift; ( timeout 300 tests/gpu_bench "$@" 2>&1 | tail -2 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
timeout 600 tests/gpu_quick 512 > $O/gpu_quick.log 2>&1; grep -E "FAIL|all ok|FAILED" $O/gpu_quick.log | head
run base 10 262144 65536 3
run base 30 262144 16384 3
run base 10 4194304 6656 2
run base 21 1048576 4096 2
run base 10 8388608 1024 2
timeout 300 python -m pytest tests/test_random_parity.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
