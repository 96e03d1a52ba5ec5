cd $GRAFT_REPO_ROOT; O=gpurun_out/r02af; mkdir -p $O
run () { name=$1; shift; ( timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
( timeout 300 tests/gpu_quick 256 13 2>&1 | tail -2 | head -1 ) | tee -a $O/summary.txt
for l in 13 15 17; do run pool4 $l 262144 16384 2; done
run pool4 13 262144 8192 2
