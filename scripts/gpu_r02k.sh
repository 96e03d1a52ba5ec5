cd $GRAFT_REPO_ROOT; O=gpurun_out/r02k; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -2 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
timeout 600 tests/gpu_quick 512 21 | tail -2; timeout 600 tests/gpu_quick 512 41 | tail -2
run base 21 262144 16384 3
run pf4_w4 21 262144 16384 3
run base 41 262144 16384 3
run base 21 1048576 4096 2
