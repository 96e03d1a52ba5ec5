cd $GRAFT_REPO_ROOT; O=gpurun_out/r02e; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 180 tests/gpu_bench "$@" 2>&1 | tail -2 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for v in base h_n10p4 h_n8p7 h_n9p5 h_old; do run $v 30 262144 16384 3; done
run base 30 4194304 1024 3
run base 21 262144 16384 3
timeout 300 tests/gpu_quick 512 30 | tail -4
