cd $GRAFT_REPO_ROOT; O=gpurun_out/r02i; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 180 tests/gpu_bench "$@" 2>&1 | tail -2 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for v in base h3_w12p5 h3_w12p4 h3_w13p3; do run $v 30 262144 16384 3; done
