cd $GRAFT_REPO_ROOT; O=gpurun_out/r02aq; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for v in base synclight base synclight; do run $v 10 262144 65536 3; done
run synclight 30 262144 16384 3
