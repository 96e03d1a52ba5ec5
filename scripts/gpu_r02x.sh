# level 10: 20-bit slots (3 check bits) -> 14 / 15 tables per CU
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02x; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for v in base x13n x14 x15; do run $v 10 262144 65536 3; done
for v in x15; do ( LD_LIBRARY_PATH=$V/$v timeout 300 tests/gpu_quick 512 10 2>&1 | tail -3 | sed "s/^/[$v] /" ) | tee -a $O/summary.txt; done
