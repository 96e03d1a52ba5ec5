# level 10 / 30: 21-bit slots (4 check bits) -> 14 tables per CU (12 + 4 at level 30)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02at; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for v in base t21_w13 t21_w14; do run $v 10 262144 65536 3; done
for v in base t21_w14; do run $v 30 262144 16384 3; done
( LD_LIBRARY_PATH=$V/t21_w14 timeout 300 tests/gpu_quick 512 10 2>&1 | tail -2 | sed "s/^/[t21_w14] /" ) | tee -a $O/summary.txt
