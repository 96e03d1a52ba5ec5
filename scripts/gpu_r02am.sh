# level 41 after the check bits: more global-table waves beside the three LDS tables?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02am; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for v in base h41_w14 h41_w16 h41_w16n4; do run $v 41 262144 16384 2; done
