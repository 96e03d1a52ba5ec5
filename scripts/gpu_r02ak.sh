# level 21 after the check bits: how many global-table waves beside the four LDS tables?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02ak; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for v in base pf_w10t10 pf_w12t10 pf_w14t10 pf_w15t10 pf_w16n3; do run $v 21 262144 16384 2; done
