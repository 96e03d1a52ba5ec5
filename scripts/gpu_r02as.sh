# hashChain: candidate measurement with X's side requested beside the walk and the candidate's side in one batch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02as; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
( timeout 300 tests/gpu_quick 256 13 2>&1 | tail -2 | head -1 ) | tee -a $O/summary.txt
( timeout 300 tests/gpu_quick 256 37 2>&1 | tail -2 | head -1 ) | tee -a $O/summary.txt
for v in base hc_prev; do run $v 13 262144 16384 2; done
for l in 15 16; do run base $l 262144 16384 2; done
