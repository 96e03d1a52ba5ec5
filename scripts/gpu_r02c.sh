cd $GRAFT_REPO_ROOT; O=gpurun_out/r02c; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 180 tests/gpu_bench "$@" 2>&1 | tail -2 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for v in base pf2_w10t11 pf2_w14t10 pf2_w8t11 pf2_w4 pf2_prio pf2_w10t11_prio; do run $v 21 262144 16384 3; done
run base 41 262144 16384 3
run base 21 1048576 4096 3
export LIZARDGPU_CHUNK_MB=4096
( LD_LIBRARY_PATH=$V/prof_w4 timeout 300 tests/gpu_quick 4096 21 1 2>&1 | grep -E "batch|prof" | sed "s/^/[prof_w4] /" ) | tee -a $O/summary.txt
