cd $GRAFT_REPO_ROOT; O=gpurun_out/r02f; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 180 tests/gpu_bench "$@" 2>&1 | tail -2 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
timeout 300 tests/gpu_quick 512 30 | tail -3
for v in base h2_n10p0 h2_n11p4 h2_n10p6; do run $v 30 262144 16384 3; done
for v in base h2_pf41_w6n4; do run $v 41 262144 16384 3; done
run base 31 262144 16384 3
run base 35 262144 8192 3
run base 30 8388608 1024 2
export LIZARDGPU_CHUNK_MB=4096
( LD_LIBRARY_PATH=$V/prof timeout 300 tests/gpu_quick 4096 30 1 2>&1 | grep -E "batch|prof|sub-phase" | sed "s/^/[prof] /" ) | tee -a $O/summary.txt
