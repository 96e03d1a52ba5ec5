cd $GRAFT_REPO_ROOT; O=gpurun_out/r02l; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -2 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for v in base x_w12 x_w11 x_w10 x_skipnt; do run $v 10 262144 65536 3; done
export LIZARDGPU_CHUNK_MB=8192
( LD_LIBRARY_PATH=$V/prof timeout 300 tests/gpu_quick 16384 10 1 2>&1 | grep -E "batch|prof" | sed "s/^/[prof] /" ) | tee -a $O/summary.txt
