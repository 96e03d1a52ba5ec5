cd $GRAFT_REPO_ROOT; O=gpurun_out/r02g; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 180 tests/gpu_bench "$@" 2>&1 | tail -2 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
timeout 300 tests/gpu_quick 512 30 | tail -2
run base 30 262144 16384 3
run base 41 262144 16384 3
run base 31 262144 16384 3
export LIZARDGPU_CHUNK_MB=4096
( LD_LIBRARY_PATH=$V/prof timeout 300 tests/gpu_quick 16384 30 1 2>&1 | grep -E "batch|prof|sub-phase" | sed "s/^/[prof] /" ) | tee -a $O/summary.txt
