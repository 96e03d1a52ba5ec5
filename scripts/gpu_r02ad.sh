# hashChain: searches walk a packed word (two links) per memory trip
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02ad; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for l in 13 15 17 34 36 38; do ( timeout 300 tests/gpu_quick 256 $l 2>&1 | tail -2 | head -1 ) | tee -a $O/summary.txt; done
for l in 13 14 15 16 17 35; do run base $l 262144 16384 2; done
export LIZARDGPU_CHUNK_MB=8192
( LD_LIBRARY_PATH=$V/prof timeout 300 tests/gpu_quick 16384 13 1 2>&1 | grep -E "batch|prof raw" | sed "s/^/[prof L13] /" ) | tee -a $O/summary.txt
