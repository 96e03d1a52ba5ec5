# hashChain: hit bits ahead of the parse
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02u; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for l in 13 14 15 16 17 34 35 36 37 38; do ( timeout 300 tests/gpu_quick 256 $l 2>&1 | tail -2 | head -1 ) | tee -a $O/summary.txt; done
for v in base hc_b8 hc_b2; do run $v 13 262144 16384 2; done
for l in 14 15 16 17 35; do run base $l 262144 16384 2; done
run base 13 4194304 1024 2
run base 13 65536 65536 2
export LIZARDGPU_CHUNK_MB=8192
( LD_LIBRARY_PATH=$V/prof timeout 300 tests/gpu_quick 16384 13 1 2>&1 | grep -E "batch|prof raw" | sed "s/^/[prof L13] /" ) | tee -a $O/summary.txt
( LD_LIBRARY_PATH=$V/prof timeout 300 tests/gpu_quick 16384 15 1 2>&1 | grep -E "batch|prof raw" | sed "s/^/[prof L15] /" ) | tee -a $O/summary.txt
