# levels 22 / 42: occupancy summary of the 2^18-slot priceFast table
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02an; mkdir -p $O
run () { name=$1; shift; ( timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for l in 22 42; do ( timeout 300 tests/gpu_quick 512 $l 2>&1 | tail -2 | head -1 ) | tee -a $O/summary.txt; done
run occ 22 262144 16384 2
run occ 42 262144 16384 2
run occ 22 1048576 4096 2
