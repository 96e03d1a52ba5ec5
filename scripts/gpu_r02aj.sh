# priceFast tables in global memory: 8 check bits beside the position (candidate bytes fetched only when they can match)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02aj; mkdir -p $O
run () { name=$1; shift; ( timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for l in 21 22 41 42; do ( timeout 300 tests/gpu_quick 512 $l 2>&1 | tail -2 | head -1 ) | tee -a $O/summary.txt; done
for l in 21 22 41 42; do run chk $l 262144 16384 2; done
run chk 21 4194304 1024 2
bash scripts/gpu_traffic2.sh r02aj "21 262144 16384" "22 262144 16384" > $O/traffic.log 2>&1; grep -E "^L" $O/traffic.log | tee -a $O/summary.txt
