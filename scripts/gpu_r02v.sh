cd $GRAFT_REPO_ROOT; O=gpurun_out/r02v; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
( timeout 300 tests/gpu_quick 256 13 2>&1 | tail -2 | head -1 ) | tee -a $O/summary.txt
for v in base hc_w12; do run $v 13 262144 16384 2; done
run base 15 262144 16384 2
bash scripts/gpu_traffic2.sh r02v "13 262144 16384" > $O/traffic.log 2>&1; grep -E "^L" $O/traffic.log | tee -a $O/summary.txt
