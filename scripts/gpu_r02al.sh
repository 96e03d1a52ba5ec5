# levels 11 / 31: occupancy summary of the 2^18-slot table in LDS (probes of groups never written skip the table)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02al; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
run () { name=$1; shift; ( if [ "$name" != base ]; then export LD_LIBRARY_PATH=$V/$name; fi; timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for l in 11 31; do ( timeout 300 tests/gpu_quick 512 $l 2>&1 | tail -2 | head -1 ) | tee -a $O/summary.txt; done
for v in base noocc; do run $v 11 262144 16384 2; run $v 31 262144 16384 2; done
run base 11 4194304 1024 2
run base 11 65536 65536 2
bash scripts/gpu_traffic2.sh r02al "11 262144 16384" > $O/traffic.log 2>&1; grep -E "^L" $O/traffic.log | tee -a $O/summary.txt
