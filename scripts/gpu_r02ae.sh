# experiment: global-memory hash tables in fine-grained / uncached device memory (request size at the fabric)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02ae; mkdir -p $O
run () { name=$1; shift; ( timeout 300 tests/gpu_bench "$@" 2>&1 | tail -1 | sed "s/^/[$name] /" ) | tee -a $O/summary.txt; }
for m in 0 1 3; do export LIZARDGPU_TABLE_MEM=$m; for l in 11 21 22; do run mem$m $l 262144 16384 2; done; done
export LIZARDGPU_TABLE_MEM=3
bash scripts/gpu_traffic2.sh r02ae "11 262144 16384" > $O/traffic.log 2>&1; grep -E "^L" $O/traffic.log | tee -a $O/summary.txt
