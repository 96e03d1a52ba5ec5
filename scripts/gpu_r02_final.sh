# Round-2 validation session: sanity, GPU tests, the bench line, rocprofv3 kernel stats of the same command, counter passes, soak.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02fin2; mkdir -p $O
( timeout 600 tests/gpu_quick 512 > $O/gpu_quick.log 2>&1; echo "gpu_quick rc=$?" ) | tee $O/summary.txt
( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ) | tee -a $O/summary.txt; tail -4 $O/pytest.log
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" ) | tee -a $O/summary.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" ) | tee -a $O/summary.txt
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/rocprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err; echo "rocprof rc=$?" ) | tee -a $O/summary.txt
find $O/rocprof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_bench.csv \;
find $O/rocprof -name "*.csv" -size +1M -delete
bash scripts/gpu_traffic2.sh r02fin2 "10 262144 65536" "10 4194304 6656" "30 262144 16384" "21 262144 16384" "13 262144 16384" "11 262144 16384" > $O/traffic.log 2>&1; grep -E "^L" $O/traffic.log | tee -a $O/summary.txt
for l in 10 21 30 41; do timeout 200 tests/gpu_bench $l 262144 16384 2 50 16 d 2>&1 | tail -2 | tee -a $O/summary.txt; done
( timeout 400 python scripts/gpu_soak.py 2000 240 > $O/soak.log 2>&1; echo "soak rc=$?" ) | tee -a $O/summary.txt; tail -2 $O/soak.log
for l in 13 14 15 16 17 35 11 31 22 41; do timeout 300 tests/gpu_bench $l 262144 16384 2 2>&1 | tail -1 | tee -a $O/summary.txt; done
