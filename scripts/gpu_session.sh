# One GPU session (run through gpurun from the repo root; everything lands under gpurun_out/<tag>/, summaries are copied into
# profiles/ by hand).  Replaces the one-off scripts of rounds 1-2.
#   bash scripts/gpu_session.sh variants <tag> "<level> <blockSize> <nBlocks> [steps [P% [verify]]]" <variant|base> [<variant> ...]
#        tests/gpu_bench on one configuration under each tuning variant (make -C lizard_amd/csrc variant NAME=.. DEFS=..), one line each
#   bash scripts/gpu_session.sh quick <tag> <variant|base> [level]
#        tests/gpu_quick (all sizes against the oracle, hang watchdog) under a variant
#   bash scripts/gpu_session.sh validate <tag>
#        the round's validation pass: gpu_quick, pytest -m gpu, smoke, bench.py, rocprofv3 --kernel-trace --stats of the same
#        command, fabric-traffic counter passes of the bench configurations (scripts/gpu_traffic.sh), all-level gpu_bench lines
#   bash scripts/gpu_session.sh validate_slim <tag>    the same without gpu_quick, the rocprofv3 pass of the whole bench and the 11/13 counter passes
#   bash scripts/gpu_session.sh traffic <tag> "<level> <blockSize> <nBlocks>" ...     (scripts/gpu_traffic.sh)
#   bash scripts/gpu_session.sh sq <tag> <variant|base> <level> <blockSize> <nBlocks>  (scripts/gpu_sq_counters.sh)
MODE=$1; TAG=$2; shift; shift
export LIZARD_REQUIRE_REF=1      # the reference-program GPU tests FAIL (not skip) when oracle/_ref did not travel
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/$TAG; mkdir -p $O
V=$R/lizard_amd/variants
with_variant () { if [ "$1" != base ]; then export LD_LIBRARY_PATH=$V/$1; fi; }
case $MODE in
variants)
  CFG=$1; shift
  for v in "$@"; do ( with_variant $v; timeout 300 tests/gpu_bench $CFG 2>&1 | tail -1 | sed "s/^/[$v] /" ) | tee -a $O/summary.txt; done ;;
quick)
  ( with_variant $1; timeout 600 tests/gpu_quick 512 ${2:-} 2>&1 | tail -4 | sed "s/^/[$1] /" ) | tee -a $O/summary.txt ;;
traffic) bash scripts/gpu_traffic.sh $TAG "$@" ;;
sq)      bash scripts/gpu_sq_counters.sh $TAG "$@" ;;
validate)
  ( timeout 600 tests/gpu_quick 512 > $O/gpu_quick.log 2>&1; echo "gpu_quick rc=$?" ) | tee $O/summary.txt
  ( timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ) | tee -a $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
  ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" ) | tee -a $O/summary.txt
  ( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" ) | tee -a $O/summary.txt
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/rocprof -o bench -- python $R/bench.py --no-cpu > $R/$O/bench_under_rocprof.json 2> $R/$O/rocprof.err; echo "rocprof rc=$?" ) | tee -a $O/summary.txt
  find $O/rocprof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_bench.csv \;
  # the headline configuration alone: its kernel's mean duration must agree with the bench line's HIP events
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/rocprof_headline -o headline -- python $R/bench.py --headline-only --no-cpu > $R/$O/bench_headline_under_rocprof.json 2> $R/$O/rocprof_headline.err; echo "rocprof headline rc=$?" ) | tee -a $O/summary.txt
  find $O/rocprof_headline -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_bench_headline_only.csv \;
  find $O/rocprof_headline -name "*.csv" -size +1M -delete; find $O/rocprof_headline -name "*.db" -delete
  find $O/rocprof -name "*.csv" -size +1M -delete; find $O/rocprof -name "*.db" -delete
  bash scripts/gpu_traffic.sh $TAG "10 262144 65536" "10 4194304 6656" "30 262144 16384" "21 262144 16384" "11 262144 16384" "13 262144 16384" "20 262144 16384" "12 262144 16384" > $O/traffic.log 2>&1
  grep -E "^L" $O/traffic.log | tee -a $O/summary.txt
  for l in 11 31 12 32 33 13 14 15 16 17 35 37 20 40 22 41 42; do timeout 300 tests/gpu_bench $l 262144 16384 2 50 1024 2>&1 | tail -1 | tee -a $O/summary.txt; done ;;
validate_slim)
  # the validation pass when GPU minutes are short: GPU tests, smoke, bench line, rocprofv3 of the headline configuration alone,
  # counter passes of the four bench configurations, one gpu_bench line per remaining level
  ( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ) | tee $O/summary.txt; tail -4 $O/pytest.log | tee -a $O/summary.txt
  ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" ) | tee -a $O/summary.txt
  ( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" ) | tee -a $O/summary.txt
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/rocprof_headline -o headline -- python $R/bench.py --headline-only --no-cpu > $R/$O/bench_headline_under_rocprof.json 2> $R/$O/rocprof_headline.err; echo "rocprof headline rc=$?" ) | tee -a $O/summary.txt
  find $O/rocprof_headline -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_bench_headline_only.csv \;
  find $O/rocprof_headline -name "*.csv" -size +1M -delete; find $O/rocprof_headline -name "*.db" -delete
  bash scripts/gpu_traffic.sh $TAG "10 262144 65536" "10 4194304 6656" "30 262144 16384" "21 262144 16384" > $O/traffic.log 2>&1
  grep -E "^L" $O/traffic.log | tee -a $O/summary.txt
  for l in 11 31 12 32 33 13 14 15 16 17 35 37 20 40 22 41 42; do timeout 300 tests/gpu_bench $l 262144 16384 2 50 1024 2>&1 | tail -1 | tee -a $O/summary.txt; done ;;
rocprof_all)
  # rocprofv3 kernel stats over EVERY BASELINE configuration of the bench line in one CSV
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/rocprof -o bench -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > $R/$O/bench_under_rocprof.json 2> $R/$O/rocprof.err; echo "rocprof rc=$?" ) | tee -a $O/summary.txt
  find $O/rocprof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_bench_all_configs.csv \;
  find $O/rocprof -name "*.csv" -size +1M -delete; find $O/rocprof -name "*.db" -delete
  head -12 $O/rocprofv3_kernel_stats_bench_all_configs.csv ;;
multirank)
  # the N > 1 bench path on the one device: the pytest form (self-launch and torchrun), then a larger self-launched line
  ( timeout 900 python -m pytest tests/test_bench_multirank.py -m gpu -q -x > $O/pytest_multirank.log 2>&1; echo "pytest multirank rc=$?" ) | tee -a $O/summary.txt; tail -5 $O/pytest_multirank.log
  ( timeout 600 python bench.py --gpus 2 --transport host-bounce --steps 2 --warmup 1 --headline-only --strong --blocks 2048 --cpu-seconds 1 --cpu-all-seconds 0 > $O/bench_n2_selflaunch.json 2> $O/bench_n2.err; echo "bench n2 self-launch rc=$?" ) | tee -a $O/summary.txt
  cut -c1-700 $O/bench_n2_selflaunch.json; tail -5 $O/bench_n2.err ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
