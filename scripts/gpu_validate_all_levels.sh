# Validation + measurement pass over every GPU level (run with gpurun). usage: TAG=r01g bash scripts/gpu_validate_all_levels.sh
TAG=${TAG:-r1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 200 ./tests/gpu_quick 2048 > $O/quick.log 2>&1; echo rc=$? >> $O/quick.log; grep -E "batch|gpu_quick|rc=|FAIL" $O/quick.log
timeout 400 python bench.py > $O/bench_L10_full.json 2> $O/bench_L10_full.err; tail -1 $O/bench_L10_full.json
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o stats -- python $R/bench.py --blocks 16384 --steps 2 --warmup 1 --no-cpu --verify 0 > $O/prof_bench_L10.json 2>&1; grep -h metric $O/prof_bench_L10.json | tail -1
pmc () { name=$1; ctr=$2; shift; shift
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$name -o $name -- python $R/bench.py --steps 1 --warmup 0 --no-cpu --verify 0 "$@" > $O/pmc_$name.log 2>&1
  f=$(find $O/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$name" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get('Kernel_Name', '')
    if 'lz_fast' in k or 'lz_price' in k or 'lz_hashchain' in k:
        agg[(k[:60], r['Counter_Name'])] += float(r['Counter_Value']); cnt[(k[:60], r['Counter_Name'])] += 1
for k, v in agg.items(): print(sys.argv[2], k, v, 'launches', cnt[k])
PY
}
pmc L10_fetch FETCH_SIZE | tee $O/pmc_traffic.txt
pmc L10_write WRITE_SIZE | tee -a $O/pmc_traffic.txt
pmc L13_fetch FETCH_SIZE --level 13 --blocks 8192 | tee -a $O/pmc_traffic.txt
pmc L13_write WRITE_SIZE --level 13 --blocks 8192 | tee -a $O/pmc_traffic.txt
cd $R
for cfg in "30 16384" "21 16384" "41 16384" "22 16384" "42 16384" "11 16384" "31 16384" "13 8192" "15 8192" "16 8192" "17 8192" "35 8192"; do
  set -- $cfg
  timeout 300 python bench.py --level $1 --blocks $2 --cpu-seconds 4 --cpu-blocks 64 > $O/bench_L$1_$2.json 2>/dev/null; tail -1 $O/bench_L$1_$2.json | cut -c1-400
done
timeout 300 python bench.py --level 10 --block-size 4194304 --blocks 4096 --cpu-blocks 16 --cpu-seconds 4 > $O/bench_L10_4MiB.json 2>/dev/null; tail -1 $O/bench_L10_4MiB.json | cut -c1-300
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete; du -sh $O
