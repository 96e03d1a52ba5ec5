# Full validation + measurement pass (run with gpurun). usage: bash scripts/gpu_full_validate.sh <tag>
TAG=${1:-r1}
set -x
mkdir -p gpurun_out/$TAG
O=gpurun_out/$TAG
timeout 120 ./tests/gpu_quick 2048 > $O/quick.log 2>&1; echo rc=$? >> $O/quick.log; grep -E "batch|gpu_quick|rc=|FAIL" $O/quick.log
timeout 500 python bench.py > $O/bench_L10_full.json 2> $O/bench_L10_full.err; tail -1 $O/bench_L10_full.json
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -8 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.log
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o stats -- python $GRAFT_REPO_ROOT/bench.py --blocks 16384 --steps 2 --warmup 1 --no-cpu --verify 0 > $GRAFT_REPO_ROOT/$O/prof_bench.json 2>&1); grep -h metric $O/prof_bench.json | tail -1
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --verify 0 > $GRAFT_REPO_ROOT/$O/pmc_fetch_bench.json 2>&1)
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -o write -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --verify 0 > $GRAFT_REPO_ROOT/$O/pmc_write_bench.json 2>&1)
python3 - <<'PY'
import csv, glob, os
O = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', os.environ.get('TAG', ''))
PY
for f in $(find $O -name "*counter_collection.csv"); do python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'lz_fast' in r.get('Kernel_Name', '') or 'lz_price' in r.get('Kernel_Name', '')]
for r in rows: print(sys.argv[1].split('/')[-1], r['Kernel_Name'][:50], r['Counter_Name'], r['Counter_Value'], 'grid', r.get('Grid_Size'))
PY
done | tee $O/pmc_traffic.txt
timeout 300 python bench.py --level 30 --blocks 16384 --cpu-seconds 6 > $O/bench_L30_16k.json 2>/dev/null; tail -1 $O/bench_L30_16k.json
timeout 300 python bench.py --level 21 --blocks 16384 --cpu-seconds 6 > $O/bench_L21_16k.json 2>/dev/null; tail -1 $O/bench_L21_16k.json
timeout 300 python bench.py --level 10 --block-size 4194304 --blocks 4096 --cpu-blocks 16 --cpu-seconds 6 > $O/bench_L10_4MiB.json 2>/dev/null; tail -1 $O/bench_L10_4MiB.json
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete; du -sh $O
