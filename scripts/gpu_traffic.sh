# Calibrated HBM-side traffic of one launch of the bench config: size-binned L2->fabric request counters.
TAG=${1:-r1}; shift; ARGS="$@"
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/traffic_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass () { name=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --verify 0 $ARGS > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$name" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r.get('Kernel_Name','') for k in ('lz_fast', 'lz_price', 'lz_hashchain')): agg[r['Counter_Name']] += float(r['Counter_Value'])
print(sys.argv[2], dict(agg))
PY
}
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum | tee $OUT/summary.txt
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum | tee -a $OUT/summary.txt
pass dram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum | tee -a $OUT/summary.txt
pass fw FETCH_SIZE | tee -a $OUT/summary.txt
pass ww WRITE_SIZE | tee -a $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete
