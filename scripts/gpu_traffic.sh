# Fabric-side (L2 -> data fabric) traffic of ONE launch of the timed kernel: separate rocprofv3 --pmc passes (--kernel-trace only)
# over tests/gpu_bench (device-resident input, 1 warm-up + 1 measured launch; the LAST dispatch is reported).
#   bash scripts/gpu_traffic2.sh <tag> "<level> <blockSize> <nBlocks>" ["<level> <blockSize> <nBlocks>" ...]
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/traffic_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CFG in "$@"; do
  set -- $CFG; LV=$1; BS=$2; NB=$3; name=L${LV}_${NB}x${BS}
  pass () { p=$1; shift
    timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name/$p -o $p -- $ROOT/tests/gpu_bench $LV $BS $NB 1 50 4 > $OUT/$name.$p.log 2>&1
    f=$(find $OUT/$name/$p -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$name" "$p" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r.get('Kernel_Name', '') for k in ('lz_fast', 'lz_price', 'lz_hashchain'))]
last = max(int(r['Dispatch_Id']) for r in rows)
agg = collections.defaultdict(float)
for r in rows:
    if int(r['Dispatch_Id']) == last: agg[r['Counter_Name']] += float(r['Counter_Value'])
print(sys.argv[2], sys.argv[3], dict(agg))
PY
  }
  pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum | tee -a $OUT/summary.txt
  pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum | tee -a $OUT/summary.txt
  pass fw FETCH_SIZE | tee -a $OUT/summary.txt
  pass ww WRITE_SIZE | tee -a $OUT/summary.txt
  grep -h "kernel" $OUT/$name.fw.log | tail -1 | tee -a $OUT/summary.txt
done
find $OUT -name "*.csv" -size +1M -delete
