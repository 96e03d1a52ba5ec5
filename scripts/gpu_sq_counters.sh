# SQ counter passes over one gpu_bench configuration: bash scripts/gpu_pmc2.sh <tag> <variant|base> <level> <blockSize> <nBlocks>
TAG=$1; VAR=$2; LV=$3; BS=$4; NB=$5
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
if [ "$VAR" != base ]; then export LD_LIBRARY_PATH=$ROOT/lizard_amd/variants/$VAR; fi
cd /tmp && export TMPDIR=/tmp
run_pass () {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- $ROOT/tests/gpu_bench $LV $BS $NB 1 50 4 > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$name" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float)
sel = [r for r in rows if 'lz_' in r.get('Kernel_Name', '') and 'datagen' not in r['Kernel_Name'] and 'scan' not in r['Kernel_Name'] and 'gather' not in r['Kernel_Name']]
disp = sorted({int(r['Dispatch_Id']) for r in sel})
last = disp[-1] if disp else None
for r in sel:
    if int(r['Dispatch_Id']) == last: agg[r['Counter_Name']] += float(r['Counter_Value'])
print(sys.argv[2], 'dispatches', len(disp), 'last only:', {k: v for k, v in agg.items()})
PY
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU | tee $OUT/summary.txt
run_pass sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM | tee -a $OUT/summary.txt
run_pass sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL | tee -a $OUT/summary.txt
grep -h "kernel" $OUT/*.log | head -4
find $OUT -name "*.csv" -size +2M -delete
