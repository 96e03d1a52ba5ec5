# PMC passes over the level-10 kernel (each pass its own run: --pmc with --kernel-trace only).
# usage: bash scripts/gpu_pmc.sh <tag> [nblocks] [level]
TAG=${1:-r1}; NB=${2:-8192}; LV=${3:-10}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|TCC_[A-Z_0-9]+_sum|GRBM_[A-Z_]+|TCP_[A-Z_0-9]+_sum)\b" | sort -u > $OUT/available_counters.txt
wc -l $OUT/available_counters.txt
run_pass () {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- $ROOT/tests/gpu_quick $NB $LV 1 > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$name" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float); n = 0
for r in rows:
    if 'lz_' in r.get('Kernel_Name', '') and 'datagen' not in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value'])
disp = {r['Dispatch_Id'] for r in rows if 'lz_' in r.get('Kernel_Name', '') and 'datagen' not in r['Kernel_Name']}
print(sys.argv[2], 'dispatches', len(disp), {k: v for k, v in agg.items()})
PY
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU | tee $OUT/summary.txt
run_pass sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM | tee -a $OUT/summary.txt
run_pass sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL | tee -a $OUT/summary.txt
run_pass fetch FETCH_SIZE | tee -a $OUT/summary.txt
run_pass write WRITE_SIZE | tee -a $OUT/summary.txt
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum | tee -a $OUT/summary.txt
grep -h "batch" $OUT/*.log | head -8
# keep only small files
find $OUT -name "*.csv" -size +2M -delete
du -sh $OUT
