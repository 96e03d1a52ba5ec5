# Second PMC set: latency / queueing view. usage: bash scripts/gpu_pmc2.sh <tag> [nblocks] [level]
TAG=${1:-r1}; NB=${2:-13312}; LV=${3:-10}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc2_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_pass () {
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- $ROOT/tests/gpu_quick $NB $LV 1 > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$name" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
big = max((int(r['Grid_Size']) for r in rows if 'lz_' in r.get('Kernel_Name','')), default=0)
agg = collections.defaultdict(float)
for r in rows:
    if 'lz_' in r.get('Kernel_Name', '') and int(r['Grid_Size']) == big: agg[r['Counter_Name']] += float(r['Counter_Value'])
print(sys.argv[2], 'grid', big, dict(agg))
PY
}
run_pass a SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES | tee $OUT/summary.txt
run_pass b SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS | tee -a $OUT/summary.txt
run_pass c TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum | tee -a $OUT/summary.txt
run_pass d TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum | tee -a $OUT/summary.txt
run_pass e TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum | tee -a $OUT/summary.txt
run_pass f GRBM_GUI_ACTIVE GRBM_TA_BUSY | tee -a $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete
