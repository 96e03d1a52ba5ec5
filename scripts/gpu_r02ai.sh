# host path: threads per staging copy
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02ai; mkdir -p $O
V=$GRAFT_REPO_ROOT/lizard_amd/variants
for v in base copy8 copy2; do
  ( if [ "$v" != base ]; then export LD_LIBRARY_PATH=$V/$v; fi; timeout 600 python bench.py --steps 1 --warmup 0 --headline-only --blocks 16384 --verify 0 --cpu-seconds 0.5 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['end_to_end']['pageable_src'], j['end_to_end']['pinned_src'], j['frames']['no_content_checksum'], j['frames']['with_xxh32_content_checksum'])" ) | tee -a $O/summary.txt
done
