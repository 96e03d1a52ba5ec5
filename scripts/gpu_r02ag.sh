# host path: chunk size of the staged pipeline
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02ag; mkdir -p $O
for mb in 64 128 256 512; do
  LIZARDGPU_CHUNK_MB=$mb timeout 600 python bench.py --steps 1 --warmup 0 --headline-only --blocks 16384 --verify 0 --cpu-seconds 0.5 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $mb MiB', j['end_to_end']['pageable_src'], j['end_to_end']['pinned_src'], j['frames']['no_content_checksum'], j['frames']['with_xxh32_content_checksum'])" | tee -a $O/summary.txt
done
