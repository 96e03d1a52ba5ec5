# host path: issuing + draining threads, three chunks in flight
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02ah; mkdir -p $O
( timeout 900 python -m pytest tests/test_host_pipeline.py tests/test_frame_stream.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -3 ) | tee -a $O/summary.txt
for mb in 256 512 1024; do
  LIZARDGPU_CHUNK_MB=$mb timeout 600 python bench.py --steps 1 --warmup 0 --headline-only --blocks 16384 --verify 0 --cpu-seconds 0.5 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $mb MiB', j['end_to_end']['pageable_src'], j['end_to_end']['pinned_src'], j['frames']['no_content_checksum'], j['frames']['with_xxh32_content_checksum'])" | tee -a $O/summary.txt
done
