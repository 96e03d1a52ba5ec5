#!/bin/bash
# The 1 -> 8 GPU scaling run of BASELINE.json (north_star: "throughput at 1/2/4/8 GPUs"), ready for the day a multi-GPU node is
# there: bench.py with identical per-GPU work at every N (weak scaling), one rank per GPU over RCCL, the library's own size gather
# inside the timed step.  bench.py refuses to print a line when the library's RCCL communicator does not report N ranks
# (config.size_gather_transport.rccl_ranks_seen), so every line of the table below is certified by the transport itself.
#   bash scripts/scale.sh [steps=20] [warmup=3] ["1 2 4 8"] [extra bench.py arguments, e.g. --headline-only]
# Prints one row per N: whole-job MB/s, MB/s per GPU, efficiency vs N = 1, slowest rank's kernel ms, largest gather us; the JSON
# lines are kept under gpurun_out/scale/.
STEPS=${1:-20}; WARMUP=${2:-3}; NS=${3:-"1 2 4 8"}; shift; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"; O=gpurun_out/scale; mkdir -p $O
NDEV=$(python -c "import torch; print(torch.cuda.device_count())")
for N in $NS; do
  if [ "$N" -gt "$NDEV" ]; then echo "N=$N: only $NDEV device(s) visible — skipped (bench.py --transport host-bounce runs the code path on fewer devices, as a functional check)"; continue; fi
  # (python bench.py --gpus N launches its own ranks under torch.distributed.run since round 6 — the driver's form)
  timeout 1800 python bench.py --gpus $N --steps $STEPS --warmup $WARMUP --no-cpu "$@" > $O/n$N.json 2> $O/n$N.err
  echo "N=$N rc=$?"
done
python - "$O" $NS <<'PY'
import json, sys
d, ns = sys.argv[1], [int(x) for x in sys.argv[2:]]
rows = {}
for n in ns:
    try:
        line = [l for l in open(f"{d}/n{n}.json") if l.startswith("{")][-1]
        rows[n] = json.loads(line)
    except (OSError, IndexError, ValueError):
        pass
if 1 not in rows:
    sys.exit("no N = 1 line: nothing to compare with")
base = rows[1]["value"]
print(f"{'N':>2} {'MB/s (job)':>12} {'MB/s per GPU':>13} {'efficiency':>10} {'max kernel ms':>14} {'max gather us':>14}  rccl ranks seen / version")
for n in sorted(rows):
    r = rows[n]
    pr = r.get("per_rank") or []
    tr = (r.get("config") or {}).get("size_gather_transport") or {}
    print(f"{n:>2} {r['value']:>12.0f} {r['value'] / n:>13.0f} {r['value'] / (n * base):>10.3f} "
          f"{max([p['kernel_ms'] for p in pr], default=(r.get('roofline') or {}).get('avg_kernel_ms', float('nan'))):>14.3f} {max([p['gather_us'] for p in pr], default=0.0):>14.1f}  "
          f"{tr.get('rccl_ranks_seen', '-')} / {tr.get('rccl_version', '-')}")
    for c in r.get("configs", []):
        print(f"   {c['workload'][:60]:<60} {c['value']:>12.0f} MB/s")
PY
