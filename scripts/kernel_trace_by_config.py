#!/usr/bin/env python3
"""Per-configuration kernel durations out of ONE rocprofv3 --kernel-trace pass over bench.py.

  python scripts/kernel_trace_by_config.py <bench_kernel_trace.csv> <bench line .json> > profiles/rNN_rocprofv3_kernel_durations_by_config.csv

rocprofv3's own *_kernel_stats.csv has one row per kernel NAME, and three of the bench line's configurations run the same kernel
(lz_fast12_split_kernel<false>: headline, 4 MiB weak, 4 MiB strong, the blocks_in_flight curve).  bench.py launches in a fixed
order — for every configuration `warmup` launches then `steps` timed ones, then blocks_in_flight (3 launches per point, the first
untimed) — so the compress dispatches of the trace, in dispatch order, are cut into those groups here and the mean of each group's
TIMED launches is printed beside the avg_kernel_ms the bench line itself reports (HIP events) for the same launches."""
import csv
import json
import re
import sys

trace, line = sys.argv[1], sys.argv[2]


def kname(r):
    return re.search(r"lz_\w+(<[^>]*>)?", r["Kernel_Name"]).group(0)


d = json.loads([l for l in open(line) if l.startswith("{")][-1])
rows = [r for r in csv.DictReader(open(trace)) if "lz_" in r["Kernel_Name"] and "datagen" not in r["Kernel_Name"] and "selfcheck" not in r["Kernel_Name"]
        and "scan" not in r["Kernel_Name"] and "gather" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
cfgs = [dict(d, workload=d["config"]["workload"])] + d.get("configs", [])
w, k = d["warmup"], d["steps"]
out = csv.writer(sys.stdout)
out.writerow(["configuration", "kernel", "timed_launches", "rocprofv3_mean_ms", "rocprofv3_min_ms", "rocprofv3_max_ms", "bench_line_hip_event_mean_ms", "grid_x", "lds_bytes", "vgprs"])
i = 0
for c in cfgs:
    grp = rows[i + w:i + w + k]; i += w + k
    ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in grp]
    name = kname(grp[0])
    out.writerow([c["workload"], name, len(ms), "%.3f" % (sum(ms) / len(ms)), "%.3f" % min(ms), "%.3f" % max(ms), c["roofline"]["avg_kernel_ms"],
                  grp[0]["Grid_Size_X"], grp[0]["LDS_Block_Size"], grp[0]["VGPR_Count"]])
for c in cfgs:                                              # the 65 536-block reading of the 16 384-block configurations: 4 launches each, the first untimed
    e = c.get("same_inputs_as_headline")
    if not e:
        continue
    grp = rows[i + 1:i + 4]; i += 4
    ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in grp]
    out.writerow(["same inputs as the headline: L%d %d x 262144" % (c["level"], e["blocks_per_gpu"]), kname(grp[0]), len(ms), "%.3f" % (sum(ms) / len(ms)),
                  "%.3f" % min(ms), "%.3f" % max(ms), e["avg_kernel_ms"], grp[0]["Grid_Size_X"], grp[0]["LDS_Block_Size"], grp[0]["VGPR_Count"]])
for p in (d.get("blocks_in_flight") or {}).get("curve", []):
    grp = rows[i + 1:i + 3]; i += 3
    ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in grp]
    out.writerow(["blocks_in_flight: L10 %d x 4 MiB" % p["blocks"], kname(grp[0]), len(ms), "%.3f" % (sum(ms) / len(ms)),
                  "%.3f" % min(ms), "%.3f" % max(ms), "%.3f" % (p["blocks"] * (4 << 20) / p["GB_s"] / 1e6), grp[0]["Grid_Size_X"], grp[0]["LDS_Block_Size"], grp[0]["VGPR_Count"]])
if i != len(rows):
    sys.exit("kernel_trace_by_config: %d compress dispatches in the trace, the bench line accounts for %d" % (len(rows), i))
