#!/usr/bin/env python3
"""Record a fabric-traffic counter pass (output of scripts/gpu_traffic.sh) in profiles/pmc_traffic.json.

    python scripts/record_traffic.py <round tag> <summary file under profiles/> [<git commit the pass ran on>]

One entry per configuration line group ("L21_16384x262144 rd {...}", "wr", "fw", "ww"): reads = 128-byte requests x 128
(every fabric read of these kernels is a 128-byte request; FETCH_SIZE counts them at 64 B on gfx950), writes = WRITE_SIZE (KB).
Each entry carries the hash of the device sources (lizard_amd/csrc/lz_*.h) of the CURRENT tree: run this on the tree the pass
ran on.  bench.py quotes an entry only while that hash is the tree's (roofline.traffic_source)."""
import ast
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    tag, summary = sys.argv[1], sys.argv[2]
    commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    import bench
    sha = bench.kernel_source_sha16()
    groups = {}
    for line in open(os.path.join(ROOT, summary)):
        m = re.match(r"^L(\d+)_(\d+)x(\d+) (rd|wr|fw|ww) (\{.*\})\s*$", line)
        if m:
            groups.setdefault((int(m.group(1)), int(m.group(3)), int(m.group(2))), {}).update(ast.literal_eval(m.group(5)))
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    doc = json.load(open(path))
    for (level, bs, nb), c in groups.items():
        if not all(k in c for k in ("TCC_EA0_RDREQ_128B_sum", "WRITE_SIZE")):
            print("incomplete pass for", level, bs, nb, sorted(c)); continue
        reads = int(c["TCC_EA0_RDREQ_128B_sum"] * 128 + c.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + c.get("TCC_EA0_RDREQ_32B_sum", 0) * 32)
        writes = int(c["WRITE_SIZE"] * 1024)
        ent = {"round": tag, "level": level, "block_size": bs, "blocks_per_gpu": nb,
               "FETCH_SIZE_KB": c.get("FETCH_SIZE"), "WRITE_SIZE_KB": c["WRITE_SIZE"],
               "TCC_EA0_RDREQ_sum": int(c.get("TCC_EA0_RDREQ_sum", 0)), "TCC_EA0_RDREQ_128B_sum": int(c["TCC_EA0_RDREQ_128B_sum"]),
               "TCC_EA0_RDREQ_64B_sum": int(c.get("TCC_EA0_RDREQ_64B_sum", 0)), "TCC_EA0_RDREQ_32B_sum": int(c.get("TCC_EA0_RDREQ_32B_sum", 0)),
               "TCC_EA0_WRREQ_sum": int(c.get("TCC_EA0_WRREQ_sum", 0)), "TCC_EA0_WRREQ_64B_sum": int(c.get("TCC_EA0_WRREQ_64B_sum", 0)),
               "read_bytes": reads, "write_bytes": writes, "traffic_bytes": reads + writes,
               "source": summary, "commit": commit, "kernel_source_sha16": sha}
        doc["entries"].append(ent)
        print("recorded", level, bs, nb, "traffic", reads + writes)
    json.dump(doc, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
