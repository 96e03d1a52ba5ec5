#!/usr/bin/env python3
"""GPU: do small launches on several streams overlap?  Wall time of R rounds of one launch of NB blocks on each of S streams, queued
without host synchronisation.  Run once with LIZARDGPU_ARENAS=1 (every launch waits for the previous one) and once without.

    python scripts/arena_probe.py [streams] [blocks per launch] [level] [rounds]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                       # noqa: E402
import torch                                             # noqa: E402
import util                                              # noqa: E402
from lizard_amd import api, _lib                         # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    level = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    R = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    bs = 262144
    host = np.frombuffer(util.datagen(bs * nb, 0.5, 0.0, 1), dtype=np.uint8).copy()
    srcs = [torch.from_numpy(host).cuda() for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    bufs = [api.compress_blocks_device(srcs[k], bs, level) for k in range(S)]
    torch.cuda.synchronize()
    t0 = time.time()
    for r in range(R):
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                api.compress_blocks_device(srcs[k], bs, level, dst=bufs[k][0], sizes=bufs[k][1])
    torch.cuda.synchronize()
    dt = time.time() - t0
    L = _lib.lib()
    print("arenas %s (in use %d): %d streams x %d launches of %d blocks at level %d: %.1f ms, %.2f GB/s" % (
        os.environ.get("LIZARDGPU_ARENAS", "default"), L.LizardGPU_arenasInUse(), S, R, nb, level, dt * 1e3, S * R * nb * bs / dt / 1e9))


if __name__ == "__main__":
    main()
