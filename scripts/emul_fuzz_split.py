#!/usr/bin/env python3
"""CPU-side soak of the producer / consumer form of levels 10 / 30 (lz_split.h: several emulated waves on OS threads, real atomics on
the words they share) against the oracle: stitched random inputs cut into blocks of random size, random producer / consumer counts and
buffers per producer (odd emulator seeds: three).  TEST INFRASTRUCTURE.

    python scripts/emul_fuzz_split.py <seed> <seconds>"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import util                                              # noqa: E402
from test_emulator import emul_split                     # noqa: E402
from test_random_parity import make_case                 # noqa: E402


def main():
    seed, box = int(sys.argv[1]), float(sys.argv[2])
    rng = random.Random(seed)
    t0, n, blocks, bad = time.time(), 0, 0, 0
    while time.time() - t0 < box:
        data = b"".join(make_case(rng, 300000) for _ in range(rng.randrange(1, 6)))
        level = rng.choice((10, 30))
        bs = rng.choice([19, 1000, 4096, 30000, 65536, 131072, 131073, 262144, 400000, 1 << 20])
        nprod, ncons = rng.randrange(1, 6), rng.randrange(1, 4)
        if rng.randrange(4) == 0:
            nprod, ncons = (13, 3) if level == 10 else (10, 6)
        es = rng.randrange(1, 1000)
        outs = emul_split(data, bs, level, nprod, ncons, seed=es)
        for i, o in enumerate(outs):
            blocks += 1
            if o != util.oracle_compress(data[i * bs:(i + 1) * bs], level):
                bad += 1
                path = f"/tmp/emul_fuzz_split_{seed}_{n}.bin"; open(path, "wb").write(data)
                print(f"MISMATCH seed {seed} case {n} level {level} bs {bs} {nprod}+{ncons} waves emulator seed {es} block {i} -> {path}", flush=True)
                break
        n += 1
    print(f"emul_fuzz_split: seed {seed}, {n} launches, {blocks} blocks in {time.time() - t0:.0f} s, {bad} mismatches", flush=True)


if __name__ == "__main__":
    main()
