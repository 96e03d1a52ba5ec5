#!/usr/bin/env python3
"""Which paths of the chained priceFast parser (lz_pricefast.h, LZ_STAT marks 32-51) a set of inputs reaches, through the SIMT
emulator, with every result compared with the oracle.  TEST INFRASTRUCTURE (uses tests/ and oracle/).

    python scripts/emul_pf_coverage.py [seconds] [seed]
"""
import ctypes
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import util                                              # noqa: E402
from test_emulator import emul_compress                  # noqa: E402
from test_random_parity import make_case, make_long_case # noqa: E402

NAMES = {32: "chain stopped: stale lane in the next stretch", 33: "later stretch without a winner", 34: "winner lengths from memory",
         35: "winner in a later stretch", 36: "... a repeat-offset match", 37: "lazy position outside the round", 38: "lazy lane repeat-measured",
         39: "lazy forward count unresolved", 40: "lazy backward count unresolved", 41: "lazy step from registers", 42: "... found a match",
         43: "  ml2 <= ml", 44: "  start2 <= ip (replaces)", 45: "  start2 - ip < 3 (replaces, again)", 46: "  overlap trimmed", 47: "  second match kept",
         48: "sequence pushed from registers", 49: "second match becomes current", 50: "repeat side tested again", 51: "sequence pushed by memory steps", 52: "lazy lane stale", 54: "... hash side of repeat-measured lanes computed", 55: "rounds"}


def stats(reset=True):
    out = (ctypes.c_ulonglong * 64)()
    util.emulator().emul_stats(out, 1 if reset else 0)
    return list(out)


def main():
    box = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    stats()
    total = [0] * 64
    t0, n, bad = time.time(), 0, 0
    fixed = [util.datagen(262144, 0.5, 0.0, 7), util.datagen(200000, 0.2, 0.0, 8), util.datagen(150000, 0.8, 0.0, 9),
             (b"the quick brown fox jumps over the lazy dog. " * 4000)[:150000], bytes(100000)]
    while time.time() - t0 < box:
        data = fixed[n] if n < len(fixed) else (make_long_case if rng.randrange(3) == 0 else make_case)(rng, 300000)
        level = rng.choice([21, 21, 21, 41, 22, 42])
        eseed = rng.randrange(1, 9)
        if emul_compress(data, level, eseed) != util.oracle_compress(data, level):
            bad += 1
            path = f"/tmp/emul_pf_{seed}_{n}_L{level}_s{eseed}.bin"
            open(path, "wb").write(data)
            print(f"MISMATCH case {n} level {level} emulator seed {eseed} len {len(data)} -> {path}", flush=True)
        s = stats()
        total = [a + b for a, b in zip(total, s)]
        n += 1
    print(f"emul_pf_coverage: seed {seed}, {n} cases in {time.time() - t0:.0f} s, {bad} mismatches")
    for k in sorted(NAMES):
        print(f"  [{k}] {NAMES[k]:52s} {total[k]}")


if __name__ == "__main__":
    main()
