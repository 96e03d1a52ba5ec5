#!/usr/bin/env python3
"""CPU-side soak: the stitched random inputs of tests/test_random_parity.py through the SIMT emulator (the product's kernel bodies)
against the oracle, every level, both table forms, until the time box is used up.  TEST INFRASTRUCTURE (uses tests/ and oracle/).

    python scripts/emul_fuzz.py <seed> <seconds> [max_size] [long|default] [levels, comma-separated]

`long`: inputs made of LONG segments instead (runs, periodic stretches and copies of 20 000 - 200 000 bytes between noise and
generator data): single matches that cross sub-block-sized distances, the case the round-3 soak found a bug in.

Prints one line per mismatch (seed, case number, level, emulator seed, length; the input is written to /tmp) and a summary line."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import util                                              # noqa: E402
from test_emulator import emul_compress                  # noqa: E402
from test_random_parity import LEVELS, make_case, make_long_case         # noqa: E402


def main():
    seed, box = int(sys.argv[1]), float(sys.argv[2])
    max_size = int(sys.argv[3]) if len(sys.argv) > 3 else 300000
    gen = make_long_case if len(sys.argv) > 4 and sys.argv[4] == "long" else make_case
    levels = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else LEVELS
    rng = random.Random(seed)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < box:
        data = gen(rng, max_size)
        level = rng.choice(levels)
        eseed = rng.randrange(1, 9)
        if emul_compress(data, level, eseed) != util.oracle_compress(data, level):
            bad += 1
            path = f"/tmp/emul_fuzz_{seed}_{n}_L{level}_s{eseed}.bin"
            open(path, "wb").write(data)
            print(f"MISMATCH seed {seed} case {n} level {level} emulator seed {eseed} len {len(data)} -> {path}", flush=True)
        n += 1
    print(f"emul_fuzz: seed {seed}, {n} cases in {time.time() - t0:.0f} s, {bad} mismatches", flush=True)


if __name__ == "__main__":
    main()
