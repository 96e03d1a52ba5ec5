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
from test_random_parity import LEVELS, make_case         # noqa: E402


def make_long_case(rng, max_size):
    target = rng.randrange(max_size // 4, max_size)
    out = bytearray()
    while len(out) < target:
        kind = rng.randrange(6)
        n = rng.choice([rng.randrange(20000, 70000), rng.randrange(60000, 140000), rng.randrange(130000, 200000), rng.randrange(1, 3000)])
        if kind == 0:
            seg = bytes([rng.randrange(256)]) * n
        elif kind == 1:
            per = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, 255, 300, 4096, 32768, 65535, 65536, 65537])
            pat = rng.randbytes(per)
            seg = (pat * (n // per + 1))[:n]
        elif kind == 2 and len(out) > 16:
            dist = min(rng.choice([8, 9, 16, 4096, 32768, 65527, 65535, 65536, 65537, 131072, 131080, len(out)]), len(out))
            start = len(out) - dist
            seg = bytes(out[start:start + min(n, dist)])
        elif kind == 3:
            seg = util.datagen(min(n, 60000), rng.choice([0.1, 0.5, 0.9, 1.0]), 0.0, rng.randrange(1 << 30))
        elif kind == 4:
            seg = rng.randbytes(rng.randrange(1, 40000))
        else:
            seg = b"".join(bytes([rng.randrange(1, 255)]) + b"\0\0\0\0" + bytes([rng.randrange(256)]) + rng.randbytes(rng.randrange(0, 12)) for _ in range(rng.randrange(1, 200)))
        out += seg
    return bytes(out[:target])


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
