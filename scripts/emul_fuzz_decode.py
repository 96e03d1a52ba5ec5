#!/usr/bin/env python3
"""CPU-side soak of the block DECODER (lizard_amd/csrc/lz_unpack.h through the SIMT emulator): random stitched inputs compressed by
the oracle at a random level must come back byte for byte; damaged copies (flipped bytes, truncations) must be refused or decoded
to something — never crash, never write past the output slot (the emulator's buffer carries a guard).  TEST INFRASTRUCTURE.

    python scripts/emul_fuzz_decode.py <seed> <seconds> [max_size]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import util                                              # noqa: E402
from test_decompress import emul_decompress              # noqa: E402
from test_random_parity import LEVELS, make_case         # noqa: E402


def main():
    seed, box = int(sys.argv[1]), float(sys.argv[2])
    max_size = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
    rng = random.Random(seed)
    t0, n, bad, damaged = time.time(), 0, 0, 0
    while time.time() - t0 < box:
        data = make_case(rng, max_size)
        level = rng.choice(LEVELS)
        comp = util.oracle_compress(data, level)
        r, out = emul_decompress(comp, len(data), seed=rng.randrange(1, 9))
        if r != len(data) or out[:r] != data:
            bad += 1
            path = f"/tmp/emul_fuzz_decode_{seed}_{n}_L{level}.bin"
            open(path, "wb").write(data)
            print(f"MISMATCH seed {seed} case {n} level {level} len {len(data)} r {r} -> {path}", flush=True)
        for _ in range(3):                               # damage: the decoder may refuse or produce anything, inside its slot
            d = bytearray(comp)
            if rng.randrange(3) == 0 and len(d) > 2:
                d = d[:rng.randrange(1, len(d))]
            else:
                for _ in range(rng.randrange(1, 4)):
                    d[rng.randrange(len(d))] = rng.randrange(256)
            emul_decompress(bytes(d), len(data), seed=rng.randrange(1, 9))
            damaged += 1
        n += 1
    print(f"emul_fuzz_decode: seed {seed}, {n} round trips + {damaged} damaged blocks in {time.time() - t0:.0f} s, {bad} mismatches", flush=True)


if __name__ == "__main__":
    main()
