#!/usr/bin/env python3
"""CPU-side soak of the Huffman stage (lz_huf.h: Lizard_writeStream + HUF_compress, through the SIMT emulator) against the oracle's
huff0 on random symbol distributions: uniform, few symbols, geometric, one symbol, power laws (the depth limiter), normal,
Fibonacci-like counts (the deepest trees), mixtures.  TEST INFRASTRUCTURE.

    python scripts/emul_fuzz_huf.py <seed> <seconds>"""
import ctypes
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import util                                              # noqa: E402


def main():
    seed, box = int(sys.argv[1]), float(sys.argv[2])
    emu = util.emulator()
    emu.emul_put_stream_huf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_uint]
    orc = util.oracle()
    rnd = random.Random(seed); rs = np.random.RandomState(seed)
    t0, n_cases, bad, huffed = time.time(), 0, 0, 0
    while time.time() - t0 < box:
        # (a stream that reaches the Huffman stage is shorter than its sub-block: Lizard_writeBlock stores the sub-block raw otherwise)
        n = rnd.choice([1024, 1025, 1500, 4001, 4002, 4003, 20000, 131056, rnd.randrange(1025, 60000), rnd.randrange(1025, 131056)])
        kind = rnd.randrange(9)
        if kind == 0: d = rs.randint(0, 256, n)
        elif kind == 1: d = rs.randint(0, rnd.randrange(1, 20), n)
        elif kind == 2: d = np.minimum(rs.geometric(rnd.uniform(0.02, 0.9), n), 255)
        elif kind == 3: d = np.full(n, rnd.randrange(256))
        elif kind == 4:
            w = np.array([2.0 ** (-i * rnd.uniform(0.3, 1.5)) for i in range(rnd.randrange(2, 256))]); w /= w.sum()
            d = rs.choice(len(w), n, p=w)
        elif kind == 5: d = np.abs(rs.normal(128, rnd.uniform(1, 60), n)).astype(np.int64) % 256
        elif kind == 6:                                  # Fibonacci-like counts: the deepest trees a stream of n symbols can have
            k = rnd.randrange(8, 26); f = [1, 1]
            while len(f) < k: f.append(f[-1] + f[-2])
            w = np.array(f, dtype=np.float64) * np.array([rnd.uniform(0.8, 1.25) for _ in f]); w /= w.sum()
            d = rs.permutation(256)[:k][rs.choice(k, n, p=w)]
        elif kind == 7:                                  # two regimes in one stream
            a = rs.randint(0, rnd.randrange(2, 256), n // 2); b = np.minimum(rs.geometric(rnd.uniform(0.05, 0.9), n - n // 2), 255)
            d = np.concatenate([a, b])
        else:                                            # a handful of symbols once each beside a dominant one
            d = np.full(n, rnd.randrange(256)); idx = rs.choice(n, rnd.randrange(1, 200), replace=False); d[idx] = rs.randint(0, 256, len(idx))
        data = np.asarray(d).astype(np.uint8).tobytes()
        cap = n + (n >> 8) + 8 + 129 + 64
        tmp = ctypes.create_string_buffer(cap)
        c = orc.lzo_huf_compress(tmp, cap, data, n) if n > 1024 else 0
        hdr = bytes([n & 255, (n >> 8) & 255, n >> 16])
        if n > 1024 and c != (1 << 64) - 1 and c > 0 and c + c // 8 + 512 < n:
            want, wh = hdr + bytes([c & 255, (c >> 8) & 255, c >> 16]) + tmp.raw[:c], 1
        else:
            want, wh = hdr + data, 0
        out = ctypes.create_string_buffer(n + 2048); h = ctypes.c_int(0)
        r = emu.emul_put_stream_huf(ctypes.create_string_buffer(data, n), n, out, ctypes.byref(h), rnd.randrange(1, 1000))
        if (out.raw[:r], h.value) != (want, wh):
            bad += 1
            path = f"/tmp/emul_fuzz_huf_{seed}_{n_cases}.bin"; open(path, "wb").write(data)
            print(f"MISMATCH seed {seed} case {n_cases} kind {kind} n {n} -> {path}", flush=True)
        huffed += wh; n_cases += 1
    print(f"emul_fuzz_huf: seed {seed}, {n_cases} streams ({huffed} Huffman-coded) in {time.time() - t0:.0f} s, {bad} mismatches", flush=True)


if __name__ == "__main__":
    main()
