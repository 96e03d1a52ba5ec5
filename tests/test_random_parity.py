"""Randomised parity: inputs stitched together from segment kinds that stress different rules of the
reference (datagen at random compressibility, noise, constant runs, short periods, copies of earlier
slices at distances around the 64 KiB window edge, text) at random ragged sizes, one block per call, every
level — against the oracle, which tests/test_oracle.py pins to the compiled reference.

CPU: a few small cases through the SIMT emulator.  GPU (-m gpu): a few hundred through the C ABI."""
import random

import pytest

import util
from test_emulator import emul_compress

LEVELS = [10, 11, 12, 13, 14, 15, 16, 17, 20, 21, 22, 30, 31, 32, 33, 34, 35, 36, 37, 38, 40, 41, 42]


def make_case(rng, max_size):
    target = rng.choice([rng.randrange(1, 64), rng.randrange(64, 5000), rng.randrange(5000, max_size), max_size,
                         131072 + rng.randrange(-40, 40)])
    target = max(1, min(target, max_size))
    out = bytearray()
    while len(out) < target:
        kind = rng.randrange(7)
        n = rng.choice([rng.randrange(1, 40), rng.randrange(40, 3000), rng.randrange(3000, 70000)])
        if kind == 0:
            seg = util.datagen(n, rng.choice([0.0, 0.1, 0.3, 0.5, 0.7, 0.9, 1.0]), 0.0, rng.randrange(1 << 30))
        elif kind == 1:
            seg = rng.randbytes(n)
        elif kind == 2:
            seg = bytes([rng.randrange(256)]) * n
        elif kind == 3:
            per = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, 255, 300])
            pat = rng.randbytes(per)
            seg = (pat * (n // per + 1))[:n]
        elif kind == 4 and len(out) > 16:                      # copy of an earlier slice: long-distance matches
            dist = rng.choice([8, 9, 15, 16, 100, 4096, 65527, 65534, 65535, 65536, 65537, 70000, len(out)])
            dist = min(dist, len(out))
            start = len(out) - dist
            seg = bytes(out[start:start + min(n, dist)])
        elif kind == 5:
            words = [b"the ", b"quick ", b"brown ", b"fox ", b"jumps ", b"over ", b"lazy ", b"dog ", b"lizard ", b"\n"]
            seg = b"".join(rng.choice(words) for _ in range(n // 5 + 1))[:n]
        else:
            seg = bytes(rng.choice(b"ab") for _ in range(min(n, 4000)))
        out += seg
    return bytes(out[:target])


def make_long_case(rng, max_size):
    """Inputs made of LONG segments (runs, periodic stretches and copies of 20 000 - 200 000 bytes between noise and generator
    data): single matches that cross sub-block-sized distances — the class of input the round-3 soak found the sweep bug with."""
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


def test_emulated_random_cases():
    rng = random.Random(20240924)
    for trial in range(48):
        data = make_case(rng, 140000)
        level = LEVELS[trial % len(LEVELS)] if trial < len(LEVELS) else rng.choice(LEVELS)
        assert emul_compress(data, level, seed=trial) == util.oracle_compress(data, level), (trial, level, len(data))


@pytest.mark.gpu
def test_gpu_random_cases():
    from lizard_amd import _lib
    L = _lib.lib()
    rng = random.Random(777)
    for trial in range(1000):
        data = make_case(rng, 400000)
        level = rng.choice(LEVELS)
        out, r = util.compress_with(L.Lizard_compress, data, level)
        assert out == util.oracle_compress(data, level), (trial, level, len(data))


@pytest.mark.gpu
def test_gpu_random_batches():
    """Same generator through the batch entry: many blocks per launch, so every kind of wave of the
    mixed-residency kernels (LDS tables, global-memory tables) takes part."""
    from lizard_amd import api
    rng = random.Random(4242)
    for trial in range(32):
        level = LEVELS[(3 * trial) % len(LEVELS)]
        bs = rng.choice([4096, 30000, 65536, 131072, 262144])
        data = b"".join(make_case(rng, 300000) for _ in range(24))
        outs = api.compress_blocks(data, bs, level)
        for i, o in enumerate(outs):
            assert o == util.oracle_compress(data[i * bs:(i + 1) * bs], level), (trial, level, bs, i)


@pytest.mark.gpu
def test_gpu_soak_time_boxed():
    """The soak as a driver-run test: stitched random inputs AND long-segment inputs (make_long_case: the generator that finds
    sweep-schedule bugs) at every level against the oracle until the time box (LIZARD_SOAK_SECONDS, default 180 s) is used up.
    One block per call from 32 host threads at once — so the calls leave in the combiner's ragged batches, blocks of different
    sizes side by side on the chip — and uniform batches through the batch entry.  The seed is taken from LIZARD_SOAK_SEED
    (default: the day number, so successive rounds walk different cases; it is printed on failure)."""
    import concurrent.futures
    import os
    import time
    from lizard_amd import _lib, api
    L = _lib.lib()
    box = float(os.environ.get("LIZARD_SOAK_SECONDS", "180"))
    seed = int(os.environ.get("LIZARD_SOAK_SEED", str(int(time.time()) // 86400)))
    rng = random.Random(seed)
    t0, n, n_long = time.time(), 0, 0

    def one(args):
        data, level = args
        out, r = util.compress_with(L.Lizard_compress, data, level)
        return out == util.oracle_compress(data, level)

    with concurrent.futures.ThreadPoolExecutor(max_workers=32) as pool:
        while time.time() - t0 < box:
            cases = []
            for k in range(48):
                long_one = k % 3 == 2
                data = make_long_case(rng, 600000) if long_one else make_case(rng, 500000)
                cases.append((data, rng.choice(LEVELS)))
                n_long += long_one
            for (data, level), ok in zip(cases, pool.map(one, cases)):
                assert ok, ("one block", seed, n, level, len(data))
                n += 1
            level = rng.choice(LEVELS); bs = rng.choice([1000, 4096, 30000, 65536, 131072, 262144, 400000])
            data = b"".join((make_long_case if rng.randrange(3) == 0 else make_case)(rng, 300000) for _ in range(40))
            for i, o in enumerate(api.compress_blocks(data, bs, level)):
                assert o == util.oracle_compress(data[i * bs:(i + 1) * bs], level), ("batch", seed, n, level, bs, i)
                n += 1
    print(f"soak: seed {seed}, {n} cases ({n_long} long-segment one-block cases) in {time.time() - t0:.0f} s, 0 mismatches")
    assert n > 100
