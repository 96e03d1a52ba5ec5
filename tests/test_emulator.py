"""CPU: runs the PRODUCT kernel bodies (lizard_amd/csrc/lz_block.h ...) lane-for-lane on the 64-lane
SIMT emulator (tests/emul) and checks them bit-exact against the oracle and the golden vectors.
This is how kernel logic is debugged without a GPU; the GPU parity tests (-m gpu) repeat the same
comparisons through the C ABI on real hardware."""
import ctypes
import json
import os
import random

import pytest

import util

with open(os.path.join(util.GOLDEN_DIR, "reference_vectors.json")) as f:
    GOLDEN = json.load(f)

EMUL_LEVELS = [10, 11, 20, 21, 22, 30, 31, 40, 41, 42]


def emul_compress(data, level, seed=1):
    n = len(data)
    bound = util.oracle().lzo_compress_bound(n)
    dst = ctypes.create_string_buffer(bound + 64)
    src = ctypes.create_string_buffer(bytes(data), n) if n else ctypes.create_string_buffer(1)
    r = util.emulator().emul_compress_block(src, n, dst, level, seed)
    assert r >= 0, "level not wired into the emulator"
    return dst.raw[:r]


@pytest.mark.parametrize("level", EMUL_LEVELS)
def test_emulated_kernel_vs_golden(level):
    for name, data in util.corpus():
        g = GOLDEN["cases"][name]["out"][str(level)]
        out = emul_compress(data, level, seed=len(name) + level)
        assert len(out) == g["size"] and util.sha(out) == g["sha256"], (name, level)


@pytest.mark.parametrize("level", [10, 30, 21, 11, 22, 13, 36, 20, 40, 12])
def test_emulated_kernel_long_range(level):
    """Window-edge / position-wrap adversaries and multi-MiB blocks (tests/util.corpus_long): all of them at the BASELINE levels,
    every third one (a different third per level) at one level of each other kernel family."""
    cases = util.corpus_long()
    if level not in (10, 30, 21, 20):
        cases = cases[level % 3::3] + [c for c in cases if c[0] == "longoff_rule"]
    for name, data in cases:
        g = GOLDEN["cases"][name]["out"][str(level)]
        out = emul_compress(data, level, seed=len(name))
        assert len(out) == g["size"] and util.sha(out) == g["sha256"], (name, level)


@pytest.mark.parametrize("level", [13, 16, 17, 35, 12, 32, 33, 34, 14, 15, 36])
def test_emulated_hashchain_vs_oracle(level):
    """hashChain kernel (lz_hashchain.h): both hash lengths, searchNum 2/8/16/256, with and without Huffman; the noChain levels
    12 / 32 / 33 (one candidate per search, hashLog 18 / 14 / 18, lizard_parser_nochain.h) through the same kernels.
    The oracle is pinned to the compiled reference for these levels by tests/test_oracle.py."""
    for name, data in util.corpus(small=True):
        if level == 17 and name in ("alpha2", "alpha4"):
            data = data[:50000]          # 256-candidate searches on 2/4-letter noise: slow under emulation
        assert emul_compress(data, level, seed=len(name) + level) == util.oracle_compress(data, level), (name, level)


def test_emulated_hashchain_slot_reuse():
    """The per-wave global slot of the hashChain levels (bins, links, saved head tables) is never cleared between
    blocks: nothing an earlier block left there may matter.  Unrelated blocks through one slot, among them one of
    more than 2^18 positions (two segments: the head tables travel through the slot) and runs (the scalar replay)."""
    cases = [util.datagen(70000 + 977 * i, 0.5, 0.0, 100 + i) for i in range(3)]
    cases.append(b"\0" * 5000 + util.datagen(3000, 0.3, 0.0, 7) + b"ab" * 3000 + b"\0" * 700)
    cases.append(util.datagen(300000, 0.6, 0.0, 9)[:150000] + util.datagen(150000, 0.6, 0.0, 9))   # repeats across the segment border
    cases.append(util.datagen(64, 0.5, 0.0, 3))
    for i, data in enumerate(cases):
        for level in (14, 16):
            assert emul_compress(data, level, seed=i) == util.oracle_compress(data, level), (i, level)


def test_emulated_global_table_forms():
    """The tables that live in global memory carry filters that must never change a result: 8 check bits beside the
    position in the priceFast u32 slots (levels 21/41 global-table waves, 22/42), and the occupancy summary of the
    2^18-slot tables (levels 11/31, 22/42; emulator seeds with bit 1 clear switch it on, odd seeds pick the global form)."""
    cases = [util.datagen(90000, 0.5, 0.0, 21), b"abcabcabd" * 400 + util.datagen(5000, 0.3, 0.0, 2) + b"\0" * 3000,
             util.datagen(280000, 0.7, 0.0, 22)]
    for level in (11, 31, 21, 41, 22, 42):
        for seed in (1, 3):
            for i, data in enumerate(cases):
                if level in (41, 42, 31) and i == 2:
                    continue                                        # (time: the Huffman levels skip the largest case)
                assert emul_compress(data, level, seed) == util.oracle_compress(data, level), (level, seed, i)


def test_emulated_kernel_schedule_independent():
    """Output must not depend on the order lanes run between cross-lane ops (LDS store races)."""
    data = dict(util.corpus(small=True))["gen262144_p0.5"]
    want = util.oracle_compress(data, 10)
    for seed in (1, 2, 3, 12345):
        assert emul_compress(data, 10, seed) == want


def test_emulated_huffman_stream_vs_oracle():
    """lz_put_stream_huf (Lizard_writeStream + HUF_compress) on synthetic symbol distributions that
    exercise RLE, 'not compressible', the depth limiter, FSE and raw weight headers."""
    import random
    import numpy as np
    emu = util.emulator()
    emu.emul_put_stream_huf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_uint]
    orc = util.oracle()
    rnd = random.Random(11); rs = np.random.RandomState(3)
    huffed = 0
    for trial in range(60):
        n = rnd.choice([1024, 1025, 1500, 4001, 4002, 4003, 20000, 131056, rnd.randrange(1025, 60000)])
        kind = trial % 6
        if kind == 0: d = rs.randint(0, 256, n)
        elif kind == 1: d = rs.randint(0, rnd.randrange(1, 20), n)
        elif kind == 2: d = np.minimum(rs.geometric(rnd.uniform(0.02, 0.9), n), 255)
        elif kind == 3: d = np.full(n, 9)
        elif kind == 4:
            w = np.array([2.0 ** (-i * rnd.uniform(0.3, 1.5)) for i in range(rnd.randrange(2, 256))]); w /= w.sum()
            d = rs.choice(len(w), n, p=w)
        else: d = np.abs(rs.normal(128, rnd.uniform(1, 60), n)).astype(np.int64) % 256
        data = d.astype(np.uint8).tobytes()
        cap = n + (n >> 8) + 8 + 129 + 64
        tmp = ctypes.create_string_buffer(cap)
        c = orc.lzo_huf_compress(tmp, cap, data, n) if n > 1024 else 0
        hdr = bytes([n & 255, (n >> 8) & 255, n >> 16])
        if n > 1024 and c != (1 << 64) - 1 and c > 0 and c + c // 8 + 512 < n:
            want, wh = hdr + bytes([c & 255, (c >> 8) & 255, c >> 16]) + tmp.raw[:c], 1
        else:
            want, wh = hdr + data, 0
        out = ctypes.create_string_buffer(n + 2048); h = ctypes.c_int(0)
        r = emu.emul_put_stream_huf(ctypes.create_string_buffer(data, n), n, out, ctypes.byref(h), trial + 1)
        assert (out.raw[:r], h.value) == (want, wh), (trial, kind, n)
        huffed += wh
    assert huffed > 10


def test_emulated_huffman_streams_whose_fate_three_bytes_decide():
    """Round 6: lz_put_stream_huf accepts or rejects a stream from the histogram alone (total code bits -> the four bitstreams' bytes
    within three) and sums the segments' code lengths only when those three bytes decide.  Streams built to sit on the acceptance
    rules of huf_compress.c:570 / lizard_compress.c:157: m incompressible bytes in front of 16-symbol bytes, m bisected to the
    flip, then every m around it.  The exact pass (LZ_STAT 60) must be reached, on both sides of the flip, oracle-equal."""
    import numpy as np
    E = util.emulator()
    E.emul_put_stream_huf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_uint]
    orc = util.oracle()
    stats = (ctypes.c_ulonglong * 64)()

    def want_of(data):
        n = len(data); cap = n + (n >> 8) + 8 + 129 + 64
        tmp = ctypes.create_string_buffer(cap)
        c = orc.lzo_huf_compress(tmp, cap, data, n)
        hdr = bytes([n & 255, (n >> 8) & 255, n >> 16])
        if c != (1 << 64) - 1 and c > 0 and c + c // 8 + 512 < n:
            return hdr + bytes([c & 255, (c >> 8) & 255, c >> 16]) + tmp.raw[:c], 1
        return hdr + data, 0

    reached = {0: 0, 1: 0}
    for n, seed in ((6000, 1), (20001, 2), (50003, 3), (131000, 4)):
        rs = np.random.RandomState(seed)
        hi_part, lo_part = rs.randint(0, 256, n).astype(np.uint8), rs.randint(0, 16, n).astype(np.uint8)
        make = lambda m: np.concatenate([hi_part[:m], lo_part[m:]]).tobytes()
        a, b = 0, n                                            # accepted at a, not at b
        assert want_of(make(a))[1] == 1 and want_of(make(b))[1] == 0
        while b - a > 1:
            m = (a + b) // 2
            if want_of(make(m))[1]: a = m
            else: b = m
        for m in range(max(0, a - 12), min(n, a + 13)):
            data = make(m)
            want, wh = want_of(data)
            E.emul_stats(stats, 1)
            out = ctypes.create_string_buffer(n + 2048); h = ctypes.c_int(0)
            r = E.emul_put_stream_huf(ctypes.create_string_buffer(data, n), n, out, ctypes.byref(h), m + 1)
            assert (out.raw[:r], h.value) == (want, wh), (n, m)
            E.emul_stats(stats, 1)
            if stats[60]:
                reached[wh] += 1
    assert reached[0] > 0 and reached[1] > 0, reached


def test_emulated_wave_helpers():
    """lz_count_fwd / lz_count_back / lz_count_both / lz_copy / scans against scalar definitions on buffers with
    planted repeats (short, around the 8-, 64- and 512-byte step sizes of the helpers, and long)."""
    import random
    import numpy as np
    emu = util.emulator()
    rnd = random.Random(5)
    n = 20000
    buf = bytearray(util.datagen(n, 0.3, 0.0, 77))
    triples = []
    for t in range(160):
        ln = rnd.choice([0, 1, 7, 8, 9, 63, 64, 65, 100, 511, 512, 513, 520, 1023, 1500, 2600])
        M = rnd.randrange(70, 6000)
        P = M + rnd.choice([1, 2, 8, 64, 700]) + rnd.randrange(0, 6000)
        if P + ln + 40 >= n:
            continue
        back = rnd.choice([0, 1, 7, 8, 63, 64, 65, 70])
        back = min(back, M - 1)
        buf[P - back:P + ln] = buf[M - back:M + ln]                    # plant the repeat (overlap is fine)
        lim = rnd.choice([P + ln // 2 + 1, P + ln, P + ln + 30, n - 16])
        triples += [P, M, min(max(lim, P), n - 16)]
    arr = (ctypes.c_uint * len(triples))(*triples)
    emu.emul_check_helpers.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint]
    for seed in (1, 2, 3):
        assert emu.emul_check_helpers(bytes(buf), n, arr, len(triples) // 3, seed) == 0


def emul_split(data, bs, level, nprod, ncons, seed=1):
    """Levels 10 / 30 in the producer / consumer form (lizard_amd/csrc/lz_split.h) on nprod + ncons emulated waves, each on an OS
    thread of its own, sharing mailboxes and free masks like the waves of one workgroup; returns the compressed blocks."""
    emu = util.emulator()
    emu.emul_compress_split.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint]
    nb = (len(data) + bs - 1) // bs
    last = len(data) - (nb - 1) * bs
    stride = util.oracle().lzo_compress_bound(bs) + 64
    dst = ctypes.create_string_buffer(nb * stride)
    sizes = (ctypes.c_uint * nb)()
    src = ctypes.create_string_buffer(bytes(data), len(data))
    assert emu.emul_compress_split(src, nb, bs, last, dst, stride, sizes, level, nprod, ncons, seed) == 0
    return [dst.raw[i * stride:i * stride + sizes[i]] for i in range(nb)]


@pytest.mark.parametrize("level", [10, 30])
def test_emulated_split_producers_and_consumers(level):
    """Parse and container on different waves: every block must come out as the one-wave form (= the oracle) writes it —
    many blocks per producer, sub-block chains (blocks of 1 to 9 sub-blocks), ragged last blocks, tiny blocks, more consumers
    than producers and the reverse, the kernel's own wave counts (13 + 3, 11 + 5)."""
    data = util.datagen(1500000, 0.5, 0.0, 3) + bytes(70000) + util.datagen(300000, 0.2, 0.0, 9) + b"abcd" * 5000 + util.datagen(7, 0.5, 0.0, 1)
    shapes = [(65536, 3, 1), (262144, 2, 2), (300000, 4, 2), (40000, 5, 3), (1 << 20, 1, 1), (19, 2, 1), (131072, 1, 3)]
    shapes.append((30000, 13, 3) if level == 10 else (30000, 11, 5))
    for bs, nprod, ncons in shapes:
        part = data if bs > 100 else data[:1000]
        outs = emul_split(part, bs, level, nprod, ncons, seed=bs + nprod)
        for i, o in enumerate(outs):
            assert o == util.oracle_compress(part[i * bs:(i + 1) * bs], level), (level, bs, nprod, ncons, i)


@pytest.mark.parametrize("level", [10, 30])
def test_emulated_split_ragged_batch_with_producer_cap(level):
    """The batches the one-block entry points' combiner launches (lizard_amd/csrc/lizard_pipeline_host.c): blocks of DIFFERENT
    sizes at a common stride (LzBatch::srcSizes) and fewer claiming producers than the workgroup has (LzBatch::activeWaves)."""
    emu = util.emulator()
    emu.emul_compress_split_ragged.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint]
    rnd = random.Random(level)
    lens = [1, 19, 20, 21, 300000, 131072, 131073, 5, 262144, 70000, 100, 262143, 40, 131071, 200000, 64]
    stride = 300032
    blocks = [util.datagen(n, rnd.choice((0.2, 0.5, 0.9)), 0.0, n + level) for n in lens]
    src = ctypes.create_string_buffer(stride * len(lens))
    for i, b in enumerate(blocks):
        ctypes.memmove(ctypes.addressof(src) + i * stride, b, len(b))
    sizes_in = (ctypes.c_uint * len(lens))(*lens)
    dstride = util.oracle().lzo_compress_bound(stride) + 64
    for nprod, ncons, active, seed in ((4, 2, 1, 1), (5, 3, 2, 2), (3, 1, 3, 3), (13, 3, 1, 4) if level == 10 else (10, 6, 2, 4)):
        dst = ctypes.create_string_buffer(len(lens) * dstride)
        sizes = (ctypes.c_uint * len(lens))()
        assert emu.emul_compress_split_ragged(src, len(lens), stride, stride, sizes_in, dst, dstride, sizes, level, nprod, ncons, active, seed) == 0
        for i, b in enumerate(blocks):
            assert dst.raw[i * dstride:i * dstride + sizes[i]] == util.oracle_compress(b, level), (level, nprod, ncons, active, i, len(b))


def _beyond_width_case(gap, seed):
    """X, a run of zeros, X again `gap` + 100 positions after its first copy, a tail: slots written during the first X survive
    the run (a run inserts next to nothing) and are `gap` + 100 old — 100 modulo the table's position width — when the second X
    probes them with the same bytes and the same check bits."""
    x = util.datagen(600000, 0.5, 0.0, seed)
    return x + bytes(gap + 100 - len(x)) + x + util.datagen(300000, 0.4, 0.0, seed + 1)


def test_emulated_fast18_block_beyond_22_bit_positions():
    """Levels 11/31 keep positions modulo 2^22 and re-stamp dead slots every 2^20 positions (LzTabWide): a block of more than
    4 MiB must come out as the reference writes it, with and without the occupancy summary."""
    data = _beyond_width_case(1 << 22, 5)
    want = util.oracle_compress(data, 11)
    for seed in (1, 3):
        assert emul_compress(data, 11, seed) == want, seed


@pytest.mark.parametrize("level,seeds", [(21, (1, 4)), (22, (1, 3)), (20, (1,))])
def test_emulated_pricefast_block_beyond_24_bit_positions(level, seeds):
    """Levels 21/22/41/42 (and 20/40, whose fastBig parser uses the same global-memory slots) keep positions modulo 2^24 in their u32
    slots and re-stamp dead slots every 2^22 positions: a block of more than 16 MiB, both residences of the table (LDS / global
    memory; level 22 with and without the occupancy summary)."""
    data = _beyond_width_case(1 << 24, 9)
    want = util.oracle_compress(data, level)
    for seed in seeds:
        assert emul_compress(data, level, seed) == want, (level, seed)


def test_emulated_hashchain_block_above_4mib():
    """hashChain keeps full positions (bins are segment-relative, heads and links absolute / distances): a 6 MiB block."""
    data = util.datagen(6 << 20, 0.5, 0.0, 3)
    assert emul_compress(data, 13, 1) == util.oracle_compress(data, 13)


def test_emulated_hashchain_searches_decided_ahead_of_the_parse():
    """Round 4, lz_hashchain.h: (1) a wider search (hashchain.h:212-214, :263-265) at a position whose hit bit is clear never
    starts — at every hashChain level; (2) at levels 16/17/37/38 the first search of a position comes out of the table the hit pass
    fills (4 candidates deep, 16 bytes each), the rest — longer matches, longer chains — stays with the parse; levels 13-15 keep
    the plain hit pass.  Counted by the emulator's LZ_STAT marks, outputs equal to the oracle's."""
    E = util.emulator()
    out = (ctypes.c_ulonglong * 64)()
    data = [util.datagen(200000, 0.5, 0.0, 5), util.datagen(140000, 0.25, 0.0, 6), (b"abcdefgh" * 3000 + bytes(range(256)) * 40) * 2]
    for level in (13, 15, 16, 17, 37):
        E.emul_stats(out, 1)
        for i, d in enumerate(data):
            assert emul_compress(d, level, 1 + i) == util.oracle_compress(d, level), (level, i)
        E.emul_stats(out, 1)
        from_table, searched, w2, w2_run, w3, w3_run = out[19], out[20], out[21], out[22], out[23], out[24]
        assert w2_run < w2 and w3_run <= w3 and w2_run > 0, (level, w2, w2_run)          # some wider searches skipped, some run
        if level in (16, 17, 37):
            assert from_table > searched > 0, (level, from_table, searched)                # most first searches read, some still searched
        else:
            assert from_table == 0 and searched > 0, (level, from_table)


def test_emulated_hashchain_prepass_edges():
    """The first-search table of levels 16/17/37/38 (lz_hc_hits) compares 16 bytes per candidate: inputs whose matches end exactly
    at, just before and just after those 16 bytes, at every distance from the end of a sub-block and of the block (where the room
    of a match shrinks to nothing and the second 8-byte load is clamped), tiny blocks, and candidates more than four deep."""
    rnd = random.Random(16)
    phrase = bytes(rnd.randrange(256) for _ in range(40))
    cases = []
    for n in list(range(8, 72)) + [100, 131072 - 30, 131072 - 17, 131072 - 1, 131072, 131072 + 1, 131072 + 16, 131072 + 25, 262144 - 9, 262144]:
        body = bytearray()
        while len(body) < n:
            k = rnd.choice((14, 15, 16, 17, 18, 23, 24, 25, 40))                 # shared prefix length of the next repeat
            body += phrase[:k] + bytes([rnd.randrange(256)])
            if rnd.randrange(4) == 0:
                body += bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 30)))
        cases.append(bytes(body[:n]))
    deep = (b"abcdefgh" + bytes(8)) * 6 + b"".join(b"abcd" + bytes([i]) * 5 for i in range(40)) + b"abcdefghijklmnopqrstuvwxyz" * 3   # many candidates per 4-byte hash
    cases += [deep, deep * 9]
    for i, data in enumerate(cases):
        for level in (16, 17, 37):
            if len(data) > 100000 and level != 16 + (i % 2):
                continue
            assert emul_compress(data, level, 1 + i % 4) == util.oracle_compress(data, level), (i, len(data), level)


def test_emulated_pricefast_chained_edges():
    """Chained priceFast rounds: matches that end exactly at, just before and just after the 24 fetched bytes (resolved / unresolved
    lengths), sequences that end at lanes 62-65 of their round (the chain must stop at the round's edge), repeats at distances
    below MIN_OFFSET, the same offset again and again (repeat-offset matches inside a chain), and every distance from the end of a
    sub-block — all three table forms."""
    rnd = random.Random(21)
    phrase = bytes(rnd.randrange(256) for _ in range(64))
    cases = []
    for n in (40, 64, 65, 127, 128, 129, 4000, 131072 - 21, 131072 - 20, 131072 - 19, 131072 - 1, 131072, 131072 + 1, 131072 + 23, 150000):
        body = bytearray()
        while len(body) < n:
            kind = rnd.randrange(6)
            if kind == 0:   body += phrase[:rnd.choice((22, 23, 24, 25, 26, 31, 32, 33))] + bytes([rnd.randrange(256)])
            elif kind == 1: body += bytes(rnd.randrange(256) for _ in range(rnd.choice((1, 2, 3, 5, 37, 57, 58, 59, 60, 61))))     # push sequence ends around lane 63
            elif kind == 2: body += bytes([rnd.randrange(256)]) * rnd.choice((3, 7, 8, 9, 30))                                    # runs: distances below MIN_OFFSET
            elif kind == 3: body += (phrase[5:12] + bytes([len(body) & 255])) * rnd.choice((2, 3, 9))                              # one offset again and again
            elif kind == 4: body += phrase[rnd.randrange(30):][:rnd.randrange(4, 20)]
            else:           body += body[-rnd.randrange(1, 200):][:rnd.randrange(4, 70)] if body else b"x"
        cases.append(bytes(body[:n]))
    for i, data in enumerate(cases):
        for level, seed in ((21, 1), (21, 2), (21, 4), (41, 1 + i % 4), (22, 1 + i % 2)):
            assert emul_compress(data, level, seed) == util.oracle_compress(data, level), (i, len(data), level, seed)


def test_emulated_pricefast_chained_paths_are_reached():
    """Levels 21 / 41 / 22 take several sequences out of one round and run the lazy step (pricefast.h:184-228) from the lanes'
    registers (lz_pricefast.h, "several sequences out of one round").  The emulator counts the parser's LZ_STAT marks: every exit
    of the arbitration, every fall-back to the memory-based steps and the stale-lane stop must be reached by this set of inputs —
    with every output equal to the oracle's — so that the CPU suite really covers them (scripts/emul_pf_coverage.py prints the
    same counters for a soak)."""
    E = util.emulator()
    out = (ctypes.c_ulonglong * 64)()
    E.emul_stats(out, 1)
    rnd = random.Random(4)
    inputs = [util.datagen(150000, 0.5, 0.0, 7), util.datagen(90000, 0.2, 0.0, 8), util.datagen(90000, 0.8, 0.0, 9),
              (b"the quick brown fox jumps over the lazy dog. " * 2000)[:70000],
              bytes(rnd.choice(b"ab") for _ in range(30000)) + (b"abcabcabd" * 3000) + bytes(20000)]
    for i, data in enumerate(inputs):
        for level, seed in ((21, 1), (21, 2), (21, 4), (41, 2), (22, 1 + i % 2)):       # global-memory table, 18 + 6 bit LDS table, u32 LDS table
            assert emul_compress(data, level, seed) == util.oracle_compress(data, level), (i, level, seed)
    E.emul_stats(out, 1)
    names = {32: "stale lane stops the chain", 33: "later stretch without a winner", 34: "winner lengths from memory", 35: "winner in a later stretch",
             37: "lazy position outside the round", 38: "lazy lane measured against its repeat candidate", 39: "lazy forward count unresolved",
             40: "lazy backward count unresolved", 41: "lazy step from registers", 42: "lazy step finds a match", 43: "ml2 <= ml", 44: "start2 <= ip",
             45: "start2 - ip < 3", 46: "overlap trimmed", 47: "second match kept", 48: "sequence pushed from registers", 49: "second match becomes current",
             50: "repeat side tested again", 51: "sequence pushed by the memory steps", 52: "lazy lane stale", 54: "hash side computed late", 55: "rounds"}
    missing = [v for k, v in names.items() if out[k] == 0]
    assert not missing, missing
    assert out[48] > 2 * out[51] and out[55] < out[48] + out[51]            # most sequences come from registers; more than one per round


def test_emulated_fastbig_long_offset_rule_paths_are_reached():
    """Levels 20 / 40 (lz_fastbig.h): a candidate 65 536 or more back counts only when forward count + backward extension reach 16
    (lizard_parser_fastbig.h:92-98; no backward extension in the post-match probe, :143-146).  The lanes decide that from the bytes
    they fetched, and the undecided ones in front of the first accepting lane are measured one by one.  tests/util.longoff_rule_case
    puts matches on both sides of the rule behind a 66 000-byte run; the emulator's LZ_STAT marks must show every path taken, with
    the output the compiled reference's (golden vector)."""
    E = util.emulator()
    out = (ctypes.c_ulonglong * 64)()
    E.emul_stats(out, 1)
    data = util.longoff_rule_case()
    for level, seed in ((20, 20), (40, 40), (20, 22), (40, 42)):       # seed bit 1: with / without the slot codes in LDS
        g = GOLDEN["cases"]["longoff_rule"]["out"][str(level)]
        got = emul_compress(data, level, seed=seed)
        assert len(got) == g["size"] and util.sha(got) == g["sha256"], (level, seed)
        assert got == util.oracle_compress(data, level)
    E.emul_stats(out, 1)
    names = {1: "long-offset lane accepted from its fetched bytes", 2: "long-offset lane refused from its fetched bytes",
             3: "undecided lane measured and accepted", 4: "undecided lane measured and refused",
             5: "post-match probe wins behind a long offset", 6: "long-offset winner with a backward extension",
             7: "sequence pushed from inside a round"}
    missing = [v for k, v in names.items() if out[k] == 0]
    assert not missing, (missing, [int(out[k]) for k in names])


def test_emulated_chained_rounds_on_generator_data():
    """lz_parse_fast takes several sequences out of one round when the lanes behind a short match already hold the reference's
    next steps (and gives up when one of them read a put of a position inside the match): the synthetic-workload generator's
    short matches are where that happens (30 % of the sequences of a bench block; ~4 % of the attempts meet such a put).  Both
    table forms (the emulator seed picks LDS or global memory), with and without the Huffman stage, block ends of every kind."""
    import random
    from tools import datagen
    rng = random.Random(11)
    for case in range(36):
        P = rng.choice((0.1, 0.3, 0.5, 0.7, 0.9, 0.97))
        size = rng.choice((300, 1000, 5000, 20000, 70000, 140000, 262144))
        buf = ctypes.create_string_buffer(size)
        datagen.datagen_host(buf, size, P, rng.choice((0.0, 0.2)), 100 + case)
        data = buf.raw
        for level in (10, 30, 11, 31):
            assert emul_compress(data, level, seed=case + 1) == util.oracle_compress(data, level), (case, P, size, level)


def _long_match_case(prefix, run, seed):
    """noise | one long run of zeros (a single match of `run` - 8 bytes: ip jumps across several sweep intervals of the 17-bit LDS
    table at once) | short records "k 0 0 0 0 k'": their first four bytes equal those inside the run, their five-byte hash names a
    slot nobody wrote.  A slot stamped "dead" by the sweep before the jump must still read as dead behind it."""
    import random
    rng = random.Random(seed)
    tail = b"".join(bytes([1 + i]) + b"\0\0\0\0" + bytes([100 + i]) + rng.randbytes(10) for i in range(50))
    return rng.randbytes(prefix) + b"\0" * run + tail


@pytest.mark.parametrize("prefix,run", [(40000, 80000), (33000, 66000), (1000, 120000), (70000, 40000), (40000, 200000)])
def test_emulated_long_match_crosses_sweep_intervals(prefix, run):
    # found by a 110 s GPU soak in round 3 (seed 4002, case 16282): levels 10/30 accepted a candidate out of a slot whose "65536 old"
    # stamp had wrapped to a young age behind a 70 000-byte match — the sweeps due inside the match are now made up at the match
    data = _long_match_case(prefix, run, prefix + run)
    for level in (10, 30, 11, 21, 22):
        want = util.oracle_compress(data, level)
        for seed in (1, 2):
            assert emul_compress(data, level, seed) == want, (level, seed)
