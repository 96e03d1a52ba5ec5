"""CPU: the decode / frame-decode / hash half of the drop-in ABI (include/lizard_amd.h parts 1b-1d) — host code, no GPU needed.

Lizard_decompress_safe* (lizard_amd/csrc/lizard_decode_host.c), LizardF_decompress / getFrameInfo (lizard_frame_host.c) and
Lizard_XXH32/64 (lizard_xxhash.c) are checked against the compiled reference (oracle/_ref/liblizard_ref.so, when it
travelled) and against the oracle's blocks (always): valid blocks decode to the input with the reference's return values
(partial decoding included), dictionaries as prefix / external / split, frames under random segmentation of input and
output, damaged input is refused or decoded exactly like the reference and never touches a byte outside its buffers."""
import ctypes
import os
import random
import struct

import pytest

import util

C = ctypes


@pytest.fixture(scope="module")
def lib():
    from lizard_amd import _lib
    _lib.build()
    L = C.CDLL(_lib.LIB_PATH)
    sig4 = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.Lizard_decompress_safe.argtypes = sig4
    L.Lizard_decompress_safe_partial.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.Lizard_decompress_safe_usingDict.argtypes = sig4 + [C.c_void_p, C.c_int]
    L.Lizard_decompress_safe_forceExtDict.argtypes = sig4 + [C.c_void_p, C.c_int]
    L.Lizard_decompress_safe_continue.argtypes = [C.c_void_p] + sig4
    L.Lizard_setStreamDecode.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.Lizard_createStreamDecode.restype = C.c_void_p
    L.Lizard_freeStreamDecode.argtypes = [C.c_void_p]
    for n in ("LizardF_createDecompressionContext", "LizardF_freeDecompressionContext", "LizardF_decompress", "LizardF_getFrameInfo",
              "LizardF_compressFrameBound", "LizardF_compressFrame", "LizardF_isError"):
        getattr(L, n).restype = C.c_size_t
    L.LizardF_isError.restype = C.c_uint
    L.LizardF_isError.argtypes = [C.c_size_t]
    L.LizardF_getErrorName.restype = C.c_char_p
    L.LizardF_getErrorName.argtypes = [C.c_size_t]
    L.LizardF_createDecompressionContext.argtypes = [C.c_void_p, C.c_uint]
    L.LizardF_freeDecompressionContext.argtypes = [C.c_void_p]
    L.LizardF_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
    L.LizardF_getFrameInfo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    L.Lizard_XXH32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]; L.Lizard_XXH32.restype = C.c_uint
    L.Lizard_XXH64.argtypes = [C.c_void_p, C.c_size_t, C.c_ulonglong]; L.Lizard_XXH64.restype = C.c_ulonglong
    L.Lizard_XXH32_reset.argtypes = [C.c_void_p, C.c_uint]
    L.Lizard_XXH32_update.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.Lizard_XXH32_digest.argtypes = [C.c_void_p]; L.Lizard_XXH32_digest.restype = C.c_uint
    L.Lizard_XXH64_reset.argtypes = [C.c_void_p, C.c_ulonglong]
    L.Lizard_XXH64_update.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.Lizard_XXH64_digest.argtypes = [C.c_void_p]; L.Lizard_XXH64_digest.restype = C.c_ulonglong
    return L


_stock = None


def stock():
    """The UNMODIFIED reference library (oracle/_ref/liblizard_ref.so): compressor at every level, decoder, frames."""
    global _stock
    if _stock is None:
        util.reference()                                     # builds oracle/_ref when the checkout is present
        path = os.path.join(util.REF_DIR, "liblizard_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.Lizard_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        R.Lizard_compressBound.argtypes = [C.c_int]
        sig4 = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        R.Lizard_decompress_safe.argtypes = sig4
        R.Lizard_decompress_safe_partial.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        R.Lizard_decompress_safe_usingDict.argtypes = sig4 + [C.c_void_p, C.c_int]
        R.Lizard_createStream.restype = C.c_void_p; R.Lizard_createStream.argtypes = [C.c_int]
        R.Lizard_freeStream.argtypes = [C.c_void_p]
        R.Lizard_loadDict.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        R.Lizard_compress_continue.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        util._reference_frame_fn(R)
        _stock = R
    return _stock


def ref_compress(data, level):
    R = stock()
    cap = R.Lizard_compressBound(len(data))
    dst = C.create_string_buffer(cap + 8)
    n = R.Lizard_compress(data, dst, len(data), cap, level)
    assert n > 0
    return dst.raw[:n]


GUARD = 64


def guarded(n, fill=0xA5):
    """n usable bytes with GUARD canary bytes on both sides"""
    buf = (C.c_ubyte * (n + 2 * GUARD))()
    C.memset(buf, fill, n + 2 * GUARD)
    return buf


def guard_ok(buf, n, fill=0xA5):
    raw = bytes(buf)
    return raw[:GUARD] == bytes([fill]) * GUARD and raw[GUARD + n:] == bytes([fill]) * GUARD


def my_decode(L, comp, cap):
    out = guarded(cap)
    src = guarded(len(comp))
    C.memmove(C.addressof(src) + GUARD, comp, len(comp))
    r = L.Lizard_decompress_safe(C.addressof(src) + GUARD, C.addressof(out) + GUARD, len(comp), cap)
    assert guard_ok(out, cap), "decoder wrote outside its output buffer"
    return r, bytes(out)[GUARD:GUARD + max(r, 0)]


SMALL = [(n, d) for n, d in util.corpus(small=True)] + [("text", dict(util.corpus())["text"]), ("alpha4", dict(util.corpus())["alpha4"]),
                                                        ("zeros300k", bytes(300000)), ("random256k", dict(util.corpus())["random256k"])]
GPU_LEVELS = [10, 11, 12, 13, 14, 15, 16, 17, 20, 21, 22, 30, 31, 32, 33, 34, 35, 36, 37, 38, 40, 41, 42]


def test_decodes_oracle_blocks(lib):
    """every GPU level's container and codewords, from the oracle restatement (always available)"""
    for name, data in SMALL:
        for level in GPU_LEVELS:
            comp = util.oracle_compress(data, level)
            r, out = my_decode(lib, comp, len(data))
            assert r == len(data) and out == data, (name, level, r)
            if len(data) > 40:                               # one byte short: must be refused, not truncated
                r2, _ = my_decode(lib, comp, len(data) - 1)
                assert r2 < 0, (name, level)
                r3, _ = my_decode(lib, comp[:-1], len(data))
                assert r3 < 0, (name, level)
                r4, _ = my_decode(lib, comp + b"\x00", len(data))
                assert r4 < 0, (name, level)


def test_golden_blocks_decode(lib):
    src = open(os.path.join(util.GOLDEN_DIR, "p50_4k_seed42.bin"), "rb").read()
    for lv in (10, 21, 30):
        comp = open(os.path.join(util.GOLDEN_DIR, f"p50_4k_seed42.L{lv}.liz_block"), "rb").read()
        r, out = my_decode(lib, comp, len(src))
        assert r == len(src) and out == src, lv


def test_every_reference_level_decodes(lib):
    """blocks of ALL 40 levels of the reference compressor (the decoder is level-agnostic: fastLZ4 / LIZv1, huff0 on every
    stream) against the reference decoder's results"""
    R = stock()
    if R is None:
        pytest.skip("oracle/_ref/liblizard_ref.so not present")
    rnd = random.Random(5)
    inputs = [util.datagen(150000, 0.5, 0.0, 3), util.datagen(300000, 0.2, 0.0, 4)[:270000], (b"the quick brown fox. " * 9000)[:140000],
              bytes(rnd.choice(b"abcd") for _ in range(70000)), util.datagen(5 << 20, 0.6, 0.0, 9)]
    for i, data in enumerate(inputs):
        levels = range(10, 50) if len(data) < (1 << 20) else (10, 17, 21, 24, 30, 41, 44)
        for level in levels:
            if level in (19, 29, 39, 49) and len(data) > 160000:
                continue                                     # the optimal parsers take seconds per MiB
            comp = ref_compress(data, level)
            r, out = my_decode(lib, comp, len(data))
            assert r == len(data) and out == data, (i, level, r)
            # partial decoding: same return value, same bytes up to it
            for target in (0, 1, 1000, len(data) // 3, len(data) - 1, len(data)):
                a = guarded(len(data)); b = C.create_string_buffer(len(data) + 64)
                ra = lib.Lizard_decompress_safe_partial(comp, C.addressof(a) + GUARD, len(comp), target, len(data))
                rb = R.Lizard_decompress_safe_partial(comp, b, len(comp), target, len(data))
                assert ra == rb and ra >= min(target, len(data)), (i, level, target, ra, rb)
                assert guard_ok(a, len(data))
                keep = min(target, ra)
                assert bytes(a)[GUARD:GUARD + keep] == data[:keep], (i, level, target)


def test_dictionaries_and_streaming(lib):
    """linked blocks from the reference's Lizard_compress_continue: decoded with the history as a prefix, as an external
    dictionary, split between both (Lizard_decompress_safe_continue after a move), and through usingDict / forceExtDict"""
    R = stock()
    if R is None:
        pytest.skip("oracle/_ref/liblizard_ref.so not present")
    data = util.datagen(1 << 20, 0.6, 0.0, 12)
    bs = 100000
    for level in (10, 15, 21, 24, 31, 41):
        st = R.Lizard_createStream(level)
        blocks = []
        src = C.create_string_buffer(data, len(data))        # contiguous input: every block may refer to the ones before it
        for off in range(0, len(data), bs):
            n = min(bs, len(data) - off)
            dst = C.create_string_buffer(n + 1000)
            c = R.Lizard_compress_continue(st, C.addressof(src) + off, dst, n, n + 1000)
            assert c > 0
            blocks.append((off, n, dst.raw[:c]))
        R.Lizard_freeStream(st)
        # (a) _continue, output contiguous: the prefix grows
        sd = lib.Lizard_createStreamDecode()
        out = guarded(len(data))
        base = C.addressof(out) + GUARD
        for off, n, comp in blocks:
            assert lib.Lizard_decompress_safe_continue(sd, comp, base + off, len(comp), n) == n, (level, off)
        assert bytes(out)[GUARD:GUARD + len(data)] == data and guard_ok(out, len(data))
        # (b) _continue, every block decoded into a different buffer: the previous output becomes the external dictionary
        #     (only ONE block of history is reachable then: compress accordingly)
        lib.Lizard_setStreamDecode(sd, None, 0)
        lib.Lizard_freeStreamDecode(sd)
        # (c) usingDict: history as a prefix (dictStart + dictSize == dest) and as a separate buffer
        for off, n, comp in blocks[1:4]:
            hist = data[:off]
            joined = guarded(off + n)
            C.memmove(C.addressof(joined) + GUARD, hist, off)
            r = lib.Lizard_decompress_safe_usingDict(comp, C.addressof(joined) + GUARD + off, len(comp), n, C.addressof(joined) + GUARD, off)
            assert r == n and bytes(joined)[GUARD + off:GUARD + off + n] == data[off:off + n] and guard_ok(joined, off + n), (level, off)
            sep = guarded(n)
            dic = C.create_string_buffer(hist, off)
            for fn in (lib.Lizard_decompress_safe_usingDict, lib.Lizard_decompress_safe_forceExtDict):
                r = fn(comp, C.addressof(sep) + GUARD, len(comp), n, dic, off)
                assert r == n and bytes(sep)[GUARD:GUARD + n] == data[off:off + n] and guard_ok(sep, n), (level, off)
            # without its history the block must not decode to the right bytes silently: refused, or (no reference into the
            # history was needed) still exact
            r = lib.Lizard_decompress_safe(comp, C.addressof(sep) + GUARD, len(comp), n)
            assert r < 0 or bytes(sep)[GUARD:GUARD + n] == data[off:off + n]
    # split history: ring-buffer style decoding — two alternating buffers, so block k sees block k-1 as external dictionary
    for level in (10, 21, 41):
        st = R.Lizard_createStream(level)
        ring = [C.create_string_buffer(bs), C.create_string_buffer(bs)]
        blocks = []
        for k, off in enumerate(range(0, 600000, bs)):
            C.memmove(ring[k & 1], data[off:off + bs], bs)
            dst = C.create_string_buffer(bs + 1000)
            c = R.Lizard_compress_continue(st, ring[k & 1], dst, bs, bs + 1000)
            assert c > 0
            blocks.append(dst.raw[:c])
        R.Lizard_freeStream(st)
        sd = lib.Lizard_createStreamDecode()
        dring = [guarded(bs), guarded(bs)]
        for k, comp in enumerate(blocks):
            r = lib.Lizard_decompress_safe_continue(sd, comp, C.addressof(dring[k & 1]) + GUARD, len(comp), bs)
            assert r == bs and bytes(dring[k & 1])[GUARD:GUARD + bs] == data[k * bs:(k + 1) * bs], (level, k)
            assert guard_ok(dring[k & 1], bs)
        lib.Lizard_freeStreamDecode(sd)


def test_damaged_blocks_are_refused_or_decoded_like_the_reference(lib):
    R = stock()
    rnd = random.Random(77)
    data = util.datagen(40000, 0.5, 0.0, 2)
    cases = 0
    for level in (10, 21, 30, 41, 17, 35):
        comp0 = util.oracle_compress(data, level)
        for _ in range(400):
            comp = bytearray(comp0)
            kind = rnd.randrange(4)
            if kind == 0:
                for _ in range(rnd.randrange(1, 4)):
                    comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
            elif kind == 1:
                comp = comp[:rnd.randrange(1, len(comp))]
            elif kind == 2:
                p = rnd.randrange(len(comp)); comp[p:p + rnd.randrange(1, 9)] = rnd.randbytes(rnd.randrange(1, 9))
            else:
                comp += rnd.randbytes(rnd.randrange(1, 20))
            comp = bytes(comp)
            cap = len(data) + rnd.choice((0, 0, 100, -100))
            r, out = my_decode(lib, comp, cap)                 # (my_decode asserts the canaries)
            cases += 1
            if R is not None and r >= 0:
                dst = C.create_string_buffer(cap + 4096)
                rr = R.Lizard_decompress_safe(comp, dst, len(comp), cap)
                assert rr == r and dst.raw[:r] == out, (level, r, rr)
    assert cases == 2400


PREF_CASES = [(bsid, mode, crc, csz) for bsid in (1, 2, 4) for mode in (0, 1) for crc in (0, 1) for csz in (0, 1)]


def _my_frame_decode(L, frame, expect_len, rnd, max_in=None, max_out=None):
    dctx = C.c_void_p()
    assert L.LizardF_createDecompressionContext(C.byref(dctx), 100) == 0
    src = C.create_string_buffer(frame, len(frame))
    out = bytearray()
    so = 0
    r = 1
    guard = 0
    while so < len(frame) or r != 0:
        avail_in = min(len(frame) - so, rnd.randrange(1, max_in) if max_in else len(frame) - so)
        cap = rnd.randrange(1, max_out) if max_out else expect_len + 16
        dst = guarded(cap)
        ds, ss = C.c_size_t(cap), C.c_size_t(avail_in)
        r = L.LizardF_decompress(dctx, C.addressof(dst) + GUARD, C.byref(ds), C.addressof(src) + so, C.byref(ss), None)
        assert not L.LizardF_isError(r), L.LizardF_getErrorName(r)
        assert guard_ok(dst, cap) and ss.value <= avail_in and ds.value <= cap
        out += bytes(dst)[GUARD:GUARD + ds.value]
        so += ss.value
        guard += 1
        assert guard < 10_000_000
        if r == 0 and so == len(frame):
            break
    assert L.LizardF_freeDecompressionContext(dctx) == 0     # 0: no frame under way
    return bytes(out)


def test_frames_of_the_reference_decode_under_any_segmentation(lib):
    R = stock()
    if R is None:
        pytest.skip("oracle/_ref/liblizard_ref.so not present")
    rnd = random.Random(3)
    data = util.datagen(700000, 0.5, 0.0, 31) + bytes(50000) + random.Random(1).randbytes(140000)   # compressible, run, stored-raw blocks
    for bsid, mode, crc, csz in PREF_CASES:
        for level in (10, 41):
            p = util.frame_prefs(level, bsid, crc, csz, block_mode=mode)
            frame = util.reference_frame(data, p)
            assert _my_frame_decode(lib, frame, len(data), rnd) == data, (bsid, mode, crc, csz, level)
            assert _my_frame_decode(lib, frame, len(data), rnd, max_in=70000, max_out=90000) == data, (bsid, mode, crc, csz, level)
    # tiny pieces on a small frame, every preference
    small = data[:30000] + data[-3000:]
    for bsid, mode, crc, csz in PREF_CASES:
        frame = util.reference_frame(small, util.frame_prefs(21, bsid, crc, csz, block_mode=mode))
        assert _my_frame_decode(lib, frame, len(small), rnd, max_in=7, max_out=9) == small
    # empty frame, one-byte frame
    for n in (0, 1):
        frame = util.reference_frame(data[:n], util.frame_prefs(10, 1, 1, 0))
        assert _my_frame_decode(lib, frame, n, rnd, max_in=3, max_out=3) == data[:n]
    # a linked frame far longer than the 16 MiB history the decoder keeps (the internal buffer slides)
    big = util.datagen(40 << 20, 0.7, 0.0, 8)
    frame = util.reference_frame(big, util.frame_prefs(10, 4, 1, 1, block_mode=0))
    assert _my_frame_decode(lib, frame, len(big), rnd, max_in=3 << 20, max_out=5 << 20) == big


def test_frame_errors_and_frame_info(lib):
    R = stock()
    if R is None:
        pytest.skip("oracle/_ref/liblizard_ref.so not present")
    data = util.datagen(300000, 0.5, 0.0, 6)
    frame = util.reference_frame(data, util.frame_prefs(10, 2, 1, 1, block_mode=0))
    dctx = C.c_void_p()
    lib.LizardF_createDecompressionContext(C.byref(dctx), 100)
    info = util.FrameInfo()
    n = C.c_size_t(6)
    assert lib.LizardF_getFrameInfo(dctx, C.byref(info), frame, C.byref(n)) == (1 << 64) - 12 and n.value == 0    # frameHeader_incomplete
    n = C.c_size_t(len(frame))
    hint = lib.LizardF_getFrameInfo(dctx, C.byref(info), frame, C.byref(n))
    assert not lib.LizardF_isError(hint) and n.value == 15
    assert (info.blockSizeID, info.blockMode, info.contentChecksumFlag, info.contentSize) == (2, 0, 1, len(data))
    # again, after the header was consumed: reports without consuming
    n2 = C.c_size_t(100)
    assert not lib.LizardF_isError(lib.LizardF_getFrameInfo(dctx, C.byref(info), frame[15:], C.byref(n2))) and n2.value == 0
    assert lib.LizardF_freeDecompressionContext(dctx) != 0                                   # a frame was under way
    # damaged frames: bad magic, bad header checksum, bad content checksum, truncated block size word
    rnd = random.Random(1)

    def decode_err(fr):
        d = C.c_void_p()
        lib.LizardF_createDecompressionContext(C.byref(d), 100)
        dst = C.create_string_buffer(len(data) + 100)
        ds, ss = C.c_size_t(len(data) + 100), C.c_size_t(len(fr))
        r = lib.LizardF_decompress(d, dst, C.byref(ds), fr, C.byref(ss), None)
        lib.LizardF_freeDecompressionContext(d)
        return r
    bad = bytearray(frame); bad[0] ^= 1
    assert decode_err(bytes(bad)) == (1 << 64) - 13                                            # frameType_unknown
    bad = bytearray(frame); bad[6] ^= 1
    assert decode_err(bytes(bad)) == (1 << 64) - 17                                            # headerChecksum_invalid
    bad = bytearray(frame); bad[-1] ^= 1
    assert decode_err(bytes(bad)) == (1 << 64) - 18                                            # contentChecksum_invalid
    bad = bytearray(frame); bad[15:19] = struct.pack("<I", 0x7FFFFFFF)
    assert lib.LizardF_isError(decode_err(bytes(bad)))                                         # block larger than the frame's block size
    for _ in range(200):                                                                       # random damage: an error or the exact data, never a crash
        bad = bytearray(frame)
        bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
        decode_err(bytes(bad))
    assert lib.LizardF_getErrorName((1 << 64) - 11) == b"ERROR_dstMaxSize_tooSmall"
    assert lib.LizardF_getErrorName(5) == b"Unspecified error code"


def test_frames_without_a_gpu_store_raw_and_round_trip(lib):
    """LizardF_compressFrame keeps the reference's contract when the block compressor cannot run (no device, or a level
    without a kernel): the blocks are stored raw (lib/lizard_frame.c:456-469) and the frame decodes — with this library's
    decoder and with the reference's."""
    import torch
    data = util.datagen(300000, 0.5, 0.0, 9)
    level = 23 if torch.cuda.is_available() else 10                     # level 23 (lowestPrice) has no GPU kernel
    lib.LizardF_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
    lib.LizardF_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p]
    for mode in (0, 1):
        p = util.frame_prefs(level, 1, 1, 1, block_mode=mode)
        cap = lib.LizardF_compressFrameBound(len(data), C.byref(p))
        dst = C.create_string_buffer(cap)
        n = lib.LizardF_compressFrame(dst, cap, data, len(data), C.byref(p))
        assert not lib.LizardF_isError(n)
        frame = dst.raw[:n]
        assert n == 15 + 3 * 4 + len(data) + 4 + 4                       # header, three raw block records, end mark, checksum
        assert _my_frame_decode(lib, frame, len(data), random.Random(2), max_in=50000, max_out=30000) == data


@pytest.mark.gpu
def test_one_byte_tail_block_in_a_buffer_of_exactly_the_bound(lib):
    """A 1-byte block costs the reference a 10-byte record where LizardF_compressBound counted 5 (the room test of
    lizard_compress.c:238 wraps at maxDstSize 0).  With dst = LizardF_compressFrameBound exactly, a 15-byte header and
    nothing saved on the other blocks the reference writes past the bound (found by tests/frametest.c, seed 7254 #1031);
    this library stays inside dstMaxSize: LizardF_compressFrame stores that byte raw when the faithful record would not
    leave room for the end mark, the strict twin reports dstMaxSize_tooSmall, and whenever the faithful frame fits it is
    the one produced."""
    lib.LizardF_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
    lib.LizardF_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.LizardGPU_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.LizardGPU_compressFrame.restype = C.c_size_t
    noise = random.Random(5).randbytes(131072)
    inputs = [b"Q", noise + b"Q", noise + noise[::-1] + b"Q", util.datagen(131072, 0.5, 0.0, 2) + b"Q"]
    squeezed = 0
    for data in inputs:
        for level in (10, 21, 30):
            for crc in (0, 1):
                for csz in (0, 1):
                    p = util.frame_prefs(level, 1, crc, csz)
                    cap = lib.LizardF_compressFrameBound(len(data), C.byref(p))
                    faithful = util.compose_frame(data, level, 1, crc, csz, util.oracle_compress)
                    for fn in (lib.LizardF_compressFrame, lib.LizardGPU_compressFrame):
                        dst = guarded(cap)
                        n = fn(C.addressof(dst) + GUARD, cap, data, len(data), C.byref(p))
                        assert guard_ok(dst, cap), "frame compressor wrote outside dstMaxSize"
                        if len(faithful) <= cap:
                            assert not lib.LizardF_isError(n) and bytes(dst)[GUARD:GUARD + n] == faithful
                        elif fn is lib.LizardGPU_compressFrame:
                            assert n == (1 << 64) - 11                                      # refused: dstMaxSize_tooSmall
                        else:
                            squeezed += 1
                            assert not lib.LizardF_isError(n) and n <= cap and n == len(faithful) - 5
                            frame = bytes(dst)[GUARD:GUARD + n]
                            assert frame[:-4 - 4 * crc - 5] == faithful[:-4 - 4 * crc - 10]     # everything before the last record
                            assert _my_frame_decode(lib, frame, len(data), random.Random(1)) == data
                            R = stock()
                            if R is not None:
                                back = C.create_string_buffer(len(data) + 16)
                                d = C.c_void_p()
                                R.LizardF_createDecompressionContext.argtypes = [C.c_void_p, C.c_uint]
                                R.LizardF_decompress.restype = C.c_size_t
                                R.LizardF_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
                                R.LizardF_createDecompressionContext(C.byref(d), 100)
                                ds, ss = C.c_size_t(len(data) + 16), C.c_size_t(n)
                                assert R.LizardF_decompress(d, back, C.byref(ds), frame, C.byref(ss), None) == 0
                                assert back.raw[:ds.value] == data
    assert squeezed >= 6          # the 1-byte input and the all-raw inputs with a content size in the header


def test_xxhash_matches_the_specification(lib):
    import xxhash
    rnd = random.Random(9)
    blob = rnd.randbytes(5000)
    for n in list(range(0, 70)) + [100, 255, 256, 1000, 4999, 5000]:
        for seed in (0, 1, 0x9E3779B1):
            assert lib.Lizard_XXH32(blob, n, seed) == xxhash.xxh32(blob[:n], seed=seed).intdigest(), (n, seed)
            assert lib.Lizard_XXH64(blob, n, seed) == xxhash.xxh64(blob[:n], seed=seed).intdigest(), (n, seed)
    # streaming in random pieces; the state is the caller's: 48 / 88 bytes as the reference declares them
    for _ in range(200):
        n = rnd.randrange(0, 5000)
        st32, st64 = guarded(48), guarded(88)
        a32, a64 = C.addressof(st32) + GUARD, C.addressof(st64) + GUARD
        seed = rnd.randrange(1 << 32)
        lib.Lizard_XXH32_reset(a32, seed); lib.Lizard_XXH64_reset(a64, seed)
        pos = 0
        while pos < n:
            k = min(n - pos, rnd.choice((1, 3, 15, 16, 17, 31, 32, 33, 100, 1000)))
            piece = blob[pos:pos + k]
            lib.Lizard_XXH32_update(a32, piece, k); lib.Lizard_XXH64_update(a64, piece, k)
            pos += k
        assert lib.Lizard_XXH32_digest(a32) == xxhash.xxh32(blob[:n], seed=seed).intdigest()
        assert lib.Lizard_XXH64_digest(a64) == xxhash.xxh64(blob[:n], seed=seed).intdigest()
        assert guard_ok(st32, 48) and guard_ok(st64, 88)


def test_decoder_and_frame_decoder_under_sanitizers(tmp_path):
    """tests/decode_fuzz.c: the three host sources of the decode / frame / hash half built with AddressSanitizer +
    UndefinedBehaviorSanitizer (no HIP needed), valid and damaged blocks and frames in exact-size heap buffers — an access one
    byte outside any buffer, a signed overflow or a misaligned access aborts the run."""
    import subprocess
    exe = str(tmp_path / "decode_fuzz")
    csrc = os.path.join(util.ROOT, "lizard_amd", "csrc")
    util.oracle()
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu99", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(util.ROOT, "tests", "decode_fuzz.c"), os.path.join(csrc, "lizard_decode_host.c"),
                           os.path.join(csrc, "lizard_frame_host.c"), os.path.join(csrc, "lizard_xxhash.c"),
                           "-I" + os.path.join(util.ROOT, "include"), "-I" + util.ORACLE_DIR, "-L" + util.ORACLE_DIR, "-llizard_oracle",
                           "-lpthread", "-Wl,-rpath," + util.ORACLE_DIR, "-o", exe])
    for seed in (11, 12):
        r = subprocess.run([exe, str(seed), "400"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert "0 misbehaved" in r.stdout
