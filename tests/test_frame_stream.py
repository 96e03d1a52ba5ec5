"""Streaming frames: LizardGPU_compressBegin / _compressUpdate / _flush / _compressEnd (the twin of the reference's
LizardF_compressBegin/Update/flush/End, lib/lizard_frame.c:362-677) must write, call for call, the bytes the
compiled reference writes for the same preferences and the same sequence of Update sizes (independent blocks,
zero-state build).  CPU part: refusals and bounds through the C ABI (no GPU needed).  GPU part (-m gpu): random call
patterns with and without autoFlush against oracle/_ref (which travels to the GPU box as a binary)."""
import ctypes
import random

import pytest

import util


@pytest.fixture(scope="module")
def lib():
    from lizard_amd import _lib
    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    c = ctypes
    L.LizardGPU_createCompressionContext.argtypes = [c.POINTER(c.c_void_p)]
    L.LizardGPU_freeCompressionContext.argtypes = [c.c_void_p]
    for name, args in (("LizardGPU_compressBegin", [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p]),
                       ("LizardGPU_compressBound", [c.c_size_t, c.c_void_p]),
                       ("LizardGPU_compressUpdate", [c.c_void_p, c.c_void_p, c.c_size_t, c.c_char_p, c.c_size_t]),
                       ("LizardGPU_flush", [c.c_void_p, c.c_void_p, c.c_size_t]),
                       ("LizardGPU_compressEnd", [c.c_void_p, c.c_void_p, c.c_size_t])):
        getattr(L, name).argtypes = args
        getattr(L, name).restype = c.c_size_t
    L.LizardGPU_frameIsError.argtypes = [c.c_size_t]
    return L


def err(code):
    return (1 << 64) - code


def ours_stream(lib, data, prefs, cuts, flush_at=()):
    """Feed `data` in pieces ending at the offsets in `cuts`; returns the frame bytes."""
    ctx = ctypes.c_void_p()
    assert lib.LizardGPU_createCompressionContext(ctypes.byref(ctx)) == 0
    out = bytearray()
    cap = 64
    dst = ctypes.create_string_buffer(cap)
    n = lib.LizardGPU_compressBegin(ctx, dst, cap, ctypes.byref(prefs))
    assert not lib.LizardGPU_frameIsError(n), n - (1 << 64)
    out += dst.raw[:n]
    pos = 0
    for k, cut in enumerate(list(cuts) + [len(data)]):
        piece = data[pos:cut]
        cap = lib.LizardGPU_compressBound(len(piece), ctypes.byref(prefs)) + 16
        dst = ctypes.create_string_buffer(cap)
        n = lib.LizardGPU_compressUpdate(ctx, dst, cap, piece, len(piece))
        assert not lib.LizardGPU_frameIsError(n), n - (1 << 64)
        out += dst.raw[:n]
        pos = cut
        if k in flush_at:
            n = lib.LizardGPU_flush(ctx, dst, cap)
            assert not lib.LizardGPU_frameIsError(n)
            out += dst.raw[:n]
    cap = util.FRAME_BLOCK_SIZES[prefs.frameInfo.blockSizeID or 1] + 32
    dst = ctypes.create_string_buffer(cap)
    n = lib.LizardGPU_compressEnd(ctx, dst, cap)
    assert not lib.LizardGPU_frameIsError(n), n - (1 << 64)
    out += dst.raw[:n]
    lib.LizardGPU_freeCompressionContext(ctx)
    return bytes(out)


def reference_stream(ref, data, prefs, cuts, flush_at=()):
    c = ctypes
    for name, args in (("LizardF_compressBegin", [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p]),
                       ("LizardF_compressBound", [c.c_size_t, c.c_void_p]),
                       ("LizardF_compressUpdate", [c.c_void_p, c.c_void_p, c.c_size_t, c.c_char_p, c.c_size_t, c.c_void_p]),
                       ("LizardF_flush", [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p]),
                       ("LizardF_compressEnd", [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p])):
        getattr(ref, name).argtypes = args
        getattr(ref, name).restype = c.c_size_t
    ref.LizardF_createCompressionContext.argtypes = [c.POINTER(c.c_void_p), c.c_uint]
    ref.LizardF_createCompressionContext.restype = c.c_size_t
    ref.LizardF_freeCompressionContext.argtypes = [c.c_void_p]
    ctx = c.c_void_p()
    assert ref.LizardF_createCompressionContext(c.byref(ctx), 100) == 0
    out = bytearray()
    dst = c.create_string_buffer(64)
    n = ref.LizardF_compressBegin(ctx, dst, 64, c.byref(prefs))
    assert n < (1 << 63)
    out += dst.raw[:n]
    pos = 0
    for k, cut in enumerate(list(cuts) + [len(data)]):
        piece = data[pos:cut]
        cap = ref.LizardF_compressBound(len(piece), c.byref(prefs)) + 16
        dst = c.create_string_buffer(cap)
        n = ref.LizardF_compressUpdate(ctx, dst, cap, piece, len(piece), None)
        assert n < (1 << 63)
        out += dst.raw[:n]
        pos = cut
        if k in flush_at:
            n = ref.LizardF_flush(ctx, dst, cap, None)
            assert n < (1 << 63)
            out += dst.raw[:n]
    cap = util.FRAME_BLOCK_SIZES[prefs.frameInfo.blockSizeID or 1] + 32
    dst = c.create_string_buffer(cap)
    n = ref.LizardF_compressEnd(ctx, dst, cap, None)
    assert n < (1 << 63)
    out += dst.raw[:n]
    ref.LizardF_freeCompressionContext(ctx)
    return bytes(out)


def test_stream_refusals_need_no_gpu(lib):
    ctx = ctypes.c_void_p()
    assert lib.LizardGPU_createCompressionContext(ctypes.byref(ctx)) == 0
    dst = ctypes.create_string_buffer(64)
    assert lib.LizardGPU_compressBegin(ctx, dst, 10, ctypes.byref(util.frame_prefs(10, 1, 0, 0))) == err(11)      # header room
    assert lib.LizardGPU_compressBegin(ctx, dst, 64, ctypes.byref(util.frame_prefs(10, 1, 0, 0, block_mode=0))) == err(3)   # linked
    assert lib.LizardGPU_compressBegin(ctx, dst, 64, ctypes.byref(util.frame_prefs(23, 1, 0, 0))) == err(5)      # no GPU kernel (lowestPrice)
    for level in (11, 21, 13):                                 # 16 MiB (and larger) frame blocks are taken at every GPU level since round 3
        c2 = ctypes.c_void_p()
        assert lib.LizardGPU_createCompressionContext(ctypes.byref(c2)) == 0
        assert lib.LizardGPU_compressBegin(c2, dst, 64, ctypes.byref(util.frame_prefs(level, 7, 0, 0))) == 7
        lib.LizardGPU_freeCompressionContext(c2)
    p = util.frame_prefs(10, 1, 0, 0)
    p.frameInfo.frameType = 1
    assert lib.LizardGPU_compressBegin(ctx, dst, 64, ctypes.byref(p)) == err(13)                                 # skippable frame
    assert lib.LizardGPU_compressUpdate(ctx, dst, 64, b"abc", 3) == err(1)                                       # Update before Begin
    n = lib.LizardGPU_compressBegin(ctx, dst, 64, ctypes.byref(util.frame_prefs(10, 2, 1, 0)))
    assert n == 7 and dst.raw[:4] == bytes([0x06, 0x22, 0x4D, 0x18])
    assert lib.LizardGPU_compressBegin(ctx, dst, 64, None) == err(1)                                             # Begin twice
    assert lib.LizardGPU_compressUpdate(ctx, dst, 10, b"x" * 1000, 1000) == err(11)                              # below the bound
    lib.LizardGPU_freeCompressionContext(ctx)
    ref = util.reference()
    if ref is not None:
        ref.LizardF_compressBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
        ref.LizardF_compressBound.restype = ctypes.c_size_t
        for n in (0, 1, 131072, 300000, 5 << 20):
            for bsid in (1, 2, 4):
                for af in (0, 1):
                    p = util.frame_prefs(10, bsid, 1, 0)
                    p.autoFlush = af
                    assert lib.LizardGPU_compressBound(n, ctypes.byref(p)) == ref.LizardF_compressBound(n, ctypes.byref(p))


@pytest.mark.gpu
def test_gpu_stream_equals_reference_stream(lib):
    ref = util.reference()
    if ref is None:
        util.need_ref("oracle/_ref")
    rnd = random.Random(7)
    data = util.datagen(3 * (1 << 20) + 4321, 0.5, 0.0, 8) + bytes(200000) + rnd.randbytes(150000)
    for trial in range(14):
        level = rnd.choice([10, 10, 30, 21, 11, 13])
        bsid = rnd.choice([1, 1, 2, 3])
        af = trial & 1
        p = util.frame_prefs(level, bsid, rnd.randrange(2), 0)
        p.autoFlush = af
        n = len(data) if trial < 4 else rnd.randrange(1, len(data))
        d = data[:n]
        k = rnd.choice([0, 1, 3, 9])
        cuts = sorted(rnd.randrange(0, n + 1) for _ in range(k))
        flush_at = {0} if (trial % 5 == 0 and not af) else ()
        got = ours_stream(lib, d, p, cuts, flush_at)
        want = reference_stream(ref, d, p, cuts, flush_at)
        assert got == want, (trial, level, bsid, af, cuts)
