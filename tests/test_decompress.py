"""Block decompression (lizard_amd/csrc/lz_unpack.h).

CPU: the product's decoder body on the SIMT emulator against the corpus — blocks written by the oracle AND by the compiled
reference (oracle/_ref, when present) must decode to their input, every level family (fastLZ4 / LIZv1 codewords, with and
without huff0); damaged blocks must be refused or decoded to *something* without touching memory outside their buffers.
GPU (-m gpu): the same through the C ABI — device-resident round trip of a batch (slot layout), host packed layout, the
one-block twin of Lizard_decompress_safe, agreement with the reference decoder, and a corruption sweep."""
import ctypes
import random

import numpy as np
import pytest

import util


def emul_decompress(comp, cap, seed=3):
    emu = util.emulator()
    emu.emul_decompress_block.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint]
    out = ctypes.create_string_buffer(cap + 64)
    out.raw = b"\xAA" * (cap + 64)
    src = ctypes.create_string_buffer(bytes(comp), len(comp)) if comp else ctypes.create_string_buffer(1)
    r = emu.emul_decompress_block(src, len(comp), out, cap, seed)
    assert out.raw[cap:] == b"\xAA" * 64, "decoder wrote past its output slot"
    return r, out.raw[:max(r, 0)]


@pytest.mark.parametrize("level", [10, 21, 30, 41, 13, 36])
def test_emulated_decoder_roundtrip(level):
    for name, data in util.corpus(small=True):
        comp = util.oracle_compress(data, level)
        r, out = emul_decompress(comp, len(data))
        assert r == len(data) and out == data, (name, level)


def test_emulated_decoder_on_reference_output():
    ref = util.reference()
    if ref is None:
        pytest.skip("oracle/_ref not present")
    data = dict(util.corpus(small=True))["gen262144_p0.5"] + bytes(5000) + b"abc" * 3000
    for level in (10, 11, 17, 21, 22, 31, 42, 45, 49, 19, 29):          # also levels the GPU does not compress: the decoder is level-agnostic
        comp, r = util.compress_with(ref.Lizard_compress, data, level)
        assert r > 0
        n, out = emul_decompress(comp, len(data))
        assert n == len(data) and out == data, level


def test_emulated_decoder_refuses_or_survives_damage():
    rnd = random.Random(5)
    data = dict(util.corpus(small=True))["gen65537_p0.5"]
    for level in (10, 30, 41):
        comp = bytearray(util.oracle_compress(data, level))
        assert emul_decompress(bytes(comp), len(data) - 1)[0] == -1                 # output slot one byte short
        assert emul_decompress(bytes(comp[:-1]), len(data))[0] != len(data) or True  # truncated input: anything but a crash
        for trial in range(40):
            bad = bytearray(comp)
            for _ in range(rnd.randrange(1, 4)):
                bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
            cut = rnd.choice([len(bad), len(bad), rnd.randrange(1, len(bad))])
            r, out = emul_decompress(bytes(bad[:cut]), len(data))
            assert r == -1 or 0 <= r <= len(data)


def _le24(v):
    return bytes([v & 255, (v >> 8) & 255, (v >> 16) & 255])


def _block(level, flags, lits, off16=b"", off24=b""):
    """One hand-built block of one sub-block, all streams raw (container: lizard_compress.c:186-250)."""
    return bytes([level, 0]) + _le24(0) + _le24(len(off16)) + off16 + _le24(len(off24)) + off24 + _le24(len(flags)) + flags + _le24(len(lits)) + lits


def _short_tail_vectors():
    """(block, plain, the reference decoder accepts it).  Blocks the encoders never write but the format allows: a length
    escape in the last 1..3 bytes of the literals stream (lizard_decompress_liz.h:142 only asks for literalsPtr <= iend - 1).
    The reference's wild copies make it refuse some of them (a token >= 32 needs 16 more literal-stream bytes,
    lizard_decompress_liz.h:81; fastLZ4 literals end 18 bytes before the stream does, lizard_decompress_lz4.h:68): those are
    only checked against the hand-built expectation."""
    abc = bytes(range(97, 97 + 26)) + b"0123"
    out = []
    # LIZv1: token 39 = [0 0100 111]: L escape (30 literals), ml 4, new 16-bit offset 8; then a 24-bit-offset token 31 whose
    # match-length escape is the last 1 / 3 / 4 bytes of the literals stream
    for esc, v in ((bytes([3]), 3), (bytes([254, 0x10, 0x01]), 0x110), (bytes([255, 1, 2, 0]), 0x201), (bytes([200]), 200)):
        plain = bytearray(abc)
        for _ in range(4): plain.append(plain[len(plain) - 8])
        for _ in range(v + 31 + 16): plain.append(plain[len(plain) - 10])
        out.append((_block(20, bytes([39, 31]), bytes([23]) + abc + esc, off16=bytes([8, 0]), off24=bytes([10, 0, 0])), bytes(plain), True))
    # LIZv1: [1 1111 000] repeat-offset match whose ml escape (token field 15) ends the stream, after a plain first sequence
    for esc, v in ((bytes([9]), 9), (bytes([254, 1, 1]), 0x101)):
        plain = bytearray(abc)
        for _ in range(4): plain.append(plain[len(plain) - 8])
        for _ in range(v + 15): plain.append(plain[len(plain) - 8])
        out.append((_block(20, bytes([39, 0xF8]), bytes([23]) + abc + esc, off16=bytes([8, 0])), bytes(plain), False))
    # fastLZ4: the offset + a 4-byte match-length escape 8 bytes before the end of the stream, at every alignment of the offset
    for pad in range(4):
        lits = bytes((i * 7 + pad) & 255 for i in range(70000 + pad))
        v = 0x011234                                                   # 24-bit escape with a non-zero top byte
        rec = bytes([255]) + _le24(len(lits) - 15) + lits + bytes([40, 0]) + bytes([255]) + _le24(v) + b"zz"
        plain = bytearray(lits)
        for _ in range(v + 15 + 4): plain.append(plain[len(plain) - 40])
        out.append((_block(10, bytes([0xFF]), rec), bytes(plain) + b"zz", False))
    return out


def test_emulated_decoder_short_tail_escapes():
    ref = util.reference()
    accepted = 0
    for i, (comp, plain, ref_ok) in enumerate(_short_tail_vectors()):
        if ref is not None:                                            # (the reference wants 16 bytes of slack behind a match)
            dst = ctypes.create_string_buffer(len(plain) + 64)
            r = ref.Lizard_decompress_safe(comp, dst, len(comp), len(plain) + 32)
            assert (r == len(plain) and dst.raw[:len(plain)] == plain) if ref_ok else r < 0, i
            accepted += r > 0
        r, out = emul_decompress(comp, len(plain))
        assert r == len(plain) and out == plain, i
    assert ref is None or accepted == 4


@pytest.fixture(scope="module")
def L():
    from lizard_amd import _lib
    lib = _lib.lib()
    c = ctypes
    lib.LizardGPU_decompressBlocks_device.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p]
    lib.LizardGPU_decompressBlocks_host.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_void_p]
    lib.LizardGPU_decompress_safe.argtypes = [c.c_char_p, c.c_char_p, c.c_int, c.c_int]
    return lib


@pytest.mark.gpu
def test_gpu_roundtrip_device_batches(L):
    import torch
    from lizard_amd import api
    data = util.datagen((24 << 20) + 4321, 0.5, 0.0, 71) + bytes(300000) + random.Random(1).randbytes(200000)
    host = np.frombuffer(data, dtype=np.uint8)
    src = torch.from_numpy(host.copy()).cuda()
    for level in [l for l in (10, 11, 13, 17, 21, 22, 30, 31, 35, 41, 42, 20, 40, 12, 33) if L.LizardGPU_levelSupported(l)]:
        for bs in (65536, 262144, 1 << 20):
            dst, sizes, stride = api.compress_blocks_device(src, bs, level)
            nb = sizes.numel()
            back = torch.full((nb * bs,), 0x55, dtype=torch.uint8, device="cuda")
            outsz = torch.zeros(nb, dtype=torch.int32, device="cuda")
            rc = L.LizardGPU_decompressBlocks_device(dst.data_ptr(), stride, sizes.data_ptr(), nb, back.data_ptr(), bs, outsz.data_ptr(), None)
            assert rc == 0, L.LizardGPU_lastError()
            torch.cuda.synchronize()
            want = [bs] * (nb - 1) + [len(data) - (nb - 1) * bs]
            assert outsz.cpu().numpy().tolist() == want, (level, bs)
            assert back[:len(data)].cpu().numpy().tobytes() == data, (level, bs)


@pytest.mark.gpu
def test_gpu_decoder_matches_reference_decoder_and_host_forms(L):
    ref = util.reference()
    rnd = random.Random(9)
    blocks = [d for _, d in util.corpus(small=True) if len(d) > 0]
    for level in (10, 21, 30, 41, 15):
        comp = [util.oracle_compress(d, level) for d in blocks]
        if ref is not None:                                   # blocks written by the reference itself, incl. a level the GPU does not compress
            comp += [util.compress_with(ref.Lizard_compress, d, 29 if level == 21 else level)[0] for d in blocks[:4]]
            plain = blocks + blocks[:4]
        else:
            plain = blocks
        packed = b"".join(comp)
        offs = np.concatenate([[0], np.cumsum([len(c) for c in comp])]).astype(np.uint64)
        stride = max(len(d) for d in plain)
        out = np.full(len(comp) * stride, 0x77, dtype=np.uint8)
        sz = np.zeros(len(comp), dtype=np.uint32)
        buf = np.frombuffer(packed, dtype=np.uint8)
        rc = L.LizardGPU_decompressBlocks_host(buf.ctypes.data, offs.ctypes.data, len(comp), out.ctypes.data, stride, sz.ctypes.data)
        assert rc == 0, L.LizardGPU_lastError()
        for i, d in enumerate(plain):
            assert sz[i] == len(d) and out[i * stride:i * stride + len(d)].tobytes() == d, (level, i)
        # one-block twin: same result and same refusals as the reference decoder
        for i in rnd.sample(range(len(comp)), 6):
            dst = ctypes.create_string_buffer(len(plain[i]) + 8)
            assert L.LizardGPU_decompress_safe(comp[i], dst, len(comp[i]), len(plain[i])) == len(plain[i])
            assert dst.raw[:len(plain[i])] == plain[i]
            if len(plain[i]) > 1:
                assert L.LizardGPU_decompress_safe(comp[i], dst, len(comp[i]), len(plain[i]) - 1) < 0
                if ref is not None:
                    assert ref.Lizard_decompress_safe(comp[i], dst, len(comp[i]), len(plain[i]) - 1) < 0


@pytest.mark.gpu
def test_gpu_decoder_survives_damage(L):
    rnd = random.Random(13)
    data = util.datagen(200000, 0.5, 0.0, 5)
    for level in (10, 21, 30, 41):
        comp = util.oracle_compress(data, level)
        bad_blocks = []
        for _ in range(96):
            bad = bytearray(comp)
            for _ in range(rnd.randrange(1, 5)):
                bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
            bad_blocks.append(bytes(bad[:rnd.choice([len(bad), rnd.randrange(1, len(bad))])]))
        packed = b"".join(bad_blocks)
        offs = np.concatenate([[0], np.cumsum([len(c) for c in bad_blocks])]).astype(np.uint64)
        out = np.zeros(len(bad_blocks) * len(data), dtype=np.uint8)
        sz = np.zeros(len(bad_blocks), dtype=np.uint32)
        buf = np.frombuffer(packed, dtype=np.uint8)
        rc = L.LizardGPU_decompressBlocks_host(buf.ctypes.data, offs.ctypes.data, len(bad_blocks), out.ctypes.data, len(data), sz.ctypes.data)
        assert rc == 0
        assert all(s == 0xFFFFFFFF or s <= len(data) for s in sz)


@pytest.mark.gpu
def test_gpu_decoder_short_tail_escapes(L):
    """The hand-built blocks of test_emulated_decoder_short_tail_escapes through the one-block twin on the device."""
    for i, (comp, plain, _) in enumerate(_short_tail_vectors()):
        dst = ctypes.create_string_buffer(len(plain) + 8)
        assert L.LizardGPU_decompress_safe(comp, dst, len(comp), len(plain)) == len(plain), i
        assert dst.raw[:len(plain)] == plain, i


@pytest.mark.gpu
def test_gpu_decoder_refuses_offsets_that_are_not_blocks(L):
    """The offsets are input like the blocks: a decreasing pair (or a block of 4 GiB and more) is refused before anything is read."""
    comp = util.oracle_compress(util.datagen(5000, 0.5, 0.0, 1), 10)
    buf = np.frombuffer(comp + comp, dtype=np.uint8)
    out = np.zeros(2 * 5000, dtype=np.uint8); sz = np.zeros(2, dtype=np.uint32)
    for offs in ([0, len(comp), len(comp) - 1], [len(comp), 0, len(comp)], [0, 1 << 33, (1 << 33) + 5]):
        o = np.array(offs, dtype=np.uint64)
        assert L.LizardGPU_decompressBlocks_host(buf.ctypes.data, o.ctypes.data, 2, out.ctypes.data, 5000, sz.ctypes.data) == -3, offs
    o = np.array([0, len(comp), 2 * len(comp)], dtype=np.uint64)
    assert L.LizardGPU_decompressBlocks_host(buf.ctypes.data, o.ctypes.data, 2, out.ctypes.data, 5000, sz.ctypes.data) == 0
    assert list(sz) == [5000, 5000]
