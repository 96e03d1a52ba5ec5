"""Shared test helpers: loading the checker libraries (oracle restatement, compiled reference when
present, SIMT emulator) and the deterministic input corpus.  Everything here is test infrastructure."""
import ctypes
import hashlib
import os
import random
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
import sys
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_c = ctypes
_COMPRESS_SIG = [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int]


def build_oracle():
    """(Re)build oracle/liblizard_oracle.so (and oracle/_ref when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liblizard_oracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            build_oracle()
        L = ctypes.CDLL(path)
        L.lzo_compress.argtypes = _COMPRESS_SIG; L.lzo_compress.restype = _c.c_int
        L.lzo_compress_bound.argtypes = [_c.c_int]; L.lzo_compress_bound.restype = _c.c_int
        L.lzo_level_supported.argtypes = [_c.c_int]
        L.lzo_datagen.argtypes = [_c.c_void_p, _c.c_size_t, _c.c_double, _c.c_double, _c.c_uint]
        L.lzo_huf_compress.argtypes = [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t]; L.lzo_huf_compress.restype = _c.c_size_t
        _oracle = L
    return _oracle


_ref = None


def reference():
    """The unmodified reference built with -DLIZARD_RESET_MEM (zero-state oracle), or None."""
    global _ref
    if _ref is None:
        path = os.path.join(REF_DIR, "liblizard_ref_reset.so")
        if not os.path.exists(path):
            if os.path.exists("/root/reference/lib/lizard_compress.c"):
                build_oracle()
            else:
                return None
        L = ctypes.CDLL(path)
        L.Lizard_compress.argtypes = _COMPRESS_SIG; L.Lizard_compress.restype = _c.c_int
        L.Lizard_compressBound.argtypes = [_c.c_int]
        L.Lizard_decompress_safe.argtypes = [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int]; L.Lizard_decompress_safe.restype = _c.c_int
        L.HUF_compress.argtypes = [_c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t]; L.HUF_compress.restype = _c.c_size_t
        _ref = L
    return _ref


def reference_datagen():
    path = os.path.join(REF_DIR, "libdatagen_ref.so")
    if not os.path.exists(path):
        return None
    L = ctypes.CDLL(path)
    L.RDG_genBuffer.argtypes = [_c.c_void_p, _c.c_size_t, _c.c_double, _c.c_double, _c.c_uint]
    return L


def need_ref(what):
    """A test needs a binary / library under oracle/_ref (git-ignored, built by oracle/Makefile where /root/reference exists and
    carried to the GPU box by gpurun).  Absent: skip — or FAIL when LIZARD_REQUIRE_REF=1 (scripts/gpu_session.sh sets it: a GPU
    session that silently skips the reference-program tests proves less than it seems to)."""
    import pytest
    msg = "%s not present (oracle/_ref is built by oracle/Makefile where /root/reference exists)" % what
    if os.environ.get("LIZARD_REQUIRE_REF") == "1":
        pytest.fail(msg + " and LIZARD_REQUIRE_REF=1")
    pytest.skip(msg)


_emul = None


def emulator():
    global _emul
    if _emul is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("emul_build", os.path.join(ROOT, "tests", "emul", "build.py"))
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        L = ctypes.CDLL(mod.build())
        L.emul_compress_block.argtypes = [_c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int, _c.c_uint]
        L.emul_compress_block.restype = _c.c_int
        _emul = L
    return _emul


def compress_with(fn, data, level, cap=None):
    """Run a (src, dst, n, cap, level) -> size C compressor over `data`; returns the output bytes."""
    n = len(data)
    bound = oracle().lzo_compress_bound(n)
    cap = bound if cap is None else cap
    dst = ctypes.create_string_buffer(max(cap, 1) + 64)
    src = ctypes.create_string_buffer(bytes(data), n) if n else ctypes.create_string_buffer(1)
    r = fn(src, dst, n, cap, level)
    return dst.raw[:max(r, 0)], r


def oracle_compress(data, level, cap=None):
    return compress_with(oracle().lzo_compress, data, level, cap)[0]


def datagen(size, match_proba=0.5, lit_proba=0.0, seed=0):
    buf = ctypes.create_string_buffer(max(size, 1))
    oracle().lzo_datagen(buf, size, match_proba, lit_proba, seed)
    return buf.raw[:size]


def sha(b):
    return hashlib.sha256(b).hexdigest()


EDGE_SIZES = [0, 1, 5, 16, 19, 20, 21, 22, 23, 40, 63, 64, 65, 100, 1000, 4096, 65535, 65536, 65537, 131071, 131072,
              131073, 131072 + 20, 131072 + 21, 131072 + 22, 200000, 262144, 300000]


def corpus(small=False):
    """Deterministic named inputs: datagen at several compressibilities and ragged sizes, degenerate
    periodic data (exercise same-hash lanes inside one round), incompressible noise (exercise the
    skip schedule and the stored-raw fallback)."""
    rnd = random.Random(1234)
    out = []
    sizes = [0, 1, 20, 21, 22, 65, 1000, 65537, 131072, 131073, 131072 + 21, 262144] if small else EDGE_SIZES
    for size in sizes:
        for p in ([0.5] if small else [0.0, 0.2, 0.5, 0.9, 1.0]):
            out.append((f"gen{size}_p{p}", datagen(size, p, 0.0, size + 7)))
    n = 262144
    out += [
        ("zeros300k", bytes(300000)),
        ("ff256k", b"\xff" * n),
        ("period3", (b"abc" * (n // 3 + 1))[:n]),
        ("period8", (b"abcdefgh" * (n // 8))[:n]),
        ("period9", (b"abcdefghi" * (n // 9 + 1))[:n]),
        ("period64", (bytes(range(64)) * (n // 64))[:n]),
        ("period65", (bytes(range(65)) * (n // 65 + 1))[:n]),
        ("random256k", rnd.randbytes(n)),
        ("alpha2", bytes(rnd.choice(b"ab") for _ in range(200000))),
        ("alpha4", bytes(rnd.choice(b"abcd") for _ in range(200000))),
        ("text", (b"the quick brown fox jumps over the lazy dog. " * 6000)[:n]),
    ]
    return out


def corpus_long():
    """1 MiB+ inputs with repeats at distances around 2^16 and 2^17 (the reference's 65535-byte window
    edge and the wrap points of position encodings that keep fewer bits than the position), sparse
    long-distance markers in noise, and multi-MiB blocks (many sub-blocks, table persistence)."""
    rnd = random.Random(99)
    out = []
    n = 1 << 20
    for per in [65528, 65535, 65536, 65537, 65544, 131064, 131071, 131072, 131073, 131080, 196608, 262144, 98304, 32768, 40000]:
        chunk = rnd.randbytes(per)
        out.append((f"randperiod{per}", (chunk * (1 + n // per))[:n]))
        chunk2 = datagen(per, 0.5, 0.0, per)
        out.append((f"p50period{per}", (chunk2 * (1 + n // per))[:n]))
    marker = rnd.randbytes(64)
    for name, step in (("markers65529", 65536 - 7), ("markers131077", 131072 + 5)):
        base = bytearray(rnd.randbytes(n))
        for pos in range(1000, n - 100, step):
            base[pos:pos + 64] = marker
        out.append((name, bytes(base)))
    out.append(("p50_4m", datagen(4 << 20, 0.5, 0.0, 5)))
    out.append(("p20_2m", datagen(2 << 20, 0.2, 0.0, 6)))
    out.append(("longoff_rule", longoff_rule_case()))
    return out


def longoff_rule_case():
    """Matches 65 536 or more bytes back whose length sits around the LIZv1 levels' long-offset rule (minMatchLongOff 16: fastBig counts
    the backward extension into it, lizard_parser_fastbig.h:92-98, its post-match probe :143-146 and priceFast :69 do not).
    Layout: a dictionary of groups B_i (24 random bytes) + C_i (40) + 16 zeros (a short match: the fast parsers' step goes back to 1, so
    every position of the next group is inserted), then every B_i once more (now the LAST position of B_i's hashes is not followed by
    C_i: a later copy of B_i's tail + C_i's head is found at C_i's head and extended backwards), a run of 66 000 zeros (one match:
    nothing inside is inserted), then per group a spacer of 0 or 20 random bytes (0: the copy starts at a post-match probe) and
    B_i[-b:] + C_i[:L] with b + L on both sides of the rule; then copies of C_k[:L] behind spacers of thousands of noise bytes."""
    rnd = random.Random(2016)
    combos = [(sp, b, L) for sp in (0, 20) for b in (0, 2, 8, 9, 12) for L in (8, 12, 15, 16, 19, 20, 21, 24, 30)][::3]
    B = [rnd.randbytes(24) for _ in combos]
    C = [rnd.randbytes(40) for _ in combos]
    z16 = b"\0" * 16
    buf = bytearray(rnd.randbytes(8))
    for i in range(len(combos)):
        buf += B[i] + C[i] + z16
    for i in range(len(combos)):
        buf += B[i] + rnd.randbytes(4) + z16
    buf += b"\0" * 66000
    for i, (sp, b, L) in enumerate(combos):
        buf += rnd.randbytes(sp) + (B[i][24 - b:] if b else b"") + C[i][:L] + z16
    # ... and copies the parser ENTERS `off` bytes in, because its step has grown past `off` over a long spacer of noise: the match is
    # extended backwards to the copy's start, and with off >= 8 the 8 bytes a lane fetches behind its position do not say how far.
    # The spacer's length puts a visit of the schedule (fast.h:75-82: s_0 = 1, s_j = (63 + j) >> 6) exactly `off` bytes into the copy.
    def visit_off(v):
        return sum(1 if j == 0 else (63 + j) >> 6 for j in range(v))
    for k, (off, L) in enumerate([(9, 20), (9, 19), (10, 21), (12, 20), (9, 26), (8, 20), (7, 20)]):
        v = 64 * (off + 1) + 2                                 # the step in front of visit v is off + 1
        spacer = 1 + visit_off(v) - off                        # behind a match that ends at `ip` the run starts at ip + 1
        buf += b"\x55" + rnd.randbytes(spacer - 1) + C[k][:L] + z16
    buf += rnd.randbytes(100)
    return bytes(buf)


# ---- frames (lizard_frame.h:111-125 LizardF_preferences_t == LizardGPU_framePrefs_t) -------------------
class FrameInfo(ctypes.Structure):
    _fields_ = [("blockSizeID", ctypes.c_uint), ("blockMode", ctypes.c_uint), ("contentChecksumFlag", ctypes.c_uint),
                ("frameType", ctypes.c_uint), ("contentSize", ctypes.c_ulonglong), ("reserved", ctypes.c_uint * 2)]


class FramePrefs(ctypes.Structure):
    _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", ctypes.c_int), ("autoFlush", ctypes.c_uint),
                ("reserved", ctypes.c_uint * 4)]


FRAME_BLOCK_SIZES = {0: 128 << 10, 1: 128 << 10, 2: 256 << 10, 3: 1 << 20, 4: 4 << 20, 5: 16 << 20, 6: 64 << 20, 7: 256 << 20}


def frame_prefs(level, bsid=0, checksum=0, content_size=0, block_mode=1):
    p = FramePrefs()
    p.frameInfo.blockSizeID = bsid
    p.frameInfo.blockMode = block_mode
    p.frameInfo.contentChecksumFlag = checksum
    p.frameInfo.contentSize = content_size
    p.compressionLevel = level
    return p


FRAME_CASES = [
    # (name, corpus case, level, blockSizeID, checksum, contentSize flag)
    ("empty_default", "gen0_p0.5", 10, 0, 0, 0),
    ("empty_crc", "gen0_p0.5", 10, 2, 1, 1),
    ("one_byte", "gen1_p0.5", 10, 0, 1, 0),
    ("ragged_128k", "gen131073_p0.5", 10, 1, 1, 0),
    ("ragged_128k_L30", "gen131073_p0.5", 30, 1, 1, 1),
    ("256k_bs256k", "gen262144_p0.5", 10, 2, 1, 0),
    ("zeros_default_level", "zeros300k", 0, 1, 0, 0),          # level 0 -> 17 (lizard_compress.c:303-308)
    ("noise_raw_blocks", "random256k", 10, 1, 1, 0),
    ("text_L21", "text", 21, 1, 0, 1),
    ("text_L41_big_id", "text", 41, 4, 1, 0),                  # optimal block size id shrinks to the input
    ("alpha4_L13", "alpha4", 13, 1, 1, 0),
    ("period65_L11", "period65", 11, 1, 0, 0),
]


def _reference_frame_fn(ref):
    ref.LizardF_compressFrameBound.restype = ctypes.c_size_t
    ref.LizardF_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
    ref.LizardF_compressFrame.restype = ctypes.c_size_t
    ref.LizardF_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    return ref


def reference_frame(data, prefs):
    """LizardF_compressFrame of the compiled reference (zero-state build); None when oracle/_ref is absent."""
    ref = reference()
    if ref is None:
        return None
    _reference_frame_fn(ref)
    cap = ref.LizardF_compressFrameBound(len(data), ctypes.byref(prefs))
    dst = ctypes.create_string_buffer(cap)
    n = ref.LizardF_compressFrame(dst, cap, data, len(data), ctypes.byref(prefs))
    assert n <= cap, "reference frame error %d" % (n - (1 << 64))
    return dst.raw[:n]


def compose_frame(data, level, bsid, checksum, content_size_flag, compress_block):
    """Restatement of LizardF_compressFrame (lib/lizard_frame.c:260-316, :403-424, :456-469, :651-658) for
    independent blocks, on top of any block compressor `compress_block(bytes, level) -> bytes`."""
    import struct
    import xxhash
    n = len(data)
    proposed, req = 1, bsid
    while req > proposed:                                   # LizardF_optimalBSID
        if n <= FRAME_BLOCK_SIZES[proposed]:
            req = proposed
            break
        proposed += 1
    bsid = req if req else 1
    bs = FRAME_BLOCK_SIZES[bsid]
    lvl = min(level, 49)
    if lvl < 10:
        lvl = 17
    hdr = bytes([(1 << 6) + (1 << 5) + (checksum << 2) + ((1 if (content_size_flag and n) else 0) << 3), bsid << 4])
    if content_size_flag and n:
        hdr += struct.pack("<Q", n)
    out = struct.pack("<I", 0x184D2206) + hdr + bytes([(xxhash.xxh32(hdr, seed=0).intdigest() >> 8) & 255])
    for off in range(0, n, bs):
        blk = data[off:off + bs]
        c = compress_block(blk, lvl)
        if len(c) > len(blk) - 1 and len(blk) != 1:      # 1-byte blocks: the room test of lizard_compress.c:238 wraps
            out += struct.pack("<I", len(blk) | 0x80000000) + blk
        else:
            out += struct.pack("<I", len(c)) + c
    out += struct.pack("<I", 0)
    if checksum:
        out += struct.pack("<I", xxhash.xxh32(data, seed=0).intdigest())
    return out
