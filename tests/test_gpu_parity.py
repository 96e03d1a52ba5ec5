"""GPU (-m gpu): parity of the HIP path, called through the C ABI (liblizard_amd.so), against
 * the oracle restatement (oracle/liblizard_oracle.so) on the same seeded inputs,
 * the golden vectors recorded from the compiled reference (tests/golden/reference_vectors.json),
 * the prebuilt reference library oracle/_ref/*.so when it travelled to this box,
 * and, at BASELINE.json sizes, size-independent properties (known-answer chained XXH64 over 64 MiB of
   datagen P50; reference decoder round trip; run-to-run determinism).
Bit-exact is the bar: this is integer/byte work."""
import ctypes
import json
import os

import numpy as np
import pytest
import xxhash

import util
from tools import datagen as tools_datagen

pytestmark = pytest.mark.gpu

with open(os.path.join(util.GOLDEN_DIR, "reference_vectors.json")) as f:
    GOLDEN = json.load(f)


@pytest.fixture(scope="module")
def L():
    from lizard_amd import _lib
    return _lib.lib()      # raises loudly if the HIP extension is not built


def gpu_levels(L):
    return [l for l in GOLDEN["levels"] if L.LizardGPU_levelSupported(l)]


def gpu_compress(L, data, level, cap=None):
    return util.compress_with(L.Lizard_compress, data, level, cap)


def test_library_loaded_is_in_tree(L):
    from lizard_amd import _lib
    assert os.path.samefile(os.path.dirname(_lib.LIB_PATH), os.path.join(util.ROOT, "lizard_amd"))
    assert L.LizardGPU_residentWaves() >= 256


def test_corpus_vs_golden_and_oracle(L):
    """Every corpus case (empty, ragged, degenerate, incompressible...) one block per call."""
    levels = gpu_levels(L)
    assert 10 in levels
    for level in levels:
        for name, data in util.corpus():
            out, r = gpu_compress(L, data, level)
            g = GOLDEN["cases"][name]["out"][str(level)]
            assert r == g["size"], (name, level, r, g["size"])
            assert util.sha(out) == g["sha256"], (name, level)
            assert out == util.oracle_compress(data, level), (name, level)


def test_long_range_corpus(L):
    """Window-edge / position-wrap adversaries, sparse long-distance repeats, multi-MiB blocks — at every GPU level."""
    for level in gpu_levels(L):
        for name, data in util.corpus_long():
            out, r = gpu_compress(L, data, level)
            g = GOLDEN["cases"][name]["out"][str(level)]
            assert r == g["size"] and util.sha(out) == g["sha256"], (name, level)


def test_long_match_crosses_sweep_intervals(L):
    """Round-3 soak finding (tests/test_emulator.py::test_emulated_long_match_crosses_sweep_intervals): behind a match that moves ip
    across several sweep intervals of the 17-bit LDS table, a slot stamped dead before the jump must still read as dead.  One block
    per call and in a batch (every kind of wave), at every GPU level."""
    from test_emulator import _long_match_case
    from lizard_amd import api
    cases = [_long_match_case(p, r, p + r) for p, r in ((40000, 80000), (33000, 66000), (1000, 120000), (70000, 40000), (40000, 200000))]
    for level in gpu_levels(L):
        for i, data in enumerate(cases):
            out, r = gpu_compress(L, data, level)
            assert out == util.oracle_compress(data, level), ("one block", level, i)
    blob = b"".join(c[:131072].ljust(131072, b"\x55") for c in cases) * 8
    for level in (10, 30, 11, 21):
        for i, o in enumerate(api.compress_blocks(blob, 131072, level)):
            assert o == util.oracle_compress(blob[i * 131072:(i + 1) * 131072], level), ("batch", level, i)


def test_blocks_above_4mib(L):
    """Every size the reference takes (lib/lizard_compress.h:121) at the fast and priceFast levels: their tables keep positions
    modulo 2^17 / 2^22 / 2^24 and sweep.  A 17 MiB block crosses every one of those widths; the one-block path runs a single
    wave, so the bulk of it is a run (cheap to parse) between two copies of the same data exactly one position width + 100
    apart.  hashChain keeps full positions; its per-wave work area grows with the block."""
    data = util.datagen((5 << 20) + 123, 0.5, 0.0, 31) + bytes(300000) + util.datagen(1 << 20, 0.2, 0.0, 32)
    for level in (10, 30, 21, 11, 22, 13, 20):
        out, r = gpu_compress(L, data, level)
        assert out == util.oracle_compress(data, level), level
    x = util.datagen(600000, 0.5, 0.0, 9)
    big = x + bytes((1 << 24) + 100 - len(x)) + x + util.datagen(300000, 0.4, 0.0, 10)       # 17.1 MiB
    for level in (11, 21, 41, 22, 31, 10, 13, 36, 20, 40, 12):
        out, r = gpu_compress(L, big, level)
        assert out == util.oracle_compress(big, level), level
    assert L.LizardGPU_maxBlockSize(11) == L.LizardGPU_maxBlockSize(10) == L.LizardGPU_maxBlockSize(42) == L.LizardGPU_maxBlockSize(13) == 0x7E000000


def test_sweeps_due_inside_long_matches(L):
    """The shape of the round-3 soak finding at EVERY modular-position table, on the GPU: long matches (a run, a periodic stretch:
    one match per sub-block) carry the position across the table's sweep points, and what follows probes slots whose entries are
    one position width + 100 old — 100 modulo the width — with the same bytes and the same check bits.  Levels 10/30: 2^17,
    sweep every 2^15 (the original bug); 11/31: 2^22, sweep every 2^20; 20/40/21/41/22/42: 2^24, sweep every 2^22."""
    import random
    rnd = random.Random(17)

    def case(width, filler):
        x = util.datagen(600000 if width > (1 << 17) else 50000, 0.5, 0.0, 5)
        gap = width + 100 - len(x)
        if filler == "run":
            mid = bytes(gap)
        else:                                                   # period below the 64 KiB window: one long match per sub-block
            pat = rnd.randbytes(filler)
            mid = (pat * (gap // filler + 1))[:gap]
        return x + mid + x + util.datagen(200000, 0.4, 0.0, 6)

    for width, levels in ((1 << 17, (10, 30)), (1 << 22, (11, 31, 10))):
        for filler in ("run", 40000, 65535):
            data = case(width, filler)
            for level in levels:
                out, r = gpu_compress(L, data, level)
                assert out == util.oracle_compress(data, level), (width, filler, level)
    # priceFast: offsets up to 4 MiB — a 3 MiB pattern repeated (every sub-block one off24 match across the sweeps at 2^22, 2^23, ...)
    pat = rnd.randbytes(3 << 20)
    data = pat * 6 + util.datagen(100000, 0.5, 0.0, 7)                       # 18.1 MiB: beyond 2^24 positions
    for level in (21, 42, 20):
        out, r = gpu_compress(L, data, level)
        assert out == util.oracle_compress(data, level), ("3 MiB period", level)


def test_frame_style_capacity(L):
    """maxDstSize = srcSize-1 (reference lib/lizard_frame.c:461): identical bytes when it fits, 0 when not."""
    for level in [l for l in (10, 21, 30) if L.LizardGPU_levelSupported(l)]:
        for name, data in util.corpus(small=True):
            if len(data) < 2:
                continue
            out, r = gpu_compress(L, data, level, cap=len(data) - 1)
            g = GOLDEN["frame_style"][name][str(level)]
            assert r == g["size"] and util.sha(out) == g["sha256"], (name, level)
    data = dict(util.corpus(small=True))["gen65537_p0.5"]
    exact = len(util.oracle_compress(data, 10))
    for cap in (1, 2, exact - 1, exact, exact + 1):
        out, r = gpu_compress(L, data, 10, cap=cap)
        want, rw = util.compress_with(util.oracle().lzo_compress, data, 10, cap=cap)
        assert r == rw and out == want[:r], cap
        assert r <= cap


def test_batch_host_ragged(L):
    """Batched host entry: many blocks, ragged last block, several block sizes."""
    from lizard_amd import api
    data = util.datagen(3 * 262144 + 12345, 0.5, 0.0, 77) + bytes(70000) + util.datagen(200000, 0.1, 0.0, 5)
    for level in gpu_levels(L):
        for bs in (4096, 65536, 131072, 262144, 1 << 20):
            outs = api.compress_blocks(data, bs, level)
            assert len(outs) == (len(data) + bs - 1) // bs
            for i, o in enumerate(outs):
                assert o == util.oracle_compress(data[i * bs:(i + 1) * bs], level), (level, bs, i)


@pytest.mark.parametrize("key", sorted(k for k in GOLDEN["p50_64m"] if k.startswith("L")))
def test_known_answers_p50_64m_device_path(L, key):
    """SURVEY.md §8c known answers through the device-resident batch entry (bench.py's path)."""
    import torch
    from lizard_amd import api
    level, bs = int(key[1:].split("_B")[0]), int(key.split("_B")[1])
    if not L.LizardGPU_levelSupported(level):
        pytest.skip("level not on the GPU path yet")
    N = 64 << 20
    host = np.frombuffer(util.datagen(N, 0.5, 0.0, 0), dtype=np.uint8)
    src = torch.from_numpy(host.copy()).cuda()
    dst, sizes, stride = api.compress_blocks_device(src, bs, level)
    torch.cuda.synchronize()
    sz = sizes.cpu().numpy().astype(np.int64)
    out = dst.cpu().numpy()
    assert int(sz.sum()) == GOLDEN["p50_64m"][key]["sum"]
    assert list(sz[:8]) == GOLDEN["p50_64m"][key]["first_sizes"]
    h = 0
    for i in range(N // bs):
        h = xxhash.xxh64(out[i * stride:i * stride + sz[i]].tobytes(), seed=h).intdigest()
    assert "%016x" % h == GOLDEN["p50_64m"][key]["xxh64_chain"]
    # determinism: a second launch produces the same bytes
    dst2, sizes2, _ = api.compress_blocks_device(src, bs, level)
    torch.cuda.synchronize()
    assert torch.equal(sizes, sizes2)
    sel = torch.arange(stride, device="cuda")[None, :] < sizes[:, None]
    assert torch.equal(dst.view(-1, stride)[sel], dst2.view(-1, stride)[sel])


def test_hashchain_many_small_blocks_reuse_slots(L):
    """The hashChain waves keep bins, links and the chain array in a per-wave global slot that is never cleared, and borrow
    their LDS region from a pool of the workgroup.  One launch over several hundred small blocks per resident wave: a
    sample of blocks from start to end must be bit-exact."""
    import torch
    from lizard_amd import api
    bs = 192
    nb = int(L.LizardGPU_residentWaves()) * 300
    rnd = np.random.RandomState(5)
    words = rnd.randint(0, 256, size=(64, 24), dtype=np.uint8)           # 64 phrases of 24 bytes: lots of matches per block
    host = words[rnd.randint(0, 64, size=nb * (bs // 24))].reshape(-1)[:nb * bs].copy()
    src = torch.from_numpy(host).cuda()
    for level in (13, 37):
        dst, sizes, stride = api.compress_blocks_device(src, bs, level)
        torch.cuda.synchronize()
        sz = sizes.cpu().numpy()
        out = dst.view(-1, stride)
        for b in list(range(0, nb, nb // 400)) + [nb - 1]:
            want = util.oracle_compress(host[b * bs:(b + 1) * bs].tobytes(), level)
            assert out[b, :int(sz[b])].cpu().numpy().tobytes() == want, (level, b)


def test_launches_on_two_streams_are_ordered(L):
    """The scratch arena, the tables and the block counter are shared by all launches of a process: a launch on
    another stream must wait (on the device) for the previous one.  Two streams, alternating, no host sync."""
    import torch
    from lizard_amd import api
    bs, nb = 65536, 3000
    a = torch.from_numpy(np.frombuffer(util.datagen(bs * nb, 0.5, 0.0, 91), dtype=np.uint8).copy()).cuda()
    b = torch.from_numpy(np.frombuffer(util.datagen(bs * nb, 0.3, 0.0, 92), dtype=np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(3):
        for src, st, level in ((a, s1, 10), (b, s2, 21), (a, s2, 13), (b, s1, 30)):
            with torch.cuda.stream(st):
                outs.append((src, level) + api.compress_blocks_device(src, bs, level))
    torch.cuda.synchronize()
    for src, level, dst, sizes, stride in outs:
        host = src.cpu().numpy()
        sz = sizes.cpu().numpy()
        for i in (0, 1, nb // 2, nb - 1):
            want = util.oracle_compress(host[i * bs:(i + 1) * bs].tobytes(), level)
            assert dst[i * stride:i * stride + int(sz[i])].cpu().numpy().tobytes() == want, (level, i)


def test_small_launches_on_several_streams_run_side_by_side(L):
    """A compress launch smaller than the machine that arrives on another stream while the context's arena is busy gets an arena
    of its own (include/lizard_amd.h, LizardGPU_arenasInUse) and runs beside the first: four streams, launches of 16-96 blocks at
    levels with LDS tables, global tables and both, queued without any host synchronisation — every block of every launch must be
    bit-exact, more than one arena must have been made, and a launch that fills the machine afterwards is still right."""
    import torch
    from lizard_amd import api
    L.LizardGPU_arenasInUse.restype = ctypes.c_int
    bs = 262144
    rnd = np.random.RandomState(3)
    streams = [torch.cuda.Stream() for _ in range(4)]
    srcs = []
    for k in range(4):
        nb = int(rnd.choice([16, 48, 96]))
        host = np.frombuffer(util.datagen(bs * nb, 0.3 + 0.15 * k, 0.0, 200 + k), dtype=np.uint8).copy()
        srcs.append((host, torch.from_numpy(host).cuda(), nb))
    torch.cuda.synchronize()
    outs = []
    for rep in range(6):
        for k, st in enumerate(streams):
            host, dev, nb = srcs[(k + rep) % 4]
            level = (10, 21, 30, 11, 41, 22)[(k + 2 * rep) % 6]
            with torch.cuda.stream(st):
                outs.append((host, nb, level) + api.compress_blocks_device(dev, bs, level))
    torch.cuda.synchronize()
    arenas = L.LizardGPU_arenasInUse()
    for host, nb, level, dst, sizes, stride in outs:
        sz = sizes.cpu().numpy()
        out = dst.cpu().numpy()
        for i in range(nb):
            want = util.oracle_compress(host[i * bs:(i + 1) * bs].tobytes(), level)
            assert out[i * stride:i * stride + int(sz[i])].tobytes() == want, (level, i)
    assert 2 <= arenas <= 4, arenas
    big = torch.from_numpy(np.frombuffer(util.datagen(bs * 4096, 0.5, 0.0, 9), dtype=np.uint8).copy()).cuda()
    with torch.cuda.stream(streams[1]):
        dst, sizes, stride = api.compress_blocks_device(big, bs, 10)
    torch.cuda.synchronize()
    host = big.cpu().numpy(); sz = sizes.cpu().numpy()
    for i in (0, 1, 2047, 4095):
        assert dst[i * stride:i * stride + int(sz[i])].cpu().numpy().tobytes() == util.oracle_compress(host[i * bs:(i + 1) * bs].tobytes(), 10)


def test_roundtrip_with_reference_decoder(L):
    """Full-size property: what the GPU writes decodes with the unmodified reference decoder."""
    ref = util.reference()
    if ref is None:
        pytest.skip("oracle/_ref not present on this box")
    from lizard_amd import api
    data = util.datagen(16 << 20, 0.5, 0.0, 3) + os.urandom(1 << 20) + bytes(1 << 20)
    for level in gpu_levels(L):
        bs = 262144
        outs = api.compress_blocks(data, bs, level)
        back = ctypes.create_string_buffer(bs)
        for i, o in enumerate(outs):
            n = ref.Lizard_decompress_safe(o, back, len(o), bs)
            assert n == len(data[i * bs:(i + 1) * bs]) and back.raw[:n] == data[i * bs:(i + 1) * bs], (level, i)


def test_device_datagen_matches_host(L):
    """bench.py's on-device input generator == the host generator == reference RDG_genBuffer."""
    import torch
    for bs, nb, p in [(262144, 70, 0.5), (4096, 300, 0.2), (65536, 65, 1.0), (1000, 64, 0.0)]:
        d = torch.empty(nb * bs, dtype=torch.uint8, device="cuda")
        tools_datagen.datagen_device(d.data_ptr(), nb, bs, p, 0.0, 1000, None)
        got = d.cpu().numpy().tobytes()
        for b in (0, 1, nb // 2, nb - 1):
            assert got[b * bs:(b + 1) * bs] == util.datagen(bs, p, 0.0, 1000 + b), (bs, b, p)


def test_lds_atomics_are_served_in_lane_order():
    """The level 10/30 round (lz_block.h, LzTab::xchg) reads and replaces the table slots of 64 positions with ONE pair of
    returning DS atomics and relies on the lanes of an instruction that hit the same dword being served in ascending lane
    order, which is what makes the result equal to the reference's sequential get-then-put (lizard_parser_fast.h:104-118).
    tests/lds_atomic_order.hip checks exactly that property on the device, 20000 random collision patterns x 64 workgroups."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lds_atomic_order")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "violations: 0 " in r.stdout, r.stdout + r.stderr


def test_lane_order_self_check_refuses_dependent_levels():
    """The library checks the lane-order property itself when a device's context is created; a device that fails is refused
    the levels whose kernels depend on it (10/30, hashChain) with a loud error, the others keep working.  The failure branch
    is forced through LIZARDGPU_FORCE_LANE_ORDER_FAILURE=1 in a fresh process (the check runs once per device)."""
    import subprocess
    import sys
    code = r'''
import ctypes, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import util
from lizard_amd import _lib
L = _lib.lib()
data = util.datagen(100000, 0.5, 0.0, 3)
for level, works in ((10, False), (30, False), (15, False), (36, False), (21, True), (11, True), (42, True)):
    out, r = util.compress_with(L.Lizard_compress, data, level)
    if works:
        assert out == util.oracle_compress(data, level), level
    else:
        assert r == 0, (level, r)
import numpy as np
src = np.zeros(4 * 65536, dtype=np.uint8); dst = np.zeros(4 * 70000, dtype=np.uint8); sz = np.zeros(4, dtype=np.uint32)
rc = L.LizardGPU_compressBlocks_host(src.ctypes.data, 4, 65536, 65536, dst.ctypes.data, 70000, sz.ctypes.data, 10)
assert rc == -7, rc
assert b"self-check" in L.LizardGPU_lastError(), L.LizardGPU_lastError()
assert L.LizardGPU_compressBlocks_host(src.ctypes.data, 4, 65536, 65536, dst.ctypes.data, 70000, sz.ctypes.data, 21) == 0
print("refused as expected")
''' % (util.ROOT, os.path.join(util.ROOT, "tests"))
    env = dict(os.environ, LIZARDGPU_FORCE_LANE_ORDER_FAILURE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "refused as expected" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert "NOT served in lane order" in r.stderr
