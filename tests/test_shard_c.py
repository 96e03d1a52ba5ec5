"""Multi-GPU side of the C library (include/lizard_amd.h, "several GPUs").
CPU: the partition / offset code, agreement with the torch.distributed form (tests/torch_sharding.py), and the library's
exchange logic itself (lizard_amd/csrc/lizard_shard_core.h) with 1..8 ranks, equal and ragged partitions, over a fake
shared-memory transport (tests/shard_fake.cpp): the in-place all-gather, the per-root broadcasts, the grouped
single-thread form and its error path.
GPU: the single-process entry LizardGPU_compressBlocks_sharded over 1/2/4/8 devices (as many as the box has) with equal and
ragged partitions through RCCL; the per-process rank communicator; and TWO ranks on ONE device through an injected
transport (LizardGPU_setCollectives) so that the N > 1 code of the library runs on a one-GPU box too."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

import util


@pytest.fixture(scope="module")
def lib():
    from lizard_amd import _lib
    return _lib.lib() if _lib.os.path.exists(_lib.LIB_PATH) else ctypes.CDLL(_lib.build())


def c_range(lib, n, r, w):
    first, count = ctypes.c_size_t(), ctypes.c_size_t()
    lib.LizardGPU_shardRange(n, r, w, ctypes.byref(first), ctypes.byref(count))
    return first.value, count.value


def test_partition_is_contiguous_balanced_and_matches_python(lib):
    from torch_sharding import shard_range
    rnd = random.Random(3)
    for n, w in [(1, 1), (8, 8), (9, 8), (65536, 8), (65537, 8), (4096, 3), (7, 7)] + [(rnd.randrange(8, 10 ** 6), rnd.randrange(1, 9)) for _ in range(50)]:
        nxt = 0
        for r in range(w):
            first, count = c_range(lib, n, r, w)
            assert (first, count) == shard_range(n, r, w)
            assert first == nxt and count in (n // w, n // w + 1)
            nxt = first + count
        assert nxt == n


def test_offsets_from_gathered_sizes_world2(lib):
    """World size 2 with the gather stubbed: each 'rank' fills its slice of the all-sizes array in place (what the RCCL
    all-gather leaves behind), then the prefix sum gives every rank the same global offsets."""
    rnd = np.random.RandomState(1)
    for n in (2, 3, 1000, 65536, 65537):
        sizes = rnd.randint(1, 262158, size=n).astype(np.uint32)
        allsz = np.zeros(n, dtype=np.uint32)
        for r in range(2):
            first, count = c_range(lib, n, r, 2)
            allsz[first:first + count] = sizes[first:first + count]
        off = np.zeros(n + 1, dtype=np.uint64)
        lib.LizardGPU_offsetsFromSizes(allsz.ctypes.data_as(ctypes.c_void_p), n, off.ctypes.data_as(ctypes.c_void_p))
        want = np.concatenate([[0], np.cumsum(sizes.astype(np.uint64))])
        assert np.array_equal(off, want)


def test_sharded_entry_refuses_bad_arguments_without_gpu(lib):
    assert lib.LizardGPU_compressBlocks_sharded(0, None, None, 10, 4096, 4096, None, 5000, None, None, 10) == -3
    assert lib.LizardGPU_setDevice(-1) == -3 and lib.LizardGPU_setDevice(10 ** 6) == -3
    assert b"LizardGPU_setDevice" in lib.LizardGPU_lastError()
    assert lib.LizardGPU_setDevice(0) == 0 and lib.LizardGPU_lastError() == b""


def test_exchange_logic_with_fake_transport_1_to_8_ranks(tmp_path):
    """lizard_shard_core.h (what LizardGPU_gatherSizes_device and LizardGPU_compressBlocks_sharded run) over a shared-memory
    transport: one thread per rank and the grouped single-thread form, equal and ragged partitions, an injected error."""
    exe = str(tmp_path / "shard_fake")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-pthread", "-o", exe,
                           os.path.join(util.ROOT, "tests", "shard_fake.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ALL OK"), out.stdout[-2000:]
    for want in ("ranks 2 blocks 1000 (equal): ok", "ranks 2 blocks 1001 (ragged): ok", "ranks 3 blocks 4099 (ragged): ok",
                 "ranks 3 blocks 12 (equal): ok", "ranks 8 blocks 524291 (ragged): ok"):
        assert want in out.stdout


def _sharded_case(lib, devices, nb, bs, tail, level):
    """LizardGPU_compressBlocks_sharded over `devices` (rank r on devices[r]); returns per-rank (sizes, offsets) and rank 0's shard check."""
    import torch
    from lizard_amd import api
    n = len(devices)
    host = np.frombuffer(util.datagen(bs * nb - tail, 0.5, 0.0, 41 + nb), dtype=np.uint8)
    stride = (api.Lizard_compressBound(bs) + 63) & ~63
    srcs, dsts, alls, offs = [], [], [], []
    for r, d in enumerate(devices):
        first, count = c_range(lib, nb, r, n)
        dev = torch.device("cuda", d)
        srcs.append(torch.from_numpy(host[first * bs:(first + count) * bs].copy()).to(dev))
        dsts.append(torch.empty(count * stride, dtype=torch.uint8, device=dev))
        alls.append(torch.full((nb,), -1, dtype=torch.int32, device=dev))
        offs.append(torch.zeros(nb + 1, dtype=torch.int64, device=dev))
    P = ctypes.c_void_p * n
    devs = (ctypes.c_int * n)(*devices)
    rc = lib.LizardGPU_compressBlocks_sharded(n, devs, P(*[t.data_ptr() for t in srcs]), nb, bs, bs - tail, P(*[t.data_ptr() for t in dsts]), stride,
                                              P(*[t.data_ptr() for t in alls]), P(*[t.data_ptr() for t in offs]), level)
    assert rc == 0, lib.LizardGPU_lastError()
    sz0 = alls[0].cpu().numpy().astype(np.int64)
    assert (sz0 > 0).all()
    for r in range(n):                                               # every rank holds all sizes and the same global offsets
        assert np.array_equal(alls[r].cpu().numpy().astype(np.int64), sz0), r
        assert np.array_equal(offs[r].cpu().numpy(), np.concatenate([[0], np.cumsum(sz0)])), r
    for r in range(n):                                               # and the bytes of a few blocks of every shard
        first, count = c_range(lib, nb, r, n)
        out = dsts[r].cpu().numpy()
        for b in {first, first + count // 2, first + count - 1}:
            want = util.oracle_compress(host[b * bs:(b + 1) * bs].tobytes(), level)
            assert out[(b - first) * stride:(b - first) * stride + sz0[b]].tobytes() == want, (r, b)
    return sz0


@pytest.mark.gpu
@pytest.mark.parametrize("ndev", [1, 2, 4, 8])
def test_sharded_single_process_n_devices(lib, ndev):
    """Equal and ragged partitions over the first `ndev` devices through RCCL (ncclCommInitAll, grouped in-place all-gather /
    per-root broadcasts); the sizes must equal the one-device result."""
    if lib.LizardGPU_deviceCount() < ndev:
        pytest.skip(f"{lib.LizardGPU_deviceCount()} device(s) visible")
    one = {}
    for nb, tail in ((304, 1234), (301, 0), (ndev, 77)):
        if nb < ndev:
            continue
        one[nb] = _sharded_case(lib, [0], nb, 65536, tail, 10)
        got = _sharded_case(lib, list(range(ndev)), nb, 65536, tail, 10)
        assert np.array_equal(got, one[nb]), (ndev, nb)
    assert lib.LizardGPU_rcclShared() in (0, 1)


class _HipCopyTransport:
    """A collective table in Python for ranks that live in ONE process: calls are queued between groupStart and groupEnd and
    carried out with device-to-device copies on the receiving rank's stream (what RCCL does between devices, here between
    buffers).  `comm` is the rank index (LizardGPU_setCollectives contract)."""
    AG = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p)
    BC = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
    GR = ctypes.CFUNCTYPE(ctypes.c_int)

    class Table(ctypes.Structure):
        pass

    def __init__(self, n_ranks):
        self.n, self.depth, self.queue, self.calls = n_ranks, 0, {}, {"ag": 0, "bc": 0, "groups": 0}
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        self.cb = (self.AG(self.all_gather), self.BC(self.broadcast), self.GR(self.start), self.GR(self.end))
        T = type("T", (ctypes.Structure,), {"_fields_": [("ag", self.AG), ("bc", self.BC), ("gs", self.GR), ("ge", self.GR)]})
        self.table = T(*self.cb)

    def all_gather(self, send, recv, count, comm, stream):
        self.calls["ag"] += 1
        self.queue.setdefault(int(comm or 0), []).append(("ag", send, recv, count, -1, stream))
        return 0 if self.depth == 1 else -9

    def broadcast(self, send, recv, count, root, comm, stream):
        self.calls["bc"] += 1
        self.queue.setdefault(int(comm or 0), []).append(("bc", send, recv, count, root, stream))
        return 0 if self.depth == 1 else -9

    def start(self):
        self.depth += 1
        return 0

    def end(self):
        self.depth -= 1
        self.calls["groups"] += 1
        seqs = [self.queue.get(r, []) for r in range(self.n)]
        self.queue = {}
        if len({len(q) for q in seqs}) != 1:
            return -9
        # sources are read before any rank's stream runs ahead: the test synchronises the device first (sizes are complete)
        import torch
        torch.cuda.synchronize()
        for ops in zip(*seqs):
            kind, _, _, count, root, _ = ops[0]
            if any((o[0], o[3], o[4]) != (kind, count, root) for o in ops):
                return -9
            for d, (_, _, recv, _, _, stream) in enumerate(ops):
                for s_rank in (range(self.n) if kind == "ag" else [root]):
                    src = ops[s_rank][1]
                    dst = recv + 4 * count * s_rank if kind == "ag" else recv
                    if dst != src and self.hip.hipMemcpyAsync(dst, src, 4 * count, 3, stream) != 0:
                        return -2
        return 0


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 3])
def test_two_and_three_ranks_on_one_device_through_an_injected_transport(lib, nranks):
    """The N > 1 code of LizardGPU_compressBlocks_sharded (per-rank launches, grouped exchange, per-rank offset scans) on a
    one-GPU box: every rank lives on device 0 and the exchange runs over an injected transport."""
    tr = _HipCopyTransport(nranks)
    lib.LizardGPU_setCollectives.argtypes = [ctypes.c_void_p]
    assert lib.LizardGPU_setCollectives(ctypes.byref(tr.table)) == 0
    try:
        for nb, tail in ((300, 99), (301, 0), (nranks, 5)):
            one = None
            got = _sharded_case(lib, [0] * nranks, nb, 65536, tail, 10)
            lib.LizardGPU_setCollectives(None)
            one = _sharded_case(lib, [0], nb, 65536, tail, 10)
            lib.LizardGPU_setCollectives(ctypes.byref(tr.table))
            assert np.array_equal(got, one), (nranks, nb)
        assert tr.calls["groups"] == 3 and tr.calls["ag"] + tr.calls["bc"] > 0
        assert (tr.calls["ag"] > 0) == (any(nb % nranks == 0 for nb in (300, 301, nranks)))
    finally:
        lib.LizardGPU_setCollectives(None)


@pytest.mark.gpu
def test_rank_communicator_one_rank(lib):
    """The torchrun form with world size 1: unique id -> ncclCommInitRank -> gather of sizes + offsets on a stream."""
    import torch
    uid = ctypes.create_string_buffer(128)
    assert lib.LizardGPU_commUniqueId(uid) == 0, lib.LizardGPU_lastError()
    assert lib.LizardGPU_commInitRank(uid, 1, 0) == 0, lib.LizardGPU_lastError()
    n = 1000
    local = torch.arange(1, n + 1, dtype=torch.int32, device="cuda")
    allsz = torch.zeros(n, dtype=torch.int32, device="cuda")
    offs = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.LizardGPU_gatherSizes_device(local.data_ptr(), n, allsz.data_ptr(), offs.data_ptr(), ctypes.c_void_p(st))
    assert rc == 0, lib.LizardGPU_lastError()
    torch.cuda.synchronize()
    assert torch.equal(allsz, local)
    assert offs[-1].item() == n * (n + 1) // 2 and offs[1].item() == 1
    assert lib.LizardGPU_commDestroy() == 0
