"""Multi-GPU side of the C library (include/lizard_amd.h, "several GPUs"): the partition / offset code on the CPU
(no GPU, no RCCL needed), agreement with the torch.distributed form (lizard_amd/sharding.py), and on the GPU box the
single-process entry LizardGPU_compressBlocks_sharded and the per-process rank communicator with ONE rank — the same
code path the N-rank runs take, RCCL included (ncclCommInitAll / ncclCommInitRank / ncclAllGather on a one-rank
communicator)."""
import ctypes
import random

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
    from lizard_amd.sharding import shard_range
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


@pytest.mark.gpu
def test_sharded_single_process_one_device(lib):
    import torch
    from lizard_amd import api
    bs, nb, level = 65536, 301, 10
    host = np.frombuffer(util.datagen(bs * nb - 1234, 0.5, 0.0, 41), dtype=np.uint8)
    src = torch.from_numpy(host.copy()).cuda()
    stride = (api.Lizard_compressBound(bs) + 63) & ~63
    dst = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
    allsz = torch.zeros(nb, dtype=torch.int32, device="cuda")
    offs = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
    P = ctypes.c_void_p * 1
    rc = lib.LizardGPU_compressBlocks_sharded(1, None, P(src.data_ptr()), nb, bs, bs - 1234, P(dst.data_ptr()), stride,
                                              P(allsz.data_ptr()), P(offs.data_ptr()), level)
    assert rc == 0, lib.LizardGPU_lastError()
    sz = allsz.cpu().numpy().astype(np.int64)
    assert np.array_equal(offs.cpu().numpy(), np.concatenate([[0], np.cumsum(sz)]))
    out = dst.cpu().numpy()
    for b in (0, 1, 150, nb - 1):
        want = util.oracle_compress(host[b * bs:(b + 1) * bs].tobytes(), level)
        assert out[b * stride:b * stride + sz[b]].tobytes() == want, b


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
