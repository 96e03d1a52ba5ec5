"""GPU: LizardGPU_setMemoryBudget / _memoryInUse / _trim (include/lizard_amd.h, "Device memory") — a library that replaces liblizard
must not help itself to half of a device it shares.  Under a 4 GiB budget every GPU level still produces the
reference's bytes (tables and hashChain work areas get fewer slots than resident waves; what another level left behind is given up
first), the context never holds more than the budget, and what the driver says is gone from the device agrees."""
import ctypes

import numpy as np
import pytest

import util

C = ctypes
GIB = 1 << 30
LEVELS = [10, 30, 11, 31, 21, 41, 22, 42, 13, 14, 15, 16, 17, 34, 35, 36, 37, 38, 12, 32, 33, 20, 40]


@pytest.fixture(scope="module")
def L():
    from lizard_amd import _lib
    lib = _lib.lib()
    lib.LizardGPU_setMemoryBudget.argtypes = [C.c_size_t]; lib.LizardGPU_setMemoryBudget.restype = C.c_int
    lib.LizardGPU_memoryBudget.restype = C.c_size_t
    lib.LizardGPU_memoryInUse.restype = C.c_size_t
    lib.LizardGPU_trim.restype = C.c_int
    lib.LizardGPU_degradedCalls.restype = C.c_ulonglong
    yield lib
    lib.LizardGPU_setMemoryBudget(0)


@pytest.mark.gpu
def test_every_level_under_a_4_gib_budget(L):
    import torch
    from lizard_amd import api
    bs, nb = 262144, 80
    data = b"".join(util.datagen(bs, 0.5, 0.0, 1000 + b) for b in range(nb - 1)) + util.datagen(70001, 0.5, 0.0, 7)   # ragged last block
    src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    stride = (api.Lizard_compressBound(bs) + 63) & ~63
    dst = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert L.LizardGPU_setMemoryBudget(1 << 20) < 0 and b"minimum" in L.LizardGPU_lastError()          # below one scratch arena: refused
    assert L.LizardGPU_setMemoryBudget(4 * GIB) == 0 and L.LizardGPU_memoryBudget() == 4 * GIB
    assert L.LizardGPU_memoryInUse() == 0                                                               # the call released everything
    free0 = torch.cuda.mem_get_info()[0]
    peak = 0
    for level in LEVELS:
        api.compress_blocks_device(src, bs, level, dst=dst, sizes=sizes)
        torch.cuda.synchronize()
        used = L.LizardGPU_memoryInUse()
        peak = max(peak, used)
        assert 0 < used <= 4 * GIB, (level, used)
        assert free0 - torch.cuda.mem_get_info()[0] <= 4 * GIB + (256 << 20), level                     # what the driver says is gone
        out, sz = dst.cpu().numpy(), sizes.cpu().numpy()
        for b in range(nb):
            blk = data[b * bs:(b + 1) * bs]
            want = util.oracle_compress(blk, level)
            assert sz[b] == len(want) and out[b * stride:b * stride + sz[b]].tobytes() == want, (level, b)
    # the host-buffer entry (chunks shrink to budget / 64) and the one-block entry under the same budget
    hb = np.frombuffer(data[:40 * bs], dtype=np.uint8)
    outs = api.compress_blocks(hb.tobytes(), bs, 21)
    for b in (0, 17, 39):
        assert outs[b] == util.oracle_compress(data[b * bs:(b + 1) * bs], 21)
    assert api.Lizard_compress(data[:bs], 36) == util.oracle_compress(data[:bs], 36)
    assert L.LizardGPU_memoryInUse() <= 4 * GIB
    assert peak > 3 * GIB // 2                                                                          # the budget was actually felt (scratch alone is 2.7 GB)
    # trim: everything but the context's own scratch arena goes back
    assert L.LizardGPU_trim() == 0
    scratch_only = L.LizardGPU_memoryInUse()
    assert 0 < scratch_only < 3 * GIB
    assert api.Lizard_compress(data[:bs], 11) == util.oracle_compress(data[:bs], 11)                    # and the library works on
    assert L.LizardGPU_setMemoryBudget(0) == 0 and L.LizardGPU_memoryBudget() == 0


@pytest.mark.gpu
def test_degraded_calls_are_counted(L):
    from lizard_amd import api
    n0 = L.LizardGPU_degradedCalls()
    data = util.datagen(100000, 0.5, 0.0, 3)
    dst = C.create_string_buffer(200000)
    lib = L
    lib.Lizard_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    assert lib.Lizard_compress(data, dst, len(data), 200000, 19) == 0        # level 19 (optimal parser) has no GPU kernel: 0, counted
    assert lib.Lizard_compress(data, dst, len(data), 200000, 23) == 0
    assert L.LizardGPU_degradedCalls() == n0 + 2
    assert lib.Lizard_compress(data, dst, len(data), 200000, 10) > 0         # a supported level is not counted
    assert L.LizardGPU_degradedCalls() == n0 + 2


@pytest.mark.gpu
def test_host_path_with_many_chunks_at_the_smallest_budget(L):
    """at the smallest budget LizardGPU_setMemoryBudget accepts (one scratch arena + 256 MiB) the host pipeline's three
    stages must still fit: its chunks are sized from the room the budget leaves beside the arena (a sixteenth each), not from the
    budget as a whole.  ~100 MiB of blocks = more than three chunks of 16 MiB; every block against the oracle; and small launches of
    a table level on a second stream while the context's own tables hold the room fall back to the context's arena instead of
    failing (lizard_gpu.hip, own_arena_after_all)."""
    import torch
    from lizard_amd import api
    L.LizardGPU_residentWaves.restype = C.c_int
    cus = L.LizardGPU_residentWaves() // 13
    floor = cus * 16 * 5 * (131072 + 32) + (256 << 20)                       # one scratch arena (16 slots of 5 padded sub-blocks per CU) + 256 MiB
    assert L.LizardGPU_setMemoryBudget(floor - 1) < 0
    assert L.LizardGPU_setMemoryBudget(floor) == 0
    bs, nb = 262144, 400
    data = b"".join(util.datagen(bs, 0.5, 0.0, 5000 + b) for b in range(nb))
    outs = api.compress_blocks(data, bs, 10)
    assert len(outs) == nb
    for b in range(0, nb, 7):
        assert outs[b] == util.oracle_compress(data[b * bs:(b + 1) * bs], 10), b
    assert L.LizardGPU_memoryInUse() <= floor
    assert L.LizardGPU_setMemoryBudget(0) == 0
    # An extra arena whose tables no longer fit: room for two scratch arenas and 300 MiB.  Two streams of small table-less launches
    # (level 10) make the second arena; level 11's tables then take the rest of the room on the context's own arena; a small level-11
    # launch that is routed to the extra arena finds no room for ITS tables and must run on the context's own arena instead of failing.
    arena = floor - (256 << 20)
    assert L.LizardGPU_setMemoryBudget(2 * arena + (300 << 20)) == 0
    L.LizardGPU_arenasInUse.restype = C.c_int
    src = torch.from_numpy(np.frombuffer(data[:64 * bs], dtype=np.uint8).copy()).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):                                                       # (the second launch must find the first still running)
        with torch.cuda.stream(s1):
            api.compress_blocks_device(src, bs, 10)
        with torch.cuda.stream(s2):
            api.compress_blocks_device(src, bs, 10)
    torch.cuda.synchronize()
    assert L.LizardGPU_arenasInUse() == 2
    with torch.cuda.stream(s1):
        first = api.compress_blocks_device(src, bs, 11)
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(s1):
            a = api.compress_blocks_device(src, bs, 11)
        with torch.cuda.stream(s2):
            b2 = api.compress_blocks_device(src[:8 * bs], bs, 11)            # round 5: LIZARDGPU_ERR_NOMEM here
        torch.cuda.synchronize()
        assert torch.equal(a[1], first[1]) and torch.equal(b2[1], first[1][:8])
    want = util.oracle_compress(data[:bs], 11)
    assert int(b2[1][0]) == len(want) and b2[0][:len(want)].cpu().numpy().tobytes() == want
    assert L.LizardGPU_memoryInUse() <= 2 * arena + (300 << 20)
    assert L.LizardGPU_setMemoryBudget(0) == 0
