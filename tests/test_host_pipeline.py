"""GPU (-m gpu): the host-buffer pipeline of the library (pinned double-buffered staging, device-side compaction,
one D2H per chunk): slot and packed forms, pageable and pinned sources, one chunk and many chunks, all bit-exact
with the oracle; per-thread device selection and error text; shutdown and re-creation of the device context."""
import ctypes
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from lizard_amd import _lib
    return _lib.lib()


def packed(L, data, bs, level):
    from lizard_amd import api
    buf = np.frombuffer(data, dtype=np.uint8)
    nb = (len(data) + bs - 1) // bs
    last = len(data) - (nb - 1) * bs
    cap = nb * api.Lizard_compressBound(bs)
    out = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(nb + 1, dtype=np.uint64)
    sizes = np.zeros(nb, dtype=np.uint32)
    rc = L.LizardGPU_compressBlocks_host_packed(buf.ctypes.data, nb, bs, last, out.ctypes.data, cap, offs.ctypes.data, sizes.ctypes.data, level)
    assert rc == 0, L.LizardGPU_lastError()
    assert offs[0] == 0 and np.array_equal(np.diff(offs), sizes.astype(np.uint64))
    return [out[int(offs[i]):int(offs[i]) + int(sizes[i])].tobytes() for i in range(nb)]


def test_packed_and_slot_forms_agree_with_oracle(L):
    from lizard_amd import api
    data = util.datagen(5 * 262144 + 4321, 0.5, 0.0, 61) + bytes(100000) + os.urandom(70000)
    for level in (10, 30, 21, 13):
        for bs in (65536, 262144):
            a = packed(L, data, bs, level)
            b = api.compress_blocks(data, bs, level)
            assert a == b
            for i, o in enumerate(a):
                assert o == util.oracle_compress(data[i * bs:(i + 1) * bs], level), (level, bs, i)
    # capacity too small is refused, not overrun
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(1000, dtype=np.uint8)
    rc = L.LizardGPU_compressBlocks_host_packed(buf.ctypes.data, 4, 65536, 65536, out.ctypes.data, 1000, None, None, 10)
    assert rc == -3 and b"dstCapacity" in L.LizardGPU_lastError()


def test_pinned_source_is_read_directly(L):
    import torch
    data = util.datagen(9 * 131072 + 17, 0.5, 0.0, 62)
    pinned = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).pin_memory()
    from lizard_amd import api
    nb = 10
    stride = api.Lizard_compressBound(131072)
    out = np.empty(nb * stride, dtype=np.uint8)
    sizes = np.zeros(nb, dtype=np.uint32)
    rc = L.LizardGPU_compressBlocks_host(pinned.data_ptr(), nb, 131072, 17, out.ctypes.data, stride, sizes.ctypes.data, 10)
    assert rc == 0, L.LizardGPU_lastError()
    for i in range(nb):
        assert out[i * stride:i * stride + int(sizes[i])].tobytes() == util.oracle_compress(data[i * 131072:(i + 1) * 131072], 10), i


def test_many_chunks_in_a_fresh_process():
    """LIZARDGPU_CHUNK_MB=1 cuts a 40 MiB batch into 40 pipeline chunks (stages alternate, sinks append in order)."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import util\nfrom lizard_amd import api\n"
            "data = util.datagen(40 * (1 << 20) + 555, 0.5, 0.0, 63)\n"
            "for level in (10, 21):\n"
            "    outs = api.compress_blocks(data, 262144, level)\n"
            "    assert len(outs) == 161\n"
            "    for i in list(range(0, 161, 7)) + [159, 160]:\n"
            "        assert outs[i] == util.oracle_compress(data[i * 262144:(i + 1) * 262144], level), (level, i)\n"
            "print('chunks ok')\n") % (util.ROOT, os.path.join(util.ROOT, "tests"))
    env = dict(os.environ, LIZARDGPU_CHUNK_MB="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "chunks ok" in r.stdout, r.stdout + r.stderr


def test_errors_are_per_thread_and_cleared_on_success(L):
    src = np.zeros(4096, dtype=np.uint8)
    out = np.zeros(8192, dtype=np.uint8)
    sizes = np.zeros(1, dtype=np.uint32)
    seen = {}

    def worker():
        seen["before"] = L.LizardGPU_lastError()
        L.LizardGPU_compressBlocks_host(src.ctypes.data, 1, 4096, 4096, out.ctypes.data, 8192, sizes.ctypes.data, 23)   # level 23 (lowestPrice): no kernel
        seen["after"] = L.LizardGPU_lastError()

    assert L.LizardGPU_compressBlocks_host(src.ctypes.data, 1, 4096, 5000, out.ctypes.data, 8192, sizes.ctypes.data, 10) == -3
    mine = L.LizardGPU_lastError()
    assert mine != b""
    t = threading.Thread(target=worker); t.start(); t.join()
    assert seen["before"] == b"" and b"level 23" in seen["after"]
    assert L.LizardGPU_lastError() == mine                      # the other thread's failure did not touch this thread's text
    assert L.LizardGPU_compressBlocks_host(src.ctypes.data, 1, 4096, 4096, out.ctypes.data, 8192, sizes.ctypes.data, 10) == 0
    assert L.LizardGPU_lastError() == b""


def test_shutdown_and_recreate(L):
    from lizard_amd import api
    data = util.datagen(300000, 0.5, 0.0, 64)
    a = api.compress_blocks(data, 131072, 11)
    L.LizardGPU_shutdown()
    b = api.compress_blocks(data, 131072, 11)
    assert a == b and a[0] == util.oracle_compress(data[:131072], 11)
    assert L.LizardGPU_deviceCount() >= 1 and L.LizardGPU_maxBlockSize(11) == 0x7E000000 and L.LizardGPU_maxBlockSize(12) == 0x7E000000 and L.LizardGPU_maxBlockSize(23) == 0
