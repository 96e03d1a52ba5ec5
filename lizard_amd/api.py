"""Host-side mirror of the reference's block-compression interface (inikep/lizard lib/lizard_compress.h)
plus the batch entry points the GPU needs. Names, argument meaning and error behaviour follow the
reference: sizes in bytes, `Lizard_compress` returns b"" on failure where the C function returns 0.

PyTorch appears only as the owner of device memory and streams (plumbing); all compute is in
liblizard_amd.so.
"""
import ctypes

import numpy as np

from . import _lib

LIZARD_MIN_CLEVEL = 10
LIZARD_MAX_CLEVEL = 49
LIZARD_DEFAULT_CLEVEL = 17
LIZARD_MAX_INPUT_SIZE = 0x7E000000
LIZARD_BLOCK_SIZE = 1 << 17


def Lizard_compressBound(isize):
    """reference lib/lizard_compress.h:124 (LIZARD_COMPRESSBOUND)."""
    if isize < 0 or isize > LIZARD_MAX_INPUT_SIZE:
        return 0
    return isize + 1 + 1 + ((isize // LIZARD_BLOCK_SIZE) + 1) * 4


def Lizard_compress(src, compressionLevel=LIZARD_MIN_CLEVEL, maxDstSize=None):
    """reference lib/lizard_compress.c:596 — one block through the GPU path. Returns the compressed bytes,
    or b"" when the C function returns 0 (does not fit in maxDstSize / level not on the GPU path)."""
    L = _lib.lib()
    src = bytes(src)
    if maxDstSize is None:
        maxDstSize = Lizard_compressBound(len(src))
    dst = ctypes.create_string_buffer(max(maxDstSize, 1))
    n = L.Lizard_compress(src, dst, len(src), maxDstSize, compressionLevel)
    return dst.raw[:n]


def level_supported(level):
    return bool(_lib.lib().LizardGPU_levelSupported(level))


def compress_blocks(data, block_size, level=LIZARD_MIN_CLEVEL):
    """Split host `data` into independent blocks of block_size (last one ragged) and compress them in one
    batched GPU call. Returns a list of bytes, block i == Lizard_compress_extState(zero state, block i)."""
    L = _lib.lib()
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    n = buf.size
    if n == 0:
        return []
    nb = (n + block_size - 1) // block_size
    last = n - (nb - 1) * block_size
    stride = Lizard_compressBound(block_size)
    out = np.empty(nb * stride, dtype=np.uint8)
    sizes = np.zeros(nb, dtype=np.uint32)
    rc = L.LizardGPU_compressBlocks_host(buf.ctypes.data, nb, block_size, last, out.ctypes.data, stride,
                                         sizes.ctypes.data, level)
    _lib.check(rc, "LizardGPU_compressBlocks_host")
    return [out[i * stride:i * stride + int(sizes[i])].tobytes() for i in range(nb)]


def compress_blocks_device(src, block_size, level=LIZARD_MIN_CLEVEL, dst=None, sizes=None, n_bytes=None):
    """Device-resident batch: `src` is a torch uint8 CUDA tensor holding the blocks back to back.
    Enqueues on torch's current stream and returns (dst, sizes, stride): dst is a uint8 tensor of
    nb*stride bytes (slot i at i*stride), sizes an int32 tensor view of the uint32 sizes."""
    import torch
    L = _lib.lib()
    assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
    n = int(src.numel()) if n_bytes is None else int(n_bytes)
    nb = (n + block_size - 1) // block_size
    last = n - (nb - 1) * block_size
    stride = (Lizard_compressBound(block_size) + 63) & ~63
    if dst is None:
        dst = torch.empty(nb * stride, dtype=torch.uint8, device=src.device)
    if sizes is None:
        sizes = torch.zeros(nb, dtype=torch.int32, device=src.device)
    L.LizardGPU_setDevice(src.device.index or 0)
    stream = torch.cuda.current_stream(src.device).cuda_stream
    rc = L.LizardGPU_compressBlocks_device(src.data_ptr(), nb, block_size, last, dst.data_ptr(), stride,
                                           sizes.data_ptr(), level, ctypes.c_void_p(stream))
    _lib.check(rc, "LizardGPU_compressBlocks_device")
    return dst, sizes, stride
