"""ctypes binding of liblizard_amd.so (the C-ABI product library, include/lizard_amd.h).

There is no Python or CPU implementation behind this module: if the shared library has not been
built (`python -c "import __graft_entry__ as g; g.build()"` or `make -C lizard_amd/csrc`) importing
any compute entry point raises, loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblizard_amd.so")

_lib = None


class LizardAmdError(RuntimeError):
    pass


def build(force=False):
    """Compile liblizard_amd.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    args = ["make", "-C", os.path.join(_HERE, "csrc")]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LizardAmdError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `make -C lizard_amd/csrc` (or __graft_entry__.build()). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    c = ctypes
    L.Lizard_versionNumber.restype = c.c_int
    L.Lizard_compressBound.argtypes = [c.c_int]; L.Lizard_compressBound.restype = c.c_int
    L.Lizard_sizeofState.argtypes = [c.c_int]; L.Lizard_sizeofState.restype = c.c_int
    L.Lizard_compress.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int]; L.Lizard_compress.restype = c.c_int
    L.Lizard_compress_extState.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int]
    L.Lizard_compress_extState.restype = c.c_int
    L.LizardGPU_levelSupported.argtypes = [c.c_int]; L.LizardGPU_levelSupported.restype = c.c_int
    L.LizardGPU_setDevice.argtypes = [c.c_int]; L.LizardGPU_setDevice.restype = c.c_int
    L.LizardGPU_lastError.restype = c.c_char_p
    L.LizardGPU_residentWaves.restype = c.c_int
    L.LizardGPU_lastKernelMs.restype = c.c_float
    L.LizardGPU_compressBlocks_device.argtypes = [c.c_void_p, c.c_size_t, c.c_size_t, c.c_size_t, c.c_void_p,
                                                  c.c_size_t, c.c_void_p, c.c_int, c.c_void_p]
    L.LizardGPU_compressBlocks_device.restype = c.c_int
    L.LizardGPU_compressBlocks_host.argtypes = [c.c_void_p, c.c_size_t, c.c_size_t, c.c_size_t, c.c_void_p,
                                                c.c_size_t, c.c_void_p, c.c_int]
    L.LizardGPU_compressBlocks_host.restype = c.c_int
    L.LizardGPU_compressBlocks_host_packed.argtypes = [c.c_void_p, c.c_size_t, c.c_size_t, c.c_size_t, c.c_void_p,
                                                       c.c_size_t, c.c_void_p, c.c_void_p, c.c_int]
    L.LizardGPU_compressBlocks_host_packed.restype = c.c_int
    L.LizardGPU_maxBlockSize.argtypes = [c.c_int]; L.LizardGPU_maxBlockSize.restype = c.c_size_t
    L.LizardGPU_deviceCount.restype = c.c_int
    L.LizardGPU_shutdown.restype = None
    L.LizardGPU_shardRange.argtypes = [c.c_size_t, c.c_int, c.c_int, c.c_void_p, c.c_void_p]; L.LizardGPU_shardRange.restype = None
    L.LizardGPU_offsetsFromSizes.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p]; L.LizardGPU_offsetsFromSizes.restype = None
    L.LizardGPU_compressBlocks_sharded.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_size_t, c.c_size_t, c.c_size_t,
                                                   c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p, c.c_int]
    L.LizardGPU_compressBlocks_sharded.restype = c.c_int
    L.LizardGPU_commUniqueId.argtypes = [c.c_void_p]; L.LizardGPU_commUniqueId.restype = c.c_int
    L.LizardGPU_commInitRank.argtypes = [c.c_void_p, c.c_int, c.c_int]; L.LizardGPU_commInitRank.restype = c.c_int
    L.LizardGPU_gatherSizes_device.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_void_p, c.c_void_p]
    L.LizardGPU_gatherSizes_device.restype = c.c_int
    L.LizardGPU_commDestroy.restype = c.c_int
    _lib = L
    return L


_ERR = {1: "no HIP device", 2: "level not implemented on the GPU path", 3: "bad argument", 4: "HIP call failed",
        5: "out of memory", 6: "RCCL failure"}


def check(rc, what):
    if rc < 0:
        detail = lib().LizardGPU_lastError().decode(errors="replace")
        raise LizardAmdError(f"{what}: {_ERR.get(-rc, rc)} {detail}".strip())
    return rc
