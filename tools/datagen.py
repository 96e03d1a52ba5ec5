"""Bench / test tooling: ctypes binding of tools/liblizard_datagen.so, the synthetic-workload generator (block b =
RDG_genBuffer(blockSize, P, litP, seed0 + b) of the reference's programs/datagen.c).  Not part of the product package."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblizard_datagen.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        c = ctypes
        L = ctypes.CDLL(LIB_PATH)
        L.LizardTools_datagen_host.argtypes = [c.c_void_p, c.c_size_t, c.c_double, c.c_double, c.c_uint]
        L.LizardTools_datagen_host.restype = None
        L.LizardTools_datagen_device.argtypes = [c.c_void_p, c.c_size_t, c.c_size_t, c.c_double, c.c_double, c.c_uint, c.c_void_p]
        L.LizardTools_datagen_device.restype = c.c_int
        _lib = L
    return _lib


def datagen_host(buffer, size, match_proba=0.5, lit_proba=0.0, seed=0):
    lib().LizardTools_datagen_host(buffer, size, match_proba, lit_proba, seed)


def datagen_device(d_dst, n_blocks, block_size, match_proba=0.5, lit_proba=0.0, seed0=0, stream=None):
    """Blocks generated on the current HIP device, synchronous; raises on a HIP error."""
    rc = lib().LizardTools_datagen_device(d_dst, n_blocks, block_size, match_proba, lit_proba, seed0, stream)
    if rc:
        raise RuntimeError(f"LizardTools_datagen_device: hipError {rc}")
