"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/lizard_amd.h
declares. No compute calls (no GPU here). Also: the product must not link or reference oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

import util
from lizard_amd import _lib, api


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def declared_functions():
    text = open(os.path.join(util.ROOT, "include", "lizard_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(Lizard(?:GPU)?_\w+)\s*\(", text)))


def test_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert "Lizard_compress" in names and "LizardGPU_compressBlocks_device" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lizard_amd.h but not exported"


def test_bound_and_version_without_gpu(lib):
    lib.Lizard_compressBound.argtypes = [ctypes.c_int]
    for n in (0, 1, 131071, 131072, 262144, 4 << 20, 0x7E000000):
        assert lib.Lizard_compressBound(n) == util.oracle().lzo_compress_bound(n) == api.Lizard_compressBound(n)
    assert lib.Lizard_compressBound(0x7E000001) == 0
    assert lib.Lizard_versionNumber() == 10000
    assert lib.LizardGPU_levelSupported(10) == 1
    assert lib.Lizard_sizeofState(10) > 0


def test_product_does_not_reference_oracle(lib):
    out = subprocess.check_output(["ldd", _lib.LIB_PATH]).decode()
    assert "oracle" not in out
    for f in os.listdir(os.path.join(util.ROOT, "lizard_amd", "csrc")) + os.listdir(os.path.join(util.ROOT, "lizard_amd")):
        p = os.path.join(util.ROOT, "lizard_amd", "csrc", f)
        if not os.path.isfile(p):
            p = os.path.join(util.ROOT, "lizard_amd", f)
        if os.path.isfile(p) and p.endswith((".h", ".hip", ".c", ".py")):
            assert "oracle/" not in open(p).read().replace("oracle/ ", ""), p


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib.Lizard_compress.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    dst = ctypes.create_string_buffer(1000)
    assert lib.Lizard_compress(b"a" * 100, dst, 100, 1000, 10) == 0   # reference: 0 == failure
    lib.LizardGPU_lastError.restype = ctypes.c_char_p
    assert lib.LizardGPU_lastError() != b""
