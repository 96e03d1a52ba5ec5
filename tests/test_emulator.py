"""CPU: runs the PRODUCT kernel bodies (lizard_amd/csrc/lz_block.h ...) lane-for-lane on the 64-lane
SIMT emulator (tests/emul) and checks them bit-exact against the oracle and the golden vectors.
This is how kernel logic is debugged without a GPU; the GPU parity tests (-m gpu) repeat the same
comparisons through the C ABI on real hardware."""
import ctypes
import json
import os

import pytest

import util

with open(os.path.join(util.GOLDEN_DIR, "reference_vectors.json")) as f:
    GOLDEN = json.load(f)

EMUL_LEVELS = [10, 11]


def emul_compress(data, level, seed=1):
    n = len(data)
    bound = util.oracle().lzo_compress_bound(n)
    dst = ctypes.create_string_buffer(bound + 64)
    src = ctypes.create_string_buffer(bytes(data), n) if n else ctypes.create_string_buffer(1)
    r = util.emulator().emul_compress_block(src, n, dst, level, seed)
    assert r >= 0, "level not wired into the emulator"
    return dst.raw[:r]


@pytest.mark.parametrize("level", EMUL_LEVELS)
def test_emulated_kernel_vs_golden(level):
    for name, data in util.corpus():
        g = GOLDEN["cases"][name]["out"][str(level)]
        out = emul_compress(data, level, seed=len(name) + level)
        assert len(out) == g["size"] and util.sha(out) == g["sha256"], (name, level)


def test_emulated_kernel_schedule_independent():
    """Output must not depend on the order lanes run between cross-lane ops (LDS store races)."""
    data = dict(util.corpus(small=True))["gen262144_p0.5"]
    want = util.oracle_compress(data, 10)
    for seed in (1, 2, 3, 12345):
        assert emul_compress(data, 10, seed) == want
