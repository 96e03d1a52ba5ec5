"""lizard_amd — MI355X-native (gfx950) implementation of Lizard's block-compress hot path.

The product is liblizard_amd.so (C ABI in include/lizard_amd.h, HIP kernels in lizard_amd/csrc);
this package is the thin host-side mirror of the reference's block API used by tests and bench.py.
"""
from .api import (LIZARD_BLOCK_SIZE, LIZARD_DEFAULT_CLEVEL, LIZARD_MAX_CLEVEL, LIZARD_MAX_INPUT_SIZE,  # noqa: F401
                  LIZARD_MIN_CLEVEL, Lizard_compress, Lizard_compressBound, compress_blocks,
                  compress_blocks_device, level_supported)
from ._lib import LizardAmdError, build  # noqa: F401
