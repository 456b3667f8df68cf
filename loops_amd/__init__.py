"""loops_amd -- MI355X-native (gfx950) load-balanced CSR SpMV path of gunrock/loops.

The product is the C++ header API under include/loops/ and the C ABI in libloops_amd.so
(include/loops_amd.h); this package is the thin host-side mirror used by the tests, the bench
and multi-GPU runs: torch for device memory / streams / torch.distributed, ctypes for the ABI.
"""
from . import _lib  # noqa: F401
from ._lib import LoopsError, build  # noqa: F401

__all__ = ["LoopsError", "build"]
__version__ = "0.2.0"
