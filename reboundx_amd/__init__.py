"""reboundx_amd -- MI355X-native implementation of REBOUNDx's additional-force path.

The arithmetic lives in ``librebxamd.so`` (hand-written HIP kernels for gfx950 behind a
C ABI, see ``include/reboundx_amd.h``).  This package is the thin Python side: ctypes
bindings, a mirror of the reference's ctypes wrapper for the plugin API, synthetic
workload generators and the multi-GPU drivers.  There is no CPU fallback: importing
:mod:`reboundx_amd.device` without the built library raises.
"""
from ._lib import lib, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
