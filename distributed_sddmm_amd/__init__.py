"""MI355X-native SDDMM+SpMM engine behind the HnH (PASSIONLab/distributed_sddmm) operator surface.

The product is native: hand-written HIP kernels for gfx950 (csrc/hip -> lib/libhnh_kernels.so, C ABI in
include/hnh_kernels.h) driven by a C++ host layer that mirrors the reference's plugin/operator classes
(csrc/host -> lib/libhnh_host.so, C ABI in include/hnh_dist.h).  This Python package is only a ctypes
binding used by tests/ and bench.py; it contains no compute path and no CPU fallback.
"""
from . import _kernels  # noqa: F401

__all__ = ["_kernels"]
