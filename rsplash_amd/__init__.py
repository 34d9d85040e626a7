"""rsplash_amd -- MI355X (gfx950) implementation of rsplash's gridded SPLASH v2.0 loop.

The compute path is librsplash_hip.so (hand-written HIP kernels behind the C ABI declared in
include/splash_hip.h).  This package is the host-side mirror of the reference's interface for
that path (splash.point / splash.grid / the SPLASH Rcpp module) over ctypes; it contains no
numerical fallback: importing `rsplash_amd.api` without the built library raises.
"""
from .build import build_check_library, build_library, check_library_path, library_path  # noqa: F401

__all__ = ["build_library", "library_path", "build_check_library", "check_library_path"]
