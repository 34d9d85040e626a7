"""In-tree build of librsplash_hip.so (hipcc --offload-arch=gfx950)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path():
    # SPLASH_HIP_LIBRARY: load another build of the library (A/B timing of kernel variants)
    return os.environ.get("SPLASH_HIP_LIBRARY") or os.path.join(_HERE, "librsplash_hip.so")


def build_library(force=False, verbose=False):
    """Compile every HIP source of the package for gfx950; returns the .so path."""
    if os.environ.get("SPLASH_HIP_LIBRARY"):
        return library_path()
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-j4", "-C", csrc]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    if not os.path.exists(library_path()):
        raise RuntimeError("librsplash_hip.so was not produced by " + " ".join(cmd))
    return library_path()


def check_library_path():
    return os.path.join(_HERE, "librsplash_hip_check.so")


def build_check_library(verbose=False):
    """The checking build (the same sources with -DSPL_CHECK_SLOTS: every access to the spin-up's record cache
    reports its slot); loaded by one GPU test only.  `make check`."""
    subprocess.check_call(["make", "-j4", "-C", os.path.join(_HERE, "csrc"), "check"],
                          stdout=None if verbose else subprocess.DEVNULL)
    return check_library_path()
