"""How fast can host arrays reach the device on this box?  (GPU box; informs splash_capi.hip's lanes)
  pageable hipMemcpy / hipMemcpy2D, the same from 2 host threads at once, hipHostRegister + async copy,
  and the cost of registering itself."""
import ctypes as C
import sys
import threading
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
GB = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(GB * (1 << 30))


def chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")


hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
H2D, D2H = 1, 2
d = [C.c_void_p(), C.c_void_p()]
for x in d:
    chk(hip.hipMalloc(C.byref(x), n), "hipMalloc")
h = [np.ones(n, np.uint8), np.ones(n, np.uint8)]
st = [C.c_void_p(), C.c_void_p()]
for s in st:
    chk(hip.hipStreamCreate(C.byref(s)), "stream")


def t(f, reps=2):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        best = min(best, time.perf_counter() - t0)
    return best


def both(f):
    th = [threading.Thread(target=f, args=(i,)) for i in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]


r = {}
r["pageable H2D hipMemcpy, 1 thread GB/s"] = GB / t(lambda: chk(hip.hipMemcpy(d[0], h[0].ctypes.data, n, H2D), "cp"))
r["pageable D2H hipMemcpy, 1 thread GB/s"] = GB / t(lambda: chk(hip.hipMemcpy(h[0].ctypes.data, d[0], n, D2H), "cp"))
rows = 365
w = n // (4 * rows) // 4096 * 4096
r["pageable H2D hipMemcpy2D (365 rows, stride 4x) GB/s"] = (w * rows / 2**30) / t(
    lambda: chk(hip.hipMemcpy2D(d[0], w, h[0].ctypes.data, 4 * w, w, rows, H2D), "cp2d"))
r["pageable H2D, 2 threads at once, aggregate GB/s"] = 2 * GB / t(
    lambda: both(lambda i: chk(hip.hipMemcpy(d[i], h[i].ctypes.data, n, H2D), "cp")))
r["pageable H2D + D2H at once, aggregate GB/s"] = 2 * GB / t(
    lambda: both(lambda i: chk(hip.hipMemcpy(d[i], h[i].ctypes.data, n, H2D) if i == 0 else
                               hip.hipMemcpy(h[i].ctypes.data, d[i], n, D2H), "cp")))
t0 = time.perf_counter()
chk(hip.hipHostRegister(h[0].ctypes.data, n, 0), "register")
r["hipHostRegister s per GB"] = (time.perf_counter() - t0) / GB
chk(hip.hipHostRegister(h[1].ctypes.data, n, 0), "register")


def pinned(i, kind):
    if kind == H2D:
        chk(hip.hipMemcpyAsync(d[i], h[i].ctypes.data, n, H2D, st[i]), "cpa")
    else:
        chk(hip.hipMemcpyAsync(h[i].ctypes.data, d[i], n, D2H, st[i]), "cpa")
    chk(hip.hipStreamSynchronize(st[i]), "sync")


r["registered H2D async GB/s"] = GB / t(lambda: pinned(0, H2D))
r["registered D2H async GB/s"] = GB / t(lambda: pinned(0, D2H))
r["registered H2D + D2H at once, aggregate GB/s"] = 2 * GB / t(lambda: both(lambda i: pinned(i, H2D if i == 0 else D2H)))
r["registered H2D x2 at once, aggregate GB/s"] = 2 * GB / t(lambda: both(lambda i: pinned(i, H2D)))
t0 = time.perf_counter()
chk(hip.hipHostUnregister(h[0].ctypes.data), "unregister")
r["hipHostUnregister s per GB"] = (time.perf_counter() - t0) / GB
for k, v in r.items():
    print(f"{k:60s} {v:8.3f}")
