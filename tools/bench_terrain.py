"""HBM fraction of the static-raster kernels on a 16384 x 16384 DEM."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
assert torch.cuda.is_available()
from rsplash_amd import api
H = W = int(os.environ.get("N", 16384))
dev = torch.device("cuda:0")
z = (800 + 600 * torch.sin(torch.arange(W, device=dev, dtype=torch.float64) / 37.0)[None, :] *
     torch.cos(torch.arange(H, device=dev, dtype=torch.float64) / 23.0)[:, None]).contiguous()
o = [torch.empty(H * W, dtype=torch.float64, device=dev) for _ in range(6)]
c9 = torch.tensor(api.FLOW_IN_TOP, dtype=torch.float64, device=dev)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
t1 = timed(lambda: api.dev_terrain(H, W, z, 60.0, 0.005, 0.005, o[0], o[1], o[2], o[3], o[4]))
t2 = timed(lambda: api.dev_ncellflow(H, W, o[4], c9, o[5]))
n = H * W
print(json.dumps({"cells": n, "terrain_ms": t1, "terrain_alg_bytes_per_cell": 48, "terrain_GBs": 48 * n / t1 / 1e6,
                  "terrain_frac_of_8TBs": 48 * n / t1 / 1e6 / 8000,
                  "ncellflow_ms": t2, "ncellflow_alg_bytes_per_cell": 16, "ncellflow_GBs": 16 * n / t2 / 1e6,
                  "ncellflow_frac_of_8TBs": 16 * n / t2 / 1e6 / 8000}))
