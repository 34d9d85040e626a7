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
# the graph part of R/upslope_area.R on a rough 4096 x 4096 tile: sink fill, flow direction of the
# filled surface, D8 accumulation (latency-bound orders, not byte streams: times and step counts)
G = int(os.environ.get("NG", 4096))
g = torch.Generator(device=dev); g.manual_seed(7)
zr = (z[:G, :G] + 40.0 * torch.rand(G, G, generator=g, device=dev, dtype=torch.float64)).contiguous()
filled, fdg, cnt, resg = (torch.empty(G * G, dtype=torch.float64, device=dev) for _ in range(4))
def wall(fn):
    torch.cuda.synchronize(); t0 = __import__("time").time(); r = fn(); torch.cuda.synchronize()
    return (__import__("time").time() - t0) * 1e3, r
api.dev_sinkfill(G, G, zr, 250.0, filled, 1.0)
t3, sweeps = wall(lambda: api.dev_sinkfill(G, G, zr, 250.0, filled, 1.0))
api.dev_terrain(G, G, filled, 60.0, 0.005, 0.005, resolution=resg, flowdir=fdg)
api.dev_flow_accumulation(G, G, fdg, cnt, elev=filled)
t4, levels = wall(lambda: api.dev_flow_accumulation(G, G, fdg, cnt, elev=filled))
mfd = torch.empty(G * G, dtype=torch.float64, device=dev)
api.dev_mfd_area(G, G, filled, 250.0, mfd)
t5, mlevels = wall(lambda: api.dev_mfd_area(G, G, filled, 250.0, mfd))
graph = {"mfd_area_ms": t5, "mfd_levels": mlevels, "graph_cells": G * G, "sinkfill_ms": t3, "sinkfill_sweeps": sweeps, "flow_accumulation_ms": t4,
         "flow_accumulation_host_levels": levels, "largest_catchment_cells": float(cnt[~torch.isnan(cnt)].max().item())}
print(json.dumps(graph))
print(json.dumps({"cells": n, "terrain_ms": t1, "terrain_alg_bytes_per_cell": 48, "terrain_GBs": 48 * n / t1 / 1e6,
                  "terrain_frac_of_8TBs": 48 * n / t1 / 1e6 / 8000,
                  "ncellflow_ms": t2, "ncellflow_alg_bytes_per_cell": 16, "ncellflow_GBs": 16 * n / t2 / 1e6,
                  "ncellflow_frac_of_8TBs": 16 * n / t2 / 1e6 / 8000}))
