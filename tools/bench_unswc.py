#!/usr/bin/env python3
"""Measurement of the unSWC epilogue kernel: 4096x4096 cells x N steps of column water
content resident in HBM -> wn_z, wtd, theta.  Algorithmic bytes: 8 B read + 24 B written = 32 B per
cell-step.  Prints one JSON line (kernel time by stream events, GB/s, fraction of 8 TB/s) and a
CPU figure of the oracle's restatement on a bounded sample."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rsplash_amd  # noqa: E402
rsplash_amd.build_library()
from rsplash_amd import api  # noqa: E402

H = W = int(os.environ.get("UNSWC_ROWS", 4096))
nt = int(os.environ.get("UNSWC_STEPS", 120))
n = H * W
dev = torch.device("cuda:0")
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(20221130, H, W, 0, n, 1, lat, elev, slop, asp, res, soil, au)
par = torch.empty(api.NPAR, n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
SAT, RES = par[4], par[11]
wn = RES[None] + (SAT - RES)[None] * torch.rand(nt, n, **f64)
outs = [torch.empty(nt, n, **f64) for _ in range(3)]
st = torch.cuda.current_stream()
for _ in range(2):
    api.dev_unswc(n, nt, wn, n, soil, 0.3, outs[0], outs[1], outs[2], n, stream=st.cuda_stream)
torch.cuda.synchronize()
ev = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    api.dev_unswc(n, nt, wn, n, soil, 0.3, outs[0], outs[1], outs[2], n, stream=st.cuda_stream)
    b.record(st)
    ev.append((a, b))
torch.cuda.synchronize()
ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
units = float(n) * nt
line = {"kernel": "k_unswc", "cells": n, "steps": nt, "avg_launch_ms": ms,
        "cell_steps_per_s": units / ms * 1e3,
        "roofline": {"bound": "hbm", "alg_bytes_per_unit": 32.0, "achieved": 32.0 * units / ms / 1e6,
                     "peak": 8000.0, "unit": "GB/s", "frac": 32.0 * units / ms / 1e6 / 8000.0}}
from oracle.binding import Oracle  # noqa: E402
o = Oracle()
hs = soil[:, :64].cpu().numpy()
hw = wn[:, :64].cpu().numpy()
t0 = time.perf_counter()
for j in range(64):
    o.unswc(hs[:, j], 0.3, np.tile(hw[:, j], 50))
dt = time.perf_counter() - t0
line["cpu_baseline"] = {"value": 64 * nt * 50 / dt, "unit": "cell-steps/s", "cores": 1, "kind": "port",
                        "sample": f"64 cells x {nt*50} steps"}
print(json.dumps(line))
# the gridded function (R/unsSWC.grid.R: its own semantics and four rasters, k_unswc_grid): 8 B read + 32 B written
outs4 = outs + [torch.empty(nt, n, **f64)]
for _ in range(2):
    api.dev_unswc_grid(n, nt, wn, n, soil, 0.3, *outs4, n, stream=st.cuda_stream)
torch.cuda.synchronize()
ev = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    api.dev_unswc_grid(n, nt, wn, n, soil, 0.3, *outs4, n, stream=st.cuda_stream)
    b.record(st)
    ev.append((a, b))
torch.cuda.synchronize()
ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
line2 = {"kernel": "k_unswc_grid", "cells": n, "steps": nt, "avg_launch_ms": ms, "cell_steps_per_s": units / ms * 1e3,
         "roofline": {"bound": "hbm", "alg_bytes_per_unit": 40.0, "achieved": 40.0 * units / ms / 1e6, "peak": 8000.0,
                      "unit": "GB/s", "frac": 40.0 * units / ms / 1e6 / 8000.0}}
t0 = time.perf_counter()
for j in range(64):
    o.unswc_grid(hs[:, j], 0.3, np.tile(hw[:, j], 50))
dt = time.perf_counter() - t0
line2["cpu_baseline"] = {"value": 64 * nt * 50 / dt, "unit": "cell-steps/s", "cores": 1, "kind": "port",
                         "sample": f"64 cells x {nt*50} steps"}
print(json.dumps(line2))
