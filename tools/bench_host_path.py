"""PCIe-inclusive rate of the host-buffer entry point (splash_grid_run): the whole splash.grid block
call -- NA scan, H2D forcing, soil_hydro, snow threshold, both spin-ups, run_all, D2H outputs --
on a synthetic terrain grid held in host memory.  Not the bench metric (bench.py keeps rasters
resident); recorded in DESIGN.md.

    python tools/bench_host_path.py [rows] [cols] [daily|monthly] [n_outputs] [f64|f32]

The output arrays are allocated and touched once before the timed calls (what the caller of a C ABI
does: R's allocMatrix, or a reused buffer); f32 hands the forcing over in single precision (the
reference's rasters are FLT4S), which halves the host->device bytes and changes no result.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import rsplash_amd.api as api

H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
monthly = len(sys.argv) > 3 and sys.argv[3] == "monthly"
nout = int(sys.argv[4]) if len(sys.argv) > 4 else 5
f32 = len(sys.argv) > 5 and sys.argv[5] == "f32"
n, nt, SEED = H * W, 365, 20221130
dev = torch.device("cuda:0")
f64 = dict(dtype=torch.float64, device=dev)
time_index = np.arange("2001-01-01", "2002-01-01", dtype="datetime64[D]")
doy, year, month = api.time_fields(time_index)
cal = api.Calendar(doy, year, month)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(SEED, H, W, 0, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(SEED, cal, 0, nt, 0, n, lat, elev, sw, tc, pn, n)
torch.cuda.synchronize()
host = {k: v.cpu().numpy() for k, v in dict(lat=lat, elev=elev, slop=slop, asp=asp, res=res, soil=soil,
                                            au=au, sw=sw, tc=tc, pn=pn).items()}
del lat, elev, slop, asp, res, soil, au, sw, tc, pn
torch.cuda.empty_cache()
names = ("wn", "snow", "ro", "aet", "pet", "cond", "bflow", "netr", "sm_lim")[:nout]
if f32:
    for k in ("sw", "tc", "pn"):
        host[k] = host[k].astype(np.float32)
bufs = {k: np.zeros(((12 if monthly else nt), n)) for k in names}
best = None
for rep in range(3):
    t0 = time.perf_counter()
    out = api.splash_grid(host["sw"], host["tc"], host["pn"], host["elev"], host["soil"], host["lat"],
                          host["slop"], host["asp"], Au=host["au"], resolution=host["res"],
                          time_index=time_index, sim_control=dict(monthly_out=monthly), outputs=names,
                          out=bufs, want_state=False)
    dt = time.perf_counter() - t0
    print(f"call {rep}: {dt:.3f} s", file=sys.stderr)
    best = dt if best is None else min(best, dt)
bytes_moved = n * nt * (12 if f32 else 24) + n * (12 if monthly else nt) * 8 * nout
print(json.dumps({"entry": "splash_grid_run (host buffers)", "forcing": "f32" if f32 else "f64", "cells": n, "days": nt, "outputs": list(names),
                  "monthly": monthly, "seconds": best, "cell_days_per_s": n * nt / best,
                  "pcie_bytes": bytes_moved, "pcie_GBps_equivalent": bytes_moved / best / 1e9,
                  "wn_mean": float(np.nanmean(out["wn"]))}))
