"""PCIe-inclusive rate of the host-buffer entry point (splash_grid_run): the whole splash.grid block
call -- NA scan, H2D forcing, soil_hydro, snow threshold, both spin-ups, run_all, D2H outputs --
on a synthetic terrain grid held in host memory.  Not the bench metric (bench.py keeps rasters
resident); recorded in DESIGN.md and profiles/.

    python tools/bench_host_path.py [rows] [cols] [daily|monthly] [n_outputs] [f64|f32] [reps] [cells per block] [out64|out32]

out32 = single-precision transport of the results (splash_grid_out.out_f32, ABI 4).
After the timed calls the results are VERIFIED: a strided set of columns (every 251st cell plus 512 cells
around every multiple of 2^20, where the library's blocks meet) is recomputed by the device-resident
pipeline (tests/gpu_pipeline.py's sequence: soil_hydro, snow threshold, both spin-ups, run_all, in one
block on one stream) from the same host forcing, and compared bit for bit with those columns of the
arrays the call filled; the JSON line carries `verify.differing_values` (must be 0).

The output arrays are allocated and touched once before the timed calls (what the caller of a C ABI
does: R's allocMatrix, or a reused buffer); f32 hands the forcing over in single precision (the
reference's rasters are FLT4S), which halves the host->device bytes and changes no result.
The grid is generated on the device in row chunks and copied to the host arrays chunk by chunk, so
that the full 4096 x 4096 x 365 grid (73 GB of f32 forcing) fits the box's host memory.
SPLASH_TRACE=1 prints the stage times of every block; SPLASH_HIP_LANES sets the pipeline's lanes.
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
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
max_cells = int(sys.argv[7]) if len(sys.argv) > 7 else 0  # cells per block (0: the library's choice)
out32 = len(sys.argv) > 8 and sys.argv[8] == "out32"
odt = np.float32 if out32 else np.float64
n, nt, SEED = H * W, 365, 20221130
dev = torch.device("cuda:0")
f64 = dict(dtype=torch.float64, device=dev)
time_index = np.arange("2001-01-01", "2002-01-01", dtype="datetime64[D]")
doy, year, month = api.time_fields(time_index)
cal = api.Calendar(doy, year, month)
fdt = np.float32 if f32 else np.float64
t_gen = time.perf_counter()
host = dict(lat=np.empty(n), elev=np.empty(n), slop=np.empty(n), asp=np.empty(n), res=np.empty(n),
            soil=np.empty((6, n)), au=np.empty((3, n)),
            sw=np.empty((nt, n), fdt), tc=np.empty((nt, n), fdt), pn=np.empty((nt, n), fdt))
rows_chunk = max(1, min(H, (1 << 21) // W))
for r0 in range(0, H, rows_chunk):
    nr = min(rows_chunk, H - r0)
    nc, c0 = nr * W, r0 * W
    lat, elev, slop, asp, res = (torch.empty(nc, **f64) for _ in range(5))
    soil, au = torch.empty(6, nc, **f64), torch.empty(3, nc, **f64)
    api.dev_synth_statics(SEED, H, W, c0, nc, 1, lat, elev, slop, asp, res, soil, au)
    sw, tc, pn = (torch.empty(nt, nc, **f64) for _ in range(3))
    api.dev_synth_forcing(SEED, cal, 0, nt, c0, nc, lat, elev, sw, tc, pn, nc)
    torch.cuda.synchronize()
    for k, v in dict(lat=lat, elev=elev, slop=slop, asp=asp, res=res).items():
        host[k][c0:c0 + nc] = v.cpu().numpy()
    host["soil"][:, c0:c0 + nc] = soil.cpu().numpy()
    host["au"][:, c0:c0 + nc] = au.cpu().numpy()
    for k, v in dict(sw=sw, tc=tc, pn=pn).items():
        host[k][:, c0:c0 + nc] = (v.float() if f32 else v).cpu().numpy()
    del lat, elev, slop, asp, res, soil, au, sw, tc, pn
torch.cuda.empty_cache()
print(f"grid generated and copied to the host in {time.perf_counter() - t_gen:.1f} s", file=sys.stderr, flush=True)
names = ("wn", "snow", "ro", "aet", "pet", "cond", "bflow", "netr", "sm_lim")[:nout]
bufs = {k: np.zeros(((12 if monthly else nt), n), odt) for k in names}
best = None
times = []
for rep in range(reps):
    t0 = time.perf_counter()
    out = api.splash_grid(host["sw"], host["tc"], host["pn"], host["elev"], host["soil"], host["lat"],
                          host["slop"], host["asp"], Au=host["au"], resolution=host["res"],
                          time_index=time_index, sim_control=dict(monthly_out=monthly), outputs=names,
                          out=bufs, want_state=False, max_cells_per_launch=max_cells, out_dtype=odt)
    dt = time.perf_counter() - t0
    print(f"call {rep}: {dt:.3f} s", file=sys.stderr, flush=True)
    times.append(dt)
    best = dt if best is None else min(best, dt)
bytes_in = n * nt * (12 if f32 else 24)
bytes_out = n * (12 if monthly else nt) * (4 if out32 else 8) * nout


def verify(out):
    """the device-resident pipeline on a strided set of the grid's cells, bitwise against the call's columns"""
    idx = [np.arange(0, n, 251)]
    for b in range(1 << 20, n, 1 << 20):
        idx.append(np.arange(max(0, b - 256), min(n, b + 256)))
    idx = np.unique(np.concatenate(idx))
    m = len(idx)
    t0 = time.perf_counter()
    up = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    lat, elev, slop, asp, res = (up(host[k][idx]) for k in ("lat", "elev", "slop", "asp", "res"))
    soil, au = up(host["soil"][:, idx]), up(host["au"][:, idx])
    sw, tc, pn = (up(host[k][:, idx]) for k in ("sw", "tc", "pn"))      # f32 forcing widens exactly
    par = torch.empty(api.NPAR, m, **f64)
    Tt = torch.empty(m, **f64)
    state = torch.empty(api.NSTATE, m, **f64)
    api.dev_soil_hydro(m, lat, elev, slop, asp, res, soil, au, par)
    api.dev_snow_threshold(m, nt, tc, m, par, m, Tt)
    work = torch.empty(api.spin_up_work_size(m), dtype=torch.uint8, device=dev)
    api.dev_spin_up(cal, m, sw, tc, pn, m, Tt, par, m, state, m, work)
    nt_out = 12 if monthly else nt
    g = {k: torch.empty(nt_out, m, **f64) for k in names}
    api.dev_run_all(cal, 0, nt, m, sw, tc, pn, m, Tt, par, m, state, m, g, m, monthly=monthly)
    torch.cuda.synchronize()
    differing, per = 0, {}
    for k in names:
        want = g[k].cpu().numpy()
        got = out[k][:, idx]
        if out32:
            d = int((want.astype(np.float32).view(np.int32) != got.view(np.int32)).sum())
        else:
            d = int((want.view(np.int64) != got.view(np.int64)).sum())
        per[k] = d
        differing += d
    return {"cells_compared": int(m), "values_compared": int(m) * nt_out * len(names), "differing_values": differing,
            "per_series": per, "seconds": round(time.perf_counter() - t0, 2),
            "against": "device-resident pipeline (one block, one stream) on the same host forcing; every 251st cell + "
                       "512 cells around every multiple of 2^20"}


ver = verify(out)
print(f"verify: {ver['differing_values']} differing values of {ver['values_compared']}", file=sys.stderr, flush=True)
print(json.dumps({"entry": "splash_grid_run (host buffers)", "forcing": "f32" if f32 else "f64", "rows": H, "cols": W,
                  "cells": n, "days": nt, "outputs": list(names), "monthly": monthly, "seconds": best,
                  "seconds_all_calls": [round(t, 3) for t in times], "cell_days_per_s": n * nt / best,
                  "pcie_bytes_in": bytes_in, "pcie_bytes_out": bytes_out,
                  "pcie_GBps_equivalent": (bytes_in + bytes_out) / best / 1e9,
                  "lanes": os.environ.get("SPLASH_HIP_LANES", "2 (default)"), "max_cells_per_block": max_cells,
                  "results": "f32" if out32 else "f64", "verify": ver,
                  "wn_mean": float(np.nanmean(out["wn"][:, ::97], dtype=np.float64))}))
