"""A multi-year series through the host-buffer entry point in time chunks of one calendar year (ABI 5:
splash_grid_in.tt / state0 / ai0), timed chunk by chunk and VERIFIED against the undivided call on the same
host arrays (every value of every series, bit for bit).  PCIe included; not the bench metric.

    python tools/bench_time_chunks.py [rows=512] [cols=4096] [years=4] [monthly|daily]

The forcing (single precision, as the reference's FLT4S rasters) is generated on the device year by year and
copied to host arrays; chunk k+1 is submitted as soon as chunk k has returned its state.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import rsplash_amd.api as api
from rsplash_amd.grid import year_chunks

H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
Y = int(sys.argv[3]) if len(sys.argv) > 3 else 4
monthly = not (len(sys.argv) > 4 and sys.argv[4] == "daily")
n, SEED = H * W, 20221130
dev = torch.device("cuda:0")
f64 = dict(dtype=torch.float64, device=dev)
dates = np.arange("2001-01-01", f"{2001 + Y}-01-01", dtype="datetime64[D]")
nt = len(dates)
doy, year, month = api.time_fields(dates)
cal = api.Calendar(doy, year, month)
cuts = year_chunks(dates, 1)
t_gen = time.perf_counter()
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(SEED, 4096, W, 1536 * W, n, 1, lat, elev, slop, asp, res, soil, au)   # low-latitude rows
host = {k: np.empty((nt, n), np.float32) for k in ("sw", "tc", "pn")}
for t0, t1 in cuts:
    sw, tc, pn = (torch.empty(t1 - t0, n, **f64) for _ in range(3))
    api.dev_synth_forcing(SEED, cal, t0, t1 - t0, 1536 * W, n, lat, elev, sw, tc, pn, n)
    torch.cuda.synchronize()
    for k, v in dict(sw=sw, tc=tc, pn=pn).items():
        host[k][t0:t1] = v.float().cpu().numpy()
    del sw, tc, pn
stat = [x.cpu().numpy() for x in (elev, soil, lat, slop, asp, au, res)]
del lat, elev, slop, asp, res, soil, au
torch.cuda.empty_cache()
print(f"{H}x{W} cells x {nt} days generated and copied to the host in {time.perf_counter() - t_gen:.1f} s",
      file=sys.stderr, flush=True)
kw = dict(sim_control=dict(monthly_out=monthly), want_aux=True)
args = lambda a, b: (host["sw"][a:b], host["tc"][a:b], host["pn"][a:b], *stat)

# ---- the chunked run (twice: the first pays the buffer allocations).  The chunks' results go straight into slices of
# caller-allocated whole-series arrays (what an R caller's bricks or allocMatrix are), touched once beforehand.
mid = api.month_ids(year, month)
nl = int(mid[-1]) + 1 if monthly else nt
full = {k: np.zeros((nl, n)) for k in api.OUT_NAMES}
lay = lambda a, b: (int(mid[a]), int(mid[b - 1]) + 1) if monthly else (a, b)
for rep in range(2):
    t_all = time.perf_counter()
    tt = None
    for a, b in cuts:
        tk = api.snow_threshold(host["tc"][a:b], stat[2], stat[0])
        tt = tk if tt is None else np.maximum(tt, tk)
    t_pre = time.perf_counter() - t_all
    state = ai = None
    parts, secs = [], []
    for a, b in cuts:
        t0 = time.perf_counter()
        l0, l1 = lay(a, b)
        r = api.splash_grid(*args(a, b), time_index=dates[a:b], tt=tt, state0=state, ai0=ai,
                            out={k: full[k][l0:l1] for k in api.OUT_NAMES}, **kw)
        secs.append(time.perf_counter() - t0)
        state, ai = r["state_end"], (r["aux"]["AI"] if ai is None else ai)
        parts.append(r)
    total = time.perf_counter() - t_all
    print(f"chunked run {rep}: threshold pre-pass {t_pre:.3f} s, chunks {[round(x, 3) for x in secs]} s, total {total:.3f} s",
          file=sys.stderr, flush=True)

# ---- the undivided call on the same arrays
t0 = time.perf_counter()
one = api.splash_grid(*args(0, nt), time_index=dates, **kw)
t_one = time.perf_counter() - t0
differing = values = 0
for k in api.OUT_NAMES:
    got = full[k]
    differing += int((got.view(np.int64) != one[k].view(np.int64)).sum())
    values += got.size
differing += int((state.view(np.int64) != one["state_end"].view(np.int64)).sum())
print(json.dumps({"entry": "splash_grid_run in time chunks (ABI 5: tt / state0 / ai0)", "rows": H, "cols": W, "cells": n,
                  "days": nt, "chunks": len(cuts), "monthly": monthly, "forcing": "f32",
                  "threshold_prepass_s": round(t_pre, 3), "chunk_seconds": [round(x, 3) for x in secs],
                  "seconds": round(total, 3), "cell_days_per_s": n * nt / total,
                  "later_chunks_cell_days_per_s": n * (nt - cuts[0][1]) / sum(secs[1:]),
                  "undivided_call_seconds": round(t_one, 3),
                  "verify": {"values_compared": values + state.size, "differing_values": differing,
                             "against": "the undivided call on the same host arrays: all nine series and the final state"}}))
