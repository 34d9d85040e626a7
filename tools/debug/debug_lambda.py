"""GPU box: parity failure rate versus pore-size index lambda (which cells are really chaotic?)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rsplash_amd
rsplash_amd.build_library()
from rsplash_amd import api
from oracle.binding import Oracle, grazing_mask
o = Oracle()
dev = torch.device("cuda:0")
Htot, W, rows, nt, seed = 4096, 4096, 48, 365, 20221130
n = rows * W
row0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dates = np.datetime64("2001-01-01") + np.arange(nt)
cal = api.Calendar.from_dates(dates)
doy, year, month = api.time_fields(dates)
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(seed, Htot, W, row0 * W, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(seed, cal, 0, nt, row0 * W, n, lat, elev, sw, tc, pn, n)
par = torch.empty(api.NPAR, n, **f64); Tt = torch.empty(n, **f64); state = torch.empty(5, n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
state0 = state.clone()
names = ("wn", "snow", "ro", "aet", "pet")
g = {k: torch.empty(nt, n, **f64) for k in names}
api.dev_run_all(cal, 0, nt, n, sw, tc, pn, n, Tt, par, n, state, n, g, n)
torch.cuda.synchronize()
hp = par.cpu().numpy()
out, _ = o.grid_transient(doy, year, month, sw.cpu().numpy(), tc.cpu().numpy(), pn.cpu().numpy(), hp[0], hp[1], hp[2], hp[3], hp[4:16],
                          Tt.cpu().numpy(), state0.cpu().numpy(), n_threads=16)
frag = o.fragile.copy()
graz = grazing_mask(o, hp[0], hp[2], hp[3], doy, year).any(0)
bad = np.zeros(n, bool)
for i, k in enumerate(names):
    a, b = g[k].cpu().numpy(), out[i]
    bad |= (np.abs(a - b) > 1e-5 * np.abs(b) + 1e-9).any(0)
lam = hp[8]
print("row0", row0, "lat", hp[0][0], "cells", n, "bad", bad.sum(), "grazing", graz.sum(), "fragile", (frag > 0).sum())
edges = [0, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 1.0]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (lam >= lo) & (lam < hi) & ~graz & (frag == 0)
    print(f"lambda [{lo:g},{hi:g}): cells {m.sum():7d}  bad {int((bad & m).sum()):6d}  ({(bad & m).sum() / max(m.sum(), 1):.4f})")
m = graz | (frag > 0)
print("grazing|fragile: cells", m.sum(), "bad", (bad & m).sum())
