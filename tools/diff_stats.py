"""Can the spin-up's stragglers be told EARLY from how their passes contract?  (VERDICT r03 #3b, dynamic
variant: a static predicate does not separate them -- profiles/r04_tail_predict.md.)

Measurement build (-DSPL_DIFF_STATS -DSPL_TAIL_STATS, tools/ab_build.sh): records |end_sm - start_sm| of the
first eight passes of spin-up #2 of every cell and the iterations each tail job really executed.

    SPLASH_HIP_LIBRARY=gpurun_ab/lib_diff.so python tools/diff_stats.py [rows=1024] [row0=0]
-> gpurun_out/diff_stats.npz (d[8][n] float32 for the cells with >= 4 passes, tail iterations) + a summary
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rsplash_amd import api
from rsplash_amd._lib import lib

dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
r0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, nt, SEED = 4096, 365, 20221130
n = H * W
cal = api.Calendar.from_dates(np.datetime64("2001-01-01") + np.arange(nt))
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(SEED, 4096, W, r0 * W, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(SEED, cal, 0, nt, r0 * W, n, lat, elev, sw, tc, pn, n)
par = torch.empty(api.NPAR, n, **f64)
Tt = torch.empty(n, **f64)
state = torch.empty(api.NSTATE, n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
dbg = torch.full((8, n), -1.0, dtype=torch.float32, device=dev)
lib.splash_dbg_set_diff.argtypes = [C.c_void_p]
assert lib.splash_dbg_set_diff(dbg.data_ptr()) == 0
work = torch.zeros(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
torch.cuda.synchronize()
al = lambda b: (b + 255) & ~255
o_meta = al(n * 4) * 2
meta = work[o_meta:o_meta + 4 * n].view(torch.int32).cpu().numpy()
d = dbg.cpu().numpy()
tail = meta <= -100000
it = np.where(tail, -(meta + 100000) - 2, 0)          # iterations executed in the tail's read-back launch
npass = (d >= 0).sum(0)                               # passes of spin-up #2 seen (<= 8)
print(f"{n} cells; passes of spin-up #2 (capped at 8): " + ", ".join(f"{k}: {(npass == k).sum()}" for k in range(1, 9)))
print(f"tail jobs (read-back launch): {tail.sum()}; >= 100 iterations: {(it >= 100).sum()}; >= 400: {(it >= 400).sum()}; "
      f">= 900: {(it >= 900).sum()}")
surv = npass >= 3                                     # what phase 0 leaves behind (2 passes of #2 done, not converged)
with np.errstate(divide="ignore", invalid="ignore"):
    ratio = np.where(surv, d[1] / d[0], np.nan)
print("survivors of phase 0:", surv.sum())
for thr in (0.3, 0.5, 0.7, 0.8, 0.9, 0.95, 1.0):
    fl = surv & (ratio >= thr)
    print(f"  d2/d1 >= {thr}: flagged {fl.sum()} ({fl.sum() / n:.4f} of all cells); recall of tail jobs {(fl & tail).sum() / max(1, tail.sum()):.4f}, "
          f"of >= 100-iteration jobs {(fl & (it >= 100)).sum()}/{(it >= 100).sum()}, of >= 400 {(fl & (it >= 400)).sum()}/{(it >= 400).sum()}")
for thr in (1.5, 2.0, 3.0, 5.0, 10.0):
    fl = surv & (d[1] >= thr)
    print(f"  d2 >= {thr} mm: flagged {fl.sum()} ({fl.sum() / n:.4f}); recall of >= 100-iteration jobs {(fl & (it >= 100)).sum()}/{(it >= 100).sum()}, "
          f">= 400: {(fl & (it >= 400)).sum()}/{(it >= 400).sum()}")
# a predicate known BEFORE the spin-up: can the cell have snow at all?  (Tt = max(tc[p_snow >= 0.5]) is -inf
# when no day qualifies, R/splash.point.R:122: f_rain == 1 all year)
snowy = np.isfinite(Tt.cpu().numpy())
snow_end = state[1].cpu().numpy() > 0
for lim in (100, 400, 900):
    L = it >= lim
    print(f"  >= {lim} iterations: {L.sum()} jobs, of which snow-capable (finite Tt): {(L & snowy).sum()}, with snow on the ground at the "
          f"end of the spin-up: {(L & snow_end).sum()}")
print(f"  snow-capable cells: {snowy.sum()} of {n} ({snowy.mean():.3f})")
keep = npass >= 4
np.savez_compressed(f"gpurun_out/diff_stats_r{r0}.npz", snowy=snowy[keep], snow_end=snow_end[keep], cell=np.nonzero(keep)[0].astype(np.int32), d=d[:, keep], it=it[keep].astype(np.int16),
                    tail=tail[keep], n=n, npass_hist=np.bincount(npass, minlength=9))
print(f"wrote gpurun_out/diff_stats_r{r0}.npz:", keep.sum(), "cells")
