"""Which cells end in the spin-up's tail?  (VERDICT r03 #3b: is there a STATIC predicate that catches the
jobs of >= 100 passes, so that those cells could be dealt into the first group of a depth-first spin-up
and start their chains early?)

Runs both spin-ups with pass counts on a few row bands of the 4096 x 4096 benchmark grid and writes, for
every cell with passes2 >= 30 and for a 1/16 sample of the others, the static features a predicate could
use (latitude, elevation, slope, aspect, soil_hydro's outputs, upslope area, aridity index) to
gpurun_out/tail_predict.npz; prints the pass histogram per band.  Analysis: tools/tail_predict_fit.py.

    python tools/tail_predict.py [rows per band = 256] [band row0 ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rsplash_amd import api

dev = torch.device("cuda:0")
HT, W, nt, SEED = 4096, 4096, 365, 20221130
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bands = [int(x) for x in sys.argv[2:]] or [0, 1024, 1920, 2560, 3584]
cal = api.Calendar.from_dates(np.datetime64("2001-01-01") + np.arange(nt))
f64 = dict(dtype=torch.float64, device=dev)
feat_names = ["lat", "elev", "slop", "asp180", "SAT", "WP", "FC", "Ksat", "lambda", "depth", "bub", "RES", "Au",
              "Ai", "ncellin", "AI", "Tt", "sum_pn", "mean_tc"]
rec = {k: [] for k in ("cell", "passes1", "passes2", "feat")}
for r0 in bands:
    n = rows * W
    lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
    soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
    api.dev_synth_statics(SEED, HT, W, r0 * W, n, 1, lat, elev, slop, asp, res, soil, au)
    sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
    api.dev_synth_forcing(SEED, cal, 0, nt, r0 * W, n, lat, elev, sw, tc, pn, n)
    par = torch.empty(api.NPAR, n, **f64)
    Tt = torch.empty(n, **f64)
    state = torch.empty(api.NSTATE, n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
    api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
    work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
    passes = torch.zeros(2, n, dtype=torch.int32, device=dev)
    api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work, passes=passes)
    torch.cuda.synchronize()
    p = passes.cpu().numpy()
    f = torch.cat([par[:16], Tt[None], pn.sum(0)[None], tc.mean(0)[None]]).cpu().numpy().astype(np.float32)
    keep = (p[1] >= 30) | (np.arange(n) % 16 == 0)
    rec["cell"].append((r0 * W + np.nonzero(keep)[0]).astype(np.int64))
    rec["passes1"].append(p[0][keep].astype(np.int16))
    rec["passes2"].append(p[1][keep].astype(np.int16))
    rec["feat"].append(f[:, keep])
    h = np.bincount(np.minimum(p[1], 1000) // 100, minlength=11)
    print(f"rows {r0}..{r0 + rows - 1} (lat {60 - 120 * (r0 + 0.5) / HT:.1f}): passes2 >= 30: {(p[1] >= 30).sum()}, "
          f">= 100: {(p[1] >= 100).sum()}, == 1000: {(p[1] >= 1000).sum()} of {n}; by hundreds {h.tolist()}", flush=True)
    del sw, tc, pn, work
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/tail_predict.npz", cell=np.concatenate(rec["cell"]), passes1=np.concatenate(rec["passes1"]),
                    passes2=np.concatenate(rec["passes2"]), feat=np.concatenate(rec["feat"], axis=1),
                    names=np.array(feat_names), rows=rows, bands=np.array(bands))
print("wrote gpurun_out/tail_predict.npz")
