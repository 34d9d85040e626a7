"""Histogram of the spin-up pass counts (spin-up #2; fast mode: #1 is a single pass) over the first
ROWS x 4096 cells of the benchmark grid at a few latitudes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rsplash_amd import api
dev = torch.device("cuda:0")
W, nt = 4096, 365
cal = api.Calendar.from_dates(np.datetime64("2001-01-01") + np.arange(nt))
f64 = dict(dtype=torch.float64, device=dev)
for row0 in (0, 1024, 2048, 3584):
    H = int(os.environ.get("ROWS", 128))
    n = H * W
    lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
    soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
    api.dev_synth_statics(20221130, 4096, W, row0 * W, n, 1, lat, elev, slop, asp, res, soil, au)
    sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
    api.dev_synth_forcing(20221130, cal, 0, nt, row0 * W, n, lat, elev, sw, tc, pn, n)
    par = torch.empty(api.NPAR, n, **f64); Tt = torch.empty(n, **f64); state = torch.empty(api.NSTATE, n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
    api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
    work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
    passes = torch.zeros(2, n, dtype=torch.int32, device=dev)
    api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work, passes=passes)
    torch.cuda.synchronize()
    p2 = passes[1].cpu().numpy()
    h = np.bincount(np.minimum(p2, 12), minlength=13)
    print(f"rows {row0}..{row0+H}: passes of spin-up #2 (count 1 = 2 passes run; last bin = 12 and more):",
          " ".join(f"{k}:{v / n:.4f}" for k, v in enumerate(h) if v), f"  max {p2.max()}  >=100: {(p2 >= 100).sum()}")
