"""How many passes the jobs of the spin-up's tail really execute (library built with -DSPL_TAIL_STATS:
the completion marker in the workspace's `meta` array then carries the executed iterations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rsplash_amd import api
dev = torch.device("cuda:0")
H, W, nt = int(os.environ.get("ROWS", 1024)), 4096, 365
n = H * W
cal = api.Calendar.from_dates(np.datetime64("2001-01-01") + np.arange(nt))
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(20221130, 4096, W, 0, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(20221130, cal, 0, nt, 0, n, lat, elev, sw, tc, pn, n)
par = torch.empty(api.NPAR, n, **f64); Tt = torch.empty(n, **f64); state = torch.empty(api.NSTATE, n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
work = torch.zeros(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
torch.cuda.synchronize()
al = lambda b: (b + 255) & ~255
o_meta = al(n * 4) * 2
meta = work[o_meta:o_meta + 4 * n].view(torch.int32).cpu().numpy()
rb = meta[meta <= -100000]
it = -(rb + 100000) - 2   # iterations executed in the read-back launch
print("jobs finished by the read-back launch:", it.size)
h = np.bincount(np.minimum(it // 50, 22))
for k, v in enumerate(h):
    if v: print(f"  {50*k:4d}..{50*k+49:4d} iterations: {v}")
long_cells = np.nonzero(meta <= -100000 - 2 - 400)[0]
print("cells with >= 400 read-back iterations:", long_cells.tolist())
