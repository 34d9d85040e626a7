"""A/B timing aid: k_run_all (daily five outputs, with and without the water-balance totals) and the
spin-up on the first ROWS x 4096 cells of the benchmark grid.  SPLASH_HIP_LIBRARY selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
assert torch.cuda.is_available()
from rsplash_amd import api
dev = torch.device("cuda:0")
H, W, nt = int(os.environ.get("ROWS", 512)), 4096, 365
row0 = int(os.environ.get("ROW0", 0))
n = H * W
dates = np.datetime64("2001-01-01") + np.arange(nt)
cal = api.Calendar.from_dates(dates)
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(20221130, 4096, W, row0 * W, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(20221130, cal, 0, nt, row0 * W, n, lat, elev, sw, tc, pn, n)
par = torch.empty(api.NPAR, n, **f64)
Tt = torch.empty(n, **f64)
state = torch.empty(api.NSTATE, n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
b.record()
torch.cuda.synchronize()
spin = a.elapsed_time(b)
state0 = state.clone()
outs = {k: torch.empty(nt, n, **f64) for k in ("wn", "snow", "ro", "aet", "pet")}
acc = torch.zeros(1, cal.nmonth, api.DIAG_REC, dtype=torch.int64, device=dev)
res_ = []
for use_acc in (True, False):
    ts = []
    for it in range(3):
        st2 = state0.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        api.dev_run_all(cal, 0, nt, n, sw, tc, pn, n, Tt, par, n, st2, n, outs, n, diag_acc=acc if use_acc else None)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    res_.append(min(ts))
chk = float(outs["wn"].sum().item())
print(f"{os.environ.get('SPLASH_HIP_LIBRARY', 'product'):60s} run_all+diag {res_[0]:7.2f} ms  run_all {res_[1]:7.2f} ms  spin-up {spin:8.1f} ms  sum(wn) {chk!r}", flush=True)
