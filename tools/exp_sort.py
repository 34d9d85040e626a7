"""Experiment: what does wave-coherence buy?  Cells are independent, so the benchmark block is
re-ordered (all per-cell arrays gathered with torch) by various keys, within 256-cell workgroups or
over the whole block, and the unchanged k_run_all is timed.  Benefit side only (no scatter cost)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
assert torch.cuda.is_available()
from rsplash_amd import api
dev = torch.device("cuda:0")
H, W, nt = int(os.environ.get("ROWS", 512)), 4096, 365
n = H * W
dates = np.datetime64("2001-01-01") + np.arange(nt)
cal = api.Calendar.from_dates(dates)
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(20221130, 4096, W, 0, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(20221130, cal, 0, nt, 0, n, lat, elev, sw, tc, pn, n)
par = torch.empty(api.NPAR, n, **f64)
Tt = torch.empty(n, **f64)
state = torch.empty(api.NSTATE, n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
torch.cuda.synchronize()
state0 = state.clone()
outs = {k: torch.empty(nt, n, **f64) for k in ("wn", "snow", "ro", "aet", "pet")}
acc = torch.zeros(1, cal.nmonth, api.DIAG_REC, dtype=torch.int64, device=dev)

def run(order=None, label=""):
    if order is None:
        s_, t_, p_, pa, T, st = sw, tc, pn, par, Tt, state0.clone()
    else:
        s_, t_, p_ = (x[:, order].contiguous() for x in (sw, tc, pn))
        pa, T, st = par[:, order].contiguous(), Tt[order].contiguous(), state0[:, order].contiguous()
    ts = []
    for it in range(3):
        st2 = st.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        api.dev_run_all(cal, 0, nt, n, s_, t_, p_, n, T, pa, n, st2, n, outs, n, diag_acc=acc)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"{label:50s} {min(ts):8.2f} ms", flush=True)
    return min(ts)

def within(key, g=256):
    """sort by key inside consecutive groups of g cells"""
    k = key.reshape(-1, g)
    o = torch.argsort(k, dim=1, stable=True)
    return (o + torch.arange(0, n, g, device=dev)[:, None]).reshape(-1)

base = run(None, "baseline (grid order)")
deep = (par[9] >= 2.0).double()
ksat, ai, el, lam = par[7], par[15], par[1], par[8]
keys = {"deep": deep, "Ksat": ksat, "AI": ai, "elev": el, "lambda": lam,
        "deep,Ksat": deep * 1e6 + ksat, "deep,AI": deep * 1e6 + ai, "deep,elev": deep * 1e6 + el}
for name, k in keys.items():
    run(within(k), f"within 256-cell workgroups by {name}")
for name in ("deep", "Ksat", "AI", "elev", "deep,Ksat"):
    run(torch.argsort(keys[name], stable=True), f"whole block by {name}")
# two-level: whole block by deep then AI quantile then Ksat
q = lambda x, m: torch.bucketize(x, torch.quantile(x[::97], torch.linspace(0, 1, m + 1, device=dev, dtype=torch.float64)[1:-1]))
k2 = deep * 1e4 + q(ai, 8).double() * 1e2 + q(ksat, 8).double() * 10 + q(el, 8).double()
run(torch.argsort(k2, stable=True), "whole block by deep, AI oct, Ksat oct, elev oct")
k3 = deep * 1e4 + q(ksat, 8).double() * 1e2 + q(ai, 8).double() * 10 + q(el, 8).double()
run(torch.argsort(k3, stable=True), "whole block by deep, Ksat oct, AI oct, elev oct")
k4 = deep * 1e4 + q(el, 8).double() * 1e2 + q(ksat, 8).double() * 10 + q(ai, 8).double()
run(torch.argsort(k4, stable=True), "whole block by deep, elev oct, Ksat oct, AI oct")
run(within(k2, 1024), "within 1024-cell tiles by deep,AI,Ksat,elev")
run(within(k2, 4096), "within 4096-cell tiles by deep,AI,Ksat,elev")
run(within(k2, 16384), "within 16384-cell tiles by deep,AI,Ksat,elev")
