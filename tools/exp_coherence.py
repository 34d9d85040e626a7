"""Experiment: what does the independence of the synthetic weather cost?  The benchmark's rain (and cloud,
and temperature noise) is drawn per cell and day, so every wavefront has a raining lane every day and pays
the rainy path; real fields are coherent over many cells.  The first row block of the benchmark grid is run
through the unchanged spin-up and k_run_all with the forcing of each group of G consecutive cells replaced by
the forcing of the group's first cell (G = 64: a wavefront sees ONE weather; statics stay per cell), and with
whole cells cloned (statics too: the floor, a wavefront of 64 copies).
    ROWS=512 python tools/exp_coherence.py"""
import os, sys
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
par0 = torch.empty(api.NPAR, n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par0)
outs = {k: torch.empty(nt, n, **f64) for k in ("wn", "snow", "ro", "aet", "pet")}
acc = torch.zeros(1, cal.nmonth, api.DIAG_REC, dtype=torch.int64, device=dev)
work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)


def shared(x, g):
    """every group of g consecutive cells gets the series of its first cell"""
    if g <= 1:
        return x
    lead = x.reshape(*x.shape[:-1], n // g, g)[..., :1]
    return lead.expand(*x.shape[:-1], n // g, g).reshape(x.shape).contiguous()


def run(label, g_rain=1, g_all=1, g_cell=1):
    p_ = shared(pn, max(g_rain, g_all, g_cell))
    s_, t_ = shared(sw, max(g_all, g_cell)), shared(tc, max(g_all, g_cell))
    par = shared(par0, g_cell).clone()
    Tt = torch.empty(n, **f64)
    state = torch.empty(api.NSTATE, n, **f64)
    api.dev_snow_threshold(n, nt, t_, n, par, n, Tt)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    api.dev_spin_up(cal, n, s_, t_, p_, n, Tt, par, n, state, n, work)
    b.record()
    torch.cuda.synchronize()
    spin = a.elapsed_time(b)
    ts = []
    for it in range(3):
        st2 = state.clone()
        a.record()
        api.dev_run_all(cal, 0, nt, n, s_, t_, p_, n, Tt, par, n, st2, n, outs, n, diag_acc=acc)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    wet = float((p_ > 0).double().mean().item())
    print(f"{label:64s} run_all {min(ts):7.2f} ms   spin-up {spin:8.1f} ms   wet cell-days {wet:.3f}", flush=True)


run("independent weather per cell and day (the benchmark)")
for g in (4, 16, 64, 256):
    run(f"rain shared by groups of {g} cells", g_rain=g)
for g in (64, 256):
    run(f"rain, radiation, temperature shared by groups of {g} cells", g_all=g)
run("whole cells cloned 64 times (a wavefront = one cell)", g_cell=64)
