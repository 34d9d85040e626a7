"""Same-process A/B of the spin-up's tail: one packed stage (round 3, SPLASH_HIP_NO_SOLO_TAIL=1) against the cascade of
spread stages (round 4; STAGE_SWEEP="32,64;16,48;..." tries other budgets).  Prints the list lengths and the time of splash_dev_spin_up_finish, and checks the states' bits.
    python tools/tail_ab.py [rows=1024] [row0=1536]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rsplash_amd import api

dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
r0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
W, nt, SEED = 4096, 365, 20221130
n = H * W
cal = api.Calendar.from_dates(np.datetime64("2001-01-01") + np.arange(nt))
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(SEED, 4096, W, r0 * W, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(SEED, cal, 0, nt, r0 * W, n, lat, elev, sw, tc, pn, n)
par0 = torch.empty(api.NPAR, n, **f64)
Tt = torch.empty(n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par0)
api.dev_snow_threshold(n, nt, tc, n, par0, n, Tt)
al = lambda b: (b + 255) & ~255
o_cnt = al(n * 4) * 3 + al(n * 8)
work = torch.zeros(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
cnt = work[o_cnt:o_cnt + 64].view(torch.int32)
ref = None
modes = ["packed only", "cascade (default)"] + [f"stages {p}" for p in os.environ.get("STAGE_SWEEP", "").split(";") if p]
for rep in range(2):
    for mode in modes:
        os.environ.pop("SPLASH_HIP_TAIL_STAGES", None)
        if mode == "packed only":
            os.environ["SPLASH_HIP_NO_SOLO_TAIL"] = "1"
        else:
            os.environ.pop("SPLASH_HIP_NO_SOLO_TAIL", None)
            if mode.startswith("stages "):
                os.environ["SPLASH_HIP_TAIL_STAGES"] = mode.split()[1]
        par, state = par0.clone(), torch.empty(api.NSTATE, n, **f64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        api.dev_spin_up_begin(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        api.dev_spin_up_finish(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        c = cnt.cpu().numpy()
        same = "-" if ref is None else bool(torch.equal(state.view(torch.int64), ref.view(torch.int64)))
        if ref is None:
            ref = state.clone()
        print(f"{mode:20s}: begin {1e3 * (t1 - t0):7.1f} ms, tail {1e3 * (t2 - t1):7.1f} ms; jobs per phase list {c[:8].tolist()}; "
              f"same bits as the first run: {same}", flush=True)
