"""Round 6: the spin-up's tail with and without the wave-uniform (SOLO) variant of its one-job-per-wavefront stage,
same process, same list; SPLASH_HIP_LIBRARY selects the build.  Prints the time of splash_dev_spin_up_begin / _finish
(best of REPS), the jobs per phase list and a sha256 of the spun-up state + aridity index (the bits must not move).
    python tools/tail_uniform_ab.py [rows=1024] [row0=1536]"""
import hashlib
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
REPS = int(os.environ.get("REPS", 3))
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
lib = os.path.basename(os.environ.get("SPLASH_HIP_LIBRARY", "product"))
modes = [("uniform (default)", {}), ("no uniform variant", {"SPLASH_HIP_NO_UNIFORM_TAIL": "1"})]
for mode, env in modes:
    for k in ("SPLASH_HIP_NO_UNIFORM_TAIL",):
        os.environ.pop(k, None)
    os.environ.update(env)
    best = None
    for rep in range(REPS):
        par, state = par0.clone(), torch.empty(api.NSTATE, n, **f64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        api.dev_spin_up_begin(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        api.dev_spin_up_finish(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if best is None or t2 - t1 < best[1]:
            best = (t1 - t0, t2 - t1)
    h = hashlib.sha256(state.cpu().numpy().tobytes() + par[15].cpu().numpy().tobytes()).hexdigest()[:16]
    c = cnt.cpu().numpy()
    print(f"{lib:22s} {mode:20s}: begin {1e3 * best[0]:7.1f} ms, tail {1e3 * best[1]:7.1f} ms; jobs per phase list "
          f"{c[:10].tolist()}; sha256(state, AI) {h}", flush=True)
