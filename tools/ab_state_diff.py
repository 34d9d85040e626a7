"""A/B aid (GPU box): spin-up + run_all of the first ROWS x 4096 benchmark cells under two library
builds (gpurun_ab/lib_<a>.so, lib_<b>.so), each in its own process; reports where the spun-up
states, the aridity index and the final wn differ.   python tools/ab_state_diff.py v2 v3"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 4 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from rsplash_amd import api
    dev = torch.device("cuda:0")
    H, W, nt = int(os.environ.get("ROWS", 256)), 4096, 365
    n = H * W
    cal = api.Calendar.from_dates(np.datetime64("2001-01-01") + np.arange(nt))
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
    st0 = state.clone()
    outs = {k: torch.empty(nt, n, **f64) for k in ("wn", "snow", "ro", "aet", "pet")}
    api.dev_run_all(cal, 0, nt, n, sw, tc, pn, n, Tt, par, n, state, n, outs, n)
    torch.cuda.synchronize()
    np.savez(sys.argv[3], state=st0.cpu().numpy(), par=par.cpu().numpy(), wn_end=outs["wn"][-1].cpu().numpy(),
             aet_sum=outs["aet"].sum(0).cpu().numpy())
    sys.exit(0)
a, b = sys.argv[1], sys.argv[2]
for v in (a, b):
    env = dict(os.environ, SPLASH_HIP_LIBRARY=f"{ROOT}/gpurun_ab/lib_{v}.so")
    subprocess.run([sys.executable, __file__, "--child", v, f"/tmp/ab_{v}.npz"], env=env, check=True)
A, B = np.load(f"/tmp/ab_{a}.npz"), np.load(f"/tmp/ab_{b}.npz")
for k in A.files:
    x, y = A[k], B[k]
    same = (x.view(np.int64) == y.view(np.int64))
    nd = int((~same).sum())
    print(f"{k:8s} differing values {nd} of {x.size}", end="")
    if nd:
        d = np.abs(x - y)[~same]
        print(f"  max |d| {np.nanmax(d):.3e}  cells {np.unique(np.nonzero(~same)[-1]).size}", end="")
    print()

# which build is closer to the oracle on the cells whose aridity index differs?
sys.path.insert(0, ROOT)
from oracle.binding import Oracle
orc = Oracle()
ai_a, ai_b = A["par"][15], B["par"][15]
idx = np.nonzero(ai_a.view(np.int64) != ai_b.view(np.int64))[0][:48]
if idx.size:
    nt = 365
    doy, year, month, mid, nm = orc.calendar(2001, nt)
    rows = []
    for c in idx:
        st = orc.synth_statics(20221130, 4096, 4096, int(c), 1, 1)
        sw, tc, pn = orc.synth_forcing(20221130, 4096, 4096, int(c), 1, 0, doy, st["lat"], st["elev"])
        o, aux = orc.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"],
                          st["soil"], st["au"], st["resolution"])
        rows.append((int(c), float(aux["AI"][0]), float(ai_a[c]), float(ai_b[c])))
    print("cell, AI oracle, (a - oracle)/oracle, (b - oracle)/oracle")
    for c, o_, x, y in rows:
        print(f"{c:9d} {o_:.15g} {(x - o_) / o_: .3e} {(y - o_) / o_: .3e}")
