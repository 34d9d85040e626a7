"""Randomised gaps in the forcing (NaN in tc, sw_in or pn on random cell-days, before and after the
spin-up year): the NaN pattern of every output must equal the oracle's (R's NA propagation,
R/splash.point.R:120-128) and the numbers must be within tolerance.  GPU box -- or, with SPLASH_SWEEP_BACKEND=host, the
host backend (ABI 6) in the build container, no GPU.
    python tools/fuzz_gaps.py [n_cases=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rsplash_amd.api as api
from oracle.binding import Oracle
from oracle.parity import _bad
o = Oracle()
backend = os.environ.get("SPLASH_SWEEP_BACKEND", "hip")
rng = np.random.default_rng(4242)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    H, W, nt = 2, 48, int(rng.choice([400, 500, 731]))
    n = H * W
    seed = int(rng.integers(1, 2**31))
    doy, year, month, mid, nm = o.calendar(2001, nt)
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    st = o.synth_statics(seed, H, W, 0, n, 1)
    sw, tc, pn = o.synth_forcing(seed, H, W, 0, n, 0, doy, st["lat"], st["elev"])
    cells = rng.choice(n, 12, replace=False)
    for c in cells:
        arr = (tc, sw, pn)[int(rng.integers(0, 3))]
        for t in rng.choice(nt, int(rng.integers(1, 4)), replace=False):
            arr[t, c] = np.nan
    g = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                        st["resolution"], time_index=dates, sim_control=dict(monthly_out=False), want_aux=True,
                        backend=backend)
    ref, aux = o.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"],
                      st["soil"], st["au"], st["resolution"])
    msgs = []
    if not np.array_equal(np.isnan(g["aux"]["Tt"]), np.isnan(aux["Tt"])):
        msgs.append("NaN pattern of Tt")
    for k in ("wn", "snow", "ro", "aet", "pet"):
        a, b = g[k][:, cells], ref[k][:, cells]
        if not np.array_equal(np.isnan(a), np.isnan(b)):
            msgs.append(f"NaN pattern of {k}: gpu {int(np.isnan(a).sum())} oracle {int(np.isnan(b).sum())}")
        elif _bad(np.nan_to_num(a), np.nan_to_num(b)).any():
            msgs.append(f"{k} out of tolerance on a gappy cell")
    # the same gaps through the monthly aggregation (R/splash.point.R:210-211, na.rm = T: NA days are left out of their
    # month -- the month of a gap keeps the mean / sum of its other days, all-NA months are NaN or 0), all nine series
    gm = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                         st["resolution"], time_index=dates, sim_control=dict(monthly_out=True), backend=backend)
    refm, _ = o.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"],
                     st["soil"], st["au"], st["resolution"], monthly_out=True)
    for k in refm:
        a, b = gm[k][:, cells], refm[k][:, cells]
        if not np.array_equal(np.isnan(a), np.isnan(b)):
            msgs.append(f"monthly NaN pattern of {k}: product {int(np.isnan(a).sum())} oracle {int(np.isnan(b).sum())}")
        elif _bad(np.nan_to_num(a), np.nan_to_num(b)).any():
            msgs.append(f"monthly {k} out of tolerance on a gappy cell")
    bad += len(msgs)
    print(f"case {case}: {nt} d, NaN cell-days in outputs {int(np.isnan(ref['wn'][:, cells]).sum())}: {'ok' if not msgs else msgs}", flush=True)
print("problems", bad)
sys.exit(1 if bad else 0)
