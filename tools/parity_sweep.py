"""Whole-pipeline parity sweep on the GPU box: splash_grid (soil_hydro, snow threshold, both
spin-ups, run_all, sm_lim) against the oracle's splash.grid restatement on several independent
synthetic grids.  Prints one JSON line per grid and a summary; exit code 1 if a well-conditioned
cell is out of tolerance.      python tools/parity_sweep.py [cells_per_grid] [n_grids] [days] [monthly]
SPLASH_SWEEP_BACKEND=host runs the host backend (ABI 6: the library's own day step compiled for the host) instead of the
GPU -- that sweep needs no GPU and runs in the build container.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import rsplash_amd.api as api
from oracle.binding import Oracle
from oracle.parity import assert_whole_pipeline

ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ngrid = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 730
monthly = len(sys.argv) > 4 and sys.argv[4] == "monthly"
backend = os.environ.get("SPLASH_SWEEP_BACKEND", "hip")
o = Oracle()
W = 128
H = ncell // W
doy, year, month, mid, nm = o.calendar(2001, nt)
dates = np.datetime64("2001-01-01") + np.arange(nt)
worst_all = {}
bad_total = 0
for g in range(ngrid):
    seed = 777000 + 7919 * g
    terrain = 0 if g == ngrid - 1 else 1  # last grid: flat terrain (sticky-NaN drainage time path)
    st = o.synth_statics(seed, H, W, 0, H * W, terrain)
    sw, tc, pn = o.synth_forcing(seed, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
    t0 = time.time()
    ref, aux = o.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"],
                      st["soil"], st["au"], st["resolution"], monthly_out=monthly)
    t_cpu = time.time() - t0
    t0 = time.time()
    got = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                          st["resolution"], time_index=dates, sim_control=dict(monthly_out=monthly),
                          want_aux=True, backend=backend)
    t_gpu = time.time() - t0
    try:
        r = assert_whole_pipeline(o, got, ref, aux, st, doy, year, f"grid {g}",
                                  monthly_mid=mid if monthly else None,
                                  forcing=(doy, year, month, mid, nm, sw, tc, pn))
        ok = r.pop("ok")
        worst, nbad = r["max_rel_err"], 0
    except AssertionError as e:
        print("FAIL", e, flush=True)
        ok = aux["critical"] == 0
        worst, nbad, r = {}, 1, dict(critical_event_cells=int((~ok).sum()), set_aside=-1)
    for k, v in worst.items():
        worst_all[k] = max(worst_all.get(k, 0.0), v)
    passes_equal = bool(np.array_equal(got["aux"]["passes2"][ok], aux["passes2"][ok]) and
                        np.array_equal(got["aux"]["passes1"][ok], aux["passes1"][ok]))
    if not passes_equal:
        # a pass count that differs in a cell within tolerance: a stopping rule met by a hair?  Then
        # the reference's own builds count differently too (oracle/parity.py)
        from oracle.parity import reference_disagrees_with_itself
        mm = np.where(ok & ((got["aux"]["passes2"] != aux["passes2"]) | (got["aux"]["passes1"] != aux["passes1"])))[0]
        exc = reference_disagrees_with_itself(mm[:4], ref, aux, st, (doy, year, month, mid, nm, sw, tc, pn),
                                              ["wn", "ro", "aet", "pet", "snow"])
        print("PASSES", g, [(int(c), int(got["aux"]["passes1"][c]), int(aux["passes1"][c]),
                             int(got["aux"]["passes2"][c]), int(aux["passes2"][c]), bool(e))
                            for c, e in zip(mm[:16], exc)], flush=True)
        passes_equal = bool(len(mm) <= 4 and exc.all())
    bad_total += nbad + (0 if passes_equal else 1)
    print(json.dumps({"grid": g, "seed": seed, "terrain": terrain, "cells": H * W, "days": nt,
                      "critical_event_cells": r["critical_event_cells"], "set_aside": r["set_aside"],
                      "reference_disagrees_with_itself": r.get("reference_disagrees_with_itself", 0),
                      "failures": nbad, "spin_up_pass_counts_equal": passes_equal,
                      "max_passes": int(aux["passes2"].max()), "max_rel_err": worst,
                      "oracle_s": round(t_cpu, 1), "gpu_call_s" if backend == "hip" else "host_backend_call_s": round(t_gpu, 1)}),
          flush=True)
print(json.dumps({"summary": "worst relative error over all grids (cells not set aside)",
                  "max_rel_err": worst_all, "failures": bad_total}))
sys.exit(1 if bad_total else 0)
