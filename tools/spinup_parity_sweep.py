"""The spin-up's RESULT against the oracle's brute-force spin-up, at scale (GPU box; VERDICT r05 #1, builder-run companion of
tests/test_gpu_fullsize.py::test_benchmark_rows_bands_own_spin_up and bench.py's `spinup_parity`): ROWS rows from each of
eight latitude bands of the 4096 x 4096 benchmark grid, each side from its OWN spin-up -- the product's phased spin-up
with exact-cycle detection, the oracle's 1000-pass loop (src/SPLASH.cpp:1594-1749 x 2, R/splash.point.R:141-152).
oracle/parity.py::spinup_summary per chunk of 4 rows: state handed to run_all (wn, snow, qin, td within 1e-5; nd exactly),
aridity index, pass counts.
    python tools/spinup_parity_sweep.py [rows_per_band=16] [row_offset=0]   (row_offset: other rows of every band)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from rsplash_amd import api
from oracle.binding import Oracle
from oracle.parity import spinup_summary
from gpu_pipeline import device_pipeline

rows_per_band = int(sys.argv[1]) if len(sys.argv) > 1 else 16
row_offset = int(sys.argv[2]) if len(sys.argv) > 2 else 0
Htot = W = 4096
chunk = 4
o = Oracle()
doy, year, month, mid, nm = o.calendar(2001, 365)
tot = dict(cells=0, capped=0, capped_within_tolerance=0, passes_equal=0, nd_equal=0, critical_event_cells=0,
           critical_event_cells_that_differ=0, excused=0, grazing_cells=0, violations=0)
worst_state, worst_capped, worst_ai = 0.0, 0.0, 0.0
t_all = time.time()
for b in range(8):
    r_band = int(round(b * (Htot - rows_per_band) / 7))
    r_band = r_band + row_offset if r_band + row_offset + rows_per_band <= Htot else r_band - row_offset
    for r0 in range(r_band, r_band + rows_per_band, chunk):
        t0 = time.time()
        p = device_pipeline(api, 20221130, Htot, W, r0 * W, chunk * W, 365, 1, want_passes=True)
        st = p["statics"]
        t1 = time.time()
        _, aux = o.grid(doy, year, month, mid, nm, p["sw"], p["tc"], p["pn"], st["lat"], st["elev"], st["slop"], st["asp"],
                        st["soil"], st["au"], st["resolution"])
        rs = spinup_summary(o, p["state0"], p["par"][15], p["passes"][1], aux, st,
                            (doy, year, month, mid, nm, p["sw"], p["tc"], p["pn"]))
        for k in tot:
            if k == "violations":
                tot[k] += len(rs["violations"])
            else:
                tot[k] += rs[k]
        worst_state = max(worst_state, max(rs["max_rel_state"].values()))
        worst_capped = max(worst_capped, rs["max_rel_state_capped"])
        worst_ai = max(worst_ai, rs["max_rel_ai"])
        print(json.dumps({"rows": [r0, r0 + chunk - 1], "lat_deg": round(60.0 - 120.0 * (r0 + chunk / 2) / Htot, 2),
                          "cells": rs["cells"], "capped": rs["capped"], "capped_within_tolerance": rs["capped_within_tolerance"],
                          "passes_equal": rs["passes_equal"], "nd_equal": rs["nd_equal"],
                          "critical_event_cells": rs["critical_event_cells"], "differ": rs["critical_event_cells_that_differ"],
                          "excused": rs["excused"], "grazing_cells": rs["grazing_cells"],
                          "max_rel_state": max(rs["max_rel_state"].values()), "max_rel_ai": rs["max_rel_ai"],
                          "violations": rs["violations"], "gpu_s": round(t1 - t0, 2), "oracle_s": round(time.time() - t1, 1)}),
              flush=True)
print(json.dumps(dict(tot, summary=True, max_rel_state=worst_state, max_rel_state_capped=worst_capped, max_rel_ai=worst_ai,
                      seconds=round(time.time() - t_all, 1))))
sys.exit(1 if tot["violations"] else 0)
