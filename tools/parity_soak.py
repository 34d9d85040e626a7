"""One-step and trajectory parity (oracle/parity.py) on many independent synthetic grids (GPU box):
the rules the GPU tests apply to four fixed cases, over seeds they have not seen.
    python tools/parity_soak.py [n_grids=12] [cells=8192] [days=730] [first_seed=911000]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rsplash_amd import api
from oracle.binding import Oracle
from oracle.parity import assert_one_step, trajectory_summary
from gpu_pipeline import device_pipeline
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 730
seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 911000
o = Oracle()
bad = 0
for g in range(ng):
    seed = seed0 + 104729 * g
    p = device_pipeline(api, seed, n // 128, 128, 0, n, nt, 0 if g % 6 == 5 else 1)
    msg = []
    try:
        r1 = assert_one_step(o, p, f"grid {g}")
    except AssertionError as e:
        r1 = {}
        msg.append("one-step: " + str(e)[:300])
    r2 = trajectory_summary(o, p)
    msg += ["trajectory: " + v[:300] for v in r2["violations"]]
    bad += len(msg)
    def bulk(prof):  # worst share of a series' cell-days beyond 1e-9 / 1e-7 (the levels assert_bulk holds)
        return {} if not prof else {"beyond_1e-9": max(v["beyond_1e_9"] / v["n"] for v in prof.values()),
                                    "beyond_1e-7": max(v["beyond_1e_7"] / v["n"] for v in prof.values())}
    b1, b2 = bulk(r1.pop("profile", None)), bulk(r2.get("profile"))
    print(json.dumps({"grid": g, "seed": seed, "bulk_one_step": b1, "bulk_trajectory": b2,
                      "one_step": {k: v for k, v in r1.items() if not hasattr(v, "shape")},
                      "trajectory": {k: r2[k] for k in ("set_aside_cell_days", "critical_event_cells",
                                                         "critical_event_cells_that_differ", "max_abs_dwn_mm")},
                      "problems": msg}, default=str)[:1400], flush=True)
print(json.dumps({"grids": ng, "problems": bad}))
sys.exit(1 if bad else 0)
