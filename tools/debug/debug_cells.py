"""debug aid: trajectory of single cells of the benchmark grid, GPU vs oracle, first divergence"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
assert torch.cuda.is_available()
from rsplash_amd import api
from oracle.binding import Oracle
from oracle.parity import one_step_report, SERIES11, _bad
from gpu_pipeline import device_pipeline
o = Oracle()
for cell in [int(x) for x in sys.argv[1:]]:
    p = device_pipeline(api, 20221130, 4096, 4096, cell, 64, 365, 1)
    hp = p["par"]
    ref, st1, ev, hs = o.grid_transient2(p["doy"], p["year"], p["month"], p["sw"], p["tc"], p["pn"], hp[0], hp[1], hp[2], hp[3], hp[4:16], p["Tt"], p["state0"], want_hs=True)
    r1 = one_step_report(o, p)
    c = 0
    print(f"=== cell {cell}: lambda {hp[8,c]:.5g} depth {hp[9,c]:.3g} SAT {hp[4,c]:.6g} RES {hp[11,c]:.6g} Au {hp[12,c]:.3g} slop {hp[2,c]:.3g} lat {hp[0,c]:.4g} state0 {p['state0'][:,c]}")
    print("   events:", [(int(t), int(ev[t, c])) for t in np.where(ev[:, c] != 0)[0]][:20])
    first = {}
    for k in SERIES11:
        a, b = p["out"][k][:, c], ref[k][:, c]
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
        big = np.where((rel > 1e-9) & (np.abs(a - b) > 1e-12))[0]
        first[k] = int(big[0]) if len(big) else None
    print("   first day rel>1e-9 (trajectory):", first)
    t0 = min([v for v in first.values() if v is not None] + [364])
    for t in range(max(0, t0 - 3), min(365, t0 + 6)):
        errs = {}
        for k in SERIES11:
            a, b = p["out"][k][t, c], r1["ref"][k][t, c]
            if np.isfinite(a) and np.isfinite(b) and b != 0:
                errs[k] = abs(a - b) / abs(b)
        top = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print(f"   t={t} ev={ev[t,c]} wn {p['out']['wn'][t,c]!r}/{ref['wn'][t,c]!r} td {p['out']['tdrain'][t,c]:.6g}/{ref['tdrain'][t,c]:.6g} qin {p['out']['qin_prev'][t,c]:.6g}/{ref['qin_prev'][t,c]:.6g} ro {p['out']['ro'][t,c]:.6g}/{ref['ro'][t,c]:.6g} | one-step worst: " + ", ".join(f"{k} {v:.1e}" for k, v in top))
