"""debug aid: where does a trajectory first leave the oracle's, and what happened to td before"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
assert torch.cuda.is_available()
from rsplash_amd import api
from oracle.binding import Oracle
from oracle.parity import _bad, OUT5
from gpu_pipeline import device_pipeline
from test_gpu_parity import CASES
case = sys.argv[1] if len(sys.argv) > 1 else "terrain"
o = Oracle()
seed, Htot, W, cell0, n, nt, terrain, start = CASES[case]
p = device_pipeline(api, seed, Htot, W, cell0, n, nt, terrain, start=start)
hp = p["par"]
ref, st1, ev, hs = o.grid_transient2(p["doy"], p["year"], p["month"], p["sw"], p["tc"], p["pn"], hp[0], hp[1], hp[2], hp[3], hp[4:16], p["Tt"], p["state0"], want_hs=True)
bad = np.zeros(ev.shape, bool)
for k in OUT5:
    bad |= _bad(p["out"][k], ref[k])
cells = np.where(bad.any(0))[0]
print("cells with any bad day:", cells)
for c in cells[:6]:
    t0 = np.argmax(bad[:, c])
    print(f"--- cell {c} first bad day {t0}; lambda {hp[8, c]:.4g} depth {hp[9, c]:.3g} Au {hp[12,c]:.3g} slop {hp[2,c]:.3g}")
    evd = np.where(ev[:, c] != 0)[0]
    print("   event days before:", [(int(t), int(ev[t, c])) for t in evd if t <= t0 + 2][-12:])
    # first day any of the 11 series differs noticeably
    for k in ("tdrain", "qin_prev", "wn"):
        d = np.abs(p["out"][k][:, c] - ref[k][:, c])
        with np.errstate(invalid="ignore"):
            rel = d / np.maximum(np.abs(ref[k][:, c]), 1e-300)
        big = np.where((rel > 1e-7) & (d > 1e-9))[0]
        print(f"   {k}: first day rel>1e-7: {big[:3]}")
    lo = max(0, t0 - 12)
    for t in range(lo, t0 + 3):
        print(f"   t={t} ev={ev[t,c]} td gpu {p['out']['tdrain'][t,c]!r} ref {ref['tdrain'][t,c]!r} qin {p['out']['qin_prev'][t,c]:.6g} {ref['qin_prev'][t,c]:.6g} wn {p['out']['wn'][t,c]:.8f} {ref['wn'][t,c]:.8f}")
from oracle.parity import one_step_report, SERIES11
r1 = one_step_report(o, p)
for c in cells[:3]:
    t0 = np.argmax(bad[:, c])
    print(f"=== one-step errors cell {c}: SAT {hp[4,c]!r} RES {hp[11,c]!r} FC {hp[6,c]!r} lambda {hp[8,c]!r} bub {hp[10,c]!r} Ksat {hp[7,c]!r}")
    for t in range(max(0, t0 - 20), t0 + 2):
        errs = {}
        for k in SERIES11:
            a, b = p["out"][k][t, c], r1["ref"][k][t, c]
            if np.isfinite(a) and np.isfinite(b) and b != 0:
                errs[k] = abs(a - b) / abs(b)
        top = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print(f"   t={t} wn {p['out']['wn'][t,c]!r} bflow {p['out']['bflow'][t,c]:.6g} ro {p['out']['ro'][t,c]:.4g} worst one-step rel: " + ", ".join(f"{k} {v:.1e}" for k, v in top))
