"""debug aid: one cell-day of the benchmark grid in one-step mode -- the GPU against the oracle as built\n(A) and its two other builds (B, C of oracle/parity.py).  python tools/debug_onestep_variants.py row cell day"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rsplash_amd import api
from oracle.binding import Oracle
from oracle.parity import reference_variants
from gpu_pipeline import device_pipeline
row, c, t = (int(x) for x in sys.argv[1:4])  # row of the 4096-row benchmark grid, cell in the row band, day
p = device_pipeline(api, 20221130, 4096, 4096, row*4096 + c, 1, 365, 1)
hp = p["par"]
prev = np.stack([p["out"][k] for k in ("wn", "snow", "qin_prev", "tdrain", "snwage")])
A = Oracle(); V = reference_variants()
res = {}
for tag, oo in [("A", A)] + list(V.items()):
    ref, st1, ev, _ = oo.grid_transient2(p["doy"], p["year"], p["month"], p["sw"], p["tc"], p["pn"], hp[0], hp[1], hp[2], hp[3], hp[4:16], p["Tt"], p["state0"], prev=prev)
    res[tag] = ref
    print(tag, "tdrain", repr(ref["tdrain"][t,0]), "wn", repr(ref["wn"][t,0]), "bflow", repr(ref["bflow"][t,0]), "ev", ev[t,0])
print("GPU tdrain", repr(p["out"]["tdrain"][t,0]), "wn", repr(p["out"]["wn"][t,0]), "bflow", repr(p["out"]["bflow"][t,0]))
print("prev day state", prev[:, t-1, 0], "par lambda", hp[8,0], "depth", hp[9,0], "SAT", hp[4,0], "RES", hp[11,0], "bub", hp[10,0], 'Au', hp[12,0])
