"""Largest deviations of the HIP SOLAR+EVAP path from the real reference's known answers
(tests/golden/ref_evap_kat.npz), per output, with the row's inputs (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rsplash_amd import api
k = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ref_evap_kat.npz"))
out = api.evap_batch(k["lat"], k["elv"], k["sw"], k["n"], k["y"], k["sw_in"], k["tc"], k["slop"], k["asp"],
                     k["snow"], k["nd"])
ref = k["evap"]
names = ("cond", "eet", "pet", "aet", "snowmelt", "sublimation", "econ", "pw", "rn_d", "visc", "pet_max")
for j, name in enumerate(names):
    scale = np.maximum(np.abs(ref[:, j]), 1e-6 * np.abs(ref[:, j]).max())
    e = np.abs(out[:, j] - ref[:, j]) / scale
    i = int(np.argmax(e))
    print(f"{name:12s} max err {e[i]:.3e} row {i} out {out[i, j]!r} ref {ref[i, j]!r}  rows>1e-10: {(e > 1e-10).sum()}")
j = 3
scale = np.maximum(np.abs(ref[:, j]), 1e-6 * np.abs(ref[:, j]).max())
e = np.abs(out[:, j] - ref[:, j]) / scale
for i in np.argsort(-e)[:6]:
    print({n: float(k[n][i]) for n in ("lat", "elv", "sw", "n", "y", "sw_in", "tc", "slop", "asp", "snow", "nd")},
          "aet", out[i, 3], ref[i, 3], "pet", ref[i, 2], "snowmelt", ref[i, 4], "sub", ref[i, 5], "rn_d", ref[i, 8])
