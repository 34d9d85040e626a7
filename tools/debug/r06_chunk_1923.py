"""Reproduce one chunk of tools/spinup_parity_sweep.py (rows 1923-1926) and look at cell 4350 from both sides."""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from rsplash_amd import api
from oracle.binding import Oracle
from oracle.parity import spinup_summary
from gpu_pipeline import device_pipeline
o = Oracle()
doy, year, month, mid, nm = o.calendar(2001, 365)
W = 4096
r0 = 1923
c = 4350
for rep in range(1):
    p = device_pipeline(api, 20221130, 4096, W, r0 * W, 4 * W, 365, 1, want_passes=True)
    st = p["statics"]
    _, aux = o.grid(doy, year, month, mid, nm, p["sw"], p["tc"], p["pn"], st["lat"], st["elev"], st["slop"], st["asp"],
                    st["soil"], st["au"], st["resolution"])
    print("rep", rep, "product AI %.16g state %s passes %d | oracle AI %.16g state %s passes %d critical %d" % (
        p["par"][15][c], p["state0"][:, c], p["passes"][1][c], aux["AI"][c], aux["state0"][c] if aux["state0"].shape[0] != 5 else aux["state0"][:, c],
        aux["passes2"][c], aux["critical"][c]))
    # the same cell alone, oracle generator
    st1 = o.synth_statics(20221130, 4096, 4096, r0 * W + c, 1, 1)
    sw1, tc1, pn1 = o.synth_forcing(20221130, 4096, 4096, r0 * W + c, 1, 0, doy, st1["lat"], st1["elev"])
    for k in ("lat", "elev", "slop", "asp"):
        print("   ", k, repr(float(st[k][c])), repr(float(st1[k][0])))
    print("    soil equal", np.array_equal(st["soil"][:, c], st1["soil"][:, 0]), "au equal", np.array_equal(st["au"][:, c], st1["au"][:, 0]),
          "forcing differing days sw/tc/pn", int((p["sw"][:, c] != sw1[:, 0]).sum()), int((p["tc"][:, c] != tc1[:, 0]).sum()), int((p["pn"][:, c] != pn1[:, 0]).sum()))
    d = np.where(p["pn"][:, c] != pn1[:, 0])[0]
    if d.size: print("    pn differs on days", d[:10], p["pn"][d[:5], c], pn1[d[:5], 0])
    d = np.where(p["tc"][:, c] != tc1[:, 0])[0]
    if d.size: print("    tc differs on days", d[:10], p["tc"][d[:5], c], tc1[d[:5], 0])
    _, aux1 = o.grid(doy, year, month, mid, nm, p["sw"][:, c:c + 1].copy(), p["tc"][:, c:c + 1].copy(), p["pn"][:, c:c + 1].copy(),
                     st["lat"][c:c + 1], st["elev"][c:c + 1], st["slop"][c:c + 1], st["asp"][c:c + 1], st["soil"][:, c:c + 1].copy(),
                     st["au"][:, c:c + 1].copy(), st["resolution"][c:c + 1])
    print("    oracle on the device-generated inputs of this cell alone: AI %.16g" % aux1["AI"][0])
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(R, "gpurun_out", "r06_cell_1924_254_device_inputs.npz"), sw=p["sw"][:, c], tc=p["tc"][:, c], pn=p["pn"][:, c],
         lat=st["lat"][c], elev=st["elev"][c], slop=st["slop"][c], asp=st["asp"][c], soil=st["soil"][:, c], au=st["au"][:, c],
         resolution=st["resolution"][c])
