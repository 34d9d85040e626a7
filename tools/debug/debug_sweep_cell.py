"""debug aid: one cell of a tools/parity_sweep.py grid, whole pipeline GPU vs oracle: spin-up results,
first day and series of divergence, oracle events around it.
    python tools/debug_sweep_cell.py <grid index g> <cell> [cells_per_grid=8192] [days=730]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rsplash_amd.api as api
from oracle.binding import Oracle
g, cell = int(sys.argv[1]), int(sys.argv[2])
ncell = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
nt = int(sys.argv[4]) if len(sys.argv) > 4 else 730
o = Oracle()
W = 128; H = ncell // W
seed = 777000 + 7919 * g
doy, year, month, mid, nm = o.calendar(2001, nt)
dates = np.datetime64("2001-01-01") + np.arange(nt)
st = o.synth_statics(seed, H, W, cell, 1, 1)
sw, tc, pn = o.synth_forcing(seed, H, W, cell, 1, 0, doy, st["lat"], st["elev"])
ref, aux = o.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"], st["soil"],
                  st["au"], st["resolution"], monthly_out=False)
got = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                      st["resolution"], time_index=dates, sim_control=dict(monthly_out=False), want_aux=True)
print("statics:", {k: (v.ravel().tolist() if hasattr(v, "ravel") else v) for k, v in st.items()})
ga = got["aux"]
print("oracle aux:", {k: aux[k][0].tolist() if hasattr(aux[k][0], "tolist") else aux[k][0] for k in aux.dtype.names})
print("gpu aux   :", {k: np.asarray(v).ravel()[:8].tolist() for k, v in ga.items()})
for k in ref:
    a, b = np.asarray(got[k]).reshape(ref[k].shape)[:, 0], ref[k][:, 0]
    d = np.abs(a - b)
    rel = d / np.maximum(np.abs(b), 1e-300)
    big = np.where((rel > 1e-9) & (d > 1e-10))[0]
    print(f"{k:7s} first day rel>1e-9: {big[:1]}  max rel {np.nanmax(np.where(d > 1e-9, rel, 0)):.3e} at day {int(np.nanargmax(np.where(d > 1e-9, rel, 0)))}")
t0 = min([int(np.where((np.abs(np.asarray(got[k]).reshape(ref[k].shape)[:, 0] - ref[k][:, 0]) > 1e-9 * np.maximum(np.abs(ref[k][:, 0]), 1e-3)))[0][:1].tolist()[0])
          for k in ref if (np.abs(np.asarray(got[k]).reshape(ref[k].shape)[:, 0] - ref[k][:, 0]) > 1e-9 * np.maximum(np.abs(ref[k][:, 0]), 1e-3)).any()] + [nt - 1])
for t in range(max(0, t0 - 3), min(nt, t0 + 8)):
    print(f"t={t} doy {doy[t]} tc {tc[t,0]:.3f} pn {pn[t,0]:.3f} sw {sw[t,0]:.1f} | " + " ".join(
        f"{k} {np.asarray(got[k]).reshape(ref[k].shape)[t,0]:.9g}/{ref[k][t,0]:.9g}" for k in ("wn", "ro", "aet", "pet", "snow", "bflow")))
