import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
assert torch.cuda.is_available()
from rsplash_amd import api
from oracle.binding import Oracle
o = Oracle()
rng = np.random.default_rng(21)
n = 20000
st = o.synth_statics(5150, 100, 200, 0, n, 1)
par = np.empty((17, n))
for i in range(n):
    si, wmax = o.soil_info(st["soil"][:, i], st["au"][:, i], st["resolution"][i])
    par[4:16, i], par[16, i] = si, wmax
par[0], par[1], par[2], par[3] = st["lat"], st["elev"], st["slop"], st["asp"] - 180
SAT, RES, Ksat = par[4], par[11], par[7]
u = rng.random(n)
wn = RES + (SAT - RES) * u
pick = rng.integers(0, 10, n)
wn = np.where(pick == 0, RES, np.where(pick == 1, SAT, np.where(pick == 2, RES - 1.0, wn)))
kv = Ksat * rng.uniform(0.4, 1.6, n)
P = np.where(rng.random(n) < 0.2, 0.0, kv * 6.0 * np.exp(rng.normal(0.5, 1.5, n)))
th, inf = api.infiltration_probe(par, wn, kv, P)
L = o.lib
th_ref = np.array([L.orc_moist_surf(par[9, i], 10.0, par[10, i], wn[i], SAT[i], RES[i], par[8, i]) for i in range(n)])
err = np.abs(th - th_ref) / np.abs(th_ref)
idx = np.argsort(-err)[:12]
for i in idx:
    print(i, 'err', err[i], 'th', th[i], th_ref[i], 'lambda', par[8, i], 'bub', par[10, i], 'depth', par[9, i], 'wn', wn[i], 'SAT', SAT[i], 'RES', RES[i], 'pick', pick[i], 'Theta', (wn[i]-RES[i])/(SAT[i]-RES[i]))
print('frac > 1e-11:', (err > 1e-11).mean())
