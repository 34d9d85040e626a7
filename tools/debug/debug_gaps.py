import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
assert torch.cuda.is_available()
from rsplash_amd import api
from oracle.binding import Oracle
oracle = Oracle()
H, W, nt = 2, 32, 450
n = H * W
doy, year, month, mid, nm = oracle.calendar(2001, nt)
st = oracle.synth_statics(99, H, W, 0, n, 1)
sw, tc, pn = oracle.synth_forcing(99, H, W, 0, n, 0, doy, st["lat"], st["elev"])
dates = np.datetime64("2001-01-01") + np.arange(nt)
tc[400, 3] = np.nan; tc[100, 7] = np.nan; sw[420, 11] = np.nan; pn[380, 13] = np.nan
g = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"], time_index=dates, sim_control=dict(monthly_out=False), want_aux=True)
o, aux = oracle.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"], st["soil"], st["au"], st["resolution"])
for k in o:
    d = np.isnan(g[k]) != np.isnan(o[k])
    if d.any():
        print(k, "NaN pattern differs at", [(int(t), int(c)) for t, c in np.argwhere(d)[:10]], "count", int(d.sum()))
        for t, c in [tuple(np.argwhere(d)[0]), tuple(np.argwhere(d)[-1])]:
            print(" day", t, "cell", c, "Tt", g["aux"]["Tt"][c], aux["Tt"][c], "tc", tc[t:t+3, c], "pn", pn[t:t+3, c])
            for q in ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr"):
                print("   ", q, g[q][t:t+3, c], o[q][t:t+3, c])
