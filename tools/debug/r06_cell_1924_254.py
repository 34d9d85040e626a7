"""One cell of the benchmark grid (row 1924, col 254) whose spin-up the ORACLE computes differently on the GPU box's host CPU
than in the build container (profiles/r06_spinup_parity_sweep2.txt): which days, which series."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.binding import Oracle
from rsplash_amd import api
o = Oracle()
cell = 1924 * 4096 + 254
doy, year, month, mid, nm = o.calendar(2001, 365)
st = o.synth_statics(20221130, 4096, 4096, cell, 1, 1)
sw, tc, pn = o.synth_forcing(20221130, 4096, 4096, cell, 1, 0, doy, st["lat"], st["elev"])
dates = np.datetime64("2001-01-01") + np.arange(365)
r, aux = o.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"], st["soil"], st["au"], st["resolution"])
print("oracle here: AI %.16g passes %d critical %d fragile %d state0 %s" % (aux["AI"][0], aux["passes2"][0], aux["critical"][0], aux["fragile"][0], aux["state0"].ravel()))
for be in ("host", "hip"):
    try:
        g = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"],
                            time_index=dates, sim_control=dict(monthly_out=False), want_aux=True, backend=be)
    except Exception as e:
        print(be, "unavailable:", str(e)[:80]); continue
    print(be, "AI %.16g passes %d" % (g["aux"]["AI"][0], g["aux"]["passes2"][0]))
    for k in ("pet", "aet", "wn", "ro", "netr", "cond", "snow"):
        d = np.abs(g[k][:, 0] - r[k][:, 0]) / np.maximum(np.abs(r[k][:, 0]), 1e-12)
        i = int(np.argmax(d))
        print("  ", k, "max rel %.2e on day %d (%.10g vs %.10g), days > 1e-6: %d" % (d.max(), i, g[k][i, 0], r[k][i, 0], int((d > 1e-6).sum())))
# the first pass of spin-up #1, day by day, through the module mirror of the oracle (pet of pass 1) against the host backend's
si, wmax = o.soil_info(st["soil"][:, 0], st["au"][:, 0], st["resolution"][0])
Tt, rain, snowf = o.snow_partition(tc[:, 0], pn[:, 0], month, st["lat"][0], st["elev"][0])
sp = o.spin_up(st["lat"][0], st["elev"][0], 365, 2001, sw[:, 0], tc[:, 0], rain, st["slop"][0], st["asp"][0] - 180, snowf, si)
print("oracle spin_up #1: passes", sp["passes"], "sum pet %.12g sum P %.12g AI %.16g" % (np.nansum(sp["pet"]), np.nansum(rain + snowf), np.nansum(sp["pet"]) / np.nansum(rain + snowf)))
m = api.SPLASH(st["lat"][0], st["elev"][0], backend="host")
hp = m.spin_up(365, 2001, sw[:, 0], tc[:, 0], rain, st["slop"][0], st["asp"][0] - 180, snowf, si)
d = np.abs(hp["pet"] - sp["pet"]) / np.maximum(np.abs(sp["pet"]), 1e-12)
print("host spin_up #1 pet vs oracle: max rel %.2e on day %d; days > 1e-9: %s" % (d.max(), int(np.argmax(d)), np.where(d > 1e-9)[0][:20]))
import platform, subprocess
print(platform.processor(), subprocess.run("grep -m1 'model name' /proc/cpuinfo", shell=True, capture_output=True, text=True).stdout.strip())
