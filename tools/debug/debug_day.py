import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rsplash_amd
rsplash_amd.build_library()
from rsplash_amd import api
from oracle.binding import Oracle
o = Oracle()
H, W, nt = 12, 32, 365
doy, year, month, mid, nm = o.calendar(2001, nt)
st = o.synth_statics(20221130, H, W, 0, H * W, 1)
sw, tc, pn = o.synth_forcing(20221130, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
j = int(sys.argv[1]) if len(sys.argv) > 1 else 71
si, wmax = o.soil_info(st["soil"][:, j], st["au"][:, j], st["resolution"][j])
Tt, rain, snf = o.snow_partition(tc[:, j], pn[:, j], month, st["lat"][j], st["elev"][j])
asp = st["asp"][j] - 180
print("si", si, "slop", st["slop"][j])
d2 = np.r_[doy, doy]; y2 = np.r_[year, year]
f2 = lambda x: np.r_[x, x]
g = api.SPLASH(st["lat"][j], st["elev"][j]).run_all(d2, y2, f2(sw[:, j]), f2(tc[:, j]), f2(rain), si[7], st["slop"][j], asp, 0.0, f2(snf), si, 0.0, 0.0, 0.0)
q = o.run_all(st["lat"][j], st["elev"][j], d2, y2, f2(sw[:, j]), f2(tc[:, j]), f2(rain), si[7], st["slop"][j], asp, 0.0, f2(snf), si, 0.0, 0.0, 0.0)
for k in g:
    dd = np.abs(g[k] - q[k]) > 1e-9 * np.abs(q[k]) + 1e-12
    if dd.any():
        t = int(np.argmax(dd))
        print("first diff", k, "day", t, g[k][max(t-2,0):t+2], q[k][max(t-2,0):t+2])
t = 364
for k in g:
    print(k, g[k][t:t+3], q[k][t:t+3])
print("forcing day0", sw[0, j], tc[0, j], rain[0], snf[0])
print("---- mirror spin_up vs run_all (both GPU) vs oracle spin_up")
for slop_v in (st["slop"][j], 0.0):
    for cin in (si[10], 3.0):
        s2 = si.copy(); s2[10] = cin
        sp = api.SPLASH(st["lat"][j], st["elev"][j])
        gs = sp.spin_up(365, 2001, sw[:, j], tc[:, j], rain, slop_v, asp, snf, s2)
        gr = sp.run_all(d2, y2, f2(sw[:, j]), f2(tc[:, j]), f2(rain), s2[7], slop_v, asp, 0.0, f2(snf), s2, 0.0, 0.0, 0.0)
        os_ = o.spin_up(st["lat"][j], st["elev"][j], 365, 2001, sw[:, j], tc[:, j], rain, slop_v, asp, snf, s2)
        print("slop", slop_v, "cellin", cin, "passes", gs["passes"], os_["passes"], "sm[0] mirror", gs["sm"][0], "run_all[365]", gr["wn"][365], "oracle", os_["sm"][0],
              "| sm[364]", gs["sm"][364], gr["wn"][729 if gs["passes"] == 2 else 364], os_["sm"][364])
print("---- vary bub")
for bubv in (si[6], -690.0):
    s2 = si.copy(); s2[6] = bubv
    sp = api.SPLASH(st["lat"][j], st["elev"][j])
    gs = sp.spin_up(365, 2001, sw[:, j], tc[:, j], rain, st["slop"][j], asp, snf, s2)
    gr = sp.run_all(d2, y2, f2(sw[:, j]), f2(tc[:, j]), f2(rain), s2[7], st["slop"][j], asp, 0.0, f2(snf), s2, 0.0, 0.0, 0.0)
    os_ = o.spin_up(st["lat"][j], st["elev"][j], 365, 2001, sw[:, j], tc[:, j], rain, st["slop"][j], asp, snf, s2)
    print("bub", bubv, "passes", gs["passes"], os_["passes"], "sm[0:3] mirror", gs["sm"][:3], "run_all[365:368]", gr["wn"][365:368], "oracle", os_["sm"][:3])
    print("   qin/td end pass1 (run_all 364):", gr["qin_prev"][364], gr["tdrain"][364], " snwage", gr["snwage"][364])
print("---- 3-year replay")
d3 = np.r_[doy, doy, doy]; y3 = np.r_[year, year, year]
f3 = lambda x: np.r_[x, x, x]
sp = api.SPLASH(st["lat"][j], st["elev"][j])
gr = sp.run_all(d3, y3, f3(sw[:, j]), f3(tc[:, j]), f3(rain), si[7], st["slop"][j], asp, 0.0, f3(snf), si, 0.0, 0.0, 0.0)
qr = o.run_all(st["lat"][j], st["elev"][j], d3, y3, f3(sw[:, j]), f3(tc[:, j]), f3(rain), si[7], st["slop"][j], asp, 0.0, f3(snf), si, 0.0, 0.0, 0.0)
print("gpu wn[364,365,729,730]", gr["wn"][[364, 365, 729, 730]], "oracle", qr["wn"][[364, 365, 729, 730]])
print("gpu snow/qin/td/nd at 364:", gr["snow"][364], gr["qin_prev"][364], gr["tdrain"][364], gr["snwage"][364], " at 729:", gr["snow"][729], gr["qin_prev"][729], gr["tdrain"][729], gr["snwage"][729])
