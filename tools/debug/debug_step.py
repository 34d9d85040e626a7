"""Debug helper (GPU box): module-mirror spin_up vs oracle on synthetic cells; first diverging day."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rsplash_amd
rsplash_amd.build_library()
from rsplash_amd import api
from oracle.binding import Oracle
o = Oracle()
terrain = int(sys.argv[1]) if len(sys.argv) > 1 else 0
H, W, nt = 12, 32, 365
doy, year, month, mid, nm = o.calendar(2001, nt)
st = o.synth_statics(20221130, H, W, 0, H * W, terrain)
sw, tc, pn = o.synth_forcing(20221130, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
nbad = 0
for j in range(0, H * W, 1):
    si, wmax = o.soil_info(st["soil"][:, j], st["au"][:, j], st["resolution"][j])
    Tt, rain, snf = o.snow_partition(tc[:, j], pn[:, j], month, st["lat"][j], st["elev"][j])
    asp = st["asp"][j] - 180
    o.lib.orc_fragile_events.restype = __import__("ctypes").c_long
    o.lib.orc_fragile_events(1)
    a = o.spin_up(st["lat"][j], st["elev"][j], 365, 2001, sw[:, j], tc[:, j], rain, st["slop"][j], asp, snf, si)
    nfrag = o.lib.orc_fragile_events(1)
    if nfrag:
        continue
    g = api.SPLASH(st["lat"][j], st["elev"][j]).spin_up(365, 2001, sw[:, j], tc[:, j], rain, st["slop"][j], asp, snf, si)
    bad = None
    for k in ("sm", "snow", "pet", "ro", "qin"):
        d = np.abs(g[k] - a[k]) > 1e-9 * np.abs(a[k]) + 1e-12
        if d.any():
            t = int(np.argmax(d))
            if bad is None or t < bad[0]:
                bad = (t, k)
    if bad and a["passes"] == g["passes"]:
        t, k = bad
        nbad += 1
        if nbad <= 6:
            print(f"cell {j} lat {st['lat'][j]:.1f} depth {si[5]:.3f} lam {si[4]:.3f} bub {si[6]:.1f} SAT {si[0]:.1f} RES {si[7]:.1f} passes {a['passes']}/{g['passes']}: first diff day {t} in {k}: gpu {g[k][t]!r} cpu {a[k][t]!r}")
            for kk in ("sm", "snow", "pet", "ro", "qin", "tdrain"):
                print("    ", kk, g[kk][max(t-1,0):t+2], a[kk][max(t-1,0):t+2])
            print("     forcing tc", tc[t, j], "rain", rain[t], "snf", snf[t], "sw", sw[t, j])
            sp = api.SPLASH(st["lat"][j], st["elev"][j])
            r = sp.run_all(doy, year, sw[:, j], tc[:, j], rain, si[7], st["slop"][j], asp, 0.0, snf, si, 0.0, 0.0, 0.0)
            q = o.run_all(st["lat"][j], st["elev"][j], doy, year, sw[:, j], tc[:, j], rain, si[7], st["slop"][j], asp, 0.0, snf, si, 0.0, 0.0, 0.0)
            for kk in ("wn", "ro", "aet", "pet", "bflow", "qin_prev", "tdrain", "snow"):
                dd = np.abs(r[kk] - q[kk]) > 1e-9 * np.abs(q[kk]) + 1e-12
                if dd.any():
                    tt = int(np.argmax(dd))
                    print("   pass-1 first diff", kk, "day", tt, r[kk][max(tt-1,0):tt+2], q[kk][max(tt-1,0):tt+2], "rain", rain[tt], "tc", tc[tt, j], "slop", st["slop"][j], "Au", si[8], "cin", si[10])
    elif a["passes"] != g["passes"]:
        nbad += 1
        print(f"cell {j}: passes differ {a['passes']} vs {g['passes']}  lat {st['lat'][j]:.1f} slop {st['slop'][j]:.4f} depth {si[5]:.3f} lam {si[4]:.3f} bub {si[6]:.1f} SAT {si[0]:.1f} RES {si[7]:.1f} Au {si[8]:.3e} cin {si[10]}")
        # first pass only: replay day by day with the module run_all from the initial state
        sp = api.SPLASH(st["lat"][j], st["elev"][j])
        r = sp.run_all(doy, year, sw[:, j], tc[:, j], rain, si[7], st["slop"][j], asp, 0.0, snf, si, 0.0, 0.0, 0.0)
        q = o.run_all(st["lat"][j], st["elev"][j], doy, year, sw[:, j], tc[:, j], rain, si[7], st["slop"][j], asp, 0.0, snf, si, 0.0, 0.0, 0.0)
        for k in ("wn", "ro", "aet", "bflow", "qin_prev", "tdrain", "snow"):
            d = np.abs(r[k] - q[k]) > 1e-9 * np.abs(q[k]) + 1e-12
            if d.any():
                t = int(np.argmax(d))
                print("   first diff", k, "day", t, r[k][max(t-1,0):t+2], q[k][max(t-1,0):t+2], "rain", rain[t], "tc", tc[t, j])
print("bad cells", nbad)
