import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle.binding import Oracle
import rsplash_amd.api as api
oracle=Oracle()
rng = np.random.default_rng(2024)
for it in range(700):
    soil = [rng.uniform(15, 70), rng.uniform(8, 25), rng.uniform(0.3, 8), rng.uniform(0, 40),
            rng.choice([np.nan, rng.uniform(0.9, 1.6)]), rng.choice([2.0, rng.uniform(0.6, 2.0)])]
    slop = rng.choice([0.0, rng.uniform(0.1, 35)])
    au = [62500.0 * 2 ** rng.uniform(0, 12), 3, 3] if slop > 0 else [0.0, 3, 3]
    si, wmax = oracle.soil_info(np.array(soil), au, 250.0)
    SAT, RES = si[0], si[7]
    lat, elv = rng.choice([rng.uniform(-60, 60), 78.0, -82.0]), rng.uniform(0, 3000)
    asp = rng.uniform(0, 360) - 180
    doy, yr = int(rng.integers(1, 366)), int(rng.choice([0, 2001, 2004]))
    wn = rng.choice([SAT, RES, rng.uniform(RES, SAT), SAT + 5.0, RES - 1.0])
    tc = rng.choice([3.0, 0.0, -0.0, rng.uniform(-25, 35)])
    sw_in = rng.choice([0.0, rng.uniform(5, 420)])
    pn = rng.choice([0.0, rng.uniform(0, 8), rng.uniform(20, 150)])
    snow = rng.choice([0.0, rng.uniform(0, 300)])
    snf = rng.choice([0.0, rng.uniform(0, 30)])
    qin = rng.choice([0.0, rng.uniform(0, 5)])
    td = rng.choice([0.0, rng.uniform(0, 30), np.nan, np.inf])
    nd = float(rng.integers(0, 40))
    if it == int(sys.argv[1]): break
print(dict(soil=soil, slop=slop, au=au, lat=lat, elv=elv, asp=asp, doy=doy, yr=yr, wn=wn, tc=tc, sw_in=sw_in, pn=pn, snow=snow, snf=snf, qin=qin, td=td, nd=nd, SAT=SAT, RES=RES))
print('si', si)
s1,e1=oracle.run_one_day(lat, elv, doy, yr, wn, sw_in, tc, pn, slop, asp, snow, snf, si, qin, td, nd)
print('oracle smr', s1); print('oracle etr', e1)
try:
    g = api.SPLASH(lat, elv).run_all([doy], [yr], [sw_in], [tc], [pn], wn, slop, asp, snow, [snf], si, qin, td, nd)
    print("gpu", {k: float(v[0]) for k,v in g.items()})
except Exception as e:
    print("no gpu:", e)
sw = (wn - RES)/ (wmax - RES) if False else None
import ctypes as C
oracle.lib.orc_debug_ptr.restype = C.POINTER(C.c_double * 16)
dbg = oracle.lib.orc_debug_ptr().contents
print('orc_debug Kb,Q,T,t_drain,q_in_f,q_in_o,R,Ksat_visc:', list(dbg)[:8])
