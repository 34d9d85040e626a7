"""Latency of the small calls (GPU box): splash_point on the bundled Bourne site (2922 days) and splash_grid on a
few small synthetic grids -- what a caller who loops over sites, as the reference's mapply does, would wait for.
    python tools/time_point.py"""
import csv, datetime, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rsplash_amd import api
from oracle.binding import Oracle
GOLD = os.path.join(ROOT, "tests", "golden")
rows = list(csv.DictReader(open(os.path.join(GOLD, "bourne_forcing.csv"))))
site = json.load(open(os.path.join(GOLD, "bourne_site.json")))
dates = np.array([datetime.date.fromisoformat(r["date"]) for r in rows], dtype="datetime64[D]")
P, Ta, sw = (np.array([float(r[k]) for r in rows]) for k in ("P", "Ta", "sw_in"))
soil = np.array([np.nan if v is None else v for v in site["soil"]])
def best(f, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts), ts[0]
from rsplash_amd._lib import lib
print(f"host backend build: {'AVX2+FMA' if lib.splash_host_backend_fma() else 'generic x86-64'}")
for backend in ("hip", "host"):
    b, first = best(lambda: api.splash_point(sw, Ta, P, site["latitude"], site["elev_m"], site["slop_250m"], site["asp_250m"],
                                             soil, site["Aups_250m"], site["resolution"], time_index=dates, backend=backend))
    print(f"splash_point, Bourne ({len(dates)} days), backend {backend}: first call {first*1e3:.1f} ms, best of 5 {b*1e3:.1f} ms")
o = Oracle()
op = lambda: o.point(*[np.asarray(x) for x in api.time_fields(dates)], api.month_ids(*api.time_fields(dates)[1:]),
                     int(api.month_ids(*api.time_fields(dates)[1:])[-1]) + 1, sw, Ta, P, site["latitude"], site["elev_m"],
                     site["slop_250m"], site["asp_250m"], soil, [site["Aups_250m"], 3, 3], site["resolution"])
b, first = best(op)
print(f"(the parity oracle's plain-C restatement of the same call, one core: best of 5 {b*1e3:.1f} ms)")
for H, W, nt in ((1, 1, 365), (1, 64, 365), (8, 128, 365), (64, 128, 365), (8, 128, 3653)):
    doy, year, month, mid, nm = o.calendar(2001, nt)
    st = o.synth_statics(20221130, H, W, 0, H * W, 1)
    f = o.synth_forcing(20221130, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
    d = np.datetime64("2001-01-01") + np.arange(nt)
    run = lambda: api.splash_grid(*f, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"],
                                  time_index=d, sim_control=dict(monthly_out=True), want_aux=True)
    b, first = best(run, 4)
    r = run()
    runh = lambda: api.splash_grid(*f, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"],
                                   time_index=d, sim_control=dict(monthly_out=True), want_aux=True, backend="host")
    bh, _ = best(runh, 2 if H * W * nt > 2e6 else 4)
    print(f"splash_grid {H}x{W} cells x {nt} days, monthly: hip best of 4 {b*1e3:.1f} ms, host {bh*1e3:.1f} ms; max passes of "
          f"spin-up #2: {int(r['aux']['passes2'].max())}")
