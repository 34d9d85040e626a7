"""Generate the golden vectors under tests/golden/ (run in the build container only).

  ref_evap_kat.npz   inputs + outputs of the REAL reference classes SOLAR and EVAP
                     (oracle/_ref = /root/reference/src/{global,SOLAR,EVAP}.cpp compiled as they
                     lie): a stratified lattice plus seeded random rows.
                     These pin the oracle (and, on the GPU, the HIP kernels) for src/SOLAR.cpp
                     and src/EVAP.cpp including the float viscosity bit patterns.
  bourne_oracle.json summary of the oracle's splash.point run on the Bourne fixture, next to
                     the numbers SURVEY.md 8c recorded from running the reference C++ itself.

    python tools/make_golden.py
"""
import csv
import datetime
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.binding import Oracle, Ref  # noqa: E402

o, r = Oracle(), Ref()
rng = np.random.default_rng(20221130)

# ---- lattice + random rows
lat_l = [0, 23, -23, 45, -45, 66, -66, 80, -80]
terr_l = [(0, 0), (10, -180 + 45), (35, 0), (60, -180), (25, 90)]  # (slope, asp-180)
n_l = [1, 80, 172, 266, 355]
y_l = [0, 2001, 2004]
rows = []
for la, (sl, asp), n, y in itertools.product(lat_l, terr_l, n_l, y_l):
    for (snow, nd, sw, tc, swin) in [(0, 0, 0.7, 12.0, 210.0), (150, 3, 0.2, -8.0, 60.0),
                                     (40, 0, 1.0, 3.0, 330.0), (0, 12, 0.0, 30.0, 0.0)]:
        rows.append((la, 500.0 + 20 * abs(la), sw, n, y, swin, tc, sl, asp, snow, nd))
lat_, elv_, sw_, n_, y_, swin_, tc_, sl_, asp_, snow_, nd_ = (np.array(c) for c in zip(*rows))
m2 = 3000
lat2 = rng.uniform(-89, 89, m2)
elv2 = rng.uniform(0, 5000, m2)
sw2 = rng.uniform(0, 1.2, m2)
n2 = rng.integers(1, 367, m2)
y2 = rng.choice([0, 1999, 2001, 2004, 2100], m2)
swin2 = np.where(rng.random(m2) < 0.05, 0.0, rng.uniform(0, 450, m2))
tc2 = np.r_[rng.uniform(-40, 45, m2 - 400), np.linspace(-30, 45, 400)]  # dense visc sweep
sl2 = np.where(rng.random(m2) < 0.3, 0.0, rng.uniform(0, 60, m2))
asp2 = rng.uniform(-180, 180, m2)
snow2 = np.where(rng.random(m2) < 0.5, 0.0, rng.uniform(0, 800, m2))
nd2 = np.floor(rng.uniform(0, 40, m2))
inp = dict(lat=np.r_[lat_, lat2], elv=np.r_[elv_, elv2], sw=np.r_[sw_, sw2],
           n=np.r_[n_, n2].astype(np.int32), y=np.r_[y_, y2].astype(np.int32),
           sw_in=np.r_[swin_, swin2], tc=np.r_[tc_, tc2], slop=np.r_[sl_, sl2],
           asp=np.r_[asp_, asp2], snow=np.r_[snow_, snow2], nd=np.r_[nd_, nd2])
inp = {k: (v.astype(np.float64) if v.dtype != np.int32 else v) for k, v in inp.items()}
evap = r.evap_batch(inp["lat"], inp["elv"], inp["sw"], inp["n"], inp["y"], inp["sw_in"], inp["tc"],
                    inp["slop"], inp["asp"], inp["snow"], inp["nd"])
solar = r.solar_batch(inp["lat"], inp["elv"], inp["n"], inp["y"], inp["sw_in"], inp["tc"],
                      inp["slop"], inp["asp"], inp["snow"], inp["nd"], inp["sw"])
np.savez_compressed(os.path.join(ROOT, "tests/golden/ref_evap_kat.npz"), evap=evap, solar=solar,
                    **inp)
print("ref KAT rows:", len(inp["lat"]), "nan rows:", int(np.isnan(evap).any(1).sum()))

# ---- Bourne summary
rows = list(csv.DictReader(open(os.path.join(ROOT, "tests/golden/bourne_forcing.csv"))))
site = json.load(open(os.path.join(ROOT, "tests/golden/bourne_site.json")))
dates = [datetime.date.fromisoformat(x["date"]) for x in rows]
doy = np.array([d.timetuple().tm_yday for d in dates], np.int32)
year = np.array([d.year for d in dates], np.int32)
month = np.array([d.month for d in dates], np.int32)
mid = np.array([(d.year - dates[0].year) * 12 + d.month - 1 for d in dates], np.int32)
P = np.array([float(x["P"]) for x in rows])
Ta = np.array([float(x["Ta"]) for x in rows])
sw = np.array([float(x["sw_in"]) for x in rows])
soil = [np.nan if v is None else v for v in site["soil"]]
res, aux = o.point(doy, year, month, mid, int(mid[-1]) + 1, sw, Ta, P, site["latitude"],
                   site["elev_m"], site["slop_250m"], site["asp_250m"], soil,
                   [site["Aups_250m"], 3, 3], site["resolution"])
summ = {
    "oracle": {
        "soil_info": [float(v) for v in aux["soil_info"]],
        "AI": float(aux["AI"]), "Tt": float(aux["Tt"]),
        "passes": [int(aux["passes1"]), int(aux["passes2"])],
        "state0": [float(v) for v in aux["state0"]],
        "mean": {k: float(res[k].mean()) for k in res},
        "sum": {k: float(res[k].sum()) for k in res},
    },
    # printed in SURVEY.md 8(c) from the survey's run of the unmodified reference C++ on Bourne
    "survey_reference_probe": {
        "SAT": 257.3165, "WP": 23.8159, "FC": 97.9485, "Ksat": 16.1334, "lambda": 0.370499,
        "AI": 1.33596, "mean": {"wn": 160.1884, "ro": 1.10814, "pet": 2.80284, "aet": 1.03750,
                                "snow": 112.1208},
        "passes_per_spin_up": 2,
    },
}
json.dump(summ, open(os.path.join(ROOT, "tests/golden/bourne_oracle.json"), "w"), indent=1)
print(json.dumps(summ["oracle"]["mean"]))
