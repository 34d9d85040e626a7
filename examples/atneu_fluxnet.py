"""The reference's second bundled site (data/ATNeu_example.RData: FLUXNET AT-Neu, Neustift, 2002-2012) comes with the
tower's measurements beside the forcing: evapotranspiration (from the eddy-covariance latent heat flux), net
radiation, soil moisture, snowmelt.  splash.point on the site, scored against them -- what the README does for Bourne
(examples/readme_bourne.py), on the fluxes Bourne has no measurements of.

    python examples/atneu_fluxnet.py [hip|host]"""
import csv
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
from readme_bourne import scores  # noqa: E402


def load_atneu():
    rows = list(csv.DictReader(open(os.path.join(GOLD, "atneu_forcing.csv"))))
    obs = list(csv.DictReader(open(os.path.join(GOLD, "atneu_obs.csv"))))
    site = json.load(open(os.path.join(GOLD, "atneu_site.json")))
    assert [r["date"] for r in rows] == [r["date"] for r in obs]
    f = lambda src, k: np.array([float(r[k]) if r[k] != "" else np.nan for r in src])
    d = dict(dates=np.array([r["date"] for r in rows], dtype="datetime64[D]"), tc=f(rows, "tc"), sw_in=f(rows, "sw_in"),
             pn=f(rows, "pn"), soil=np.array(site["soil"]), site=site)
    d.update({k: f(obs, k + "_obs") for k in ("aet", "pet", "netr", "sm", "snowmelt")})
    return d


def monthly(x, dates):
    m = dates.astype("datetime64[M]").astype(int)
    m -= m[0]
    out = np.full(m[-1] + 1, np.nan)
    for k in range(out.size):
        v = x[m == k]
        if np.isfinite(v).sum() >= 25:     # a month with (nearly) all its days measured
            out[k] = np.nanmean(v)
    return out


def run(backend="hip"):
    from rsplash_amd import api
    A = load_atneu()
    s = A["site"]
    # the tower stands on a valley floor meadow: flat, no lateral flow (README.md:47: asp, Au and resolution can be omitted)
    r = api.splash_point(sw_in=A["sw_in"], tc=A["tc"], pn=A["pn"], lat=s["latitude"], elev=s["elev_m"], slop=0.0, asp=0.0,
                         soil_data=A["soil"], Au=0.0, resolution=250.0, time_index=A["dates"], backend=backend)
    out = {}
    for k in ("aet", "pet", "netr"):
        out[k + " daily"] = scores(r[k], A[k])
        out[k + " monthly"] = scores(monthly(r[k], A["dates"]), monthly(A[k], A["dates"]))
    # soil moisture: the sensor reports volumetric % of a shallow layer, the model the water of the whole column:
    # only the correlation of the relative saturation is comparable
    out["sm_lim vs sensor"] = scores(r["sm_lim"], A["sm"])
    return out, r


if __name__ == "__main__":
    res, _ = run(sys.argv[1] if len(sys.argv) > 1 else "hip")
    for k, v in res.items():
        print(f"{k:18s} sim. against obs. (n = {v['n']:4d}): r = {v['r']:.3f}, bias = {v['bias']:+.2f}, NSE = {v['nse']:.3f}")
