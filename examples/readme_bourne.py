"""The reference's README example (README.md:28-61), line for line, on this library: splash.point on the bundled Bourne
SNOTEL site, simulated snow water equivalent against the observed one, and the simulated soil water content of the
measured part of the profile (unSWC to 0.49 m) against the sensors'.  The reference plots the pairs; this script
prints what the plots show (correlation, bias, Nash-Sutcliffe efficiency) -- the reference has no other test
(SURVEY.md 4), so tests/test_readme_example.py asserts these scores.

    python examples/readme_bourne.py [hip|host]

`host` runs both steps on a CPU core (ABI 6: no GPU touched)."""
import csv
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def load_bourne():
    """data(Bourne): forcing (P, Ta, sw_in), observations (swe, sm), soil, site metadata -- decoded from
    data/Bourne.RData by tools/make_bourne_fixture.py (a data fixture, no reference code)"""
    rows = list(csv.DictReader(open(os.path.join(GOLD, "bourne_forcing.csv"))))
    site = json.load(open(os.path.join(GOLD, "bourne_site.json")))
    f = lambda k: np.array([float(r[k]) if r[k] not in ("", "NA", "nan") else np.nan for r in rows])
    return dict(dates=np.array([r["date"] for r in rows], dtype="datetime64[D]"), P=f("P"), Ta=f("Ta"), sw_in=f("sw_in"),
                swe=f("swe_obs"), sm=f("sm_obs"), soil=np.array([np.nan if v is None else v for v in site["soil"]]),
                md=site, max_depth_sm=site["max_depth_sm"])


def scores(sim, obs):
    ok = np.isfinite(sim) & np.isfinite(obs)
    s, o = sim[ok], obs[ok]
    return dict(n=int(ok.sum()), r=float(np.corrcoef(s, o)[0, 1]), bias=float((s - o).mean()),
                nse=float(1.0 - ((s - o) ** 2).sum() / ((o - o.mean()) ** 2).sum()))


def run(backend="hip", with_unswc=True):
    from rsplash_amd import api
    Bourne = load_bourne()
    md = Bourne["md"]
    # run splash (README.md:33-45)
    run1 = api.splash_point(sw_in=Bourne["sw_in"], tc=Bourne["Ta"], pn=Bourne["P"], lat=md["latitude"], elev=md["elev_m"],
                            slop=md["slop_250m"], asp=md["asp_250m"], soil_data=Bourne["soil"], Au=md["Aups_250m"],
                            resolution=250.0, time_index=Bourne["dates"], backend=backend)
    out = {"swe": scores(run1["snow"], Bourne["swe"])}                      # README.md:49-50
    if with_unswc:  # README.md:52-59: water content of the measured part of the profile
        swc = api.unSWC(soil_data=Bourne["soil"], uns_depth=Bourne["max_depth_sm"], wn=run1["wn"], backend=backend)
        out["swc"] = scores(swc["wn_z"], Bourne["sm"])
    return out, run1


if __name__ == "__main__":
    backend = sys.argv[1] if len(sys.argv) > 1 else "hip"
    res, _ = run(backend)
    for k, v in res.items():
        print(f"{k.upper()} sim. against obs. ({v['n']} days): r = {v['r']:.3f}, bias = {v['bias']:+.1f} mm, NSE = {v['nse']:.3f}")
