"""Turn the reference's other bundled example data into fixtures (run once in the build container):

  data/ATNeu_example.RData -> tests/golden/atneu_forcing.csv (date,tc,sw_in,pn; 4018 daily rows,
                              2002-01-01..2012-12-31) + tests/golden/atneu_site.json (soil vector)
                              + tests/golden/atneu_obs.csv (the measured aet, pet, netr, sm, snowmelt columns)
  data/SA_cru.RData        -> tests/golden/sa_cru_statics.npz: elev [139*159] and soil [6][139*159]
                              (sand, clay, SOM, fgravel, bdens, depth2br) of the 0.5 degree South
                              America grid, 72 % NA (ocean), stored as float32 (the rasters are FLT4S)

These are DATA the reference ships, not reference source.  The monthly tc/pn/sw_in bricks of SA_cru
are not converted: the reference's path for monthly forcing is the stochastic disaggregation of
R/splash.point.R:66-86 (month2day_rain draws from R's unseeded RNG; SURVEY.md 2 #15: out of scope).
"""
import datetime
import json
import os
import sys

import numpy as np

sys.setrecursionlimit(100000)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rdata  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests/golden")

A = rdata.named_list(rdata.load("/root/reference/data/ATNeu_example.RData")["ATNeu_example"])
f = A["forcing"]
nr, nc = (int(x) for x in f.attrs["dim"].value)
cols = f.attrs["dimnames"].value[1].value
mat = f.value.reshape(nc, nr).T
dates = [datetime.datetime.fromtimestamp(s, datetime.timezone.utc).date() for s in f.attrs["index"].value]
ix = [cols.index(k) for k in ("tc", "SW_in", "pn")]
assert not np.isnan(mat[:, ix]).any()
with open(os.path.join(G, "atneu_forcing.csv"), "w") as fh:
    fh.write("date,tc,sw_in,pn\n")
    for dt, row in zip(dates, mat):
        fh.write(dt.isoformat() + "," + ",".join(repr(float(row[i])) for i in ix) + "\n")
# the measurements that come with the forcing (FLUXNET AT-Neu: evapotranspiration from the eddy-covariance latent heat
# flux, net radiation, soil moisture of the sensor depth; NA kept as empty fields): examples/readme_bourne.py scores
# the simulation against them
ox = [cols.index(k) for k in ("aet", "pet", "netr", "sm", "snowmelt")]
with open(os.path.join(G, "atneu_obs.csv"), "w") as fh:
    fh.write("date,aet_obs,pet_obs,netr_obs,sm_obs,snowmelt_obs\n")
    for dt, row in zip(dates, mat):
        fh.write(dt.isoformat() + "," + ",".join("" if np.isnan(row[i]) else repr(float(row[i])) for i in ox) + "\n")
site = {
    "soil": [float(v) for v in A["soil"].value],
    "soil_names": ["sand", "clay", "OM", "fgravel", "bd", "depth"],
    "latitude": 47.1167, "elev_m": 970.0,
    "note": "soil from data/ATNeu_example.RData; the file holds no site metadata: latitude and elevation "
            "are the published FLUXNET coordinates of AT-Neu (Neustift) and only need to be the same on "
            "both sides of a parity test",
    "source": "data/ATNeu_example.RData of dsval/rsplash (decoded by tools/rdata.py)",
}
json.dump(site, open(os.path.join(G, "atneu_site.json"), "w"), indent=1)
print("ATNeu", nr, "rows", dates[0], dates[-1], site["soil"])

S = rdata.named_list(rdata.load("/root/reference/data/SA_cru.RData")["SA_cru"])
elev = S["elev"].attrs["data"].attrs["values"].value
soil = S["soil"].attrs["data"].attrs["values"].value.reshape(6, -1)  # [cells x 6] column-major
nrows, ncols = int(S["elev"].attrs["nrows"].value[0]), int(S["elev"].attrs["ncols"].value[0])
ext = S["elev"].attrs["extent"].attrs
assert elev.size == nrows * ncols and soil.shape[1] == elev.size
assert np.array_equal(elev.astype(np.float32).astype(np.float64)[~np.isnan(elev)], elev[~np.isnan(elev)])
np.savez_compressed(os.path.join(G, "sa_cru_statics.npz"), elev=elev.astype(np.float32),
                    soil=soil.astype(np.float32), nrows=nrows, ncols=ncols,
                    extent=np.array([float(ext[k].value[0]) for k in ("xmin", "xmax", "ymin", "ymax")]),
                    soil_names=np.array(S["soil"].attrs["data"].attrs["names"].value))
print("SA_cru", nrows, ncols, "land cells", int((~np.isnan(elev)).sum()),
      "soil float32-exact:", bool(np.array_equal(soil.astype(np.float32).astype(np.float64)[~np.isnan(soil)],
                                                soil[~np.isnan(soil)])))
