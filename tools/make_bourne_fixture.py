"""Turn the reference's bundled example data (data/Bourne.RData) into a plain-text fixture.

Run once in the build container (needs /root/reference):
    python tools/make_bourne_fixture.py
Writes tests/golden/bourne_forcing.csv (date,P,Ta,sw_in,swe_obs,sm_obs; 2922 daily rows,
2008-01-01..2015-12-31) and tests/golden/bourne_site.json (site metadata used by the README
call, README.md:31-43).  These are DATA the reference ships, not reference source.
"""
import datetime
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rdata  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = rdata.load("/root/reference/data/Bourne.RData")
B = rdata.named_list(d["Bourne"])
f = B["forcing"]
nr, nc = (int(x) for x in f.attrs["dim"].value)
cols = f.attrs["dimnames"].value[1].value
mat = f.value.reshape(nc, nr).T  # R matrices are column-major
secs = f.attrs["index"].value
dates = [datetime.datetime.fromtimestamp(s, datetime.timezone.utc).date() for s in secs]
assert cols == ["P", "Ta", "sw_in", "swe", "sm"], cols
with open(os.path.join(ROOT, "tests/golden/bourne_forcing.csv"), "w") as fh:
    fh.write("date,P,Ta,sw_in,swe_obs,sm_obs\n")
    for dt, row in zip(dates, mat):
        fh.write(dt.isoformat() + "," + ",".join(repr(float(v)) for v in row) + "\n")
md = rdata.named_list(B["md"])
site = {
    "soil": [None if np.isnan(v) else float(v) for v in B["soil"].value],
    "soil_names": B["soil"].attrs["names"].value,
    "max_depth_sm": float(B["max_depth_sm"].value[0]),
    "latitude": float(md["latitude"].value[0]),
    "elev_m": float(md["elev_m"].value[0]),
    "slop_250m": float(md["slop_250m"].value[0]),
    "asp_250m": float(md["asp_250m"].value[0]),
    "Aups_250m": float(md["Aups_250m"].value[0]),
    "resolution": 250.0,
    "source": "data/Bourne.RData of dsval/rsplash (decoded by tools/rdata.py)",
}
json.dump(site, open(os.path.join(ROOT, "tests/golden/bourne_site.json"), "w"), indent=1)
print(nr, "rows", dates[0], dates[-1], site)
