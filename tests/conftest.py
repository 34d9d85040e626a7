import csv
import datetime
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def api():
    """the Python mirror over the built library ON A GPU (the GPU test modules; tests/test_host_backend.py has its own,
    device-free one)"""
    import torch
    assert torch.cuda.is_available(), "GPU suite needs a GPU"
    import rsplash_amd
    rsplash_amd.build_library()
    from rsplash_amd import api as a
    assert a.device_count() >= 1
    return a


@pytest.fixture(scope="session")
def bourne():
    """the reference's bundled example site (data/Bourne.RData -> tests/golden, README.md:28-61)"""
    rows = list(csv.DictReader(open(os.path.join(GOLD, "bourne_forcing.csv"))))
    site = json.load(open(os.path.join(GOLD, "bourne_site.json")))
    dates = [datetime.date.fromisoformat(r["date"]) for r in rows]
    d = dict(
        dates=np.array(dates, dtype="datetime64[D]"),
        doy=np.array([x.timetuple().tm_yday for x in dates], np.int32),
        year=np.array([x.year for x in dates], np.int32),
        month=np.array([x.month for x in dates], np.int32),
        P=np.array([float(r["P"]) for r in rows]),
        Ta=np.array([float(r["Ta"]) for r in rows]),
        sw_in=np.array([float(r["sw_in"]) for r in rows]),
        soil=np.array([np.nan if v is None else v for v in site["soil"]]),
        site=site,
    )
    d["month_id"] = ((d["year"] - d["year"][0]) * 12 + d["month"] - 1).astype(np.int32)
    d["nmonth"] = int(d["month_id"][-1]) + 1
    return d


@pytest.fixture(scope="session")
def atneu():
    """the reference's second bundled site (data/ATNeu_example.RData -> tests/golden): 4018 days of
    FLUXNET AT-Neu forcing and a soil vector with a measured bulk density"""
    rows = list(csv.DictReader(open(os.path.join(GOLD, "atneu_forcing.csv"))))
    site = json.load(open(os.path.join(GOLD, "atneu_site.json")))
    dates = [datetime.date.fromisoformat(r["date"]) for r in rows]
    d = dict(
        dates=np.array(dates, dtype="datetime64[D]"),
        doy=np.array([x.timetuple().tm_yday for x in dates], np.int32),
        year=np.array([x.year for x in dates], np.int32),
        month=np.array([x.month for x in dates], np.int32),
        P=np.array([float(r["pn"]) for r in rows]),
        Ta=np.array([float(r["tc"]) for r in rows]),
        sw_in=np.array([float(r["sw_in"]) for r in rows]),
        soil=np.array(site["soil"]),
        site=site,
    )
    d["month_id"] = ((d["year"] - d["year"][0]) * 12 + d["month"] - 1).astype(np.int32)
    d["nmonth"] = int(d["month_id"][-1]) + 1
    return d


@pytest.fixture(scope="session")
def sa_cru():
    """static rasters of the reference's bundled 0.5 degree South America grid (data/SA_cru.RData):
    elev [nrows*ncols], soil [6][nrows*ncols], NaN over the ocean (72 %), and the row latitudes"""
    z = np.load(os.path.join(GOLD, "sa_cru_statics.npz"))
    nrows, ncols = int(z["nrows"]), int(z["ncols"])
    xmin, xmax, ymin, ymax = z["extent"]
    lat_rows = ymax - (np.arange(nrows) + 0.5) * (ymax - ymin) / nrows  # yFromRow: cell centres
    return dict(elev=z["elev"].astype(np.float64), soil=z["soil"].astype(np.float64), nrows=nrows,
                ncols=ncols, lat=np.repeat(lat_rows, ncols))


@pytest.fixture(scope="session")
def minigrid():
    """committed whole-pipeline anchor (tools/make_minigrid_golden.py): the oracle's splash.grid
    outputs on a 64-cell x 730-day synthetic terrain grid, monthly"""
    return dict(np.load(os.path.join(GOLD, "minigrid_monthly.npz")))


@pytest.fixture(scope="session")
def kat():
    return dict(np.load(os.path.join(GOLD, "ref_evap_kat.npz")))


ATOL = 1e-9  # mm (or mm/day): below any physical meaning; guards values that should be ~0


def rel_err(x, ref, atol=ATOL):
    """worst |x-ref| / (|ref| + atol/rtol-free form): returns (max relative error over entries with
    |ref| > 1e-4, max absolute error over all entries).  The NaN pattern must match exactly."""
    x, ref = np.asarray(x, float), np.asarray(ref, float)
    assert x.shape == ref.shape
    nanx, nanr = np.isnan(x), np.isnan(ref)
    assert (nanx == nanr).all(), f"NaN pattern differs: {nanx.sum()} vs {nanr.sum()}"
    ok = ~nanr
    err = np.abs(x[ok] - ref[ok])
    r = np.abs(ref[ok])
    excess = err - atol  # error beyond the absolute floor, relative to |ref|
    rel = float(np.max(np.where(excess > 0, excess / np.maximum(r, 1e-300), 0.0))) if err.size else 0.0
    return rel, float(err.max()) if err.size else 0.0
