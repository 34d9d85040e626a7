"""Output bricks (R/splash.grid.R:186-216): file names, variable names, units, axes."""
import os

import numpy as np
import pytest
from scipy.io import netcdf_file

from rsplash_amd import io


@pytest.mark.parametrize("monthly", [True, False])
def test_bricks_are_named_like_the_reference(tmp_path, monthly):
    t = np.arange("2014-01-01", "2016-01-01", dtype="datetime64[D]")
    z, zval, zunit, y0, y1 = io.output_time_axis(t, monthly)
    assert (y0, y1) == (2014, 2015) and zunit == "days since 2013-12-31"
    assert z.size == (24 if monthly else 730) and zval[0] == 1 and (zval[1] == (32 if monthly else 2))
    H, W = 5, 7
    rng = np.random.default_rng(0)
    res = {k: rng.uniform(0, 300, (z.size, H, W)) for k in io.BRICKS}
    res["wn"][:, 2, 3] = np.nan  # an NA cell
    lat, lon = 10.25 - 0.5 * np.arange(H), -70.25 + 0.5 * np.arange(W)
    paths = io.write_bricks(str(tmp_path), res, lon, lat, t, monthly)
    tt = "monthly" if monthly else "daily"
    assert os.path.basename(paths["wn"]) == f"SPLASH_v2.0_swc_{tt}_2014-2015.nc"
    assert os.path.basename(paths["netr"]) == f"SPLASH_v2.0_netrad_{tt}_2014-2015.nc"
    assert os.path.basename(paths["snow"]) == f"SPLASH_v2.0_swe_{tt}_2014-2015.nc"
    assert len(paths) == 9
    with netcdf_file(paths["wn"], "r", mmap=False) as f:
        assert set(f.dimensions) == {"lon", "lat", "time"}
        v = f.variables["wm"]
        assert v.units == b"mm" and v.long_name == b"Soil water content"
        assert v.shape == (z.size, H, W)
        a = v[:].astype(np.float64)
        assert a[0, 2, 3] == pytest.approx(float(io.NA_FLT4S))
        ok = ~np.isnan(res["wn"])
        assert np.allclose(a[ok], res["wn"][ok], rtol=1e-6)
        assert np.allclose(f.variables["lat"][:], lat) and np.allclose(f.variables["lon"][:], lon)
        assert f.variables["time"].units == zunit.encode()
        assert np.array_equal(f.variables["time"][:], zval)
    with netcdf_file(paths["netr"], "r", mmap=False) as f:
        assert f.variables["netrad"].units == b"MJ"


def test_partial_years_are_refused(tmp_path):
    t = np.arange("2014-01-01", "2014-07-01", dtype="datetime64[D]")
    res = {"wn": np.zeros((6, 2, 2))}
    with pytest.raises(ValueError, match="whole calendar years"):
        io.write_bricks(str(tmp_path), res, [0, 1], [1, 0], t, True)
