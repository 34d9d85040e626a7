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
    assert os.path.basename(paths["wn"]) == f"SPLASH_v2.0_sm_{tt}_2014-2015.nc"
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


def test_streamed_bricks_equal_the_one_shot_writer(tmp_path):
    """io.BrickWriter (header laid out by hand, data memory-mapped, row blocks stored as they arrive:
    R/splash.grid.R:370-400) produces the files io.write_bricks (scipy) produces: the same dimensions,
    variables, attributes and values when read back (scipy orders the variables differently in the header)."""
    t = np.arange("2014-01-01", "2016-01-01", dtype="datetime64[D]")
    H, W = 5, 7
    lat, lon = 10.25 - 0.5 * np.arange(H), -70.25 + 0.5 * np.arange(W)
    rng = np.random.default_rng(0)
    for monthly in (True, False):
        nl = 24 if monthly else 730
        res = {k: rng.uniform(0, 300, (nl, H, W)) for k in io.BRICKS}
        res["wn"][:, 2, 3] = np.nan
        a, b = tmp_path / f"s{monthly}", tmp_path / f"o{monthly}"
        a.mkdir(), b.mkdir()
        with io.BrickWriter(str(a), list(io.BRICKS), lon, lat, t, monthly) as w:
            for k in io.BRICKS:
                w.write(k, 3, res[k][:, 3:5])     # blocks arrive in any order
                w.write(k, 0, res[k][:, 0:3])
            with pytest.raises(ValueError, match="does not fit"):
                w.write("wn", 4, res["wn"][:, 0:3])
        ref = io.write_bricks(str(b), res, lon, lat, t, monthly)
        assert set(w.paths) == set(ref)
        for k in ref:
            assert os.path.basename(w.paths[k]) == os.path.basename(ref[k])
            assert os.path.getsize(w.paths[k]) == os.path.getsize(ref[k])
            with netcdf_file(w.paths[k], "r", mmap=False) as f, netcdf_file(ref[k], "r", mmap=False) as g:
                assert f.dimensions == g.dimensions and set(f.variables) == set(g.variables) and f.title == g.title
                for name in g.variables:
                    vf, vg = f.variables[name], g.variables[name]
                    assert vf.dimensions == vg.dimensions and vf.data.dtype == vg.data.dtype, (k, name)
                    assert np.array_equal(vf[:], vg[:]), (k, name)
                    assert vf._attributes.keys() == vg._attributes.keys(), (k, name)
                    for an, av in vg._attributes.items():
                        assert np.array_equal(vf._attributes[an], av), (k, name, an)


def test_unwritten_stripes_read_back_as_na_and_a_failed_run_leaves_no_bricks(tmp_path):
    """the streamed brick is created with all-zero data; a row block or layer that never arrives must
    read back as the fill value (raster's FLT4S NA), not as 0.0 -- and a run that fails midway removes its partial
    files instead of leaving bricks whose missing blocks look like data (R/splash.grid.R:370-400 writes blocks as they
    arrive; raster's writeStart initialises with NA)."""
    t = np.arange("2014-01-01", "2015-01-01", dtype="datetime64[D]")
    H, W = 4, 3
    lat, lon = 10.25 - 0.5 * np.arange(H), -70.25 + 0.5 * np.arange(W)
    a = tmp_path / "partial"
    a.mkdir()
    with io.BrickWriter(str(a), ["wn"], lon, lat, t, True) as w:
        w.write("wn", 1, np.full((12, 2, W), 5.0))            # rows 1-2 of 0..3, all layers
        w.write("wn", 3, np.full((6, 1, W), 7.0), t0=6)       # row 3, layers 6..11 only
    with netcdf_file(w.paths["wn"], "r", mmap=False) as f:
        v = f.variables[io.BRICKS["wn"][1]][:]
    na = np.float32(io.NA_FLT4S)
    assert (v[:, 1:3] == 5.0).all() and (v[6:, 3] == 7.0).all()
    assert (v[:, 0] == na).all() and (v[:6, 3] == na).all()
    b = tmp_path / "failed"
    b.mkdir()
    with pytest.raises(RuntimeError, match="midway"):
        with io.BrickWriter(str(b), ["wn", "ro"], lon, lat, t, True) as w2:
            w2.write("wn", 0, np.zeros((12, 2, W)))
            raise RuntimeError("the block loop failed midway")
    assert os.listdir(str(b)) == [] and w2.paths == {}


def test_classic_header_of_a_brick_beyond_4_gib(tmp_path):
    """the data variable is the last one, so it may exceed 4 GiB: its vsize field is then 2^32 - 1
    (NetCDF classic format specification) and its offset is a 64-bit one; scipy opens the file"""
    path = str(tmp_path / "big.nc")
    lay = io.nc3_create(path, {"time": 365, "lon": 2048, "lat": 2048},
                        [("lon", io._NC_DOUBLE, ("lon",), {}), ("v", io._NC_FLOAT, ("time", "lat", "lon"), {"units": "mm"})])
    off, shp, dt = lay["v"]
    assert shp == (365, 2048, 2048) and os.path.getsize(path) == off + 365 * 2048 * 2048 * 4   # (sparse)
    m = np.memmap(path, dtype=dt, mode="r+", offset=off, shape=shp)
    m[364, 2047, 2040:] = np.arange(8)
    m.flush()
    del m
    with netcdf_file(path, "r", mmap=True) as f:
        v = f.variables["v"]
        assert v.shape == shp and v.units == b"mm"
        assert np.array_equal(np.array(v[364, 2047, 2040:]), np.arange(8, dtype=np.float32))
        del v
    with pytest.raises(ValueError, match="last variable"):
        io.nc3_create(path, {"a": 1 << 30, "b": 2}, [("x", io._NC_DOUBLE, ("a",), {}), ("y", io._NC_INT, ("b",), {})])


def test_row_blocks_and_raster_slices():
    from rsplash_amd import grid
    for nrow, ncol, kw in ((4096, 4096, {}), (7, 3, dict(block_rows=2)), (1, 5, {}), (100, 10, dict(minblocks=7))):
        bs = grid.row_blocks(nrow, ncol, 3 * 365 * 8, **kw)
        assert bs[0][0] == 0 and sum(n for _, n in bs) == nrow
        assert all(bs[i][0] + bs[i][1] == bs[i + 1][0] for i in range(len(bs) - 1))
        assert len(bs) >= min(nrow, kw.get("minblocks", 2)) or "block_rows" in kw
    assert max(n for _, n in grid.row_blocks(4096, 4096, 3 * 365 * 8)) * 4096 * 3 * 365 * 8 <= 2 << 30
    v = np.arange(2 * 6 * 4, dtype=">f4").reshape(2, 6, 4)       # big-endian, as a NetCDF variable is
    r = grid.Raster(v, ymax=10.0, xmin=-70.0, yres=0.5, xres=0.25, z=["2001-01-01", "2001-01-02"])
    a = r.rows(2, 5)
    assert a.shape == (2, 3, 4) and a.dtype == np.float32 and a.flags.c_contiguous and np.array_equal(a, v[:, 2:5])
    assert np.allclose(r.lat(), [9.75, 9.25, 8.75, 8.25, 7.75, 7.25]) and np.allclose(r.lon()[:2], [-69.875, -69.625])
    e = r.like(np.zeros((6, 4)))
    assert e.rows(0, 6).shape == (6, 4)
    grid._compare([r, e])
    with pytest.raises(ValueError, match="rows/columns"):
        grid._compare([r, r.like(np.zeros((5, 4)))])
    with pytest.raises(ValueError, match="extent"):
        grid._compare([r, grid.Raster(np.zeros((6, 4)), 10.5, -70.0, 0.5, 0.25)])
    with pytest.raises(ValueError, match="resolution"):
        grid._compare([r, grid.Raster(np.zeros((6, 4)), 10.0, -70.0, 0.9, 0.25)])
