"""rsplash_amd/grid.py -- splash.grid with the reference's signature (R/splash.grid.R:28): row blocks in flight, the
series in year chunks, the nine NetCDF bricks streamed block by block -- run END TO END IN THE BUILD CONTAINER on the
host backend (ABI 6: the library's own day step on CPU threads; the static rasters handed over, since terrain and
upslope area are device kernels).  The GPU suite runs the same loop on the device
(tests/test_gpu_parity.py::test_splash_grid_end_to_end_from_a_dem_in_row_blocks); here the host logic around the
library is what is under test: blocks tile the grid once, three in flight, chunks continue bit for bit, files hold the
float32 of the one-call result under the reference's names, a failed run leaves nothing behind."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def api():
    import rsplash_amd
    rsplash_amd.build_library()
    from rsplash_amd import api as a
    return a


def _case(oracle, H=9, W=12, nt=730, seed=31):
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    st = oracle.synth_statics(seed, H, W, 0, H * W, 1)
    sw, tc, pn = oracle.synth_forcing(seed, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    rng = np.random.default_rng(seed)
    na = rng.random((H, W)) < 0.1
    dem = st["elev"].reshape(H, W).copy()
    dem[na] = np.nan
    r2 = lambda a: a.reshape(H, W)
    statics = dict(lat=r2(st["lat"]), slop=r2(st["slop"]), asp=r2(st["asp"]), resolution=r2(st["resolution"]),
                   Au=st["au"].reshape(3, H, W))
    r3 = lambda a: a.reshape(nt, H, W).astype(np.float32)
    return dict(H=H, W=W, nt=nt, dates=dates, na=na, dem=dem, statics=statics, soil=st["soil"].reshape(6, H, W),
                sw=r3(sw), tc=r3(tc), pn=r3(pn))


RET = dict(wn="wn", ro="ro", pet="pet", aet="aet", snow="snow", bflow="bflow", cond="cond", netr="netrad", sm_lim="sm_lim")


def test_raster_level_splash_grid_on_the_host_blocks_chunks_and_bricks(api, oracle, tmp_path):
    from scipy.io import netcdf_file
    from rsplash_amd import grid, io
    c = _case(oracle)
    H, W, nt, st = c["H"], c["W"], c["nt"], c["statics"]
    R = lambda v, z=None: grid.Raster(v, ymax=46.5, xmin=7.0, yres=0.01, xres=0.01, z=z)
    bricks = [R(c[k], c["dates"]) for k in ("sw", "tc", "pn")]
    elev, soil = R(c["dem"]), R(c["soil"])
    basin = (np.arange(H * W).reshape(H, W) % 3).astype(np.int32)
    with pytest.raises(ValueError, match="statics"):      # terrain and upslope area are device kernels
        grid.splash_grid(*bricks, elev, soil, sim_control=dict(inmem=True), backend="host")
    seen = []
    kw = dict(statics=st, want_diag=True, basin=basin, backend="host")
    g = grid.splash_grid(*bricks, elev, soil, sim_control=dict(monthly_out=True, inmem=True), block_rows=2, inflight=3,
                         progress=lambda i, n: seen.append((i, n)), **kw)
    assert g["blocks"] == [(r, min(2, H - r)) for r in range(0, H, 2)] and seen[-1] == (5, 5)
    one = api.splash_grid(c["sw"], c["tc"], c["pn"], c["dem"], c["soil"], st["lat"], st["slop"], st["asp"], st["Au"],
                          st["resolution"], time_index=c["dates"], sim_control=dict(monthly_out=True), want_diag=True,
                          basin=basin, n_basin=3, backend="host")
    for k, rk in RET.items():
        assert g[rk].shape == (24, H, W)
        assert np.array_equal(g[rk].view(np.int64), one[k].view(np.int64)), k          # blocks == one call, bit for bit
        assert np.isnan(g[rk][:, c["na"]]).all() and np.isfinite(g[rk][:, ~c["na"]]).all(), k
    assert np.array_equal(g["diag_acc"], one["diag_acc"])                                  # integer limbs: sums over blocks
    # the series in year chunks (ABI 5: threshold pre-pass, state carried from chunk to chunk), daily, to disk
    out = str(tmp_path / "out")
    gd = grid.splash_grid(*bricks, elev, soil, outdir=out, sim_control=dict(monthly_out=False, inmem=False),
                          chunk_years=1, block_rows=4, statics=st, backend="host")
    oned = api.splash_grid(c["sw"], c["tc"], c["pn"], c["dem"], c["soil"], st["lat"], st["slop"], st["asp"], st["Au"],
                           st["resolution"], time_index=c["dates"], sim_control=dict(monthly_out=False), backend="host")
    assert os.path.basename(gd["wn"]) == "SPLASH_v2.0_sm_daily_2001-2002.nc"                # writeStart()'s name, R/splash.grid.R:321
    assert sorted(os.listdir(out)) == sorted(os.path.basename(gd[rk]) for rk in RET.values())
    for k, rk in RET.items():
        with netcdf_file(gd[rk], "r", mmap=False) as fo:
            a = np.array(fo.variables[io.BRICKS[k][1]][:])
            want = np.where(np.isnan(oned[k]), io.NA_FLT4S, oned[k]).astype(np.float32)
            assert a.shape == (nt, H, W) and np.array_equal(a, want), k                  # chunks == the undivided run
    # a run that fails mid-way leaves no partial bricks (io.BrickWriter.discard)
    bad = str(tmp_path / "bad")

    def boom(i, n):
        if i == 2:
            raise RuntimeError("stop here")
    with pytest.raises(RuntimeError, match="stop here"):
        grid.splash_grid(*bricks, elev, soil, outdir=bad, sim_control=dict(monthly_out=True, inmem=False), block_rows=2,
                         statics=st, backend="host", progress=boom)
    assert not os.path.exists(bad) or os.listdir(bad) == []


def test_snow_threshold_on_the_host_and_in_chunks(api, oracle):
    """K2 on host arrays with backend="host" (splash_snow_threshold_run_backend): against the oracle's threshold, NaN
    for a gap, -inf where no day qualifies; the maximum over time chunks is the threshold of the whole series; float32
    forcing gives the threshold of its values"""
    c = _case(oracle, H=2, W=16, nt=400)
    H, W = c["H"], c["W"]
    tc = c["tc"].astype(np.float64).reshape(c["nt"], H * W)
    lat, elev = c["statics"]["lat"].ravel(), np.where(c["na"], 500.0, c["dem"]).ravel()
    tc[:, 1] = 40.0               # never cold enough: -inf
    tc[17, 2] = np.nan            # a gap: NA
    tt = api.snow_threshold(tc, lat, elev, backend="host")
    ref = np.array([oracle.snow_partition(tc[:, j], np.zeros(c["nt"]), np.ones(c["nt"], np.int32), lat[j], elev[j])[0]
                    for j in range(H * W)])
    assert np.array_equal(tt, ref, equal_nan=True)
    assert tt[1] == -np.inf and np.isnan(tt[2])
    parts = [api.snow_threshold(tc[a:b], lat, elev, backend="host") for a, b in ((0, 100), (100, 365), (365, 400))]
    assert np.array_equal(np.maximum(np.maximum(parts[0], parts[1]), parts[2]), tt, equal_nan=True)
    t32 = api.snow_threshold(tc.astype(np.float32), lat, elev, backend="host")
    assert np.array_equal(t32, api.snow_threshold(tc.astype(np.float32).astype(np.float64), lat, elev, backend="host"),
                          equal_nan=True)
