"""SPLASH_BACKEND_HOST (ABI 6): the product's own day step -- the library's device functions compiled for the host
(rsplash_amd/csrc/splash_host_backend.hip) -- against the oracle, on the CPU: these tests need no GPU and run in the
build container every round.  BASELINE.json configs[0] ("splash.point on the bundled Bourne SNOTEL station, CPU
reference path (plumbing, no GPU)") is exercised here as written; the GPU suite compares the two backends with each
other (tests/test_gpu_parity.py::test_host_and_device_backends_agree).

The host backend is an explicit option, never a fallback: the default backend without a device still fails loudly
(tests/test_abi.py::test_no_gpu_means_loud_failure)."""
import time

import numpy as np
import pytest

RTOL, ATOL = 1e-5, 1e-9


@pytest.fixture(scope="module")
def api():
    import rsplash_amd
    rsplash_amd.build_library()
    from rsplash_amd import api as a
    return a


def close(x, ref, name="", rtol=RTOL, atol=ATOL):
    x, ref = np.asarray(x, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert x.shape == ref.shape, (name, x.shape, ref.shape)
    assert (np.isnan(x) == np.isnan(ref)).all(), f"{name}: NaN pattern differs"
    with np.errstate(invalid="ignore"):
        d = np.abs(x - ref)
        bad = ~((d <= rtol * np.abs(ref) + atol) | (x == ref) | np.isnan(ref))
    assert not bad.any(), f"{name}: {int(bad.sum())} of {bad.size} out of tolerance, worst {np.nanmax(d / np.maximum(np.abs(ref), 1e-300)):.3e}"
    with np.errstate(invalid="ignore", divide="ignore"):   # reported: the raw relative error of values that are not tiny
        e = np.where(np.isfinite(ref) & (np.abs(ref) > 1e-6), d / np.abs(ref), 0.0)
    return float(np.nanmax(e)) if e.size else 0.0


@pytest.mark.parametrize("flat", [False, True])
@pytest.mark.parametrize("monthly", [False, True])
def test_splash_point_bourne_on_the_host(api, oracle, bourne, flat, monthly):
    """BASELINE.json configs[0] as written: splash.point on the bundled Bourne SNOTEL site (README.md:28-61;
    R/splash.point.R:29-225), no GPU.  All nine outputs against the oracle."""
    b, s = bourne, bourne["site"]
    slop, asp = (0.0, 0.0) if flat else (s["slop_250m"], s["asp_250m"])
    t0 = time.perf_counter()
    g = api.splash_point(b["sw_in"], b["Ta"], b["P"], s["latitude"], s["elev_m"], slop, asp, b["soil"], s["Aups_250m"],
                         s["resolution"], time_index=b["dates"], monthly_out=monthly, backend="host")
    dt = time.perf_counter() - t0
    o, aux = oracle.point(b["doy"], b["year"], b["month"], b["month_id"], b["nmonth"], b["sw_in"], b["Ta"], b["P"],
                          s["latitude"], s["elev_m"], slop, asp, b["soil"], [s["Aups_250m"], 3, 3], s["resolution"],
                          monthly_out=monthly)
    worst = {k: close(g[k], o[k], f"host point[{'flat' if flat else 'slope'}].{k}") for k in o}
    print(f"host backend, Bourne ({len(b['doy'])} days): {1e3 * dt:.1f} ms; max rel err", {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) < 1e-7      # (the tolerance is 1e-5: the host build sits near rounding level)


@pytest.mark.parametrize("slop,asp", [(0.0, 0.0), (12.0, 215.0)])
def test_splash_point_atneu_on_the_host(api, oracle, atneu, slop, asp):
    b, s = atneu, atneu["site"]
    au = 0.0 if slop == 0.0 else 40000.0
    g = api.splash_point(b["sw_in"], b["Ta"], b["P"], s["latitude"], s["elev_m"], slop, asp, b["soil"], au, 250.0,
                         time_index=b["dates"], monthly_out=False, backend="host")
    o, aux = oracle.point(b["doy"], b["year"], b["month"], b["month_id"], b["nmonth"], b["sw_in"], b["Ta"], b["P"],
                          s["latitude"], s["elev_m"], slop, asp, b["soil"], [au, 3, 3], 250.0)
    worst = {k: close(g[k], o[k], f"host atneu[{slop}].{k}") for k in o}
    print("max rel err", {k: f"{v:.1e}" for k, v in worst.items()})


def test_module_mirrors_on_the_host(api, oracle, bourne):
    """the Rcpp module's methods (src/SPLASH.h:85-103) with the host backend: spin_up, run_all, run_one_year,
    moist_surf, inf_GA against the oracle's"""
    b, s = bourne, bourne["site"]
    si, wmax_r = oracle.soil_info(b["soil"], [s["Aups_250m"], 3, 3], s["resolution"])
    Tt, rain, snowf = oracle.snow_partition(b["Ta"], b["P"], b["month"], s["latitude"], s["elev_m"])
    asp = s["asp_250m"] - 180
    sp = api.SPLASH(s["latitude"], s["elev_m"], backend="host")
    y0 = int(b["year"][0])
    g = sp.spin_up(365, y0, b["sw_in"][:365], b["Ta"][:365], rain[:365], s["slop_250m"], asp, snowf[:365], si)
    o = oracle.spin_up(s["latitude"], s["elev_m"], 365, y0, b["sw_in"][:365], b["Ta"][:365], rain[:365],
                       s["slop_250m"], asp, snowf[:365], si)
    assert g["passes"] == o["passes"]
    for k in ("sm", "snow", "qin", "tdrain", "ro", "pet"):
        close(g[k], o[k], "host spin_up." + k)
    assert np.array_equal(g["snwage"], o["snwage"])
    r = sp.run_all(b["doy"], b["year"], b["sw_in"], b["Ta"], rain, o["sm"][-1], s["slop_250m"], asp, o["snow"][-1],
                   snowf, si, o["qin"][-1], o["tdrain"][-1], o["snwage"][-1])
    q = oracle.run_all(s["latitude"], s["elev_m"], b["doy"], b["year"], b["sw_in"], b["Ta"], rain, o["sm"][-1],
                       s["slop_250m"], asp, o["snow"][-1], snowf, si, o["qin"][-1], o["tdrain"][-1], o["snwage"][-1])
    for k in ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev", "tdrain"):
        close(r[k], q[k], "host run_all." + k)
    assert np.array_equal(r["snwage"], q["snwage"])
    yr = sp.run_one_year(365, y0, b["sw_in"][:365], b["Ta"][:365], rain[:365], o["sm"], s["slop_250m"], asp, o["snow"],
                         snowf[:365], si, o["qin"], o["tdrain"], o["snwage"])
    close(yr["wn"], r["wn"][:365], "host run_one_year.wn", rtol=1e-12)
    # the two helper methods, literal forms (src/SPLASH.cpp:1921-1961, 1963-2023)
    rng = np.random.default_rng(3)
    L = oracle.lib
    for _ in range(200):
        depth = rng.uniform(0.6, 2.0)
        SAT, RES = rng.uniform(200, 600), rng.uniform(10, 60)
        wn = rng.choice([RES, SAT, rng.uniform(RES, SAT), RES - 1.0])
        lam, bub = rng.uniform(0.1, 0.6), -rng.uniform(100, 3000)
        a = api.SPLASH.moist_surf(depth, 10.0, bub, wn, SAT, RES, lam, backend="host")
        f = L.orc_moist_surf(depth, 10.0, bub, wn, SAT, RES, lam)
        assert a == pytest.approx(f, rel=1e-12)
        th_s = SAT / (depth * 1000)
        th_i = rng.choice([th_s, th_s * rng.uniform(0.1, 0.99), th_s + 0.01])
        Ks = rng.uniform(0.05, 40)
        P = rng.choice([0.0, rng.uniform(0, 5), rng.uniform(5, 300)])
        slop = rng.choice([0.0, rng.uniform(0, 60)])
        a = api.SPLASH.inf_GA(bub, th_i, Ks, th_s, lam, P, 6.0, slop, backend="host")
        f = L.orc_inf_GA(bub, th_i, Ks, th_s, lam, P, 6.0, slop)
        assert a == pytest.approx(f, rel=1e-11, abs=1e-300)


def test_splash_grid_on_the_host_with_na_cells_monthly_and_totals(api, oracle):
    """splash_grid_run with splash_opts.backend = SPLASH_BACKEND_HOST: a small synthetic terrain grid with NA cells,
    monthly outputs of all nine series, state, aridity index, pass counts and the exact monthly totals per basin --
    against the oracle's splash.grid restatement (R/splash.grid.R:277-308), each side from its own spin-up."""
    from oracle.parity import assert_whole_pipeline
    H, W, nt = 4, 24, 500
    n = H * W
    doy, year, month, mid, nm = oracle.calendar(2003, nt)
    st = oracle.synth_statics(91, H, W, 0, n, 1)
    sw, tc, pn = oracle.synth_forcing(91, H, W, 0, n, 0, doy, st["lat"], st["elev"])
    rng = np.random.default_rng(3)
    na = rng.random(n) < 0.2
    elev = np.where(na, np.nan, st["elev"])
    basin = (np.arange(n) % 3).astype(np.int32)
    dates = np.datetime64("2003-01-01") + np.arange(nt)
    for monthly in (False, True):
        g = api.splash_grid(sw, tc, pn, elev, st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"],
                            time_index=dates, sim_control=dict(monthly_out=monthly), want_aux=True, want_diag=True,
                            basin=basin, n_basin=3, backend="host")
        o, aux = oracle.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], elev, st["slop"], st["asp"], st["soil"],
                             st["au"], st["resolution"], monthly_out=monthly)
        for k in o:
            assert np.isnan(g[k][:, na]).all() and np.isnan(o[k][:, na]).all(), k
        r = assert_whole_pipeline(oracle, g, o, aux, st, doy, year, f"host grid monthly={monthly}",
                                  monthly_mid=mid if monthly else None)
        ok = r.pop("ok")
        assert np.array_equal(g["aux"]["passes2"][ok & ~na], aux["passes2"][ok & ~na])
        close(g["aux"]["AI"][ok & ~na], aux["AI"][ok & ~na], "host AI", rtol=1e-9)
        assert np.array_equal(g["aux"]["Tt"][~na], aux["Tt"][~na])
        # exact totals == sums of the returned rasters (daily run: monthly sums of the daily values)
        if monthly:
            for b_ in range(3):
                cells = (basin == b_) & ~na
                for k in ("aet", "ro", "bflow"):
                    ref = g[k][:, cells].astype(np.longdouble).sum(1).astype(float)
                    assert np.allclose(g["diag"][k][b_], ref, rtol=1e-12, atol=1e-9), (b_, k)
                assert (g["diag_cells"][b_] == cells.sum()).all()


@pytest.mark.parametrize("submit", [False, True])
def test_no_cpp_exception_leaves_the_c_abi(api, oracle, monkeypatch, submit):
    """The callers of include/splash_hip.h are C (R's .Call shim, ctypes, cgo): a C++ exception in flight there is
    terminate() without a message.  Every entry point that allocates or starts threads runs under a function-try-
    block (splash_host.hpp: exception_code); SPLASH_HIP_TEST_THROW=1 throws std::bad_alloc in the body of grid_run
    -- on the caller's thread, and on the job thread of splash_grid_submit -- and the call must come back with
    SPLASH_ENOMEM and a message, the process alive and the next call fine."""
    from rsplash_amd._lib import SplashError
    H, W, nt = 1, 4, 365
    n = H * W
    doy, year, month, mid, nm = oracle.calendar(2003, nt)
    st = oracle.synth_statics(5, H, W, 0, n, 1)
    sw, tc, pn = oracle.synth_forcing(5, H, W, 0, n, 0, doy, st["lat"], st["elev"])
    dates = np.datetime64("2003-01-01") + np.arange(nt)

    def call():
        r = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                            st["resolution"], time_index=dates, backend="host", submit=submit)
        return r.wait() if submit else r

    monkeypatch.setenv("SPLASH_HIP_TEST_THROW", "1")
    with pytest.raises(SplashError, match=r"\(-3\).*out of host memory"):
        call()
    monkeypatch.delenv("SPLASH_HIP_TEST_THROW")
    g = call()
    assert np.isfinite(g["wn"]).all()


def test_unswc_on_the_host(api, oracle, bourne):
    """R/unsSWC.R:14-88 with backend="host" (ABI 6: the kernel's own two functions, splash_unswc.hpp, on the calling
    thread): the README workflow (README.md:52-59: water to the 0.49 m sensor depth from the simulated column water)
    and a synthetic grid, against the oracle's unSWC; the same comparison the GPU suite makes
    (tests/test_gpu_parity.py::test_unswc_bourne_and_grid)."""
    b, s = bourne, bourne["site"]
    g = api.splash_point(b["sw_in"], b["Ta"], b["P"], s["latitude"], s["elev_m"], s["slop_250m"], s["asp_250m"], b["soil"],
                         s["Aups_250m"], s["resolution"], time_index=b["dates"], backend="host")
    got = api.unSWC(b["soil"], s["max_depth_sm"], g["wn"], backend="host")
    ref = oracle.unswc(b["soil"], s["max_depth_sm"], g["wn"])
    for k in ("wn_z", "wtd", "theta_i"):
        tol = 1e-5 * np.abs(ref[k]) + 1e-9 + 1e-15 * ref["cond"] * np.abs(ref[k])
        assert (np.abs(got[k] - ref[k]) <= tol).all(), k
    assert np.all(got["wn_z"] > 0) and np.all(got["wtd"] <= b["soil"][5] + 1e-12)
    H, W, nt = 6, 32, 120
    rng = np.random.default_rng(9)
    st = oracle.synth_statics(20221130, H, W, 0, H * W, 1)
    soil = st["soil"]
    sat = np.array([oracle.soil_info(soil[:, j], None, 250.0)[0][[0, 7]] for j in range(H * W)])
    wn = sat[:, 1] + (sat[:, 0] - sat[:, 1]) * rng.uniform(0, 1, (nt, H * W))
    wn[0], wn[1] = sat[:, 1], sat[:, 0]
    got = api.unSWC_cells(soil, 0.3, wn, backend="host")
    bad = 0
    for j in range(H * W):
        ref = oracle.unswc(soil[:, j], 0.3, wn[:, j])
        for k in ("wn_z", "wtd", "theta_i"):
            tol = 1e-5 * np.abs(ref[k]) + 1e-9 + 1e-14 * ref["cond"] * np.abs(ref[k])
            bad += int((np.abs(got[k][:, j] - ref[k]) > tol).sum())
    assert bad == 0
    with pytest.raises(ValueError, match="backend"):
        api.unSWC(b["soil"], 0.49, g["wn"], backend="cpu")


def _unswc_grid_case(oracle, H=6, W=32, nt=60, seed=11):
    """a synthetic grid for unSWC.grid: soils of the benchmark generator, wn anywhere between residual and saturation
    (first two layers: exactly those), some cells NA in the soil (a NaN gravel fraction alone makes a cell NA, as
    soil_hydro(fgravel = NA * 0) does) and one column of NA water"""
    rng = np.random.default_rng(seed)
    st = oracle.synth_statics(20221130, H, W, 0, H * W, 1)
    soil = st["soil"].copy()
    sat = np.array([oracle.soil_info(soil[:, j], None, 250.0)[0][[0, 7]] for j in range(H * W)])
    wn = sat[:, 1] + (sat[:, 0] - sat[:, 1]) * rng.uniform(0, 1, (nt, H * W))
    wn[0], wn[1] = sat[:, 1], sat[:, 0]
    soil[3, 5] = np.nan          # gravel NA
    soil[:, 9] = np.nan          # an ocean cell
    wn[:, 17] = np.nan
    return soil, wn


def _assert_unswc_grid(oracle, got, soil, wn, uns_depth):
    bad = 0
    for j in range(soil.shape[1]):
        ref = oracle.unswc_grid(soil[:, j], uns_depth, wn[:, j])
        for k in ("w_z", "wtd", "Se", "theta_mean"):
            g = got[k][:, j]
            assert np.array_equal(np.isnan(g), np.isnan(ref[k])), (k, j)
            with np.errstate(invalid="ignore"):
                # the integral's own cancellation factor widens the bound where the reference's value is rounding noise
                c = ref["cond"] if k in ("w_z", "Se") else 0.0
                tol = 1e-5 * np.abs(ref[k]) + 1e-9 + 1e-14 * c * np.abs(ref["w_z"] if k == "w_z" else 1.0)
                bad += int((np.abs(g - ref[k]) > tol).sum())
    assert bad == 0


@pytest.mark.parametrize("uns_depth", [0.3, 2.5])
def test_unswc_grid_semantics_on_the_host(api, oracle, uns_depth):
    """R/unsSWC.grid.R:14-141 is NOT unSWC applied cell by cell (no stoniness correction, no patch of a non-positive
    integral, the integral's water table capped at uns_depth, its own four rasters): splash_unswc_grid_run with the
    host backend against the oracle's restatement of the gridded function -- sensor depth inside the soil (0.3 m) and
    below many cells' bedrock (2.5 m: where the two functions' water-table caps part ways)."""
    soil, wn = _unswc_grid_case(oracle)
    got = api.unSWC_grid(soil, uns_depth, wn, backend="host")
    _assert_unswc_grid(oracle, got, soil, wn, uns_depth)
    assert np.isnan(got["w_z"][:, [5, 9, 17]]).all() and np.isfinite(got["w_z"][:, 0]).all()
    # and it does differ from the site function where gravel is present
    site = api.unSWC_cells(soil, uns_depth, wn, backend="host")
    j = int(np.nanargmax(soil[3]))
    assert not np.allclose(site["wn_z"][:, j], got["w_z"][:, j], rtol=1e-3)
    only = api.unSWC_grid(soil, uns_depth, wn, backend="host", want=("Se",))
    assert list(only) == ["Se"] and np.array_equal(only["Se"], got["Se"], equal_nan=True)


def test_unswc_grid_writes_the_references_four_files(api, oracle, tmp_path):
    """grid.unSWC_grid(soil_data, uns_depth, wn, d_name, outdir): the reference's signature, its four files with its
    names, variable names, units and long names (R/unsSWC.grid.R:87, 108, 115, 126), its returned list"""
    from scipy.io import netcdf_file
    from rsplash_amd import grid, io
    H, W, nt = 6, 32, 60
    soil, wn = _unswc_grid_case(oracle, H, W, nt)
    R = lambda v, z=None: grid.Raster(v, ymax=46.5, xmin=7.0, yres=0.01, xres=0.01, z=z)
    dates = np.datetime64("2001-12-15") + np.arange(nt)          # two years in the file name
    res = grid.unSWC_grid(R(soil.reshape(6, H, W)), 0.49, R(wn.reshape(nt, H, W), dates), "0.49", outdir=str(tmp_path),
                          backend="host")
    want = {"theta_mean": ("SPLASH_v2.0_theta_mean_daily_2001-2002.nc", "theta", b"m3/m3"),
            "wtd": ("SPLASH_v2.0_wtd_daily_2001-2002.nc", "wtd", b"m"),
            "w_z": ("SPLASH_v2.0_swc_top_0.49_m_daily_2001-2002.nc", "swc", b"mm"),
            "Se": ("SPLASH_v2.0_Se_top_0.49_m_daily_2001-2002.nc", "Se", b"fraction")}
    for k, (fname, var, unit) in want.items():
        path = res["paths"][k]
        assert path == str(tmp_path / fname)
        with netcdf_file(path, "r", mmap=False) as f:
            v = f.variables[var]
            assert v.units == unit and v.shape == (nt, H, W) and f.variables["time"].units == b"days since 2000-12-31"
            a = np.where(v[:] == io.NA_FLT4S, np.nan, v[:].astype(np.float64))
            with np.errstate(over="ignore"):
                assert np.allclose(a, res[k].values.astype(np.float32), equal_nan=True)
    monthly = np.arange("2003-01", "2003-07", dtype="datetime64[M]").astype("datetime64[D]")
    r2 = grid.unSWC_grid(R(soil.reshape(6, H, W)), 0.3, R(wn[:6].reshape(6, H, W), monthly), "0.3", outdir=str(tmp_path),
                         backend="host")
    assert r2["paths"]["Se"].endswith("SPLASH_v2.0_Se_top_0.3_m_monthly_2003-2003.nc")


def _gap_case(oracle, H=1, W=16, nt=450, seed=99):
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    st = oracle.synth_statics(seed, H, W, 0, H * W, 1)
    sw, tc, pn = oracle.synth_forcing(seed, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
    tc[400, 3] = np.nan     # a gap in tc: Tt is NA, rain and snowfall NA on every possible-snow day from the start
    sw[420, 5] = np.nan     # one day without radiation: that day's fluxes NA, the state after it NA for good
    pn[380, 6] = np.nan
    tc[100, 7] = np.nan     # inside the spin-up year
    return dict(doy=doy, year=year, month=month, mid=mid, nm=nm, st=st, sw=sw, tc=tc, pn=pn,
                dates=np.datetime64("2001-01-01") + np.arange(nt))


def _assert_monthly_with_gaps(g, o):
    for k in o:
        assert np.array_equal(np.isnan(g[k]), np.isnan(o[k])), f"{k}: NaN pattern of the months differs"
        close(g[k], o[k], "monthly with gaps." + k)
    gaps = [3, 5, 6]
    # the month the gap falls in has a value (the mean / sum of the days before it), the months after it NaN (means) or 0 (sums)
    assert np.isfinite(o["wn"][13, 3]) and np.isnan(o["wn"][14, 3]) and o["pet"][14, 3] == 0.0
    assert np.isfinite(o["wn"][13, 5]) and np.isnan(o["wn"][14, 5]) and o["pet"][14, 5] > 0.0   # (pet needs no state)
    assert np.isfinite(g["wn"][:12, gaps]).all()


def test_monthly_aggregation_leaves_na_days_out(api, oracle):
    """R/splash.point.R:210-211: ctapply(x, ind.mont, FUN = mean | sum, na.rm = T) -- a day without a value (a gap in
    the forcing, and every day after the state has turned NA) is left out of its month: the month of the gap keeps
    the mean / sum of its other days, an all-NA month is NaN (means: wn, snow, sm_lim) or 0 (sums).  Rounds 1-5
    summed the NaN in: every month from the gap on was NaN.  Host backend against the oracle, all nine series."""
    d = _gap_case(oracle)
    st = d["st"]
    g = api.splash_grid(d["sw"], d["tc"], d["pn"], st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                        st["resolution"], time_index=d["dates"], sim_control=dict(monthly_out=True), backend="host")
    o, aux = oracle.grid(d["doy"], d["year"], d["month"], d["mid"], d["nm"], d["sw"], d["tc"], d["pn"], st["lat"],
                         st["elev"], st["slop"], st["asp"], st["soil"], st["au"], st["resolution"], monthly_out=True)
    _assert_monthly_with_gaps(g, o)


def test_survey_named_hooks_on_the_host(api, oracle, bourne, sa_cru):
    """the test hooks SURVEY.md 8(b) names, with the host backend: splash_day_step == SPLASH::run_one_day
    (src/SPLASH.cpp:920-1588) against the oracle's, state and fluxes, on days of the Bourne series from the oracle's own
    states; splash_soil_hydro (K1 alone) against orc_soil_info on every land cell of the reference's South America
    soils; splash_hip_version"""
    from rsplash_amd import _lib
    assert _lib.lib.splash_hip_version() == _lib.lib.splash_hip_abi_version() == 6
    b, s = bourne, bourne["site"]
    si, wmax_r = oracle.soil_info(b["soil"], [s["Aups_250m"], 3, 3], s["resolution"])
    Tt, rain, snowf = oracle.snow_partition(b["Ta"], b["P"], b["month"], s["latitude"], s["elev_m"])
    asp = s["asp_250m"] - 180
    state = np.array([si[0] * 0.6, 40.0, 0.0, 0.0, 3.0])      # wn, snow, qin, td, nd
    worst = 0.0
    for t in range(0, 400, 7):
        smr, etr = oracle.run_one_day(s["latitude"], s["elev_m"], int(b["doy"][t]), int(b["year"][t]), state[0], b["sw_in"][t],
                                      b["Ta"][t], rain[t], s["slop_250m"], asp, state[1], snowf[t], si, state[2], state[3],
                                      state[4])
        g = api.day_step(s["latitude"], s["elev_m"], int(b["doy"][t]), int(b["year"][t]), b["sw_in"][t], b["Ta"][t], rain[t],
                         snowf[t], s["slop_250m"], asp, si, state, backend="host")
        pairs = dict(wn=smr["sm"], ro=smr["ro"], snow=smr["swe"], bflow=smr["bflow"], qin=smr["sqout"], tdrain=smr["tdr"],
                     snwage=smr["nd"], pet=etr["pet"], aet=etr["aet"], cond=etr["cond"])
        for k, ref in pairs.items():
            worst = max(worst, close(np.array([g[k]]), np.array([ref]), f"day_step[{t}].{k}"))
        state = np.array([smr["sm"], smr["swe"], smr["sqout"], smr["tdr"], smr["nd"]])
    print("splash_day_step vs orc_run_one_day, 58 days: max rel err", worst)
    land = ~np.isnan(sa_cru["elev"])
    soil = sa_cru["soil"][:, land].astype(np.float64)
    n = soil.shape[1]
    rng = np.random.default_rng(2)
    au = np.stack([rng.uniform(0, 1e7, n), rng.integers(1, 4, n).astype(float), np.full(n, 3.0)])
    res = rng.uniform(100, 60000, n)
    lat, elev, slop, asp_ = rng.uniform(-60, 60, n), rng.uniform(0, 4000, n), rng.uniform(0, 40, n), rng.uniform(0, 360, n)
    par = api.soil_hydro(lat, elev, slop, asp_, res, soil, au, backend="host")
    assert par.shape == (api.NPAR, n) and n > 6000
    assert np.array_equal(par[0], lat) and np.array_equal(par[3], asp_ - 180)
    ref = np.empty((13, n))
    for i in range(n):
        si_i, wmax = oracle.soil_info(soil[:, i], au[:, i], res[i])
        ref[:12, i], ref[12, i] = si_i, wmax
    for k in range(12):
        close(par[4 + k], ref[k], f"soil_info[{k}]", rtol=1e-12)
    close(par[16], ref[12], "Wmax", rtol=1e-12)


def test_host_backend_is_explicit_and_unknown_backends_are_refused(api, bourne):
    import ctypes as C
    from rsplash_amd import _lib
    b, s = bourne, bourne["site"]
    with pytest.raises(ValueError, match="backend"):
        api.splash_point(b["sw_in"], b["Ta"], b["P"], s["latitude"], s["elev_m"], 0, 0, b["soil"], 0.0, 250.0,
                         time_index=b["dates"], backend="cpu")
    # a monthly series (first two dates > 20 days apart, R/splash.point.R:56, 66) is refused, not run as if daily
    monthly = np.arange("2008-01", "2011-01", dtype="datetime64[M]").astype("datetime64[D]")
    with pytest.raises(ValueError, match="monthly inputs"):
        api.splash_point(np.ones(36), np.ones(36), np.ones(36), s["latitude"], s["elev_m"], 0, 0, b["soil"], 0.0, 250.0,
                         time_index=monthly, backend="host")
    assert _lib.lib.splash_mod_backend(7) == -1          # SPLASH_EINVAL, setting unchanged
    assert _lib.lib.splash_mod_backend(0) in (0, 1)
    assert _lib.lib.splash_host_backend_fma() in (0, 1)
    assert _lib.lib.splash_hip_abi_version() == 6


def test_cpp_host_mirror_with_the_host_backend(oracle, bourne, tmp_path):
    """include/splash.hpp: class SPLASH(lat, elv, SPLASH_BACKEND_HOST) from a plain C++ program (spin_up + run_all on
    one year of the Bourne site), no GPU -- the argument lists of src/SPLASH.h:85-103"""
    import subprocess
    from test_abi import build_cpp_mirror
    b, s = bourne, bourne["site"]
    si, wmax_r = oracle.soil_info(b["soil"], [s["Aups_250m"], 3, 3], s["resolution"])
    Tt, rain, snowf = oracle.snow_partition(b["Ta"], b["P"], b["month"], s["latitude"], s["elev_m"])
    asp, n, y = s["asp_250m"] - 180, 365, int(b["year"][0])
    head = [n, y, s["latitude"], s["elev_m"], s["slop_250m"], asp]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    np.concatenate([head, si[:12], b["sw_in"][:n], b["Ta"][:n], rain[:n], snowf[:n]]).astype("f8").tofile(fin)
    subprocess.check_call([build_cpp_mirror(str(tmp_path)), fin, fout, "host"])
    got = np.fromfile(fout).reshape(18, n)
    o = oracle.spin_up(s["latitude"], s["elev_m"], n, y, b["sw_in"][:n], b["Ta"][:n], rain[:n], s["slop_250m"], asp,
                       snowf[:n], si)
    for i, k in enumerate(("sm", "snow", "qin", "tdrain", "ro", "snwage", "pet")):
        close(got[i], o[k], "cpp host spin_up." + k)
    q = oracle.run_all(s["latitude"], s["elev_m"], np.arange(1, n + 1), np.full(n, y), b["sw_in"][:n], b["Ta"][:n],
                       rain[:n], o["sm"][-1], s["slop_250m"], asp, o["snow"][-1], snowf[:n], si, o["qin"][-1],
                       o["tdrain"][-1], o["snwage"][-1])
    for i, k in enumerate(("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev", "tdrain", "snwage")):
        close(got[7 + i], q[k], "cpp host run_all." + k)
