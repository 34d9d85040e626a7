"""GPU suite: the HIP path, called through the C ABI, against the oracle on identical inputs.

Tolerance (BASELINE.json north_star): wn, snow, ro, aet, pet within 1e-5 relative of the
reference, written as  |x - ref| <= 1e-5*|ref| + 1e-9  (1e-9 mm of water is the absolute floor for
entries that are ~0 by cancellation or clamping).  Index-like quantities (spin-up pass counts, snow
age, NaN patterns, viscosity float bit patterns) must match exactly.
"""
import os

import numpy as np
import pytest

from conftest import ROOT, rel_err
from gpu_pipeline import (RTOL, assert_one_step, assert_trajectory, assert_whole_pipeline, close, device_pipeline,  # noqa: F401
                          synth_grid as _synth)

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- SOLAR + EVAP vs the REAL reference
def test_evap_vs_reference_kat(api, kat):
    k = kat
    out = api.evap_batch(k["lat"], k["elv"], k["sw"], k["n"], k["y"], k["sw_in"], k["tc"],
                         k["slop"], k["asp"], k["snow"], k["nd"])
    ref = k["evap"]
    # viscosity (float emulation): identical bit patterns
    v = out[:, 9].astype(np.float32).view(np.uint32)
    vr = ref[:, 9].astype(np.float32).view(np.uint32)
    assert (v == vr).mean() == 1.0, f"viscosity bit mismatches: {(v != vr).sum()} of {len(v)}"
    assert np.array_equal(out[:, 9], ref[:, 9])
    for j, name in enumerate(("cond", "eet", "pet", "aet", "snowmelt", "sublimation", "econ", "pw",
                              "rn_d", "visc", "pet_max")):
        scale = np.maximum(np.abs(ref[:, j]), 1e-6 * np.abs(ref[:, j]).max())
        if name == "aet":
            # aet is a difference of pet-sized terms in the cross-over angles hn and hi
            # (src/EVAP.cpp:254-263), and for a low sun those are arc cosines near 1: the
            # reference's own doubles are 1.6e-12 mm (3e-12 of pet) away from a long-double
            # evaluation of the same statements on the supply-limited winter row that decides
            # this bound (tools/debug_kat.py prints it).  1e-9 relative + 1e-11 of pet.
            scale = scale + 1e-2 * np.abs(ref[:, 2])
        err = np.max(np.abs(out[:, j] - ref[:, j]) / scale)
        assert err < 1e-9, f"{name}: {err:.3e}"


def test_viscosity_bits_dense_sweep(api, oracle):
    """all daily temperatures the forcing can take, several pressures: float bit patterns"""
    tc = np.tile(np.linspace(-35, 48, 60000).astype(np.float32).astype(np.float64), 3)
    elv = np.repeat([0.0, 1783.08, 4500.0], 60000)
    m = len(tc)
    z = np.zeros(m)
    n = np.full(m, 100, np.int32)
    y = np.full(m, 2001, np.int32)
    args = (np.full(m, 40.0), elv, np.full(m, 0.5), n, y, np.full(m, 150.0), tc, z, z, z, z)
    out = api.evap_batch(*args)
    ref = oracle.evap_batch(*args)
    bad = out[:, 9] != ref[:, 9]
    assert bad.sum() == 0, f"{bad.sum()} viscosity mismatches of {m}"


def test_device_math_vs_libm(api):
    """device build of splash_math.hpp (flog/fexp/fpow/facos) against numpy/libm"""
    rng = np.random.default_rng(11)
    n = 400000
    x = np.r_[rng.uniform(0, 1.5, n // 2), np.exp(rng.uniform(-300, 300, n // 4)),
              1 + rng.uniform(-1e-6, 1e-6, n // 4)]
    y = rng.uniform(0.3, 12, n)
    out = api.math_probe(x, y)
    ulp = lambda a, b: np.abs(a - b) / np.spacing(np.abs(b))
    assert ulp(out[0], np.log(x)).max() <= 2.5  # table-driven log: <~ 2 ulp
    assert ulp(out[4], np.sqrt(x)).max() <= 1.5
    xe = np.r_[rng.uniform(-40, 40, n // 2), rng.uniform(-740, 700, n // 2)]
    oe = api.math_probe(xe, y)[1]
    assert ulp(oe, np.exp(xe)).max() <= 1.5
    th = rng.uniform(0, 1, n)
    op = api.math_probe(th, y)[2]
    ref = np.power(th, y)
    ok = ref > 1e-300
    assert (np.abs(op[ok] - ref[ok]) / ref[ok]).max() < 1e-13
    c = np.r_[rng.uniform(-1, 1, n // 2), 1 - rng.uniform(0, 1e-9, n // 4),
              -1 + rng.uniform(0, 1e-9, n // 4)]
    oa = api.math_probe(c, y)[3]
    assert ulp(oa, np.arccos(c)).max() <= 2.5
    # arc tangents of the static-raster kernel: every reduction range, both signs, the quadrants
    t = np.r_[rng.uniform(-3, 3, n // 2), np.exp(rng.uniform(-30, 30, n // 4)) * rng.choice([-1, 1], n // 4),
              rng.choice([0.4375, 0.6875, 1.1875, 2.4375], n // 4) * (1 + rng.uniform(-1e-12, 1e-12, n // 4))]
    yy = rng.uniform(-5, 5, n)
    yy[:100] = 0.0           # x = 0 in fatan2(y = t, x = yy): +-pi/2
    ot = api.math_probe(t, yy)
    assert ulp(ot[5], np.arctan(t)).max() <= 2.0
    assert ulp(ot[6], np.arctan2(t, yy)).max() <= 2.5
    st = api.math_probe(np.array([np.inf, -np.inf, np.nan, 0.0, -0.0]), np.ones(5))[5]
    assert st[0] == np.pi / 2 and st[1] == -np.pi / 2 and np.isnan(st[2]) and st[3] == 0 and st[4] == 0
    sp = api.math_probe(np.array([0.0, -1.0, np.inf, np.nan, 1.0]), np.ones(5))
    assert sp[0][0] == -np.inf and np.isnan(sp[0][1]) and sp[0][2] == np.inf and np.isnan(sp[0][3])
    assert sp[0][4] == 0.0 and sp[1][0] == 1.0 and sp[1][2] == np.inf and np.isnan(sp[1][3])
    assert sp[2][0] == 0.0 and np.isnan(sp[2][1])
    assert api.math_probe(np.array([-np.inf, -800.0]), np.ones(2))[1].tolist() == [0.0, 0.0]
    assert sp[4][0] == 0.0 and np.isnan(sp[4][1]) and np.isnan(sp[4][3]) and sp[4][4] == 1.0


# ---------------------------------------------------------------- module mirrors
def test_moist_surf_and_inf_GA(api, oracle):
    rng = np.random.default_rng(3)
    L = oracle.lib
    for _ in range(200):
        depth = rng.uniform(0.6, 2.0)
        SAT, RES = rng.uniform(200, 600), rng.uniform(10, 60)
        wn = rng.choice([RES, SAT, rng.uniform(RES, SAT), RES - 1.0])
        lam, bub = rng.uniform(0.1, 0.6), -rng.uniform(100, 3000)
        a = api.SPLASH.moist_surf(depth, 10.0, bub, wn, SAT, RES, lam)
        b = L.orc_moist_surf(depth, 10.0, bub, wn, SAT, RES, lam)
        assert a == pytest.approx(b, rel=1e-12)
        th_s = SAT / (depth * 1000)
        th_i = rng.choice([th_s, th_s * rng.uniform(0.1, 0.99), th_s + 0.01])
        Ks = rng.uniform(0.05, 40)
        P = rng.choice([0.0, rng.uniform(0, 5), rng.uniform(5, 300)])
        slop = rng.choice([0.0, rng.uniform(0, 60)])
        a = api.SPLASH.inf_GA(bub, th_i, Ks, th_s, lam, P, 6.0, slop)
        b = L.orc_inf_GA(bub, th_i, Ks, th_s, lam, P, 6.0, slop)
        assert a == pytest.approx(b, rel=1e-11, abs=1e-300)


def test_day_step_transitions_random_and_edges(api, oracle):
    """(state, forcing, soil_info) -> (state', 11 outputs) for single days, seeded
    random plus the edge cases the reference's control flow hinges on: bucket exactly at SAT / RES,
    flat terrain (NaN drainage time), depth exactly 2.0, tc exactly 3.0 and 0.0, zero radiation,
    polar day/night, sticky NaN / +inf drainage time carried in"""
    import ctypes as C
    rng = np.random.default_rng(2024)
    names = ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev", "tdrain")
    worst = 0.0
    ncase = 0
    skipped = dict(fragile=0, grazing=0, residual=0)
    ncmp = nloose = 0  # the bulk: comparisons beyond 1e-9 relative (+1e-12)
    for it in range(700):
        soil = [rng.uniform(15, 70), rng.uniform(8, 25), rng.uniform(0.3, 8), rng.uniform(0, 40),
                rng.choice([np.nan, rng.uniform(0.9, 1.6)]), rng.choice([2.0, rng.uniform(0.6, 2.0)])]
        slop = rng.choice([0.0, rng.uniform(0.1, 35)])
        au = [62500.0 * 2 ** rng.uniform(0, 12), 3, 3] if slop > 0 else [0.0, 3, 3]
        si, wmax = oracle.soil_info(np.array(soil), au, 250.0)
        SAT, RES = si[0], si[7]
        lat, elv = rng.choice([rng.uniform(-60, 60), 78.0, -82.0]), rng.uniform(0, 3000)
        asp = rng.uniform(0, 360) - 180
        doy, yr = int(rng.integers(1, 366)), int(rng.choice([0, 2001, 2004]))
        wn = rng.choice([SAT, RES, rng.uniform(RES, SAT), SAT + 5.0, RES - 1.0])
        tc = rng.choice([3.0, 0.0, -0.0, rng.uniform(-25, 35)])
        sw_in = rng.choice([0.0, rng.uniform(5, 420)])
        pn = rng.choice([0.0, rng.uniform(0, 8), rng.uniform(20, 150)])
        snow = rng.choice([0.0, rng.uniform(0, 300)])
        snf = rng.choice([0.0, rng.uniform(0, 30)])
        qin = rng.choice([0.0, rng.uniform(0, 5)])
        td = rng.choice([0.0, rng.uniform(0, 30), np.nan, np.inf])
        nd = float(rng.integers(0, 40))
        oracle.fragile_events()  # reset the counter
        s1, e1 = oracle.run_one_day(lat, elv, doy, yr, wn, sw_in, tc, pn, slop, asp, snow, snf, si, qin,
                                    td, nd)
        from oracle.binding import grazing_mask
        # conditioning classes (DESIGN "Conditioning"): tallied by kind and bounded below -- a change
        # that made the test skip more of its cases fails it
        if oracle.fragile_events() > 0:
            skipped["fragile"] += 1
            continue
        if grazing_mask(oracle, np.array([lat]), np.array([slop]), np.array([asp]),
                        np.array([doy], np.int32), np.array([yr], np.int32)).any():
            skipped["grazing"] += 1
            continue
        oracle.lib.orc_debug_ptr.restype = C.POINTER(C.c_double * 16)
        dbg = oracle.lib.orc_debug_ptr().contents
        # the bucket started at SAT and overflowed: what is left of the recharge after
        # `R -= ro_d` (src/SPLASH.cpp:1386-1390) is the rounding residual of (SAT + R) - SAT,
        # +-1e-13, and its SIGN switches the drainage branch of :1470 on or off -- no two
        # evaluations of R that differ in the last bit agree on it (only reachable when a day
        # starts exactly at SAT, which a preceding day with drainage never produces).  The series
        # upstream of that branch (radiation, evaporation, snow) are still compared.
        residual = 0.0 < abs(dbg[6]) < 1e-10
        if residual:
            skipped["residual"] += 1
        g = api.SPLASH(lat, elv).run_all([doy], [yr], [sw_in], [tc], [pn], wn, slop, asp, snow, [snf], si,
                                         qin, td, nd)
        ref = dict(wn=s1["sm"], ro=s1["ro"], pet=e1["pet"], aet=e1["aet"], snow=s1["swe"], cond=e1["cond"],
                   bflow=s1["bflow"], netr=e1["rn_d"] / 1e6, qin_prev=s1["sqout"], tdrain=s1["tdr"])
        ncase += 0 if residual else 1
        for k in (("pet", "aet", "snow", "cond", "netr") if residual else names):
            a, b = float(g[k][0]), float(ref[k])
            if np.isnan(b) or np.isinf(b):
                assert (np.isnan(a) and np.isnan(b)) or a == b, (it, k, a, b)
            else:
                err = max(abs(a - b) - 1e-9, 0.0) / max(abs(b), 1e-300)
                worst = max(worst, err)
                assert err <= 1e-5, (it, k, a, b, dict(wn=wn, tc=tc, sw_in=sw_in, pn=pn, slop=slop, td=td))
                ncmp += 1
                nloose += abs(a - b) > 1e-9 * abs(b) + 1e-12
        assert float(g["snwage"][0]) == s1["nd"]
    print("cases", ncase, "skipped by class", skipped, "worst rel err", worst, "beyond 1e-9:", nloose, "of", ncmp)
    # 700 seeded cases: 642 compared in full, 58 start exactly at SAT and overflow (compared upstream of
    # the drainage branch only), none meets a noise denominator or a grazing sun
    assert ncase >= 640 and skipped["residual"] <= 60 and skipped["fragile"] <= 2 and skipped["grazing"] <= 2, \
        (ncase, skipped)


def _bourne_prelude(oracle, b):
    s = b["site"]
    si, wmax_r = oracle.soil_info(b["soil"], [s["Aups_250m"], 3, 3], s["resolution"])
    Tt, rain, snowf = oracle.snow_partition(b["Ta"], b["P"], b["month"], s["latitude"], s["elev_m"])
    return si, rain, snowf


def test_module_spin_up_and_run_all_bourne(api, oracle, bourne):
    b, s = bourne, bourne["site"]
    si, rain, snowf = _bourne_prelude(oracle, b)
    asp = s["asp_250m"] - 180
    sp = api.SPLASH(s["latitude"], s["elev_m"])
    g = sp.spin_up(365, int(b["year"][0]), b["sw_in"][:365], b["Ta"][:365], rain[:365],
                   s["slop_250m"], asp, snowf[:365], si)
    o = oracle.spin_up(s["latitude"], s["elev_m"], 365, int(b["year"][0]), b["sw_in"][:365],
                       b["Ta"][:365], rain[:365], s["slop_250m"], asp, snowf[:365], si)
    assert g["passes"] == o["passes"]
    for k in ("sm", "snow", "qin", "tdrain", "ro", "pet"):
        close(g[k], o[k], "spin_up." + k)
    assert np.array_equal(g["snwage"], o["snwage"])
    r = sp.run_all(b["doy"], b["year"], b["sw_in"], b["Ta"], rain, o["sm"][-1], s["slop_250m"],
                   asp, o["snow"][-1], snowf, si, o["qin"][-1], o["tdrain"][-1], o["snwage"][-1])
    q = oracle.run_all(s["latitude"], s["elev_m"], b["doy"], b["year"], b["sw_in"], b["Ta"], rain,
                       o["sm"][-1], s["slop_250m"], asp, o["snow"][-1], snowf, si, o["qin"][-1],
                       o["tdrain"][-1], o["snwage"][-1])
    for k in ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev", "tdrain"):
        close(r[k], q[k], "run_all." + k)
    assert np.array_equal(r["snwage"], q["snwage"])


def test_module_run_one_year(api, oracle, bourne):
    """SPLASH::run_one_year (src/SPLASH.cpp:1751): year y, days 1..n, start = last vector elements"""
    b, s = bourne, bourne["site"]
    si, rain, snowf = _bourne_prelude(oracle, b)
    asp = s["asp_250m"] - 180
    n = 365
    st = dict(wn=np.full(n, 150.0), snow=np.full(n, 30.0), qin=np.full(n, 0.01),
              td=np.full(n, 2.0), nd=np.full(n, 3.0))
    g = api.SPLASH(s["latitude"], s["elev_m"]).run_one_year(
        n, 2010, b["sw_in"][:n], b["Ta"][:n], rain[:n], st["wn"], s["slop_250m"], asp, st["snow"],
        snowf[:n], si, st["qin"], st["td"], st["nd"])
    q = oracle.run_all(s["latitude"], s["elev_m"], np.arange(1, n + 1), np.full(n, 2010),
                       b["sw_in"][:n], b["Ta"][:n], rain[:n], 150.0, s["slop_250m"], asp, 30.0,
                       snowf[:n], si, 0.01, 2.0, 3.0)
    for k in ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev", "tdrain"):
        close(g[k], q[k], "run_one_year." + k)
    assert np.array_equal(g["snwage"], q["snwage"])


def test_cpp_host_mirror_bourne(oracle, bourne, tmp_path):
    """include/splash.hpp: class SPLASH from a plain C++ program (spin_up + run_all on one year)"""
    import subprocess
    from test_abi import build_cpp_mirror
    b, s = bourne, bourne["site"]
    si, rain, snowf = _bourne_prelude(oracle, b)
    asp, n, y = s["asp_250m"] - 180, 365, int(b["year"][0])
    head = [n, y, s["latitude"], s["elev_m"], s["slop_250m"], asp]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    np.concatenate([head, si[:12], b["sw_in"][:n], b["Ta"][:n], rain[:n], snowf[:n]]).astype("f8").tofile(fin)
    subprocess.check_call([build_cpp_mirror(str(tmp_path)), fin, fout])
    got = np.fromfile(fout).reshape(18, n)
    o = oracle.spin_up(s["latitude"], s["elev_m"], n, y, b["sw_in"][:n], b["Ta"][:n], rain[:n],
                       s["slop_250m"], asp, snowf[:n], si)
    for i, k in enumerate(("sm", "snow", "qin", "tdrain", "ro", "snwage", "pet")):
        close(got[i], o[k], "cpp.spin_up." + k)
    q = oracle.run_all(s["latitude"], s["elev_m"], np.arange(1, n + 1), np.full(n, y), b["sw_in"][:n],
                       b["Ta"][:n], rain[:n], o["sm"][-1], s["slop_250m"], asp, o["snow"][-1],
                       snowf[:n], si, o["qin"][-1], o["tdrain"][-1], o["snwage"][-1])
    for i, k in enumerate(("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev",
                           "tdrain", "snwage")):
        close(got[7 + i], q[k], "cpp.run_all." + k)


# ---------------------------------------------------------------- splash.point (BASELINE config 0)
@pytest.mark.parametrize("flat", [False, True])
@pytest.mark.parametrize("monthly", [False, True])
def test_splash_point_bourne(api, oracle, bourne, flat, monthly):
    b, s = bourne, bourne["site"]
    slop, asp = (0.0, 0.0) if flat else (s["slop_250m"], s["asp_250m"])
    g = api.splash_point(b["sw_in"], b["Ta"], b["P"], s["latitude"], s["elev_m"], slop, asp,
                         b["soil"], s["Aups_250m"], s["resolution"], time_index=b["dates"],
                         monthly_out=monthly)
    o, aux = oracle.point(b["doy"], b["year"], b["month"], b["month_id"], b["nmonth"], b["sw_in"],
                          b["Ta"], b["P"], s["latitude"], s["elev_m"], slop, asp, b["soil"],
                          [s["Aups_250m"], 3, 3], s["resolution"], monthly_out=monthly)
    worst = {k: close(g[k], o[k], f"point[{'flat' if flat else 'slope'}].{k}") for k in o}
    print("max rel err", {k: f"{v:.2e}" for k, v in worst.items()})
    print("max |dwn| =", np.max(np.abs(g["wn"] - o["wn"])))


def test_host_and_device_backends_agree(api, oracle, bourne):
    """ABI 6: SPLASH_BACKEND_HOST is the library's own device functions compiled for the host -- a
    second executable of the product's arithmetic.  The two backends must agree to rounding: on the bundled Bourne site
    (all nine series, daily and monthly; BASELINE.json configs[0]) and, cell for cell, on a synthetic terrain grid of
    1 024 cells (each backend from its OWN spin-up: pass counts, aridity index, snowfall threshold, every daily value
    of every cell in which neither run meets a critical event of the reference).  The host side is compared with the
    oracle on the CPU in tests/test_host_backend.py; R/splash.point.R:92-172."""
    import time
    b, s = bourne, bourne["site"]
    for monthly in (False, True):
        kw = dict(time_index=b["dates"], monthly_out=monthly)
        args = (b["sw_in"], b["Ta"], b["P"], s["latitude"], s["elev_m"], s["slop_250m"], s["asp_250m"], b["soil"],
                s["Aups_250m"], s["resolution"])
        d = api.splash_point(*args, **kw)
        t0 = time.perf_counter()
        d = api.splash_point(*args, **kw)
        t1 = time.perf_counter()
        h = api.splash_point(*args, backend="host", **kw)
        t2 = time.perf_counter()
        worst = {}
        for k in d:
            with np.errstate(invalid="ignore", divide="ignore"):
                e = np.where(np.abs(d[k]) > 1e-6, np.abs(h[k] - d[k]) / np.abs(d[k]), np.abs(h[k] - d[k]))
            worst[k] = float(np.nanmax(e))
        print(f"Bourne, monthly={monthly}: device {1e3 * (t1 - t0):.1f} ms, host {1e3 * (t2 - t1):.1f} ms; host vs device max rel "
              f"{ {k: f'{v:.1e}' for k, v in worst.items()} }")
        assert max(worst.values()) < 1e-10, worst
    H, W, nt = 16, 64, 400
    dd = _synth(oracle, H, W, nt, 1, seed=4711)
    st = dd["st"]
    args = (dd["sw"], dd["tc"], dd["pn"], st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"])
    kw = dict(time_index=dd["dates"], sim_control=dict(monthly_out=False), want_aux=True)
    g = api.splash_grid(*args, **kw)
    h = api.splash_grid(*args, backend="host", **kw)
    _, aux = oracle.grid(dd["doy"], dd["year"], dd["month"], dd["mid"], dd["nm"], dd["sw"], dd["tc"], dd["pn"], st["lat"],
                         st["elev"], st["slop"], st["asp"], st["soil"], st["au"], st["resolution"])
    calm = aux["critical"] == 0
    assert calm.mean() > 0.9
    assert np.array_equal(g["aux"]["Tt"], h["aux"]["Tt"])
    assert np.array_equal(g["aux"]["passes2"][calm], h["aux"]["passes2"][calm])
    assert np.allclose(g["aux"]["AI"][calm], h["aux"]["AI"][calm], rtol=1e-10, atol=0)
    from oracle.parity import _bad, _floor
    for k in ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "sm_lim"):
        bad = _bad(h[k], g[k], floor=_floor(k, g))[:, calm]
        assert bad.mean() <= 1e-5, (k, int(bad.sum()))


@pytest.mark.parametrize("slop,asp", [(0.0, 0.0), (12.0, 215.0)])
def test_splash_point_atneu(api, oracle, atneu, slop, asp):
    """the reference's second bundled site (FLUXNET AT-Neu, 11 years, measured bulk density)"""
    b, s = atneu, atneu["site"]
    au = 0.0 if slop == 0.0 else 40000.0
    g = api.splash_point(b["sw_in"], b["Ta"], b["P"], s["latitude"], s["elev_m"], slop, asp,
                         b["soil"], au, 250.0, time_index=b["dates"], monthly_out=False)
    o, aux = oracle.point(b["doy"], b["year"], b["month"], b["month_id"], b["nmonth"], b["sw_in"],
                          b["Ta"], b["P"], s["latitude"], s["elev_m"], slop, asp, b["soil"],
                          [au, 3, 3], 250.0)
    assert aux["critical"] == 0
    worst = {k: close(g[k], o[k], f"atneu[{slop}].{k}") for k in o}
    print("max rel err", {k: f"{v:.2e}" for k, v in worst.items()})


def test_splash_grid_sa_cru_statics(api, oracle, sa_cru):
    """real static rasters: the reference's bundled 0.5 degree South America grid (139 x 159 cells,
    72 % ocean = NA, soils with up to 45 % organic matter) under the synthetic daily forcing"""
    H, W, nt = sa_cru["nrows"], sa_cru["ncols"], 365
    n = H * W
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    lat, elev, soil = sa_cru["lat"], sa_cru["elev"], sa_cru["soil"]
    sw, tc, pn = oracle.synth_forcing(20221130, H, W, 0, n, 0, doy, lat, np.nan_to_num(elev))
    sea = np.isnan(elev)
    for a in (sw, tc, pn):
        a[:, sea] = np.nan
    res = np.full(n, 55000.0) * np.sqrt(np.cos(np.radians(lat)))  # ~sqrt(cell area) in m
    flat = np.zeros(n)
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    g = api.splash_grid(sw, tc, pn, elev, soil, lat, flat, flat, None, res, time_index=dates,
                        sim_control=dict(monthly_out=True), want_aux=True)
    o, aux = oracle.grid(doy, year, month, mid, nm, sw, tc, pn, lat, elev, flat, flat, soil, None,
                         res, monthly_out=True)
    land = ~np.isnan(o["wn"][0])
    assert 6000 < land.sum() < 6200
    for k in o:
        assert np.isnan(g[k][:, ~land]).all() and np.isfinite(g[k][:, land]).all(), k
    r = assert_whole_pipeline(oracle, g, o, aux, dict(lat=lat, slop=flat, asp=flat), doy, year, "sa_cru",
                              monthly_mid=mid)
    ok = r.pop("ok") & land
    assert np.array_equal(g["aux"]["passes2"][ok], aux["passes2"][ok])
    print(r)


# ---------------------------------------------------------------- splash.grid
@pytest.mark.parametrize("terrain,monthly", [(0, False), (1, False), (1, True)])
def test_splash_grid_synthetic(api, oracle, terrain, monthly):
    H, W, nt = 12, 32, 730
    d = _synth(oracle, H, W, nt, terrain)
    st = d["st"]
    g = api.splash_grid(d["sw"], d["tc"], d["pn"], st["elev"], st["soil"], st["lat"], st["slop"],
                        st["asp"], st["au"], st["resolution"], time_index=d["dates"],
                        sim_control=dict(monthly_out=monthly), want_aux=True, want_diag=True)
    o, aux = oracle.grid(d["doy"], d["year"], d["month"], d["mid"], d["nm"], d["sw"], d["tc"],
                         d["pn"], st["lat"], st["elev"], st["slop"], st["asp"], st["soil"],
                         st["au"], st["resolution"], monthly_out=monthly)
    assert np.array_equal(g["aux"]["Tt"], aux["Tt"])
    # strict for every cell without a critical event of the reference (DESIGN.md "Conditioning");
    # of the cells with one, at most a few per cent may differ
    r = assert_whole_pipeline(oracle, g, o, aux, st, d["doy"], d["year"], f"grid[{terrain},{monthly}]",
                              monthly_mid=d["mid"] if monthly else None)
    ok = r.pop("ok")
    print(r)
    assert np.array_equal(g["aux"]["passes1"][ok], aux["passes1"][ok])
    assert np.array_equal(g["aux"]["passes2"][ok], aux["passes2"][ok])
    close(g["aux"]["AI"][ok], aux["AI"][ok], "AI", rtol=1e-10)
    close(g["state_end"][[0, 1, 2]][:, ok], aux["state1"].T[[0, 1, 2]][:, ok], "state_end")
    assert np.array_equal(g["state_end"][4][ok], aux["state1"][:, 4][ok])
    for k in o:
        assert np.isfinite(g[k]).all(), k
    # diagnostics: monthly grid totals == sums over the oracle's cells
    if not monthly and ok.all():
        for k in ("aet", "ro", "bflow"):
            ref = np.array([o[k][d["mid"] == m].sum() for m in range(d["nm"])])
            assert np.allclose(g["diag"][k], ref, rtol=1e-9), k
        last = np.array([np.where(d["mid"] == m)[0][-1] for m in range(d["nm"])])
        assert np.allclose(g["diag"]["wn"], o["wn"][last].sum(1), rtol=1e-10)
        assert np.allclose(g["diag"]["snow"], o["snow"][last].sum(1), rtol=1e-9, atol=1e-9)
        assert (g["diag_cells"] == H * W).all()
    if not monthly:
        # the totals are exact sums of the GPU's own cell values (fixed point, order free): equal
        # to a long-double sum of the returned rasters to the last bits of a double
        mids = d["mid"]
        for k in ("aet", "ro", "bflow"):
            ref = np.array([g[k][mids == m].astype(np.longdouble).sum() for m in range(d["nm"])])
            assert np.allclose(g["diag"][k], ref.astype(np.float64), rtol=1e-13, atol=1e-10), k


def test_splash_grid_writes_reference_named_bricks(api, oracle, tmp_path):
    """sim.control$inmem = FALSE (R/splash.grid.R:186-216): [nt, H, W] rasters in, NetCDF bricks out"""
    from scipy.io import netcdf_file
    H, W, nt = 6, 16, 730
    d = _synth(oracle, H, W, nt, 1)
    st = d["st"]
    r3 = lambda a: a.reshape(nt, H, W)
    g = api.splash_grid(r3(d["sw"]), r3(d["tc"]), r3(d["pn"]), st["elev"], st["soil"], st["lat"],
                        st["slop"], st["asp"], st["au"], st["resolution"], time_index=d["dates"],
                        sim_control=dict(monthly_out=True, inmem=False), outdir=str(tmp_path))
    assert len(g["files"]) == 9 and g["wn"].shape == (24, H, W)
    assert os.path.basename(g["files"]["wn"]) == "SPLASH_v2.0_sm_monthly_2001-2002.nc"
    with netcdf_file(g["files"]["aet"], "r", mmap=False) as f:
        a = f.variables["aet"][:].astype(np.float64)
        assert a.shape == (24, H, W) and np.allclose(a, g["aet"], rtol=1e-6, atol=1e-30)
        assert np.allclose(f.variables["lat"][:], st["lat"].reshape(H, W)[:, 0])


def test_splash_grid_end_to_end_from_a_dem_in_row_blocks(api, oracle, tmp_path):
    """splash.grid with the reference's signature (rsplash_amd/grid.py, the twin of r/splash_hip.R): the
    static rasters derived from the elevation raster alone (R/splash.grid.R:76-110), the grid cut into
    row blocks, three of them in the library at a time (the receive loop of :359-400), the nine bricks
    written block by block (:370-392) -- against ONE call over the whole grid with the same statics:
    every series bit for bit, the files the float32 of it, the water-balance limbs equal (integer sums
    over blocks).  NA cells in the DEM and the forcing; forcing read from a NetCDF file through a memory
    map (big-endian float32, as raster's FLT4S files are)."""
    from scipy.io import netcdf_file
    from rsplash_amd import grid, io
    H, W, nt = 23, 40, 730
    d = _synth(oracle, H, W, nt, 1)
    st = d["st"]
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:H, 0:W]
    dem = 900 + 400 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + 6.0 * yy + rng.uniform(0, 15, (H, W))
    na = rng.random((H, W)) < 0.08
    dem[na] = np.nan
    geo = dict(ymax=46.5, xmin=7.0, yres=0.01, xres=0.01)
    r3 = lambda a: a.reshape(nt, H, W).copy()
    sw, tc, pn = (r3(d[k]).astype(np.float32) for k in ("sw", "tc", "pn"))
    for a in (sw, tc, pn):
        a[:, na] = np.nan
    # the forcing goes through a NetCDF file and comes back as memory-mapped big-endian variables
    fpath = str(tmp_path / "forcing.nc")
    lay = io.nc3_create(fpath, {"time": nt, "lat": H, "lon": W},
                        [(k, io._NC_FLOAT, ("time", "lat", "lon"), {}) for k in ("sw", "tc", "pn")])
    for k, a in (("sw", sw), ("tc", tc), ("pn", pn)):
        o, shp, dt = lay[k]
        m = np.memmap(fpath, dtype=dt, mode="r+", offset=o, shape=shp)
        m[:] = a
        m.flush()
        del m
    f = netcdf_file(fpath, "r", mmap=True)
    R = lambda v, z=None: grid.Raster(v, z=z, **geo)
    elev = R(dem)
    soil = R(st["soil"].reshape(6, H, W))
    bricks = [R(f.variables[k], d["dates"]) for k in ("sw", "tc", "pn")]
    basin = (xx // 14 + 3 * (yy // 12)).astype(np.int32)
    statics = grid.grid_statics(elev)
    assert statics["method"] == "top" and np.isnan(statics["lat"][na]).all()
    assert np.isfinite(statics["slop"][~na]).all()
    # cells whose 3 x 3 window holds no flow code (an island inside NA cells) have no ncellin/ncellout
    # (R/upslope_area.R:140-165): NA statics make an NA cell, here as in the reference
    na = na | np.isnan(statics["Au"]).any(0)
    assert (statics["Au"][1][~na] >= 1).all() and (~na).sum() > 0.7 * H * W   # (the outer ring has no flow direction either)
    kw = dict(sim_control=dict(monthly_out=True, inmem=True), statics=statics, want_diag=True, basin=basin)
    seen = []
    g = grid.splash_grid(*bricks, elev, soil, block_rows=4, inflight=3, progress=lambda i, n: seen.append((i, n)), **kw)
    assert g["blocks"] == [(r, min(4, H - r)) for r in range(0, H, 4)] and seen[-1] == (6, 6)
    one = api.splash_grid(sw, tc, pn, dem, soil.values, statics["lat"], statics["slop"], statics["asp"],
                          statics["Au"], statics["resolution"], time_index=d["dates"],
                          sim_control=dict(monthly_out=True), want_diag=True, basin=basin, n_basin=int(basin.max()) + 1)
    ret = dict(wn="wn", ro="ro", pet="pet", aet="aet", snow="snow", bflow="bflow", cond="cond", netr="netrad", sm_lim="sm_lim")
    for k, rk in ret.items():
        assert g[rk].shape == (24, H, W)
        assert np.array_equal(g[rk].view(np.int64), one[k].view(np.int64)), k
        assert np.isnan(g[rk][:, na]).all() and np.isfinite(g[rk][:, ~na]).all(), k
    assert np.array_equal(g["diag_acc"], one["diag_acc"])
    assert np.array_equal(g["diag"]["aet"], one["diag"]["aet"])
    # the same run to disk, daily, two blocks in flight, default block size (two blocks)
    gd = grid.splash_grid(*bricks, elev, soil, outdir=str(tmp_path / "out"), tmpdir=str(tmp_path),
                          sim_control=dict(monthly_out=False, inmem=False), statics=statics)
    assert len(gd["blocks"]) == 2
    oned = api.splash_grid(sw, tc, pn, dem, soil.values, statics["lat"], statics["slop"], statics["asp"],
                           statics["Au"], statics["resolution"], time_index=d["dates"],
                           sim_control=dict(monthly_out=False))
    assert os.path.basename(gd["wn"]) == "SPLASH_v2.0_sm_daily_2001-2002.nc"
    assert os.path.basename(gd["netrad"]) == "SPLASH_v2.0_netrad_daily_2001-2002.nc"
    for k, rk in ret.items():
        with netcdf_file(gd[rk], "r", mmap=False) as fo:
            v = fo.variables[io.BRICKS[k][1]]
            a = np.array(v[:])
            assert a.shape == (nt, H, W) and a.dtype.itemsize == 4
            want = np.where(np.isnan(oned[k]), io.NA_FLT4S, oned[k]).astype(np.float32)
            assert np.array_equal(a, want), k
            assert np.allclose(fo.variables["lat"][:], elev.lat()) and np.allclose(fo.variables["lon"][:], elev.lon())
    del bricks, R
    try:
        f.close()
    except Exception:
        pass


def test_time_chunks_continue_a_run_bit_for_bit(api, oracle, tmp_path):
    """ABI 5: a series processed in time chunks.  The module's run_all starts from the state it is given
    (wn_last ... nds_last, src/SPLASH.h:93-97); splash_grid_in.state0 / ai0 / tt expose that at the block
    level: chunk 1 with the spin-up and the threshold of the WHOLE series (the maximum of the chunks'
    thresholds, R/splash.point.R:120-122), every later chunk from the state the one before ended in ->
    every series, the final state and the water-balance limbs of the undivided call, bit for bit; and
    rsplash_amd.grid.splash_grid(chunk_years=1) does the same through row blocks and brick files."""
    from scipy.io import netcdf_file
    from rsplash_amd import grid, io
    H, W, nt = 6, 20, 1096                       # 2001-2003 and 1 January 2004: the last chunk is ONE day
    d = _synth(oracle, H, W, nt, 1, seed=31337)
    st = d["st"]
    n = H * W
    sw, tc, pn = d["sw"].copy(), d["tc"].copy(), d["pn"].copy()
    elev = st["elev"].copy()
    na = np.zeros(n, bool)
    na[[7, 63]] = True
    for a in (sw, tc, pn):
        a[:, na] = np.nan
    elev[na] = np.nan
    stat = (elev, st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"])
    for monthly in (False, True):
        kw = dict(sim_control=dict(monthly_out=monthly), want_diag=True, want_aux=True)
        one = api.splash_grid(sw, tc, pn, *stat, time_index=d["dates"], **kw)
        cuts = [(0, 365), (365, 730), (730, 1095), (1095, 1096)]
        tts = [api.snow_threshold(tc[a:b], st["lat"], elev) for a, b in cuts]
        tt = tts[0]
        for t in tts[1:]:
            tt = np.maximum(tt, t)
        ok = ~na
        assert np.array_equal(tt[ok], one["aux"]["Tt"][ok])      # the whole series' threshold, from its pieces
        assert not np.array_equal(tts[1][ok], tt[ok])            # (and no single chunk has it)
        state = ai = None
        parts, accs = [], []
        for a, b in cuts:
            r = api.splash_grid(sw[a:b], tc[a:b], pn[a:b], *stat, time_index=d["dates"][a:b], tt=tt,
                                state0=state, ai0=ai, **kw)
            state = r["state_end"]
            ai = r["aux"]["AI"] if ai is None else ai
            parts.append(r)
            accs.append(r["diag_acc"])
        for k in api.OUT_NAMES:
            got = np.concatenate([p[k] for p in parts])
            assert got.shape == one[k].shape, k
            assert np.array_equal(got[:, ok].view(np.int64), one[k][:, ok].view(np.int64)), (monthly, k)
            assert np.isnan(got[:, na]).all(), k
        assert np.array_equal(state[:, ok].view(np.int64), one["state_end"][:, ok].view(np.int64))
        assert np.array_equal(np.concatenate(accs, axis=1), one["diag_acc"])
        assert (parts[1]["aux"]["passes2"][ok] == 0).all() and np.array_equal(parts[2]["aux"]["AI"][ok], ai[ok])
    # refusals: the three come together
    with pytest.raises(api.SplashError, match="state0, ai0 and tt"):
        api.splash_grid(sw[:10], tc[:10], pn[:10], *stat, time_index=d["dates"][:10], state0=state)
    with pytest.raises(api.SplashError, match="365"):
        api.splash_grid(sw[:10], tc[:10], pn[:10], *stat, time_index=d["dates"][:10], tt=tt)
    # the block loop with time chunks: three whole years, rows in blocks of 4, bricks on disk
    nt3 = 1095
    geo = dict(ymax=46.5, xmin=7.0, yres=0.01, xres=0.01)
    R = lambda v, z=None: grid.Raster(v, z=z, **geo)
    r3 = lambda a: a[:nt3].reshape(nt3, H, W)
    statics = dict(lat=st["lat"].reshape(H, W), slop=st["slop"].reshape(H, W), asp=st["asp"].reshape(H, W),
                   Au=st["au"].reshape(3, H, W), resolution=st["resolution"].reshape(H, W))
    bricks = [R(r3(x), d["dates"][:nt3]) for x in (sw, tc, pn)]
    ref = api.splash_grid(r3(sw), r3(tc), r3(pn), elev.reshape(H, W), st["soil"].reshape(6, H, W), statics["lat"],
                          statics["slop"], statics["asp"], statics["Au"], statics["resolution"],
                          time_index=d["dates"][:nt3], sim_control=dict(monthly_out=True), want_diag=True)
    g = grid.splash_grid(*bricks, R(elev.reshape(H, W)), R(st["soil"].reshape(6, H, W)), outdir=str(tmp_path),
                         sim_control=dict(monthly_out=True, inmem=False), statics=statics, block_rows=4,
                         chunk_years=1, want_diag=True)
    assert grid.year_chunks(d["dates"][:nt3], 1) == [(0, 365), (365, 730), (730, 1095)]
    assert grid.year_chunks(d["dates"], 2) == [(0, 730), (730, 1096)]
    assert np.array_equal(g["diag_acc"], ref["diag_acc"])
    for k, rk in dict(wn="wn", aet="aet", snow="snow", netr="netrad").items():
        with netcdf_file(g[rk], "r", mmap=False) as fo:
            a = np.array(fo.variables[io.BRICKS[k][1]][:])
            want = np.where(np.isnan(ref[k]), io.NA_FLT4S, ref[k]).astype(np.float32)
            assert a.shape == (36, H, W) and np.array_equal(a, want), k
    gm = grid.splash_grid(*bricks, R(elev.reshape(H, W)), R(st["soil"].reshape(6, H, W)),
                          sim_control=dict(monthly_out=False, inmem=True), statics=statics, block_rows=3, chunk_years=2)
    refd = api.splash_grid(r3(sw), r3(tc), r3(pn), elev.reshape(H, W), st["soil"].reshape(6, H, W), statics["lat"],
                           statics["slop"], statics["asp"], statics["Au"], statics["resolution"],
                           time_index=d["dates"][:nt3], sim_control=dict(monthly_out=False))
    okg = ~na.reshape(H, W)
    for k, rk in dict(wn="wn", ro="ro", pet="pet", bflow="bflow", sm_lim="sm_lim").items():
        assert np.array_equal(gm[rk][:, okg].view(np.int64), refd[k][:, okg].view(np.int64)), k


def test_splash_grid_na_cells_and_blocking(api, oracle):
    """NA cells (72 % of the reference's SA_cru example) come back NaN; results do not depend on
    how the grid is cut into launches"""
    H, W, nt = 8, 40, 400
    d = _synth(oracle, H, W, nt, 1)
    st = d["st"]
    rng = np.random.default_rng(5)
    mask = rng.random(H * W) < 0.6
    mask[:3] = True
    sw, tc, pn = d["sw"].copy(), d["tc"].copy(), d["pn"].copy()
    elev = st["elev"].copy()
    sw[:, mask] = np.nan
    tc[:, mask] = np.nan
    pn[:, mask] = np.nan
    elev[mask] = np.nan
    kw = dict(time_index=d["dates"], sim_control=dict(monthly_out=False))
    g = api.splash_grid(sw, tc, pn, elev, st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                        st["resolution"], **kw)
    for k in ("wn", "ro", "pet", "aet", "snow"):
        assert np.isnan(g[k][:, mask]).all() and np.isfinite(g[k][:, ~mask]).all()
    full = api.splash_grid(d["sw"], d["tc"], d["pn"], st["elev"], st["soil"], st["lat"],
                           st["slop"], st["asp"], st["au"], st["resolution"], **kw)
    blk = api.splash_grid(d["sw"], d["tc"], d["pn"], st["elev"], st["soil"], st["lat"], st["slop"],
                          st["asp"], st["au"], st["resolution"], max_cells_per_launch=100, **kw)
    for k in ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "sm_lim"):
        assert np.array_equal(full[k][:, ~mask], g[k][:, ~mask]), k
        assert np.array_equal(full[k], blk[k]), k


# ---------------------------------------------------------------- device-resident stages
def test_device_pipeline_chunking_and_generator(api, oracle):
    """splash_dev_*: (i) the device generator reproduces the oracle's generator, (ii) a run cut
    into time chunks (state carried in HBM) equals the single run bit for bit, (iii) monthly
    output of a chunked run equals the aggregation of the daily run."""
    import torch
    dev = torch.device("cuda:0")
    H, W, nt, seed = 16, 64, 800, 20221130
    n = H * W
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    cal = api.Calendar(doy, year, month)
    f64 = dict(dtype=torch.float64, device=dev)
    lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
    soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
    api.dev_synth_statics(seed, H, W, 0, n, 1, lat, elev, slop, asp, res, soil, au)
    sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
    api.dev_synth_forcing(seed, cal, 0, nt, 0, n, lat, elev, sw, tc, pn, n)
    torch.cuda.synchronize()
    st = oracle.synth_statics(seed, H, W, 0, n, 1)
    osw, otc, opn = oracle.synth_forcing(seed, H, W, 0, n, 0, doy, st["lat"], st["elev"])
    for k, t in (("lat", lat), ("elev", elev), ("slop", slop), ("asp", asp)):
        assert np.allclose(t.cpu().numpy(), st[k], rtol=1e-15, atol=0), k
    assert np.allclose(soil.cpu().numpy(), st["soil"], rtol=1e-15)
    assert np.allclose(au.cpu().numpy(), st["au"], rtol=1e-14)
    for a, b in ((sw, osw), (tc, otc), (pn, opn)):
        a = a.cpu().numpy()
        ulp = np.abs(a - b) / np.maximum(np.abs(np.spacing(b.astype(np.float32))), 1e-45)
        assert (a == b).mean() > 0.9999 and ulp.max() <= 1.0

    par = torch.empty(api.NPAR, n, **f64)
    Tt = torch.empty(n, **f64)
    state = torch.empty(api.NSTATE, n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
    api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
    work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
    passes = torch.zeros(2, n, dtype=torch.int32, device=dev)
    par_cellout = par[15].clone()  # spin_up overwrites ncellout with the aridity index
    api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work, passes=passes)
    torch.cuda.synchronize()
    assert int(passes.min()) >= 2 and int(passes.max()) <= 1000
    state0 = state.clone()
    # without `passes` spin-up #1 stops after the pass that yields the aridity index (its state is
    # discarded by the reference as well): the same bits must come out
    par2, state2 = par.clone(), torch.empty_like(state)
    par2[15] = par_cellout
    api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par2, n, state2, n, work)
    torch.cuda.synchronize()
    assert torch.equal(par2.view(torch.int64), par.view(torch.int64))
    assert torch.equal(state2.view(torch.int64), state0.view(torch.int64))
    names = ("wn", "snow", "ro", "aet", "pet")
    one = {k: torch.empty(nt, n, **f64) for k in names}
    api.dev_run_all(cal, 0, nt, n, sw, tc, pn, n, Tt, par, n, state, n, one, n)
    end1 = state.clone()
    # (ii) three uneven chunks, second one on a sub-range of cells pointers (ld > ncell)
    state.copy_(state0)
    chk = {k: torch.empty(nt, n, **f64) for k in names}
    for t0, t1 in ((0, 45), (45, 400), (400, nt)):
        outs = {k: chk[k][t0:] for k in names}
        api.dev_run_all(cal, t0, t1 - t0, n, sw[t0:], tc[t0:], pn[t0:], n, Tt, par, n, state, n,
                        outs, n)
    torch.cuda.synchronize()
    for k in names:
        assert torch.equal(one[k], chk[k]), k
    assert torch.equal(end1[:3], state[:3]) and torch.equal(end1[4], state[4])
    # (iii) monthly, chunked at non-month boundaries
    state.copy_(state0)
    mon = {k: torch.zeros(nm, n, **f64) for k in names}
    for t0, t1 in ((0, 45), (45, 400), (400, nt)):
        m0 = int(mid[t0])
        outs = {k: mon[k][m0:] for k in names}
        api.dev_run_all(cal, t0, t1 - t0, n, sw[t0:], tc[t0:], pn[t0:], n, Tt, par, n, state, n,
                        outs, n, monthly=True)
    torch.cuda.synchronize()
    midt = torch.as_tensor(mid.astype(np.int64), device=dev)
    for k in names:
        agg = torch.zeros(nm, n, **f64).index_add_(0, midt, one[k])
        if k in ("wn", "snow"):
            cnt = torch.bincount(midt, minlength=nm).to(torch.float64)
            agg = agg / cnt[:, None]
        assert torch.allclose(mon[k], agg, rtol=1e-12, atol=1e-12), k
    # (iv) NA days (R/splash.point.R:210-211, na.rm = T) in a month that two calls share (cut at t = 45: February has
    # days 31..58): a NA day in the SECOND part (t = 50, cells 0..9) -- the month's means are over the days that have a
    # value, as in one call; a NA day in the FIRST part (t = 40, cells 10..19) cannot be carried in the one stored
    # number: those means are NaN (the sums of the fluxes before the gap still add up, and the months before too)
    pn2 = pn.clone()
    pn2[50, 0:10] = float("nan")
    pn2[40, 10:20] = float("nan")
    res = {}
    for cuts in (((0, nt),), ((0, 45), (45, nt))):
        state.copy_(state0)
        mon2 = {k: torch.zeros(nm, n, **f64) for k in names}
        for t0, t1 in cuts:
            m0 = int(mid[t0])
            api.dev_run_all(cal, t0, t1 - t0, n, sw[t0:], tc[t0:], pn2[t0:], n, Tt, par, n, state, n,
                            {k: mon2[k][m0:] for k in names}, n, monthly=True)
        torch.cuda.synchronize()
        res[len(cuts)] = {k: v.cpu().numpy() for k, v in mon2.items()}
    a, b = res[1], res[2]
    for k in names:
        assert np.isfinite(a[k][1, :20]).all(), k                                  # one call: February has a value everywhere
        assert np.allclose(a[k][:, 20:], b[k][:, 20:], rtol=1e-12, atol=1e-12), k  # cells without a gap: as before
        assert np.allclose(a[k][:, :10], b[k][:, :10], rtol=1e-12, atol=1e-12, equal_nan=True), k   # gap in the second part
        assert np.allclose(a[k][0, 10:20], b[k][0, 10:20], rtol=1e-12), k          # January of the others
        if k in ("wn", "snow"):
            assert np.isnan(b[k][1, 10:20]).all(), k                               # gap in the first part: the mean is given up
        else:
            assert np.allclose(a[k][1, 10:20], b[k][1, 10:20], rtol=1e-12, atol=1e-12), k   # ... the sums are not
    # row-block partition == whole grid (what each rank of a multi-GPU run computes)
    state.copy_(state0)
    half = n // 2
    part = {k: torch.empty(nt, n, **f64) for k in names}
    for c0, c1 in ((0, half), (half, n)):
        outs = {k: part[k][:, c0:] for k in names}
        api.dev_run_all(cal, 0, nt, c1 - c0, sw[:, c0:], tc[:, c0:], pn[:, c0:], n, Tt[c0:],
                        par[:, c0:], n, state[:, c0:], n, outs, n)
    torch.cuda.synchronize()
    for k in names:
        assert torch.equal(one[k], part[k]), k
    cal.close()


CASES = {
    # name: (seed, Htot, W, cell0, cells, days, terrain, start)
    "terrain": (777000, 64, 128, 0, 8192, 730, 1, "01-01"),
    # first rows of the 4096-row benchmark grid: lat ~60 N, slopes up to 30 deg, winter sun grazing
    # the slopes (0 < hs < 2e-3 deg: the reference's daily radiation is rounding noise there)
    "benchmark-rows-60N": (20221130, 4096, 4096, 0, 4 * 4096, 365, 1, "01-01"),
    "flat": (816595, 32, 128, 0, 4096, 400, 0, "01-01"),
    # a series that starts in October: the spin-up still runs day-of-year 1..365 of the first year
    "starts-in-october": (784919, 32, 128, 0, 4096, 500, 1, "10-01"),
}


@pytest.mark.parametrize("case", list(CASES))
def test_every_cell_day_one_step_and_trajectory(api, oracle, case):
    """The device pipeline (soil_hydro, snow threshold, both spin-ups, run_all with all 11 daily
    series) against the oracle on EVERY cell-day, two ways (tests/gpu_pipeline.py): one-step (the
    oracle re-does each day from the GPU's previous-day state: nothing can hide behind an earlier
    difference) and trajectory (the oracle carries its own state like SPLASH::run_all).  Set aside
    are only cell-days on which the reference disagrees with itself (profiles/r02_conditioning.md)."""
    seed, Htot, W, cell0, n, nt, terrain, start = CASES[case]
    p = device_pipeline(api, seed, Htot, W, cell0, n, nt, terrain, start=start)
    r1 = assert_one_step(oracle, p, case)
    print("one-step", r1)
    r2 = assert_trajectory(oracle, p, case)
    print("trajectory", r2)
    if case == "benchmark-rows-60N":
        assert r1["grazing"] > 0  # the class exists on this grid
    # snow age is index-like: identical wherever the snow pack is
    assert np.isfinite(p["out"]["wn"]).all()


def test_spin_up_runs_on_day_of_year_1_to_365_of_the_first_year(api, oracle):
    """src/SPLASH.cpp:1658/1674 + R/splash.point.R:141-152: spin_up(365, y[1], ...) evaluates
    quick_run(i+1, y, ...) -- orbit of day-of-year 1..365 of the first year -- on the first 365
    RECORDS, whatever their dates (while the rain/snow split keeps the records' real months).
    Whole pipeline against the oracle's splash.grid restatement for a series starting 2001-10-01."""
    H, W, nt = 8, 64, 500
    n = H * W
    dates = np.datetime64("2001-10-01") + np.arange(nt)
    doy, year, month = api.time_fields(dates)
    mid = api.month_ids(year, month)
    nm = int(mid[-1]) + 1
    st = oracle.synth_statics(4242, H, W, 0, n, 1)
    sw, tc, pn = oracle.synth_forcing(4242, H, W, 0, n, 273, doy, st["lat"], st["elev"])
    g = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                        st["resolution"], time_index=dates, sim_control=dict(monthly_out=False),
                        want_aux=True)
    o, aux = oracle.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"],
                         st["asp"], st["soil"], st["au"], st["resolution"])
    r = assert_whole_pipeline(oracle, g, o, aux, st, doy, year, "october")
    ok = r.pop("ok")
    print(r)
    close(g["aux"]["AI"][ok], aux["AI"][ok], "AI", rtol=1e-10)
    assert np.array_equal(g["aux"]["passes2"][ok], aux["passes2"][ok])
    # and it matters: spinning up on the records' own day numbers gives another aridity index
    st0 = np.stack([g["state_end"][k] for k in range(3)])
    assert np.isfinite(st0).all()


def test_gaps_in_the_forcing_propagate_like_NA_in_R(api, oracle):
    """NaN later in a series is not a cell mask (R/splash.point.R:120-128): one missing tc makes
    Tt = max(tc[p_snow >= 0.5]) NA, hence NA rain and snowfall on every day with p_snow >= 0.5 and on
    the gap day itself; a missing sw_in or pn poisons the day it falls on.  Same NaN pattern and
    same numbers as the oracle."""
    H, W, nt = 2, 32, 450
    n = H * W
    d = _synth(oracle, H, W, nt, 1, seed=99)
    st = d["st"]
    sw, tc, pn = d["sw"].copy(), d["tc"].copy(), d["pn"].copy()
    tc[400, 3] = np.nan   # after the spin-up year
    tc[100, 7] = np.nan   # inside it
    sw[420, 11] = np.nan
    pn[380, 13] = np.nan
    g = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                        st["resolution"], time_index=d["dates"], sim_control=dict(monthly_out=False),
                        want_aux=True)
    o, aux = oracle.grid(d["doy"], d["year"], d["month"], d["mid"], d["nm"], sw, tc, pn, st["lat"],
                         st["elev"], st["slop"], st["asp"], st["soil"], st["au"], st["resolution"])
    assert np.isnan(aux["Tt"][[3, 7]]).all() and np.isnan(g["aux"]["Tt"][[3, 7]]).all()
    assert np.array_equal(np.isnan(g["aux"]["Tt"]), np.isnan(aux["Tt"]))
    gaps = [3, 7, 11, 13]
    for k in ("wn", "snow", "ro", "aet", "pet"):
        assert np.array_equal(np.isnan(g[k][:, gaps]), np.isnan(o[k][:, gaps])), k
        assert np.isnan(o[k][:, gaps]).any()
        close(g[k][:, gaps], o[k][:, gaps], "gaps." + k)
    r = assert_whole_pipeline(oracle, g, o, aux, st, d["doy"], d["year"], "gaps")
    r.pop("ok")
    print(r)


def test_monthly_aggregation_leaves_na_days_out(api, oracle):
    """R/splash.point.R:210-211, na.rm = T (tests/test_host_backend.py has the host backend's twin of this test): the
    month a gap falls in keeps the mean / sum of its other days, all-NA months are NaN (means) or 0 (sums) -- the
    device's monthly outputs against the oracle's and, for the NaN pattern, the host backend's."""
    from test_host_backend import _assert_monthly_with_gaps, _gap_case
    d = _gap_case(oracle)
    st = d["st"]
    g = api.splash_grid(d["sw"], d["tc"], d["pn"], st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                        st["resolution"], time_index=d["dates"], sim_control=dict(monthly_out=True))
    o, aux = oracle.grid(d["doy"], d["year"], d["month"], d["mid"], d["nm"], d["sw"], d["tc"], d["pn"], st["lat"],
                         st["elev"], st["slop"], st["asp"], st["soil"], st["au"], st["resolution"], monthly_out=True)
    _assert_monthly_with_gaps(g, o)
    host = api.splash_grid(d["sw"], d["tc"], d["pn"], st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                           st["resolution"], time_index=d["dates"], sim_control=dict(monthly_out=True), backend="host")
    for k in o:
        assert np.array_equal(np.isnan(g[k]), np.isnan(host[k])), k


def test_soil_hydro_cellpar_direct(api, oracle, sa_cru):
    """K1 alone: dev_soil_hydro's cellpar rows against orc_soil_info for every land cell of the
    reference's bundled South America soils (up to 45 % organic matter, 51 % gravel) and for
    synthetic soils with a missing bulk density (R/splash.point.R:276)."""
    import torch
    land = ~np.isnan(sa_cru["elev"])
    soil = sa_cru["soil"][:, land].copy()
    n0 = soil.shape[1]
    rng = np.random.default_rng(8)
    extra = np.stack([rng.uniform(5, 90, 500), rng.uniform(2, 60, 500), rng.uniform(0.1, 40, 500),
                      rng.uniform(0, 60, 500), np.full(500, np.nan), rng.uniform(0.3, 3.0, 500)])
    soil = np.concatenate([soil, extra], axis=1)
    n = soil.shape[1]
    au = np.stack([rng.uniform(0, 1e7, n), rng.integers(1, 4, n).astype(float), np.full(n, 3.0)])
    res = rng.uniform(100, 60000, n)
    lat, elev, slop, asp = rng.uniform(-60, 60, n), rng.uniform(0, 4000, n), rng.uniform(0, 40, n), \
        rng.uniform(0, 360, n)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    par = torch.empty(api.NPAR, n, dtype=torch.float64, device="cuda")
    api.dev_soil_hydro(n, t(lat), t(elev), t(slop), t(asp), t(res), t(soil), t(au), par)
    torch.cuda.synchronize()
    par = par.cpu().numpy()
    ref = np.empty((13, n))
    for i in range(n):
        si, wmax = oracle.soil_info(soil[:, i], au[:, i], res[i])
        ref[:12, i], ref[12, i] = si, wmax
    assert n0 > 6000
    assert np.array_equal(par[0], lat) and np.array_equal(par[1], elev) and np.array_equal(par[2], slop)
    assert np.array_equal(par[3], asp - 180)
    names = ("SAT", "WP", "FC", "Ksat", "lambda", "depth", "bub_press", "RES", "Au", "Ai", "ncellin",
             "ncellout", "Wmax_R")
    for k, nm_ in enumerate(names):
        rel, ab = rel_err(par[4 + k], ref[k])
        assert rel <= 1e-12, f"cellpar.{nm_}: rel {rel:.2e} abs {ab:.2e}"


def test_hot_path_moist_surf_and_inf_GA(api, oracle):
    """The day loop's OWN moist_surf()/inf_GA() (shared log/exp algebra, reciprocal divisions,
    surface moisture only inside the ponding branch: splash_device.hpp) against the literal
    restatement of src/SPLASH.cpp:1921-1961, 1963-2023 -- including every branch: no ponding,
    saturated surface, tp guard, cap at P, dry end (wn = RES), wn < RES, wn = SAT."""
    rng = np.random.default_rng(21)
    n = 20000
    st = oracle.synth_statics(5150, 100, 200, 0, n, 1)
    par = np.empty((17, n))
    for i in range(n):
        si, wmax = oracle.soil_info(st["soil"][:, i], st["au"][:, i], st["resolution"][i])
        par[4:16, i], par[16, i] = si, wmax
    par[0], par[1], par[2], par[3] = st["lat"], st["elev"], st["slop"], st["asp"] - 180
    SAT, RES, Ksat = par[4], par[11], par[7]
    u = rng.random(n)
    wn = RES + (SAT - RES) * u
    pick = rng.integers(0, 10, n)
    wn = np.where(pick == 0, RES, np.where(pick == 1, SAT, np.where(pick == 2, RES - 1.0, wn)))
    kv = Ksat * rng.uniform(0.4, 1.6, n)
    P = np.where(rng.random(n) < 0.2, 0.0, kv * 6.0 * np.exp(rng.normal(0.5, 1.5, n)))
    th, inf = api.infiltration_probe(par, wn, kv, P)
    L = oracle.lib
    th_ref, inf_ref = np.empty(n), np.empty(n)
    for i in range(n):
        depth, bub, lam = par[9, i], par[10, i], par[8, i]
        th_ref[i] = L.orc_moist_surf(depth, 10.0, bub, wn[i], SAT[i], RES[i], lam)
        inf_ref[i] = L.orc_inf_GA(bub, th_ref[i], kv[i], SAT[i] / (depth * 1000.0), lam, P[i], 6.0,
                                  st["slop"][i])
    ponding = P / 6.0 > kv
    assert 0.2 < ponding.mean() < 0.8 and (inf_ref < P).mean() > 0.1 and (inf_ref == P).mean() > 0.1
    r1, a1 = rel_err(th, th_ref)
    r2, a2 = rel_err(inf, inf_ref)
    print(f"moist_surf rel {r1:.2e}, inf_GA rel {r2:.2e}")
    assert r1 <= 1e-11 and r2 <= 1e-9


def test_basin_totals_are_exact_and_split_independent(api, oracle):
    """K7: monthly water-balance totals per basin id.  The accumulators are fixed-point integers
    (include/splash_hip.h), so the totals are (i) the exact sums of the cells' monthly values,
    (ii) bitwise identical however the grid is cut into launches or spread over devices (here: two
    workers on the one GPU), and an all-reduce of the limbs over ranks is exact."""
    H, W, nt = 16, 64, 400
    n = H * W
    d = _synth(oracle, H, W, nt, 1, seed=606)
    st = d["st"]
    rng = np.random.default_rng(4)
    nb = 7
    # spatially coherent basins (blocks of columns) with a noisy border and a few cells outside any basin
    basin = (np.arange(n) % W) * nb // W
    flip = rng.random(n) < 0.1
    basin[flip] = rng.integers(0, nb, flip.sum())
    basin[rng.random(n) < 0.02] = -1
    kw = dict(time_index=d["dates"], sim_control=dict(monthly_out=False), want_diag=True, basin=basin,
              n_basin=nb)
    args = (d["sw"], d["tc"], d["pn"], st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
            st["resolution"])
    g = api.splash_grid(*args, **kw)
    g2 = api.splash_grid(*args, max_cells_per_launch=300, **kw)
    g3 = api.splash_grid(*args, devices=[0, 0], **kw)
    assert np.array_equal(g["diag_acc"], g2["diag_acc"]) and np.array_equal(g["diag_acc"], g3["diag_acc"])
    for k in ("wn", "ro", "aet", "pet", "snow", "bflow", "cond", "netr", "sm_lim"):
        assert np.array_equal(g[k], g3[k]), k
    # Round 6: the in-library split over devices is a round-robin deal of chunks of 256 cells (1 024 cells: four
    # chunks, so with three workers the first holds two chunks that are NOT neighbours: a block of two runs on
    # the dense fast path), with and without NA cells (gather path), with single-precision result transport
    # -- bit for bit the one-device call, totals included (R/splash.grid.R:264-268, 312-314)
    g4 = api.splash_grid(*args, devices=[0, 0, 0], want_aux=True, **kw)
    g1a = api.splash_grid(*args, want_aux=True, **kw)
    assert np.array_equal(g["diag_acc"], g4["diag_acc"])
    for k in ("wn", "ro", "aet", "pet", "snow", "bflow", "cond", "netr", "sm_lim"):
        assert np.array_equal(g[k], g4[k]), k
    for k in g1a["aux"]:
        assert np.array_equal(g1a["aux"][k], g4["aux"][k], equal_nan=True), k
    for k in range(5):
        assert np.array_equal(g1a["state_end"][k], g4["state_end"][k], equal_nan=True), k
    elev_na = st["elev"].copy()
    elev_na[rng.random(n) < 0.3] = np.nan
    args_na = (d["sw"], d["tc"], d["pn"], elev_na) + args[4:]
    h1 = api.splash_grid(*args_na, **kw)
    h3 = api.splash_grid(*args_na, devices=[0, 0, 0], **kw)
    assert np.array_equal(h1["diag_acc"], h3["diag_acc"])
    for k in ("wn", "ro", "aet", "snow", "sm_lim"):
        assert np.array_equal(h1[k], h3[k], equal_nan=True), k
    mid, nm = d["mid"], d["nm"]
    last = np.array([np.where(mid == m)[0][-1] for m in range(nm)])
    for b in range(nb):
        cells = basin == b
        assert (g["diag_cells"][b] == cells.sum()).all()
        for k in ("aet", "ro", "bflow"):
            ref = np.array([g[k][mid == m][:, cells].astype(np.longdouble).sum() for m in range(nm)])
            assert np.allclose(g["diag"][k][b], ref.astype(float), rtol=1e-13, atol=1e-10), (b, k)
        for k in ("wn", "snow"):
            ref = g[k][last][:, cells].astype(np.longdouble).sum(1).astype(float)
            assert np.allclose(g["diag"][k][b], ref, rtol=1e-13, atol=1e-10), (b, k)
    # against the oracle's cells (tolerance of the series themselves)
    o, aux = oracle.grid(d["doy"], d["year"], d["month"], mid, nm, d["sw"], d["tc"], d["pn"], st["lat"],
                         st["elev"], st["slop"], st["asp"], st["soil"], st["au"], st["resolution"])
    for b in range(nb):
        cells = (basin == b) & (aux["critical"] == 0)
        if cells.sum() == (basin == b).sum():
            ref = np.array([o["aet"][mid == m][:, cells].sum() for m in range(nm)])
            assert np.allclose(g["diag"]["aet"][b], ref, rtol=1e-6)
    # host-side finalize of summed limbs == what the library returned
    tot, cnt = api.diag_finalize_host(g["diag_acc"])
    assert np.array_equal(tot[:, :, 1], g["diag"]["aet"])


def test_single_precision_transport_is_lossless(api, oracle):
    """forcing handed over as float32 (the reference's rasters are FLT4S; the synthetic forcing is
    float-representable) gives bit-identical results to the float64 hand-over"""
    H, W, nt = 4, 64, 400
    d = _synth(oracle, H, W, nt, 1, seed=17)
    st = d["st"]
    f32 = [x.astype(np.float32) for x in (d["sw"], d["tc"], d["pn"])]
    for a, b in zip(f32, (d["sw"], d["tc"], d["pn"])):
        assert np.array_equal(a.astype(np.float64), b)
    rest = (st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"])
    kw = dict(time_index=d["dates"], sim_control=dict(monthly_out=False))
    g64 = api.splash_grid(d["sw"], d["tc"], d["pn"], *rest, **kw)
    g32 = api.splash_grid(*f32, *rest, **kw)
    for k in ("wn", "ro", "aet", "pet", "snow"):
        assert np.array_equal(g64[k], g32[k]), k
    # with NA cells too (gather path)
    f32[1][:, 5] = np.nan
    g32 = api.splash_grid(*f32, *rest, **kw)
    assert np.isnan(g32["wn"][:, 5]).all() and np.array_equal(g32["wn"][:, 6:], g64["wn"][:, 6:])


# ---------------------------------------------------------------- unSWC epilogue
def test_unswc_bourne_and_grid(api, oracle, bourne):
    """R/unsSWC.R on the README workflow (README.md:52-59: water to the 0.49 m sensor depth from the
    simulated column water) and on a synthetic grid.  The analytical integral (:41-43) is a
    difference of two terms that cancel for dry columns; entries whose cancellation factor
    exceeds 1e7 (reference = rounding noise) are compared with the matching looser bound."""
    b, s = bourne, bourne["site"]
    g = api.splash_point(b["sw_in"], b["Ta"], b["P"], s["latitude"], s["elev_m"], s["slop_250m"],
                         s["asp_250m"], b["soil"], s["Aups_250m"], s["resolution"],
                         time_index=b["dates"])
    got = api.unSWC(b["soil"], s["max_depth_sm"], g["wn"])
    ref = oracle.unswc(b["soil"], s["max_depth_sm"], g["wn"])
    for k in ("wn_z", "wtd", "theta_i"):
        tol = 1e-5 * np.abs(ref[k]) + 1e-9 + 1e-15 * ref["cond"] * np.abs(ref[k])
        assert (np.abs(got[k] - ref[k]) <= tol).all(), k
    assert np.all(got["wn_z"] > 0) and np.all(got["wtd"] <= b["soil"][5] + 1e-12)
    # grid form
    H, W, nt = 6, 32, 120
    rng = np.random.default_rng(9)
    st = oracle.synth_statics(20221130, H, W, 0, H * W, 1)
    soil = st["soil"]
    sat = np.array([oracle.soil_info(soil[:, j], None, 250.0)[0][[0, 7]] for j in range(H * W)])
    wn = sat[:, 1] + (sat[:, 0] - sat[:, 1]) * rng.uniform(0, 1, (nt, H * W))
    wn[0] = sat[:, 1]
    wn[1] = sat[:, 0]
    got = api.unSWC_cells(soil, 0.3, wn)
    os.environ["SPLASH_HIP_FAKE_DEVICE_BYTES"] = "1"        # a pretend device: column blocks of 256 cells (192 here: one)
    try:
        wide = np.concatenate([wn] * 3, axis=1)              # 576 cells: three blocks
        blocked = api.unSWC_cells(np.concatenate([soil] * 3, axis=1), 0.3, wide)
    finally:
        os.environ.pop("SPLASH_HIP_FAKE_DEVICE_BYTES")
    for k in got:
        assert np.array_equal(blocked[k], np.concatenate([got[k]] * 3, axis=1), equal_nan=True), "column blocks: " + k
    bad = 0
    for j in range(H * W):
        ref = oracle.unswc(soil[:, j], 0.3, wn[:, j])
        for k in ("wn_z", "wtd", "theta_i"):
            tol = 1e-5 * np.abs(ref[k]) + 1e-9 + 1e-14 * ref["cond"] * np.abs(ref[k])
            bad += int((np.abs(got[k][:, j] - ref[k]) > tol).sum())
    assert bad == 0


@pytest.mark.parametrize("uns_depth", [0.3, 2.5])
def test_unswc_grid_semantics(api, oracle, uns_depth):
    """R/unsSWC.grid.R:14-141 (the GRIDDED function: no stoniness correction, no patch, the integral's water table
    capped at uns_depth, four rasters) on the device: splash_unswc_grid_run (host buffers, here forced into several
    column blocks) and splash_dev_unswc_grid (device rasters) against the oracle's restatement of the gridded function
    and, to rounding, against the host backend running the same two functions."""
    import torch
    from test_host_backend import _assert_unswc_grid, _unswc_grid_case
    soil, wn = _unswc_grid_case(oracle, H=8, W=64, nt=90, seed=5)
    got = api.unSWC_grid(soil, uns_depth, wn)
    _assert_unswc_grid(oracle, got, soil, wn, uns_depth)
    os.environ["SPLASH_HIP_FAKE_DEVICE_BYTES"] = "1"        # a pretend device: column blocks of the minimum, 256 cells
    try:
        blocked = api.unSWC_grid(soil, uns_depth, wn)
    finally:
        os.environ.pop("SPLASH_HIP_FAKE_DEVICE_BYTES")
    for k in got:
        assert np.array_equal(blocked[k], got[k], equal_nan=True), "column blocks: " + k
    host = api.unSWC_grid(soil, uns_depth, wn, backend="host")
    for k in got:
        assert np.array_equal(np.isnan(got[k]), np.isnan(host[k])), k
        # (the integral is a difference of two terms that cancel for dry columns: host and device may differ by the
        # cancellation factor times an ulp there, like either of them and the oracle)
    nt, n = wn.shape
    dev = torch.device("cuda:0")
    d_wn, d_soil = torch.from_numpy(wn).to(dev), torch.from_numpy(soil).to(dev)
    outs = [torch.empty(nt, n, dtype=torch.float64, device=dev) for _ in range(4)]
    api.dev_unswc_grid(n, nt, d_wn, n, d_soil, uns_depth, *outs, n)
    torch.cuda.synchronize()
    for k, o in zip(("w_z", "wtd", "Se", "theta_mean"), outs):
        assert np.array_equal(o.cpu().numpy(), got[k], equal_nan=True), k      # same kernel: same bits


def test_survey_named_hooks_on_the_device(api, oracle, bourne, sa_cru):
    """splash_day_step and splash_soil_hydro (the test hooks SURVEY.md 8(b) names) with the HIP backend: the device's
    day step against the oracle's run_one_day on days of the Bourne series, K1 against the host backend's (the same
    function compiled twice) on the reference's South America soils.  tests/test_host_backend.py has the host twin."""
    b, s = bourne, bourne["site"]
    si, _ = oracle.soil_info(b["soil"], [s["Aups_250m"], 3, 3], s["resolution"])
    Tt, rain, snowf = oracle.snow_partition(b["Ta"], b["P"], b["month"], s["latitude"], s["elev_m"])
    asp = s["asp_250m"] - 180
    state = np.array([si[0] * 0.6, 40.0, 0.0, 0.0, 3.0])
    for t in range(0, 200, 9):
        smr, etr = oracle.run_one_day(s["latitude"], s["elev_m"], int(b["doy"][t]), int(b["year"][t]), state[0], b["sw_in"][t],
                                      b["Ta"][t], rain[t], s["slop_250m"], asp, state[1], snowf[t], si, state[2], state[3],
                                      state[4])
        g = api.day_step(s["latitude"], s["elev_m"], int(b["doy"][t]), int(b["year"][t]), b["sw_in"][t], b["Ta"][t], rain[t],
                         snowf[t], s["slop_250m"], asp, si, state)
        for k, ref in dict(wn=smr["sm"], ro=smr["ro"], snow=smr["swe"], bflow=smr["bflow"], qin=smr["sqout"],
                           snwage=smr["nd"], pet=etr["pet"], aet=etr["aet"], cond=etr["cond"]).items():
            close(np.array([g[k]]), np.array([ref]), f"day_step[{t}].{k}")
        state = np.array([smr["sm"], smr["swe"], smr["sqout"], smr["tdr"], smr["nd"]])
    land = ~np.isnan(sa_cru["elev"])
    soil = sa_cru["soil"][:, land].astype(np.float64)
    n = soil.shape[1]
    rng = np.random.default_rng(2)
    args = (rng.uniform(-60, 60, n), rng.uniform(0, 4000, n), rng.uniform(0, 40, n), rng.uniform(0, 360, n),
            rng.uniform(100, 60000, n), soil)
    dev, host = api.soil_hydro(*args), api.soil_hydro(*args, backend="host")
    for k in range(api.NPAR):
        close(dev[k], host[k], f"cellpar[{k}]", rtol=1e-12)


def test_gdaldem_terrain_vs_oracle(api, oracle):
    """splash_terrain_gdal_run / splash_dev_terrain_gdal (the reference's large-grid terrain, R/splash.grid.R:83-84,
    restated from gdaldem's published algorithm) against the oracle's restatement on a rough DEM with NA patches, a
    plateau and fractional elevations -- interior, edges, corners, cells next to NA --, with and without the rounding to
    whole metres; grid_statics(terrain="gdal" / "reference") hands it to splash.grid."""
    import torch
    from rsplash_amd import grid
    H, W = 157, 203
    rng = np.random.default_rng(17)
    yy, xx = np.mgrid[0:H, 0:W]
    dem = 800 + 500 * np.sin(xx / 17.0) * np.cos(yy / 11.0) + 3.0 * yy + rng.uniform(0, 40, (H, W))
    dem[rng.random((H, W)) < 0.03] = np.nan
    dem[40:60, 70:95] = 1234.0                                  # a plateau: flat cells
    dem[100:110, 150:170] = np.nan
    dem[0, 5] = dem[H - 1, W - 1] = np.nan                      # NA on the edge lines and in a corner
    for rnd in (True, False):
        sl, asp = api.terrain_gdal(dem, 0.05, 0.05, 111120.0, round_int=rnd)
        rsl, rasp = oracle.terrain_gdal(dem, 0.05, 0.05, 111120.0, round_int=rnd)
        assert np.array_equal(np.isnan(sl), np.isnan(dem)) and np.array_equal(np.isnan(asp), np.isnan(dem))
        assert np.allclose(sl, rsl, rtol=1e-12, atol=1e-12, equal_nan=True)
        assert np.allclose(asp, rasp, rtol=1e-12, atol=1e-9, equal_nan=True)
        assert (asp[45:55, 75:90] == 0).all() and (sl[45:55, 75:90] == 0).all()
    dev = torch.device("cuda:0")
    d = torch.from_numpy(dem).to(dev)
    o1, o2 = torch.empty(H, W, dtype=torch.float64, device=dev), torch.empty(H, W, dtype=torch.float64, device=dev)
    api.check(api.lib.splash_dev_terrain_gdal(H, W, d.data_ptr(), 0.05, 0.05, 111120.0, 0, o1.data_ptr(), o2.data_ptr(), None),
              "splash_dev_terrain_gdal")
    torch.cuda.synchronize()
    assert np.array_equal(o1.cpu().numpy(), sl, equal_nan=True) and np.array_equal(o2.cpu().numpy(), asp, equal_nan=True)
    R = grid.Raster(dem, ymax=46.5, xmin=7.0, yres=0.05, xres=0.05)
    g_ref = grid.grid_statics(R)                                 # the default follows the reference: 0.05 degree < 0.1, the small-grid branch
    g_gd = grid.grid_statics(R, terrain="gdal")
    assert g_ref["terrain"] == "raster" and g_gd["terrain"] == "gdal"
    assert np.allclose(g_gd["slop"], api.terrain_gdal(dem, 0.05, 0.05)[0], equal_nan=True)
    assert np.array_equal(g_gd["lat"], g_ref["lat"], equal_nan=True) and np.array_equal(g_gd["Au"], g_ref["Au"], equal_nan=True)
    R2 = grid.Raster(dem, ymax=46.5, xmin=7.0, yres=0.25, xres=0.25)
    assert grid.grid_statics(R2)["terrain"] == "gdal"            # 0.1 < res < 1: the large-grid branch
    assert grid.grid_statics(R2, terrain="raster")["terrain"] == "raster"
    # the large-grid branch's AREA comes from the DEM as 16-bit integers (R/upslope_area.R:64): whole metres
    g2 = grid.grid_statics(R2)
    Rint = grid.Raster(np.rint(dem), ymax=46.5, xmin=7.0, yres=0.25, xres=0.25)
    assert g2["method"] == "tau" and np.array_equal(g2["Au"], grid.grid_statics(Rint)["Au"], equal_nan=True)
    assert not np.array_equal(g2["Au"][0], grid.grid_statics(R2, method="top")["Au"][0], equal_nan=True)


def test_splash_grid_leap_year_calendar(api, oracle):
    """calendar crossing a leap year (2003-2005: doy 366, kN = 366 in the orbit, months of 29 days)
    on flat terrain, monthly output (the reference's default for splash.grid)"""
    H, W, nt = 6, 32, 1000
    doy, year, month, mid, nm = oracle.calendar(2003, nt)
    assert 366 in doy
    st = oracle.synth_statics(7, H, W, 0, H * W, 0)
    sw, tc, pn = oracle.synth_forcing(7, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
    dates = np.datetime64("2003-01-01") + np.arange(nt)
    g = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"],
                        st["au"], st["resolution"], time_index=dates)  # monthly_out default TRUE
    o, aux = oracle.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"],
                         st["asp"], st["soil"], st["au"], st["resolution"], monthly_out=True)
    assert g["wn"].shape == (nm, H * W)
    print(assert_whole_pipeline(oracle, g, o, aux, st, doy, year, "leap", monthly_mid=mid))


def test_polar_day_and_night(api, oracle):
    """latitudes 66-85 N/S: polar night (hs = 0: src/SOLAR.cpp:152-154, rw branch :220-221) and
    polar day (hs = 180), deep snowpacks, tc < 0 most of the year (viscosity at 0 C)"""
    H, W, nt = 4, 32, 500
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    st = oracle.synth_statics(11, H, W, 0, H * W, 1)
    n = H * W
    lat = np.r_[np.linspace(66, 85, n // 2), -np.linspace(66, 85, n - n // 2)]
    st["lat"] = lat
    sw, tc, pn = oracle.synth_forcing(11, H, W, 0, n, 0, doy, lat, st["elev"])
    sw[sw < 20] = 0.0  # dark days: sw_in == 0 branch
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    g = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], lat, st["slop"], st["asp"], st["au"],
                        st["resolution"], time_index=dates, sim_control=dict(monthly_out=False))
    o, aux = oracle.grid(doy, year, month, mid, nm, sw, tc, pn, lat, st["elev"], st["slop"],
                         st["asp"], st["soil"], st["au"], st["resolution"])
    assert (sw == 0).mean() > 0.05
    for k in o:
        assert np.isfinite(g[k]).all(), k
    print(assert_whole_pipeline(oracle, g, o, aux, st, doy, year, "polar"))


def test_config3_thirty_years_streamed_in_year_chunks(api, oracle):
    """BASELINE configs[3] at test size: 30 years (2001-2030, real leap days), spin-up, then the
    transient run streamed one calendar year at a time through ONE reused forcing buffer filled by
    the device generator (a 30-year 4096^2 raster cannot be resident), state carried in HBM,
    monthly outputs.  Checked against the oracle's single 10 957-day run."""
    import torch
    dev = torch.device("cuda:0")
    H, W, seed = 2, 32, 31
    n = H * W
    nt = 10957
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    assert year[-1] == 2030 and nm == 360
    cal = api.Calendar(doy, year, month)
    f64 = dict(dtype=torch.float64, device=dev)
    lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
    soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
    api.dev_synth_statics(seed, H, W, 0, n, 1, lat, elev, slop, asp, res, soil, au)
    par = torch.empty(api.NPAR, n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
    # Tt needs the whole series: one streaming pass over tc, chunk by chunk (max of chunk maxima)
    ybounds = [0] + [int(i) + 1 for i in np.where(np.diff(year) != 0)[0]] + [nt]
    buf = [torch.empty(366, n, **f64) for _ in range(3)]
    Tt = torch.full((n,), -np.inf, **f64)
    Ttc = torch.empty(n, **f64)
    for t0, t1 in zip(ybounds[:-1], ybounds[1:]):
        api.dev_synth_forcing(seed, cal, t0, t1 - t0, 0, n, lat, elev, buf[0], buf[1], buf[2], n)
        api.dev_snow_threshold(n, t1 - t0, buf[1], n, par, n, Ttc)
        Tt = torch.maximum(Tt, Ttc)
    # spin-up on the first 365 days
    api.dev_synth_forcing(seed, cal, 0, 365, 0, n, lat, elev, buf[0], buf[1], buf[2], n)
    state = torch.empty(api.NSTATE, n, **f64)
    work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
    api.dev_spin_up(cal, n, buf[0], buf[1], buf[2], n, Tt, par, n, state, n, work)
    names = ("wn", "snow", "ro", "aet", "pet")
    mon = {k: torch.zeros(nm, n, **f64) for k in names}
    for t0, t1 in zip(ybounds[:-1], ybounds[1:]):
        api.dev_synth_forcing(seed, cal, t0, t1 - t0, 0, n, lat, elev, buf[0], buf[1], buf[2], n)
        outs = {k: mon[k][int(mid[t0]):] for k in names}
        api.dev_run_all(cal, t0, t1 - t0, n, buf[0], buf[1], buf[2], n, Tt, par, n, state, n, outs,
                        n, monthly=True)
    torch.cuda.synchronize()
    st = oracle.synth_statics(seed, H, W, 0, n, 1)
    sw, tc, pn = oracle.synth_forcing(seed, H, W, 0, n, 0, doy, st["lat"], st["elev"])
    o, aux = oracle.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"],
                         st["asp"], st["soil"], st["au"], st["resolution"], monthly_out=True)
    assert np.array_equal(Tt.cpu().numpy(), aux["Tt"])
    print(assert_whole_pipeline(oracle, {k: mon[k].cpu().numpy() for k in names}, o, aux, st, doy, year,
                                "30y", names=names, monthly_mid=mid))
    cal.close()


def test_config4_cru_shaped_land_mask_with_grid_totals(api, oracle):
    """BASELINE configs[4] at test size: a lon/lat grid with 72 % NA (ocean) cells in the layout of
    data/SA_cru.RData, cell size varying with latitude (resolution = sqrt(area)*1000,
    R/splash.grid.R:98), monthly outputs and the monthly grid totals used for the basin /
    water-balance diagnostics."""
    H, W, nt = 20, 24, 730
    n = H * W
    doy, year, month, mid, nm = oracle.calendar(2014, nt)
    st = oracle.synth_statics(5, H, W, 0, n, 1)
    lat = np.repeat(13.25 - 0.5 * np.arange(H) * 3.4, W)  # 13 N ... 19 S in coarse steps
    st["lat"] = lat
    st["resolution"] = np.sqrt(55.6 * 55.6 * np.cos(np.radians(lat))) * 1000.0
    st["au"][0] = st["resolution"] ** 2 * np.exp2(6 * np.random.default_rng(1).random(n))
    sw, tc, pn = oracle.synth_forcing(5, H, W, 0, n, 0, doy, lat, st["elev"])
    land = np.random.default_rng(2).random(n) < 0.28
    land[:2] = False
    for x in (sw, tc, pn):
        x[:, ~land] = np.nan
    elev = np.where(land, st["elev"], np.nan)
    soil = np.where(land[None, :], st["soil"], np.nan)
    dates = np.datetime64("2014-01-01") + np.arange(nt)
    g = api.splash_grid(sw, tc, pn, elev, soil, lat, st["slop"], st["asp"], st["au"],
                        st["resolution"], time_index=dates, want_diag=True, want_aux=True)
    o, aux = oracle.grid(doy, year, month, mid, nm, sw, tc, pn, lat, elev, st["slop"], st["asp"],
                         soil, st["au"], st["resolution"], monthly_out=True)
    for k in o:
        assert np.isnan(g[k][:, ~land]).all() and np.isnan(o[k][:, ~land]).all()
    r = assert_whole_pipeline(oracle, g, o, aux, st, doy, year, "cru", monthly_mid=mid)
    ok = r.pop("ok")
    print(r)
    # grid totals over the land cells (what an all-reduce would sum across GPUs): exact sums of
    # the GPU's own monthly values
    assert (g["diag_cells"] == land.sum()).all()
    for k in ("aet", "ro", "bflow"):
        assert np.allclose(g["diag"][k], np.nansum(g[k].astype(np.longdouble), axis=1).astype(float),
                           rtol=1e-12, atol=1e-9)
        if (ok | ~land).all():
            assert np.allclose(g["diag"][k], np.nansum(o[k], axis=1), rtol=1e-9)


def test_minigrid_against_committed_golden(api, oracle, minigrid):
    """whole pipeline on device-generated inputs against COMMITTED vectors (no live oracle run in
    the comparison): 64 cells x 730 days, monthly outputs of all nine series"""
    import torch
    g = minigrid
    H, W, nt, seed = int(g["H"]), int(g["W"]), int(g["nt"]), int(g["seed"])
    n = H * W
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    cal = api.Calendar.from_dates(dates)
    f64 = dict(dtype=torch.float64, device="cuda")
    lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
    soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
    api.dev_synth_statics(seed, H, W, 0, n, 1, lat, elev, slop, asp, res, soil, au)
    sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
    api.dev_synth_forcing(seed, cal, 0, nt, 0, n, lat, elev, sw, tc, pn, n)
    torch.cuda.synchronize()
    cal.close()
    h = lambda t: t.cpu().numpy()
    out = api.splash_grid(h(sw), h(tc), h(pn), h(elev), h(soil), h(lat), h(slop), h(asp), h(au), h(res),
                          time_index=dates, sim_control=dict(monthly_out=True), want_aux=True)
    ok = g["well_conditioned"]
    assert ok.sum() >= 50
    assert np.array_equal(out["aux"]["passes2"][ok], g["passes2"][ok])
    for k in ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "sm_lim"):
        close(out[k][:, ok], g[k][:, ok], "minigrid." + k)
    close(out["state_end"][:3][:, ok], g["state_end"][:3][:, ok], "minigrid.state_end")


# ---------------------------------------------------------------- error behaviour of the boundary
def test_edge_shapes_all_na_one_cell_and_thirty_years_through_the_host_entry_point(api, oracle):
    """Ragged and extreme shapes through splash_grid_run (what R binds): a grid of NA cells only (no
    launch, NaN everywhere, no error), one valid cell among NA cells, one cell, a width that is no
    multiple of the wavefront in several launches, and a 30-year series (10 957 days with the real
    leap days: month bookkeeping, the year table, 32-bit products of nt and ncell) against the oracle."""
    d = _synth(oracle, 3, 37, 400, 1)
    st = d["st"]
    n = 3 * 37
    args = lambda sw, tc, pn, elev: (sw, tc, pn, elev, st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"])
    kw = dict(time_index=d["dates"], sim_control=dict(monthly_out=False), want_diag=True)
    # every cell NA
    nan_f = np.full_like(d["sw"], np.nan)
    g = api.splash_grid(*args(nan_f, nan_f, nan_f, np.full(n, np.nan)), **kw)
    for k in api.OUT_NAMES:
        assert g[k].shape == (400, n) and np.isnan(g[k]).all(), k
    assert np.isnan(g["state_end"]).all() and (g["diag_cells"] == 0).all()
    full = api.splash_grid(*args(d["sw"], d["tc"], d["pn"], st["elev"]), **kw)
    # one valid cell among NA cells == that cell of the full grid, bit for bit
    keep = 53
    sw, tc, pn, elev = d["sw"].copy(), d["tc"].copy(), d["pn"].copy(), st["elev"].copy()
    m = np.ones(n, bool)
    m[keep] = False
    for a in (sw, tc, pn):
        a[:, m] = np.nan
    elev[m] = np.nan
    one = api.splash_grid(*args(sw, tc, pn, elev), **kw)
    for k in api.OUT_NAMES:
        assert np.isnan(one[k][:, m]).all() and np.array_equal(one[k][:, keep], full[k][:, keep]), k
    assert (one["diag_cells"] == 1).all()
    # ncell = 1 as a grid, and as a point: the same numbers
    c1 = api.splash_grid(d["sw"][:, keep:keep + 1], d["tc"][:, keep:keep + 1], d["pn"][:, keep:keep + 1],
                         st["elev"][keep:keep + 1], st["soil"][:, keep:keep + 1], st["lat"][keep:keep + 1],
                         st["slop"][keep:keep + 1], st["asp"][keep:keep + 1], st["au"][:, keep:keep + 1],
                         st["resolution"][keep:keep + 1], **kw)
    for k in api.OUT_NAMES:
        assert np.array_equal(c1[k][:, 0], full[k][:, keep]), k
    # 111 cells in launches of 40: the last one ragged
    blk = api.splash_grid(*args(d["sw"], d["tc"], d["pn"], st["elev"]), max_cells_per_launch=40, **kw)
    for k in api.OUT_NAMES:
        assert np.array_equal(blk[k], full[k]), k
    assert np.array_equal(blk["diag_acc"], full["diag_acc"])
    # thirty years, monthly and daily
    H, W, nt = 2, 24, 10957
    d = _synth(oracle, H, W, nt, 1, seed=606)
    st = d["st"]
    assert str(d["dates"][-1]) == "2030-12-31" and d["nm"] == 360
    for monthly in (True, False):
        g = api.splash_grid(d["sw"], d["tc"], d["pn"], st["elev"], st["soil"], st["lat"], st["slop"], st["asp"],
                            st["au"], st["resolution"], time_index=d["dates"], sim_control=dict(monthly_out=monthly))
        o, aux = oracle.grid(d["doy"], d["year"], d["month"], d["mid"], d["nm"], d["sw"], d["tc"], d["pn"], st["lat"],
                             st["elev"], st["slop"], st["asp"], st["soil"], st["au"], st["resolution"], monthly_out=monthly)
        assert g["wn"].shape == ((360 if monthly else nt), H * W)
        r = assert_whole_pipeline(oracle, g, o, aux, st, d["doy"], d["year"], f"30 years[{monthly}]",
                                  monthly_mid=d["mid"] if monthly else None)
        r.pop("ok")
        print(r)


def test_boundary_rejects_bad_arguments(api, oracle, bourne):
    """errors are codes + messages (never a crash, never a silent NaN grid): too-short series
    (R/splash.point.R:141-144 needs 365 days), malformed Au, mismatched time axis, bad leading
    dimensions and NULL pointers at the device-level entry points"""
    import ctypes as C
    import torch
    b, s = bourne, bourne["site"]
    with pytest.raises(api.SplashError, match="365"):
        api.splash_point(b["sw_in"][:200], b["Ta"][:200], b["P"][:200], s["latitude"], s["elev_m"], 0.0,
                         0.0, b["soil"], 0.0, 250.0, time_index=b["dates"][:200])
    with pytest.raises(api.SplashError, match="1 or 3"):
        api.splash_point(b["sw_in"], b["Ta"], b["P"], s["latitude"], s["elev_m"], 0.0, 0.0, b["soil"],
                         [1.0, 2.0], 250.0, time_index=b["dates"])
    with pytest.raises(ValueError, match="time_index"):
        api.splash_point(b["sw_in"], b["Ta"], b["P"], s["latitude"], s["elev_m"], 0.0, 0.0, b["soil"],
                         0.0, 250.0, time_index=b["dates"][:-3])
    d = _synth(oracle, 4, 16, 365, 1)
    st = d["st"]
    with pytest.raises(api.SplashError, match="365"):
        api.splash_grid(d["sw"][:100], d["tc"][:100], d["pn"][:100], st["elev"], st["soil"], st["lat"],
                        st["slop"], st["asp"], st["au"], st["resolution"], time_index=d["dates"][:100])
    lib = api.lib
    err = C.create_string_buffer(256)
    assert lib.splash_grid_run(None, None, None, err, 256) < 0 and b"NULL" in err.value
    # device-level entry points: leading dimension smaller than ncell, day range outside the calendar
    cal = api.Calendar.from_dates(d["dates"])
    n = 64
    f64 = dict(dtype=torch.float64, device="cuda")
    tc = torch.zeros(365, n, **f64)
    par, Tt, state = torch.zeros(api.NPAR, n, **f64), torch.zeros(n, **f64), torch.zeros(api.NSTATE, n, **f64)
    with pytest.raises(api.SplashError, match="leading dimension"):
        api.dev_snow_threshold(n, 365, tc, n - 1, par, n, Tt)
    outs = {k: torch.zeros(365, n, **f64) for k in ("wn",)}
    with pytest.raises(api.SplashError, match="day range"):
        api.dev_run_all(cal, 300, 100, n, tc, tc, tc, n, Tt, par, n, state, n, outs, n)
    assert b"day range" in lib.splash_hip_last_error()
    cal.close()


# ---------------------------------------------------------------- static rasters
def test_terrain_and_ncellflow_vs_oracle(api, oracle):
    """slope/aspect (Horn), latitude, resolution, D8 flow direction and ncellflow
    (R/splash.grid.R:96-110, R/upslope_area.R:31-33, 140-165) on a rough lon/lat DEM with NA
    patches, against the oracle's restatement (published algorithms: parity unpinned)."""
    rng = np.random.default_rng(12)
    H, W = 301, 517
    y, x = np.mgrid[0:H, 0:W]
    z = 800 + 600 * np.sin(x / 37.0) * np.cos(y / 23.0) + 40 * rng.standard_normal((H, W))
    z[rng.random((H, W)) < 0.03] = np.nan
    z[100:140, 200:260] = np.nan            # a lake of NA
    z[180:200, 300:330] = 500.0             # a plateau: flat cells (aspect 90, ties in the D8 choice)
    for na0 in (True, False):
        g = api.terrain(z, 47.5, 0.01, 0.0125, na_to_zero=na0)
        o = oracle.terrain(z, 47.5, 0.01, 0.0125, na_to_zero=na0)
        for k in ("flowdir", "ncellin", "ncellout"):
            assert np.array_equal(g[k], o[k], equal_nan=True), k
        for k in ("lat", "resolution"):  # (device cos / fused multiply-add: an ulp)
            assert np.allclose(g[k], o[k], rtol=1e-14, atol=0, equal_nan=True), k
        for k in ("slope", "aspect"):
            assert np.array_equal(np.isnan(g[k]), np.isnan(o[k])), k
            m = ~np.isnan(o[k])
            assert np.allclose(g[k][m], o[k][m], rtol=1e-12, atol=1e-10), k
    assert (g["aspect"][185:195, 305:325] == 90.0).all() and (g["slope"][185:195, 305:325] == 0).all()
    assert np.isnan(g["slope"][0]).all() and 0 < np.nanmax(g["slope"]) < 60


def test_sinkfill_and_flow_accumulation_vs_oracle(api, oracle):
    """SURVEY 8f-1, the graph part of R/upslope_area.R: sink fill (fixed point of a monotone
    relaxation: bit-identical to the serial sweeps whatever the update order) and D8 accumulation
    (integer counts: identical whatever the order of the atomic adds), on a rough DEM with NA
    patches, pits and a long valley, both code conventions; then the whole upslope_area() chain."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(20221130)
    H, W = 301, 517
    rows, cols = np.mgrid[0:H, 0:W]
    z = (400 + 3.0 * np.hypot(rows - 150, 0.6 * (cols - 250)) + 60 * np.sin(rows / 17.0) * np.cos(cols / 23.0)
         + 25 * rng.random((H, W)))
    z[100:110, 200:240] = np.nan                      # a lake / missing tile
    z[40:43, 40:43] -= 200                            # a deep pit
    z[250, :] = np.minimum(z[250, :], 300 - 0.05 * cols[250])  # a long valley draining east
    zt = torch.as_tensor(z, device=dev)
    for degree in (0.0, 1.0):
        filled = torch.empty(H * W, dtype=torch.float64, device=dev)
        sweeps = api.dev_sinkfill(H, W, zt, 250.0, filled, degree)
        ref = oracle.sinkfill(z, 250.0, degree)
        got = filled.cpu().numpy().reshape(H, W)
        assert sweeps > 0
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)]), f"sink fill, degree {degree}"
        assert (got[~np.isnan(z)] >= z[~np.isnan(z)]).all() and (got > z).sum() > 100
    # flow direction of the filled surface, then the accumulation, against the oracle's
    t = oracle.terrain(ref, 60.0, 0.0025, 0.0025, na_to_zero=False)
    fd, res = t["flowdir"], t["resolution"]
    cnt_o, area_o = oracle.flow_accumulation(fd, elev=ref, resolution=res)
    cnt = torch.empty(H * W, dtype=torch.float64, device=dev)
    area = torch.empty(H * W, dtype=torch.float64, device=dev)
    levels = api.dev_flow_accumulation(H, W, torch.as_tensor(fd, device=dev), cnt, area,
                                       elev=torch.as_tensor(ref, device=dev),
                                       resolution=torch.as_tensor(res, device=dev))
    g = cnt.cpu().numpy().reshape(H, W)
    assert levels >= 1
    assert np.array_equal(np.isnan(g), np.isnan(cnt_o)) and np.array_equal(g[~np.isnan(g)], cnt_o[~np.isnan(g)])
    ga = area.cpu().numpy().reshape(H, W)
    m = ~np.isnan(area_o)
    assert np.array_equal(np.isnan(ga), ~m) and np.array_equal(ga[m], area_o[m])
    assert np.nanmax(g) > 2000 and np.nanmin(g) == 1   # the valley collects a large catchment
    # without the elevation mask cells are where the flow code is not NA; TauDEM codes agree
    cnt2_o, _ = oracle.flow_accumulation(fd)
    api.dev_flow_accumulation(H, W, torch.as_tensor(fd, device=dev), cnt)
    g2 = cnt.cpu().numpy().reshape(H, W)
    assert np.array_equal(np.isnan(g2), np.isnan(cnt2_o)) and np.array_equal(g2[~np.isnan(g2)], cnt2_o[~np.isnan(g2)])
    q = np.log2(np.where(np.isnan(fd) | (fd < 1), 1.0, fd)).astype(int)   # 0 E, 1 SE, ... 7 NE
    tau = np.where(np.isnan(fd) | (fd < 1), fd, ((9 - q) % 8 + (((9 - q) % 8) == 0) * 8).astype(float))
    tau = np.where(np.isnan(fd) | (fd < 1), fd, np.array([1, 8, 7, 6, 5, 4, 3, 2], float)[q])
    api.dev_flow_accumulation(H, W, torch.as_tensor(tau, device=dev), cnt, tau_codes=True)
    g3 = cnt.cpu().numpy().reshape(H, W)
    assert np.array_equal(g3[~np.isnan(g3)], cnt2_o[~np.isnan(g3)])
    # the whole chain of upslope_area(dem, resolution)
    u = api.upslope_area(z, 60.0, 0.0025, 0.0025)
    rv = res[~np.isnan(res)].astype(np.longdouble)
    mean_res = float(rv.sum() / rv.size + (rv - rv.sum() / rv.size).sum() / rv.size)   # R's mean()
    assert np.array_equal(u["filled"][~np.isnan(z)], oracle.sinkfill(z, mean_res, 1.0)[~np.isnan(z)])
    assert np.nanmin(u["ups_ncell"]) == 1 and np.isnan(u["areacatch"][105, 220])
    # multiple-flow-direction area of the filled surface (topidx()$area): pull order fixed, no
    # contraction: the same doubles as the serial statement
    filled_t = torch.as_tensor(ref, device=dev)
    mfd = torch.empty(H * W, dtype=torch.float64, device=dev)
    lv = api.dev_mfd_area(H, W, filled_t, 250.0, mfd)
    mo = oracle.mfd_area(ref, 250.0)
    gm = mfd.cpu().numpy().reshape(H, W)
    assert lv > 10 and np.array_equal(np.isnan(gm), np.isnan(mo))
    assert np.array_equal(gm[~np.isnan(mo)], mo[~np.isnan(mo)])
    assert np.array_equal(u["areacatch"][~np.isnan(z)], oracle.mfd_area(u["filled"], mean_res)[~np.isnan(z)])
    u_tau = api.upslope_area(z, 60.0, 0.0025, 0.0025, method="tau")
    # the host-buffer entry point the R shim binds gives the same rasters in one call
    sr = api.static_rasters(z, 60.0, 0.0025, 0.0025)
    for k, ku in (("filled", "filled"), ("flowdir", "flowdir"), ("ups_ncell", "ups_ncell"),
                  ("ups_area", "areacatch"), ("ncellin", "ncellin"), ("ncellout", "ncellout")):
        a, b = sr[k], u_tau[ku]
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(a)]), k
    assert np.array_equal(sr["ups_area_mfd"][~np.isnan(z)], u["areacatch"][~np.isnan(z)])
    t0 = api.terrain(z, 60.0, 0.0025, 0.0025)
    for k in ("slope", "aspect", "lat", "resolution"):
        assert np.array_equal(np.isnan(sr[k]), np.isnan(t0[k])) and np.array_equal(sr[k][~np.isnan(sr[k])], t0[k][~np.isnan(sr[k])]), k
    with pytest.raises(Exception):
        api.static_rasters(np.full((4, 4), np.nan), 60.0, 0.0025, 0.0025)   # all NA: no mean resolution


@pytest.mark.gpu
def test_product_bits_are_pinned(api, oracle):
    """The product's OWN bits, pinned across commits: the exact int64 limbs of the monthly
    water-balance totals and the sha256 of all nine result series, the end state, the aridity index and
    the snow threshold of a fixed 64 x 128 x 400-day grid must equal tests/golden/product_bits.json, which
    tools/make_product_bits.py wrote on an MI355X.  "Same bits" claims elsewhere (any blocking, lanes,
    devices, pass mode) are relative to ONE build; this one is absolute: a commit that moves a bit has to
    regenerate the file and say why in DESIGN.md ("product bits changelog")."""
    import json
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_product_bits", os.path.join(ROOT, "tools", "make_product_bits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "product_bits.json")))
    got = mod.compute(api, oracle)
    assert got["diag_acc_shape"] == want["diag_acc_shape"]
    diff = [i for i, (a, b) in enumerate(zip(got["diag_acc"], want["diag_acc"])) if a != b]
    assert not diff, f"{len(diff)} of {len(want['diag_acc'])} limbs of the water-balance totals moved (first at {diff[0]})"
    moved = [k for k in want["sha256"] if got["sha256"][k] != want["sha256"][k]]
    assert not moved, f"bits moved in: {moved} (pinned by: {want.get('why', '')!r})"
    assert got["passes2_sum"] == want["passes2_sum"]
