"""GPU suite at the benchmark's and the BASELINE configurations' TRUE sizes: size-independent properties where the oracle
cannot follow (idempotence, bounds, chunking invariance, totals == sums of rasters), the oracle on bands of rows of the
4096 x 4096 grid of the metric (one-step, trajectory, and each side from its own spin-up), and the bench line's contract.
"""
import os

import numpy as np
import pytest

from conftest import ROOT, rel_err
from gpu_pipeline import (RTOL, assert_one_step, assert_trajectory, assert_whole_pipeline, close, device_pipeline,  # noqa: F401
                          synth_grid as _synth)

pytestmark = pytest.mark.gpu


def test_config1_flat_256x256_properties(api):
    """BASELINE configs[1] size (256x256 flat, 365 d), GPU only: size-independent properties.
    Idempotence (same inputs -> same bits), bounds (RES <= wn <= SAT, 0 <= sm_lim <= 1, snow,
    ro, aet, pet >= 0) and no NaN in the five headline outputs."""
    import torch
    dev = torch.device("cuda:0")
    H = W = 256
    n, nt, seed = H * W, 365, 20221130
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    cal = api.Calendar.from_dates(dates)
    f64 = dict(dtype=torch.float64, device=dev)
    lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
    soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
    api.dev_synth_statics(seed, H, W, 0, n, 0, lat, elev, slop, asp, res, soil, au)
    sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
    api.dev_synth_forcing(seed, cal, 0, nt, 0, n, lat, elev, sw, tc, pn, n)
    par = torch.empty(api.NPAR, n, **f64)
    Tt = torch.empty(n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
    api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
    runs = []
    for _ in range(2):
        state = torch.empty(api.NSTATE, n, **f64)
        p = par.clone()
        work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
        api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, p, n, state, n, work)
        out = {k: torch.empty(nt, n, **f64) for k in ("wn", "snow", "ro", "aet", "pet", "sm_lim")}
        api.dev_run_all(cal, 0, nt, n, sw, tc, pn, n, Tt, p, n, state, n, out, n)
        torch.cuda.synchronize()
        runs.append((out, p))
    a, b = runs[0][0], runs[1][0]
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert torch.isfinite(a[k]).all(), k
    p = runs[0][1]
    SAT, RES = p[4], p[11]
    assert (a["wn"] <= SAT[None] * (1 + 1e-15)).all() and (a["wn"] >= RES[None] * (1 - 1e-15)).all()
    assert (a["sm_lim"] >= 0).all() and (a["sm_lim"] <= 1).all()
    for k in ("snow", "ro", "aet", "pet"):
        assert (a[k] >= 0).all(), k
    cal.close()


def _chk(t):
    """order-independent bit checksum of a device tensor (wrap-around sum of its 64-bit words)"""
    return int(t.contiguous().view(__import__("torch").int64).sum().item())


def test_config4_full_size_0p5_degree_land_mask_30_years_properties(api):
    """BASELINE configs[4] at its TRUE size, GPU only: the 360 x 720 half-degree grid, ~26 % land
    by a hash mask (67 k cells), 30 years (2001-2030) streamed in calendar-year chunks, monthly outputs and monthly
    water-balance totals for 72 basins.  No oracle at this size (30 years x 67 k cells is hours of CPU): the
    size-independent properties -- idempotence (two runs: the same bits in every raster, every exact limb of the
    totals and the end state), bounds (RES <= monthly mean wn <= SAT, snow/ro/aet/pet >= 0, all finite), a sane
    pass-count histogram of the spin-up (1 <= passes <= 1000, median <= 6, under 1 % capped), and totals == the sums
    of the rasters (per basin and month, to 1e-12).  R/splash.grid.R:260-314 with data/SA_cru.RData's layout."""
    import torch
    dev = torch.device("cuda:0")
    torch.cuda.empty_cache()
    Htot, W, years, seed = 360, 720, 30, 20221130
    dates = np.arange(np.datetime64("2001-01-01"), np.datetime64(f"{2001 + years}-01-01"))
    nt = len(dates)
    cal = api.Calendar.from_dates(dates)
    doy, year, month = api.time_fields(dates)
    ybounds = [0] + [int(i) + 1 for i in np.where(np.diff(year) != 0)[0]] + [nt]
    nm = cal.nmonth
    assert nt == 10957 and nm == 360
    f64 = dict(dtype=torch.float64, device=dev)
    nfull = Htot * W
    land = np.random.default_rng(seed).random(nfull) < 0.26
    land_idx = np.where(land)[0]
    n = len(land_idx)
    assert 60000 < n < 75000
    rr, cc = np.divmod(land_idx, W)
    n_basin = 72
    basin = torch.tensor(((rr * 6 // Htot) * 12 + cc * 12 // W).astype(np.int32), device=dev)
    stat = [torch.empty(nfull, **f64) for _ in range(5)]
    soil_f, au_f = torch.empty(6, nfull, **f64), torch.empty(3, nfull, **f64)
    api.dev_synth_statics(seed, Htot, W, 0, nfull, 1, *stat, soil_f, au_f)
    li = torch.tensor(land_idx, device=dev)
    lat, elev, slop, asp, res = (x[li].contiguous() for x in stat)
    soil, au = soil_f[:, li].contiguous(), au_f[:, li].contiguous()
    par0 = torch.empty(api.NPAR, n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par0)
    full = [torch.empty(366, nfull, **f64) for _ in range(3)]
    buf = [torch.empty(366, n, **f64) for _ in range(3)]

    def generate(t0, t1):
        api.dev_synth_forcing(seed, cal, t0, t1 - t0, 0, nfull, stat[0], stat[1], *full, nfull)
        for v in range(3):
            torch.index_select(full[v][:t1 - t0], 1, li, out=buf[v][:t1 - t0])

    Tt = torch.full((n,), -np.inf, **f64)
    Ttc = torch.empty(n, **f64)
    for t0, t1 in zip(ybounds[:-1], ybounds[1:]):
        generate(t0, t1)
        api.dev_snow_threshold(n, t1 - t0, buf[1], n, par0, n, Ttc)
        Tt = torch.maximum(Tt, Ttc)
    names = ("wn", "snow", "ro", "aet", "pet", "bflow")
    runs = []
    for rep in range(2):
        par = par0.clone()
        generate(0, 365)
        state = torch.empty(api.NSTATE, n, **f64)
        work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
        passes = torch.zeros(2, n, dtype=torch.int32, device=dev)
        api.dev_spin_up(cal, n, *buf, n, Tt, par, n, state, n, work, passes=passes)
        mon = {k: torch.zeros(nm, n, **f64) for k in names}
        acc = torch.zeros(n_basin, nm, api.DIAG_REC, dtype=torch.int64, device=dev)
        for t0, t1 in zip(ybounds[:-1], ybounds[1:]):
            generate(t0, t1)
            o_ = {k: mon[k][int(cal.month_id[t0]):] for k in names}
            api.dev_run_all(cal, t0, t1 - t0, n, *buf, n, Tt, par, n, state, n, o_, n, monthly=True,
                            diag_acc=acc, basin=basin, n_basin=n_basin)
        torch.cuda.synchronize()
        runs.append(dict(mon=mon, acc=acc, state=state, passes=passes, par=par))
        del work
    a, b = runs
    for k in names:
        assert torch.equal(a["mon"][k].view(torch.int64), b["mon"][k].view(torch.int64)), k
        assert torch.isfinite(a["mon"][k]).all(), k
    assert torch.equal(a["acc"], b["acc"]) and torch.equal(a["passes"], b["passes"])
    assert torch.equal(a["state"].view(torch.int64), b["state"].view(torch.int64))
    SAT, RES = a["par"][4], a["par"][11]
    assert (a["mon"]["wn"] <= SAT[None] * (1 + 1e-15)).all() and (a["mon"]["wn"] >= RES[None] * (1 - 1e-15)).all()
    for k in ("snow", "ro", "aet", "pet", "bflow"):
        assert (a["mon"][k] >= 0).all(), k
    p2 = a["passes"][1].cpu().numpy()
    hist = np.bincount(np.minimum(p2, 12), minlength=13)
    print(f"config 4 at full size: {n} land cells x {nt} days; passes of spin-up #2: histogram 0..11, >=12 = {hist.tolist()}, "
          f"capped {int((p2 >= 1000).sum())}")
    assert p2.min() >= 1 and p2.max() <= 1000 and np.median(p2) <= 6 and (p2 >= 1000).mean() < 0.01
    diag = torch.zeros(n_basin, nm, api.NDIAG, **f64)
    api.dev_diag_finalize(a["acc"], n_basin, nm, diag)
    d = diag.cpu().numpy()
    bas = basin.cpu().numpy()
    for j, k in ((1, "aet"), (2, "ro"), (3, "bflow")):   # records: precipitation, aet, ro, bflow sums; wn, snow at month end
        m = a["mon"][k].cpu().numpy().astype(np.longdouble)
        tot = np.zeros((n_basin, nm), np.longdouble)
        np.add.at(tot, bas, m.T)
        assert np.allclose(d[:, :, j], tot.astype(float), rtol=1e-12, atol=1e-9), k
    cal.close()


def test_config3_one_ranks_share_three_years_in_year_chunks_properties(api):
    """BASELINE configs[3] at ONE RANK'S TRUE SHARE (512 x 4096 = 2.1 M cells = 1/8 of the 4096 x 4096 grid), three
    years of daily forcing with the spin-up, GPU only.  Properties that do not depend on the size:
    the run streamed in calendar-year chunks with the five-double state as carry must equal the undivided
    1 096-day launch BIT FOR BIT (checksums of the wn and aet rasters year by year, the end state word for word);
    bounds and finiteness of every daily output of every chunk; the water-balance totals of the chunked run equal
    those of the undivided one limb for limb.  src/SPLASH.cpp:1875-1909 (the state carried by run_all)."""
    import torch
    dev = torch.device("cuda:0")
    torch.cuda.empty_cache()
    H, W, Htot, seed = 512, 4096, 4096, 20221130
    n = H * W
    dates = np.arange(np.datetime64("2001-01-01"), np.datetime64("2004-01-01"))
    nt = len(dates)
    cal = api.Calendar.from_dates(dates)
    doy, year, month = api.time_fields(dates)
    ybounds = [0] + [int(i) + 1 for i in np.where(np.diff(year) != 0)[0]] + [nt]
    nm = cal.nmonth
    f64 = dict(dtype=torch.float64, device=dev)
    cell0 = 3 * n   # the fourth rank's rows (1536..2047: latitudes 15 N .. 0)
    lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
    soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
    api.dev_synth_statics(seed, Htot, W, cell0, n, 1, lat, elev, slop, asp, res, soil, au)
    sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))                    # 3 x 18.4 GB resident
    api.dev_synth_forcing(seed, cal, 0, nt, cell0, n, lat, elev, sw, tc, pn, n)
    par = torch.empty(api.NPAR, n, **f64)
    Tt = torch.empty(n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
    api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
    state0 = torch.empty(api.NSTATE, n, **f64)
    work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
    api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state0, n, work)
    del work
    SAT, RES = par[4], par[11]
    # undivided: one launch over the 1 096 days, two of the five rasters kept (2 x 18 GB)
    st_a = state0.clone()
    acc_a = torch.zeros(1, nm, api.DIAG_REC, dtype=torch.int64, device=dev)
    whole = {k: torch.empty(nt, n, **f64) for k in ("wn", "aet")}
    api.dev_run_all(cal, 0, nt, n, sw, tc, pn, n, Tt, par, n, st_a, n, whole, n, diag_acc=acc_a)
    chk_whole = {k: [_chk(whole[k][t0:t1]) for t0, t1 in zip(ybounds[:-1], ybounds[1:])] for k in whole}
    del whole
    # in calendar-year chunks: one reused year of all five daily outputs, state carried in HBM
    st_b = state0.clone()
    acc_b = torch.zeros_like(acc_a)
    outs = {k: torch.empty(366, n, **f64) for k in ("wn", "snow", "ro", "aet", "pet")}
    chk_chunks = {"wn": [], "aet": []}
    for t0, t1 in zip(ybounds[:-1], ybounds[1:]):
        api.dev_run_all(cal, t0, t1 - t0, n, sw[t0:], tc[t0:], pn[t0:], n, Tt, par, n, st_b, n, outs, n, diag_acc=acc_b)
        for k in outs:
            o = outs[k][:t1 - t0]
            assert torch.isfinite(o).all(), (k, t0)
            if k == "wn":
                assert (o <= SAT[None] * (1 + 1e-15)).all() and (o >= RES[None] * (1 - 1e-15)).all(), t0
            else:
                assert (o >= 0).all(), (k, t0)
        for k in chk_chunks:
            chk_chunks[k].append(_chk(outs[k][:t1 - t0]))
    torch.cuda.synchronize()
    assert chk_chunks == chk_whole, (chk_chunks, chk_whole)
    assert torch.equal(st_a.view(torch.int64), st_b.view(torch.int64))
    assert torch.equal(acc_a, acc_b)
    print(f"config 3, one rank's share: {n} cells x {nt} days, year chunks == undivided launch (checksums {chk_whole['wn']})")
    cal.close()


def test_benchmark_grid_latitude_bands_one_step_and_trajectory(api, oracle):
    """`benchmark-rows-bands`: the 4096 x 4096 grid of the metric (BASELINE.json configs[2]) sampled in eight
    bands of two rows from 60 N to 60 S -- rows 0, 585, 1170, ..., 4094 -- each through the whole device
    pipeline and compared on every cell-day, one-step and trajectory (src/SPLASH.cpp:920-1588,
    src/SOLAR.cpp:149-165).  The 60 N rows of CASES see grazing winter sun and snow; the low latitudes
    are the snow-free, ponding-prone, slower blocks that only builder-run tools had compared."""
    Htot = W = 4096
    rows = 2
    snow_frac = []
    for b in range(8):
        r0 = int(round(b * (Htot - rows) / 7))
        p = device_pipeline(api, 20221130, Htot, W, r0 * W, rows * W, 365, 1)
        r1 = assert_one_step(oracle, p, f"rows {r0}+")
        r2 = assert_trajectory(oracle, p, f"rows {r0}+", n_worst=3)
        lat = 60.0 - 120.0 * (r0 + 1.0) / Htot
        print(f"rows {r0}-{r0 + rows - 1} (lat {lat:+.1f}): one-step max rel err {max(r1['max_rel_err'].values()):.2e}, "
              f"trajectory {max(r2['max_rel_err'].values()):.2e}, set aside {r2['set_aside_cell_days']} cell-days, "
              f"{r2['critical_event_cells_that_differ']} of {r2['critical_event_cells']} critical-event cells differ; worst {r2['worst'][:1]}")
        assert np.isfinite(p["out"]["wn"]).all()
        snow_frac.append(float((p["out"]["snow"] > 0).any(0).mean()))
    print("cells that see snow, per band:", [round(x, 3) for x in snow_frac])
    assert max(snow_frac) > 0.99 and min(snow_frac) < 0.7   # all-snow bands and bands where 40 % of the cells never see any


@pytest.mark.gpu
def test_cycle_detector_ignores_an_idle_snow_age(api, oracle):
    """Round 4: the spin-up's exact-cycle detector (k_spin_phase) no longer needs the snow age nd to repeat
    in a dry year (no record of the 365 has snowfall: nd is never reset) once it has passed 64 days (from
    there on the snow albedo 0.5563 + 0.4437 exp(-0.895 nd) IS 0.5563): wn, snow, qin and td then evolve
    without it, a bit-for-bit repeat of those four is a cycle for good, and nd is advanced by 365 per
    skipped pass.  Two rows at the equator of the benchmark grid (no snow all year,
    0.1 % of the cells in an exact 2-cycle of more than 1 mm): the spun-up state -- nd included --, the
    aridity index, the pass counts and every daily series must equal, bit for bit, what the round-3
    detector (SPLASH_HIP_NO_ND_SKIP=1: compares nd as well, i.e. runs those cells ~1000 passes) leaves;
    the pass counts must also be the oracle's, which iterates all passes (src/SPLASH.cpp:1697-1743)."""
    import os
    Htot = W = 4096
    r0, rows = 2040, 2
    os.environ.pop("SPLASH_HIP_NO_ND_SKIP", None)
    a = device_pipeline(api, 20221130, Htot, W, r0 * W, rows * W, 365, 1, want_passes=True)
    os.environ["SPLASH_HIP_NO_ND_SKIP"] = "1"
    os.environ["SPLASH_HIP_NO_SOLO_TAIL"] = "1"   # and the tail in one packed stage, as in round 3
    try:
        b = device_pipeline(api, 20221130, Htot, W, r0 * W, rows * W, 365, 1, want_passes=True)
    finally:
        os.environ.pop("SPLASH_HIP_NO_ND_SKIP")
        os.environ.pop("SPLASH_HIP_NO_SOLO_TAIL")
    capped = a["passes"][1] >= 1000
    bare = ~(a["out"]["snow"] > 0).any(0) & (a["state0"][1] == 0)   # the case this is about: no snow at all
    assert (capped & bare).sum() >= 5, (capped.sum(), (capped & bare).sum())
    assert (a["state0"][4][capped & bare] >= 365000 - 2 * 365).all()   # nd after 1000 passes of 365 days
    assert np.array_equal(a["passes"], b["passes"])
    for k in ("state0", "state1", "par", "Tt"):
        assert np.array_equal(a[k].view(np.int64), b[k].view(np.int64)), k
    for k in a["out"]:
        assert np.array_equal(a["out"][k].view(np.int64), b["out"][k].view(np.int64)), k
    # the oracle's pass counts for the capped cells and a sample of the others (it really iterates)
    doy, year, month, mid, nm = oracle.calendar(2001, 365)
    sel = np.r_[np.nonzero(capped)[0][:12], np.arange(0, rows * W, 997)]
    st = {k: (v[:, sel] if v.ndim == 2 else v[sel]) for k, v in a["statics"].items()}
    ref, aux = oracle.grid(doy, year, month, mid, nm, np.ascontiguousarray(a["sw"][:, sel]),
                           np.ascontiguousarray(a["tc"][:, sel]), np.ascontiguousarray(a["pn"][:, sel]), st["lat"],
                           st["elev"], st["slop"], st["asp"], st["soil"], st["au"], st["resolution"])
    agree = aux["passes2"].astype(np.int64) == a["passes"][1][sel]
    assert agree.mean() >= 0.98 and agree[:min(12, int(capped.sum()))].all(), (aux["passes2"], a["passes"][1][sel])
    # Round 6: not only the pass counts -- the spin-up's RESULT.  The state handed to run_all
    # (wn, snow, qin within tolerance; td with its NaN pattern; nd EXACTLY: the detector adds 365 per skipped
    # pass) and the aridity index of the capped cells and of the strided sample against the oracle's brute-force
    # 1000-pass loop (src/SPLASH.cpp:1697-1743), each side from its own spin-up
    from oracle.parity import spinup_summary
    r = spinup_summary(oracle, a["state0"][:, sel], a["par"][15][sel], a["passes"][1][sel], aux, st,
                       (doy, year, month, mid, nm, a["sw"][:, sel], a["tc"][:, sel], a["pn"][:, sel]))
    print("spin-up result vs the oracle:", r)
    assert not r["violations"], r["violations"]
    slack = r["critical_event_cells_that_differ"] + r["excused"] + r["grazing_cells"]
    assert r["capped"] >= min(12, int(capped.sum())) and r["capped_within_tolerance"] >= r["capped"] - slack
    assert r["nd_equal"] >= r["cells"] - slack


@pytest.mark.gpu
def test_benchmark_rows_bands_own_spin_up(api, oracle):
    """`benchmark-rows-bands-own-spin-up` (VERDICT r05 #1b): the eight two-row bands of the 4096 x 4096 grid of
    the metric, 60 N to 60 S, through the WHOLE pipeline with each side starting from its OWN spin-up -- the
    product's phased spin-up with exact-cycle detection, the oracle's brute-force loop
    (src/SPLASH.cpp:1594-1749 x 2, R/splash.point.R:141-152) -- where the one-step and trajectory tests above start
    the oracle from the GPU's spun-up state and cannot see a wrong spin-up.  Per band: every series of every cell
    without a critical event of the reference within tolerance on every day (assert_whole_pipeline), the spin-up's
    state, aridity index and pass counts by spinup_summary (capped cells included: 0.1-0.3 % of the cells)."""
    from oracle.parity import spinup_summary
    Htot = W = 4096
    rows, nt = 2, 365
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    names = ("wn", "snow", "ro", "aet", "pet", "cond", "bflow", "netr")
    tot = dict(cells=0, capped=0, capped_ok=0, flips=0, excused=0, set_aside=0, crit=0)
    for b in range(8):
        r0 = int(round(b * (Htot - rows) / 7))
        p = device_pipeline(api, 20221130, Htot, W, r0 * W, rows * W, nt, 1, want_passes=True)
        st = p["statics"]
        ref, aux = oracle.grid(doy, year, month, mid, nm, p["sw"], p["tc"], p["pn"], st["lat"], st["elev"],
                               st["slop"], st["asp"], st["soil"], st["au"], st["resolution"])
        forcing = (doy, year, month, mid, nm, p["sw"], p["tc"], p["pn"])
        rs = spinup_summary(oracle, p["state0"], p["par"][15], p["passes"][1], aux, st, forcing)
        assert not rs["violations"], (r0, rs["violations"])
        rw = assert_whole_pipeline(oracle, p["out"], ref, aux, st, doy, year, f"rows {r0}+ (own spin-up)",
                                   names=names, forcing=forcing)
        ok = rw.pop("ok")
        # the pass counts of spin-up #2 on every cell the comparison kept
        assert np.array_equal(p["passes"][1][ok], aux["passes2"][ok].astype(p["passes"].dtype)), r0
        lat = 60.0 - 120.0 * (r0 + 1.0) / Htot
        print(f"rows {r0}-{r0 + rows - 1} (lat {lat:+.1f}): spin-up {rs['capped']} capped cells "
              f"({rs['capped_within_tolerance']} within tolerance), {rs['critical_event_cells_that_differ']} of "
              f"{rs['critical_event_cells']} critical-event cells differ, {rs['grazing_cells']} cells with a grazing-sun day "
              f"set aside ({rs['grazing_cells_that_differ']} differ), {rs['excused']} excused, max rel state "
              f"{max(rs['max_rel_state'].values()):.2e} (capped {rs['max_rel_state_capped']:.2e}), AI {rs['max_rel_ai']:.2e}; "
              f"whole pipeline max rel err {max(rw['max_rel_err'].values()):.2e}, {rw['set_aside']} set aside, "
              f"{rw['reference_disagrees_with_itself']} excused")
        tot["cells"] += rs["cells"]; tot["capped"] += rs["capped"]; tot["capped_ok"] += rs["capped_within_tolerance"]
        tot["flips"] += rs["critical_event_cells_that_differ"] + rs["grazing_cells"]
        tot["excused"] += rs["excused"] + rw["reference_disagrees_with_itself"]
        tot["set_aside"] += rw["set_aside"]; tot["crit"] += rs["critical_event_cells"]
    print("all bands:", tot)
    assert tot["capped"] >= 40                       # the class exists at this sample size (0.1-0.3 % of 65 536 cells)
    assert tot["capped_ok"] >= tot["capped"] - tot["flips"] - tot["excused"]


def test_bench_line_keeps_the_contract():
    """bench.py prints ONE JSON line with the keys the driver and the judge read"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rows", "64", "--cols", "64",
                        "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "cell-days/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    # the default splits THE grid over the ranks of --gpus N (total work fixed): the label says so at every N
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0
    assert d["value"] > 0 and abs(d["value"] - 64 * 64 * 365 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    # round 4: what the whole-call figure is made of, and a parity sample that says where it looked
    assert d["inclusive"]["tail_seconds"] > 0 and d["inclusive"]["seconds"] >= d["inclusive"]["tail_seconds"]
    ps = d["parity_sample"]
    nb = len(ps["bands"]["row0"])
    assert nb >= 1 and all(len(ps["bands"][k]) == nb for k in ("lat_deg", "set_aside_cell_days", "flips", "max_rel_err",
                                                                "kept_out_of_tolerance"))
    assert ps["n_violations"] == 0, ps["violations"]
    assert ps["excuse_builds"] in ("not needed", "built") or ps["excuse_builds"].startswith("unavailable")
    assert ps["out_of_tolerance_columns"] == ["all", "inside_set_aside", "kept"]
    assert all(v[2] == 0 or ps["cells_on_which_the_reference_disagrees_with_itself"] > 0
               for v in ps["out_of_tolerance"].values())
    # round 6: the record of the driver keeps 8 KB of the line and `config` whole: the whole-call figures sit
    # in config, the parity block is compact (its per-band / per-decade detail goes to stderr), and the spin-up's
    # RESULT -- not only its pass counts -- is compared with the oracle's own brute-force spin-up
    assert len(lines[0]) < 8000, len(lines[0])
    for k in ("value_inclusive", "inclusive_seconds", "inclusive_tail_seconds", "spinup_seconds"):
        assert d["config"][k] > 0, k
    assert abs(d["config"]["value_inclusive"] - d["value_inclusive"]) < 1e-6 * d["value_inclusive"]
    assert any(l.startswith("PARITY_DETAIL ") for l in r.stderr.splitlines())
    sp = d["spinup_parity"]
    assert sp["n_violations"] == 0 and sp["same_bits_as_the_grid_run"] is True, sp
    assert sp["cells"] >= 64 and sp["passes_equal"] >= sp["cells"] - sp["critical_event_cells_that_differ"] - sp["excused"] - sp["grazing_cells_set_aside"]
    assert d["ranks"]["cells"] == [64 * 64] and len(d["ranks"]["run_all_ms_per_step"]) == 1
