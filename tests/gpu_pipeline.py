"""Helpers of the GPU parity tests (not a test module): run the device-resident pipeline on
synthetic grids and compare it with the oracle in the two ways DESIGN.md "Parity" describes.

  one-step     every cell-day is an independent one-day transition: the oracle starts day t from the
               GPU's own state after day t-1, so a disagreement cannot propagate.  Only the cell-days
               on which the reference's own result is rounding noise (event bits of
               orc_grid_transient2; evidence in profiles/r02_conditioning.md) are set aside.
  trajectory   the oracle carries its own state (what SPLASH::run_all does).  Strict except on a
               grazing-sun day and from a cell's critical event on (a sticky infinite drainage time
               the reference itself takes or not depending on the last bit of pow()); at most 5 % of
               the critical-event cells may actually differ, and >= 1 - 1e-5 of ALL cell-days must
               be within tolerance with nothing set aside.
"""
import numpy as np

from oracle.parity import (ATOL, OUT5, SERIES11, assert_one_step, assert_trajectory,  # noqa: F401
                           assert_whole_pipeline, one_step_report, trajectory_report)


RTOL = 1e-5     # north_star tolerance (shadows oracle.parity's of the same value: one name for the test modules)


def close(x, ref, name="", rtol=RTOL):
    from conftest import rel_err
    rel, ab = rel_err(x, ref)
    assert rel <= rtol, f"{name}: rel {rel:.3e} (max abs err {ab:.3e})"
    return rel


def synth_grid(oracle, H, W, nt, terrain, seed=20221130):
    """the oracle's counter-based generator: statics + forcing of an H x W grid from 2001-01-01"""
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    st = oracle.synth_statics(seed, H, W, 0, H * W, terrain)
    sw, tc, pn = oracle.synth_forcing(seed, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    return dict(doy=doy, year=year, month=month, mid=mid, nm=nm, st=st, sw=sw, tc=tc, pn=pn,
                dates=dates)


def device_pipeline(api, seed, Htot, W, cell0, n, nt, terrain, year0=2001, start="01-01",
                    want_passes=False):
    """soil_hydro -> snow threshold -> both spin-ups -> run_all (all 11 daily series) for cells
    [cell0, cell0+n) of the Htot x W synthetic grid, everything generated and kept on cuda:0.
    Returns numpy copies."""
    import torch
    dev = torch.device("cuda:0")
    dates = np.datetime64(f"{year0}-{start}") + np.arange(nt)
    cal = api.Calendar.from_dates(dates)
    doy, year, month = api.time_fields(dates)
    f64 = dict(dtype=torch.float64, device=dev)
    lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
    soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
    api.dev_synth_statics(seed, Htot, W, cell0, n, terrain, lat, elev, slop, asp, res, soil, au)
    sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
    api.dev_synth_forcing(seed, cal, 0, nt, cell0, n, lat, elev, sw, tc, pn, n)
    par = torch.empty(api.NPAR, n, **f64)
    Tt = torch.empty(n, **f64)
    state = torch.empty(api.NSTATE, n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
    api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
    work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
    passes = torch.zeros(2, n, dtype=torch.int32, device=dev) if want_passes else None
    api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work, passes=passes)
    state0 = state.clone()
    g = {k: torch.empty(nt, n, **f64) for k in SERIES11}
    api.dev_run_all(cal, 0, nt, n, sw, tc, pn, n, Tt, par, n, state, n, g, n)
    torch.cuda.synchronize()
    r = dict(doy=doy, year=year, month=month, dates=dates,
             sw=sw.cpu().numpy(), tc=tc.cpu().numpy(), pn=pn.cpu().numpy(), par=par.cpu().numpy(),
             Tt=Tt.cpu().numpy(), state0=state0.cpu().numpy(), state1=state.cpu().numpy(),
             out={k: v.cpu().numpy() for k, v in g.items()},
             statics=dict(lat=lat.cpu().numpy(), elev=elev.cpu().numpy(), slop=slop.cpu().numpy(),
                          asp=asp.cpu().numpy(), resolution=res.cpu().numpy(),
                          soil=soil.cpu().numpy(), au=au.cpu().numpy()))
    if want_passes:
        r["passes"] = passes.cpu().numpy()
    cal.close()
    return r
