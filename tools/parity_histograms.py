"""The whole distribution of the product's error against the oracle (GPU box), for profiles/:

  1. one-step and trajectory comparisons of the four grids of tests/test_gpu_parity.py::CASES
     (every cell-day, all 11 / 5 series);
  2. the whole-pipeline sweep (splash_grid vs orc_grid, each from its own spin-up) over independent
     synthetic grids, histograms merged, and the WORST cell-days that are not set aside, each with
     what the oracle knows about it (lambda, depth, slope sunset angle hs, event bits of the day,
     events of the cell before that day).

Histograms are over ALL cell-days, nothing set aside (oracle/parity.py::error_profile).

    python tools/parity_histograms.py [--grids 24] [--cells 8192] [--days 730] [--out gpurun_out/r03_parity_histograms.md]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import rsplash_amd.api as api
from gpu_pipeline import device_pipeline
from oracle.binding import HS_LIM_DEG, Oracle
from oracle.parity import (OUT5, assert_one_step, assert_whole_pipeline, format_profiles, merge_profiles,
                           trajectory_summary)

CASES = {
    "terrain": (777000, 64, 128, 0, 8192, 730, 1, "01-01"),
    "benchmark-rows-60N": (20221130, 4096, 4096, 0, 4 * 4096, 365, 1, "01-01"),
    "flat": (816595, 32, 128, 0, 4096, 400, 0, "01-01"),
    "starts-in-october": (784919, 32, 128, 0, 4096, 500, 1, "10-01"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grids", type=int, default=24)
    ap.add_argument("--cells", type=int, default=8192)
    ap.add_argument("--days", type=int, default=730)
    ap.add_argument("--worst", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_parity_histograms.md"))
    a = ap.parse_args()
    o = Oracle()
    L = ["# Error histograms of the HIP path against the oracle -- `tools/parity_histograms.py`", "",
         "Counts of cell-days over ALL cell-days (nothing set aside): `bit-equal` (includes matching NaN and "
         "0 == 0), `|d|<=1e-12` (a non-zero difference below the survey's absolute tolerance for exact zeros: "
         "transmittances of 1e-30 mm/day live here), then the rest by decade of |x - ref|/|ref|.  `beyond r` = "
         "|x - ref| > r|ref| + 1e-12 (the levels `assert_bulk` holds); `ref=0` = cell-days whose reference value "
         "is exactly 0, of which `|x|>1e-12` have a product value beyond 1e-12; `only atol 1e-9` = pass the "
         "tolerance 1e-5|ref| + 1e-9 but not 1e-5|ref| + 1e-12; `only floor` = pass only through the series' "
         "floor (aet: 1e-7 pet, ro: 1e-10 wn); `out of tol.` = fail 1e-5|ref| + 1e-9 + floor (cells set aside by "
         "the tests are INCLUDED here).", ""]
    one, tra = {}, {}
    for case, (seed, Htot, W, cell0, n, nt, terrain, start) in CASES.items():
        p = device_pipeline(api, seed, Htot, W, cell0, n, nt, terrain, start=start)
        r1 = assert_one_step(o, p, case, bulk=None)
        r2 = trajectory_summary(o, p, bulk=None)
        one[case], tra[case] = r1["profile"], r2["profile"]
        print(case, "done", flush=True)
    ser11 = list(next(iter(one.values())))
    L.append(format_profiles({k: merge_profiles([one[c][k] for c in one]) for k in ser11},
                             "## One-step (the oracle redoes every day from the product's previous-day state): 4 test grids merged"))
    L.append("")
    L.append(format_profiles({k: merge_profiles([tra[c][k] for c in tra]) for k in OUT5},
                             "## Trajectory (the oracle carries its own state through run_all): 4 test grids merged"))
    L.append("")
    for case in CASES:
        L.append(format_profiles(one[case], f"### one-step, `{case}`"))
        L.append("")
        L.append(format_profiles(tra[case], f"### trajectory, `{case}`"))
        L.append("")

    # ---- whole pipeline sweep
    W = 128
    H = a.cells // W
    nt = a.days
    doy, year, month, mid, nm = o.calendar(2001, nt)
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    names = ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "sm_lim")
    profs, rows = [], []
    for g in range(a.grids):
        seed = 777000 + 7919 * g
        terrain = 0 if g == a.grids - 1 else 1
        st = o.synth_statics(seed, H, W, 0, H * W, terrain)
        sw, tc, pn = o.synth_forcing(seed, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
        ref, aux = o.grid(doy, year, month, mid, nm, sw, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"],
                          st["soil"], st["au"], st["resolution"])
        got = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                              st["resolution"], time_index=dates, sim_control=dict(monthly_out=False))
        r = assert_whole_pipeline(o, got, ref, aux, st, doy, year, f"grid {g}", names=names,
                                  forcing=(doy, year, month, mid, nm, sw, tc, pn), bulk=None, n_worst=a.worst)
        profs.append(r["profile"])
        for w in r["worst"]:
            c = w["cell"]
            sl = slice(c, c + 1)
            si = np.ascontiguousarray(aux["soil_info"][sl].T)
            _, _, ev, hs = o.grid_transient2(doy, year, month, sw[:, sl].copy(), tc[:, sl].copy(), pn[:, sl].copy(),
                                             st["lat"][sl], st["elev"][sl], st["slop"][sl], st["asp"][sl] - 180.0, si,
                                             aux["Tt"][sl].copy(), np.ascontiguousarray(aux["state0"][sl].T),
                                             names=("wn",), want_hs=True, n_threads=1)
            t = w["day"]
            w.update(grid=g, lam=float(si[4, 0]), depth=float(si[5, 0]), hs=float(hs[t, 0]), ev_day=int(ev[t, 0]),
                     ev_before=int(np.bitwise_or.reduce(ev[:t + 1, 0])), min_hs_before=float(hs[:t + 1, 0][hs[:t + 1, 0] > 0].min()),
                     critical_in_spinup=int(aux["critical"][c]), passes2=int(aux["passes2"][c]))
            rows.append(w)
        print("grid", g, "done", flush=True)
    rows.sort(key=lambda r: -r["rel"])
    L.append(format_profiles({k: merge_profiles([p[k] for p in profs]) for k in names},
                             f"## Whole pipeline (splash_grid vs the oracle's splash.grid, each from its own spin-up): "
                             f"{a.grids} grids x {H * W} cells x {nt} days merged"))
    L += ["", f"## The {a.worst} worst cell-days of the sweep that are NOT set aside", "",
          f"(set aside: cells that took the other side of a critical event, cell-days at/after a grazing-sun day "
          f"hs < {HS_LIM_DEG:g} deg; event bits of the oracle's run: 1 noise-level denominator, 2 grazing sun, "
          f"4 overflow residual, 8 critical denominator)", "",
          "| grid | cell | day | series | product | oracle | rel. diff | lambda | depth | hs that day [deg] | smallest hs so far | events that day | events so far | critical events in spin-up | passes #2 |",
          "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for w in rows[:a.worst]:
        L.append(f"| {w['grid']} | {w['cell']} | {w['day']} | {w['series']} | {w['product']:.12g} | {w['ref']:.12g} | "
                 f"{w['rel']:.2e} | {w['lam']:.4g} | {w['depth']:.3g} | {w['hs']:.4g} | {w['min_hs_before']:.3g} | "
                 f"{w['ev_day']} | {w['ev_before']} | {w['critical_in_spinup']} | {w['passes2']} |")
    txt = "\n".join(L) + "\n"
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    open(a.out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
