"""Comparison rules of the parity checks (TEST INFRASTRUCTURE ONLY, like everything in oracle/:
used by tests/, __graft_entry__.smoke() and the cpu_baseline/parity leg of bench.py).

Inputs are dicts `p` as tests/gpu_pipeline.py::device_pipeline builds them (numpy arrays: calendar,
forcing, cellpar, Tt, spun-up state0 and the product's daily series) -- see DESIGN.md "Parity":

  one-step     every cell-day is an independent one-day transition: the oracle starts day t from the
               product's own state after day t-1, so a disagreement cannot propagate.  Only the
               cell-days on which the reference's own result is rounding noise (event bits of
               orc_grid_transient2; evidence in profiles/r02_conditioning.md) are set aside.
  trajectory   the oracle carries its own state (what SPLASH::run_all does).  Strict except from a
               cell's grazing-sun day on and from a cell's critical event on (a sticky infinite
               drainage time the reference itself takes or not depending on the last bit of pow());
               at most 5 % of the critical-event cells may actually differ, and >= 1 - 1e-5 of ALL
               cell-days must be within tolerance with nothing set aside.
"""
import os

import numpy as np

RTOL, ATOL = 1e-5, 1e-9
OUT5 = ("wn", "snow", "ro", "aet", "pet")
SERIES11 = ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev", "tdrain", "snwage")
EV_FRAGILE, EV_GRAZING, EV_RESIDUAL, EV_CRITICAL = 1, 2, 4, 8


def _bad(a, b, rtol=RTOL, atol=ATOL, floor=None):
    """|a - b| > rtol*|b| + atol (+ floor, an array: see aet_floor)"""
    with np.errstate(invalid="ignore"):
        d = np.abs(a - b)
        lim = rtol * np.abs(b) + atol
        if floor is not None:
            lim = lim + np.nan_to_num(floor, nan=0.0, posinf=0.0)
        out = ~(d <= lim)
    out &= ~((a == b) | (np.isnan(a) & np.isnan(b)))  # equal infinities / matching NaN are fine
    return out


def aet_floor(pet_ref):
    """Absolute floor of the aet comparison.  aet is an integral of the day's evaporative supply and
    demand minus the snowmelt energy, clamped at 0 (src/EVAP.cpp:254-263); when it cancels to ~0
    its rounding error is that of its terms, which are of the size of pet.  1e-7*pet keeps the
    comparison two orders tighter than the tolerance relative to those terms."""
    return 1e-7 * np.abs(pet_ref)


def ro_floor(wn_ref):
    """Absolute floor of the runoff comparison.  Dunne runoff is the bucket's overflow, ro_d = sm - SAT
    (src/SPLASH.cpp:1383-1386): a difference of two numbers of the size of the column water, so its
    absolute error is that of the bucket.  1e-10*wn (5e-8 mm for a 500 mm column) is five orders
    tighter than the bucket's own tolerance and still far below any runoff that matters."""
    return 1e-10 * np.abs(wn_ref)


def _floor(k, ref):
    if k == "aet" and "pet" in ref:
        return aet_floor(ref["pet"])
    if k == "ro" and "wn" in ref:
        return ro_floor(ref["wn"])
    return None


# ---- the whole distribution of the error, not just its tail ---------------------------------------
# The tolerance of north_star (1e-5) is four to seven decades wider than what the product really
# delivers; a change that moved the BULK of cell-days from 1e-12 to 5e-6 would pass every `_bad`
# test.  error_profile() therefore records, per series and over ALL cell-days handed in (nothing set
# aside), the histogram of |x - ref|/|ref| by decade, what happens where the reference is exactly 0,
# and how many cell-days pass ONLY through an absolute floor; assert_bulk() holds the bulk to levels
# read off those histograms (profiles/r03_parity_histograms.md) with a decade of margin.
# A cell (or cell-day) out of tolerance WITHOUT an event bit is excused only if two other legal builds
# of the reference's own statements, run on it, leave the tolerance too -- and only this many per
# comparison (round 2 allowed 16; its 24-grid sweep excused at most one cell per grid, five in all).
MAX_SELF_DISAGREEMENTS = 4

DECADES = tuple(10.0 ** e for e in range(-15, -4))  # 1e-15 ... 1e-5 (upper edges; last bin: > 1e-5)
ATOL_STRICT = 1e-12                                   # exact zeros within 1e-12
# (fraction, rtol) pairs: at least `fraction` of all cell-days within rtol*|ref| + ATOL_STRICT.
# Trajectory / whole-pipeline comparisons carry the state, so an ulp of one day is amplified by the
# bucket's own dynamics over the following year; one-step comparisons restart every day.
# Levels (profiles/r03_parity_histograms.md, a decade above what was measured): one-step, 4 grids, 1.6e7
# cell-days: at most 5.2e-5 of a series' cell-days beyond 1e-9 (tdrain; aet 2.4e-5, pet 1.5e-5: low sun),
# 2.6e-6 beyond 1e-7; trajectory: 2.6e-5 (wn) and 2.5e-6 (aet).  Whole pipeline (each side from its own
# spin-up, 24 grids, 1.4e8 cell-days, cells set aside for a flipped critical event left out): 7e-5 and
# 4e-5 -- a spin-up that ends 1e-6 mm apart stays apart for the whole run of that cell, so those
# levels are counted in cells: `min_cells` cells' worth of cell-days may always differ (a soak over 18
# unseen grids met one trajectory cell that stayed 1e-7 off for 550 days: profiles/r03_parity_soak.txt).
BULK_ONE_STEP = dict(levels=((1.0 - 5e-4, 1e-9), (1.0 - 5e-5, 1e-7)), min_cells=0)
BULK_TRAJECTORY = dict(levels=((1.0 - 5e-4, 1e-9), (1.0 - 5e-5, 1e-7)), min_cells=2)
BULK_PIPELINE = dict(levels=((1.0 - 2e-3, 1e-9), (1.0 - 1e-3, 1e-7)), min_cells=2)


def error_profile(a, f, floor=None):
    """a, f: product and reference arrays of one series (any shape).  Returns a dict of plain ints/floats:
      n            cell-days compared
      exact        bit-for-bit equal values (NaN == NaN and equal infinities count as exact)
      tiny         0 < |a-f| <= 1e-12 absolute (SURVEY.md 8d's tolerance for exact zeros): neither exact nor
                   in the decades below -- a relative error of a value of 1e-30 mm/day says nothing
      hist         the rest, by decade of |a-f|/|f|: hist[i] = count with error <= DECADES[i] (and >
                   DECADES[i-1]); hist[-1]: > 1e-5 (a reference value of exactly 0 with |a| > 1e-12 lands here)
      ref_zero     cell-days with f == 0;  ref_zero_over_1e_12: of which |a| > 1e-12
      beyond_1e_9, beyond_1e_7   cell-days with |a-f| > r|f| + 1e-12 for r = 1e-9, 1e-7 (the bulk levels)
      only_atol    pass with atol 1e-9 but not with 1e-12 (|d| <= rtol|f| + 1e-9, > rtol|f| + 1e-12)
      only_floor   pass only through the series' floor (aet: 1e-7 pet, ro: 1e-10 wn)
      out_of_tol   fail the parity tolerance (floor included)"""
    rows = int(np.shape(a)[0]) if np.ndim(a) == 2 else 1  # cell-days per cell ([nt, ncell] arrays)
    a = np.asarray(a, dtype=np.float64).ravel()
    f = np.asarray(f, dtype=np.float64).ravel()
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        same = (a == f) | (np.isnan(a) & np.isnan(f))
        d = np.abs(a - f)
        d = np.where(same, 0.0, np.where(np.isnan(d), np.inf, d))
        tiny = ~same & (d <= ATOL_STRICT)
        rest = ~same & ~tiny
        rel = d[rest] / np.abs(f[rest])
        hist = np.bincount(np.searchsorted(np.array(DECADES), np.nan_to_num(rel, nan=np.inf), side="left"),
                           minlength=len(DECADES) + 1)
        af = np.where(np.isnan(f), 0.0, np.abs(f))
        lim0 = RTOL * af
        p_strict = d <= lim0 + ATOL_STRICT
        p_atol = d <= lim0 + ATOL
        fl = np.nan_to_num(np.asarray(floor, dtype=np.float64).ravel(), nan=0.0, posinf=0.0) if floor is not None else 0.0
        p_floor = d <= lim0 + ATOL + fl
        beyond = {r: int((d > r * af + ATOL_STRICT).sum()) for r in (1e-9, 1e-7)}
    fz = f == 0.0
    return dict(n=int(a.size), rows_per_cell=rows, exact=int(same.sum()), tiny=int(tiny.sum()), hist=[int(x) for x in hist],
                ref_zero=int(fz.sum()), ref_zero_over_1e_12=int((np.abs(a[fz]) > ATOL_STRICT).sum()),
                beyond_1e_9=beyond[1e-9], beyond_1e_7=beyond[1e-7],
                only_atol=int((p_atol & ~p_strict).sum()), only_floor=int((p_floor & ~p_atol).sum()),
                out_of_tol=int((~p_floor).sum()))


def merge_profiles(ps):
    """sum of error_profile dicts (several grids)"""
    out = None
    for p in ps:
        if out is None:
            out = {k: (list(v) if isinstance(v, list) else v) for k, v in p.items()}
            continue
        for k, v in p.items():
            if k == "hist":
                out[k] = [x + y for x, y in zip(out[k], v)]
            elif k == "rows_per_cell":
                out[k] = max(out[k], v)
            else:
                out[k] += v
    return out


def bulk_violations(profiles, bulk, name=""):
    """profiles: {series: error_profile}.  bulk: BULK_ONE_STEP / BULK_TRAJECTORY / BULK_PIPELINE."""
    viol = []
    for k, p in profiles.items():
        for frac, rt in bulk["levels"]:
            beyond = p["beyond_1e_9" if rt == 1e-9 else "beyond_1e_7"]
            allowed = max((1.0 - frac) * p["n"], bulk["min_cells"] * p["rows_per_cell"])
            if beyond > allowed:
                viol.append(f"{name}{k}: {beyond} of {p['n']} cell-days beyond {rt:g}|ref| + {ATOL_STRICT:g} "
                            f"(the bulk allows {allowed:.0f})")
    return viol


def format_profiles(profiles, title=""):
    """markdown table of error_profile dicts, one row per series"""
    head = ["series", "cell-days", "bit-equal", "|d|<=1e-12"] + [f"<={e:.0e}" for e in DECADES] + \
        [">1e-5", "beyond 1e-9", "beyond 1e-7", "ref=0", "ref=0: |x|>1e-12", "only atol 1e-9", "only floor", "out of tol."]
    L = ([title, ""] if title else []) + ["| " + " | ".join(head) + " |", "|" + "---|" * len(head)]
    for k, p in profiles.items():
        L.append("| " + " | ".join([k, str(p["n"]), str(p["exact"]), str(p["tiny"])] + [str(x) for x in p["hist"]] +
                                   [str(p["beyond_1e_9"]), str(p["beyond_1e_7"]), str(p["ref_zero"]),
                                    str(p["ref_zero_over_1e_12"]), str(p["only_atol"]), str(p["only_floor"]),
                                    str(p["out_of_tol"])]) + " |")
    return "\n".join(L)


def worst_cell_days(a, ref, keep, n=20, events=None):
    """the n cell-days with the largest relative error among those `keep` marks (not set aside):
    list of dicts(series, day, cell, product, ref, rel, events)"""
    rows = []
    for k in a:
        x, f = np.asarray(a[k]), ref[k]
        with np.errstate(invalid="ignore", divide="ignore"):
            e = np.where(keep & np.isfinite(f) & (f != 0), np.abs(x - f) / np.abs(f), 0.0)
            fl = _floor(k, ref)
            lim = ATOL if fl is None else ATOL + np.nan_to_num(fl, nan=0.0, posinf=0.0)
            e = np.where(np.abs(x - f) <= lim, 0.0, e)  # differences below the absolute floors are not errors
        e = np.nan_to_num(e, nan=0.0, posinf=0.0)
        flat = e.ravel()
        m = min(n, flat.size)
        for i in np.argpartition(-flat, m - 1)[:m]:
            if flat[i] > 0:
                t, c = np.unravel_index(i, e.shape)
                rows.append(dict(series=k, day=int(t), cell=int(c), product=float(x[t, c]), ref=float(f[t, c]),
                                 rel=float(flat[i]), events=int(events[t, c]) if events is not None else None))
    rows.sort(key=lambda r: -r["rel"])
    return rows[:n]


def one_step_report(oracle, p, n_threads=0):
    """p: result of device_pipeline.  Every cell-day of the GPU's run re-done by the oracle from the
    GPU's previous-day state.  Returns dict(events, bad={series: bool [nt, n]}, ref)."""
    g = p["out"]
    prev = np.stack([g["wn"], g["snow"], g["qin_prev"], g["tdrain"], g["snwage"]])
    hp = p["par"]
    ref, _, ev, _ = oracle.grid_transient2(p["doy"], p["year"], p["month"], p["sw"], p["tc"], p["pn"],
                                           hp[0], hp[1], hp[2], hp[3], hp[4:16], p["Tt"], p["state0"],
                                           prev=prev, n_threads=n_threads)
    bad = {k: _bad(g[k], ref[k], floor=_floor(k, ref)) for k in SERIES11}
    return dict(events=ev, bad=bad, ref=ref)


def assert_one_step(oracle, p, name="", max_event_frac=5e-3, n_threads=0, bulk=BULK_ONE_STEP):
    """asserts the one-step parity of a device_pipeline result; returns a small summary dict
    (with 'profile': error_profile per series over ALL cell-days, and the bulk held to `bulk`)"""
    r = one_step_report(oracle, p, n_threads)
    ev, bad = r["events"], r["bad"]
    prof = {k: error_profile(p["out"][k], r["ref"][k], _floor(k, r["ref"])) for k in SERIES11}
    # tdrain on a noise-denominator day is noise in the reference itself (below): not part of its bulk
    td_ok = (ev & (EV_FRAGILE | EV_CRITICAL | EV_RESIDUAL)) == 0
    prof["tdrain"] = error_profile(p["out"]["tdrain"][td_ok], r["ref"]["tdrain"][td_ok])
    bv = bulk_violations(prof, bulk, name + ": one-step ") if bulk else []
    assert not bv, "; ".join(bv)
    graz = (ev & EV_GRAZING) != 0
    # the drainage time of a noise-denominator day is noise in the reference itself; it is an
    # OUTPUT of that day only (tdrain), the other series do not depend on it within the day
    td_noise = (ev & (EV_FRAGILE | EV_CRITICAL | EV_RESIDUAL)) != 0
    resid = (ev & EV_RESIDUAL) != 0
    n_cd = ev.size
    assert graz.mean() <= max_event_frac and td_noise.mean() <= max_event_frac
    worst = {}
    self_dis = 0
    for k in SERIES11:
        mask = graz | resid | (td_noise if k == "tdrain" else False)
        b = bad[k] & ~mask
        if b.any() and b.sum() <= MAX_SELF_DISAGREEMENTS:
            # no event bit names these cell-days: does the reference agree with ITSELF on them?  Its
            # two other builds redo the day from the same previous state (reference_variants()).
            prev = np.stack([p["out"][s_] for s_ in ("wn", "snow", "qin_prev", "tdrain", "snwage")])
            hp = p["par"]
            for t, c in np.argwhere(b):
                sl = slice(int(c), int(c) + 1)
                for oo in reference_variants().values():
                    r2 = oo.grid_transient2(p["doy"], p["year"], p["month"], p["sw"][:, sl].copy(),
                                            p["tc"][:, sl].copy(), p["pn"][:, sl].copy(), hp[0, sl], hp[1, sl],
                                            hp[2, sl], hp[3, sl], hp[4:16, sl].copy(), p["Tt"][sl],
                                            p["state0"][:, sl].copy(), prev=prev[:, :, sl].copy(), names=(k,),
                                            n_threads=1)[0]
                    if _bad(r2[k][t:t + 1, 0], r["ref"][k][t:t + 1, c]).any():
                        b[t, c] = False
                        self_dis += 1
                        break
        if b.any():
            t, c = np.argwhere(b)[0]
            raise AssertionError(
                f"{name}: one-step {k} out of tolerance on {int(b.sum())} of {n_cd} cell-days outside the "
                f"event mask; first: day {t} cell {c}: gpu {p['out'][k][t, c]!r} ref {r['ref'][k][t, c]!r} "
                f"events {ev[t, c]}")
        a, f = p["out"][k], r["ref"][k]
        with np.errstate(invalid="ignore", divide="ignore"):
            e = np.where(mask | ~np.isfinite(f), 0.0,
                         np.maximum(np.abs(a - f) - ATOL, 0.0) / np.maximum(np.abs(f), 1e-300))
        worst[k] = float(np.nanmax(e))
    return dict(cell_days=n_cd, grazing=int(graz.sum()), noise_denominator=int(td_noise.sum()),
                reference_disagrees_with_itself=self_dis, max_rel_err=worst, profile=prof)


def trajectory_report(oracle, p, n_threads=0):
    hp = p["par"]
    ref, st1, ev, _ = oracle.grid_transient2(p["doy"], p["year"], p["month"], p["sw"], p["tc"], p["pn"],
                                             hp[0], hp[1], hp[2], hp[3], hp[4:16], p["Tt"], p["state0"],
                                             n_threads=n_threads)
    bad = {k: _bad(p["out"][k], ref[k]) for k in OUT5}
    return dict(events=ev, bad=bad, ref=ref, state1=st1)


def trajectory_summary(oracle, p, min_frac_all=1.0 - 1e-5, max_flip_frac=0.05, n_threads=0, names=OUT5,
                       bulk=BULK_TRAJECTORY, n_worst=0):
    """Strict for every cell-day that is not at/after a conditioning event of its cell:
      * a grazing-sun day (the reference's radiation of that day is rounding noise, and the day
        leaves a slightly different bucket behind: the cell is set aside from that day on);
      * a critical event (the reference takes a sticky infinite drainage time or not depending on the
        last bit of pow(), profiles/r02_conditioning.md): such cells MAY differ from that day on,
        but only `max_flip_frac` of them do.
    And `min_frac_all` of all cell-days, nothing set aside but the cells that took the other side of a
    critical event (counted, bounded by `max_flip_frac`), must be within tolerance.
    Returns the figures plus 'violations' (list of messages)."""
    hp = p["par"]
    want = tuple(dict.fromkeys(tuple(names) + ("pet", "wn")))
    ref, st1, ev, hs = oracle.grid_transient2(p["doy"], p["year"], p["month"], p["sw"], p["tc"], p["pn"],
                                              hp[0], hp[1], hp[2], hp[3], hp[4:16], p["Tt"], p["state0"],
                                              names=want, n_threads=n_threads, want_hs=bool(n_worst))
    graz_after = np.cumsum((ev & EV_GRAZING) != 0, axis=0) > 0
    crit_after = np.cumsum((ev & (EV_CRITICAL | EV_RESIDUAL)) != 0, axis=0) > 0
    flagged = crit_after.any(0)
    frac_all, frac_kept, worst, viol = {}, {}, {}, []
    self_dis = set()
    bads = {k: _bad(p["out"][k], ref[k], floor=_floor(k, ref)) for k in names}
    flipped = np.zeros(flagged.shape, bool)
    for k in names:
        flipped |= (bads[k] & crit_after & ~graz_after).any(0)
    for k in names:
        b = bads[k]
        # of ALL cell-days, nothing set aside (reported); the floor is asserted on the cells that did
        # not take the other side of a critical event -- those are counted, and bounded, below
        frac_all[k] = 1.0 - float(b.mean())
        frac_kept[k] = 1.0 - float(b[:, ~flipped].mean()) if (~flipped).any() else 1.0
        if frac_kept[k] < min_frac_all:
            viol.append(f"only {frac_kept[k]:.8f} of the cell-days of cells without a flipped critical event "
                        f"within tolerance for {k}")
        bm = b & ~graz_after & ~crit_after
        if bm.any() and np.unique(np.argwhere(bm)[:, 1]).size <= MAX_SELF_DISAGREEMENTS:
            # cells that leave the tolerance without an event: does the reference's own trajectory
            # from the same state survive another legal build (reference_variants())?
            for c in np.unique(np.argwhere(bm)[:, 1]):
                sl = slice(int(c), int(c) + 1)
                for oo in reference_variants().values():
                    r2 = oo.grid_transient2(p["doy"], p["year"], p["month"], p["sw"][:, sl].copy(),
                                            p["tc"][:, sl].copy(), p["pn"][:, sl].copy(), hp[0, sl], hp[1, sl],
                                            hp[2, sl], hp[3, sl], hp[4:16, sl].copy(), p["Tt"][sl],
                                            p["state0"][:, sl].copy(), names=(k,), n_threads=1)[0]
                    if _bad(r2[k][:, 0], ref[k][:, c]).any():
                        bm[:, c] = False
                        self_dis.add(int(c))
                        break
        if bm.any():
            t, c = np.argwhere(bm)[0]
            viol.append(f"{k} out of tolerance on {int(bm.sum())} cell-days outside the mask; first: day {t} "
                        f"cell {c}: product {p['out'][k][t, c]!r} ref {ref[k][t, c]!r}")
        if not np.isfinite(p["out"][k]).all():
            viol.append(f"{k} has non-finite values")
    keep = ~graz_after & ~(crit_after & flipped[None, :])
    for k in names:
        a, f = p["out"][k], ref[k]
        e = np.where(keep, np.maximum(np.abs(a - f) - ATOL, 0.0) / np.maximum(np.abs(f), 1e-300), 0.0)
        fl = _floor(k, ref)
        if fl is not None:
            e = np.where(np.abs(a - f) <= fl + ATOL, 0.0, e)
        worst[k] = float(e.max())
    if flipped.sum() > max(3, max_flip_frac * flagged.sum()):
        viol.append(f"{int(flipped.sum())} of {int(flagged.sum())} critical-event cells differ")
    prof = {k: error_profile(p["out"][k], ref[k], _floor(k, ref)) for k in names}
    if bulk:
        viol += bulk_violations(prof, bulk, "trajectory ")
    worst_rows = worst_cell_days({k: p["out"][k] for k in names}, ref, keep, n_worst, events=ev) if n_worst else []
    for r in worst_rows:  # what the explanation of a worst row needs: the sun's height and the soil's pore-size index
        r["hs_deg"] = float(hs[r["day"], r["cell"]])
        r["lambda"] = float(hp[8, r["cell"]])
        r["depth_m"] = float(hp[9, r["cell"]])
    set_aside = int((~keep).sum())
    # of each series' cell-days out of tolerance, how many lie inside the set-aside mask (the rest would
    # have raised a violation above unless another build of the reference excused their cell)
    oot = {k: dict(all=int(bads[k].sum()), inside_set_aside=int((bads[k] & ~keep).sum()),
                   kept=int((bads[k] & keep).sum())) for k in names}
    if set_aside > 5e-3 * ev.size:
        viol.append(f"{set_aside} of {ev.size} cell-days set aside")
    return dict(cell_days=int(ev.size), tolerance=RTOL, set_aside_cell_days=set_aside,
                grazing_cell_days=int(((ev & EV_GRAZING) != 0).sum()),
                cells_after_a_grazing_day=int(graz_after.any(0).sum()),
                critical_event_cells=int(flagged.sum()),
                critical_event_cells_that_differ=int(flipped.sum()),
                cells_on_which_the_reference_disagrees_with_itself=len(self_dis),
                fraction_of_all_cell_days_within_tolerance=frac_all,
                fraction_within_tolerance_without_flipped_cells=frac_kept, max_rel_err=worst,
                max_abs_dwn_mm=float(np.abs(p["out"]["wn"] - ref["wn"])[keep].max()), violations=viol,
                out_of_tolerance=oot, excuse_builds=variants_status(), profile=prof, worst=worst_rows)


def assert_trajectory(oracle, p, name="", **kw):
    r = trajectory_summary(oracle, p, **kw)
    assert not r["violations"], f"{name}: " + "; ".join(r["violations"])
    return r


_VARIANTS = None
_VARIANTS_STATUS = "not needed"


def variants_status():
    """'not needed' (no stray cell asked for them), 'built' or 'unavailable: <why>'"""
    return _VARIANTS_STATUS


def reference_variants():
    """The oracle's own source in two further legal builds (tools/conditioning_evidence.py):
    B  -O3 -march=x86-64-v3 -ffp-contract=fast;  C  every libm result moved by -1/0/+1 ulp.
    Built on first use into the temp directory (gcc is on the build container and the GPU box).
    A host without gcc gets an EMPTY dict, not an exception: no stray cell can then be excused, so a
    comparison that needed the excuse reports a violation -- a verdict ('excuse builds unavailable' in
    variants_status() and in the summaries) instead of a crash of the checker."""
    global _VARIANTS, _VARIANTS_STATUS
    if _VARIANTS is None:
        import shutil
        import subprocess
        import tempfile
        from .binding import Oracle
        src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "splash_oracle.c")
        out = {}
        cc = os.environ.get("SPLASH_ORACLE_CC", "gcc")
        if shutil.which(cc) is None:
            _VARIANTS, _VARIANTS_STATUS = {}, f"unavailable: no `{cc}` on this host"
            return _VARIANTS
        try:
            for tag, flags in (("B", ["-O3", "-march=x86-64-v3", "-ffp-contract=fast"]),
                               ("C", ["-O2", "-DORC_PERTURB_LIBM"])):
                so = os.path.join(tempfile.gettempdir(), f"liboracle_parity_{tag}_{os.getpid()}.so")
                subprocess.check_call([cc, "-std=c11", "-fPIC", "-fopenmp", "-shared", "-Wl,-Bsymbolic"] + flags +
                                      ["-o", so, src, "-lm"])
                out[tag] = Oracle(so)
            _VARIANTS, _VARIANTS_STATUS = out, "built"
        except (OSError, subprocess.CalledProcessError) as e:
            _VARIANTS, _VARIANTS_STATUS = {}, f"unavailable: {e}"
    return _VARIANTS


def input_ulp_variants(st1):
    """The statics of ONE cell with one input moved by one unit in the last place, up or down: the six soil inputs and
    the elevation (14 variants).  The synthetic generator itself exists in two builds whose doubles differ in the last
    place on one cell in 1e4 (host C and device, SURVEY.md 8d: 'agree to <= 1 ulp'), and no measured soil or DEM is
    known to an ulp: a cell whose reference result leaves the tolerance under such a step -- a drainage time that
    reaches 0 by a hair (`td <= 0`, src/SPLASH.cpp:1367-1371, after the cancelling `td -= 1`, :1468), a bucket that
    lands on its bound -- has no reference value to be compared with (class 4).  Found in round 6: cell
    (1924, 254) of the benchmark grid, where one ulp of the bulk density moves the oracle's aridity index by 1.7e-4
    (profiles/r06_spinup_parity_sweep2.txt, tools/debug/r06_cell_1924_254.py)."""
    for key, idx in [("soil", k) for k in range(6)] + [("elev", None)]:
        for direction in (-np.inf, np.inf):
            v = {k: np.array(x, dtype=np.float64, copy=True) for k, x in st1.items()}
            a = v[key]
            if idx is None:
                if not np.isfinite(a).all():
                    continue
                a[...] = np.nextafter(a, direction)
            else:
                if not np.isfinite(a[idx]).all():
                    continue
                a[idx] = np.nextafter(a[idx], direction)
            yield v


def reference_disagrees_with_itself(cells, ref, aux, statics, forcing, names):
    """For each of `cells` (whole-pipeline inputs: statics dict, forcing = (doy, year, month, mid, nm,
    sw, tc, pn)): does another legal build of the reference's statements (B or C) leave the tolerance
    of the pinned build's result, or count other spin-up passes?  Such a cell -- a spin-up on a
    non-periodic orbit that never meets the 1 mm criterion, a threshold met by a hair -- has no
    reference value to be compared with (class 4)."""
    doy, year, month, mid, nm, sw, tc, pn = forcing
    res = np.zeros(len(cells), bool)
    for n_, c in enumerate(cells):
        sl = slice(int(c), int(c) + 1)
        st1 = {k: (v[:, sl] if np.ndim(v) == 2 else v[sl]) for k, v in statics.items()}
        for oo in reference_variants().values():
            r2, a2 = oo.grid(doy, year, month, mid, nm, np.ascontiguousarray(sw[:, sl]),
                             np.ascontiguousarray(tc[:, sl]), np.ascontiguousarray(pn[:, sl]), st1["lat"],
                             st1["elev"], st1["slop"], st1["asp"], st1["soil"], st1["au"], st1["resolution"],
                             monthly_out=(ref[names[0]].shape[0] != len(doy)), n_threads=1)
            differs = any(_bad(r2[k][:, 0], ref[k][:, c]).any() for k in names)
            differs |= int(a2["passes1"][0]) != int(aux["passes1"][c]) or int(a2["passes2"][0]) != int(aux["passes2"][c])
            res[n_] |= bool(differs)
        if not res[n_]:   # ... or the PINNED build with one input moved by one ulp (input_ulp_variants)
            from .binding import Oracle
            pinned = Oracle()
            for v in input_ulp_variants(st1):
                r2, a2 = pinned.grid(doy, year, month, mid, nm, np.ascontiguousarray(sw[:, sl]),
                                     np.ascontiguousarray(tc[:, sl]), np.ascontiguousarray(pn[:, sl]), v["lat"],
                                     v["elev"], v["slop"], v["asp"], v["soil"], v["au"], v["resolution"],
                                     monthly_out=(ref[names[0]].shape[0] != len(doy)), n_threads=1)
                differs = any(_bad(r2[k][:, 0], ref[k][:, c]).any() for k in names)
                differs |= int(a2["passes1"][0]) != int(aux["passes1"][c]) or int(a2["passes2"][0]) != int(aux["passes2"][c])
                if differs:
                    res[n_] = True
                    break
    return res


def assert_whole_pipeline(oracle, g, ref, aux, statics, doy, year, name="", names=None,
                          monthly_mid=None, max_flip_frac=0.05, forcing=None, bulk=BULK_PIPELINE, n_worst=0):
    """splash_grid (GPU: soil_hydro, snow threshold, both spin-ups, run_all, sm_lim) against the
    oracle's splash.grid restatement, each from its own spin-up.

    Strict for every cell in which the oracle met no critical event (aux['critical'] == 0): all
    cell-days within tolerance except from a grazing-sun day of the cell on.  Cells with a critical event (a few per
    cent on the synthetic soils) take a sticky infinite drainage time or not depending on the last
    bit of pow() -- the reference built against another libm flips ~1.5 % of them
    (profiles/r02_conditioning.md) -- so they are allowed to differ, but only `max_flip_frac` of them.
    monthly_mid: month id per day when g/ref are monthly (a grazing day then taints its month)."""
    from oracle.binding import HS_LIM_DEG, grazing_mask
    names = names or [k for k in ref if k in g]
    flagged = aux["critical"] > 0
    graz = grazing_mask(oracle, statics["lat"], statics["slop"], statics["asp"] - 180.0, doy, year,
                        hs_deg=HS_LIM_DEG)
    graz = np.cumsum(graz, axis=0) > 0  # a grazing day leaves a different bucket behind: from that day on
    if monthly_mid is not None:
        nm = int(monthly_mid[-1]) + 1
        gm = np.zeros((nm, graz.shape[1]), bool)
        for m in range(nm):
            gm[m] = graz[monthly_mid == m].any(0)
        graz = gm
    bad_cell = np.zeros(flagged.shape, bool)
    worst = {}
    prof = {}
    for k in names:
        a, b = np.asarray(g[k]).reshape(ref[k].shape), ref[k]
        assert (np.isnan(a) == np.isnan(b)).all(), f"{name}: NaN pattern of {k} differs"
        fl = _floor(k, ref) if monthly_mid is None else None
        prof[k] = error_profile(a, b, fl)
        bad = _bad(a, b, floor=fl) & ~graz
        bad_cell |= bad.any(0)
    ok = ~bad_cell  # the cells finally kept (a cell set aside for ANY series is out of every series' figure)
    for k in names:
        a, b = np.asarray(g[k]).reshape(ref[k].shape), ref[k]
        with np.errstate(invalid="ignore", divide="ignore"):
            e = np.where(graz | np.isnan(b), 0.0, np.maximum(np.abs(a - b) - ATOL, 0.0) / np.maximum(np.abs(b), 1e-300))
        worst[k] = float(e[:, ok].max()) if ok.any() else 0.0
    stray = bad_cell & ~flagged
    self_dis = 0
    if stray.any() and forcing is not None and stray.sum() <= MAX_SELF_DISAGREEMENTS:
        # no event of the oracle's run explains these cells: does the reference agree with ITSELF there?
        cells = np.where(stray)[0]
        excused = reference_disagrees_with_itself(cells, ref, aux, statics, forcing, list(names))
        stray[cells[excused]] = False
        self_dis = int(excused.sum())
    assert not stray.any(), f"{name}: cells {np.where(stray)[0][:8]} out of tolerance without a critical event"
    flips = int((bad_cell & flagged).sum())
    assert flips <= max(3, max_flip_frac * flagged.sum()), \
        f"{name}: {flips} of {int(flagged.sum())} critical-event cells differ"
    assert bad_cell.mean() <= 5e-3, f"{name}: {bad_cell.mean():.2e} of the cells set aside"
    assert graz.mean() <= 2e-2, f"{name}: {graz.mean():.2e} of the cell-days after a grazing-sun day"
    if bulk:  # the bulk of the cells that are not set aside (those are counted and bounded above)
        kept = {}
        for k in names:
            a, b = np.asarray(g[k]).reshape(ref[k].shape), ref[k]
            fl = _floor(k, ref) if monthly_mid is None else None
            kept[k] = error_profile(a[:, ok], b[:, ok], fl[:, ok] if fl is not None else None)
        bv = bulk_violations(kept, bulk, name + ": whole pipeline ")
        assert not bv, "; ".join(bv)
    worst_rows = []
    if n_worst:
        keep = ~graz & ~bad_cell[None, :]
        worst_rows = worst_cell_days({k: np.asarray(g[k]).reshape(ref[k].shape) for k in names}, ref, keep, n_worst)
    return dict(cells=int(flagged.size), critical_event_cells=int(flagged.sum()), set_aside=flips,
                reference_disagrees_with_itself=self_dis,
                grazing_cell_days=int(graz.sum()), max_rel_err=worst, ok=~bad_cell, profile=prof,
                worst=worst_rows)


# spinup_summary: a stray cell whose spin-up amplifies an ulp of libm to >= AMPLIFIED_FROM in the reference's own other
# builds is excused when the product is at most AMPLIFICATION_MARGIN times further from the pinned build than they are
AMPLIFIED_FROM = 1.0e-7
AMPLIFICATION_MARGIN = 1000.0   # the ratio of the seeds: the product's one-pass distance from the reference is 1e-13..1e-12,
                                # a thousand ulp; a libm ulp is one.  Seen: 12x (cell (2057, 1855)), 93x (cell (2192, 523))


def spinup_summary(oracle, state0, ai, passes2, aux, statics=None, forcing=None, max_flip_frac=0.05,
                   rtol_ai=1e-6):
    """The spin-up's RESULT against the oracle's brute-force loop (src/SPLASH.cpp:1594-1749 twice, with the
    aridity index in between, R/splash.point.R:141-152): the product ends a cell that never meets the 1 mm rule
    by exact-cycle detection and extrapolates the snow age; the oracle iterates all 1000 passes.

    state0 [5, n] (wn, snow, qin, td, nd), ai [n], passes2 [n] (or None): the product's; aux: the oracle's
    orc_cell_aux records of the same cells (Oracle.grid).  Rules (per cell):
      wn, snow, qin   |x - ref| <= 1e-5 |ref| + 1e-9
      td              the same, and the same NaN / inf pattern (the sticky drainage time of flat terrain); not
                      compared in a cell whose run met a noise-level drainage denominator (aux['fragile'] > 0: the
                      drainage time of such a day is rounding noise in the reference itself)
      nd              EXACTLY equal (an integer count of days; the detector adds 365 per skipped pass)
      aridity index   relative rtol_ai = 1e-6, a decade inside the tolerance of the series it is formed from (the year's
                      PET of spin-up #1's first pass over the year's precipitation; days on which the net radiation is
                      negative almost all day carry relative PET errors up to 1e-6).  Measured over 1 048 576 benchmark
                      cells (tools/spinup_parity_sweep.py, profiles/r06_spinup_parity_sweep.txt): median of the chunks'
                      maxima 1.4e-10, ONE cell at 1.3e-7, none above -- a first version of the rule, 1e-7, met that cell
      passes          equal
    Set aside (counted, bounded like assert_whole_pipeline's): a cell whose year holds a grazing-sun day (statics given:
    0 < hs < HS_LIM_DEG on one of the 365 spin-up days -- the reference's own radiation of that day is rounding
    noise and every pass of the spin-up goes through it).
    A cell that breaks a rule is classified: a cell in which the oracle met a critical event (aux['critical'] > 0:
    the sticky infinite drainage time the reference itself takes or not depending on the last bit of pow(),
    DESIGN.md 3.2) may differ, at most `max_flip_frac` of them (or 3); any other cell only if two other legal
    builds of the reference's own statements disagree on it too (needs statics and forcing = (doy, year, month,
    mid, nm, sw, tc, pn)); at most MAX_SELF_DISAGREEMENTS of those.
    Returns a dict of plain numbers with 'violations' (list of str)."""
    from oracle.binding import HS_LIM_DEG, grazing_mask
    ref = np.asarray(aux["state0"]).T  # [5, n]
    n = ref.shape[1]
    graz = np.zeros(n, bool)
    if statics is not None and forcing is not None:
        year0 = int(forcing[1][0])
        graz = grazing_mask(oracle, statics["lat"], statics["slop"], statics["asp"] - 180.0, np.arange(1, 366),
                            np.full(365, year0), hs_deg=HS_LIM_DEG).any(0)
    fragile = np.asarray(aux["fragile"]) > 0
    bad = np.zeros(n, bool)
    why = {}
    worst = {}
    for k, name in enumerate(("wn", "snow", "qin", "td")):
        a, f = np.asarray(state0[k]), ref[k]
        b = _bad(a, f)
        if name == "td":
            b &= ~fragile
        with np.errstate(invalid="ignore", divide="ignore"):
            e = np.where(np.isfinite(f) & (f != 0) & np.isfinite(a), np.maximum(np.abs(a - f) - ATOL, 0.0) / np.abs(f), 0.0)
        if name == "td":
            e = np.where(fragile, 0.0, e)
        worst[name] = e
        why[name] = b
        bad |= b
    nd_bad = ~((np.asarray(state0[4]) == ref[4]) | (np.isnan(state0[4]) & np.isnan(ref[4])))
    why["nd"] = nd_bad
    bad |= nd_bad
    with np.errstate(invalid="ignore", divide="ignore"):
        ai_ref = np.asarray(aux["AI"])
        ai_bad = ~((np.abs(ai - ai_ref) <= rtol_ai * np.abs(ai_ref)) | (ai == ai_ref) | (np.isnan(ai) & np.isnan(ai_ref)))
    why["AI"] = ai_bad
    bad |= ai_bad
    passes_equal = None
    if passes2 is not None:
        pdiff = np.asarray(passes2).astype(np.int64) != np.asarray(aux["passes2"]).astype(np.int64)
        passes_equal = int((~pdiff).sum())
        why["passes"] = pdiff
        bad |= pdiff
    flagged = np.asarray(aux["critical"]) > 0
    capped = np.asarray(aux["passes2"]) >= 1000
    viol = []
    bad_g = bad & graz          # set aside
    bad &= ~graz
    if graz.mean() > 2e-2:
        viol.append(f"{int(graz.sum())} of {n} cells hold a grazing-sun day")
    flips = bad & flagged
    if flips.sum() > max(3, max_flip_frac * flagged.sum()):
        viol.append(f"{int(flips.sum())} of {int(flagged.sum())} critical-event cells end the spin-up differently")
    stray = bad & ~flagged
    excused = 0
    if stray.any():
        cells = np.where(stray)[0]
        if statics is not None and forcing is not None and len(cells) <= MAX_SELF_DISAGREEMENTS:
            ex = np.zeros(len(cells), bool)
            doy, year, month, mid, nm, sw, tc, pn = forcing

            def rel_dev(x, r):   # [wn, snow, qin, td, AI]: relative distance of a spin-up result from the pinned build's
                with np.errstate(invalid="ignore", divide="ignore"):
                    d = np.abs(np.asarray(x, float) - np.asarray(r, float)) / np.maximum(np.abs(np.asarray(r, float)), 1e-300)
                return np.where(np.isfinite(d), d, 0.0)

            for i, c in enumerate(cells):
                sl = slice(int(c), int(c) + 1)
                st1 = {k: (v[:, sl] if np.ndim(v) == 2 else v[sl]) for k, v in statics.items()}
                pin = np.array([ref[0, c], ref[1, c], ref[2, c], ref[3, c], ai_ref[c]])
                dev_product = rel_dev([state0[0][c], state0[1][c], state0[2][c], state0[3][c], ai[c]], pin)
                dev_builds = np.zeros(5)
                for oo in reference_variants().values():
                    _, a2 = oo.grid(doy, year, month, mid, nm, np.ascontiguousarray(sw[:, sl]),
                                    np.ascontiguousarray(tc[:, sl]), np.ascontiguousarray(pn[:, sl]), st1["lat"],
                                    st1["elev"], st1["slop"], st1["asp"], st1["soil"], st1["au"], st1["resolution"],
                                    n_threads=1)
                    r2 = np.asarray(a2["state0"][0])
                    ai2 = float(a2["AI"][0])
                    dev_builds = np.maximum(dev_builds, rel_dev([r2[0], r2[1], r2[2], r2[3], ai2], pin))
                    differs = abs(ai2 - ai_ref[c]) > rtol_ai * abs(ai_ref[c]) or _bad(r2[:3], ref[:3, c]).any() or (not fragile[c] and _bad(r2[3:4], ref[3:4, c]).any()) or \
                        not (r2[4] == ref[4, c] or (r2[4] != r2[4] and ref[4, c] != ref[4, c])) or \
                        int(a2["passes2"][0]) != int(aux["passes2"][c])
                    ex[i] |= bool(differs)
                if not ex[i]:   # ... or the pinned build with one input moved by one ulp (input_ulp_variants)
                    for v in input_ulp_variants(st1):
                        _, a2 = oracle.grid(doy, year, month, mid, nm, np.ascontiguousarray(sw[:, sl]),
                                            np.ascontiguousarray(tc[:, sl]), np.ascontiguousarray(pn[:, sl]), v["lat"],
                                            v["elev"], v["slop"], v["asp"], v["soil"], v["au"], v["resolution"], n_threads=1)
                        r2 = np.asarray(a2["state0"][0])
                        ai2 = float(a2["AI"][0])
                        differs = abs(ai2 - ai_ref[c]) > rtol_ai * abs(ai_ref[c]) or _bad(r2[:3], ref[:3, c]).any() or \
                            (not fragile[c] and _bad(r2[3:4], ref[3:4, c]).any()) or \
                            not (r2[4] == ref[4, c] or (r2[4] != r2[4] and ref[4, c] != ref[4, c])) or \
                            int(a2["passes2"][0]) != int(aux["passes2"][c])
                        if differs:
                            ex[i] = True
                            break
                if not ex[i]:
                    # ... or the spin-up of this cell AMPLIFIES: another legal build of the reference's own statements
                    # (one ulp of every libm result, or contraction and wider vectors: seeds of 1e-16) ends the spin-up
                    # at least AMPLIFIED_FROM away from the pinned build in every quantity the product breaks a rule on,
                    # and the product -- whose one-pass distance from the reference is 1e-13..1e-12, a thousand ulp -- is
                    # no more than AMPLIFICATION_MARGIN (= that ratio of the seeds) times further.  (Found in round 6: cell (2057, 1855) of the
                    # benchmark grid, 49 passes on an expanding orbit: pass-to-pass distance 4.6e-13 for ten passes, then
                    # x30 every ten; build C ends 1.9e-6 from the pinned build in qin, the product 2.2e-5.)
                    broke = np.array([why["wn"][c], why["snow"][c], why["qin"][c], why["td"][c], why["AI"][c]])
                    other = why["nd"][c] or ("passes" in why and why["passes"][c])   # counts: never excused this way
                    if broke.any() and not other and (dev_builds[broke] >= AMPLIFIED_FROM).all() and \
                            (dev_product[broke] <= AMPLIFICATION_MARGIN * dev_builds[broke]).all():
                        ex[i] = True
            stray[cells[ex]] = False
            excused = int(ex.sum())
        if stray.any():
            c = int(np.where(stray)[0][0])
            broken = [k for k, v in why.items() if v[c]]
            viol.append(f"{int(stray.sum())} cells end the spin-up out of tolerance without a critical event "
                        f"(rules broken, cells: { {k: int((v & stray).sum()) for k, v in why.items()} }); first: cell {c} "
                        f"breaks {broken}: product {[float(state0[k][c]) for k in range(5)]} AI {float(ai[c])!r} ref "
                        f"{[float(ref[k][c]) for k in range(5)]} AI {float(ai_ref[c])!r} passes "
                        f"{None if passes2 is None else int(passes2[c])}/{int(aux['passes2'][c])} fragile {int(aux['fragile'][c])}")
    keep = ~bad & ~graz
    return dict(cells=int(n), capped=int(capped.sum()), capped_within_tolerance=int((capped & keep).sum()),
                passes_equal=passes_equal, critical_event_cells=int(flagged.sum()),
                critical_event_cells_that_differ=int(flips.sum()), excused=excused,
                grazing_cells=int(graz.sum()), grazing_cells_that_differ=int(bad_g.sum()),
                nd_equal=int((~nd_bad).sum()),
                max_rel_state={k: float(v[keep].max()) if keep.any() else 0.0 for k, v in worst.items()},
                max_rel_state_capped=float(max((v[capped & keep].max() if (capped & keep).any() else 0.0)
                                               for v in worst.values())),
                max_rel_ai=float(np.nanmax(np.where(keep & (ai_ref != 0), np.abs(ai - ai_ref) / np.abs(ai_ref), 0.0)))
                if keep.any() else 0.0,
                excuse_builds=variants_status(), violations=viol)
