"""Evidence for the conditioning masks of the parity tests (build container or GPU box; CPU only).

The parity tests set aside a few cell-days on which the reference's own result is rounding noise
(DESIGN.md "Conditioning").  This script shows, with the oracle alone, that those cell-days are where
the reference disagrees WITH ITSELF, and that little else is set aside:

  A  oracle/liboracle.so            pinned flags (gcc -O2, no -march, no contraction)
  B  -O3 -march=x86-64-v3 -ffp-contract=fast     another legal build of the same source
  C  -DORC_PERTURB_LIBM             the same source against "another correct libm": every pow/exp/
                                    log/sin/cos/tan/acos result moved by -1/0/+1 ulp (hash of the
                                    arguments); exact cases and the float viscosity left alone
  D  ... -DORC_PERTURB_ULPS=4       the same with up to 4 ulp: a math library of a few ulp, which is
                                    what the product's log/exp/acos are (< 2.5 ulp)
(in B and C the float viscosity routine and the density it rounds to float keep the pinned
arithmetic: the product reproduces those bit for bit, they are not a source of disagreement.)

For every grid the TRANSIENT loop (run_all from A's spun-up state) is run with A, B and C, once as
a trajectory and once in one-step mode (every day restarted from A's state series).  Reported:

  * one-step: cell-days on which B or C differ from A by more than the parity tolerance, and how
    many of them carry an event bit of orc_grid_transient2 (grazing sun / noise denominator /
    overflow residual) -- the event mask must cover them;
  * how the disagreement depends on the slope sunset angle hs (to place HS_LIM_DEG);
  * trajectory: days a cell stays out of tolerance after an event (to place the decay window);
  * the share of cell-days the masks set aside.

    python tools/conditioning_evidence.py [--cells 8192] [--days 730] [--out profiles/r02_conditioning.md]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle.binding import HS_LIM_DEG, Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "oracle", "splash_oracle.c")
OUT5 = ("wn", "snow", "ro", "aet", "pet")
RTOL, ATOL = 1e-5, 1e-9


def build_variant(tag, flags):
    so = os.path.join(tempfile.gettempdir(), f"liboracle_{tag}.so")
    cmd = ["gcc", "-std=c11", "-fPIC", "-fopenmp", "-shared", "-Wl,-Bsymbolic"] + flags + \
        ["-o", so, SRC, "-lm"]
    subprocess.check_call(cmd)
    return so


def bad(a, b):
    return np.abs(a - b) > RTOL * np.abs(b) + ATOL


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=8192)
    ap.add_argument("--days", type=int, default=730)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_conditioning.md"))
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()

    A = Oracle()
    variants = {
        "B (-O3 -march=x86-64-v3 -ffp-contract=fast)":
            Oracle(build_variant("fast", ["-O3", "-march=x86-64-v3", "-ffp-contract=fast"])),
        "C (libm results +-1 ulp)": Oracle(build_variant("pert", ["-O2", "-DORC_PERTURB_LIBM"])),
        "D (libm results up to +-4 ulp)": Oracle(build_variant("pert4", ["-O2", "-DORC_PERTURB_LIBM",
                                                                         "-DORC_PERTURB_ULPS=4"])),
    }
    nt, W = a.days, 128
    H = a.cells // W
    doy, year, month, mid, nm = A.calendar(2001, nt)
    # grids: two generic terrain grids (lat 60..-60), the first rows of the 4096-row benchmark grid
    # (lat ~60 N, winter sun grazing the slopes), one flat grid
    grids = [("terrain, seed 777000", 777000, H, H, 0, 1), ("terrain, seed 784919", 784919, H, H, 0, 1),
             ("benchmark grid rows 0..%d of 4096 (lat 60 N)" % (H * W // 4096 or 1), 20221130, 4096, None, 0, 1),
             ("flat, seed 816595", 816595, H, H, 0, 0)]
    lines = []
    hs_edges = np.array([0, 1e-5, 1e-4, 3e-4, 1e-3, 2e-3, 5e-3, 1e-2, 3e-2, 1e-1, 1.0, 180.0])
    hs_tot = np.zeros(len(hs_edges) - 1, np.int64)
    hs_bad = {v: np.zeros(len(hs_edges) - 1, np.int64) for v in variants}
    tot = dict(cd=0, masked_cd=0, ev=np.zeros(4, np.int64), traj_masked=0)
    onestep = {v: dict(bad=0, covered=0, uncovered=0, worst_unc=0.0) for v in variants}
    traj = {v: dict(bad_cd=0, bad_cells=0, outside_window=0, cells_with_event=0) for v in variants}
    recov = []
    spin = {v: dict(cells=0, passes_differ=0, state_differ=0, state_differ_flagged=0, flagged=0) for v in variants}

    for name, seed, Hs, Hg, cell0, terrain in grids:
        if Hg is None:  # benchmark grid: 4096 columns
            Wg, n = 4096, max(4096, a.cells // 4096 * 4096)
        else:
            Wg, n = W, Hg * W
        st = A.synth_statics(seed, Hs, Wg, cell0, n, terrain)
        sw, tc, pn = A.synth_forcing(seed, Hs, Wg, cell0, n, 0, doy, st["lat"], st["elev"])
        # spin-up with A (the pass count is a conditioning class of its own: see the end)
        _, aux = A.grid(doy[:365], year[:365], month[:365], mid[:365], 12, sw[:365], tc[:365], pn[:365],
                        st["lat"], st["elev"], st["slop"], st["asp"], st["soil"], st["au"],
                        st["resolution"], n_threads=a.threads)
        si = np.ascontiguousarray(aux["soil_info"].T)
        Tt = aux["Tt"].copy()
        st0 = np.ascontiguousarray(aux["state0"].T)
        args = (doy, year, month, sw, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"] - 180.0, si, Tt, st0)
        refA, _, ev, hs = A.grid_transient2(*args, want_hs=True, n_threads=a.threads)
        prev = np.stack([refA[k] for k in ("wn", "snow", "qin_prev", "tdrain", "snwage")])
        lam_small = si[4] < 0.01
        event = ev != 0
        tot["cd"] += ev.size
        tot["masked_cd"] += int(event.sum())
        for b in range(4):
            tot["ev"][b] += int(((ev >> b) & 1).sum())
        # trajectory mask of the tests: a cell from its first grazing-sun day or critical event on
        win = np.cumsum((ev & (2 | 4 | 8)) != 0, axis=0) > 0
        tot["traj_masked"] += int((np.cumsum((ev & 2) != 0, axis=0) > 0).sum())
        hsd = hs.astype(np.float64)
        hb = np.digitize(hsd, hs_edges) - 1
        hb = np.clip(hb, 0, len(hs_edges) - 2)
        pos = hsd > 0
        hs_tot += np.bincount(hb[pos], minlength=len(hs_edges) - 1)
        for vname, V in variants.items():
            # the spin-up's stopping rule |end_sm - start_sm| > 1.0 (src/SPLASH.cpp:1697) is a
            # discontinuity of its own: does another build stop after the same number of passes?
            _, auxv = V.grid(doy[:365], year[:365], month[:365], mid[:365], 12, sw[:365], tc[:365], pn[:365],
                             st["lat"], st["elev"], st["slop"], st["asp"], st["soil"], st["au"],
                             st["resolution"], n_threads=a.threads)
            pd_ = (auxv["passes1"] != aux["passes1"]) | (auxv["passes2"] != aux["passes2"])
            sd = np.zeros(n, bool)
            for q in range(3):  # wn, snow, qin
                sd |= bad(auxv["state0"][:, q], aux["state0"][:, q])
            spin[vname]["cells"] += n
            spin[vname]["passes_differ"] += int(pd_.sum())
            spin[vname]["state_differ"] += int(sd.sum())
            flagged = (aux["critical"] > 0) | (aux["passes2"] >= 1000)
            spin[vname]["state_differ_flagged"] += int((sd & flagged).sum())
            spin[vname]["flagged"] += int(flagged.sum())
            one, _, _, _ = V.grid_transient2(*args, prev=prev, n_threads=a.threads)
            b1 = np.zeros(ev.shape, bool)
            worst = np.zeros(ev.shape)
            for k in OUT5:
                b1 |= bad(one[k], refA[k])
                worst = np.maximum(worst, np.abs(one[k] - refA[k]) / np.maximum(np.abs(refA[k]), 1e-300) *
                                   (np.abs(one[k] - refA[k]) > ATOL))
            r = onestep[vname]
            r["bad"] += int(b1.sum())
            r["covered"] += int((b1 & event).sum())
            unc = b1 & ~event
            r["uncovered"] += int(unc.sum())
            if unc.any():
                r["worst_unc"] = max(r["worst_unc"], float(worst[unc].max()))
                for t, c in np.argwhere(unc)[:5]:
                    lines.append(f"  uncovered one-step difference: grid '{name}' cell {c} day {t} hs={hsd[t, c]:.3g} "
                                 f"lambda={si[4, c]:.4g} worst rel {worst[t, c]:.2e}")
            hs_bad[vname] += np.bincount(hb[pos & b1], minlength=len(hs_edges) - 1)
            tr, _, _, _ = V.grid_transient2(*args, n_threads=a.threads)
            bt = np.zeros(ev.shape, bool)
            for k in OUT5:
                bt |= bad(tr[k], refA[k])
            q = traj[vname]
            q["bad_cd"] += int(bt.sum())
            q["bad_cells"] += int(bt.any(0).sum())
            q["outside_window"] += int((bt & ~win).sum())
            q["cells_with_event"] += int(event.any(0).sum())
            # recovery time after the last event of a cell
            for c in np.where(bt.any(0) & event.any(0))[0]:
                te = np.where(event[:, c])[0]
                tb = np.where(bt[:, c])[0]
                recov.append(int(tb.max() - te[te <= tb.max()].max()) if (te <= tb.max()).any() else -1)
        print("done:", name, flush=True)

    md = ["# Conditioning evidence (round 2) -- `tools/conditioning_evidence.py`", "",
          f"Grids: {len(grids)} synthetic grids, {a.cells} cells x {nt} days each (2 terrain, first rows of the "
          "4096-row benchmark grid at 60 N, 1 flat); transient loop from the pinned oracle's spun-up "
          f"state.  Tolerance |x-ref| <= {RTOL:g}|ref| + {ATOL:g} on wn, snow, ro, aet, pet.  "
          "A = pinned oracle; variants are other legal builds of the SAME source (docstring).", "",
          f"Cell-days: {tot['cd']}.  Event cell-days (any bit): {tot['masked_cd']} "
          f"({100.0 * tot['masked_cd'] / tot['cd']:.4f} %): noise denominator {tot['ev'][0]}, grazing sun "
          f"(0 < hs < {HS_LIM_DEG:g} deg) {tot['ev'][1]}, overflow residual {tot['ev'][2]}, critical (zero-or-one-ulp) "
          f"denominator {tot['ev'][3]}.  "
          f"Cell-days at/after a grazing-sun day of their cell (set aside by the trajectory tests): "
          f"{tot['traj_masked']} ({100.0 * tot['traj_masked'] / tot['cd']:.4f} %).", "",
          "## One-step mode (each day restarted from A's state): where the reference disagrees with itself", "",
          "| variant | cell-days out of tolerance | carrying an event bit | not covered | worst uncovered rel. diff |",
          "|---|---|---|---|---|"]
    for v, r in onestep.items():
        md.append(f"| {v} | {r['bad']} | {r['covered']} | {r['uncovered']} | {r['worst_unc']:.2e} |")
    md += ["", "## Disagreement against the slope sunset angle hs (one-step, all grids)", "",
           "| hs [deg] | cell-days | " + " | ".join(f"out of tol., {v.split()[0]}" for v in variants) + " |",
           "|---|---|" + "---|" * len(variants)]
    for i in range(len(hs_edges) - 1):
        md.append(f"| {hs_edges[i]:g} .. {hs_edges[i + 1]:g} | {hs_tot[i]} | " +
                  " | ".join(str(int(hs_bad[v][i])) for v in variants) + " |")
    md += ["", "## Trajectory mode (state carried): propagation of an event", "",
           "| variant | cell-days out of tol. | cells affected | cells with an event | out of tol. before any grazing/critical event of the cell |",
           "|---|---|---|---|---|"]
    for v, q in traj.items():
        md.append(f"| {v} | {q['bad_cd']} | {q['bad_cells']} | {q['cells_with_event']} | {q['outside_window']} |")
    if recov:
        rc = np.array([r for r in recov if r >= 0])
        if rc.size:
            md += ["", f"Days from a cell's last event to its last out-of-tolerance day (cells that have both, "
                   f"n = {rc.size}): median {int(np.median(rc))}, 90 % {int(np.percentile(rc, 90))}, "
                   f"max {int(rc.max())}."]
    md += ["", "## Spin-up (both spin-ups of every cell, 365 days, 2..1000 passes each)", "",
           "Flagged = cells in which A met a critical event (zero-or-one-ulp transmittance, "
           "`orc_critical_events`) during the spin-ups or never met the 1 mm stopping rule (1000 passes).", "",
           "| variant | cells | pass count differs from A | spun-up wn/snow/qin out of tolerance | ... of which flagged | cells flagged |",
           "|---|---|---|---|---|---|"]
    for v, q in spin.items():
        md.append(f"| {v} | {q['cells']} | {q['passes_differ']} | {q['state_differ']} | {q['state_differ_flagged']} | {q['flagged']} |")
    if lines:
        md += ["", "## Uncovered one-step differences (first few)", ""] + lines
    open(a.out, "w").write("\n".join(md) + "\n")
    print("\n".join(md))
    print(json.dumps({"onestep": onestep, "traj": traj, "spin": spin}))


if __name__ == "__main__":
    main()
