"""Class 1 of DESIGN.md 3.2 (sun grazing the slope) shown on the REAL reference (build container; CPU).

The grazing-sun class lives entirely in src/SOLAR.cpp:129-165, 220-224 and src/EVAP.cpp:82-264 -- the
part of the reference that compiles here from its own sources (oracle/Makefile).  This script runs
those unmodified classes in three legal builds

  A  oracle/_ref/libsplash_ref.so    g++ -O2, no -march, no contraction (the pinned build)
  B  oracle/_ref/libsplash_ref_B.so  g++ -O3 -march=x86-64-v3 -ffp-contract=fast
  K  oracle/_ref/libsplash_ref_K.so  clang++ -O2 -march=x86-64-v3 (another compiler; contraction "on")

on the SOLAR/EVAP calls of the benchmark grid's cell-days with a low sun on the slope (0 < hs < 1 deg),
with the inputs those calls really get in a run (supply fraction, snow pack and snow age from the
oracle's trajectory of the cell), and tabulates against hs how often two builds of the reference
differ by more than the parity tolerance in the quantities that reach the five outputs (pet, aet,
cond; snowmelt).  The restatement (oracle/liboracle.so) is run on the same rows: it must be
bit-identical to A.

    python tools/conditioning_real_reference.py [--rows 256] [--out profiles/r03_conditioning_real_reference.md]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle.binding import HS_LIM_DEG, Oracle, Ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL, ATOL = 1e-5, 1e-9
SEED, H, W = 20221130, 4096, 4096


def ruv_of(lat, slop, asp_c, sd, cd):
    """ru/rv of src/SOLAR.cpp:129-140 (numpy, approximate: only used to SELECT the band)"""
    pir = np.pi / 180.0
    sl, cl = np.sin(lat * pir), np.cos(lat * pir)
    ss, cs = np.sin(slop * pir), np.cos(slop * pir)
    sa, ca = np.sin(asp_c * pir), np.cos(asp_c * pir)
    a = sd * (cl * ss * ca - sl * cs)
    b = cd * (cl * cs + sl * ss * ca)
    c = cd * (ss * sa)
    b2c2 = b * b + c * c
    d = b2c2 - a * a
    with np.errstate(invalid="ignore", divide="ignore"):
        sinfirst = np.where(d < 0, a * c / b2c2, (a * c + b * np.sqrt(np.abs(d))) / b2c2)
        return (-a + c * sinfirst) / b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=256, help="rows of the 4096-row benchmark grid that are scanned (spread evenly)")
    ap.add_argument("--days", type=int, default=365)
    ap.add_argument("--band", type=float, default=1.0, help="upper hs of the band [deg]")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_conditioning_real_reference.md"))
    a = ap.parse_args()
    O = Oracle()
    refs = {"A": Ref(), "B": Ref("B"), "K": Ref("K")}
    nt = a.days
    doy, year, month, mid, nm = O.calendar(2001, nt)
    orb = np.array([O.orbit(int(n), int(y)) for n, y in zip(doy, year)])
    pir = np.pi / 180.0
    sd, cd = np.sin(orb[:, 4] * pir)[:, None], np.cos(orb[:, 4] * pir)[:, None]
    lim = 1.0 - np.cos(a.band * pir)

    rows = np.unique(np.linspace(0, H - 1, a.rows).astype(int))
    sel_cells = []
    n_scanned = 0
    for r in rows:                                   # 1) find the cells that have a day in the band
        st = O.synth_statics(SEED, H, W, int(r) * W, W, 1)
        ruv = ruv_of(st["lat"], st["slop"], st["asp"] - 180.0, sd, cd)
        hit = ((ruv > -1.0) & (1.0 + ruv < lim)).any(0)
        sel_cells += [int(r) * W + int(c) for c in np.where(hit)[0]]
        n_scanned += W * nt
    sel_cells = np.array(sel_cells, dtype=np.int64)
    print(f"{len(sel_cells)} cells of {len(rows) * W} have a day with 0 < hs < {a.band} deg", flush=True)

    # 2) the oracle's run of those cells: spin-ups + transient year, to get the inputs of every EVAP call
    stc = {k: [] for k in ("lat", "elev", "slop", "asp", "resolution")}
    soil, au, F = [], [], []
    for c in sel_cells:                              # statics/forcing are counter based: any cell anywhere
        s1 = O.synth_statics(SEED, H, W, int(c), 1, 1)
        for k in stc:
            stc[k].append(s1[k][0])
        soil.append(s1["soil"][:, 0])
        au.append(s1["au"][:, 0])
        F.append([x[:, 0] for x in O.synth_forcing(SEED, H, W, int(c), 1, 0, doy, s1["lat"], s1["elev"])])
    st = {k: np.array(v) for k, v in stc.items()}
    st["soil"] = np.ascontiguousarray(np.array(soil).T)
    st["au"] = np.ascontiguousarray(np.array(au).T)
    sw_in = np.ascontiguousarray(np.array([f[0] for f in F]).T)
    tc = np.ascontiguousarray(np.array([f[1] for f in F]).T)
    pn = np.ascontiguousarray(np.array([f[2] for f in F]).T)
    _, aux = O.grid(doy, year, month, mid, nm, sw_in, tc, pn, st["lat"], st["elev"], st["slop"], st["asp"],
                    st["soil"], st["au"], st["resolution"])
    si = np.ascontiguousarray(aux["soil_info"].T)
    st0 = np.ascontiguousarray(aux["state0"].T)
    ser, _, ev, hs = O.grid_transient2(doy, year, month, sw_in, tc, pn, st["lat"], st["elev"], st["slop"],
                                       st["asp"] - 180.0, si, aux["Tt"].copy(), st0, want_hs=True)
    hs = hs.astype(np.float64)
    # state BEFORE day t, then the prelude of run_one_day (src/SPLASH.cpp:1042-1048, 1225-1231)
    prev = lambda k, j: np.vstack([st0[j][None, :], ser[k][:-1]])
    wn_p, snow_p, nd_p = prev("wn", 0), prev("snow", 1), prev("snwage", 4)
    snf = np.empty_like(tc)
    for j in range(len(sel_cells)):
        _, _, snf[:, j] = O.snow_partition(tc[:, j], pn[:, j], month, st["lat"][j], st["elev"][j])
    SAT, FC, lam, depth, RES = si[0], si[2], si[4], si[5], si[7]
    KG_o = 1000.0 / (997 * 9.80665)
    coeff_A = np.exp(np.log(33.0) + (1.0 / lam) * np.log(FC / (depth * 1000.0)))
    Wmax = np.power(coeff_A * KG_o / depth, 1.0 / ((1 / lam) + 1.0)) * (depth * 1000.0)
    with np.errstate(invalid="ignore"):
        sw = (wn_p - RES) / (Wmax - RES)
    sw = np.where((sw < 0) | np.isnan(sw), 0.0, np.where(sw > 1.0, 1.0, sw))
    nd = np.where(snf > 0.0, 0.0, nd_p + 1.0)
    snow = snow_p + snf

    band = (hs > 0) & (hs < a.band)
    tt, cc = np.where(band)
    m = len(tt)
    print(f"{m} cell-days in the band", flush=True)
    args = (st["lat"][cc], st["elev"][cc], sw[tt, cc], doy[tt], year[tt], sw_in[tt, cc], tc[tt, cc],
            st["slop"][cc], (st["asp"] - 180.0)[cc], snow[tt, cc], nd[tt, cc])
    E = {k: r.evap_batch(*args) for k, r in refs.items()}
    sargs = (st["lat"][cc], st["elev"][cc], doy[tt], year[tt], sw_in[tt, cc], tc[tt, cc], st["slop"][cc],
             (st["asp"] - 180.0)[cc], snow[tt, cc], nd[tt, cc], sw[tt, cc])
    S = {k: r.solar_batch(*sargs) for k, r in refs.items()}
    Eo = O.evap_batch(*args)
    same = np.array_equal(Eo.view(np.uint64), E["A"].view(np.uint64))
    # control rows: an ordinary sun on the same cells (hs > 10 deg)
    ct, cc2 = np.where(hs > 10.0)
    pick = np.random.default_rng(1).choice(len(ct), size=min(len(ct), 400000), replace=False)
    ct, cc2 = ct[pick], cc2[pick]
    cargs = (st["lat"][cc2], st["elev"][cc2], sw[ct, cc2], doy[ct], year[ct], sw_in[ct, cc2], tc[ct, cc2],
             st["slop"][cc2], (st["asp"] - 180.0)[cc2], snow[ct, cc2], nd[ct, cc2])
    Ec = {k: r.evap_batch(*cargs) for k, r in refs.items()}

    # quantities that reach the five outputs: cond 0, pet 2, aet 3, snowmelt 4 (src/etr.h:44-56)
    cols = {"cond": 0, "pet": 2, "aet": 3, "snowmelt": 4}

    def out_of_tol(X, Y):
        bad = np.zeros(len(X), bool)
        worst = np.zeros(len(X))
        for k, j in cols.items():
            x, y = X[:, j], Y[:, j]
            fl = 1e-7 * np.abs(Y[:, 2]) if k == "aet" else 0.0   # oracle/parity.py aet_floor
            with np.errstate(invalid="ignore"):
                d = np.abs(x - y)
                b = ~(d <= RTOL * np.abs(y) + ATOL + fl) & ~((x == y) | (np.isnan(x) & np.isnan(y)))
                rel = np.where(b, d / np.maximum(np.abs(y), 1e-300), 0.0)
            bad |= b
            worst = np.maximum(worst, np.nan_to_num(rel, nan=0.0, posinf=1e300))
        return bad, worst

    edges = np.array([0, 1e-4, 3e-4, 1e-3, 2e-3, 5e-3, 1e-2, 3e-2, 1e-1, 3e-1, 1.0])
    edges = edges[edges <= a.band + 1e-12]
    hb = np.clip(np.digitize(hs[tt, cc], edges) - 1, 0, len(edges) - 2)
    L = ["# Class 1 (sun grazing the slope) on the REAL reference -- `tools/conditioning_real_reference.py`", "",
         f"Source: unmodified `src/{{global,SOLAR,EVAP}}.cpp` compiled where they lie (`oracle/Makefile`, targets `ref`, "
         f"`ref-variants`): A = g++ -O2 (pinned), B = g++ -O3 -march=x86-64-v3 -ffp-contract=fast, K = clang++ -O2 "
         f"-march=x86-64-v3.  Rows: every `EVAP::calculate_daily_fluxes` call of one year on the cells of {len(rows)} "
         f"rows (spread over all latitudes) of the 4096x4096 benchmark grid that have a day with 0 < hs < {a.band:g} deg "
         f"({len(sel_cells)} cells of {len(rows) * W}; {m} cell-days in the band), with the supply fraction, snow pack "
         f"and snow age of the oracle's run of the cell.  Out of tolerance = |x - A| > 1e-5|A| + 1e-9 in cond, pet, "
         f"aet (+1e-7 pet) or snowmelt.", "",
         f"Restatement (`oracle/liboracle.so`) vs A on the band rows: {'bit-identical' if same else 'DIFFERS'} "
         f"({m} rows x 11 outputs).", "",
         "| hs [deg] | cell-days | B out of tol. | K out of tol. | worst rel. diff B | worst rel. diff K |", "|---|---|---|---|---|---|"]
    bB, wB = out_of_tol(E["B"], E["A"])
    bK, wK = out_of_tol(E["K"], E["A"])
    for i in range(len(edges) - 1):
        mk = hb == i
        L.append(f"| {edges[i]:g} .. {edges[i + 1]:g} | {int(mk.sum())} | {int(bB[mk].sum())} | {int(bK[mk].sum())} | "
                 f"{(wB[mk].max() if mk.any() else 0):.2e} | {(wK[mk].max() if mk.any() else 0):.2e} |")
    cB, cwB = out_of_tol(Ec["B"], Ec["A"])
    cK, cwK = out_of_tol(Ec["K"], Ec["A"])
    L.append(f"| control: hs > 10 (same cells) | {len(ct)} | {int(cB.sum())} | {int(cK.sum())} | {cwB.max():.2e} | {cwK.max():.2e} |")
    below = hs[tt, cc] < HS_LIM_DEG
    L += ["", f"Below the tests' limit HS_LIM_DEG = {HS_LIM_DEG:g} deg: {int(below.sum())} cell-days, "
          f"{int((bB | bK)[below].sum())} out of tolerance in B or K.  At or above it: {int((~below).sum())} cell-days, "
          f"{int((bB | bK)[~below].sum())} out of tolerance.", ""]
    # the radiation itself: relative difference of rw (src/SOLAR.cpp:220-224) between builds, by hs
    L += ["## The shortwave term rw (src/SOLAR.cpp:220-224), relative difference between builds", "",
          "| hs [deg] | median B | max B | median K | max K | 1e-16/hs_rad^2 at the bin's lower edge |", "|---|---|---|---|---|---|"]
    with np.errstate(invalid="ignore", divide="ignore"):
        rB = np.abs(S["B"][:, 2] - S["A"][:, 2]) / np.abs(S["A"][:, 2])
        rK = np.abs(S["K"][:, 2] - S["A"][:, 2]) / np.abs(S["A"][:, 2])
    for i in range(len(edges) - 1):
        mk = (hb == i) & np.isfinite(rB) & np.isfinite(rK)
        if not mk.any():
            L.append(f"| {edges[i]:g} .. {edges[i + 1]:g} | - | - | - | - | - |")
            continue
        lo = max(edges[i], 1e-5) * pir
        L.append(f"| {edges[i]:g} .. {edges[i + 1]:g} | {np.median(rB[mk]):.1e} | {rB[mk].max():.1e} | "
                 f"{np.median(rK[mk]):.1e} | {rK[mk].max():.1e} | {1e-16 / lo ** 2:.1e} |")
    worst = [i for i in np.argsort(-np.maximum(wB, wK))[:20] if max(wB[i], wK[i]) > 0]
    L += ["", "## Band rows out of tolerance in B or K (worst first, at most 20)", "", "| cell | doy | hs [deg] | lat | slope | aspect-180 | pet A | pet B | pet K | aet A | aet B | aet K |",
          "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for i in worst:
        c = cc[i]
        L.append(f"| {int(sel_cells[c])} | {int(doy[tt[i]])} | {hs[tt[i], c]:.3g} | {st['lat'][c]:.3f} | {st['slop'][c]:.3f} | "
                 f"{st['asp'][c] - 180:.2f} | {E['A'][i, 2]:.9g} | {E['B'][i, 2]:.9g} | {E['K'][i, 2]:.9g} | "
                 f"{E['A'][i, 3]:.9g} | {E['B'][i, 3]:.9g} | {E['K'][i, 3]:.9g} |")
    txt = "\n".join(L) + "\n"
    print(txt)
    open(a.out, "w").write(txt)
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
