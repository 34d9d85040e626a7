"""Randomised invariants of the host-buffer entry point (splash_grid_run through api.splash_grid), GPU box:
for random grid shapes, NA masks, launch blockings, device lists {0,0}/{0,0,0}, pipeline lanes (1-4),
straggler pools too small for a block (finished in place), a series in two time chunks (ABI 5: continued from the
first chunk's state), calls in flight side by side
(splash_grid_submit/_wait), single-precision transport of forcing and of results, the large-call paths at
small size (page mapping per block, page-locked caller arrays, the library's own blocking; outputs handed in
filled with a sentinel), monthly/daily outputs and basin maps, the
results must equal the plain call BITWISE on
every valid cell, NA cells must come back NaN, and the per-basin totals must be identical limbs.
    python tools/fuzz_grid_run.py [n_cases=12] [rng_seed=20221130]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rsplash_amd.api as api
from oracle.binding import Oracle
o = Oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20221130)
bad = 0
for case in range(n_cases):
    H, W = int(rng.integers(3, 20)), int(rng.integers(5, 90))
    nt = int(rng.choice([365, 400, 731]))
    monthly = bool(rng.integers(0, 2))
    seed = int(rng.integers(1, 2**31))
    doy, year, month, mid, nm = o.calendar(2001, nt)
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    st = o.synth_statics(seed, H, W, 0, H * W, 1)
    sw, tc, pn = o.synth_forcing(seed, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
    n = H * W
    mask = rng.random(n) < rng.choice([0.0, 0.3, 0.7])
    mask[int(rng.integers(0, n))] = False
    elev = st["elev"].copy(); elev[mask] = np.nan
    swm, tcm, pnm = sw.copy(), tc.copy(), pn.copy()
    swm[:, mask] = np.nan; tcm[:, mask] = np.nan; pnm[:, mask] = np.nan
    basin = rng.integers(0, 5, n).astype(np.int32)
    kw = dict(time_index=dates, sim_control=dict(monthly_out=monthly), want_diag=True, basin=basin, n_basin=5)
    args = (elev, st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"])
    ref = api.splash_grid(swm, tcm, pnm, *args, **kw)
    variants = {
        "blocked": dict(max_cells_per_launch=int(rng.integers(64, 400))),
        "two workers": dict(devices=[0, 0]),
        "three workers, blocked": dict(devices=[0, 0, 0], max_cells_per_launch=int(rng.integers(64, 300))),
    }
    names = [k for k in ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "sm_lim") if k in ref]
    msgs = []
    for k in names:
        if not (np.isnan(ref[k][:, mask]).all() and np.isfinite(ref[k][:, ~mask]).all()):
            msgs.append(f"NaN pattern of {k}")
    lanes = int(rng.integers(1, 5))
    variants[f"{lanes} lanes, blocked"] = dict(max_cells_per_launch=int(rng.integers(64, 400)), _env=dict(SPLASH_HIP_LANES=str(lanes)))
    variants["tiny pool, blocked"] = dict(max_cells_per_launch=int(rng.integers(64, 400)),
                                          _env=dict(SPLASH_HIP_POOL_CAP=str(int(rng.integers(1, 4)))))
    # round 4: the paths that only ran for big calls (result pages mapped per block, caller arrays page-locked
    # by a helper thread, the library's own many-blocks plan), sentinel-filled outputs handed in, and
    # single-precision results
    big = dict(SPLASH_HIP_PREFAULT_MIN_BYTES="0", SPLASH_HIP_REGISTER_MIN_BYTES="0",
               SPLASH_HIP_MIN_BLOCK_CELLS=str(int(rng.integers(32, 200))), SPLASH_HIP_BLOCKS_PER_LANE=str(int(rng.integers(2, 7))),
               SPLASH_HIP_PREFAULT=str(rng.choice(["populate", "touch"])), SPLASH_HIP_LANES=str(int(rng.integers(1, 5))))
    variants["large-call paths"] = dict(_env=big, _sentinel=True)
    variants["large-call paths, tiny pool"] = dict(_env=dict(big, SPLASH_HIP_POOL_CAP="2"), _sentinel=True)
    variants["f32 results"] = dict(out_dtype=np.float32, max_cells_per_launch=int(rng.integers(64, 400)), _env=dict(big))
    variants["in flight (a)"] = dict(max_cells_per_launch=int(rng.integers(64, 400)), submit=True, mem_fraction=0.3)
    variants["in flight (b)"] = dict(devices=[0, 0], submit=True, mem_fraction=0.3)
    jobs = {}
    for tag, extra in variants.items():
        extra = dict(extra)
        env = extra.pop("_env", {})
        if extra.pop("_sentinel", False):
            nto = nm if monthly else nt
            extra["out"] = {k: np.full((nto, n), -12345.678) for k in names}
        os.environ.update(env)
        g = api.splash_grid(swm, tcm, pnm, *args, **kw, **extra)
        for e in env:
            os.environ.pop(e)
        if extra.get("submit"):
            jobs[tag] = g
            continue
        for k in names:
            if extra.get("out_dtype") is np.float32:   # (float) of the double-precision call, bit for bit
                same = g[k].dtype == np.float32 and np.array_equal(g[k][:, ~mask].view(np.int32),
                                                                   ref[k][:, ~mask].astype(np.float32).view(np.int32))
            else:
                same = np.array_equal(g[k][:, ~mask].view(np.int64), ref[k][:, ~mask].view(np.int64))
            if not same:
                msgs.append(f"{tag}: {k} differs")
            if not np.isnan(g[k][:, mask]).all():
                msgs.append(f"{tag}: NA cells of {k} are not NaN")
        for k in g["diag"]:
            if not np.array_equal(np.asarray(g["diag"][k]).view(np.int64), np.asarray(ref["diag"][k]).view(np.int64)):
                msgs.append(f"{tag}: basin totals of {k} differ")
    for tag, job in jobs.items():  # both were in the library at the same time
        g = job.wait()
        for k in names:
            if not np.array_equal(g[k][:, ~mask].view(np.int64), ref[k][:, ~mask].view(np.int64)):
                msgs.append(f"{tag}: {k} differs")
        for k in g["diag"]:
            if not np.array_equal(np.asarray(g["diag"][k]).view(np.int64), np.asarray(ref["diag"][k]).view(np.int64)):
                msgs.append(f"{tag}: basin totals of {k} differ")
    if nt > 366:  # round 5 (ABI 5): the series in two time chunks, cut at a random month start after the first year
        starts = np.flatnonzero(np.r_[False, mid[1:] != mid[:-1]])
        cut = int(rng.choice(starts[starts >= 365]))
        tt = np.maximum(api.snow_threshold(tcm[:cut], st["lat"], elev), api.snow_threshold(tcm[cut:], st["lat"], elev))
        blk = dict(max_cells_per_launch=int(rng.integers(64, 400)))
        kwc = dict(kw, want_aux=True)
        a = api.splash_grid(swm[:cut], tcm[:cut], pnm[:cut], *args, **dict(kwc, time_index=dates[:cut]), tt=tt, **blk)
        b = api.splash_grid(swm[cut:], tcm[cut:], pnm[cut:], *args, **dict(kwc, time_index=dates[cut:]), tt=tt,
                            state0=a["state_end"], ai0=a["aux"]["AI"], **blk)
        for k in names:
            got = np.concatenate([a[k], b[k]])
            if got.shape != ref[k].shape or not np.array_equal(got[:, ~mask].view(np.int64), ref[k][:, ~mask].view(np.int64)):
                msgs.append(f"two time chunks (cut at day {cut}): {k} differs")
        if not np.array_equal(np.concatenate([a["diag_acc"], b["diag_acc"]], axis=1), ref["diag_acc"]):
            msgs.append("two time chunks: water-balance limbs differ")
        if not np.array_equal(b["state_end"][:, ~mask].view(np.int64), ref["state_end"][:, ~mask].view(np.int64)):
            msgs.append("two time chunks: final state differs")
    g32 = api.splash_grid(swm.astype(np.float32), tcm.astype(np.float32), pnm.astype(np.float32), *args, **kw)
    for k in names:  # the synthetic forcing is float-representable: single-precision transport is lossless
        if not np.array_equal(g32[k][:, ~mask].view(np.int64), ref[k][:, ~mask].view(np.int64)):
            msgs.append(f"f32 transport: {k} differs")
    bad += len(msgs)
    print(f"case {case}: {H}x{W} x {nt} d, {int(mask.sum())} NA cells, monthly={monthly}: {'ok' if not msgs else msgs[:4]}", flush=True)
print("problems", bad)
sys.exit(1 if bad else 0)
