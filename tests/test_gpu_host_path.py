"""GPU suite, the host-buffer path: splash_grid_run / _submit / _wait (rsplash_amd/csrc/splash_grid_host.hip) and the
spin-up's workspace -- blocking, lanes, the straggler pool, page mapping, single-precision transport, out-of-memory:
whatever the plan, the bits of the one-block call (R/splash.grid.R:260-314, 359-400 is what stands there in the reference).
"""
import os

import numpy as np
import pytest

from conftest import ROOT, rel_err
from gpu_pipeline import (RTOL, assert_one_step, assert_trajectory, assert_whole_pipeline, close, device_pipeline,  # noqa: F401
                          synth_grid as _synth)

pytestmark = pytest.mark.gpu


def test_spin_up_blocking_and_cache_overflow_give_the_same_bits():
    """The spin-up's launches depend on the capacity of the day-term cache: phase 0 walks the grid in
    blocks of it (a fill launch + a read-back launch per block), the later phases send the jobs beyond
    it through the general kernel.  With a capacity of 128 slots (SPLASH_HIP_SPIN_CACHE_CAP, read once
    per process: hence the child processes) a 64 x 128 grid exercises every one of those paths; states,
    aridity index and pass counts must be bit-identical to the default layout, in both pass modes."""
    import subprocess, sys, tempfile
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from rsplash_amd import api
dev = torch.device("cuda:0")
H, W, nt = 64, 128, 365
n = H * W
cal = api.Calendar.from_dates(np.datetime64("2001-01-01") + np.arange(nt))
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(777000, H, W, 0, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(777000, cal, 0, nt, 0, n, lat, elev, sw, tc, pn, n)
par0 = torch.empty(api.NPAR, n, **f64); Tt = torch.empty(n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par0)
api.dev_snow_threshold(n, nt, tc, n, par0, n, Tt)
out = {}
for mode in ("fast", "passes"):
    par, state = par0.clone(), torch.empty(api.NSTATE, n, **f64)
    work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
    passes = torch.zeros(2, n, dtype=torch.int32, device=dev) if mode == "passes" else None
    api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work, passes=passes)
    torch.cuda.synchronize()
    out[mode + "_state"] = state.cpu().numpy()
    out[mode + "_ai"] = par[15].cpu().numpy()
    if passes is not None: out["passes"] = passes.cpu().numpy()
np.savez(sys.argv[1], **out)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for tag, cap in (("default", None), ("cap128", "128")):
            env = dict(os.environ)
            env.pop("SPLASH_HIP_SPIN_CACHE_CAP", None)
            if cap:
                env["SPLASH_HIP_SPIN_CACHE_CAP"] = cap
            f = os.path.join(d, tag + ".npz")
            subprocess.run([sys.executable, "-c", code, f], env=env, check=True)
            res[tag] = dict(np.load(f))
    a, b = res["default"], res["cap128"]
    for k in a:
        assert np.array_equal(a[k].view(np.int64) if a[k].dtype == np.float64 else a[k],
                              b[k].view(np.int64) if b[k].dtype == np.float64 else b[k]), k
    assert np.array_equal(a["fast_state"].view(np.int64), a["passes_state"].view(np.int64))
    assert a["passes"][1].max() == 1000 and a["passes"][1].min() >= 1   # the tail is exercised


def test_read_back_never_addresses_a_slot_beyond_the_cache():
    """the read-back kernel formed record addresses from slots the cache does not have (lanes
    beyond the capacity are `done` but step along with their wavefront; with records laid out job by job
    that was (slot - cap) x 58 KB past the workspace, harmless only because torch's segment was mapped).
    The checking build of the library (librsplash_hip_check.so: the same sources with -DSPL_CHECK_SLOTS)
    reports every access to the record cache with its slot and the capacity the launch was given.  A
    capacity of 128 compact = 64 rich slots on 64 low-latitude rows of the benchmark grid (262 144 cells,
    where the long chains are densest) gives every later phase a list far longer
    than the cache (packed launches) and the tail a list longer than 64 jobs (spread launches: before the
    fix they dealt ALL jobs of the list); no access may name a slot >= the capacity, and the states must
    equal the product library's with its default layout, bit for bit."""
    import subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    chk = os.path.join(root, "rsplash_amd", "librsplash_hip_check.so")
    import rsplash_amd
    rsplash_amd.build_check_library()   # `make check`: a no-op when the prebuilt file is current
    assert os.path.exists(chk), "librsplash_hip_check.so is built by rsplash_amd.build_check_library() (make check)"
    code = r'''
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, %r)
from rsplash_amd import api
from rsplash_amd._lib import lib
dev = torch.device("cuda:0")
H, W, nt = 4096, 4096, 365
cell0, n = 1800 * W, 64 * W
cal = api.Calendar.from_dates(np.datetime64("2001-01-01") + np.arange(nt))
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(20221130, H, W, cell0, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(20221130, cal, 0, nt, cell0, n, lat, elev, sw, tc, pn, n)
par = torch.empty(api.NPAR, n, **f64); Tt = torch.empty(n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par)
api.dev_snow_threshold(n, nt, tc, n, par, n, Tt)
state = torch.empty(api.NSTATE, n, **f64)
# the workspace is the LAST bytes of its allocation (exact size, nothing of ours behind the cache)
work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
api.dev_spin_up_begin(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
pending = api.dev_spin_up_pending_count(work, n)
api.dev_spin_up_finish(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
torch.cuda.synchronize()
out = dict(state=state.cpu().numpy(), ai=par[15].cpu().numpy(), pending=np.array([pending]))
if hasattr(lib, "splash_dbg_slot_check"):
    v = (C.c_ulonglong * 2)()
    lib.splash_dbg_slot_check.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    assert lib.splash_dbg_slot_check(v, 0) == 0
    out["slots"] = np.array([v[0], v[1]], dtype=np.uint64)
np.savez(sys.argv[1], **out)
''' % root
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for tag, lib_, cap in (("product", None, None), ("check", chk, "128")):
            env = dict(os.environ)
            env.pop("SPLASH_HIP_SPIN_CACHE_CAP", None)
            env.pop("SPLASH_HIP_LIBRARY", None)
            if lib_:
                env["SPLASH_HIP_LIBRARY"] = lib_
            if cap:
                env["SPLASH_HIP_SPIN_CACHE_CAP"] = cap
            f = os.path.join(d, tag + ".npz")
            subprocess.run([sys.executable, "-c", code, f], env=env, check=True)
            res[tag] = dict(np.load(f))
    a, b = res["product"], res["check"]
    assert "slots" in b and "slots" not in a          # the checking build ran where it was meant to
    accesses, beyond = int(b["slots"][0]), int(b["slots"][1])
    assert accesses > 10_000, accesses                 # the cached paths were exercised
    assert int(b["pending"][0]) > 64, b["pending"]     # a tail list longer than the 64 rich slots
    assert beyond == 0, f"{beyond} record accesses named a slot beyond the cache's capacity"
    for k in ("state", "ai"):
        assert np.array_equal(a[k].view(np.int64), b[k].view(np.int64)), k
    assert int(a["pending"][0]) == int(b["pending"][0])


@pytest.mark.gpu
def test_overlapped_spin_up_gives_the_same_bits(api, oracle):
    """splash_dev_spin_up_begin / _finish with the transient run of the finished cells in between
    (skip mask) and the late cells in list mode afterwards == the plain sequence, bit for bit, water-
    balance totals included (what splash_grid_run does to hide the spin-up's tail)."""
    import torch
    dev = torch.device("cuda:0")
    seed, H, W, nt = 777000, 64, 128, 400
    n = H * W
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    cal = api.Calendar.from_dates(dates)
    f64 = dict(dtype=torch.float64, device=dev)
    lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
    soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
    api.dev_synth_statics(seed, H, W, 0, n, 1, lat, elev, slop, asp, res, soil, au)
    sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
    api.dev_synth_forcing(seed, cal, 0, nt, 0, n, lat, elev, sw, tc, pn, n)
    par0 = torch.empty(api.NPAR, n, **f64)
    Tt = torch.empty(n, **f64)
    api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par0)
    api.dev_snow_threshold(n, nt, tc, n, par0, n, Tt)
    names = ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "sm_lim")
    nm = cal.nmonth

    def plain():
        par, state = par0.clone(), torch.empty(api.NSTATE, n, **f64)
        work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
        api.dev_spin_up(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
        out = {k: torch.empty(nt, n, **f64) for k in names}
        acc = torch.zeros(1, nm, api.DIAG_REC, dtype=torch.int64, device=dev)
        api.dev_run_all(cal, 0, nt, n, sw, tc, pn, n, Tt, par, n, state, n, out, n, diag_acc=acc)
        torch.cuda.synchronize()
        return out, state, acc, par

    def overlapped():
        par, state = par0.clone(), torch.empty(api.NSTATE, n, **f64)
        work = torch.empty(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
        s1, s2 = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
        api.dev_spin_up_begin(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work, stream=s1.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(s1)
        mask_p, cells_p, _ = api.dev_spin_up_pending(work, n)
        out = {k: torch.empty(nt, n, **f64) for k in names}
        acc = torch.zeros(1, nm, api.DIAG_REC, dtype=torch.int64, device=dev)
        api.dev_run_all(cal, 0, nt, n, sw, tc, pn, n, Tt, par, n, state, n, out, n, diag_acc=acc,
                        skip=mask_p, stream=s1.cuda_stream)
        s2.wait_event(ev)
        api.dev_spin_up_finish(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work, stream=s2.cuda_stream)
        torch.cuda.synchronize()
        n_late = api.dev_spin_up_pending_count(work, n, s1.cuda_stream)
        assert 0 < n_late < n // 20
        cells = torch.empty(n_late, dtype=torch.int32, device=dev)
        import ctypes
        # the list lives in the workspace: view it through a tensor of the workspace's bytes
        off = cells_p - work.data_ptr()
        cells = work[off:off + 4 * n_late].view(torch.int32).clone()
        late = {k: torch.empty(nt, n_late, **f64) for k in names}
        api.dev_run_all(cal, 0, nt, n_late, sw, tc, pn, n, Tt, par, n, state, n, late, n_late,
                        diag_acc=acc, cells=cells_p)
        torch.cuda.synchronize()
        idx = cells.long()
        for k in names:
            out[k][:, idx] = late[k]
        return out, state, acc, par, n_late

    a_out, a_state, a_acc, a_par = plain()
    b_out, b_state, b_acc, b_par, n_late = overlapped()
    print("late cells:", n_late)
    for k in names:
        assert torch.equal(a_out[k].view(torch.int64), b_out[k].view(torch.int64)), k
    assert torch.equal(a_state.view(torch.int64), b_state.view(torch.int64))
    assert torch.equal(a_acc, b_acc) and torch.equal(a_par.view(torch.int64), b_par.view(torch.int64))
    cal.close()


@pytest.mark.gpu
def test_host_path_pipeline_lanes_pool_and_jobs_give_the_same_bits(api, oracle):
    """splash_grid_run pipelines its blocks over two lanes (host thread + stream + buffers each) and
    finishes the stragglers of ALL blocks once, from a pool (splash_capi.hip).  Whatever the blocking,
    the number of lanes, a pool too small for a block's stragglers (they are then finished in place),
    a masked grid, or two calls in flight at once (splash_grid_submit/_wait): the same bits as the
    one-block call -- results, end state, aridity index, pass counts and the exact totals."""
    import os
    H, W, nt = 48, 128, 400
    n = H * W
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    st = oracle.synth_statics(777000, H, W, 0, n, 1)
    sw, tc, pn = oracle.synth_forcing(777000, H, W, 0, n, 0, doy, st["lat"], st["elev"])
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    basin = (np.arange(n) // (n // 5)).astype(np.int32).clip(0, 4)

    def call(elev=None, **kw):
        return api.splash_grid(sw, tc, pn, st["elev"] if elev is None else elev, st["soil"], st["lat"], st["slop"],
                               st["asp"], st["au"], st["resolution"], time_index=dates,
                               sim_control=dict(monthly_out=False), want_aux=True, want_diag=True,
                               basin=basin, n_basin=5, **kw)

    def same(a, b, what):
        for k in ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "sm_lim", "state_end"):
            assert np.array_equal(a[k].view(np.int64), b[k].view(np.int64)), (what, k)
        for k in ("Tt", "AI", "passes1", "passes2"):
            assert np.array_equal(a["aux"][k].view(np.int64), b["aux"][k].view(np.int64)), (what, k)
        assert np.array_equal(a["diag_acc"], b["diag_acc"]), what

    ref = call()
    n_late = int((ref["aux"]["passes2"] > 28).sum())   # jobs the first seven phases (28 passes) leave
    assert n_late >= 3, n_late
    knobs = ("SPLASH_HIP_LANES", "SPLASH_HIP_POOL_CAP", "SPLASH_HIP_PREFAULT", "SPLASH_HIP_PREFAULT_MIN_BYTES",
             "SPLASH_HIP_REGISTER_MIN_BYTES", "SPLASH_HIP_BLOCKS_PER_LANE", "SPLASH_HIP_MIN_BLOCK_CELLS")
    env0 = {k: os.environ.get(k) for k in knobs}
    for k in knobs:
        os.environ.pop(k, None)
    try:
        same(ref, call(max_cells_per_launch=1024), "6 blocks on 2 lanes")
        # The code that only runs for BIG calls, at test size: the result
        # pages mapped per block while its kernels run (>= 64 MiB per block by default), caller arrays
        # page-locked by a helper thread while the first blocks load and compute (>= 8 GiB of results by
        # default), and the library's own plan of four blocks per lane (grids > 2^20 cells by default).
        # Outputs are handed over filled with a sentinel: every value must have been delivered.
        os.environ["SPLASH_HIP_PREFAULT_MIN_BYTES"] = "0"
        os.environ["SPLASH_HIP_REGISTER_MIN_BYTES"] = "0"
        os.environ["SPLASH_HIP_MIN_BLOCK_CELLS"] = "512"
        nt_out = nt
        for mode in ("populate", "touch"):
            os.environ["SPLASH_HIP_PREFAULT"] = mode
            bufs = {k: np.full((nt_out, n), -12345.678) for k in ("wn", "ro", "pet", "aet", "snow", "cond", "bflow",
                                                                   "netr", "sm_lim")}
            g = call(out=bufs)    # the library's own blocking: 2 lanes x 4 blocks of 768 cells
            same(ref, g, f"prefault={mode} + registered arrays + 4 blocks per lane")
            assert all(g[k] is bufs[k] for k in bufs)
        os.environ["SPLASH_HIP_BLOCKS_PER_LANE"] = "6"
        os.environ["SPLASH_HIP_LANES"] = "3"
        same(ref, call(), "prefault + registered arrays, 3 lanes x 6 blocks")
        # round 6: three workers on the one GPU -- 6 144 cells are dealt as 24 chunks of 256, eight to a worker, so a
        # worker's blocks are SEVERAL runs of consecutive cells: the dense fast paths (strided copies from and to the
        # caller's arrays, page mapping per run, page-locked arrays) loop over them (for_runs), on two lanes with the pool
        os.environ["SPLASH_HIP_LANES"] = "2"
        os.environ["SPLASH_HIP_BLOCKS_PER_LANE"] = "2"
        bufs = {k: np.full((nt_out, n), -12345.678) for k in ("wn", "ro", "pet", "aet", "snow", "cond", "bflow",
                                                               "netr", "sm_lim")}
        same(ref, call(out=bufs, devices=[0, 0, 0]), "three workers, chunks dealt round-robin, blocks of several runs")
        os.environ["SPLASH_HIP_LANES"] = "3"
        for k in ("SPLASH_HIP_PREFAULT", "SPLASH_HIP_PREFAULT_MIN_BYTES", "SPLASH_HIP_REGISTER_MIN_BYTES",
                  "SPLASH_HIP_BLOCKS_PER_LANE", "SPLASH_HIP_MIN_BLOCK_CELLS"):
            os.environ.pop(k)
        os.environ["SPLASH_HIP_LANES"] = "2"
        os.environ["SPLASH_HIP_LANES"] = "1"
        same(ref, call(max_cells_per_launch=1024), "6 blocks on 1 lane")
        os.environ["SPLASH_HIP_LANES"] = "3"
        same(ref, call(max_cells_per_launch=700), "9 ragged blocks on 3 lanes")
        os.environ["SPLASH_HIP_LANES"] = "2"
        os.environ["SPLASH_HIP_POOL_CAP"] = "2"       # fewer slots than stragglers: the fallback path
        same(ref, call(max_cells_per_launch=1024), "pool overflow")
        same(ref, call(), "pool overflow, one block")
        # several overflowing blocks per lane: the lane's buffer for late cells is allocated once and
        # goes back to the cache with the rest (it was re-allocated per block and the earlier
        # ones stayed busy for the life of the process) -- after the call everything held is idle
        os.environ["SPLASH_HIP_LANES"] = "4"          # and four lanes reserve pool slots at once (CAS)
        same(ref, call(max_cells_per_launch=256), "pool overflow, 24 blocks on 4 lanes")
        os.environ["SPLASH_HIP_LANES"] = "2"
        same(ref, call(max_cells_per_launch=512), "pool overflow, 6 blocks per lane")
        held, idle = api.cache_stats(0)
        assert held == idle and held > 0, (held, idle)
        os.environ.pop("SPLASH_HIP_POOL_CAP")
        # two calls in flight on the same device, each with its share of the memory
        j1 = call(max_cells_per_launch=2048, submit=True, mem_fraction=0.4)
        j2 = call(max_cells_per_launch=1500, submit=True, mem_fraction=0.4)
        same(ref, j2.wait(), "second job in flight")
        same(ref, j1.wait(), "first job in flight")
        # a masked grid (NA cells are compacted on the host by several threads): valid cells unchanged
        elev = st["elev"].copy()
        elev[::7] = np.nan
        m = call(elev=elev, max_cells_per_launch=900)
        ok = ~np.isnan(elev)
        for k in ("wn", "ro", "aet", "snow", "sm_lim"):
            assert np.array_equal(m[k][:, ok].view(np.int64), ref[k][:, ok].view(np.int64)), k
            assert np.isnan(m[k][:, ~ok]).all(), k
    finally:
        for k, v in env0.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.gpu
def test_repeated_multi_block_calls_keep_their_block_size(api, oracle):
    """The host-buffer entry points keep their device buffers between calls.  Those idle buffers are
    not 'free' for hipMemGetInfo: a call that sized its blocks from the free figure alone shrank with
    every call after the first multi-block one until it ran out of memory.  Idle pooled bytes count
    as free now: the same call repeated under a tight memory share plans the same blocks each time
    (SPLASH_TRACE prints the plan) and returns the same bits."""
    import os
    import re
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import os, sys, numpy as np
        sys.path.insert(0, %r)
        from rsplash_amd import api
        from oracle.binding import Oracle
        o = Oracle()
        H, W, nt = 32, 128, 365
        doy, year, month, mid, nm = o.calendar(2001, nt)
        st = o.synth_statics(777000, H, W, 0, H * W, 1)
        sw, tc, pn = o.synth_forcing(777000, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
        dates = np.datetime64("2001-01-01") + np.arange(nt)
        ref = None
        for rep in range(4):
            g = api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                                st["resolution"], time_index=dates, sim_control=dict(monthly_out=False),
                                mem_fraction=float(os.environ["FRAC"]))
            if ref is None:
                ref = g
            assert np.array_equal(ref["wn"].view(np.int64), g["wn"].view(np.int64))
        print("done")
    """) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # a pretend device of 0.7 GB (SPLASH_HIP_FAKE_DEVICE_BYTES: free = that minus what the pool holds): the
    # 4096-cell grid needs ~0.33 GB for the straggler pool and ~0.65 GB per lane in one block
    env = dict(os.environ, SPLASH_TRACE="1", FRAC="0.85", SPLASH_HIP_FAKE_DEVICE_BYTES=str(700 * 1000 * 1000))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "done" in r.stdout, r.stderr[-3000:]
    plans = re.findall(r"(\d+) cells in (\d+) blocks of <= (\d+) on", r.stderr)
    assert len(plans) == 4, r.stderr[-2000:]
    assert len(set(plans)) == 1, plans          # the same plan every time
    assert int(plans[0][1]) >= 2, plans         # and it is a multi-block plan


@pytest.mark.gpu
def test_host_path_out_of_memory_is_an_error_not_a_hang(api, oracle):
    """A device too small for the call (a pretend one: SPLASH_HIP_FAKE_DEVICE_BYTES) makes splash_grid_run
    return SPLASH_ENOMEM with a message -- from the blocking call and from submit/wait alike -- and the
    next call on the real device works."""
    import os
    from rsplash_amd._lib import SplashError
    H, W, nt = 8, 64, 365
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    st = oracle.synth_statics(777000, H, W, 0, H * W, 1)
    sw, tc, pn = oracle.synth_forcing(777000, H, W, 0, H * W, 0, doy, st["lat"], st["elev"])
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    args = (sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"])
    from rsplash_amd import _lib
    _lib.lib.splash_hip_release_cache()  # the pretend device starts empty (buffers kept by earlier tests would
                                          # count as its idle memory)
    os.environ["SPLASH_HIP_FAKE_DEVICE_BYTES"] = str(20 * 1000 * 1000)
    try:
        with pytest.raises(SplashError, match="memory"):
            api.splash_grid(*args, time_index=dates)
        job = api.splash_grid(*args, time_index=dates, submit=True)
        with pytest.raises(SplashError, match="memory"):
            job.wait()
    finally:
        os.environ.pop("SPLASH_HIP_FAKE_DEVICE_BYTES")
    g = api.splash_grid(*args, time_index=dates)
    assert np.isfinite(g["wn"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("where,devices", [(2, None), (3, [0, 0]), (1, None)])
def test_a_cpp_exception_on_any_thread_of_a_call_becomes_its_error(api, oracle, monkeypatch, where, devices):
    """No C++ exception leaves the C ABI, and none ends the process from a helper thread: thrown
    (SPLASH_HIP_TEST_THROW, fault injection) in the body of grid_run (1), on a lane's thread (2: the other lanes
    drain the block list, the straggler pool's wait ends) or on a device worker's thread (3), the call returns an
    error code and a message -- from the blocking call and from submit/wait -- and the next call works."""
    from rsplash_amd._lib import SplashError
    H, W, nt = 8, 64, 365
    g_ = _synth(oracle, H, W, nt, 1, seed=4242)
    st = g_["st"]
    args = (g_["sw"], g_["tc"], g_["pn"], st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"], st["resolution"])
    kw = dict(time_index=g_["dates"], devices=devices)
    ref = api.splash_grid(*args, **kw)
    monkeypatch.setenv("SPLASH_HIP_TEST_THROW", str(where))
    pat = "out of host memory" if where == 1 else "injected on a"
    with pytest.raises(SplashError, match=pat):
        api.splash_grid(*args, **kw)
    job = api.splash_grid(*args, submit=True, **kw)
    with pytest.raises(SplashError, match=pat):
        job.wait()
    monkeypatch.delenv("SPLASH_HIP_TEST_THROW")
    g = api.splash_grid(*args, **kw)
    for k in ("wn", "ro", "snow"):
        assert np.array_equal(g[k], ref[k], equal_nan=True), k


@pytest.mark.gpu
def test_prefault_touches_only_its_own_columns(api, oracle):
    """Regression test for the race fixed by inspection in ff89765 (round 3).  While a block's kernels
    run, its lane maps the caller's result pages.  The first version did that ONCE, from the first lane
    to arrive, over the whole worker range -- columns [idx[v0], idx[v0] + (v1 - v0)) of every row -- by
    storing a zero every 4 KiB, while the other lane's helper thread was delivering ITS block into the
    same arrays: a delivered value could be overwritten.  Now a block touches columns [c0, c0 + nb), its
    own, which nothing else writes before its own delivery (and by default it stores nothing at all:
    madvise(MADV_POPULATE_WRITE)).
    By construction, not by luck of timing: SPLASH_HIP_PREFAULT=touch selects the storing variant,
    SPLASH_HIP_PREFAULT_DELAY_MS makes every lane sleep before it touches, so that the blocks handed out
    before have been delivered by then.  With the old range (kept as fault injection behind
    SPLASH_HIP_TEST_PREFAULT_WHOLE_RANGE, nothing a caller would set) the same call MUST come back
    with overwritten values: the negative control shows that this test can see the bug."""
    import os
    H, W, nt = 48, 128, 365
    n = H * W
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    st = oracle.synth_statics(777000, H, W, 0, n, 1)
    sw, tc, pn = oracle.synth_forcing(777000, H, W, 0, n, 0, doy, st["lat"], st["elev"])
    dates = np.datetime64("2001-01-01") + np.arange(nt)
    names = ("wn", "ro", "pet", "aet", "snow")

    def call(**kw):
        bufs = {k: np.full((nt, n), -12345.678) for k in names}
        return api.splash_grid(sw, tc, pn, st["elev"], st["soil"], st["lat"], st["slop"], st["asp"], st["au"],
                               st["resolution"], time_index=dates, sim_control=dict(monthly_out=False),
                               outputs=names, out=bufs, want_state=False, **kw)

    knobs = ("SPLASH_HIP_PREFAULT", "SPLASH_HIP_PREFAULT_MIN_BYTES", "SPLASH_HIP_PREFAULT_DELAY_MS",
             "SPLASH_HIP_TEST_PREFAULT_WHOLE_RANGE", "SPLASH_HIP_LANES")
    env0 = {k: os.environ.get(k) for k in knobs}
    for k in knobs:
        os.environ.pop(k, None)
    try:
        ref = call()                                   # one block, nothing touched (6 144 cells: below 64 MiB)
        os.environ.update(SPLASH_HIP_PREFAULT="touch", SPLASH_HIP_PREFAULT_MIN_BYTES="0",
                          SPLASH_HIP_PREFAULT_DELAY_MS="150", SPLASH_HIP_LANES="2")
        g = call(max_cells_per_launch=768)             # 8 blocks on 2 lanes
        for k in names:
            assert np.array_equal(g[k].view(np.int64), ref[k].view(np.int64)), k
        os.environ["SPLASH_HIP_TEST_PREFAULT_WHOLE_RANGE"] = "1"
        bad = call(max_cells_per_launch=768)
        n_diff = sum(int((bad[k].view(np.int64) != ref[k].view(np.int64)).sum()) for k in names)
        cols = np.unique(np.concatenate([np.nonzero((bad[k].view(np.int64) != ref[k].view(np.int64)).any(axis=0))[0]
                                         for k in names]))
        print("negative control: old range overwrote", n_diff, "delivered values, in columns", cols[:12], "...")
        assert n_diff > 0, "the fault injection did not reproduce the overwrite: the test has no teeth"
    finally:
        for k, v in env0.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.gpu
def test_single_precision_result_transport_is_the_rounded_double(api, oracle):
    """splash_grid_out.out_f32 (ABI 4, opt-in): the series arrive as floats rounded on the device.  Bit
    for bit (float) of what the double-precision call delivers -- dense and masked grids, daily and
    monthly, several blocks, the straggler pool and its overflow path (late cells are rounded on the
    host) -- and asking for one series in both precisions is refused."""
    import os
    H, W, nt = 24, 128, 400
    n = H * W
    doy, year, month, mid, nm = oracle.calendar(2001, nt)
    st = oracle.synth_statics(777000, H, W, 0, n, 1)
    sw, tc, pn = oracle.synth_forcing(777000, H, W, 0, n, 0, doy, st["lat"], st["elev"])
    dates = np.datetime64("2001-01-01") + np.arange(nt)

    def call(elev=None, monthly=False, **kw):
        return api.splash_grid(sw, tc, pn, st["elev"] if elev is None else elev, st["soil"], st["lat"], st["slop"],
                               st["asp"], st["au"], st["resolution"], time_index=dates,
                               sim_control=dict(monthly_out=monthly), **kw)

    def same32(a64, a32, what):
        for k in api.OUT_NAMES:
            assert a32[k].dtype == np.float32, k
            want = a64[k].astype(np.float32)
            assert np.array_equal(want.view(np.int32), a32[k].view(np.int32)), (what, k)
        assert np.array_equal(a64["state_end"].view(np.int64), a32["state_end"].view(np.int64)), what

    env0 = {k: os.environ.get(k) for k in ("SPLASH_HIP_POOL_CAP",)}
    try:
        for monthly in (False, True):
            ref = call(monthly=monthly)
            same32(ref, call(monthly=monthly, out_dtype=np.float32), f"one block, monthly={monthly}")
            same32(ref, call(monthly=monthly, out_dtype=np.float32, max_cells_per_launch=700),
                   f"ragged blocks, monthly={monthly}")
        # per series: some in single, some in double precision in the same call
        mix = call(monthly=True, out_dtype=dict(wn=np.float32, aet=np.float32, sm_lim=np.float32), max_cells_per_launch=700)
        for k in api.OUT_NAMES:
            if k in ("wn", "aet", "sm_lim"):
                assert mix[k].dtype == np.float32 and np.array_equal(ref[k].astype(np.float32).view(np.int32), mix[k].view(np.int32)), k
            else:
                assert mix[k].dtype == np.float64 and np.array_equal(ref[k].view(np.int64), mix[k].view(np.int64)), k
        os.environ["SPLASH_HIP_POOL_CAP"] = "2"
        same32(ref, call(monthly=True, out_dtype=np.float32, max_cells_per_launch=1024), "pool overflow")
        os.environ.pop("SPLASH_HIP_POOL_CAP")
        elev = st["elev"].copy()
        elev[::5] = np.nan
        m64, m32 = call(elev=elev, max_cells_per_launch=900), call(elev=elev, max_cells_per_launch=900,
                                                                  out_dtype=np.float32)
        ok = ~np.isnan(elev)
        for k in api.OUT_NAMES:
            assert np.array_equal(m64[k][:, ok].astype(np.float32).view(np.int32), m32[k][:, ok].view(np.int32)), k
            assert np.isnan(m32[k][:, ~ok]).all(), k
    finally:
        for k, v in env0.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    with pytest.raises(ValueError):       # a caller's array must have the dtype asked for
        call(out=dict(wn=np.empty((nt, n))), out_dtype=np.float32)
