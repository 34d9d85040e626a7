"""How long is the spin-up's chain when the long job has a wavefront to ITSELF?  (round 4 experiment)

In the tail launch the jobs are packed 64 to a wavefront, and a wavefront executes the union of its lanes'
branches (finished lanes keep stepping).  If a long job alone executed markedly fewer instructions per day
step, a final "solo" stage for the few hundred longest jobs would shorten the chain.  This measures it:
  1. a full spin-up with the measurement build (-DSPL_TAIL_STATS) tells which cells run >= 900 tail iterations;
  2. spin_up_begin again, then the pending list in the workspace is REPLACED by (a) one such cell, (b) 64 such
     cells (one wavefront of long jobs), (c) left as it is; spin_up_finish is timed for each.
    SPLASH_HIP_LIBRARY=gpurun_ab/lib_diff.so python tools/solo_tail.py [rows=1024] [row0=1536]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rsplash_amd import api

dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
r0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
W, nt, SEED = 4096, 365, 20221130
n = H * W
cal = api.Calendar.from_dates(np.datetime64("2001-01-01") + np.arange(nt))
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(SEED, 4096, W, r0 * W, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(SEED, cal, 0, nt, r0 * W, n, lat, elev, sw, tc, pn, n)
par0 = torch.empty(api.NPAR, n, **f64)
Tt = torch.empty(n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par0)
api.dev_snow_threshold(n, nt, tc, n, par0, n, Tt)
al = lambda b: (b + 255) & ~255
o_meta = al(n * 4) * 2
o_cnt = al(n * 4) * 3 + al(n * 8)
work = torch.zeros(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)


def begin():
    par, state = par0.clone(), torch.empty(api.NSTATE, n, **f64)
    api.dev_spin_up_begin(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
    torch.cuda.synchronize()
    return par, state


par, state = begin()
t0 = time.perf_counter()
api.dev_spin_up_finish(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
torch.cuda.synchronize()
t_full = time.perf_counter() - t0
meta = work[o_meta:o_meta + 4 * n].view(torch.int32).cpu().numpy()
it = np.where(meta <= -100000, -(meta + 100000) - 2, 0)
long_cells = np.nonzero(it >= int(os.environ.get("LONG_IT", "900")))[0]
cnt = work[o_cnt:o_cnt + 64].view(torch.int32)
n_pend = int(cnt[6].item())
print(f"{n} cells, {n_pend} jobs in the tail, {len(long_cells)} of them run >= 900 iterations; the whole tail: {t_full * 1e3:.1f} ms", flush=True)
ref_state = state.clone()


def tail_with(cells, label):
    par, state = begin()
    lst = torch.tensor(np.asarray(cells, dtype=np.int32), device=dev)
    work[0:4 * len(cells)].view(torch.int32).copy_(lst)
    cnt[6] = len(cells)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    api.dev_spin_up_finish(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = torch.equal(state[:, lst.long()].view(torch.int64), ref_state[:, lst.long()].view(torch.int64))
    m = work[o_meta:o_meta + 4 * n].view(torch.int32)[lst.long()].cpu().numpy()
    its = -(m + 100000) - 2
    print(f"{label}: {dt * 1e3:.1f} ms, iterations {its.min()}..{its.max()}, same bits as in the full tail: {ok}", flush=True)
    return dt


if os.environ.get("SOLO_WHERE"):
    # per job: when it ended (ms after its wavefront started) and on which XCC / SE / CU / SIMD it ran
    import ctypes as C
    from rsplash_amd._lib import lib
    dbg = torch.full((8, n), -1.0, dtype=torch.float32, device=dev)
    lib.splash_dbg_set_diff.argtypes = [C.c_void_p]
    assert lib.splash_dbg_set_diff(dbg.data_ptr()) == 0
    os.environ["SPLASH_HIP_TAIL_STAGES"] = "1,0"
    for m in (16, 64):
        cells = [int(c) for c in long_cells[:m]]
        tail_with(cells, f"{m} jobs")
        d = dbg[:, torch.tensor(cells, device=dev)].cpu().numpy()
        for j, c in enumerate(cells):
            hw = int(d[7, j])
            print(f"   job {j:2d} cell {c}: ended at {d[6, j]:8.1f} ms; xcc {hw >> 16} se {(hw >> 13) & 7} cu {(hw >> 8) & 15} simd {(hw >> 4) & 3} wave {hw & 15}")
    sys.exit(0)
if os.environ.get("PACKED_WHERE"):
    # the round-3 tail (one packed stage) over the whole list: when does each WORKGROUP end, and where did it run?
    import ctypes as C
    from rsplash_amd._lib import lib
    dbg = torch.full((8, n), -1.0, dtype=torch.float32, device=dev)
    lib.splash_dbg_set_diff.argtypes = [C.c_void_p]
    assert lib.splash_dbg_set_diff(dbg.data_ptr()) == 0
    os.environ["SPLASH_HIP_NO_SOLO_TAIL"] = "1"
    par, state = begin()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    api.dev_spin_up_finish(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
    torch.cuda.synchronize()
    print(f"packed tail over {n_pend} jobs: {1e3 * (time.perf_counter() - t0):.1f} ms")
    d = dbg[:, :n_pend].cpu().numpy()
    for g in range((n_pend + 255) // 256):
        seg = d[4, 256 * g:256 * (g + 1)]
        hw = int(d[5, 256 * g:256 * (g + 1)].max())
        per_wave = [float(seg[64 * w:64 * (w + 1)].max()) for w in range(4) if seg[64 * w:64 * (w + 1)].size]
        print(f"   workgroup {g:2d}: last job of each wavefront ends at {[round(x) for x in per_wave]} ms; xcc {hw >> 16} se {(hw >> 13) & 7} cu {(hw >> 8) & 15}")
    sys.exit(0)
if os.environ.get("SOLO_SAME"):
    # the SAME job on 64 wavefronts (16 workgroups) at once: is the speed a property of the job or of the place?
    import ctypes as C
    from rsplash_amd._lib import lib
    dbg = torch.full((8, n), -1.0, dtype=torch.float32, device=dev)
    lib.splash_dbg_set_diff.argtypes = [C.c_void_p]
    assert lib.splash_dbg_set_diff(dbg.data_ptr()) == 0
    os.environ["SPLASH_HIP_TAIL_STAGES"] = "1,0"
    lanes_list = [int(x) for x in os.environ.get("ACTIVE_LANES", "0").split(",")]
    for which in ([0] * len(lanes_list) if len(lanes_list) > 1 else (0, 12)):
        if len(lanes_list) > 1:   # measurement build: only that many lanes of a spread wavefront stay active
            os.environ["SPLASH_DBG_ACTIVE_LANES"] = str(lanes_list.pop(0))
            print("active lanes per wavefront:", os.environ["SPLASH_DBG_ACTIVE_LANES"])
        cells = [int(long_cells[which])] * 64
        par, state = begin()
        lst = torch.tensor(np.asarray(cells, dtype=np.int32), device=dev)
        work[0:4 * len(cells)].view(torch.int32).copy_(lst)
        cnt[6] = len(cells)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        api.dev_spin_up_finish(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
        torch.cuda.synchronize()
        print(f"64 copies of job {which}: {1e3 * (time.perf_counter() - t0):.1f} ms")
        mm = work[o_meta:o_meta + 4 * n].view(torch.int32)[cells[0]].item()
        print("   iterations executed in the spread stage:", -(mm + 100000) - 2)
        d = dbg[:, :64].cpu().numpy()
        for g in range(16):
            hw = int(d[5, 4 * g])
            print(f"   workgroup {g:2d}: ends {d[4, 4 * g:4 * g + 4].round(0).tolist()} ms; xcc {hw >> 16} se {(hw >> 13) & 7} cu {(hw >> 8) & 15}")
    sys.exit(0)
if os.environ.get("SOLO_REPEAT"):
    os.environ["SPLASH_HIP_TAIL_STAGES"] = "1,0"
    for rep in range(10):
        tail_with([int(long_cells[0])], f"run {rep}: the same ONE job")
    for rep in range(4):
        tail_with([int(c) for c in long_cells[:16]], f"run {rep}: the same 16 jobs")
    sys.exit(0)
if os.environ.get("SOLO_PAD"):
    # long jobs at every 4th list position, short ones (done within 3 iterations) between: one long-running wavefront
    # per workgroup -- against the same long jobs side by side (four to a workgroup, on the four SIMDs of one CU)
    os.environ["SPLASH_HIP_TAIL_STAGES"] = "1,0"
    short = [int(c) for c in np.nonzero((it >= 1) & (it <= 3))[0]]
    for m in (4, 16, 64):
        padded = []
        for j in range(m):
            padded += [int(long_cells[j])] + short[3 * j:3 * j + 3]
        tail_with(padded, f"{m} long jobs, one per workgroup (padded with short ones)")
        tail_with([int(c) for c in long_cells[:m]], f"{m} long jobs side by side")
    sys.exit(0)
if os.environ.get("SOLO_SWEEP"):
    os.environ["SPLASH_HIP_TAIL_STAGES"] = "1,0"
    for m in (1, 4, 16, 64, 128, 236, 700):
        cells = list(long_cells[:m]) if m <= len(long_cells) else list(long_cells) + [int(c) for c in np.nonzero((it > 0) & (it < 900))[0][:m - len(long_cells)]]
        tail_with(cells, f"{len(cells)} jobs, one packed pass then spread")
    sys.exit(0)
for c in long_cells[:3]:
    tail_with([int(c)], f"ONE long job (cell {int(c)}) on its own wavefront")
tail_with(long_cells[:64], "64 long jobs = one wavefront")
tail_with(long_cells[:8], "8 long jobs on one wavefront")
tail_with(long_cells[:min(len(long_cells), 1024)], f"{min(len(long_cells), 1024)} long jobs packed 64 to a wavefront")
