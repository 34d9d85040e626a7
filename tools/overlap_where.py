"""Which wavefronts of the spin-up's last stage lose time when the transient run shares the chip?  (round 4)
bench.py's whole-call figure overlaps the tail (splash_dev_spin_up_finish on a high-priority stream) with the transient
run of the finished cells; the tail then takes 0.68 s or 0.85-0.9 s, run by run.  Measurement build
(-DSPL_TAIL_STATS -DSPL_DIFF_STATS): every job records when its wavefront started, when it ended and where it ran.
    SPLASH_HIP_LIBRARY=gpurun_ab/lib_diff.so python tools/overlap_where.py [reps=3]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rsplash_amd import api
from rsplash_amd._lib import lib

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W, nt, SEED = 4096, 4096, 365, 20221130
n = H * W
cal = api.Calendar.from_dates(np.datetime64("2001-01-01") + np.arange(nt))
f64 = dict(dtype=torch.float64, device=dev)
lat, elev, slop, asp, res = (torch.empty(n, **f64) for _ in range(5))
soil, au = torch.empty(6, n, **f64), torch.empty(3, n, **f64)
api.dev_synth_statics(SEED, H, W, 0, n, 1, lat, elev, slop, asp, res, soil, au)
sw, tc, pn = (torch.empty(nt, n, **f64) for _ in range(3))
api.dev_synth_forcing(SEED, cal, 0, nt, 0, n, lat, elev, sw, tc, pn, n)
par0 = torch.empty(api.NPAR, n, **f64)
Tt = torch.empty(n, **f64)
api.dev_soil_hydro(n, lat, elev, slop, asp, res, soil, au, par0)
del lat, elev, slop, asp, res, soil, au
api.dev_snow_threshold(n, nt, tc, n, par0, n, Tt)
al = lambda b: (b + 255) & ~255
o_meta = al(n * 4) * 2
work = torch.zeros(api.spin_up_work_size(n), dtype=torch.uint8, device=dev)
mask_p, cells_p, ncells_p = api.dev_spin_up_pending(work, n)
dbg = torch.full((8, n), -1.0, dtype=torch.float32, device=dev)
lib.splash_dbg_set_diff.argtypes = [C.c_void_p]
assert lib.splash_dbg_set_diff(dbg.data_ptr()) == 0
OUT5 = ("wn", "snow", "ro", "aet", "pet")
bn = 512 * W
outb = {k: torch.empty(nt, bn, **f64) for k in OUT5}
acc = torch.zeros(1, cal.nmonth, api.DIAG_REC, dtype=torch.int64, device=dev)
sp = torch.cuda.current_stream().cuda_stream
s2 = torch.cuda.Stream(device=dev, priority=-1)
state = torch.empty(api.NSTATE, n, **f64)


def one(overlap):
    par = par0.clone()
    dbg.fill_(-1.0)
    api.dev_spin_up_begin(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work, stream=sp)
    e_begin = torch.cuda.Event()
    e_begin.record()
    if overlap:
        for c0 in range(0, n, bn):
            api.dev_run_all(cal, 0, nt, bn, sw[:, c0:], tc[:, c0:], pn[:, c0:], n, Tt[c0:], par[:, c0:], n, state[:, c0:], n,
                            outb, bn, monthly=False, diag_acc=acc, stream=sp, skip=mask_p + c0)
    s2.wait_event(e_begin)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s2)
    api.dev_spin_up_finish(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work, stream=s2.cuda_stream)
    e1.record(s2)
    torch.cuda.synchronize()
    tail = e0.elapsed_time(e1)
    meta = work[o_meta:o_meta + 4 * n].view(torch.int32).cpu().numpy()
    it = np.where(meta <= -100000, -(meta + 100000) - 2, -1)   # iterations of the launch the job ended in
    d = dbg.cpu().numpy()
    long_ = np.nonzero(it >= 300)[0]                            # ended deep inside the last stage
    start, dur, hw = d[3, long_], d[6, long_], d[7, long_].astype(np.int64)
    us_per_pass = 1e3 * dur / (it[long_] + 1)
    q = np.percentile(us_per_pass, [0, 50, 90, 99, 100])
    print(f"{'overlapped' if overlap else 'tail alone'}: tail {tail:7.1f} ms; {len(long_)} jobs of >= 300 iterations in the last stage; "
          f"us per pass: min {q[0]:.1f} median {q[1]:.1f} p90 {q[2]:.1f} p99 {q[3]:.1f} max {q[4]:.1f}; "
          f"last-stage wavefronts start within {start.max() - start.min():.1f} ms of each other; latest end {np.max(start - start.min() + dur):.1f} ms after the first start", flush=True)
    # where the slow ones ran: SIMD = (xcc, se, cu, simd); how many long jobs share a CU / a SIMD with them
    simd = (hw >> 16) * 100000 + ((hw >> 13) & 7) * 10000 + ((hw >> 8) & 15) * 100 + ((hw >> 4) & 3)
    cu = simd // 100
    order = np.argsort(-us_per_pass)
    cu_count = {c: int((cu == c).sum()) for c in np.unique(cu)}
    simd_count = {c: int((simd == c).sum()) for c in np.unique(simd)}
    hist = {}
    for j in range(len(long_)):
        key = (simd_count[simd[j]], cu_count[cu[j]])
        hist.setdefault(key, []).append(us_per_pass[j])
    for key in sorted(hist):
        v = np.asarray(hist[key])
        print(f"     long jobs sharing their SIMD with {key[0] - 1} and their CU with {key[1] - 1} other long jobs: {len(v):4d}, us per pass median {np.median(v):.1f} max {v.max():.1f}")
    print("     slowest ten: " + ", ".join(f"{us_per_pass[j]:.0f} us (xcc {hw[j] >> 16} se {(hw[j] >> 13) & 7} cu {(hw[j] >> 8) & 15} simd {(hw[j] >> 4) & 3}, {it[long_[j]]} it)" for j in order[:10]), flush=True)
    return tail


for r in range(reps):
    one(False)
    one(True)
