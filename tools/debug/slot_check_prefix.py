"""Evidence for ADVICE r04 (medium): the slot counters of the checking build on the PRE-FIX indexing.
gpurun_ab/lib_oldchk.so = this tree's kernels with `cs_ld = cs` and without the spread clamp, built
with -DSPL_CHECK_SLOTS.  The workspace gets 64 MB of slack behind it so that the out-of-range reads of
the old indexing stay inside our own allocation (the run is safe); the counter reports them.
  SPLASH_HIP_LIBRARY=gpurun_ab/lib_oldchk.so SPLASH_HIP_SPIN_CACHE_CAP=128 python tools/debug/slot_check_prefix.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
work = torch.zeros(api.spin_up_work_size(n) + (64 << 20), dtype=torch.uint8, device=dev)
api.dev_spin_up_begin(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
pending = api.dev_spin_up_pending_count(work, n)
api.dev_spin_up_finish(cal, n, sw, tc, pn, n, Tt, par, n, state, n, work)
torch.cuda.synchronize()
v = (C.c_ulonglong * 2)()
lib.splash_dbg_slot_check.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert lib.splash_dbg_slot_check(v, 0) == 0
print(f"library {os.environ.get('SPLASH_HIP_LIBRARY', 'product')}: tail list {pending} jobs, "
      f"{v[0]} wavefront accesses to the record cache, {v[1]} lane accesses with a slot beyond the capacity")
