"""Host-side mirror of the reference's interface for the gridded SPLASH path.

  splash_point(...)  <->  R/splash.point.R:29   (same argument names / meaning / outputs)
  splash_grid(...)   <->  R/splash.grid.R:28    (arrays instead of Raster* objects)
  SPLASH(lat, elv)   <->  Rcpp module "splash_module", src/SPLASH.cpp:12-21
  Calendar, dev_*    <->  the device-resident stages of include/splash_hip.h (torch tensors or
                          raw device pointers; used by bench.py and multi-GPU drivers)

Everything computes through librsplash_hip.so; there is no numerical fallback here.
"""
import ctypes as C
import datetime as _dt

import numpy as np

from . import _lib
from ._lib import (DIAG_NAMES, DIAG_REC, EXTRA_NAMES, NDIAG, NEXTRA, NOUT, NPAR, NSTATE, OUT_NAMES,
                   SplashError, check, lib)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def device_count():
    return lib.splash_hip_device_count()


def cache_stats(device=0):
    """(held, idle) bytes of the library's device-buffer cache on `device`"""
    held, idle = C.c_size_t(0), C.c_size_t(0)
    check(lib.splash_hip_cache_stats(int(device), C.byref(held), C.byref(idle)), "splash_hip_cache_stats")
    return held.value, idle.value


def orbit_day(doy, year):
    """(dr, sin delta, cos delta) of one calendar day -- host-only, no GPU needed"""
    v = [C.c_double() for _ in range(3)]
    check(lib.splash_orbit_day(int(doy), int(year), *[C.byref(x) for x in v]), "splash_orbit_day")
    return tuple(x.value for x in v)


def time_fields(time_index):
    """doy / year / month vectors (what R derives with format(time_index, '%j' / '%Y' / '%m'),
    R/splash.point.R:57-59, 547)."""
    t = np.asarray(time_index)
    if t.dtype.kind == "M":
        days = t.astype("datetime64[D]")
        years = days.astype("datetime64[Y]")
        months = days.astype("datetime64[M]")
        year = years.astype(int) + 1970
        month = months.astype(int) % 12 + 1
        doy = (days - years.astype("datetime64[D]")).astype(int) + 1
    else:
        dates = [d if isinstance(d, _dt.date) else _dt.date.fromisoformat(str(d)) for d in t]
        year = np.array([d.year for d in dates])
        month = np.array([d.month for d in dates])
        doy = np.array([d.timetuple().tm_yday for d in dates])
    return doy.astype(np.int32), year.astype(np.int32), month.astype(np.int32)


def month_ids(year, month):
    """running month number (groups of format(time_index,'%Y-%m'), R/splash.point.R:208)."""
    key = np.asarray(year).astype(np.int64) * 12 + np.asarray(month)
    chg = np.r_[0, (key[1:] != key[:-1]).astype(np.int64)]
    return np.cumsum(chg).astype(np.int32)


# ---------------------------------------------------------------------------- splash.point
def _refuse_monthly_inputs(time_index):
    """R/splash.point.R:56, 66-86: a series whose first two dates are more than 20 days apart is taken to be MONTHLY
    and disaggregated -- tc and sw_in by linear interpolation, the rain by month2day_rain() (:460-516), which draws the
    wet days and their amounts from R's unseeded RNG (out of scope).  Such a series is refused here
    rather than run as if it were daily."""
    t = np.asarray(time_index, dtype="datetime64[D]")
    if t.size >= 2 and abs(int((t[1] - t[0]).astype(np.int64))) > 20:
        raise ValueError("monthly inputs: disaggregate to daily first -- the reference draws the daily rain of a monthly "
                         "series from an unseeded RNG (R/splash.point.R:66-86, month2day_rain), which has no reproducible twin")


def splash_point(sw_in, tc, pn, lat, elev, slop=0.0, asp=0.0, soil_data=None, Au=0.0,
                 resolution=250.0, time_index=None, monthly_out=False, backend="hip"):
    """R/splash.point.R:29.  Daily series in, dict of the 9 reference outputs out
    (wn, ro, pet, aet, snow, cond, bflow, netr, sm_lim).  `time_index`: dates of the series.
    Monthly inputs (R/splash.point.R:66-86) are not supported: their disaggregation draws from
    an unseeded RNG in the reference.
    backend: "hip" (default: the GPU; no device = an error) or "host" -- the same day step, the library's own device
    functions compiled for the host, on a CPU core (include/splash_hip.h, splash_opts.backend): a single site is a
    chain of sequential day steps that one core steps through faster than a lone wavefront.  Never chosen for you."""
    if time_index is None:
        raise ValueError("time_index is required (dates of the daily series)")
    if soil_data is None or len(soil_data) != 6:
        raise ValueError("soil_data = c(sand, clay, OM, gravel, bulk density (may be NaN), depth)")
    doy, year, month = time_fields(time_index)
    nt = len(doy)
    _refuse_monthly_inputs(time_index)
    f = [_d(x) for x in (sw_in, tc, pn)]
    if any(len(x[0]) != nt for x in f):
        raise ValueError("sw_in, tc, pn and time_index must have the same length")
    sd, sdp = _d(soil_data)
    au = np.atleast_1d(np.asarray(Au, dtype=np.float64))
    aua, aup = _d(au)
    nm = int(month_ids(year, month)[-1]) + 1
    nt_out = nm if monthly_out else nt
    out = np.empty((NOUT, nt_out))
    ptrs = (_dp * NOUT)(*[out[k].ctypes.data_as(_dp) for k in range(NOUT)])
    err = C.create_string_buffer(512)
    ia = [_i(x) for x in (doy, year, month)]
    rc = lib.splash_point_run_backend(_lib.backend_code(backend), nt, ia[0][1], ia[1][1], ia[2][1], f[0][1], f[1][1],
                                      f[2][1], float(lat), float(elev), float(slop), float(asp), sdp, aup, len(au),
                                      float(resolution), int(bool(monthly_out)), ptrs, err, 512)
    if rc:
        raise SplashError(f"splash_point_run failed ({rc}): {err.value.decode()}")
    return {k: out[i] for i, k in enumerate(OUT_NAMES)}


# ---------------------------------------------------------------------------- splash.grid
def splash_grid(sw_in, tc, pn, elev, soil, lat, slop, asp, Au=None, resolution=250.0,
                time_index=None, sim_control=None, outputs=OUT_NAMES, device=0,
                max_cells_per_launch=0, want_aux=False, want_diag=False, outdir=None, lon=None,
                basin=None, n_basin=0, devices=None, out=None, want_state=True, submit=False,
                mem_fraction=0.0, out_dtype=np.float64, tt=None, state0=None, ai0=None, backend="hip"):
    """R/splash.grid.R:28 on arrays.  sw_in/tc/pn: [nt, ncell] (or [nt, H, W]) daily rasters;
    elev/lat/slop/asp: [ncell]; soil: [6, ncell]; Au: [3, ncell] (area, ncellin, ncellout) or
    None.  The static rasters the reference derives with raster/GDAL/TauDEM (lat, slope, aspect,
    upslope area, resolution: R/splash.grid.R:76-110) are inputs here.  sim_control =
    dict(monthly_out=True) mirrors sim.control (monthly is the reference's default).
    Returns dict of [nt_out, ...] arrays named as R/splash.grid.R:449 (netr is 'netrad' there).
    With `outdir` (and [nt, H, W] rasters) the bricks are also written as NetCDF files named, and
    with variables named, as the reference names them (sim.control$inmem = FALSE,
    R/splash.grid.R:186-216; rsplash_amd/io.py); `lon` [W] gives the x axis (default: column index),
    the y axis is the latitude raster's first column.  Paths come back under "files".
    Forcing arrays of dtype float32 are handed over as they are (single-precision transport, widened
    on the device; the reference's rasters are FLT4S) -- the arithmetic is FP64 either way.
    basin/n_basin: optional integer basin id per cell -> want_diag totals per basin
    ([n_basin, nmonth] per quantity).  devices: list of HIP ordinals to split the cells over inside
    the one call (one host thread per entry; the same ordinal may repeat).
    out: dict name -> preallocated array [nt_out, ...] of `out_dtype` to receive that series (what an R
    caller's allocMatrix is); series not in `out` are allocated here.
    out_dtype=np.float32 (or a dict name -> dtype): single-precision transport of the results (splash_grid_out.out_f32): every
    value is rounded to the nearest float on the device -- what the reference's FLT4S bricks hold,
    R/splash.grid.R:186-216 -- which halves the device->host bytes; the arithmetic is FP64 either way.
    tt, state0, ai0 (ABI 5): a long series in time chunks.  tt [ncell]: the snowfall threshold of the WHOLE
    series (the maximum of snow_threshold() over the chunks), used instead of this call's own; state0
    [5, ncell] = "state_end" of the chunk before and ai0 [ncell] = aux["AI"] of the first chunk: the call
    continues that run -- no spin-up, any number of days -- and equals the undivided call bit for bit.
    submit=True: splash_grid_submit -- the call runs on its own host thread and streams; a GridJob
    comes back at once and its wait() returns what this function would have (a host loop over row
    blocks keeps two or more in flight, each with its mem_fraction of the device's free memory)."""
    if time_index is None:
        raise ValueError("time_index is required")
    monthly = True if sim_control is None else bool(sim_control.get("monthly_out", True))
    f32 = all(getattr(x, "dtype", None) == np.float32 for x in (sw_in, tc, pn))
    fdt = np.float32 if f32 else np.float64
    sw = np.asarray(sw_in, dtype=fdt)
    shp = sw.shape[1:]
    nt = sw.shape[0]
    ncell = int(np.prod(shp))
    doy, year, month = time_fields(time_index)
    _refuse_monthly_inputs(time_index)
    if len(doy) != nt:
        raise ValueError("time_index length differs from the rasters' time dimension")

    def flat(a, lead):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(lead + (ncell,))
        return a, a.ctypes.data_as(_dp)

    def flatf(a):
        a = np.ascontiguousarray(np.asarray(a, dtype=fdt)).reshape((nt, ncell))
        return a, a.ctypes.data_as(C.POINTER(C.c_float) if f32 else _dp)

    f = [flatf(x) for x in (sw, tc, pn)]
    st = [flat(np.broadcast_to(np.asarray(x, dtype=np.float64), shp) if np.ndim(x) == 0 else x, ())
          for x in (lat, elev, slop, asp, resolution)]
    so = flat(soil, (6,))
    au = flat(Au, (3,)) if Au is not None else None
    ia = [_i(x) for x in (doy, year, month)]
    gin = _lib.GridIn(ncell, nt, ia[0][1], ia[1][1], ia[2][1],
                      None if f32 else f[0][1], None if f32 else f[1][1], None if f32 else f[2][1],
                      st[0][1], st[1][1], st[2][1], st[3][1], st[4][1], so[1],
                      au[1] if au else None)
    if f32:
        gin.sw_in_f32, gin.tc_f32, gin.pn_f32 = f[0][1], f[1][1], f[2][1]
    cont = None
    if tt is not None or state0 is not None or ai0 is not None:
        cont = [None if x is None else flat(x, lead) for x, lead in ((tt, ()), (state0, (NSTATE,)), (ai0, ()))]
        gin.tt, gin.state0, gin.ai0 = (None if c is None else c[1] for c in cont)
    nbas = 1
    bas = None
    if basin is not None:
        bas = np.ascontiguousarray(np.asarray(basin, dtype=np.int32)).reshape(ncell)
        nbas = int(n_basin) if n_basin else int(bas.max()) + 1
        gin.basin, gin.n_basin = bas.ctypes.data_as(_ip), nbas
    nm = int(month_ids(year, month)[-1]) + 1
    nt_out = nm if monthly else nt
    res = {}
    # one precision for all series, or a dict name -> dtype (series not named: float64)
    odts = {k: np.dtype(out_dtype.get(k, np.float64) if isinstance(out_dtype, dict) else out_dtype) for k in outputs}
    if any(d not in (np.dtype(np.float64), np.dtype(np.float32)) for d in odts.values()):
        raise ValueError("out_dtype must be float64 or float32")
    for k in outputs:
        a = out.get(k) if out else None
        if a is None:
            a = np.empty((nt_out,) + shp, odts[k])
        elif a.shape != (nt_out,) + shp or a.dtype != odts[k] or not a.flags.c_contiguous:
            raise ValueError(f"out[{k!r}] must be a C-contiguous {odts[k]} array of shape {(nt_out,) + shp}")
        res[k] = a
    gout = _lib.GridOut()
    for i, k in enumerate(OUT_NAMES):
        if k not in res:
            continue
        if odts[k] == np.float64:
            gout.out[i] = res[k].ctypes.data_as(_dp)
        else:
            gout.out_f32[i] = res[k].ctypes.data_as(C.POINTER(C.c_float))
    state_end = None
    if want_state:
        state_end = np.empty((NSTATE,) + shp)
        gout.state_end = state_end.ctypes.data_as(_dp)
    aux = diag = acc = None
    if want_aux:
        aux = np.empty((4,) + shp)
        gout.aux = aux.ctypes.data_as(_dp)
    if want_diag:
        diag = np.empty((nbas, nm, NDIAG))
        gout.diag = diag.ctypes.data_as(_dp)
        acc = np.zeros((nbas, nm, DIAG_REC), np.int64)
        gout.diag_acc = acc.ctypes.data_as(C.POINTER(C.c_int64))
    opt = _lib.Opts(int(monthly), int(device), int(max_cells_per_launch), 0, None, float(mem_fraction),
                    _lib.backend_code(backend))
    dv = None
    if devices is not None:
        dv = np.ascontiguousarray(np.asarray(devices, dtype=np.int32))
        opt.n_devices, opt.devices = len(dv), dv.ctypes.data_as(_ip)
    err = C.create_string_buffer(512)
    if submit:
        job = C.c_void_p()
        rc = lib.splash_grid_submit(C.byref(gin), C.byref(opt), C.byref(gout), C.byref(job), err, 512)
        if rc:
            raise SplashError(f"splash_grid_submit failed ({rc}): {err.value.decode()}")
        keep = (f, st, so, au, ia, gin, opt, gout, dv, res, state_end, aux, diag, bas, acc, cont)
        return GridJob(job, keep, lambda: _grid_results(res, shp, lat, lon, outdir, time_index, monthly, want_state,
                                                         state_end, want_aux, aux, want_diag, diag, acc, basin))
    rc = lib.splash_grid_run(C.byref(gin), C.byref(opt), C.byref(gout), err, 512)
    if rc:
        raise SplashError(f"splash_grid_run failed ({rc}): {err.value.decode()}")
    return _grid_results(res, shp, lat, lon, outdir, time_index, monthly, want_state, state_end, want_aux, aux,
                         want_diag, diag, acc, basin)


def snow_threshold(tc, lat, elev, backend="hip"):
    """K2 on host arrays (splash_snow_threshold_run): Tt [ncell] = max(tc[p_snow >= 0.5]) of the series
    tc [nt, ...] (R/splash.point.R:120-122); -inf where no day qualifies, NaN where tc has a gap.  The
    threshold of a series processed in time chunks is np.maximum over the chunks' thresholds.
    backend="host" (ABI 6): on the calling thread."""
    f32 = getattr(tc, "dtype", None) == np.float32
    a = np.ascontiguousarray(np.asarray(tc, dtype=np.float32 if f32 else np.float64))
    nt, shp = a.shape[0], a.shape[1:]
    n = int(np.prod(shp))
    la, ll = _d(np.broadcast_to(np.asarray(lat, dtype=np.float64), shp).reshape(n))
    ea, el = _d(np.broadcast_to(np.asarray(elev, dtype=np.float64), shp).reshape(n))
    out = np.empty(n)
    err = C.create_string_buffer(256)
    rc = lib.splash_snow_threshold_run_backend(_lib.backend_code(backend), n, nt, None if f32 else a.ctypes.data_as(_dp),
                                               a.ctypes.data_as(C.POINTER(C.c_float)) if f32 else None, ll, el,
                                               out.ctypes.data_as(_dp), err, 256)
    if rc:
        raise SplashError(f"splash_snow_threshold_run failed ({rc}): {err.value.decode()}")
    return out.reshape(shp)


class GridJob:
    """a splash_grid call in flight (splash_grid_submit); wait() -> the results dict.

    The library's thread reads the forcing arrays and writes the result arrays until splash_grid_wait
    returns (include/splash_hip.h), so the job keeps them alive (`_keep`) and NEVER lets go of them
    before it has waited: close() (also the context manager's exit and the finaliser) waits for a job
    that was not waited for and discards its outcome -- e.g. the second of two jobs in flight when the
    first one's wait() raised.  r/splash_shim.c's grid_handle_finalizer does the same for R."""

    def __init__(self, job, keep, finish, wait_fn=None):
        self._job, self._keep, self._finish = job, keep, finish
        self._wait_fn = wait_fn or lib.splash_grid_wait   # (injectable: the CPU test has no device)

    def wait(self):
        if self._job is None:
            raise SplashError("GridJob.wait() called twice (or after close())")
        err = C.create_string_buffer(512)
        job, self._job = self._job, None
        try:
            rc = self._wait_fn(job, err, 512)
        finally:
            finish, self._finish = self._finish, None
        if rc:
            self._keep = None   # the library is done with the arrays whatever the outcome
            raise SplashError(f"splash_grid_run failed ({rc}): {err.value.decode()}")
        r = finish()
        self._keep = None
        return r

    def close(self):
        """wait for a job nobody waited for (its result is dropped), then release the arrays"""
        job, self._job = self._job, None
        if job is not None:
            try:
                self._wait_fn(job, None, 0)
            finally:
                self._finish = None
        self._keep = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: nothing left to report to
            pass


def _grid_results(res, shp, lat, lon, outdir, time_index, monthly, want_state, state_end, want_aux, aux,
                  want_diag, diag, acc, basin):
    if outdir is not None:
        if len(shp) != 2:
            raise ValueError("outdir needs [nt, H, W] rasters")
        from . import io as _io
        latr = np.asarray(lat, dtype=np.float64).reshape(shp)[:, 0]
        lonv = np.arange(shp[1], dtype=np.float64) if lon is None else np.asarray(lon, dtype=np.float64)
        res["files"] = _io.write_bricks(outdir, res, lonv, latr, time_index, monthly)
    if want_state:
        res["state_end"] = state_end
    if want_aux:
        res["aux"] = dict(Tt=aux[0], AI=aux[1], passes1=aux[2], passes2=aux[3])
    if want_diag:
        # [nmonth] per quantity for the whole grid, [n_basin, nmonth] with basin ids
        res["diag"] = {k: (diag[:, :, i] if basin is not None else diag[0, :, i])
                       for i, k in enumerate(DIAG_NAMES)}
        res["diag_acc"] = acc
        res["diag_cells"] = acc[:, :, 2 * NDIAG] if basin is not None else acc[0, :, 2 * NDIAG]
    return res


# ---------------------------------------------------------------------------- unSWC
def day_step(lat, elv, n, y, sw_in, tc, pn, snowfall, slop, asp, soil_info, state, backend="hip"):
    """test hook splash_day_step: ONE day step of one cell = SPLASH::run_one_day (src/SPLASH.cpp:920-1588).
    state = (wn, snow, qin, td, nd) of the day before; returns dict(wn, ro, pet, aet, snow, cond, bflow, netr, qin,
    tdrain, snwage)."""
    si, sip = _d(soil_info)
    st, stp = _d(state)
    out = np.empty(11)
    prev = lib.splash_mod_backend(_lib.backend_code(backend))
    try:
        check(lib.splash_day_step(float(lat), float(elv), int(n), int(y), float(sw_in), float(tc), float(pn),
                                  float(snowfall), float(slop), float(asp), sip, stp, out.ctypes.data_as(_dp)),
              "splash_day_step")
    finally:
        lib.splash_mod_backend(prev)
    return dict(zip(("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin", "tdrain", "snwage"), out))


def soil_hydro(lat, elev, slop, asp, resolution, soil, au=None, backend="hip"):
    """test hook splash_soil_hydro: K1 alone on host arrays -> cellpar [NPAR, ncell] (rows: include/splash_hip.h)"""
    a = [_d(np.atleast_1d(x)) for x in (lat, elev, slop, asp, resolution)]
    n = a[0][0].size
    so, sop = _d(soil)
    if so.shape != (6, n):
        raise ValueError("soil must be [6, ncell]")
    aup = None
    if au is not None:
        aua, aup = _d(au)
        if aua.shape != (3, n):
            raise ValueError("au must be [3, ncell]")
    par = np.empty((NPAR, n))
    err = C.create_string_buffer(256)
    prev = lib.splash_mod_backend(_lib.backend_code(backend))
    try:
        rc = lib.splash_soil_hydro(n, *[x[1] for x in a], sop, aup, par.ctypes.data_as(_dp), err, 256)
    finally:
        lib.splash_mod_backend(prev)
    if rc:
        raise SplashError(f"splash_soil_hydro failed ({rc}): {err.value.decode()}")
    return par


def unSWC(soil_data, uns_depth, wn, backend="hip"):
    """R/unsSWC.R:14.  soil_data = c(sand, clay, OM, gravel, bulk density, depth); wn: series of
    column water content [mm].  Returns dict(wn_z, wtd, theta_i) like the reference's list
    (its first element is the water to uns_depth, :85-86).  backend="host" (ABI 6): on the calling thread, no GPU."""
    soil = np.asarray(soil_data, dtype=np.float64).reshape(6, 1)
    w = np.ascontiguousarray(np.asarray(wn, dtype=np.float64).reshape(-1, 1))
    r = unSWC_cells(soil, uns_depth, w, backend=backend)
    return {k: v[:, 0] for k, v in r.items()}


def unSWC_cells(soil, uns_depth, wn, backend="hip"):
    """unSWC (R/unsSWC.R:14: the SITE function, with its stoniness correction and its patch of a non-positive
    integral) for many sites at once: soil [6, ncell], wn [nt, ncell] -> dict of [nt, ncell].
    (The gridded function of the reference has other semantics: unSWC_grid.)"""
    w, wp = _d(wn)
    nt, ncell = w.shape
    so, sp = _d(soil)
    if so.shape != (6, ncell):
        raise ValueError("soil must be [6, ncell]")
    out = {k: np.empty((nt, ncell)) for k in ("wn_z", "wtd", "theta_i")}
    err = C.create_string_buffer(512)
    rc = lib.splash_unswc_run_backend(_lib.backend_code(backend), ncell, nt, wp, sp, float(uns_depth),
                                      *[out[k].ctypes.data_as(_dp) for k in ("wn_z", "wtd", "theta_i")],
                                      err, 512)
    if rc:
        raise SplashError(f"splash_unswc_run failed ({rc}): {err.value.decode()}")
    return out


def unSWC_grid(soil, uns_depth, wn, backend="hip", want=("w_z", "wtd", "Se", "theta_mean")):
    """R/unsSWC.grid.R:14-141 on arrays: soil [6, ncell], wn [nt, ncell] -> dict of [nt, ncell] with the gridded
    function's four rasters: w_z (its "swc_top" file, mm), wtd (m), Se (effective saturation of the layer) and
    theta_mean (the column's clamped mean theta).  Its semantics are its own (include/splash_hip.h): no stoniness
    correction, no patch, the integral's water table capped at uns_depth.  rsplash_amd.grid.unSWC_grid is the
    raster-level twin that also writes the four NetCDF files."""
    w, wp = _d(wn)
    nt, ncell = w.shape
    so, sp = _d(soil)
    if so.shape != (6, ncell):
        raise ValueError("soil must be [6, ncell]")
    names = ("w_z", "wtd", "Se", "theta_mean")
    out = {k: np.empty((nt, ncell)) for k in names if k in want}
    err = C.create_string_buffer(512)
    rc = lib.splash_unswc_grid_run(_lib.backend_code(backend), ncell, nt, wp, sp, float(uns_depth),
                                   *[out[k].ctypes.data_as(_dp) if k in out else None for k in names], err, 512)
    if rc:
        raise SplashError(f"splash_unswc_grid_run failed ({rc}): {err.value.decode()}")
    return out


def dev_unswc(ncell, nt, wn, ldw, soil, uns_depth, wn_z, wtd, theta, ldo, stream=None):
    check(lib.splash_dev_unswc(ncell, nt, _ptr(wn), ldw, _ptr(soil), float(uns_depth), _ptr(wn_z),
                               _ptr(wtd), _ptr(theta), ldo, stream), "splash_dev_unswc")


def dev_unswc_grid(ncell, nt, wn, ldw, soil, uns_depth, w_z, wtd, se, theta_mean, ldo, stream=None):
    check(lib.splash_dev_unswc_grid(ncell, nt, _ptr(wn), ldw, _ptr(soil), float(uns_depth), _ptr(w_z), _ptr(wtd),
                                    _ptr(se), _ptr(theta_mean), ldo, stream), "splash_dev_unswc_grid")


# ---------------------------------------------------------------------------- Rcpp module mirror
class SPLASH:
    """Mirror of the Rcpp class exposed by src/SPLASH.cpp:12-21 (`new(SPLASH, lat, elev)`)."""

    def __init__(self, lat, elv, backend="hip"):
        """backend: "hip" (default) or "host" -- where this object's methods compute (splash_mod_backend: the
        library keeps the choice per calling thread; it is set before every call of this object)"""
        self.lat, self.elv = float(lat), float(elv)
        self.backend = _lib.backend_code(backend)

    def _enter(self):
        rc = lib.splash_mod_backend(self.backend)
        if rc < 0:
            check(rc, "splash_mod_backend")

    def spin_up(self, n, y, sw_in, tair, pn, slop, asp, snowfall, soil_info):
        """src/SPLASH.cpp:1594 -> dict(sm, snow, qin, tdrain, ro, snwage, pet) (+ 'passes')."""
        self._enter()
        a = [_d(x) for x in (sw_in, tair, pn, snowfall, soil_info)]
        names = ("sm", "snow", "qin", "tdrain", "ro", "snwage", "pet")
        out = {k: np.empty(n) for k in names}
        passes = C.c_int(0)
        check(lib.splash_mod_spin_up(self.lat, self.elv, int(n), int(y), a[0][1], a[1][1], a[2][1],
                                     float(slop), float(asp), a[3][1], a[4][1],
                                     *[out[k].ctypes.data_as(_dp) for k in names],
                                     C.byref(passes)), "splash_mod_spin_up")
        out["passes"] = passes.value
        return out

    def run_all(self, doys, yrs, sw_in, tair, pn, wn_last, slop, asp, snow_last, snowfall,
                soil_info, qin_last, td_last, nds_last):
        """src/SPLASH.cpp:1833 -> dict of the 11 series of :1913-1915."""
        self._enter()
        nt = len(doys)
        d, dp_ = _i(doys)
        y, yp = _i(yrs)
        a = [_d(x) for x in (sw_in, tair, pn, snowfall, soil_info)]
        names = ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev", "tdrain",
                 "snwage")
        out = {k: np.empty(nt) for k in names}
        check(lib.splash_mod_run_all(self.lat, self.elv, nt, dp_, yp, a[0][1], a[1][1], a[2][1],
                                     float(wn_last), float(slop), float(asp), float(snow_last),
                                     a[3][1], a[4][1], float(qin_last), float(td_last),
                                     float(nds_last),
                                     *[out[k].ctypes.data_as(_dp) for k in names]),
              "splash_mod_run_all")
        return out

    def run_one_year(self, n, y, sw_in, tair, pn, wn_vec, slop, asp, snow, snowfall, soil_info,
                     qin_vec, td_vec, nds):
        """src/SPLASH.cpp:1751 -> dict of the 11 series of :1827-1829, each [n]"""
        self._enter()
        a = [_d(x) for x in (sw_in, tair, pn, wn_vec, snow, snowfall, soil_info, qin_vec, td_vec, nds)]
        names = ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev", "tdrain",
                 "snwage")
        out = {k: np.empty(n) for k in names}
        check(lib.splash_mod_run_one_year(self.lat, self.elv, int(n), int(y), len(a[3][0]), a[0][1],
                                          a[1][1], a[2][1], a[3][1], float(slop), float(asp), a[4][1],
                                          a[5][1], a[6][1], a[7][1], a[8][1], a[9][1],
                                          *[out[k].ctypes.data_as(_dp) for k in names]),
              "splash_mod_run_one_year")
        return out

    @staticmethod
    def moist_surf(depth, z, bub_p, wn, SAT, RES, lam, backend="hip"):
        """src/SPLASH.cpp:1921"""
        check(min(lib.splash_mod_backend(_lib.backend_code(backend)), 0), "splash_mod_backend")
        v = C.c_double()
        check(lib.splash_mod_moist_surf(depth, z, bub_p, wn, SAT, RES, lam, C.byref(v)),
              "splash_mod_moist_surf")
        return v.value

    @staticmethod
    def inf_GA(bub_press, theta_i, Ksat, theta_s, lam, P, tdur, slop, backend="hip"):
        """src/SPLASH.cpp:1963"""
        check(min(lib.splash_mod_backend(_lib.backend_code(backend)), 0), "splash_mod_backend")
        v = C.c_double()
        check(lib.splash_mod_inf_GA(bub_press, theta_i, Ksat, theta_s, lam, P, tdur, slop,
                                    C.byref(v)), "splash_mod_inf_GA")
        return v.value


def evap_batch(lat, elv, sw, n, y, sw_in, tc, slop, asp, snow, nd):
    """test hook: SOLAR+EVAP rows on the GPU -> [m, 11] (order of src/etr.h:44-56)."""
    m = len(lat)
    out = np.empty((m, 11))
    f = [_d(x) for x in (lat, elv, sw)]
    i = [_i(x) for x in (n, y)]
    g = [_d(x) for x in (sw_in, tc, slop, asp, snow, nd)]
    check(lib.splash_dev_evap_batch(m, f[0][1], f[1][1], f[2][1], i[0][1], i[1][1],
                                    *[x[1] for x in g], out.ctypes.data_as(_dp), None),
          "splash_dev_evap_batch")
    return out


def math_probe(x, y):
    """test hook: device flog(x), fexp(x), fpow(x, y), facos(x), fsqrt(x), fatan(x), fatan2(x, y) -> [7, n]"""
    xa, xp = _d(x)
    ya, yp = _d(y)
    out = np.empty((7, len(xa)))
    check(lib.splash_dev_math_probe(len(xa), xp, yp, out.ctypes.data_as(_dp)),
          "splash_dev_math_probe")
    return out


# ---------------------------------------------------------------------------- device-resident API
def _ptr(x):
    """device pointer of a torch tensor (or an int / None)."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()


class Calendar:
    """Per-day tables on the current device (splash_calendar_create)."""

    def __init__(self, doy, year, month):
        ia = [_i(x) for x in (doy, year, month)]
        self.nt = len(ia[0][0])
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib.splash_calendar_create(self.nt, ia[0][1], ia[1][1], ia[2][1], C.byref(h), err, 512)
        if rc:
            raise SplashError(f"splash_calendar_create failed ({rc}): {err.value.decode()}")
        self.h = h
        self.nmonth = lib.splash_calendar_nmonth(h)
        self.month_id = month_ids(year, month)

    @classmethod
    def from_dates(cls, time_index):
        return cls(*time_fields(time_index))

    def orbit(self):
        dr, sd, cd = (np.empty(self.nt) for _ in range(3))
        check(lib.splash_calendar_orbit(self.h, *[x.ctypes.data_as(_dp) for x in (dr, sd, cd)]))
        return dr, sd, cd

    def close(self):
        if self.h:
            lib.splash_calendar_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dev_soil_hydro(ncell, lat, elev, slop, asp, resolution, soil, au, cellpar, stream=None):
    check(lib.splash_dev_soil_hydro(ncell, _ptr(lat), _ptr(elev), _ptr(slop), _ptr(asp),
                                    _ptr(resolution), _ptr(soil), _ptr(au), _ptr(cellpar), stream),
          "splash_dev_soil_hydro")


def dev_snow_threshold(ncell, nt, tc, ldf, cellpar, ldp, Tt, stream=None):
    check(lib.splash_dev_snow_threshold(ncell, nt, _ptr(tc), ldf, _ptr(cellpar), ldp, _ptr(Tt),
                                        stream), "splash_dev_snow_threshold")


def spin_up_work_size(ncell):
    return lib.splash_dev_spin_up_work_size(ncell)


def dev_spin_up(cal, ncell, sw_in, tc, pn, ldf, Tt, cellpar, ldp, state, lds, work, passes=None,
                stream=None):
    """work: device scratch of spin_up_work_size(ncell) bytes"""
    check(lib.splash_dev_spin_up(cal.h, ncell, _ptr(sw_in), _ptr(tc), _ptr(pn), ldf, _ptr(Tt),
                                 _ptr(cellpar), ldp, _ptr(state), lds, _ptr(passes), _ptr(work),
                                 stream), "splash_dev_spin_up")


def dev_spin_up_begin(cal, ncell, sw_in, tc, pn, ldf, Tt, cellpar, ldp, state, lds, work, passes=None,
                      stream=None):
    """all phases of the spin-up but the long tail (see include/splash_hip.h)"""
    check(lib.splash_dev_spin_up_begin(cal.h, ncell, _ptr(sw_in), _ptr(tc), _ptr(pn), ldf, _ptr(Tt),
                                       _ptr(cellpar), ldp, _ptr(state), lds, _ptr(passes), _ptr(work),
                                       stream), "splash_dev_spin_up_begin")


def dev_spin_up_finish(cal, ncell, sw_in, tc, pn, ldf, Tt, cellpar, ldp, state, lds, work, passes=None,
                       stream=None):
    """the tail: the cells _begin left unfinished"""
    check(lib.splash_dev_spin_up_finish(cal.h, ncell, _ptr(sw_in), _ptr(tc), _ptr(pn), ldf, _ptr(Tt),
                                        _ptr(cellpar), ldp, _ptr(state), lds, _ptr(passes), _ptr(work),
                                        stream), "splash_dev_spin_up_finish")


def dev_spin_up_pending(work, ncell):
    """device pointers (ints) into the workspace: byte mask [ncell], uint32 cell list, uint32 count"""
    m, c, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
    check(lib.splash_dev_spin_up_pending(_ptr(work), ncell, C.byref(m), C.byref(c), C.byref(n)),
          "splash_dev_spin_up_pending")
    return m.value, c.value, n.value


def dev_spin_up_pending_count(work, ncell, stream=None):
    """number of cells _begin left for _finish (waits for `stream`)"""
    v = C.c_uint32(0)
    check(lib.splash_dev_spin_up_pending_count(_ptr(work), ncell, stream, C.byref(v)),
          "splash_dev_spin_up_pending_count")
    return v.value


def diag_acc_size(n_basin, nmonth):
    """bytes of the exact water-balance accumulators: int64 [n_basin][nmonth][NDIAG+1][2]"""
    return lib.splash_dev_diag_acc_size(int(n_basin), int(nmonth))


def dev_diag_finalize(diag_acc, n_basin, nmonth, diag, counts=None, stream=None):
    """limbs -> doubles on the device: diag [n_basin, nmonth, NDIAG]"""
    check(lib.splash_dev_diag_finalize(_ptr(diag_acc), int(n_basin), int(nmonth), _ptr(diag),
                                       _ptr(counts), stream), "splash_dev_diag_finalize")


def diag_finalize_host(acc):
    """limbs [n_basin, nmonth, NDIAG+1, 2] (numpy int64) -> (totals [n_basin, nmonth, NDIAG],
    counts [n_basin, nmonth, 2])"""
    a = np.ascontiguousarray(acc, dtype=np.int64)
    nb, nm = a.shape[0], a.shape[1]
    out, cnt = np.empty((nb, nm, NDIAG)), np.empty((nb, nm, 2))
    lib.splash_diag_finalize_host(a.ctypes.data_as(C.POINTER(C.c_int64)), nb, nm,
                                  out.ctypes.data_as(_dp), cnt.ctypes.data_as(_dp))
    return out, cnt


def infiltration_probe(cellpar, wn, ksat_visc, P):
    """test hook: the day loop's own moist_surf()/inf_GA() -> (theta_surf, infiltration)"""
    cp, cpp = _d(cellpar)
    n = cp.shape[1]
    a = [_d(x) for x in (wn, ksat_visc, P)]
    th, inf = np.empty(n), np.empty(n)
    check(lib.splash_dev_infiltration_probe(n, cpp, a[0][1], a[1][1], a[2][1],
                                            th.ctypes.data_as(_dp), inf.ctypes.data_as(_dp)),
          "splash_dev_infiltration_probe")
    return th, inf


def dev_run_all(cal, t0, nt, ncell, sw_in, tc, pn, ldf, Tt, cellpar, ldp, state, lds, out_ptrs,
                ldo, monthly=False, diag_acc=None, basin=None, n_basin=0, stream=None, skip=None,
                cells=None):
    """out_ptrs: dict name -> tensor/pointer for the series to store (others are skipped); the
    names of OUT_NAMES plus, daily mode only, qin_prev / tdrain / snwage.
    diag_acc: int64 tensor [n_basin, cal.nmonth, NDIAG+1, 2] the water-balance totals are ADDED to."""
    o = _lib.DevOut()
    for i, k in enumerate(OUT_NAMES):
        o.out[i] = _ptr(out_ptrs.get(k)) if out_ptrs else None
    for i, k in enumerate(EXTRA_NAMES):
        o.extra[i] = _ptr(out_ptrs.get(k)) if out_ptrs else None
    o.ldo = ldo
    o.monthly = int(bool(monthly))
    o.diag_acc = _ptr(diag_acc)
    o.basin = _ptr(basin)
    o.n_basin = int(n_basin)
    o.skip = _ptr(skip)
    o.cells = _ptr(cells)
    check(lib.splash_dev_run_all(cal.h, t0, nt, ncell, _ptr(sw_in), _ptr(tc), _ptr(pn), ldf,
                                 _ptr(Tt), _ptr(cellpar), ldp, _ptr(state), lds, C.byref(o),
                                 stream), "splash_dev_run_all")


def dev_synth_statics(seed, H, W, cell0, ncell, terrain, lat, elev, slop, asp, resolution, soil,
                      au, stream=None):
    check(lib.splash_dev_synth_statics(seed, H, W, cell0, ncell, int(terrain), _ptr(lat),
                                       _ptr(elev), _ptr(slop), _ptr(asp), _ptr(resolution),
                                       _ptr(soil), _ptr(au), stream), "splash_dev_synth_statics")


def dev_synth_forcing(seed, cal, t0, nt, cell0, ncell, lat, elev, sw_in, tc, pn, ldf, stream=None):
    check(lib.splash_dev_synth_forcing(seed, cal.h, t0, nt, cell0, ncell, _ptr(lat), _ptr(elev),
                                       _ptr(sw_in), _ptr(tc), _ptr(pn), ldf, stream),
          "splash_dev_synth_forcing")


# ---------------------------------------------------------------------------- static rasters
FLOW_IN_TOP = (2, 1, 128, 4, 0, 64, 8, 16, 32)     # matrix(c(...), nrow=3), R/upslope_area.R:142
FLOW_OUT_TOP = tuple(reversed(FLOW_IN_TOP))        # :144


def dev_terrain(H, W, elev, ymax, yres, xres, slope=None, aspect=None, lat=None, resolution=None,
                flowdir=None, na_to_zero=True, stream=None):
    """R/splash.grid.R:96-110 on a device-resident lon/lat elevation raster [H*W] (see
    include/splash_hip.h: splash_dev_terrain)"""
    check(lib.splash_dev_terrain(int(H), int(W), _ptr(elev), float(ymax), float(yres), float(xres),
                                 int(bool(na_to_zero)), _ptr(slope), _ptr(aspect), _ptr(lat),
                                 _ptr(resolution), _ptr(flowdir), stream), "splash_dev_terrain")


def dev_ncellflow(H, W, flowdir, codes9, out, pad=False, stream=None):
    """R/upslope_area.R:140-165; codes9: device tensor of the 9 window codes in row order"""
    check(lib.splash_dev_ncellflow(int(H), int(W), _ptr(flowdir), _ptr(codes9), int(bool(pad)),
                                   _ptr(out), stream), "splash_dev_ncellflow")


def dev_sinkfill(H, W, elev, resolution_m, filled, degree=1.0, stream=None):
    """R/upslope_area.R:15-16 (sinkfill(as.matrix(dem), resolution, 1)) on a device raster; returns the
    number of relaxation sweeps (see include/splash_hip.h: splash_dev_sinkfill)"""
    n = C.c_int(0)
    check(lib.splash_dev_sinkfill(int(H), int(W), _ptr(elev), float(resolution_m), float(degree),
                                  _ptr(filled), C.byref(n), stream), "splash_dev_sinkfill")
    return n.value


def dev_flow_accumulation(H, W, flowdir, ups_ncell=None, ups_area=None, elev=None, resolution=None,
                          tau_codes=False, stream=None):
    """R/upslope_area.R:21 / :74, 85-96: cells draining through each cell and the contributing area
    in m2; returns the number of host-synchronised levels"""
    n = C.c_int(0)
    check(lib.splash_dev_flow_accumulation(int(H), int(W), _ptr(flowdir), _ptr(elev), int(bool(tau_codes)),
                                           _ptr(resolution), _ptr(ups_ncell), _ptr(ups_area),
                                           C.byref(n), stream), "splash_dev_flow_accumulation")
    return n.value


def dev_mfd_area(H, W, elev, resolution_m, area, stream=None):
    """R/upslope_area.R:21 (topidx(...)$area): multiple-flow-direction contributing area in m2 of a
    (sink-filled) device raster; returns the number of levels"""
    n = C.c_int(0)
    check(lib.splash_dev_mfd_area(int(H), int(W), _ptr(elev), float(resolution_m), _ptr(area), C.byref(n),
                                  stream), "splash_dev_mfd_area")
    return n.value


def upslope_area(dem, ymax, yres, xres, degree=1.0, method="top"):
    """upslope_area(dem, resolution) of R/upslope_area.R:12-58 for a host lon/lat elevation array
    [H, W]: sink fill with the mean resolution (:14-16), contributing area (method "top": :21, the
    multiple-flow-direction area of topidx; "tau": the D8 count times the cell area of
    upslope_areav2, :74, 85-96), flow direction of the FILLED surface and the cells draining in and
    out (:31-33).  Returns dict(areacatch, ncellin,
    ncellout) like the reference's stack, plus ups_ncell, flowdir and the filled surface."""
    import torch
    e = torch.as_tensor(np.ascontiguousarray(dem, dtype=np.float64), device="cuda")
    H, W = e.shape
    t = lambda: torch.empty(H * W, dtype=torch.float64, device="cuda")
    res, filled, fd, nin, nout, cnt, area = (t() for _ in range(7))
    dev_terrain(H, W, e, ymax, yres, xres, resolution=res)
    rv = res.cpu().numpy()
    rv = rv[~np.isnan(rv)].astype(np.longdouble)  # cellStats(resolution, 'mean', na.rm=TRUE): R's mean is
    m0 = rv.sum() / rv.size                        # a long-double sum and one refinement pass
    mean_res = float(m0 + (rv - m0).sum() / rv.size)
    dev_sinkfill(H, W, e, mean_res, filled, degree)
    dev_terrain(H, W, filled, ymax, yres, xres, flowdir=fd)
    dev_flow_accumulation(H, W, fd, cnt, area, elev=filled, resolution=res)  # :74, 85-96 (TauDEM twin)
    if method == "top":  # :21 topidx(as.matrix(elev), resolution = mean)$area
        dev_mfd_area(H, W, filled, mean_res, area)
    for out, codes in ((nin, FLOW_IN_TOP), (nout, FLOW_OUT_TOP)):
        dev_ncellflow(H, W, fd, torch.tensor(codes, dtype=torch.float64, device="cuda"), out)
    torch.cuda.synchronize()
    g = lambda x: x.cpu().numpy().reshape(H, W)
    return dict(areacatch=g(area), ncellin=g(nin), ncellout=g(nout), ups_ncell=g(cnt), flowdir=g(fd),
                filled=g(filled))


def static_rasters(elev, ymax, yres, xres, na_to_zero=True, degree=1.0, want=None):
    """host arrays through the host-buffer entry point (splash_static_rasters_run, what the R shim
    binds): elev [H, W] -> dict of [H, W] arrays named like _lib.STATIC_NAMES (want: subset)"""
    from ._lib import STATIC_NAMES, StaticOut
    e = np.ascontiguousarray(elev, dtype=np.float64)
    H, W = e.shape
    names = STATIC_NAMES if want is None else tuple(want)
    o = {k: np.empty((H, W)) for k in names}
    so = StaticOut(**{k: o[k].ctypes.data_as(_dp) for k in names})
    buf = C.create_string_buffer(256)
    rc = lib.splash_static_rasters_run(H, W, e.ctypes.data_as(_dp), float(ymax), float(yres), float(xres),
                                       int(bool(na_to_zero)), float(degree), C.byref(so), buf, 256)
    if rc:
        raise SplashError(rc, "splash_static_rasters_run: " + buf.value.decode())
    return o


def terrain_gdal(elev, xres, yres, scale=111120.0, round_int=True):
    """The reference's LARGE-GRID terrain (R/splash.grid.R:83-84: `gdaldem slope -s 111120 -compute_edges`, `gdaldem
    aspect -zero_for_flat -compute_edges` on the DEM written as INT2S): elev [H, W] -> (slope, aspect) in degrees.
    gdaldem's published algorithm restated (include/splash_hip.h); parity unpinned."""
    e = np.ascontiguousarray(elev, dtype=np.float64)
    H, W = e.shape
    sl, asp = np.empty((H, W)), np.empty((H, W))
    buf = C.create_string_buffer(256)
    rc = lib.splash_terrain_gdal_run(H, W, e.ctypes.data_as(_dp), float(xres), float(yres), float(scale),
                                     int(bool(round_int)), sl.ctypes.data_as(_dp), asp.ctypes.data_as(_dp), buf, 256)
    if rc:
        raise SplashError(f"splash_terrain_gdal_run failed ({rc}): {buf.value.decode()}")
    return sl, asp


def terrain(elev, ymax, yres, xres, na_to_zero=True):
    """host arrays: elev [H, W] -> dict(slope, aspect, lat, resolution, flowdir, ncellin, ncellout),
    the static rasters splash.grid prepares before its loop (the upslope area: upslope_area())"""
    import torch
    e = torch.as_tensor(np.ascontiguousarray(elev, dtype=np.float64), device="cuda")
    H, W = e.shape
    o = {k: torch.empty(H * W, dtype=torch.float64, device="cuda")
         for k in ("slope", "aspect", "lat", "resolution", "flowdir", "ncellin", "ncellout")}
    dev_terrain(H, W, e, ymax, yres, xres, o["slope"], o["aspect"], o["lat"], o["resolution"],
                o["flowdir"], na_to_zero)
    for name, codes in (("ncellin", FLOW_IN_TOP), ("ncellout", FLOW_OUT_TOP)):
        c9 = torch.tensor(codes, dtype=torch.float64, device="cuda")
        dev_ncellflow(H, W, o["flowdir"], c9, o[name])
    torch.cuda.synchronize()
    return {k: v.cpu().numpy().reshape(H, W) for k, v in o.items()}
