"""ctypes doorway to the parity oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (rsplash_amd) never does.

  Oracle  -> oracle/liboracle.so        our plain-C restatement (splash_oracle.c)
  Ref     -> oracle/_ref/libsplash_ref.so  the real reference SOLAR/EVAP classes (may be absent)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    """Compile liboracle.so (and _ref when the reference tree is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "splash_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


class Srad(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("ru", "rv", "rw", "rnl", "hn", "rn_d", "rnn_d", "ts")]


class Etr(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("cond", "eet", "pet", "aet", "snowmelt", "sublimation",
                                          "econ", "pw", "rn_d", "visc", "pet_max")]


class Smr(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("sm", "ro", "swe", "bflow", "sqout", "tdr", "nd", "pet")]


class Soil(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("SAT", "FC", "WP", "bd", "AWC", "Ksat", "A", "B",
                                          "theta_c", "RES", "bubbling_p")]


class Opts(C.Structure):
    _fields_ = [("monthly_out", C.c_int), ("n_threads", C.c_int)]


class CellAux(C.Structure):
    _fields_ = [("soil_info", C.c_double * 12), ("wmax_r", C.c_double), ("Tt", C.c_double),
                ("AI", C.c_double), ("passes1", C.c_int), ("passes2", C.c_int),
                ("state0", C.c_double * 5), ("state1", C.c_double * 5), ("fragile", C.c_int), ("critical", C.c_int)]


AUX_DTYPE = np.dtype([("soil_info", "f8", 12), ("wmax_r", "f8"), ("Tt", "f8"), ("AI", "f8"),
                      ("passes1", "i4"), ("passes2", "i4"), ("state0", "f8", 5),
                      ("state1", "f8", 5), ("fragile", "i4"), ("critical", "i4")], align=True)
assert AUX_DTYPE.itemsize == C.sizeof(CellAux)

OUT_NAMES = ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "sm_lim")

# Slope sunset angle below which a cell-day counts as "sun grazing the slope" (event bit 1 of
# orc_grid_transient2): the reference's own daily radiation carries relative rounding noise
# ~1e-16/hs^2 there.  The value comes from tools/conditioning_evidence.py (profiles/r02_conditioning.md).
HS_LIM_DEG = 2e-3


class Oracle:
    def __init__(self, lib_path=None):
        """lib_path: an alternative build of splash_oracle.c (tools/conditioning_evidence.py);
        default = the pinned parity oracle oracle/liboracle.so"""
        if lib_path is None:
            build()
            lib_path = os.path.join(_HERE, "liboracle.so")
        self.lib = L = C.CDLL(lib_path)
        L.orc_viscosity_h2o.restype = C.c_float
        L.orc_viscosity_h2o.argtypes = [C.c_float, C.c_float]
        for f in ("orc_density_h2o", "orc_moist_surf", "orc_inf_GA", "orc_get_wtd",
                  "orc_snowfall_prob", "orc_elv2pres", "orc_synth_u", "orc_snow_partition"):
            getattr(L, f).restype = C.c_double
        L.orc_density_h2o.argtypes = [C.c_double] * 2
        L.orc_elv2pres.argtypes = [C.c_double]
        L.orc_moist_surf.argtypes = [C.c_double] * 7
        L.orc_inf_GA.argtypes = [C.c_double] * 8
        L.orc_get_wtd.argtypes = [C.c_double] * 6
        L.orc_snowfall_prob.argtypes = [C.c_double] * 3
        L.orc_synth_u.argtypes = [C.c_uint64] * 4
        L.orc_solar.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int] + [C.c_double] * 7 + \
            [C.POINTER(Srad)]
        L.orc_evap.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int] + \
            [C.c_double] * 6 + [C.POINTER(Etr)]
        L.orc_evap_batch.argtypes = [C.c_int, _dp, _dp, _dp, _ip, _ip] + [_dp] * 7
        L.orc_solar_batch.argtypes = [C.c_int, _dp, _dp, _ip, _ip] + [_dp] * 8
        L.orc_orbit.argtypes = [C.c_int, C.c_int] + [_dp] * 5
        L.orc_run_one_day.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int] + \
            [C.c_double] * 8 + [_dp] + [C.c_double] * 3 + [C.POINTER(Smr), C.POINTER(Etr)]
        L.orc_spin_up.restype = C.c_int
        L.orc_spin_up.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, _dp, _dp, _dp,
                                  C.c_double, C.c_double, _dp, _dp] + [_dp] * 7
        L.orc_run_all.argtypes = [C.c_double, C.c_double, C.c_int, _ip, _ip, _dp, _dp, _dp,
                                  C.c_double, C.c_double, C.c_double, C.c_double, _dp, _dp,
                                  C.c_double, C.c_double, C.c_double] + [_dp] * 11
        L.orc_soil_hydro.argtypes = [C.c_double] * 5 + [C.POINTER(Soil)]
        L.orc_soil_info.argtypes = [_dp, _dp, C.c_double, _dp, _dp]
        L.orc_snow_partition.argtypes = [C.c_int, _dp, _dp, _ip, C.c_double, C.c_double, _dp, _dp]
        L.orc_point.argtypes = [C.c_int, _ip, _ip, _ip, _ip, C.c_int, _dp, _dp, _dp, C.c_size_t] + \
            [C.c_double] * 4 + [_dp, _dp, C.c_double, C.POINTER(Opts), C.POINTER(_dp), C.c_size_t,
                                C.POINTER(CellAux)]
        L.orc_grid.argtypes = [C.c_int, C.c_int, _ip, _ip, _ip, _ip, C.c_int] + [_dp] * 10 + \
            [C.POINTER(Opts), C.POINTER(_dp), C.c_void_p]
        L.orc_grid_transient.argtypes = [C.c_int, C.c_int, _ip, _ip, _ip] + [_dp] * 10 + \
            [C.POINTER(_dp), C.c_int, _ip]
        L.orc_grid_transient2.argtypes = [C.c_int, C.c_int, _ip, _ip, _ip] + [_dp] * 10 + \
            [_dp, C.POINTER(_dp), C.c_void_p, C.c_double, C.c_void_p, C.c_int]
        L.orc_terrain.argtypes = [C.c_int, C.c_int, _dp, C.c_double, C.c_double, C.c_double, C.c_int] + [_dp] * 5
        L.orc_ncellflow.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int, _dp]
        L.orc_sinkfill.argtypes = [C.c_int, C.c_int, _dp, C.c_double, C.c_double, _dp]
        L.orc_mfd_area.argtypes = [C.c_int, C.c_int, _dp, C.c_double, _dp]
        L.orc_flow_accumulation.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp]
        L.orc_synth_statics.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int] + \
            [_dp] * 7
        L.orc_synth_forcing.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                                        C.c_int, _ip, _dp, _dp, _dp, _dp, _dp]
        L.orc_unswc.argtypes = [_dp, C.c_double, C.c_int, _dp, C.c_size_t, _dp, _dp, _dp, _dp,
                                C.c_size_t]
        L.orc_terrain_gdal.argtypes = [C.c_int, C.c_int, _dp, C.c_double, C.c_double, C.c_double, C.c_int, _dp, _dp]
        L.orc_unswc_grid.argtypes = [_dp, C.c_double, C.c_int, _dp, C.c_size_t, _dp, _dp, _dp, _dp, _dp,
                                     C.c_size_t]
        L.orc_calendar.restype = C.c_int
        L.orc_calendar.argtypes = [C.c_int, C.c_int, _ip, _ip, _ip, _ip]
        L.orc_max_threads.restype = C.c_int

    # ---- scalar pieces -------------------------------------------------
    def solar(self, lat, elv, n, y, sw_in, tc, slop, asp, snow, nd, sw):
        s = Srad()
        self.lib.orc_solar(lat, elv, n, y, sw_in, tc, slop, asp, snow, nd, sw, C.byref(s))
        return np.array([getattr(s, k) for k, _ in Srad._fields_])

    def evap(self, lat, elv, sw, n, y, sw_in, tc, slop, asp, snow, nd):
        e = Etr()
        self.lib.orc_evap(lat, elv, sw, n, y, sw_in, tc, slop, asp, snow, nd, C.byref(e))
        return np.array([getattr(e, k) for k, _ in Etr._fields_])

    def evap_batch(self, lat, elv, sw, n, y, sw_in, tc, slop, asp, snow, nd):
        m = len(lat)
        out = np.empty((m, 11))
        f = [_d(x) for x in (lat, elv, sw)]
        i = [_i(x) for x in (n, y)]
        g = [_d(x) for x in (sw_in, tc, slop, asp, snow, nd)]
        self.lib.orc_evap_batch(m, f[0][1], f[1][1], f[2][1], i[0][1], i[1][1],
                                *[x[1] for x in g], out.ctypes.data_as(_dp))
        return out

    def solar_batch(self, lat, elv, n, y, sw_in, tc, slop, asp, snow, nd, sw):
        m = len(lat)
        out = np.empty((m, 8))
        f = [_d(x) for x in (lat, elv)]
        i = [_i(x) for x in (n, y)]
        g = [_d(x) for x in (sw_in, tc, slop, asp, snow, nd, sw)]
        self.lib.orc_solar_batch(m, f[0][1], f[1][1], i[0][1], i[1][1], *[x[1] for x in g],
                                 out.ctypes.data_as(_dp))
        return out

    def viscosity(self, tc, p):
        return self.lib.orc_viscosity_h2o(C.c_float(tc), C.c_float(p))

    def orbit(self, n, y):
        v = [C.c_double() for _ in range(5)]
        self.lib.orc_orbit(n, y, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)  # kN, nu, lam, dr, delta

    def run_one_day(self, lat, elv, n, y, wn, sw_in, tc, pn, slop, asp, snow, snowfall, soil_info,
                    qin, td, nd):
        si, sip = _d(soil_info)
        s, e = Smr(), Etr()
        self.lib.orc_run_one_day(lat, elv, n, y, wn, sw_in, tc, pn, slop, asp, snow, snowfall, sip,
                                 qin, td, nd, C.byref(s), C.byref(e))
        return ({k: getattr(s, k) for k, _ in Smr._fields_},
                {k: getattr(e, k) for k, _ in Etr._fields_})

    def fragile_events(self, reset=True):
        """noise-denominator events (DESIGN "Conditioning") counted on this thread since the last reset"""
        self.lib.orc_fragile_events.restype = C.c_long
        return int(self.lib.orc_fragile_events(1 if reset else 0))

    def spin_up(self, lat, elv, n, y, sw_in, tair, pn, slop, asp, snowfall, soil_info):
        a = [_d(x) for x in (sw_in, tair, pn, snowfall, soil_info)]
        outs = {k: np.empty(n) for k in ("sm", "snow", "qin", "tdrain", "ro", "snwage", "pet")}
        passes = self.lib.orc_spin_up(lat, elv, n, y, a[0][1], a[1][1], a[2][1], slop, asp,
                                      a[3][1], a[4][1],
                                      *[outs[k].ctypes.data_as(_dp) for k in outs])
        outs["passes"] = passes
        return outs

    def run_all(self, lat, elv, doys, yrs, sw_in, tair, pn, wn_last, slop, asp, snow_last,
                snowfall, soil_info, qin_last, td_last, nds_last):
        nt = len(doys)
        d, dp_ = _i(doys)
        y, yp = _i(yrs)
        a = [_d(x) for x in (sw_in, tair, pn, snowfall, soil_info)]
        names = ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev", "tdrain",
                 "snwage")
        outs = {k: np.empty(nt) for k in names}
        self.lib.orc_run_all(lat, elv, nt, dp_, yp, a[0][1], a[1][1], a[2][1], wn_last, slop, asp,
                             snow_last, a[3][1], a[4][1], qin_last, td_last, nds_last,
                             *[outs[k].ctypes.data_as(_dp) for k in names])
        return outs

    def soil_hydro(self, sand, clay, OM, fgravel, bd):
        s = Soil()
        self.lib.orc_soil_hydro(sand, clay, OM, fgravel, bd, C.byref(s))
        return {k: getattr(s, k) for k, _ in Soil._fields_}

    def soil_info(self, soil_data, au, resolution):
        sd, sdp = _d(soil_data)
        out = np.empty(12)
        w = C.c_double()
        aup = None
        if au is not None:
            au_, aup = _d(au)
        self.lib.orc_soil_info(sdp, aup, resolution, out.ctypes.data_as(_dp), C.byref(w))
        return out, w.value

    def snow_partition(self, tc, pn, month, lat, elev):
        nt = len(tc)
        t, tp = _d(tc)
        p, pp = _d(pn)
        m, mp = _i(month)
        rain, snow = np.empty(nt), np.empty(nt)
        Tt = self.lib.orc_snow_partition(nt, tp, pp, mp, lat, elev, rain.ctypes.data_as(_dp),
                                         snow.ctypes.data_as(_dp))
        return Tt, rain, snow

    def unswc(self, soil_data, uns_depth, wn):
        """R/unsSWC.R: dict(wn_z, wtd, theta_i) + 'cond' (cancellation factor of the integral)"""
        sd, sdp = _d(soil_data)
        w, wp = _d(wn)
        nt = len(w)
        o = {k: np.empty(nt) for k in ("wn_z", "wtd", "theta_i", "cond")}
        self.lib.orc_unswc(sdp, float(uns_depth), nt, wp, 1, *[o[k].ctypes.data_as(_dp) for k in o], 1)
        return o

    def terrain_gdal(self, elev, xres, yres, scale=111120.0, round_int=True):
        """the reference's large-grid terrain (gdaldem on the INT2S DEM, R/splash.grid.R:83-84): slope, aspect [H, W]"""
        e, ep = _d(elev)
        H, W = e.shape
        sl, asp = np.empty((H, W)), np.empty((H, W))
        self.lib.orc_terrain_gdal(H, W, ep, float(xres), float(yres), float(scale), int(bool(round_int)),
                                  sl.ctypes.data_as(_dp), asp.ctypes.data_as(_dp))
        return sl, asp

    def unswc_grid(self, soil_data, uns_depth, wn):
        """R/unsSWC.grid.R for one cell: dict(w_z, wtd, Se, theta_mean) + 'cond'"""
        sd, sdp = _d(soil_data)
        w, wp = _d(wn)
        nt = len(w)
        o = {k: np.empty(nt) for k in ("w_z", "wtd", "Se", "theta_mean", "cond")}
        self.lib.orc_unswc_grid(sdp, float(uns_depth), nt, wp, 1, *[o[k].ctypes.data_as(_dp) for k in o], 1)
        return o

    # ---- drivers ---------------------------------------------------------
    def calendar(self, year0, nt):
        doy, year, month, mid = (np.empty(nt, np.int32) for _ in range(4))
        nm = self.lib.orc_calendar(year0, nt, *[x.ctypes.data_as(_ip) for x in (doy, year, month, mid)])
        return doy, year, month, mid, nm

    def point(self, doy, year, month, month_id, nmonth, sw_in, tc, pn, lat, elev, slop, asp,
              soil_data, au, resolution, monthly_out=False):
        nt = len(doy)
        ia = [_i(x) for x in (doy, year, month, month_id)]
        fa = [_d(x) for x in (sw_in, tc, pn)]
        sd, sdp = _d(soil_data)
        aup = None
        if au is not None and np.size(au) == 3:
            au_, aup = _d(au)
        nt_out = nmonth if monthly_out else nt
        out = np.empty((9, nt_out))
        ptrs = (_dp * 9)(*[out[k].ctypes.data_as(_dp) for k in range(9)])
        opt = Opts(int(monthly_out), 1)
        aux = CellAux()
        self.lib.orc_point(nt, ia[0][1], ia[1][1], ia[2][1], ia[3][1], nmonth, fa[0][1], fa[1][1],
                           fa[2][1], 1, lat, elev, slop, asp, sdp, aup, resolution, C.byref(opt),
                           ptrs, 1, C.byref(aux))
        res = {k: out[i] for i, k in enumerate(OUT_NAMES)}
        a = np.frombuffer(bytes(aux), dtype=AUX_DTYPE)[0]
        return res, a

    def grid(self, doy, year, month, month_id, nmonth, sw_in, tc, pn, lat, elev, slop, asp, soil,
             au, resolution, monthly_out=False, n_threads=0, want_aux=True):
        nt, ncell = sw_in.shape
        ia = [_i(x) for x in (doy, year, month, month_id)]
        fa = [_d(x) for x in (sw_in, tc, pn, lat, elev, slop, asp, soil)]
        aup = None
        if au is not None:
            au_, aup = _d(au)
        rs, rsp = _d(resolution)
        nt_out = nmonth if monthly_out else nt
        out = np.empty((9, nt_out, ncell))
        ptrs = (_dp * 9)(*[out[k].ctypes.data_as(_dp) for k in range(9)])
        opt = Opts(int(monthly_out), n_threads)
        aux = np.zeros(ncell, dtype=AUX_DTYPE) if want_aux else None
        self.lib.orc_grid(ncell, nt, ia[0][1], ia[1][1], ia[2][1], ia[3][1], nmonth,
                          *[x[1] for x in fa], aup, rsp, C.byref(opt), ptrs,
                          aux.ctypes.data if want_aux else None)
        return {k: out[i] for i, k in enumerate(OUT_NAMES)}, aux

    def grid_transient(self, doy, year, month, sw_in, tc, pn, lat, elev, slop, asp, soil_info, Tt,
                       state, n_threads=0, store=True):
        nt, ncell = sw_in.shape
        ia = [_i(x) for x in (doy, year, month)]
        fa = [_d(x) for x in (sw_in, tc, pn, lat, elev, slop, asp, soil_info, Tt)]
        st = np.array(state, dtype=np.float64, order="C", copy=True)
        out = np.empty((5, nt, ncell)) if store else None
        if store:
            ptrs = (_dp * 5)(*[out[k].ctypes.data_as(_dp) for k in range(5)])
        else:
            ptrs = (_dp * 5)(*[None] * 5)
        self.fragile = np.zeros(ncell, np.int32)
        self.lib.orc_grid_transient(ncell, nt, ia[0][1], ia[1][1], ia[2][1], *[x[1] for x in fa],
                                    st.ctypes.data_as(_dp), ptrs, n_threads,
                                    self.fragile.ctypes.data_as(_ip))
        return out, st

    SERIES11 = ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "qin_prev", "tdrain",
                "snwage")

    def grid_transient2(self, doy, year, month, sw_in, tc, pn, lat, elev, slop, asp, soil_info, Tt,
                        state, prev=None, names=SERIES11, hs_lim_deg=HS_LIM_DEG, want_hs=False,
                        n_threads=0):
        """run_all for every cell from `state` with the per-cell-day conditioning events.
        prev = [5, nt, ncell] state series (wn, snow, qin, td, nd after each day) switches to
        one-step mode: day t starts from prev[:, t-1] (see orc_grid_transient2).
        Returns (dict name -> [nt, ncell], state_end [5, ncell], events uint8 [nt, ncell], hs|None)."""
        nt, ncell = sw_in.shape
        ia = [_i(x) for x in (doy, year, month)]
        fa = [_d(x) for x in (sw_in, tc, pn, lat, elev, slop, asp, soil_info, Tt)]
        st = np.array(state, dtype=np.float64, order="C", copy=True)
        pv = None
        if prev is not None:
            pv = np.ascontiguousarray(prev, dtype=np.float64)
            assert pv.shape == (5, nt, ncell)
        out = {k: np.empty((nt, ncell)) for k in names}
        ptrs = (_dp * 11)(*[out[k].ctypes.data_as(_dp) if k in out else None for k in self.SERIES11])
        ev = np.zeros((nt, ncell), np.uint8)
        hs = np.empty((nt, ncell), np.float32) if want_hs else None
        self.lib.orc_grid_transient2(ncell, nt, ia[0][1], ia[1][1], ia[2][1], *[x[1] for x in fa],
                                     st.ctypes.data_as(_dp),
                                     pv.ctypes.data_as(_dp) if pv is not None else None, ptrs,
                                     ev.ctypes.data, float(hs_lim_deg),
                                     hs.ctypes.data if want_hs else None, n_threads)
        return out, st, ev, hs

    def terrain(self, elev, ymax, yres, xres, na_to_zero=True):
        """R/splash.grid.R:96-110 + flow direction and ncellflow of R/upslope_area.R:31-33, 140-165"""
        e, ep = _d(elev)
        H, W = e.shape
        o = {k: np.empty((H, W)) for k in ("slope", "aspect", "lat", "resolution", "flowdir")}
        self.lib.orc_terrain(H, W, ep, ymax, yres, xres, int(na_to_zero),
                             *[o[k].ctypes.data_as(_dp) for k in ("slope", "aspect", "lat", "resolution", "flowdir")])
        for name, codes in (("ncellin", (2, 1, 128, 4, 0, 64, 8, 16, 32)),
                            ("ncellout", (32, 16, 8, 64, 0, 4, 128, 1, 2))):
            c9, c9p = _d(np.array(codes, dtype=np.float64))
            o[name] = np.empty((H, W))
            self.lib.orc_ncellflow(H, W, o["flowdir"].ctypes.data_as(_dp), c9p, 0, o[name].ctypes.data_as(_dp))
        return o

    def sinkfill(self, elev, resolution_m, degree=1.0):
        """Planchon & Darboux relaxation (serial restatement of the published algorithm)"""
        e, ep = _d(elev)
        H, W = e.shape
        out = np.empty((H, W))
        self.lib.orc_sinkfill(H, W, ep, C.c_double(resolution_m), C.c_double(degree), out.ctypes.data_as(_dp))
        return out

    def mfd_area(self, elev, resolution_m):
        """Quinn et al. (1991) multiple-flow-direction contributing area (serial restatement)"""
        e, ep = _d(elev)
        H, W = e.shape
        out = np.empty((H, W))
        self.lib.orc_mfd_area(H, W, ep, C.c_double(resolution_m), out.ctypes.data_as(_dp))
        return out

    def flow_accumulation(self, flowdir, elev=None, resolution=None, tau=False):
        f, fp = _d(flowdir)
        H, W = f.shape
        ep = rp = None
        if elev is not None:
            e_, ep = _d(elev)
        cnt, area = np.empty((H, W)), None
        ap = None
        if resolution is not None:
            r_, rp = _d(resolution)
            area = np.empty((H, W))
            ap = area.ctypes.data_as(_dp)
        self.lib.orc_flow_accumulation(H, W, fp, ep, int(bool(tau)), rp, cnt.ctypes.data_as(_dp), ap)
        return cnt, area

    def synth_statics(self, seed, H, W, cell0, ncell, terrain):
        lat, elev, slop, asp, res = (np.empty(ncell) for _ in range(5))
        soil = np.empty((6, ncell))
        au = np.empty((3, ncell))
        self.lib.orc_synth_statics(seed, H, W, cell0, ncell, int(terrain),
                                   *[x.ctypes.data_as(_dp) for x in (lat, elev, slop, asp, res, soil, au)])
        return dict(lat=lat, elev=elev, slop=slop, asp=asp, resolution=res, soil=soil, au=au)

    def synth_forcing(self, seed, H, W, cell0, ncell, t0, doy, lat, elev):
        nt = len(doy)
        d, dp_ = _i(doy)
        la, lap = _d(lat)
        el, elp = _d(elev)
        sw, tc, pn = (np.empty((nt, ncell)) for _ in range(3))
        self.lib.orc_synth_forcing(seed, H, W, cell0, ncell, t0, nt, dp_, lap, elp,
                                   *[x.ctypes.data_as(_dp) for x in (sw, tc, pn)])
        return sw, tc, pn

    def threads_obtained(self, n):
        """threads an OpenMP region asked for n really gets"""
        return int(self.lib.orc_threads_obtained(int(n)))

    def max_threads(self):
        return self.lib.orc_max_threads()


def grazing_mask(oracle, lat, slop, asp_c, doy, year, hs_deg=0.01):
    """Cell-days on which the reference itself is ill-conditioned (numpy, approximate).

    When the sun only grazes the slope, the slope sunset angle hs of src/SOLAR.cpp:129-159 is
    ~0 although not exactly 0: ru/rv = -cos(hs) = -1 + O(hs^2), the daily geometry integral
    ru*hs + rv*sin(hs) (src/SOLAR.cpp:165) cancels to O(hs^3) and src/SOLAR.cpp:223 divides by
    it, so rw, and with it every flux of that day, carries relative rounding noise of order
    1e-16/hs^2.  Two correct libm implementations (glibc vs the GPU's) then differ by far more
    than 1e-5 on that cell-day -- as the reference does between two of its own builds -- and the
    cell's state carries the difference forward.  hs depends only on (lat, slope, aspect, day).
    Returns a bool array [nt, ncell]: 0 < hs < hs_deg.   asp_c is asp-180."""
    pir = np.pi / 180.0
    orb = np.array([oracle.orbit(int(n), int(y)) for n, y in zip(doy, year)])
    delta = orb[:, 4]
    sd, cd = np.sin(delta * pir)[:, None], np.cos(delta * pir)[:, None]
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
        ruv = (-a + c * sinfirst) / b
    lim = 1.0 - np.cos(hs_deg * pir)
    return (ruv > -1.0) & (1.0 + ruv < lim)


class Ref:
    """The real reference SOLAR/EVAP classes (oracle/_ref); raises if it was never built."""

    def __init__(self, variant=None):
        """variant: None = the pinned build; "B" / "K" = the other legal builds of the same sources
        (oracle/Makefile `ref-variants`; tools/conditioning_real_reference.py)"""
        path = os.path.join(_HERE, "_ref", f"libsplash_ref{'_' + variant if variant else ''}.so")
        if not os.path.exists(path):
            build()
            if variant:
                subprocess.check_call(["make", "-C", _HERE, "ref-variants"], stdout=subprocess.DEVNULL)
        self.lib = L = C.CDLL(path)
        L.ref_solar.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int] + [C.c_double] * 7 + [_dp]
        L.ref_evap.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int] + \
            [C.c_double] * 6 + [_dp]
        L.ref_evap_batch.argtypes = [C.c_int, _dp, _dp, _dp, _ip, _ip] + [_dp] * 7
        L.ref_solar_batch.argtypes = [C.c_int, _dp, _dp, _ip, _ip] + [_dp] * 8

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libsplash_ref.so")) or \
            os.path.isdir("/root/reference/src")

    def evap_batch(self, lat, elv, sw, n, y, sw_in, tc, slop, asp, snow, nd):
        m = len(lat)
        out = np.empty((m, 11))
        f = [_d(x) for x in (lat, elv, sw)]
        i = [_i(x) for x in (n, y)]
        g = [_d(x) for x in (sw_in, tc, slop, asp, snow, nd)]
        self.lib.ref_evap_batch(m, f[0][1], f[1][1], f[2][1], i[0][1], i[1][1],
                                *[x[1] for x in g], out.ctypes.data_as(_dp))
        return out

    def solar_batch(self, lat, elv, n, y, sw_in, tc, slop, asp, snow, nd, sw):
        m = len(lat)
        out = np.empty((m, 8))
        f = [_d(x) for x in (lat, elv)]
        i = [_i(x) for x in (n, y)]
        g = [_d(x) for x in (sw_in, tc, slop, asp, snow, nd, sw)]
        self.lib.ref_solar_batch(m, f[0][1], f[1][1], i[0][1], i[1][1], *[x[1] for x in g],
                                 out.ctypes.data_as(_dp))
        return out
