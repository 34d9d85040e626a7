"""ctypes prototypes for include/splash_hip.h.  Fails loudly when the library is absent."""
import ctypes as C
import os

from .build import library_path

NPAR, NSTATE, NOUT, NDIAG, NEXTRA = 17, 5, 9, 6, 3
DIAG_REC = (NDIAG + 1) * 2  # int64 limbs per (basin, month) record, see include/splash_hip.h
EXTRA_NAMES = ("qin_prev", "tdrain", "snwage")
OUT_NAMES = ("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr", "sm_lim")
DIAG_NAMES = ("precip", "aet", "ro", "bflow", "wn", "snow")

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_int64)
_vp = C.c_void_p


class GridIn(C.Structure):
    _fields_ = [("ncell", C.c_size_t), ("nt", C.c_int), ("doy", _ip), ("year", _ip), ("month", _ip),
                ("sw_in", _dp), ("tc", _dp), ("pn", _dp), ("lat", _dp), ("elev", _dp),
                ("slop", _dp), ("asp", _dp), ("resolution", _dp), ("soil", _dp), ("au", _dp),
                ("basin", _ip), ("n_basin", C.c_int),
                ("sw_in_f32", _fp), ("tc_f32", _fp), ("pn_f32", _fp),
                ("tt", _dp), ("state0", _dp), ("ai0", _dp)]          # ABI 5


class Opts(C.Structure):
    _fields_ = [("monthly_out", C.c_int), ("device", C.c_int), ("max_cells_per_launch", C.c_size_t),
                ("n_devices", C.c_int), ("devices", _ip), ("mem_fraction", C.c_double), ("backend", C.c_int)]


class GridOut(C.Structure):
    _fields_ = [("out", _dp * NOUT), ("state_end", _dp), ("diag", _dp), ("aux", _dp),
                ("diag_acc", _lp), ("out_f32", _fp * NOUT)]


class DevOut(C.Structure):
    _fields_ = [("out", _vp * NOUT), ("ldo", C.c_size_t), ("monthly", C.c_int),
                ("extra", _vp * NEXTRA), ("diag_acc", _vp), ("basin", _vp), ("n_basin", C.c_int),
                ("skip", _vp), ("cells", _vp)]


STATIC_NAMES = ("slope", "aspect", "lat", "resolution", "filled", "flowdir", "ups_ncell", "ups_area",
                "ncellin", "ncellout", "ups_area_mfd")


class StaticOut(C.Structure):  # splash_static_out
    _fields_ = [(k, _dp) for k in STATIC_NAMES]


def _load():
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with rsplash_amd.build_library() "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64/libhsa-runtime64, this
    # library links the system's.  Whichever is loaded FIRST serves both (same SONAME); loaded the
    # other way round the process ends up with two HSA runtimes and the second sees no device.
    # So when PyTorch is there (it is the device-memory plumbing of the Python mirror), it goes first.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(path)
    L.splash_hip_abi_version.restype = C.c_int
    L.splash_hip_device_count.restype = C.c_int
    L.splash_hip_last_error.restype = C.c_char_p
    L.splash_hip_release_cache.restype = None
    L.splash_hip_cache_stats.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.splash_calendar_create.argtypes = [C.c_int, _ip, _ip, _ip, C.POINTER(_vp), C.c_char_p, C.c_size_t]
    L.splash_calendar_destroy.argtypes = [_vp]
    L.splash_calendar_destroy.restype = None
    L.splash_calendar_nt.argtypes = [_vp]
    L.splash_calendar_nmonth.argtypes = [_vp]
    L.splash_calendar_orbit.argtypes = [_vp, _dp, _dp, _dp]
    L.splash_orbit_day.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
    L.splash_dev_soil_hydro.argtypes = [C.c_size_t] + [_vp] * 8 + [_vp]
    L.splash_dev_snow_threshold.argtypes = [C.c_size_t, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t,
                                            _vp, _vp]
    L.splash_dev_spin_up_work_size.restype = C.c_size_t
    L.splash_dev_spin_up_work_size.argtypes = [C.c_size_t]
    L.splash_dev_spin_up.argtypes = [_vp, C.c_size_t, _vp, _vp, _vp, C.c_size_t, _vp, _vp,
                                     C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp]
    for fn in ("splash_dev_spin_up_begin", "splash_dev_spin_up_finish"):
        getattr(L, fn).argtypes = L.splash_dev_spin_up.argtypes
    L.splash_dev_spin_up_pending.argtypes = [_vp, C.c_size_t, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]
    L.splash_dev_spin_up_pending_count.argtypes = [_vp, C.c_size_t, _vp, C.POINTER(C.c_uint32)]
    L.splash_dev_diag_acc_size.restype = C.c_size_t
    L.splash_dev_diag_acc_size.argtypes = [C.c_int, C.c_int]
    L.splash_dev_diag_finalize.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _vp]
    L.splash_diag_finalize_host.argtypes = [_lp, C.c_int, C.c_int, _dp, _dp]
    L.splash_diag_finalize_host.restype = None
    L.splash_dev_infiltration_probe.argtypes = [C.c_size_t] + [_dp] * 6
    L.splash_dev_terrain.argtypes = [C.c_int, C.c_int, _vp, C.c_double, C.c_double, C.c_double, C.c_int,
                                     _vp, _vp, _vp, _vp, _vp, _vp]
    L.splash_dev_ncellflow.argtypes = [C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp]
    if not (os.environ.get("SPLASH_HIP_LIBRARY") and not hasattr(L, "splash_dev_sinkfill")):
        # (an A/B build of an earlier commit, selected by SPLASH_HIP_LIBRARY, may predate these three)
        L.splash_static_rasters_run.argtypes = [C.c_int, C.c_int, _dp, C.c_double, C.c_double, C.c_double,
                                                C.c_int, C.c_double, C.POINTER(StaticOut), C.c_char_p,
                                                C.c_size_t]
        L.splash_dev_sinkfill.argtypes = [C.c_int, C.c_int, _vp, C.c_double, C.c_double, _vp,
                                          C.POINTER(C.c_int), _vp]
        L.splash_dev_mfd_area.argtypes = [C.c_int, C.c_int, _vp, C.c_double, _vp, C.POINTER(C.c_int), _vp]
        L.splash_dev_flow_accumulation.argtypes = [C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp,
                                                   C.POINTER(C.c_int), _vp]
    L.splash_dev_run_all.argtypes = [_vp, C.c_int, C.c_int, C.c_size_t, _vp, _vp, _vp, C.c_size_t,
                                     _vp, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(DevOut), _vp]
    L.splash_dev_synth_statics.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_size_t,
                                           C.c_int] + [_vp] * 7 + [_vp]
    L.splash_dev_synth_forcing.argtypes = [C.c_uint64, _vp, C.c_int, C.c_int, C.c_int64,
                                           C.c_size_t, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]
    L.splash_dev_evap_batch.argtypes = [C.c_size_t, _dp, _dp, _dp, _ip, _ip] + [_dp] * 7 + [_vp]
    L.splash_dev_unswc.argtypes = [C.c_size_t, C.c_int, _vp, C.c_size_t, _vp, C.c_double, _vp, _vp,
                                   _vp, C.c_size_t, _vp]
    L.splash_unswc_run.argtypes = [C.c_size_t, C.c_int, _dp, _dp, C.c_double, _dp, _dp, _dp,
                                   C.c_char_p, C.c_size_t]
    L.splash_unswc_run_backend.argtypes = [C.c_int, C.c_size_t, C.c_int, _dp, _dp, C.c_double, _dp, _dp, _dp,
                                           C.c_char_p, C.c_size_t]
    L.splash_terrain_gdal_run.argtypes = [C.c_int, C.c_int, _dp, C.c_double, C.c_double, C.c_double, C.c_int, _dp, _dp,
                                          C.c_char_p, C.c_size_t]
    L.splash_dev_terrain_gdal.argtypes = [C.c_int, C.c_int, _vp, C.c_double, C.c_double, C.c_double, C.c_int, _vp, _vp, _vp]
    L.splash_day_step.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int] + [C.c_double] * 6 + [_dp, _dp, _dp]
    L.splash_soil_hydro.argtypes = [C.c_size_t] + [_dp] * 8 + [C.c_char_p, C.c_size_t]
    L.splash_unswc_grid_run.argtypes = [C.c_int, C.c_size_t, C.c_int, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp,
                                        C.c_char_p, C.c_size_t]
    L.splash_dev_unswc_grid.argtypes = [C.c_size_t, C.c_int, _vp, C.c_size_t, _vp, C.c_double, _vp, _vp, _vp, _vp,
                                        C.c_size_t, _vp]
    L.splash_dev_math_probe.argtypes = [C.c_size_t, _dp, _dp, _dp]
    L.splash_snow_threshold_run.argtypes = [C.c_size_t, C.c_int, _dp, _fp, _dp, _dp, _dp, C.c_char_p, C.c_size_t]
    L.splash_snow_threshold_run_backend.argtypes = [C.c_int, C.c_size_t, C.c_int, _dp, _fp, _dp, _dp, _dp, C.c_char_p,
                                                    C.c_size_t]
    L.splash_grid_submit.argtypes = [C.POINTER(GridIn), C.POINTER(Opts), C.POINTER(GridOut),
                                     C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    L.splash_grid_wait.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.splash_grid_run.argtypes = [C.POINTER(GridIn), C.POINTER(Opts), C.POINTER(GridOut),
                                  C.c_char_p, C.c_size_t]
    L.splash_point_run.argtypes = [C.c_int, _ip, _ip, _ip, _dp, _dp, _dp] + [C.c_double] * 4 + \
        [_dp, _dp, C.c_int, C.c_double, C.c_int, C.POINTER(_dp), C.c_char_p, C.c_size_t]
    L.splash_point_run_backend.argtypes = [C.c_int] + L.splash_point_run.argtypes
    L.splash_mod_backend.argtypes = [C.c_int]
    L.splash_mod_spin_up.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, _dp, _dp, _dp,
                                     C.c_double, C.c_double, _dp, _dp] + [_dp] * 7 + [_ip]
    L.splash_mod_run_all.argtypes = [C.c_double, C.c_double, C.c_int, _ip, _ip, _dp, _dp, _dp,
                                     C.c_double, C.c_double, C.c_double, C.c_double, _dp, _dp,
                                     C.c_double, C.c_double, C.c_double] + [_dp] * 11
    L.splash_mod_run_one_year.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, _dp, _dp,
                                          _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp, _dp,
                                          _dp] + [_dp] * 11
    L.splash_mod_moist_surf.argtypes = [C.c_double] * 7 + [_dp]
    L.splash_mod_inf_GA.argtypes = [C.c_double] * 8 + [_dp]
    return L


lib = _load()

# every symbol include/splash_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = (
    "splash_hip_abi_version", "splash_hip_device_count", "splash_hip_last_error",
    "splash_hip_release_cache", "splash_hip_cache_stats",
    "splash_calendar_create", "splash_calendar_destroy", "splash_calendar_nt",
    "splash_calendar_nmonth", "splash_calendar_orbit", "splash_orbit_day", "splash_dev_soil_hydro",
    "splash_dev_snow_threshold", "splash_dev_spin_up_work_size", "splash_dev_spin_up",
    "splash_dev_spin_up_begin", "splash_dev_spin_up_finish", "splash_dev_spin_up_pending",
    "splash_dev_spin_up_pending_count",
    "splash_dev_diag_acc_size", "splash_dev_diag_finalize", "splash_diag_finalize_host",
    "splash_dev_infiltration_probe", "splash_dev_terrain", "splash_dev_ncellflow",
    "splash_dev_sinkfill", "splash_dev_flow_accumulation", "splash_dev_mfd_area", "splash_static_rasters_run",
    "splash_dev_run_all", "splash_dev_synth_statics", "splash_dev_synth_forcing",
    "splash_dev_evap_batch", "splash_dev_math_probe", "splash_dev_unswc", "splash_unswc_run", "splash_unswc_run_backend", "splash_hip_version", "splash_terrain_gdal_run", "splash_dev_terrain_gdal", "splash_day_step", "splash_soil_hydro", "splash_unswc_grid_run", "splash_dev_unswc_grid",
    "splash_snow_threshold_run", "splash_snow_threshold_run_backend", "splash_grid_run", "splash_grid_submit", "splash_grid_wait", "splash_point_run", "splash_mod_spin_up",
    "splash_mod_run_all", "splash_mod_run_one_year", "splash_mod_moist_surf", "splash_mod_inf_GA",
    "splash_point_run_backend", "splash_host_backend_fma", "splash_mod_backend",
)

BACKENDS = {"hip": 0, "host": 1, 0: 0, 1: 1}   # SPLASH_BACKEND_HIP / SPLASH_BACKEND_HOST (include/splash_hip.h)


def backend_code(b):
    try:
        return BACKENDS[b]
    except KeyError:
        raise ValueError(f"backend must be 'hip' (default) or 'host', not {b!r}") from None


class SplashError(RuntimeError):
    pass


def check(rc, where=""):
    if rc != 0:
        msg = lib.splash_hip_last_error()
        raise SplashError(f"{where} failed ({rc}): {msg.decode() if msg else ''}")
