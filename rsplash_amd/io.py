"""Output side of splash.grid: the nine result bricks as NetCDF files with the reference's file
names, variable names, units and axis names (R/splash.grid.R:186-216: `brick(..., filename=
paste0(outdir,"/SPLASH_v2.0_",name,"_",time_text,"_",y[1],"-",y[length(y)],".nc"), format="CDF",
varname=..., varunit=..., longname=..., xname="lon", yname="lat", zname="time")`, written block by
block at :370-400 and closed at :409-417).

The reference writes through raster/ncdf4; this image has neither, so the files are NetCDF-3
classic written with scipy.io.netcdf_file (any NetCDF reader, raster::brick() included, opens
them).  Values are stored as float32 like raster's default FLT4S with its NA flag as _FillValue.
Host I/O only -- nothing here touches the GPU.
"""
import os

import numpy as np
from scipy.io import netcdf_file

# result key -> (file tag, varname, unit, longname), R/splash.grid.R:190-214
BRICKS = {
    "wn": ("swc", "wm", "mm", "Soil water content"),
    "ro": ("ro", "ro", "mm", "surface runoff"),
    "pet": ("pet", "pet", "mm", "potential evapotranspiration"),
    "aet": ("aet", "aet", "mm", "actual evapotranspiration"),
    "snow": ("swe", "swe", "mm", "snow water equivalent"),
    "cond": ("cond", "cond", "mm", "condensation"),
    "bflow": ("bflow", "bflow", "mm", "drainage"),
    "netr": ("netrad", "netrad", "MJ", "daytime net radiation"),
    "sm_lim": ("sm_lim", "sm_lim", "mm", "fraction of available water content"),
}
NA_FLT4S = np.float32(-3.4e38)  # raster's NA flag for FLT4S


def output_time_axis(time_index, monthly_out):
    """ztime_out (R/splash.grid.R:161-181): the first of every month, or every day, of the years
    covered; values are days since (first year - 1)-12-31 (`zunit`, :164)."""
    t = np.asarray(time_index, dtype="datetime64[D]")
    y0, y1 = int(str(t[0])[:4]), int(str(t[-1])[:4])
    origin = np.datetime64(f"{y0 - 1:04d}-12-31")
    if monthly_out:
        z = np.arange(f"{y0:04d}-01", f"{y1 + 1:04d}-01", dtype="datetime64[M]").astype("datetime64[D]")
    else:
        z = np.arange(f"{y0:04d}-01-01", f"{y1 + 1:04d}-01-01", dtype="datetime64[D]")
    return z, (z - origin).astype(np.int64), f"days since {y0 - 1}-12-31", y0, y1


def brick_path(outdir, key, monthly_out, y0, y1):
    tag = BRICKS[key][0]
    return os.path.join(outdir, f"SPLASH_v2.0_{tag}_{'monthly' if monthly_out else 'daily'}_{y0}-{y1}.nc")


def write_bricks(outdir, res, lon, lat, time_index, monthly_out):
    """Write every output of `res` that is one of the nine bricks.  res[key]: [nl, nrow, ncol]
    (row 0 = northernmost, as raster stores it); lon [ncol], lat [nrow] cell centres.
    Returns {key: path}."""
    z, zval, zunit, y0, y1 = output_time_axis(time_index, monthly_out)
    lon, lat = np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64)
    paths = {}
    for key, (tag, varname, unit, longname) in BRICKS.items():
        if key not in res:
            continue
        a = np.asarray(res[key])
        if a.ndim != 3 or a.shape[1:] != (lat.size, lon.size):
            raise ValueError(f"{key}: expected [nl, {lat.size}, {lon.size}], got {a.shape}")
        if a.shape[0] != z.size:
            raise ValueError(f"{key}: {a.shape[0]} layers for a time axis of {z.size} "
                             "(the inputs must cover whole calendar years, R/splash.grid.R:161-166)")
        path = brick_path(outdir, key, monthly_out, y0, y1)
        with netcdf_file(path, "w", version=2) as f:
            f.createDimension("time", z.size)
            f.createDimension("lon", lon.size)
            f.createDimension("lat", lat.size)
            vx = f.createVariable("lon", "d", ("lon",))
            vx.units, vx.long_name = "degrees_east", "lon"
            vy = f.createVariable("lat", "d", ("lat",))
            vy.units, vy.long_name = "degrees_north", "lat"
            vt = f.createVariable("time", "i", ("time",))
            vt.units, vt.long_name = zunit, "time"
            v = f.createVariable(varname, "f", ("time", "lat", "lon"))
            v.units, v.long_name = unit, longname
            v._FillValue = NA_FLT4S
            v.missing_value = NA_FLT4S
            vx[:], vy[:] = lon, lat
            vt[:] = zval.astype(np.int32)
            v[:] = np.where(np.isnan(a), NA_FLT4S, a).astype(np.float32)
            f.title = "SPLASH v2.0 " + longname
        paths[key] = path
    return paths
