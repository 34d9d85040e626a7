"""Output side of splash.grid: the nine result bricks as NetCDF files with the reference's file
names, variable names, units and axis names (R/splash.grid.R:321-337 `writeStart(..., filename=...)`, 186-216: `brick(..., filename=
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

# result key -> (file tag, varname, unit, longname), R/splash.grid.R:321-337: the names of the writeStart() calls, which
# open the files the blocks are written into.  (The brick() calls at :190-214 carry the same names except for the
# soil water, "swc" there and "sm" at :321 -- but with values=FALSE raster::brick() returns an empty brick and writes
# nothing: the file a run leaves behind is SPLASH_v2.0_sm_*.nc.  Rounds 1-5 used "swc" here.)
BRICKS = {
    "wn": ("sm", "wm", "mm", "Soil water content"),
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


def _write_one(path, a, varname, unit, longname, lon, lat, zval, zunit):
    """one [nl, nrow, ncol] brick as a NetCDF file laid out like raster's writeRaster(format="CDF", xname="lon",
    yname="lat", zname="time") output (R/splash.grid.R:190-216)"""
    with netcdf_file(path, "w", version=2) as f:
        f.createDimension("time", len(zval))
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
        vt[:] = np.asarray(zval).astype(np.int32)
        with np.errstate(over="ignore"):   # FLT4S: a double beyond the float range is stored as inf, as raster stores it
            v[:] = np.where(np.isnan(a), NA_FLT4S, a).astype(np.float32)
        f.title = "SPLASH v2.0 " + longname


# the four rasters of unSWC.grid (R/unsSWC.grid.R:87, 108, 115, 126): key -> (file tag, varname, unit, longname);
# "{d}" is the caller's d_name (the depth as text, e.g. "0.49")
UNSWC_BRICKS = {
    "theta_mean": ("theta_mean", "theta", "m3/m3", "volumetric soil moisture"),
    "wtd": ("wtd", "wtd", "m", "water table depth"),
    "w_z": ("swc_top_{d}_m", "swc", "mm", "soil water content"),
    "Se": ("Se_top_{d}_m", "Se", "fraction", "effective saturation or water filled porosity"),
}


def write_unswc_bricks(outdir, res, lon, lat, ztime, d_name):
    """The files unSWC.grid writes, with its names: SPLASH_v2.0_<tag>_<daily|monthly>_<y0>-<y1>.nc, where the
    frequency is read off the first two layer dates (< 2 days apart: daily, R/unsSWC.grid.R:19-26) and the years
    off all of them (:76).  res[key]: [nl, nrow, ncol].  The time variable holds the layers' dates as days since
    (first year - 1)-12-31, like the bricks of splash.grid (overlay() drops the Z of its inputs: the reference's
    files number their layers 1..nl).  Returns {key: path}."""
    t = np.asarray(ztime, dtype="datetime64[D]")
    if t.size < 2:
        raise ValueError("unSWC.grid reads the time step off the first two layer dates: at least two layers")
    step = abs(int((t[1] - t[0]).astype(np.int64)))
    time_text = "daily" if step < 2 else "monthly"
    years = np.unique(t.astype("datetime64[Y]").astype(int) + 1970)
    y0, y1 = int(years[0]), int(years[-1])
    origin = np.datetime64(f"{y0 - 1:04d}-12-31")
    zval, zunit = (t - origin).astype(np.int64), f"days since {y0 - 1}-12-31"
    lon, lat = np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64)
    paths = {}
    for key, (tag, varname, unit, longname) in UNSWC_BRICKS.items():
        if key not in res:
            continue
        a = np.asarray(res[key])
        if a.shape != (t.size, lat.size, lon.size):
            raise ValueError(f"{key}: expected [{t.size}, {lat.size}, {lon.size}], got {a.shape}")
        path = os.path.join(outdir, f"SPLASH_v2.0_{tag.format(d=d_name)}_{time_text}_{y0}-{y1}.nc")
        _write_one(path, a, varname, unit, longname, lon, lat, zval, zunit)
        paths[key] = path
    return paths


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
        _write_one(path, a, varname, unit, longname, lon, lat, zval, zunit)
        paths[key] = path
    return paths


# ---------------------------------------------------------------------------------------------------
# Streaming bricks: the file is created once (header + coordinate variables), the data variable is a
# window of the file that row blocks are stored into as they arrive -- what raster's writeStart /
# writeValues(brick, block, row) / writeStop do around the reference's receive loop
# (R/splash.grid.R:186-216, 370-400, 409-417).  scipy's writer keeps a whole variable in memory until
# close(); a brick of the benchmark grid is tens of GB, so the classic format (64-bit offsets) is laid
# out here directly -- header per the NetCDF-3 specification, big-endian data, the data variable last
# so that it may exceed 4 GB -- and mapped with numpy.memmap.
_NC_BYTE, _NC_CHAR, _NC_INT, _NC_FLOAT, _NC_DOUBLE = 1, 2, 4, 5, 6
_NC_DIMENSION, _NC_VARIABLE, _NC_ATTRIBUTE = 0x0A, 0x0B, 0x0C
_NC_DT = {_NC_INT: ">i4", _NC_FLOAT: ">f4", _NC_DOUBLE: ">f8"}


def _pad4(b):
    return b + b"\0" * (-len(b) % 4)


def _be32(v):
    return int(v).to_bytes(4, "big")


def _nc_name(s):
    b = s.encode()
    return _be32(len(b)) + _pad4(b)


def _nc_atts(atts):
    if not atts:
        return _be32(0) + _be32(0)
    out = _be32(_NC_ATTRIBUTE) + _be32(len(atts))
    for k, v in atts.items():
        if isinstance(v, (str, bytes)):
            b = v.encode() if isinstance(v, str) else v
            out += _nc_name(k) + _be32(_NC_CHAR) + _be32(len(b)) + _pad4(b)
        else:
            a = np.atleast_1d(np.asarray(v))
            t = _NC_FLOAT if a.dtype == np.float32 else _NC_INT if a.dtype.kind in "iu" else _NC_DOUBLE
            out += _nc_name(k) + _be32(t) + _be32(a.size) + _pad4(a.astype(_NC_DT[t]).tobytes())
    return out


def nc3_create(path, dims, variables, gatts=None):
    """Lay out a NetCDF-3 file (64-bit offset variant, no record dimension).  dims: ordered
    {name: length}; variables: list of (name, nc_type, dim names, attributes) in file order.  Every
    variable but the last must stay below 4 GiB.  Returns {name: (offset, shape, big-endian dtype)};
    the file has its final size, data all zero."""
    dim_names = list(dims)
    var_sizes = []
    for name, t, vd, atts in variables:
        n = int(np.prod([dims[d] for d in vd])) if vd else 1
        var_sizes.append(n * np.dtype(_NC_DT[t]).itemsize)
    if any(sz >= (1 << 32) - 4 for sz in var_sizes[:-1]):
        raise ValueError("only the last variable of a classic NetCDF file may exceed 4 GiB")

    def header(begins):
        h = b"CDF\x02" + _be32(0)
        h += _be32(_NC_DIMENSION) + _be32(len(dims))
        for d in dim_names:
            h += _nc_name(d) + _be32(dims[d])
        h += _nc_atts(gatts or {})
        h += _be32(_NC_VARIABLE) + _be32(len(variables))
        for (name, t, vd, atts), sz, b in zip(variables, var_sizes, begins):
            h += _nc_name(name) + _be32(len(vd)) + b"".join(_be32(dim_names.index(d)) for d in vd)
            vsize = sz + (-sz % 4)
            h += _nc_atts(atts) + _be32(t) + _be32(min(vsize, 0xFFFFFFFF)) + int(b).to_bytes(8, "big")
        return h

    hlen = len(header([0] * len(variables)))
    begins, off = [], hlen
    for sz in var_sizes:
        begins.append(off)
        off += sz + (-sz % 4)
    with open(path, "wb") as f:
        f.write(header(begins))
        f.truncate(off)
    return {name: (b, tuple(dims[d] for d in vd), np.dtype(_NC_DT[t]))
            for (name, t, vd, atts), b in zip(variables, begins)}


class BrickWriter:
    """The result bricks of one splash.grid run, written row block by row block.  Same files, names,
    variables and attributes as write_bricks(); `write(key, row0, block)` stores block [nl, nrows, ncol]
    at grid rows row0..row0+nrows (NaN -> raster's FLT4S flag), close() flushes and unmaps."""

    def __init__(self, outdir, keys, lon, lat, time_index, monthly_out):
        z, zval, zunit, y0, y1 = output_time_axis(time_index, monthly_out)
        lon, lat = np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64)
        self.nl, self.shape = z.size, (lat.size, lon.size)
        self.paths, self._maps, self._seen = {}, {}, {}
        os.makedirs(outdir, exist_ok=True)
        for key in keys:
            if key not in BRICKS:
                continue
            tag, varname, unit, longname = BRICKS[key]
            path = brick_path(outdir, key, monthly_out, y0, y1)
            lay = nc3_create(
                path, {"time": z.size, "lon": lon.size, "lat": lat.size},
                [("lon", _NC_DOUBLE, ("lon",), {"units": "degrees_east", "long_name": "lon"}),
                 ("lat", _NC_DOUBLE, ("lat",), {"units": "degrees_north", "long_name": "lat"}),
                 ("time", _NC_INT, ("time",), {"units": zunit, "long_name": "time"}),
                 (varname, _NC_FLOAT, ("time", "lat", "lon"),
                  {"units": unit, "long_name": longname, "_FillValue": NA_FLT4S, "missing_value": NA_FLT4S})],
                {"title": "SPLASH v2.0 " + longname})
            for name, val in (("lon", lon), ("lat", lat), ("time", zval)):
                o, shp, dt = lay[name]
                m = np.memmap(path, dtype=dt, mode="r+", offset=o, shape=shp)
                m[:] = val
                m.flush()
                del m
            o, shp, dt = lay[varname]
            self._maps[key] = np.memmap(path, dtype=dt, mode="r+", offset=o, shape=shp)
            self.paths[key] = path
            # which (layer range, row) stripes have been stored: whatever is still unwritten at close() becomes the
            # fill value -- the file is created with all-zero data, and a layer or row that never arrives (a time
            # axis longer than the months computed, a run that ends early) must read back as NA, not as 0.0.
            # Kept per distinct layer range (a run writes row blocks of ALL layers, or of a few time chunks): one
            # row mask each, not a [layers x rows] table (30 years of daily layers x 4096 rows x 9 bricks)
            self._seen[key] = {}

    def write(self, key, row0, block, t0=0):
        """block [nl_b, nrows, ncol] -> layers t0..t0+nl_b, rows row0..row0+nrows (t0 > 0: a later time
        chunk of a series processed in chunks)"""
        m = self._maps.get(key)
        if m is None:
            return
        b = np.asarray(block)
        if (b.ndim != 3 or t0 < 0 or t0 + b.shape[0] > self.nl or b.shape[2] != self.shape[1] or row0 + b.shape[1] > self.shape[0]):
            raise ValueError(f"{key}: block {b.shape} at layer {t0}, row {row0} does not fit a brick of {(self.nl,) + self.shape}")
        m[t0:t0 + b.shape[0], row0:row0 + b.shape[1], :] = np.where(np.isnan(b), NA_FLT4S, b)
        rows = self._seen[key].setdefault((int(t0), int(t0 + b.shape[0])), np.zeros(self.shape[0], bool))
        rows[row0:row0 + b.shape[1]] = True

    def close(self):
        """flush and unmap; every (layer, row) stripe that was never written is set to the fill value first"""
        for key, m in self._maps.items():
            seen = self._seen[key]
            cuts = sorted({0, self.nl} | {t for rng in seen for t in rng})
            for a, b in zip(cuts[:-1], cuts[1:]):     # elementary layer intervals: rows some range covering it has stored
                rows = np.zeros(self.shape[0], bool)
                for (t0, t1), r in seen.items():
                    if t0 <= a and b <= t1:
                        rows |= r
                if not rows.all():
                    m[a:b, ~rows, :] = NA_FLT4S
            m.flush()
        self._maps, self._seen = {}, {}

    def discard(self):
        """a run that failed midway: unmap and REMOVE the partial bricks (a reader must not find files whose
        missing blocks look like data)"""
        self._maps, self._seen = {}, {}
        for p in self.paths.values():
            try:
                os.remove(p)
            except OSError:
                pass
        self.paths = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if exc and exc[0] is not None:
            self.discard()
        else:
            self.close()
        return False
