"""splash.grid() end to end, with the reference's signature (R/splash.grid.R:28):

    splash_grid(sw_in, tc, pn, elev, soil, outdir=os.getcwd(), tmpdir=None,
                sim_control=dict(monthly_out=True, inmem=False))

What the reference does around its per-cell calls, and where each part lives here:

  R/splash.grid.R:40       compareRaster of the five inputs        -> _compare()
  :76-110                  resolution, upslope area (Au), latitude,
                           slope/aspect with NA slopes of valid
                           cells as flats                           -> api.static_rasters (HIP: §8 f-1, f-2)
  :120-181                 time axis of the inputs and the outputs  -> io.output_time_axis
  :186-216                 nine empty bricks on disk                -> io.BrickWriter (NetCDF, streamed)
  :260-275                 row blocks (blockSize)                   -> row_blocks()
  :277-308, 315-400        clFun on the cluster's nodes, the
                           receive loop that writes a block and
                           hands the free node the next one        -> one splash_grid_submit per block, `inflight`
                                                                       of them in the library at a time; a block
                                                                       is written while the next ones compute
  :409-449                 writeStop, the returned list             -> paths (inmem=False) or arrays (inmem=True)

This is the Python twin of r/splash_hip.R::splash.grid() (which cannot be run in this image: no R);
the GPU tests run THIS loop and compare it with one call over the whole grid, bit for bit.

Rasters: a `Raster` carries what splash.grid reads off a RasterLayer/RasterBrick -- values
[H, W] or [nl, H, W] (anything that can be sliced by rows: ndarray, numpy.memmap, a variable of
scipy.io.netcdf_file opened with mmap=True), the extent's ymax/xmin, the resolution and, for the
forcing bricks, the layer dates (getZ).  A lon/lat grid is assumed, as raster::area() and the
reference's latitude raster do.
"""
import os
from dataclasses import dataclass
from typing import Any, Optional, Sequence

import numpy as np

from . import api, io as _io


@dataclass
class Raster:
    values: Any                      # [H, W] or [nl, H, W], row 0 = northernmost
    ymax: float                      # extent: northern edge (degrees)
    xmin: float                      # extent: western edge
    yres: float                      # cell height (degrees, > 0)
    xres: float                      # cell width
    z: Optional[Sequence] = None     # layer dates (getZ) of a forcing brick

    @property
    def nrow(self):
        return int(np.shape(self.values)[-2])

    @property
    def ncol(self):
        return int(np.shape(self.values)[-1])

    def like(self, values, z=None):
        return Raster(values, self.ymax, self.xmin, self.yres, self.xres, z)

    def lat(self):
        return self.ymax - self.yres * (np.arange(self.nrow) + 0.5)

    def lon(self):
        return self.xmin + self.xres * (np.arange(self.ncol) + 0.5)

    def rows(self, r0, r1, t0=None, t1=None):
        """getValues(x, row, nrows): rows [r0, r1) (of layers [t0, t1)) as a C-contiguous float array"""
        v = self.values
        a = v[r0:r1] if np.ndim(v) == 2 else v[t0:t1, r0:r1]
        a = np.asarray(a)
        if a.dtype not in (np.float32, np.float64) or not a.dtype.isnative:
            a = a.astype(np.float32 if a.dtype.itemsize == 4 and a.dtype.kind == "f" else np.float64)
        return np.ascontiguousarray(a)


def _compare(rasters, tol=0.5):
    """compareRaster(..., extent=TRUE, res=TRUE, rotation=FALSE, values=FALSE) with
    rasterOptions(tolerance = 0.5) (R/splash.grid.R:39-40): same rows/columns, and extent and resolution
    equal to within `tol` cells.  The reference only warns (stopiffalse=FALSE) and then fails somewhere
    inside a worker; here a mismatch is an error."""
    a = rasters[0]
    for b in rasters[1:]:
        if (a.nrow, a.ncol) != (b.nrow, b.ncol):
            raise ValueError(f"rasters differ in rows/columns: {(a.nrow, a.ncol)} vs {(b.nrow, b.ncol)}")
        if abs(a.ymax - b.ymax) > tol * a.yres or abs(a.xmin - b.xmin) > tol * a.xres:
            raise ValueError("rasters differ in extent")
        if abs(a.yres - b.yres) > tol * a.yres or abs(a.xres - b.xres) > tol * a.xres:
            raise ValueError("rasters differ in resolution")


def row_blocks(nrow, ncol, bytes_per_cell, target_bytes=2 << 30, minblocks=2, block_rows=None):
    """raster::blockSize: [(row0, nrows)] covering the grid once, whole rows, at least `minblocks`
    blocks (the reference asks for 2 per node, 10 on one node, R/splash.grid.R:266-270) of at most
    `target_bytes` of values each."""
    if block_rows is None:
        by_mem = max(1, int(target_bytes // max(1, bytes_per_cell * ncol)))
        by_count = max(1, -(-nrow // max(1, minblocks)))
        block_rows = min(by_mem, by_count)
    block_rows = max(1, min(int(block_rows), nrow))
    return [(r, min(block_rows, nrow - r)) for r in range(0, nrow, block_rows)]


def grid_statics(elev, degree=1.0, method=None, terrain="reference"):
    """R/splash.grid.R:76-110 from the elevation raster alone: latitude, slope/aspect (NA slopes of
    valid cells are flats, :107), resolution = sqrt(cell area) and the Au stack (contributing area,
    cells draining in, cells draining out; R/upslope_area.R:56).  Like the reference, large or coarse
    grids (ncell > 1e7, or 0.1 < resolution < 1 degree) take the D8 area of `upslope_areav2`, the others
    the multiple-flow-direction area of `upslope_area` (:76-77); `method` ("tau" / "top") overrides.
    terrain: "reference" (default) = whichever of the two the reference takes for this grid's size: "raster" =
    raster::terrain's slope and aspect, the small-grid branch's (:104-107: spacing from the row's latitude, NA ring set
    to 0); "gdal" = the large-grid branch's (:83-84: `gdaldem slope -s 111120 -compute_edges` / `gdaldem aspect
    -zero_for_flat -compute_edges` on the DEM rounded to whole metres: one scale at every latitude, edges computed, flat
    aspect 0 -- api.terrain_gdal, restated from gdaldem's published algorithm).  ("raster" on a large grid is the
    better terrain -- the scale of "gdal" understates east-west gradients by cos(latitude) -- but not the reference's.)"""
    e = np.ascontiguousarray(np.asarray(elev.values, dtype=np.float64))
    large = bool(e.size > 1e7 or 0.1 < elev.xres < 1)
    if method is None:
        method = "tau" if large else "top"
    if terrain not in ("raster", "gdal", "reference"):
        raise ValueError("terrain must be 'raster', 'gdal' or 'reference'")
    if terrain == "reference":
        terrain = "gdal" if large else "raster"
    area_key = "ups_area" if method == "tau" else "ups_area_mfd"
    graph = (area_key, "ncellin", "ncellout")
    if method == "tau" and not np.array_equal(np.rint(e), e, equal_nan=True):
        # upslope_areav2 hands TauDEM the DEM as 16-bit integers (R/upslope_area.R:64): pits, flow directions and the
        # contributing area come from elevations rounded to whole metres; latitude, resolution and raster::terrain do not
        s = api.static_rasters(e, elev.ymax, elev.yres, elev.xres, na_to_zero=True, degree=degree,
                               want=("slope", "aspect", "lat", "resolution"))
        s.update(api.static_rasters(np.rint(e), elev.ymax, elev.yres, elev.xres, na_to_zero=True, degree=degree, want=graph))
    else:
        s = api.static_rasters(e, elev.ymax, elev.yres, elev.xres, na_to_zero=True, degree=degree,
                               want=("slope", "aspect", "lat", "resolution") + graph)
    if terrain == "gdal":
        s["slope"], s["aspect"] = api.terrain_gdal(e, elev.xres, elev.yres, 111120.0, round_int=True)
    return dict(lat=s["lat"], slop=s["slope"], asp=s["aspect"], resolution=s["resolution"],
                Au=np.stack([s[area_key], s["ncellin"], s["ncellout"]]), method=method, terrain=terrain)


def year_chunks(ztime, chunk_years):
    """[(t0, t1)] covering the series once, each chunk `chunk_years` whole calendar years (the last one
    what is left): a chunk starts on a 1 January, so the monthly outputs of the chunks are the months of
    the undivided run"""
    y = np.asarray(ztime, dtype="datetime64[Y]").astype(int)
    starts = np.flatnonzero(np.r_[True, y[1:] != y[:-1]])
    cuts = list(starts[::max(1, int(chunk_years))]) + [len(y)]
    return [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:])]


def splash_grid(sw_in, tc, pn, elev, soil, outdir=None, tmpdir=None, sim_control=None, *,
                devices=None, inflight=2, block_rows=None, statics=None, out_dtype=np.float64,
                want_diag=False, basin=None, n_basin=0, progress=None, chunk_years=None, backend="hip"):
    """R/splash.grid.R:28.  sw_in, tc, pn: Raster bricks [nl, H, W] with layer dates (`z`, taken from
    pn like getZ(pn), :144); elev: Raster [H, W]; soil: Raster [6, H, W] (sand, clay, OM, gravel, bulk
    density, depth).  sim_control = dict(monthly_out=True, inmem=False) as the reference's default.
    Returns the reference's list (:449) as a dict wn, ro, pet, aet, snow, bflow, cond, netrad, sm_lim:
    paths of the NetCDF bricks under `outdir` (inmem=False) or [nl_out, H, W] arrays (inmem=True), plus
    "statics" (what :76-110 derived) and, with want_diag, the exact water-balance totals summed over the
    blocks ("diag", "diag_acc": integer limbs, so the sum over blocks IS the one-call total).
    `tmpdir` is accepted for the signature's sake (the reference parks intermediate rasters there; nothing
    is parked here).  devices: HIP ordinals every block is split over inside the library; inflight:
    blocks in the library at a time; block_rows: rows per block (default: ~2 GB of forcing per block, at
    least two blocks).
    chunk_years: process the series in time chunks of that many calendar years (a 30-year series of a
    large grid does not fit host memory even one row block at a time): per row block a pre-pass takes the
    snowfall threshold of the whole series (the maximum of the chunks'), the first chunk runs with the
    spin-up, every later one continues from the state the one before ended in (splash_grid_in.state0 /
    ai0 / tt, ABI 5) while the results of the one before are being written -- bit for bit the undivided
    run.
    backend="host" (ABI 6): every block on CPU threads, the library's own day step compiled for the host -- for grids
    of a few thousand cells, and for running THIS loop (blocks in flight, chunks, bricks) where there is no GPU.  The
    static rasters are device kernels: pass `statics` (lat, slop, asp, resolution, Au) then.  Never chosen for you."""
    ctl = dict(monthly_out=True, inmem=False)
    ctl.update(sim_control or {})
    monthly, inmem = bool(ctl["monthly_out"]), bool(ctl["inmem"])
    _compare([sw_in, tc, pn, elev, soil])
    if pn.z is None:
        raise ValueError("pn carries no layer dates (Raster.z): getZ(pn), R/splash.grid.R:144")
    ztime = np.asarray(pn.z, dtype="datetime64[D]")
    nt = int(np.shape(pn.values)[0])
    if ztime.size != nt or np.shape(sw_in.values)[0] != nt or np.shape(tc.values)[0] != nt:
        raise ValueError("the forcing bricks and the layer dates differ in length")
    if np.shape(soil.values)[0] != 6:
        raise ValueError("soil needs six layers: sand, clay, OM, gravel, bulk density, depth")
    H, W = elev.nrow, elev.ncol
    if statics is None and backend != "hip":
        raise ValueError("backend='host': the static rasters (terrain, upslope area) are device kernels -- pass statics=dict("
                         "lat, slop, asp, resolution, Au)")
    st = statics or grid_statics(elev)
    z_out, _, _, y0, y1 = _io.output_time_axis(ztime, monthly)
    nl = z_out.size
    if not monthly and nl != nt:
        raise ValueError("daily output needs whole calendar years of input (R/splash.grid.R:161-181)")
    nm_in = int(api.month_ids(*api.time_fields(ztime)[1:])[-1]) + 1
    if monthly and nm_in != nl:   # (write_bricks raises on the same mismatch; a streamed brick would keep layers nobody writes)
        raise ValueError(f"monthly output: the input covers {nm_in} calendar months, the output time axis of whole "
                         f"years {y0}-{y1} has {nl} layers (R/splash.grid.R:161-181)")
    keys = api.OUT_NAMES
    writer = None
    mats = None
    if inmem:
        mats = {k: np.empty((nl, H, W), out_dtype) for k in keys}
    else:
        outdir = outdir or os.getcwd()
        writer = _io.BrickWriter(outdir, keys, elev.lon(), elev.lat(), ztime, monthly)
    itemsize = 4 if all(np.asarray(x.values[:1, :1, :1]).dtype == np.float32 for x in (sw_in, tc, pn)) else 8
    bs = row_blocks(H, W, 3 * nt * itemsize, minblocks=2, block_rows=block_rows)
    inflight = max(1, min(int(inflight), len(bs)))
    acc_total = None
    if basin is not None and not n_basin:   # one numbering for every block (a block sees only its own ids)
        n_basin = int(np.nanmax(np.asarray(basin))) + 1

    def submit(i):
        r0, nr = bs[i]
        r1 = r0 + nr
        cut = lambda a: np.ascontiguousarray(np.asarray(a)[..., r0:r1, :])
        return api.splash_grid(
            sw_in.rows(r0, r1), tc.rows(r0, r1), pn.rows(r0, r1), elev.rows(r0, r1), soil.rows(r0, r1),
            cut(st["lat"]), cut(st["slop"]), cut(st["asp"]), cut(st["Au"]), cut(st["resolution"]),
            time_index=ztime, sim_control=dict(monthly_out=monthly), devices=devices, submit=True,
            mem_fraction=0.8 / inflight, want_state=False, out_dtype=out_dtype, want_diag=want_diag,
            basin=None if basin is None else cut(np.asarray(basin).reshape(H, W)), n_basin=n_basin, backend=backend)

    def store(res, r0, nr, t0_out, m0):
        nonlocal acc_total
        for k in keys:
            if inmem:
                mats[k][t0_out:t0_out + res[k].shape[0], r0:r0 + nr, :] = res[k]
            else:
                writer.write(k, r0, res[k], t0_out)
        if want_diag:
            a = res["diag_acc"]
            if acc_total is None:
                acc_total = np.zeros((a.shape[0], nm_total, a.shape[2]), np.int64)
            acc_total[:, m0:m0 + a.shape[1]] += a

    nm_total = int(api.month_ids(*api.time_fields(ztime)[1:])[-1]) + 1
    if chunk_years:
        # time chunks: per row block, threshold pre-pass, then the chunks in order, each continuing the one before
        tcs = year_chunks(ztime, chunk_years)
        mid_all = api.month_ids(*api.time_fields(ztime)[1:])
        jobs = []
        try:
            for i, (r0, nr) in enumerate(bs):
                r1 = r0 + nr
                cut = lambda a: np.ascontiguousarray(np.asarray(a)[..., r0:r1, :])
                st_b = {k: cut(st[k]) for k in ("lat", "slop", "asp", "Au", "resolution")}
                ev, so = elev.rows(r0, r1), soil.rows(r0, r1)
                tt = None
                for t0, t1 in tcs:   # Tt = max(tc[p_snow >= 0.5]) over the WHOLE series (R/splash.point.R:120-122)
                    tk = api.snow_threshold(tc.rows(r0, r1, t0, t1), st_b["lat"], ev, backend=backend)
                    tt = tk if tt is None else np.maximum(tt, tk)
                bas = None if basin is None else cut(np.asarray(basin).reshape(H, W))
                state = ai = None
                prev = None   # (job, t0_out, m0) of the chunk in the library
                for t0, t1 in tcs:
                    if prev is not None:   # the chunk before must have ended: its state starts this one
                        res_p = prev[0].wait()
                        jobs.remove(prev[0])
                        state = res_p["state_end"]
                        ai = res_p["aux"]["AI"] if ai is None else ai
                    job = api.splash_grid(
                        sw_in.rows(r0, r1, t0, t1), tc.rows(r0, r1, t0, t1), pn.rows(r0, r1, t0, t1), ev, so,
                        st_b["lat"], st_b["slop"], st_b["asp"], st_b["Au"], st_b["resolution"],
                        time_index=ztime[t0:t1], sim_control=dict(monthly_out=monthly), devices=devices, submit=True,
                        mem_fraction=0.8, want_state=True, want_aux=True, out_dtype=out_dtype, want_diag=want_diag,
                        basin=bas, n_basin=n_basin, tt=tt, state0=state, ai0=ai, backend=backend)
                    jobs.append(job)
                    if prev is not None:   # written while this chunk computes
                        store(res_p, r0, nr, prev[1], prev[2])
                    prev = (job, int(mid_all[t0]) if monthly else t0, int(mid_all[t0]))
                res_p = prev[0].wait()
                jobs.remove(prev[0])
                store(res_p, r0, nr, prev[1], prev[2])
                if progress:
                    progress(i + 1, len(bs))
        except BaseException:
            for j in jobs:
                j.close()
            jobs = []
            if writer:
                writer.discard()   # no partial bricks whose missing blocks would read as data
                writer = None
            raise
        finally:
            for j in jobs:
                j.close()
            if writer:
                writer.close()
        return _returned(inmem, mats, writer, st, bs, want_diag, acc_total, basin)

    # the receive loop of R/splash.grid.R:359-400: collect block i, hand the free slot block i + inflight,
    # write block i while the others compute
    jobs = [None] * len(bs)
    try:
        for i in range(inflight):
            jobs[i] = submit(i)
        for i, (r0, nr) in enumerate(bs):
            res = jobs[i].wait()
            jobs[i] = None
            if i + inflight < len(bs):
                jobs[i + inflight] = submit(i + inflight)
            store(res, r0, nr, 0, 0)
            if progress:
                progress(i + 1, len(bs))
    except BaseException:
        for j in jobs:          # an error mid-way: wait for what is still in the library before the
            if j is not None:   # arrays it reads and writes go away
                j.close()
        jobs = []
        if writer:
            writer.discard()    # ... and leave no partial bricks behind
            writer = None
        raise
    finally:
        for j in jobs:
            if j is not None:
                j.close()
        if writer:
            writer.close()
    return _returned(inmem, mats, writer, st, bs, want_diag, acc_total, basin)


def _returned(inmem, mats, writer, st, bs, want_diag, acc_total, basin):
    ret_names = dict(wn="wn", ro="ro", pet="pet", aet="aet", snow="snow", bflow="bflow", cond="cond",
                     netr="netrad", sm_lim="sm_lim")     # R/splash.grid.R:449
    src = mats if inmem else writer.paths
    out = {ret_names[k]: src[k] for k in ("wn", "ro", "pet", "aet", "snow", "bflow", "cond", "netr", "sm_lim")}
    out["statics"] = st
    out["blocks"] = bs
    if want_diag:
        out["diag_acc"] = acc_total
        d, _ = api.diag_finalize_host(acc_total)
        out["diag"] = {k: (d[:, :, j] if basin is not None else d[0, :, j]) for j, k in enumerate(api.DIAG_NAMES)}
    return out


def unSWC_grid(soil_data, uns_depth, wn, d_name, outdir=None, backend="hip"):
    """unSWC.grid(soil_data, uns_depth, wn, d_name, outdir = getwd()) -- R/unsSWC.grid.R:14-141 with its signature.

    soil_data: Raster of six layers (sand, clay, OM, coarse fragments, bulk density, depth to bedrock), wn: Raster
    brick of the column water content with its layer dates in `z` (the `wn` brick splash.grid returns), d_name: the
    depth as it should appear in the file names.  Writes the reference's four files into outdir (io.UNSWC_BRICKS;
    outdir=None: nothing is written) and returns what it returns, list(w_z, wtd, Se), as Rasters -- plus theta_mean,
    the raster its first file holds.  The arithmetic is the library's (splash_unswc_grid_run: the gridded function's
    own semantics, include/splash_hip.h), in column blocks sized to the device."""
    _compare([soil_data, wn])
    if wn.z is None:
        raise ValueError("unSWC.grid: wn needs its layer dates (getZ)")
    H, W = wn.nrow, wn.ncol
    soil = np.ascontiguousarray(np.asarray(soil_data.values, dtype=np.float64).reshape(6, H * W))
    w = np.asarray(wn.values)
    nl = w.shape[0]
    w = np.ascontiguousarray(w.reshape(nl, H * W), dtype=np.float64)
    r = api.unSWC_grid(soil, uns_depth, w, backend=backend)
    res = {k: v.reshape(nl, H, W) for k, v in r.items()}
    out = {k: wn.like(v, z=wn.z) for k, v in res.items()}
    if outdir is not None:
        os.makedirs(outdir, exist_ok=True)
        out["paths"] = _io.write_unswc_bricks(outdir, res, wn.lon(), wn.lat(), wn.z, d_name)
    return out

