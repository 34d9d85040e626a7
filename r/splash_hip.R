# splash_hip.R -- R wrappers with the reference's signatures (R/splash.point.R:29,
# R/splash.grid.R:28) that route the hot path through the HIP library instead of the
# per-cell Rcpp module.  Everything outside the hot path (terrain/upslope preparation,
# raster and NetCDF I/O, time axis handling) stays exactly as in the reference.

.time_fields <- function(time_index) {
  list(doy   = as.integer(format(time_index, "%j")),   # R/splash.point.R:57
       year  = as.integer(format(time_index, "%Y")),   # :59
       month = as.integer(format(time_index, "%m")))   # :547
}

# backend = "hip" (default) or "host": a single site is a chain of sequential day steps that one CPU core steps
# through faster than a lone wavefront; "host" runs the library's own day step compiled for the host
# (include/splash_hip.h, ABI 6).  Never chosen for the caller: without a HIP device the default is an error.
splash.point <- function(sw_in, tc, pn, lat, elev, slop = 0, asp = 0, soil_data, Au = 0,
                         resolution = 250, time_index = NULL, monthly_out = FALSE,
                         ts_out = TRUE, verbose = TRUE, backend = c("hip", "host")) {
  backend <- match(match.arg(backend), c("hip", "host")) - 1L
  if (class(tc)[1] %in% c("xts", "zoo", "ts")) time_index <- time(tc)
  if (is.null(time_index)) stop("time info missing")
  time.freq <- abs(as.numeric(time_index[1] - time_index[2], units = "days"))
  if (time.freq > 20) stop("monthly inputs: disaggregate to daily first (R/splash.point.R:66-86)")
  tf <- .time_fields(time_index)
  result <- .Call("C_splash_point_run", as.numeric(sw_in), as.numeric(tc), as.numeric(pn),
                  lat, elev, slop, asp, as.numeric(soil_data), as.numeric(Au), resolution,
                  tf$doy, tf$year, tf$month, monthly_out, backend)
  if (ts_out & !monthly_out) {
    result <- xts::xts(base::as.data.frame(result), time_index)
  } else if (ts_out & monthly_out) {
    result <- xts::xts(base::as.data.frame(result),
                       seq(time_index[1], time_index[length(time_index)], "month"))
  }
  result
}

# Replaces clFun + the send/receive loop of R/splash.grid.R:277-400: one .Call per row block.
# `bs` is raster::blockSize(sw_in, ...) as in the reference; the static rasters (lat, terraines,
# resolution, Au) are prepared exactly as R/splash.grid.R:76-110 does.  `devices`: HIP ordinals the
# block is split over inside the call (this is what stands where the SOCK cluster stood); `devices = -1L`:
# no device -- the host backend (ABI 6), for blocks of a few cells.
# .splash_grid_submit starts the block's call and returns a handle at once; .splash_grid_collect
# waits for it and returns
#   list(wn, ro, pet, aet, snow, cond, bflow, netr, sm_lim) of [ncell_blk x nl] matrices, which
#   writeValues(brick, m, bs$row[i]) / matXX[startind[i]:endind[i], ] <- m consume unchanged
#   (R/splash.grid.R:370-392)
.splash_grid_submit <- function(i, bs, sw_in, tc, pn, elev, lat, terraines, soil, resolution, Au,
                                ztime, monthly_out, devices = 0L, mem_fraction = 0) {
  gv <- function(x) raster::getValues(x, bs$row[i], bs$nrows[i])
  tf <- .time_fields(ztime)
  stat <- cbind(gv(lat), gv(elev), gv(terraines[[1]]), gv(terraines[[2]]), gv(resolution))
  .Call("C_splash_grid_submit", gv(sw_in), gv(tc), gv(pn), stat, gv(soil), gv(Au),
        tf$doy, tf$year, tf$month, monthly_out, as.integer(devices), as.numeric(mem_fraction))
}
.splash_grid_collect <- function(handle) .Call("C_splash_grid_wait", handle)

# A series too long for host memory, in time chunks (ABI 5).  The module's run_all starts from the state it is
# given (src/SPLASH.h:93-97); .splash_grid_chunk is that for a row block: `tt` = the snowfall threshold of the
# WHOLE series, do.call(pmax, lapply(chunks, function(ch) .splash_snow_threshold(tc_of(ch), lat, elev)))
# (R/splash.point.R:120-122: max over all days); the first chunk runs the spin-up (state0 = ai0 = NULL), every
# later one continues from res$state_end of the chunk before with ai0 = aux[, 2] of the FIRST chunk.  Chunks of
# whole calendar years give the monthly outputs of the undivided run, bit for bit (tested on the Python twin,
# rsplash_amd/grid.py: chunk_years).
.splash_snow_threshold <- function(tc, lat, elev, backend = 0L)      # backend = 1L with devices = -1L: the host backend
  .Call("C_splash_snow_threshold", tc, as.numeric(lat), as.numeric(elev), as.integer(backend))
.splash_grid_chunk <- function(sw_in, tc, pn, stat, soil, Au, ztime, monthly_out, tt, state0 = NULL, ai0 = NULL,
                               devices = 0L) {
  tf <- .time_fields(ztime)
  .Call("C_splash_grid_chunk", sw_in, tc, pn, stat, soil, Au, tf$doy, tf$year, tf$month, monthly_out,
        as.integer(devices), as.numeric(tt), state0, ai0)
}

# file stem, NetCDF variable, unit and long name of the nine result bricks, in the order
# .splash_grid_collect returns them (names as R/splash.grid.R:190-216 and :449)
.splash_outputs <- data.frame(
  key  = c("wn",  "ro", "pet", "aet", "snow", "cond", "bflow", "netr",   "sm_lim"),
  stem = c("sm", "ro", "pet", "aet", "swe",  "cond", "bflow", "netrad", "sm_lim"),   # writeStart()'s names, R/splash.grid.R:321-337
  var  = c("wm",  "ro", "pet", "aet", "swe",  "cond", "bflow", "netrad", "sm_lim"),
  unit = c("mm",  "mm", "mm",  "mm",  "mm",   "mm",   "mm",    "MJ",     "mm"),
  long = c("Soil water content", "surface runoff", "potential evapotranspiration",
           "actual evapotranspiration", "snow water equivalent", "condensation", "drainage",
           "daytime net radiation", "fraction of available water content"),
  # name of the same brick in the list splash.grid returns (R/splash.grid.R:449) -- POSITIONAL with
  # `key`: the returned list is re-ordered by name afterwards, never relabelled
  ret  = c("wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netrad", "sm_lim"),
  stringsAsFactors = FALSE)

# splash.grid with the reference's signature (R/splash.grid.R:28).  Static preparation (upslope
# area, latitude, slope/aspect, resolution: R/splash.grid.R:76-110) and the output time axis
# (:145-181) are the package's own code and are reused through `prep`; what changes is the middle:
# no cluster, no per-cell mapply -- every row block goes through ONE call of the HIP library, which
# splits it over `devices` itself; `inflight` of them are in the library at a time.  Blocks are
# sized by raster::blockSize as before, so memory use on the R side is `inflight` blocks.
splash.grid <- function(sw_in, tc, pn, elev, soil, outdir = getwd(),
                        tmpdir = dirname(raster::rasterTmpFile()),
                        sim.control = list(monthly_out = TRUE, inmem = FALSE), devices = 0L,
                        inflight = 2L, prep = splash.grid.statics(elev, tmpdir)) {
  raster::rasterOptions(tolerance = 0.5)
  raster::compareRaster(sw_in, tc, pn, elev, soil, extent = TRUE, crs = TRUE, res = TRUE,
                        orig = FALSE, rotation = FALSE, values = FALSE, stopiffalse = FALSE,
                        showwarning = TRUE)
  ztime <- raster::getZ(pn)
  if (!methods::is(ztime[1], "Date")) stop("layer dates (setZ) of class Date are required")
  yrs <- as.numeric(unique(format(ztime, "%Y")))
  first <- as.Date(paste(yrs[1], 1, 1, sep = "-"))
  last <- as.Date(paste(yrs[length(yrs)], 12, 31, sep = "-"))
  monthly <- isTRUE(sim.control$monthly_out)
  ztime_out <- seq(first, last, by = if (monthly) "month" else "day")   # :161-181
  nl <- length(ztime_out)
  if (!monthly && nl != length(ztime)) stop("daily output needs whole calendar years of input")
  text <- if (monthly) "monthly" else "daily"
  inmem <- isTRUE(sim.control$inmem)
  bricks <- lapply(seq_len(nrow(.splash_outputs)), function(k) {
    o <- .splash_outputs[k, ]
    if (inmem) return(NULL)
    b <- raster::brick(elev, values = FALSE, nl = nl, format = "CDF", overwrite = TRUE,
                       filename = file.path(outdir, paste0("SPLASH_v2.0_", o$stem, "_", text, "_",
                                                           yrs[1], "-", yrs[length(yrs)], ".nc")),
                       varname = o$var, varunit = o$unit, longname = o$long, xname = "lon",
                       yname = "lat", zname = "time")
    raster::writeStart(raster::setZ(b, ztime_out), filename = raster::filename(b), overwrite = TRUE)
  })
  mats <- if (inmem) lapply(1:9, function(k) matrix(NA_real_, raster::ncell(elev), nl)) else NULL
  bs <- raster::blockSize(sw_in, minblocks = 2)
  pb <- utils::txtProgressBar(min = 0, max = bs$n, style = 3)
  # The receive loop of R/splash.grid.R:359-400 with the GPU where the cluster was: `inflight` blocks
  # are in the library at any time (each on its own host thread and streams, with its share of the
  # device memory), so reading block i+1 from disk, the device work of block i and writing block i-1
  # overlap, and the long tails of the blocks' spin-ups run side by side instead of one after another.
  inflight <- max(1L, min(as.integer(inflight), bs$n))
  submit <- function(i) .splash_grid_submit(i, bs, sw_in, tc, pn, elev, prep$lat, prep$terraines, soil,
                                            prep$resolution, prep$Au, ztime, monthly, devices,
                                            mem_fraction = 0.8 / inflight)
  handles <- vector("list", bs$n)
  for (i in seq_len(inflight)) handles[[i]] <- submit(i)
  for (i in seq_len(bs$n)) {
    res <- .splash_grid_collect(handles[[i]])
    handles[i] <- list(NULL)
    if (i + inflight <= bs$n) handles[[i + inflight]] <- submit(i + inflight)   # :395-399: next block to the free slot
    for (k in 1:9) {
      if (inmem) {
        rows <- raster::cellFromRow(elev, bs$row[i]:(bs$row[i] + bs$nrows[i] - 1))
        mats[[k]][rows, ] <- res[[k]]
      } else {
        bricks[[k]] <- raster::writeValues(bricks[[k]], res[[k]], bs$row[i])
      }
    }
    utils::setTxtProgressBar(pb, i)
  }
  close(pb)
  out <- lapply(1:9, function(k) {
    if (inmem) {
      b <- raster::brick(nrows = nrow(elev), ncols = ncol(elev), crs = raster::crs(elev), nl = nl)
      raster::extent(b) <- raster::extent(elev)
      raster::setZ(raster::setValues(b, mats[[k]]), ztime_out)
    } else {
      raster::writeStop(bricks[[k]])
    }
  })
  names(out) <- .splash_outputs$ret
  out[c("wn", "ro", "pet", "aet", "snow", "bflow", "cond", "netrad", "sm_lim")]      # :449
}

# the static rasters of R/splash.grid.R:76-110 with the package's own helpers (upslope_area,
# raster::terrain, raster::area); NA slopes of valid cells are flats (:107)
splash.grid.statics <- function(elev, tmpdir = tempdir()) {
  resolution <- sqrt(raster::area(elev)) * 1000
  Au <- upslope_area(elev, resolution)
  lat <- raster::init(elev, "y")
  lat[is.na(elev)] <- NA
  terraines <- raster::terrain(elev, opt = c("slope", "aspect"), unit = "degrees")
  terraines[is.na(terraines) & !is.na(elev)] <- 0
  list(lat = lat, terraines = terraines, resolution = resolution, Au = Au)
}

# The same rasters from the HIP library (Horn slope/aspect, latitude, resolution, sink fill, D8
# contributing area, cells draining in/out: include/splash_hip.h, splash_static_rasters_run); a
# lon/lat grid is assumed, as raster::area() and the rest of splash.grid do.
# terrain = "gdal": slope and aspect as the reference's large-grid branch gets them (R/splash.grid.R:83-84: gdaldem on
# the INT2S DEM; published algorithm restated in the library), "raster": raster::terrain's; "reference" (default):
# whichever the reference takes for the grid's size (:77).
splash.grid.statics.hip <- function(elev, degree = 1, terrain = c("reference", "raster", "gdal")) {
  terrain <- match.arg(terrain)
  large <- raster::ncell(elev) > 1e7 | (raster::res(elev)[1] > 0.1 & raster::res(elev)[1] < 1)     # R/splash.grid.R:77
  if (terrain == "reference") terrain <- if (large) "gdal" else "raster"
  z <- matrix(raster::getValues(elev), nrow = ncol(elev), ncol = nrow(elev))   # [ncol x nrow]: row-major cells
  r <- .Call("C_splash_static_rasters", z, as.numeric(raster::ymax(elev)), as.numeric(raster::yres(elev)),
             as.numeric(raster::xres(elev)), as.numeric(degree))
  if (large) {   # upslope_areav2: D8 area (TauDEM's aread8) of the DEM as 16-bit integers (R/upslope_area.R:64-96)
    ri <- .Call("C_splash_static_rasters", round(z), as.numeric(raster::ymax(elev)), as.numeric(raster::yres(elev)),
                as.numeric(raster::xres(elev)), as.numeric(degree))
    r$areacatch <- ri$areacatch_d8
    r$ncellin <- ri$ncellin
    r$ncellout <- ri$ncellout
  }
  if (terrain == "gdal") {
    g <- .Call("C_splash_terrain_gdal", z, as.numeric(raster::xres(elev)), as.numeric(raster::yres(elev)), 111120, 1L)
    r$slope <- g$slope
    r$aspect <- g$aspect
  }
  as_raster <- function(m) raster::setValues(raster::raster(elev), as.vector(m))
  terraines <- raster::stack(as_raster(r$slope), as_raster(r$aspect))
  names(terraines) <- c("slope", "aspect")
  Au <- raster::stack(as_raster(r$areacatch), as_raster(r$ncellin), as_raster(r$ncellout))   # R/upslope_area.R:56
  list(lat = as_raster(r$lat), terraines = terraines, resolution = as_raster(r$resolution), Au = Au)
}

# upslope_area (R/upslope_area.R:12) from the HIP library: the same RasterStack(areacatch, ncellin, ncellout).  The
# reference hands sinkfill()/topidx() the MEAN of its resolution raster (:14); the library forms that mean itself
# from the cell areas, so the argument is accepted for the signature's sake only.
upslope_area.hip <- function(dem, resolution = NULL) splash.grid.statics.hip(dem)$Au

# unSWC.grid (R/unsSWC.grid.R:14): same signature, same four files (names, variable names, units, long names of
# :87, :108, :115, :126), same returned list(w_z, wtd, Se); the arithmetic -- the gridded function's own semantics,
# include/splash_hip.h -- runs in the HIP library over the brick's values ([ncell x nt], as getValues() gives them).
unSWC.grid <- function(soil_data, uns_depth, wn, d_name, outdir = getwd(), backend = c("hip", "host")) {
  backend <- match.arg(backend)
  ztime <- raster::getZ(wn)
  time.freq <- ztime[2] - ztime[1]
  time_text <- if (abs(as.numeric(time.freq, units = "days")) < 2) "daily" else "monthly"
  y <- as.integer(unique(format(ztime, "%Y")))
  res <- .Call("C_splash_unswc_grid_run", raster::getValues(wn), raster::getValues(soil_data), as.numeric(uns_depth),
               if (backend == "host") 1L else 0L)
  put <- function(m, tag, varname, varunit, longname) {
    b <- raster::setValues(raster::brick(wn), m)
    raster::writeRaster(b, paste0(outdir, "/", "SPLASH_v2.0_", tag, "_", time_text, "_", y[1], "-", y[length(y)], ".nc"),
                        format = "CDF", overwrite = TRUE, varname = varname, varunit = varunit, longname = longname,
                        xname = "lon", yname = "lat", zname = "time")
  }
  theta_i <- put(res$theta_mean, "theta_mean", "theta", "m3/m3", "volumetric soil moisture")
  wtd <- put(res$wtd, "wtd", "wtd", "m", "water table depth")
  w_z <- put(res$w_z, paste0("swc_top_", d_name, "_m"), "swc", "mm", "soil water content")
  Se <- put(res$Se, paste0("Se_top_", d_name, "_m"), "Se", "fraction", "effective saturation or water filled porosity")
  list(w_z = w_z, wtd = wtd, Se = Se)
}

# unSWC (R/unsSWC.R:14): same signature; the analytical integral runs in the HIP library.
# One site = one "cell": wn is handed over as a [1 x nt] matrix, soil as [1 x 6].
unSWC <- function(soil_data, uns_depth, wn, backend = c("hip", "host")) {
  backend <- match.arg(backend)          # "host" (ABI 6): on a CPU core, as splash.point(..., backend = "host")
  res <- .Call("C_splash_unswc_run", matrix(as.numeric(wn), nrow = 1),
               matrix(as.numeric(soil_data), nrow = 1), as.numeric(uns_depth), if (backend == "host") 1L else 0L)
  result <- list(wn_z = res$wn_z[1, ], wtd = res$wtd[1, ])
  if (xts::is.xts(wn)) result <- mapply(FUN = xts::xts, result, MoreArgs = list(order.by = time(wn)), SIMPLIFY = FALSE)
  result$theta_i <- if (xts::is.xts(wn)) xts::xts(res$theta_i[1, ], time(wn)) else res$theta_i[1, ]
  result
}
