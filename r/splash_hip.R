# splash_hip.R -- R wrappers with the reference's signatures (R/splash.point.R:29,
# R/splash.grid.R:28) that route the hot path through the HIP library instead of the
# per-cell Rcpp module.  Everything outside the hot path (terrain/upslope preparation,
# raster and NetCDF I/O, time axis handling) stays exactly as in the reference.

.time_fields <- function(time_index) {
  list(doy   = as.integer(format(time_index, "%j")),   # R/splash.point.R:57
       year  = as.integer(format(time_index, "%Y")),   # :59
       month = as.integer(format(time_index, "%m")))   # :547
}

splash.point <- function(sw_in, tc, pn, lat, elev, slop = 0, asp = 0, soil_data, Au = 0,
                         resolution = 250, time_index = NULL, monthly_out = FALSE,
                         ts_out = TRUE, verbose = TRUE) {
  if (class(tc)[1] %in% c("xts", "zoo", "ts")) time_index <- time(tc)
  if (is.null(time_index)) stop("time info missing")
  time.freq <- abs(as.numeric(time_index[1] - time_index[2], units = "days"))
  if (time.freq > 20) stop("monthly inputs: disaggregate to daily first (R/splash.point.R:66-86)")
  tf <- .time_fields(time_index)
  result <- .Call("C_splash_point_run", as.numeric(sw_in), as.numeric(tc), as.numeric(pn),
                  lat, elev, slop, asp, as.numeric(soil_data), as.numeric(Au), resolution,
                  tf$doy, tf$year, tf$month, monthly_out)
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
# resolution, Au) are prepared exactly as R/splash.grid.R:76-110 does.
.splash_grid_block <- function(i, bs, sw_in, tc, pn, elev, lat, terraines, soil, resolution, Au,
                               ztime, monthly_out, device = 0L) {
  gv <- function(x) raster::getValues(x, bs$row[i], bs$nrows[i])
  tf <- .time_fields(ztime)
  stat <- cbind(gv(lat), gv(elev), gv(terraines[[1]]), gv(terraines[[2]]), gv(resolution))
  .Call("C_splash_grid_run", gv(sw_in), gv(tc), gv(pn), stat, gv(soil), gv(Au),
        tf$doy, tf$year, tf$month, monthly_out, as.integer(device))
  # -> list(wn, ro, pet, aet, snow, cond, bflow, netr, sm_lim) of [ncell_blk x nl] matrices, which
  #    writeValues(brick, m, bs$row[i]) / matXX[startind[i]:endind[i], ] <- m consume unchanged
  #    (R/splash.grid.R:370-392)
}

# unSWC (R/unsSWC.R:14): same signature; the analytical integral runs in the HIP library
unSWC <- function(soil_data, uns_depth, wn) {
  res <- .Call("C_splash_unswc_run", matrix(as.numeric(wn), ncol = 1), matrix(as.numeric(soil_data), nrow = 1),
               as.numeric(uns_depth))
  result <- list(wn_z = res$wn_z[, 1], wtd = res$wtd[, 1])
  if (xts::is.xts(wn)) result <- mapply(FUN = xts::xts, result, MoreArgs = list(order.by = time(wn)), SIMPLIFY = FALSE)
  result$theta_i <- if (xts::is.xts(wn)) xts::xts(res$theta_i[, 1], time(wn)) else res$theta_i[, 1]
  result
}
