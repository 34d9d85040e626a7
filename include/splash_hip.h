/*
 * splash_hip.h -- C ABI of librsplash_hip.so: the MI355X (gfx950) implementation of rsplash's
 * gridded daily water/energy balance loop.
 *
 * This is the drop-in boundary.  The reference exposes the path through the Rcpp module
 * "splash_module" (src/SPLASH.cpp:12-21: class SPLASH(lat, elv) with methods spin_up, run_all,
 * run_one_year, moist_surf, inf_GA) which R/splash.point.R:92-172 drives once per cell, and
 * R/splash.grid.R:277-308 maps over the cells of a row block.  A per-cell call cannot feed a GPU,
 * so the boundary is offered at three levels, all plain C (pointers + sizes, no R, no torch):
 *
 *   1. splash_grid_run / splash_point_run   host buffers in, host buffers out: what a `.Call`
 *      shim behind splash.grid()/splash.point() binds (INTEGRATION.md shows the shim);
 *   2. splash_dev_*                         device-resident stages for callers that keep the
 *      rasters in HBM (bench.py, multi-GPU drivers): soil_hydro -> snow_threshold -> spin_up ->
 *      run_all (+ diagnostics);
 *   3. splash_mod_*                         one-cell mirrors of the Rcpp module's methods with
 *      the reference's argument lists, used by the parity tests.
 *
 * Layouts: every raster series is time-major [nt][ncell] double (== R's [ncell x nt] column-major
 * matrix from raster::getValues, R/splash.grid.R:278-280), cell index = row-major raster cell.
 * Every function returns 0 on success or a negative SPLASH_E* code; when `err` is non-NULL a
 * message is written there.  Nothing is freed or retained except what the *_create calls return.
 * The library has no CPU FALLBACK: without a HIP device every compute entry point fails with
 * SPLASH_ENODEV.  (ABI 6: a caller may CHOOSE the host backend for one-site / few-cell calls --
 * splash_opts.backend, splash_point_run_backend, splash_mod_backend: the same day step, the
 * library's own device functions compiled for the host; it is never chosen for the caller.)
 */
#ifndef SPLASH_HIP_H
#define SPLASH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPLASH_ABI_VERSION 6

enum {
    SPLASH_OK = 0,
    SPLASH_EINVAL = -1,  /* bad argument */
    SPLASH_ENODEV = -2,  /* no HIP device / HIP runtime error */
    SPLASH_ENOMEM = -3,  /* host or device allocation failed */
    SPLASH_EHIP = -4     /* kernel launch / runtime failure */
};

/* rows of the per-cell parameter array cellpar[SPLASH_NPAR][ld] produced by soil_hydro:
 * lat, elev, slop, asp-180 (R/splash.point.R:131), then soil_info[0..11] in the order of
 * R/splash.point.R:110-114 (SAT, WP, FC, Ksat, lambda, depth, bub_press, RES, Au, Ai, ncellin,
 * ncellout), then the R-side Wmax = theta_c*depth*1000 used by sm_lim (R/splash.point.R:102,197). */
#define SPLASH_NPAR 17
#define SPLASH_NSTATE 5 /* wn, snow, qin, td, nd : the carry of SPLASH::run_all, src/SPLASH.cpp:1878-1889 */
#define SPLASH_NOUT 9   /* wn, ro, pet, aet, snow, cond, bflow, netr, sm_lim (R/splash.point.R:182,201) */
#define SPLASH_NDIAG 6  /* precip(rain+snowfall), aet, ro, bflow, wn, snow : monthly totals over cells */
#define SPLASH_NEXTRA 3 /* qin_prev, tdrain, snwage : the state series run_all also returns, src/SPLASH.cpp:1913-1915 */
/* Water-balance totals are accumulated EXACTLY, as two 64-bit fixed-point limbs per total
 * (value = limb0 / 2^16 + limb1 / 2^48; every cell's monthly figure is split without rounding
 * above 2^-48 mm and added with integer atomics), so a total is bitwise identical for any
 * launch blocking, any number of GPUs and any reduction order, and an all-reduce of the limbs
 * over ranks is exact.  One accumulator record per (basin, calendar month):
 *   int64 acc[n_basin][nmonth][SPLASH_NDIAG + 1][2]
 * record SPLASH_NDIAG holds {cell-months added, values skipped because non-finite or >= 2^20}. */
#define SPLASH_DIAG_LIMBS 2
#define SPLASH_DIAG_REC ((SPLASH_NDIAG + 1) * SPLASH_DIAG_LIMBS)

int splash_hip_abi_version(void);
int splash_hip_device_count(void);            /* 0 when no device; never fails */
const char* splash_hip_last_error(void);      /* thread-local text of the last failure */
/* The host-buffer entry points keep their device buffers between calls (re-acquiring ~100 GB
 * costs seconds); this returns the idle ones to the runtime.  SPLASH_HIP_NO_POOL=1 in the
 * environment disables the retention altogether. */
void splash_hip_release_cache(void);
/* bytes of `device` the library's buffer cache holds (in use + idle) and the idle part alone; after a
 * call has returned everything it held is idle again (tests check that no buffer is lost) */
int splash_hip_cache_stats(int device, size_t* held_bytes, size_t* idle_bytes);

/* ---------------------------------------------------------------- calendar ---------------- */
/* Per-day tables shared by all cells: orbit terms (src/SOLAR.cpp:98-124, 291-374), the month
 * factors of frain_func (R/splash.point.R:549-550) and month bookkeeping for the monthly
 * aggregation (R/splash.point.R:208).  doy/year/month are what R passes as doys/yrs and
 * format(time_index,"%m").  The calendar owns a device copy on the current device. */
typedef struct splash_calendar splash_calendar;
int splash_calendar_create(int nt, const int* doy, const int* year, const int* month,
                           splash_calendar** out, char* err, size_t errlen);
void splash_calendar_destroy(splash_calendar* cal);
int splash_calendar_nt(const splash_calendar* cal);
int splash_calendar_nmonth(const splash_calendar* cal);
/* host-only (works without a GPU): distance factor dr and sin/cos of the solar declination of
 * one calendar day (src/SOLAR.cpp:98-124), the day-static terms the calendar tabulates */
int splash_orbit_day(int doy, int year, double* dr, double* sin_delta, double* cos_delta);
/* host copy of the orbit table, for tests: dr, sin(delta), cos(delta), each [nt] */
int splash_calendar_orbit(const splash_calendar* cal, double* dr, double* sd, double* cd);

/* ---------------------------------------------------------------- device-resident stages -- */
/* All pointers below are DEVICE pointers on the current device; `stream` is a hipStream_t
 * passed as void* (NULL = default stream).  Calls are asynchronous on that stream. */

/* K1 soil_hydro + soil_info (R/splash.point.R:96-115, 232-416).
 * soil [6][ncell] = sand, clay, OM, gravel, bulk density (NaN => estimated), depth;
 * au [3][ncell] = upslope area, ncellin, ncellout, or NULL (=> 0,3,3). */
int splash_dev_soil_hydro(size_t ncell, const double* lat, const double* elev, const double* slop,
                          const double* asp, const double* resolution, const double* soil,
                          const double* au, double* cellpar /*[SPLASH_NPAR][ncell]*/, void* stream);

/* K2 snowfall threshold Tt = max(tc[p_snow >= 0.5]) over the whole series
 * (R/splash.point.R:120-122); -Inf where no day qualifies.  tc is [nt][ldf]. */
int splash_dev_snow_threshold(size_t ncell, int nt, const double* tc, size_t ldf,
                              const double* cellpar, size_t ldp, double* Tt, void* stream);

/* K4 the two spin-ups of splash.point (R/splash.point.R:141-152 -> SPLASH::spin_up,
 * src/SPLASH.cpp:1594-1749) on the first 365 records of the series, run as day-of-year 1..365 of
 * the first year whatever date the series starts on (quick_run(i+1, y[1], ...), src/SPLASH.cpp:1658;
 * the rain/snow split keeps the months of the real dates): spin-up #1, aridity index
 * AI = sum(pet)/sum(P), cellout := AI (written to cellpar row 15), spin-up #2.  Writes the state
 * handed to run_all.  passes (optional) [2][ncell] int: executed passes of each spin-up.
 * Spin-up #1 contributes nothing but the PET of its first pass (the reference iterates it to
 * convergence and discards the result): it is run to convergence only when `passes` is given.
 * The pass count is data dependent (2..1000 per spin-up), so the stage runs as a few phases over
 * shrinking worklists; `work` is scratch of splash_dev_spin_up_work_size(ncell) bytes. */
size_t splash_dev_spin_up_work_size(size_t ncell);
int splash_dev_spin_up(const splash_calendar* cal, size_t ncell, const double* sw_in,
                       const double* tc, const double* pn, size_t ldf, const double* Tt,
                       double* cellpar, size_t ldp, double* state /*[5][lds]*/, size_t lds,
                       int* passes, void* work, void* stream);

/* The spin-up in two parts, for callers that overlap its tail with other work.  A few cells per
 * thousand never meet the 1 mm stopping rule and run 1000 passes (src/SPLASH.cpp:1697): a chain of
 * 365 000 sequential day steps that keeps a handful of wavefronts busy for over a second while the
 * rest of the GPU idles.  _begin runs the phases that finish all other cells (their state is final
 * when it completes on its stream); _finish runs the rest, typically on a second stream, while the
 * caller already runs run_all for the finished cells (splash_dev_out.skip) and copies their
 * results; the late cells follow in list mode (splash_dev_out.cells).  Same workspace and same
 * arguments for both; splash_dev_spin_up == _begin + _finish on one stream. */
int splash_dev_spin_up_begin(const splash_calendar* cal, size_t ncell, const double* sw_in,
                             const double* tc, const double* pn, size_t ldf, const double* Tt,
                             double* cellpar, size_t ldp, double* state, size_t lds, int* passes,
                             void* work, void* stream);
int splash_dev_spin_up_finish(const splash_calendar* cal, size_t ncell, const double* sw_in,
                              const double* tc, const double* pn, size_t ldf, const double* Tt,
                              double* cellpar, size_t ldp, double* state, size_t lds, int* passes,
                              void* work, void* stream);
/* device pointers into `work`, valid once _begin has completed: the byte mask [ncell] of the cells
 * left for _finish, their list (uint32 cell ids) and the list's length (one uint32) */
int splash_dev_spin_up_pending(void* work, size_t ncell, const unsigned char** mask,
                               const uint32_t** cells, const uint32_t** n_cells);
/* the list's length on the host (waits for `stream`, the one _begin ran on) */
int splash_dev_spin_up_pending_count(void* work, size_t ncell, void* stream, uint32_t* n_pending);

typedef struct {
    double* out[SPLASH_NOUT]; /* each [nt_out][ldo] or NULL to skip that series */
    size_t ldo;
    int monthly;              /* 0: one row per day; 1: one row per calendar month
                                 (mean of wn, snow, sm_lim; sum of the others) */
    double* extra[SPLASH_NEXTRA]; /* daily mode only, optional: qin_prev, tdrain, snwage [nt][ldo] */
    /* K7 water-balance diagnostics (optional): monthly totals of SPLASH_NDIAG quantities per
     * basin, ADDED into diag_acc (the caller zeroes it once and may pass the same buffer to every
     * row block, time chunk and -- after an integer all-reduce -- rank).  Month index = the
     * calendar's running month number, so the buffer spans the whole calendar:
     * int64 [n_basin][splash_calendar_nmonth][SPLASH_NDIAG+1][2].  basin: [ncell] ids in
     * [0, n_basin) or NULL (= one basin, the whole grid; n_basin is then taken as 1).  Cells with
     * an id outside the range are left out.  The reference only has a TODO here (src/smr.h:44-46). */
    int64_t* diag_acc;
    const int32_t* basin;
    int n_basin;
    /* Overlapping the spin-up's long tail with the transient run (splash_dev_spin_up_begin/_finish):
     * skip [ncell] bytes (splash_dev_spin_up_pending): cells flagged there are still spinning up --
     * this call computes them from whatever state they have, but they take no part in the totals,
     * their state is left alone, and their outputs are to be rewritten by a later call in list mode;
     * cells [ncell] uint32: list mode -- lane j works on cell cells[j] (forcing, parameters, state
     * and basin of THAT cell; the array arguments of run_all then address cell 0) and writes its
     * outputs to column j of compact [nt_out][ldo] arrays. */
    const unsigned char* skip;
    const uint32_t* cells;
} splash_dev_out;

size_t splash_dev_diag_acc_size(int n_basin, int nmonth); /* bytes */
/* limbs -> doubles: diag[n_basin][nmonth][SPLASH_NDIAG] (device pointers; exact up to the final
 * rounding to double); counts (optional) [n_basin][nmonth][2] as doubles */
int splash_dev_diag_finalize(const int64_t* diag_acc, int n_basin, int nmonth, double* diag,
                             double* counts, void* stream);
/* the same conversion on the host (e.g. after summing the limbs of several devices or ranks) */
void splash_diag_finalize_host(const int64_t* diag_acc, int n_basin, int nmonth, double* diag,
                               double* counts);

/* K5 transient run: SPLASH::run_all (src/SPLASH.cpp:1833-1916) over days [t0, t0+nt) of the
 * calendar for ncell cells, with the rain/snow split (K2 part 2) fused into the loads.  Forcing
 * pointers address day t0 (row stride ldf).  state is read and overwritten (time-chunking).
 * In monthly mode, row m of an output is month (month_id(t0+..) - month_id(t0)). */
int splash_dev_run_all(const splash_calendar* cal, int t0, int nt, size_t ncell,
                       const double* sw_in, const double* tc, const double* pn, size_t ldf,
                       const double* Tt, const double* cellpar, size_t ldp, double* state,
                       size_t lds, const splash_dev_out* out, void* stream);

/* synthetic inputs on the device (generator spec: DESIGN.md "Synthetic inputs"; same
 * counter-based splitmix64 streams as the oracle's generator).  Cells [cell0, cell0+ncell) of an
 * H x W grid, row-major. */
int splash_dev_synth_statics(uint64_t seed, int H, int W, int64_t cell0, size_t ncell, int terrain,
                             double* lat, double* elev, double* slop, double* asp,
                             double* resolution, double* soil /*[6][ncell]*/,
                             double* au /*[3][ncell]*/, void* stream);
int splash_dev_synth_forcing(uint64_t seed, const splash_calendar* cal, int t0, int nt,
                             int64_t cell0, size_t ncell, const double* lat, const double* elev,
                             double* sw_in, double* tc, double* pn, size_t ldf, void* stream);

/* test hook: SOLAR+EVAP for m independent rows (src/EVAP.cpp:82-264), out [m][11] in the order
 * of src/etr.h:44-56 (cond, eet, pet, aet, snowmelt, sublimation, econ, pw, rn_d, visc, pet_max) */
int splash_dev_evap_batch(size_t m, const double* lat, const double* elv, const double* sw,
                          const int* n, const int* y, const double* sw_in, const double* tc,
                          const double* slop, const double* asp, const double* snow,
                          const double* nd, double* out, void* stream);

/* ---------------------------------------------------------------- static rasters */
/* What splash.grid derives from the elevation raster before the loop, for a north-up
 * longitude/latitude grid of H rows x W columns (row-major, NaN = NA, elevation in metres):
 * slope and aspect in degrees (raster::terrain(opt=c('slope','aspect'), unit='degrees'): Horn's
 * 3x3 differences, aspect clockwise from north, NA on the outer ring and next to NA cells, which
 * R/splash.grid.R:107 turns into 0 where an elevation exists: na_to_zero), cell-centre latitude
 * (:101-103), resolution = sqrt(cell area) in metres (:98) and the D8 flow direction code
 * (raster::terrain(opt='flowdir'), R/upslope_area.R:31).  ymax = northern edge, yres/xres = cell
 * size, all in degrees.  Device pointers; outputs [H*W], any may be NULL.  The arithmetic of
 * raster::terrain / raster::area is not in the reference tree: this follows the published
 * algorithms (parity unpinned). */
int splash_dev_terrain(int H, int W, const double* elev, double ymax, double yres, double xres,
                       int na_to_zero, double* slope_deg, double* aspect_deg, double* lat,
                       double* resolution_m, double* flowdir, void* stream);
/* ncellflow (R/upslope_area.R:140-165): count of the 3x3 neighbours whose flow code equals
 * codes9[position] (window in row order; device pointer to 9 doubles, e.g. 2,1,128,4,0,64,8,16,32
 * for inout='in', met='top' and the reverse for 'out'); 0 -> 1; NA if the window is all NA.
 * pad = 0: outer ring NA (raster::focal's default); pad = 1: NA outside the grid. */
int splash_dev_ncellflow(int H, int W, const double* flowdir, const double* codes9, int pad,
                         double* ncell, void* stream);
/* Sink filling (R/upslope_area.R:15-16 `topmodel::sinkfill(as.matrix(dem), resolution, 1)`; :68 TauDEM
 * pitremove): the surface is raised until every cell has a path to an outlet (grid edge or NA
 * neighbour) that descends by at least tan(degree) * resolution_m per cell (sqrt 2 times that
 * along diagonals); degree = 0 fills to level.  NA stays NA.  filled: [H*W], not elev itself.
 * sweeps (optional, host): relaxation sweeps run.  The third-party codes are not in the reference
 * tree: this is Planchon & Darboux (2002) (parity unpinned).  Synchronises the stream. */
int splash_dev_sinkfill(int H, int W, const double* elev, double resolution_m, double degree,
                        double* filled, int* sweeps, void* stream);
/* Contributing area by D8 accumulation (R/upslope_area.R:21 `topidx()$area`; :74 TauDEM aread8 and
 * :85-96): ups_ncell = number of cells draining through a cell, itself included (>= 1 wherever
 * there is a cell: the reference's fix_coast); ups_area = ups_ncell * resolution_m^2 (the reference's
 * ups_ncell * cell area).  flowdir: raster's codes (splash_dev_terrain) or, tau_codes = 1, TauDEM's
 * 1..8.  elev (optional) says where cells are (NaN = none); without it, where flowdir is not NaN.
 * Cells on a closed loop of codes (pits left unfilled) keep what drains into them and pass nothing
 * on.  levels (optional, host): host-synchronised levels of the order.  Synchronises the stream. */
int splash_dev_flow_accumulation(int H, int W, const double* flowdir, const double* elev,
                                 int tau_codes, const double* resolution_m, double* ups_ncell,
                                 double* ups_area, int* levels, void* stream);
/* Contributing area by the multiple-flow-direction algorithm of Quinn et al. (1991) -- what
 * `topmodel::topidx(as.matrix(elev), resolution)$area` returns for R/upslope_area.R:21 (small areas;
 * third-party code outside the reference tree: the published algorithm, parity unpinned).  Every
 * cell starts with resolution_m^2 and passes its accumulated area to all lower neighbours in
 * proportion to tan(beta) * contour length (0.5 and 0.354 cell sizes).  elev: the (sink-filled)
 * surface, NaN = NA; area [H*W] in m2.  Deterministic (pull order fixed).  Synchronises the stream. */
int splash_dev_mfd_area(int H, int W, const double* elev, double resolution_m, double* area,
                        int* levels, void* stream);

/* The static rasters in ONE call on host buffers (what an R caller binds): everything
 * R/splash.grid.R:96-110 and upslope_area() (R/upslope_area.R:12-58) derive from the elevation
 * matrix.  elev [H*W] row-major (raster::as.matrix / getValues order), NaN = NA.  Outputs [H*W], any
 * may be NULL: slope, aspect (degrees, of the RAW dem as splash.grid takes them), lat, resolution (m);
 * filled (sink-filled surface: degree as in sinkfill(dem, mean resolution, degree)), flowdir (raster
 * codes, of the FILLED surface), ups_ncell, ups_area (m2: D8, as upslope_areav2), ncellin, ncellout
 * (met='top' windows), ups_area_mfd (m2: as upslope_area's topidx). */
typedef struct splash_static_out {
    double *slope, *aspect, *lat, *resolution;
    double *filled, *flowdir, *ups_ncell, *ups_area, *ncellin, *ncellout;
    double* ups_area_mfd; /* topidx()$area: multiple-flow-direction area of the filled surface, m2 */
} splash_static_out;
int splash_static_rasters_run(int H, int W, const double* elev, double ymax, double yres, double xres,
                              int na_to_zero, double degree, const splash_static_out* out,
                              char* err, size_t errlen);

/* The reference's LARGE-GRID terrain (R/splash.grid.R:77-94: grids of more than 1e7 cells or of 0.1-1 degree): the DEM
 * is written as 16-bit integers (R/upslope_area.R:64) and `gdaldem slope -s 111120 -compute_edges`, `gdaldem aspect
 * -zero_for_flat -compute_edges` give slope and aspect (:83-84): Horn's formula with ONE metres-per-degree scale for
 * both axes, aspect as an azimuth with flat = 0, edge cells and cells next to NA computed (no NA ring to zero).  GDAL is
 * a system tool, in neither tree nor image: gdaldem's published algorithm restated, parity unpinned.  elev [H][W]
 * row-major, north row first, NaN = NA; xres, yres in degrees; scale = 111120 for the reference's call; round_int != 0
 * rounds the elevations to whole metres first (the INT2S file).  Either output may be NULL. */
int splash_dev_terrain_gdal(int H, int W, const double* elev, double xres, double yres, double scale, int round_int,
                            double* slope_deg, double* aspect_deg, void* stream);
int splash_terrain_gdal_run(int H, int W, const double* elev, double xres, double yres, double scale, int round_int,
                            double* slope_deg, double* aspect_deg, char* err, size_t errlen);

/* ---------------------------------------------------------------- unSWC epilogue */
/* R/unsSWC.R:14-88 / R/unsSWC.grid.R: from the column water content wn [mm] (output of the path
 * above), the water held between the surface and uns_depth [m] (wn_z, mm), the water-table depth
 * (wtd, m) and the mean volumetric content of that layer (theta).  soil [6][ncell] as for
 * soil_hydro (the gravel fraction is applied as the reference does, :55-56, :82).  Any of the three
 * outputs may be NULL. */
int splash_dev_unswc(size_t ncell, int nt, const double* wn, size_t ldw, const double* soil,
                     double uns_depth, double* wn_z, double* wtd, double* theta, size_t ldo,
                     void* stream);
int splash_unswc_run(size_t ncell, int nt, const double* wn /*[nt][ncell]*/,
                     const double* soil /*[6][ncell]*/, double uns_depth, double* wn_z, double* wtd,
                     double* theta, char* err, size_t errlen);
/* ABI 6: the same with the backend chosen by the caller (SPLASH_BACKEND_HOST: the kernel's two functions on the
 * calling thread -- the README example, README.md:52-59, then runs without a GPU; splash_unswc_run == backend HIP) */
int splash_unswc_run_backend(int backend, size_t ncell, int nt, const double* wn, const double* soil,
                             double uns_depth, double* wn_z, double* wtd, double* theta, char* err, size_t errlen);

/* unSWC.grid (R/unsSWC.grid.R:14-141): the GRIDDED function has its own semantics -- no stoniness correction (:65, :73),
 * no patch of a non-positive integral, the integral's water table capped at uns_depth (:115 against :29) while the one
 * written out is capped at the soil depth (:98-106) -- and its own outputs, the four rasters it writes: w_z = water
 * between the surface and uns_depth [mm] ("swc_top", :115), wtd [m] (:108), Se = effective saturation of that layer
 * (:120-126), theta_mean = the column's clamped mean volumetric content (:79-87).  soil [6][ncell]; any output NULL.
 * splash_unswc_grid_run: host buffers, in column blocks sized to the device; backend as in splash_unswc_run_backend. */
int splash_dev_unswc_grid(size_t ncell, int nt, const double* wn, size_t ldw, const double* soil, double uns_depth,
                          double* w_z, double* wtd, double* se, double* theta_mean, size_t ldo, void* stream);
int splash_unswc_grid_run(int backend, size_t ncell, int nt, const double* wn /*[nt][ncell]*/,
                          const double* soil /*[6][ncell]*/, double uns_depth, double* w_z, double* wtd, double* se,
                          double* theta_mean, char* err, size_t errlen);

/* The names SURVEY.md 8(b) lists for the core library that the entry points above do not already carry:
 * splash_hip_version == splash_hip_abi_version; two test hooks on host arrays, both on the calling thread's module
 * backend (splash_mod_backend):
 * splash_day_step: ONE day step of one cell = SPLASH::run_one_day (src/SPLASH.cpp:920-1588; splash_mod_run_all with
 *   nt = 1).  state_in[5]: wn, snow, qin, td, nd of the day before; out[11]: wn, ro, pet, aet, snow, cond, bflow, netr,
 *   qin, tdrain, snwage.  soil_info as for splash_mod_run_all.
 * splash_soil_hydro: K1 alone -- soil_hydro and the soil_info vector (R/splash.point.R:96-115, 232-416) -> cellpar
 *   [SPLASH_NPAR][ncell]; soil [6][ncell]; au [3][ncell] or NULL (Au = 0, 3 cells in, 3 out). */
int splash_hip_version(void);
int splash_day_step(double lat, double elv, int n, int y, double sw_in, double tc, double pn, double snowfall,
                    double slop, double asp, const double* soil_info, const double* state_in, double* out);
int splash_soil_hydro(size_t ncell, const double* lat, const double* elev, const double* slop, const double* asp,
                      const double* resolution, const double* soil, const double* au, double* cellpar, char* err,
                      size_t errlen);

/* test hook: the device build of the FP64 log/exp/pow/acos used by the day step, for n arguments
 * (host pointers): out [7][n] = flog(x), fexp(x), fpow(x,y), facos(x), fsqrt(x), fatan(x), fatan2(x,y) */
int splash_dev_math_probe(size_t n, const double* x, const double* y, double* out);

/* test hook: the day loop's own forms of SPLASH::moist_surf (z = 10 cm, src/SPLASH.cpp:1265,
 * 1921-1961) and SPLASH::inf_GA (tdur = 6 h, :1268, 1963-2023) for n independent rows (host
 * pointers): cellpar [SPLASH_NPAR][n], column water wn, viscosity-corrected conductivity and
 * inflow P -> surface moisture and infiltration.  (splash_mod_moist_surf / splash_mod_inf_GA are
 * the literal general-argument mirrors of the module methods.) */
int splash_dev_infiltration_probe(size_t n, const double* cellpar, const double* wn,
                                  const double* ksat_visc, const double* P, double* theta_surf,
                                  double* infiltration);

/* ---------------------------------------------------------------- host-buffer entry points - */
typedef struct {
    size_t ncell;
    int nt;
    const int *doy, *year, *month;           /* [nt] */
    const double *sw_in, *tc, *pn;           /* [nt][ncell] (W m-2, deg C, mm) */
    const double *lat, *elev, *slop, *asp;   /* [ncell] degrees / m; asp clockwise from north */
    const double* resolution;                /* [ncell] m */
    const double* soil;                      /* [6][ncell] */
    const double* au;                        /* [3][ncell] or NULL */
    const int32_t* basin;                    /* [ncell] basin ids in [0, n_basin) or NULL */
    int n_basin;                             /* 0/1 without basin ids */
    /* optional single-precision transport of the forcing (the reference's rasters are FLT4S:
     * widening is lossless; arithmetic stays FP64).  When sw_in is NULL these are used instead. */
    const float *sw_in_f32, *tc_f32, *pn_f32; /* [nt][ncell] */
    /* (ABI 5) A long series in time chunks: continuing a run instead of starting one.  The module's
     * run_all takes the state to start from (wn_last, snow_last, qin_last, td_last, nds_last,
     * src/SPLASH.h:93-97); splash.point always hands it the end of its own spin-up
     * (R/splash.point.R:158-172).  A caller whose series does not fit host memory in one piece runs
     * the first chunk as usual (spin-up on its first 365 days) and every later chunk with
     *   state0 [5][ncell] = splash_grid_out.state_end of the chunk before (wn, snow, qin, td, nd),
     *   ai0    [ncell]    = the aridity index of the run's spin-up (splash_grid_out.aux row 1 of the
     *                       first chunk; it is soil_info[12] of every run_all call, :150),
     *   tt     [ncell]    = the snowfall threshold (below);
     * such a call runs no spin-up (nt may be any length >= 1) and starts run_all from state0.  Chunks
     * that begin on the first day of a month give the monthly outputs of the undivided call.
     * tt: the snowfall threshold Tt = max(tc[p_snow >= 0.5]) is a property of the WHOLE series
     * (R/splash.point.R:120-122).  When tt is given (with or without state0) it is used instead of the
     * threshold of this call's tc: the maximum over the chunks' thresholds (splash_snow_threshold_run on
     * each chunk; NaN if any is NaN) is the threshold of the whole series, and with it the chunked run
     * equals the undivided one bit for bit.  state0 and ai0 come together and need tt. */
    const double* tt;
    const double* state0;
    const double* ai0;
} splash_grid_in;

typedef struct {
    int monthly_out;    /* sim.control$monthly_out, R/splash.grid.R:28 (default TRUE there) */
    int device;         /* HIP device ordinal (used when n_devices <= 1 and devices == NULL) */
    size_t max_cells_per_launch; /* 0 = size blocks from free HBM */
    /* Multi-GPU inside one call (replaces the SOCK-cluster block scheduler of
     * R/splash.grid.R:260-314, 359-400): the valid cells are cut into chunks (about sixteen per
     * device, 256 cells at least) that are dealt round-robin to the listed devices -- every device
     * sees every latitude band, so they finish together (round 6; a contiguous range per device left
     * the low-latitude one behind) --, one host thread + stream set per device.  devices == NULL
     * means ordinals 0..n_devices-1; the same ordinal may appear more than once (two workers on one
     * GPU).  Results are independent of the split, bit for bit, totals included. */
    int n_devices;
    const int* devices;
    /* (ABI 3) share of a device's free memory this call may plan its blocks with; 0 = 0.85.  A
     * caller that keeps several calls in flight on one device (splash_grid_submit) gives each its
     * share, e.g. 0.4 for two. */
    double mem_fraction;
    /* (ABI 6) SPLASH_BACKEND_HIP (0, the default) or SPLASH_BACKEND_HOST: the same day step -- the
     * product's own device functions compiled for the host, nothing else -- on CPU cores (one thread
     * per few cells), for calls of one or a few cells: a single site is a chain of sequential day
     * steps that one core steps through faster than a lone wavefront (BASELINE.json configs[0]: "CPU
     * reference path (plumbing, no GPU)").  An explicit choice, never a fallback: with
     * the default backend a machine without a HIP device gets SPLASH_ENODEV.  Host and device results
     * agree to rounding (<= 1e-12 relative on the bundled sites), not to the bit. */
    int backend;
} splash_opts;
#define SPLASH_BACKEND_HIP 0
#define SPLASH_BACKEND_HOST 1

typedef struct {
    double* out[SPLASH_NOUT]; /* caller-allocated [nt_out][ncell], NULL to skip */
    double* state_end;        /* optional [5][ncell] */
    double* diag;             /* optional [n_basin][nmonth][SPLASH_NDIAG] (n_basin = 1 without ids) */
    double* aux;              /* optional [4][ncell]: Tt, AI, passes1, passes2 */
    int64_t* diag_acc;        /* optional: the exact limbs behind diag, [n_basin][nmonth][SPLASH_NDIAG+1][2],
                                 for callers that sum over several calls or ranks */
    /* (ABI 4) opt-in single-precision transport of the results: series k is delivered to out_f32[k]
     * (caller-allocated [nt_out][ncell] float) instead of out[k], which must then be NULL.  The
     * arithmetic stays FP64; each value is rounded to the nearest float on the device, i.e. the array
     * holds exactly (float)x of what out[k] would have held -- what the reference's output bricks
     * store anyway (datatype FLT4S, R/splash.grid.R:186-216).  Halves the device->host bytes, which
     * bound a call with daily outputs.  All NULL = the default: double precision. */
    float* out_f32[SPLASH_NOUT];
} splash_grid_out;

/* splash.grid's inner computation for a block of cells (R/splash.grid.R:277-308): every cell gets
 * what splash.point computes (R/splash.point.R:92-214).  Cells with NaN statics or NaN day-0
 * forcing are returned as NaN (R's NA cells, e.g. ocean).  A gap (NaN) later in a series is not
 * a mask: it propagates through the arithmetic as it does in R (Tt = NA, NA rain and snow on the
 * days of possible snowfall, NaN state from the first such day). */
int splash_grid_run(const splash_grid_in* in, const splash_opts* opt, const splash_grid_out* out,
                    char* err, size_t errlen);

/* The same call in two halves, for a host loop over row blocks that wants more than one block in
 * flight (the reference's receive loop, R/splash.grid.R:359-400, keeps every worker of its cluster
 * busy while it writes the block that came back): _submit starts the call on its own host thread
 * and streams and returns at once; _wait blocks until the results are in `out` and returns the
 * call's code (the job is freed either way).  Every array behind `in`, `opt` and `out` must stay
 * valid and untouched until _wait returns; the three structs themselves are copied.  Calls in flight
 * on the same device share its memory: give each a splash_opts.mem_fraction.  Inside ONE call the
 * blocks are pipelined already (transfers of neighbouring blocks overlap the kernels, and the
 * spin-up's long tail runs once per call): a caller that can hand over a large block should. */
/* (ABI 5) K2 on host buffers: Tt [ncell] of the tc series given (exactly one of tc / tc_f32, [nt][ncell]);
 * -Inf where no day qualifies, NaN where tc has a gap (R's max() with NA).  The threshold of a series
 * processed in time chunks is the maximum of the chunks' thresholds (NaN if any is NaN). */
int splash_snow_threshold_run(size_t ncell, int nt, const double* tc, const float* tc_f32,
                              const double* lat, const double* elev, double* tt, char* err, size_t errlen);
/* (ABI 6) the same with the backend chosen by the caller (SPLASH_BACKEND_HOST: on the calling thread, for a series in
 * time chunks on the host backend; splash_snow_threshold_run == backend HIP) */
int splash_snow_threshold_run_backend(int backend, size_t ncell, int nt, const double* tc, const float* tc_f32,
                                      const double* lat, const double* elev, double* tt, char* err, size_t errlen);

typedef struct splash_grid_job splash_grid_job;
int splash_grid_submit(const splash_grid_in* in, const splash_opts* opt, const splash_grid_out* out,
                       splash_grid_job** job, char* err, size_t errlen);
int splash_grid_wait(splash_grid_job* job, char* err, size_t errlen);

/* splash.point (R/splash.point.R:29-225) for one site: the same pipeline with ncell = 1.
 * au: pointer to 1 value (scalar Au, ncellin=ncellout=3) when n_au==1, or 3 values. */
int splash_point_run(int nt, const int* doy, const int* year, const int* month,
                     const double* sw_in, const double* tc, const double* pn, double lat,
                     double elev, double slop, double asp, const double* soil_data /*[6]*/,
                     const double* au, int n_au, double resolution, int monthly_out,
                     double* const out[SPLASH_NOUT], char* err, size_t errlen);

/* (ABI 6) the same with the backend chosen (SPLASH_BACKEND_HIP / SPLASH_BACKEND_HOST, see splash_opts.backend):
 * what `splash.point(..., backend = "host")` of r/splash_hip.R binds.  splash_point_run IS this call with
 * SPLASH_BACKEND_HIP. */
int splash_point_run_backend(int backend, int nt, const int* doy, const int* year, const int* month,
                             const double* sw_in, const double* tc, const double* pn, double lat,
                             double elev, double slop, double asp, const double* soil_data /*[6]*/,
                             const double* au, int n_au, double resolution, int monthly_out,
                             double* const out[SPLASH_NOUT], char* err, size_t errlen);
/* 1 when the host backend runs its build for cores with fused multiply-add (x86-64: AVX2 + FMA), 0 for the
 * generic build; informational (timings) */
int splash_host_backend_fma(void);

/* ---------------------------------------------------------------- Rcpp-module mirrors ------ */
/* (ABI 6) backend of the splash_mod_* calls of the CALLING THREAD (the reference's module object lives in
 * one R process; SPLASH_BACKEND_HIP is the default): returns the previous setting, or SPLASH_EINVAL. */
int splash_mod_backend(int backend);
/* One-cell GPU evaluations with the argument lists of src/SPLASH.h:85-103 (host pointers).
 * soil_info has 12 (or 13) entries as built by R/splash.point.R:110-114. */
int splash_mod_spin_up(double lat, double elv, int n, int y, const double* sw_in,
                       const double* tair, const double* pn, double slop, double asp,
                       const double* snowfall, const double* soil_info,
                       double* sm, double* snow, double* qin, double* tdrain, double* ro,
                       double* snwage, double* pet /* each [n] or NULL */, int* passes);
int splash_mod_run_all(double lat, double elv, int nt, const int* doys, const int* yrs,
                       const double* sw_in, const double* tair, const double* pn, double wn_last,
                       double slop, double asp, double snow_last, const double* snowfall,
                       const double* soil_info, double qin_last, double td_last, double nds_last,
                       double* wn, double* ro, double* pet, double* aet, double* snow, double* cond,
                       double* bflow, double* netr, double* qin_prev, double* tdrain,
                       double* snwage /* each [nt] or NULL */);
/* run_one_year (src/SPLASH.cpp:1751): n days of year y from the last elements of the state
 * vectors (each of length n_end >= n) */
int splash_mod_run_one_year(double lat, double elv, int n, int y, int n_end, const double* sw_in,
                            const double* tair, const double* pn, const double* wn_vec, double slop,
                            double asp, const double* snow, const double* snowfall,
                            const double* soil_info, const double* qin_vec, const double* td_vec,
                            const double* nds, double* wn, double* ro, double* pet, double* aet,
                            double* snow_out, double* cond, double* bflow, double* netr,
                            double* qin_prev, double* tdrain, double* snwage /* each [n] or NULL */);
int splash_mod_moist_surf(double depth, double z, double bub_p, double wn, double SAT, double RES,
                          double lambda, double* theta);
int splash_mod_inf_GA(double bub_press, double theta_i, double Ksat, double theta_s, double lambda,
                      double P, double tdur, double slop, double* infiltration);

#ifdef __cplusplus
}
#endif
#endif
