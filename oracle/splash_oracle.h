/*
 * splash_oracle.h -- CPU restatement of the rsplash hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: a plain-C, line-by-line restatement of the reference
 * algorithm (splash.grid -> splash.point -> SPLASH::spin_up / run_all -> run_one_day).
 * It is NOT part of the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (rsplash_amd/) never
 * links, imports or falls back to anything in oracle/.
 *
 * Pin status (see DESIGN.md "Oracle"):
 *   - SOLAR / EVAP (src/SOLAR.cpp, src/EVAP.cpp, src/global.cpp) are pinned by execution
 *     against oracle/_ref (those three reference files compile from their own sources).
 *   - SPLASH.cpp (run_one_day, spin_up, run_all, moist_surf, inf_GA) includes <Rcpp.h>,
 *     which this image lacks; it is unbuildable here without writing a stand-in header,
 *     so that part of the restatement is "parity unpinned" (by execution).
 *   - The R prelude/epilogue (soil_hydro, snowfall split, AI, sm_lim, monthly) has no R
 *     interpreter here: "parity unpinned" (by execution).
 *
 * All citations are relative to /root/reference/.
 */
#ifndef SPLASH_ORACLE_H
#define SPLASH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/srad.h:44-53 */
typedef struct {
    double ru, rv, rw, rnl, hn, rn_d, rnn_d, ts;
} orc_srad;

/* src/etr.h:44-56 */
typedef struct {
    double cond, eet, pet, aet, snowmelt, sublimation, econ, pw, rn_d, visc, pet_max;
} orc_etr;

/* src/smr.h:51-60 */
typedef struct {
    double sm, ro, swe, bflow, sqout, tdr, nd, pet;
} orc_smr;

/* R/splash.point.R:401-413 (the fields the hot path uses) */
typedef struct {
    double SAT, FC, WP, bd, AWC, Ksat, A, B, theta_c, RES, bubbling_p;
} orc_soil;

/* ---- SOLAR / EVAP (pinned against oracle/_ref) ---- */
void orc_solar(double lat, double elv, int n, int y, double sw_in, double tc, double slop,
               double asp, double snow, double nd, double sw, orc_srad* out);
void orc_evap(double lat, double elv, double sw, int n, int y, double sw_in, double tc,
              double slop, double asp, double snow, double nd, orc_etr* out);
/* row-wise batches of the two calls above: out [m][8] / [m][11] */
void orc_solar_batch(int m, const double* lat, const double* elv, const int* n, const int* y,
                     const double* sw_in, const double* tc, const double* slop, const double* asp,
                     const double* snow, const double* nd, const double* sw, double* out);
void orc_evap_batch(int m, const double* lat, const double* elv, const double* sw, const int* n,
                    const int* y, const double* sw_in, const double* tc, const double* slop,
                    const double* asp, const double* snow, const double* nd, double* out);
float orc_viscosity_h2o(float tc, float p);
double orc_density_h2o(double tc, double p);
double orc_elv2pres(double z);
int orc_julian_day(int y, int m, int i);
/* day-static orbit terms: kN, nu, lambda, dr, delta (degrees) */
void orc_orbit(int n, int y, double* kN, double* nu, double* lam, double* dr, double* delta);

/* ---- SPLASH.cpp ---- */
double orc_moist_surf(double depth, double z, double bub_p, double wn, double SAT, double RES,
                      double lambda);
double orc_inf_GA(double bub_press, double theta_i, double Ksat, double theta_s, double lambda,
                  double P, double tdur, double slop);
double orc_get_wtd(double wn, double depth, double bub_press, double theta_s, double theta_r,
                   double lambda);
/* run_one_day == quick_run arithmetic (src/SPLASH.cpp:920-1588 == 150-918) */
void orc_run_one_day(double lat, double elv, int n, int y, double wn, double sw_in, double tc,
                     double pn, double slop, double asp, double snow, double snowfall,
                     const double* soil_info /*[12]*/, double qin, double td, double nd,
                     orc_smr* dsoil, orc_etr* dvap);
/* spin_up (src/SPLASH.cpp:1594-1749): outputs are [n] series (any may be NULL); returns #passes */
int orc_spin_up(double lat, double elv, int n, int y, const double* sw_in, const double* tair,
                const double* pn, double slop, double asp, const double* snowfall,
                const double* soil_info, double* sm, double* snow, double* qin, double* tdrain,
                double* ro, double* snwage, double* pet);
/* run_all (src/SPLASH.cpp:1833-1916): outputs [nt] (any may be NULL) */
void orc_run_all(double lat, double elv, int nt, const int* doys, const int* yrs,
                 const double* sw_in, const double* tair, const double* pn, double wn_last,
                 double slop, double asp, double snow_last, const double* snowfall,
                 const double* soil_info, double qin_last, double td_last, double nds_last,
                 double* wn, double* ro, double* pet, double* aet, double* snow, double* cond,
                 double* bflow, double* netr, double* qin_prev, double* tdrain, double* snwage);

/* ---- R prelude / epilogue (R/splash.point.R) ---- */
void orc_soil_hydro(double sand, double clay, double OM, double fgravel, double bd, orc_soil* out);
/* builds soil_info[12] + R-side Wmax (R/splash.point.R:96-115); au may be NULL (=> Au=0,3,3) */
void orc_soil_info(const double* soil_data /*[6]*/, const double* au /*[3] or NULL*/,
                   double resolution, double* soil_info /*[12]*/, double* wmax_r);
double orc_snowfall_prob(double tc, double lat, double elev);
/* rain/snow split over a series (R/splash.point.R:120-128, 521-558); returns Tt */
double orc_snow_partition(int nt, const double* tc, const double* pn, const int* month,
                          double lat, double elev, double* rain, double* snowfall);

/* R/unsSWC.R:14-88 (post-processing: water to depth uns_depth, water table, layer theta) */
void orc_unswc(const double* soil_data /*[6]*/, double uns_depth, int nt, const double* wn,
               size_t stride, double* wn_z, double* wtd, double* theta, double* cond,
               size_t ostride);
/* R/unsSWC.grid.R:14-141, one cell of the rasters (its four output files: swc_top, wtd, Se_top, theta_mean) */
void orc_unswc_grid(const double* soil_data /*[6]*/, double uns_depth, int nt, const double* wn, size_t stride,
                    double* w_z, double* wtd, double* se, double* theta_mean, double* cond, size_t ostride);

/* options for the point/grid drivers */
typedef struct {
    int monthly_out; /* 0 daily, 1 monthly aggregation (R/splash.point.R:207-214) */
    int n_threads;   /* grid driver only; <=0 => all */
} orc_opts;

/* per-cell auxiliary results of the driver (for parity tests of the intermediate stages) */
typedef struct {
    double soil_info[12];
    double wmax_r;
    double Tt;
    double AI;
    int passes1, passes2;
    double state0[5]; /* wn, snow, qin, td, nd handed to run_all */
    double state1[5]; /* state after the last transient day */
    int fragile;      /* day steps whose drainage time came from a rounding-noise transmittance */
    int critical;     /* ... of which zero-or-one-ulp: sticky infinite drainage time or not */
} orc_cell_aux;

/* count of such steps on the calling thread since the last reset (see splash_oracle.c) */
long orc_fragile_events(int reset);
long orc_critical_events(int reset);

/*
 * splash.point for one cell (R/splash.point.R:29-225).  Forcing is read with a stride
 * (stride = ncell for the [nt][ncell] grid layout, 1 for a point series).  Outputs are
 * written with out_stride; out[k] for k = wn,ro,pet,aet,snow,cond,bflow,netr,sm_lim
 * (order of R/splash.point.R:182,201), each [nt_out]; NULL entries are skipped.
 * month_id[t] in [0,nmonth) groups days for the monthly aggregation.
 */
void orc_point(int nt, const int* doy, const int* year, const int* month, const int* month_id,
               int nmonth, const double* sw_in, const double* tc, const double* pn,
               size_t stride, double lat, double elev, double slop, double asp,
               const double* soil_data, const double* au, double resolution,
               const orc_opts* opt, double* const out[9], size_t out_stride, orc_cell_aux* aux);

/* splash.grid inner loop (R/splash.grid.R:277-308): every cell, OpenMP over cells.
 * Layouts: forcing [nt][ncell]; soil [6][ncell]; au [3][ncell] or NULL; out[k] [nt_out][ncell].
 * Cells whose statics or day-0 forcing are NaN are emitted as NaN (R NA cells). */
void orc_grid(int ncell, int nt, const int* doy, const int* year, const int* month,
              const int* month_id, int nmonth, const double* sw_in, const double* tc,
              const double* pn, const double* lat, const double* elev, const double* slop,
              const double* asp, const double* soil, const double* au, const double* resolution,
              const orc_opts* opt, double* const out[9], orc_cell_aux* aux /*[ncell] or NULL*/);

/* transient-only variant for the CPU baseline: run_all per cell from a given state
 * (Tt given; rain/snow split done on the fly; asp is ALREADY asp-180).  outputs
 * wn,snow,ro,aet,pet [nt][ncell]. */
void orc_grid_transient(int ncell, int nt, const int* doy, const int* year, const int* month,
                        const double* sw_in, const double* tc, const double* pn,
                        const double* lat, const double* elev, const double* slop,
                        const double* asp, const double* soil_info /*[12][ncell]*/,
                        const double* Tt, double* state /*[5][ncell] in/out*/,
                        double* const out5[5], int n_threads, int* fragile /*[ncell] or NULL*/);

/* the same loop with all 11 series of SPLASH::run_all, per-cell-day conditioning events and an
 * optional one-step mode (every day restarted from a given state series): see the definition */
void orc_grid_transient2(int ncell, int nt, const int* doy, const int* year, const int* month,
                         const double* sw_in, const double* tc, const double* pn,
                         const double* lat, const double* elev, const double* slop,
                         const double* asp, const double* soil_info, const double* Tt,
                         double* state, const double* prev /*[5][nt][ncell] or NULL*/,
                         double* const out11[11], unsigned char* events /*[nt][ncell] or NULL*/,
                         double hs_lim_deg, float* hs_out /*[nt][ncell] or NULL*/, int n_threads);
double orc_last_hs(void);
double* orc_debug_ptr(void);

/* ---- static rasters; parity unpinned, see the definitions ---- */
void orc_terrain(int H, int W, const double* elev, double ymax, double yres, double xres,
                 int na_to_zero, double* slope, double* aspect, double* lat, double* resolution,
                 double* flowdir);
/* the reference's large-grid terrain: gdaldem slope -s <scale> -compute_edges / gdaldem aspect -zero_for_flat -compute_edges
 * on the (optionally integer-rounded) DEM, R/splash.grid.R:83-84 (restated from gdaldem's published algorithm) */
void orc_terrain_gdal(int H, int W, const double* elev, double xres, double yres, double scale, int round_int,
                      double* slope, double* aspect);
void orc_ncellflow(int H, int W, const double* flowdir, const double* codes9, int pad, double* out);
void orc_mfd_area(int H, int W, const double* z, double res, double* area);
void orc_sinkfill(int H, int W, const double* z, double resolution_m, double degree, double* w);
void orc_flow_accumulation(int H, int W, const double* fd, const double* elev, int tau,
                           const double* resolution, double* ups_ncell, double* ups_area);

/* ---- synthetic inputs (SURVEY.md 8d spec; splitmix64 counter RNG) ---- */
double orc_synth_u(uint64_t seed, uint64_t stream, uint64_t cell, uint64_t day);
/* statics for cells [cell0, cell0+ncell) of an H x W grid (row-major cell index).
 * terrain=0: flat config; terrain=1: slope/aspect/upslope config.
 * soil [6][ncell], au [3][ncell]. */
void orc_synth_statics(uint64_t seed, int H, int W, int64_t cell0, int ncell, int terrain,
                       double* lat, double* elev, double* slop, double* asp, double* resolution,
                       double* soil, double* au);
/* forcing [nt][ncell] for the same cells; day index t0+t, doy[t] given */
void orc_synth_forcing(uint64_t seed, int H, int W, int64_t cell0, int ncell, int t0, int nt,
                       const int* doy, const double* lat, const double* elev, double* sw_in,
                       double* tc, double* pn);
/* calendar starting at year0-01-01: fills doy, year, month (1..12), month_id (0..) ; returns nmonth */
int orc_calendar(int year0, int nt, int* doy, int* year, int* month, int* month_id);

int orc_max_threads(void);
void orc_set_spin_trace(double* buf, int cap);
int orc_threads_obtained(int n);

#ifdef __cplusplus
}
#endif
#endif
