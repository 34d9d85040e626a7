// splash_kernels.hpp -- argument blocks and launcher prototypes shared by the kernels
// (splash_kernels.hip) and the C ABI (splash_capi.hip).  Internal; not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace splash {

#define SPLASH_NDIAG_K 6

// rows of the cell-parameter array cellpar[SPLASH_NPAR][ncell] (see include/splash_hip.h)
enum {
    P_LAT = 0, P_ELV, P_SLOP, P_ASP /* already asp-180 */,
    P_SAT, P_WP, P_FC, P_KSAT, P_LAMBDA, P_DEPTH, P_BUB, P_RES, P_AU, P_AI, P_CELLIN, P_CELLOUT,
    P_WMAXR, P_NPAR
};

// one calendar day of the shared per-day table (uniform index => scalar loads)
struct DayRec {
    double dr, sd, cd;   // orbit: distance factor, sin/cos of declination (src/SOLAR.cpp:114-124)
    double ttm_s, inv_trm14; // month factors of frain_func (R/splash.point.R:549-550): sin term of Ttm, 1/(1.4 Trm)
    int month_id;        // running month number since the first day of the calendar
    int mdays;           // days of that month present in the calendar
    int flags;           // bit0: last calendar day of its month
    int doy;             // day of year (synthetic generator)
};

// per-device lookup table of the float viscosity routine (splash_device.hpp, k_visc_tables)
struct ViscTables {
    const float* rtab; // [kViscNR][8]: coef2[0..5], 2 x padding   (index: bits(rbar) - kViscR0)
};
size_t visc_table_floats_r();
hipError_t launch_visc_tables(float* rtab, hipStream_t s);

struct SpinArgs {
    const DayRec* days; // first 365 days
    int n;
    size_t ncell;
    const double *sw, *tc, *pn;
    size_t ldf;
    const double* Tt;
    double* par;
    size_t ldp;
    double* state;
    size_t lds;
    int* passes;
};

// one phase of the grid spin-up (see k_spin_phase)
struct SpinPhaseArgs {
    const DayRec* days; // first 365 days
    int n;
    size_t ncell;
    const double *sw, *tc, *pn;
    size_t ldf;
    const double* Tt;
    double* par;
    size_t ldp;
    double* state; // final state, and parking space of unfinished jobs
    size_t lds;
    int* passes;            // optional [2][ncell]
    const uint32_t* wl_in;  // cell ids of this phase (NULL: all cells, phase 0)
    const uint32_t* n_in;   // device-side length of wl_in (NULL: ncell)
    uint32_t* wl_out;       // cells that need another phase
    uint32_t* n_out;
    int* meta;              // [ncell] (stage << 16) | pass count of parked jobs
    double* ai;             // [ncell] aridity index
    int budget;             // passes per job in this phase
    int fresh;              // phase 0
    unsigned char* pending; // [ncell] or NULL: set to 1 for every job parked by this launch
    ViscTables vt;
    double* cache;          // records of the day terms (compact: [365][15][cache_cap]; rich: [cache_cap][365][20]), or NULL
    size_t cache_cap;       // jobs (launch positions) the cache has room for
    size_t cell0;           // phase 0 runs in blocks of cache_cap cells: first cell of this launch
    size_t ntotal;          // cells of the whole call (row stride of `passes`)
    size_t slot0;           // first launch position this launch handles (tail: jobs beyond the cache)
    int readback;           // 1: the read-back kernel (every job's records are in the cache already)
    int no_queue;           // 1: park unfinished jobs without queueing them (a launch over the same
                            //    cells follows and does the queueing)
    const uint32_t* first_dev; // optional (packed launches): device-side first list position of this launch
    int slot_by_cell;       // 1: a job's cache slot is its cell's position in the block (cell - cell0), not its launch position
    int spread;             // read-back kernel only: deal the jobs over 1024 wavefronts (ceil(n/1024) each) instead of packing
                            //    64 to a wavefront -- the tail's last stage (see k_spin_phase)
    const uint32_t* wl_slots; // optional: the launch's list holds POSITIONS of wl_in (and cache slots), not cells
    int queue_slots;        // 1: unfinished jobs are queued by their position (for a wl_slots launch), not by cell
    int dbg_active_lanes;   // measurement builds only (tools/solo_tail.py): lanes of a spread wavefront that stay active
    int no_nd_skip;         // test aid (SPLASH_HIP_NO_ND_SKIP): the cycle detector compares the snow age too, always
    int lds_pad;            // spread launches: bytes of dynamic LDS asked for and never touched (one workgroup per CU), 0: none
    int rich;               // 1: the cache holds rich records, 20 words each, record by record (cache_cap = spin_cache_rich_cap of the
                            //    workspace's capacity), see splash_kernels.hip "Two layouts"
};
size_t spin_cache_bytes_per_slot(); // compact records
size_t spin_cache_rich_cap(size_t cap); // slots of rich records the same bytes hold

struct RunArgs {
    const DayRec* days; // day t0 onwards
    int nt;
    size_t ncell;
    const double *sw, *tc, *pn, *snowfall; // snowfall != NULL: explicit split (module mirror)
    size_t ldf;
    const double* Tt;
    const double* par;
    size_t ldp;
    double* state;
    size_t lds;
    double* out[9];
    double* extra[3]; // module mirror only: qin_prev, tdrain, snwage
    size_t ldo;
    int monthly;
    int month0;       // month_id of day t0
    int head_partial; // month0 began before t0 (add to the stored partial)
    int tail_continues; // the call's last month goes on after its last day (a later call adds its days to the slot)
    int nmonth;       // months touched by [t0, t0+nt)
    // K7 water-balance totals (see include/splash_hip.h): exact fixed-point limbs, added with
    // integer atomics into acc[basin][month_id][SPLASH_NDIAG+1][2]
    long long* diag_acc;
    const int* basin; // [ncell] or NULL
    int n_basin;
    int nmonth_cal;   // months in the whole calendar (row stride of diag_acc)
    ViscTables vt;
    // list mode (the cells whose spin-up finished late, splash_capi.hip): lane j works on cell
    // cells[j] and writes its outputs to COLUMN j of compact [nt_out][ldo] arrays
    const uint32_t* cells;
    // main pass of an overlapped call: cells flagged here are still spinning up -- their outputs are
    // rewritten later in list mode; they take no part in the totals and their state is left alone
    const unsigned char* skip;
};

// Straggler pool of the host-buffer pipeline (splash_capi.hip): the jobs a block's spin-up left for
// its tail phase are copied -- forcing columns, parameters, threshold, parked state, job record -- to
// compact arrays shared by all blocks of the call, so that the block's buffers can be reused at
// once and the tail runs ONCE per call, over all blocks' stragglers together.
struct PoolGatherArgs {
    // the block (source)
    const uint32_t* cells;   // worklist of the block's pending jobs (cell ids in the block)
    const uint32_t* n_cells; // its device-side length
    int nt;
    const double *sw, *tc, *pn; // [nt][ldf]
    size_t ldf;
    const double* par;       // [P_NPAR][ldp]
    size_t ldp;
    const double* state;     // [5][lds]
    size_t lds;
    const double* Tt;
    const int* meta;
    const double* ai;
    const int* passes;       // [2][ntotal] or NULL
    size_t ntotal;
    const int* basin;        // or NULL
    uint64_t origin0;        // position of the block's cell 0 in the worker's list of valid cells
    // the pool (destination), capacity cap; slot = *res_base + j
    size_t cap;
    double *p_sw, *p_tc, *p_pn; // [nt][cap]
    double* p_par;           // [P_NPAR][cap]
    double* p_state;         // [5][cap]
    double* p_Tt;
    int* p_meta;
    double* p_ai;
    int* p_passes;           // [2][cap] or NULL
    int* p_basin;            // or NULL
    uint64_t* p_origin;
    uint32_t* p_wl;          // worklist of the pool's tail phase: identity
    uint32_t* p_n;           // jobs in the pool (atomic counter; the tail phase's input count)
    uint32_t* res;           // [3] per block, written by the reservation: base slot, jobs that fit, jobs of the block
};
hipError_t launch_pool_gather(const PoolGatherArgs& a, hipStream_t s);

hipError_t launch_soil_hydro(size_t ncell, const double* lat, const double* elev,
                             const double* slop, const double* asp, const double* resolution,
                             const double* soil, const double* au, double* par, hipStream_t s);
hipError_t launch_snow_threshold(size_t ncell, int nt, const double* tc, size_t ldf,
                                 const double* par, size_t ldp, double* Tt, hipStream_t s);
hipError_t launch_spin_phase(const SpinPhaseArgs& a, hipStream_t s);
hipError_t launch_spin_up_mod(const SpinArgs& a, double* const* series_dev, size_t sld,
                              const double* snowfall, hipStream_t s);
hipError_t launch_run_all(const RunArgs& a, hipStream_t s);
hipError_t launch_run_all_mod(const RunArgs& a, hipStream_t s);
hipError_t launch_diag_finalize(const long long* acc, int n_rec, double* diag, double* counts,
                                hipStream_t s);
hipError_t launch_terrain(int H, int W, const double* elev, double ymax, double yres, double xres,
                          int na_to_zero, double* slope, double* aspect, double* lat, double* res,
                          double* flowdir, hipStream_t s);
hipError_t launch_ncellflow(int H, int W, const double* fd, const double* codes, int pad,
                            double* out, hipStream_t s);
hipError_t launch_sinkfill(int H, int W, const double* elev, double resolution_m, double degree,
                           double* filled, int* sweeps_out, hipStream_t s);
hipError_t launch_mfd_area(int H, int W, const double* elev, double resolution_m, double* area,
                           int* levels_out, hipStream_t s);
hipError_t launch_flow_accumulation(int H, int W, const double* fd, const double* elev, int tau,
                                    const double* resolution, double* ups_ncell, double* ups_area,
                                    int* levels_out, hipStream_t s);
hipError_t launch_widen_f32(const float* src, size_t n, double* dst, hipStream_t s);
// src and dst 16- and 8-byte aligned (whole series of a block buffer are)
hipError_t launch_narrow_f32(const double* src, size_t n, float* dst, hipStream_t s);
hipError_t launch_evap_batch(size_t m, const double* lat, const double* elv, const double* sw,
                             const double* dr, const double* sd, const double* cd,
                             const double* sw_in, const double* tc, const double* slop,
                             const double* asp, const double* snow, const double* nd, double* out,
                             ViscTables vt, hipStream_t s);
hipError_t launch_unswc(size_t ncell, int nt, const double* wn, size_t ldw, const double* soil,
                        double uns_depth, double* wn_z, double* wtd, double* theta, size_t ldo,
                        hipStream_t s);
hipError_t launch_unswc_grid(size_t ncell, int nt, const double* wn, size_t ldw, const double* soil, double uns_depth,
                             double* w_z, double* wtd, double* se, double* theta_mean, size_t ldo, hipStream_t s);
hipError_t launch_math_probe(size_t n, const double* x, const double* y, double* out, hipStream_t s);
hipError_t launch_infiltration_probe(size_t n, const double* par, const double* wn,
                                     const double* ksat, const double* P, double* theta_surf,
                                     double* infi, hipStream_t s);
hipError_t launch_moist_surf(double depth, double z, double bub_p, double wn, double SAT,
                             double RES, double lambda, double* out, hipStream_t s);
hipError_t launch_inf_GA(double bub_press, double theta_i, double Ksat, double theta_s,
                         double lambda, double P, double tdur, double slop, double* out,
                         hipStream_t s);
hipError_t launch_synth_statics(uint64_t seed, int H, int W, int64_t cell0, size_t ncell,
                                int terrain, double* lat, double* elev, double* slop, double* asp,
                                double* resolution, double* soil, double* au, hipStream_t s);
hipError_t launch_synth_forcing(uint64_t seed, const DayRec* days, int t0, int nt, int64_t cell0,
                                size_t ncell, const double* lat, const double* elev, double* sw_in,
                                double* tc, double* pn, size_t ldf, hipStream_t s);

} // namespace splash
