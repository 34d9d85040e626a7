// splash_host.hpp -- what the translation units of the host side share: error reporting, the
// per-device buffer pool, the calendar's layout and the spin-up's phase driver.
//   splash_capi.hip       the C ABI's small entry points: calendar, device-resident stages, module
//                         mirrors, static rasters, unSWC, the buffer pool
//   splash_grid_host.hip  splash_grid_run / _submit / _wait: the pipelined host-buffer path
//                         (Worker / Lane / Pool) that stands where R/splash.grid.R:260-314, 359-400 stand
// Nothing here is part of the ABI (include/splash_hip.h is).
#pragma once
#include "../../include/splash_hip.h"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "splash_kernels.hpp"

namespace splash {
// (splash_statics.hip; declared here rather than in splash_kernels.hpp, whose text is part of the hash that ties the PMC
// constants of profiles/traffic.json to the day-loop kernels' sources)
hipError_t launch_terrain_gdal(int H, int W, const double* elev, double xres, double yres, double scale, int round_int,
                               double* slope, double* aspect, hipStream_t s);
namespace host {

// text of the calling thread's last failure (splash_hip_last_error)
extern thread_local std::string g_err;
int fail(int code, char* err, size_t errlen, const char* fmt, ...) __attribute__((format(printf, 4, 5)));
int need_device(char* err, size_t errlen);
int enter_device(char* err, size_t errlen); // knobs_refresh() + need_device(): public entry points (caller's thread)

// Environment knobs (tests, A/B timing, tuning aids; INTEGRATION.md lists them).  getenv() is only
// ever called on the thread that ENTERS the library (knobs_refresh(), first statement of the public
// entry points that start work): lane, pool and helper threads read the snapshot, so no thread of
// the library reads the environment while the caller's thread may be changing it between calls.
// A call (or a job in flight, splash_grid_submit) KEEPS the knobs it started with: the entry point copies the
// snapshot, and every thread that works for the call -- the job thread, per-device workers, lanes, delivery
// helpers -- installs that copy with a KnobScope; under a scope knobs() returns the copy and knobs_refresh() (which
// the public splash_dev_* stages call when a lane uses them) is a no-op (round 6, ADVICE r05: those stages
// re-read the environment on lane threads, the very race the snapshot was introduced against).
struct Knobs {
    bool no_pool = false;              // SPLASH_HIP_NO_POOL
    long spin_cache_cap = 0;           // SPLASH_HIP_SPIN_CACHE_CAP (frozen at the first entry: the workspace layout depends on it)
    int dbg_active_lanes = 0;          // SPLASH_DBG_ACTIVE_LANES (measurement builds)
    bool no_nd_skip = false;           // SPLASH_HIP_NO_ND_SKIP
    bool no_block_phase1 = false;      // SPLASH_HIP_NO_BLOCK_PHASE1
    bool no_solo_tail = false;         // SPLASH_HIP_NO_SOLO_TAIL
    int tail_stage[2] = {-1, -1};      // SPLASH_HIP_TAIL_STAGES="a,b"
    int tail_lds_pad = 57344;          // SPLASH_HIP_TAIL_LDS_PAD
    int prefault_mode = 0;             // SPLASH_HIP_PREFAULT (populate | touch | off)
    size_t prefault_min_bytes = (size_t)64 << 20; // SPLASH_HIP_PREFAULT_MIN_BYTES
    size_t prefault_delay_ms = 0;      // SPLASH_HIP_PREFAULT_DELAY_MS (test aid)
    bool test_prefault_whole_range = false; // SPLASH_HIP_TEST_PREFAULT_WHOLE_RANGE (fault injection, tests only)
    int test_throw = 0;                // SPLASH_HIP_TEST_THROW (fault injection, tests only): 1 = std::bad_alloc in the body of
                                       // grid_run, 2 = std::runtime_error on a lane's thread, 3 = on a device worker's thread
    bool trace = false;                // SPLASH_TRACE
    int lanes = 0;                     // SPLASH_HIP_LANES (0: chosen by the call)
    bool fake_device = false;          // SPLASH_HIP_FAKE_DEVICE_BYTES given
    size_t fake_device_bytes = 0;
    size_t pool_cap = 0;               // SPLASH_HIP_POOL_CAP (0: chosen by the call)
    size_t blocks_per_lane = 4;        // SPLASH_HIP_BLOCKS_PER_LANE
    size_t min_block_cells = (size_t)1 << 20; // SPLASH_HIP_MIN_BLOCK_CELLS
    bool no_register = false;          // SPLASH_HIP_NO_REGISTER
    size_t register_min_bytes = (size_t)8 << 30; // SPLASH_HIP_REGISTER_MIN_BYTES
};
enum { PREFAULT_POPULATE = 0, PREFAULT_TOUCH = 1, PREFAULT_OFF = 2 };
void knobs_refresh(); // the calling thread of a public entry point (no-op under a KnobScope)
Knobs knobs();        // the snapshot (any thread); the scope's copy under a KnobScope
// installs `k` as the calling thread's knobs for the scope's lifetime (nests: the previous one comes back)
class KnobScope {
public:
    explicit KnobScope(const Knobs& k);
    ~KnobScope();
    KnobScope(const KnobScope&) = delete;
    KnobScope& operator=(const KnobScope&) = delete;
private:
    Knobs mine_;
    const Knobs* prev_;
};

// device buffers of the host-buffer entry points are kept between calls (splash_capi.hip)
hipError_t pool_alloc(void** out, size_t bytes);
void pool_free(void* p);
size_t pool_idle_bytes(int dev);
size_t pool_held_bytes(int dev);
std::recursive_mutex& device_alloc_mutex(int dev);
int visc_tables(ViscTables& vt);

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes)
    {
        o.p = nullptr;
        o.bytes = 0;
    }
    ~DevBuf() { pool_free(p); }
    // a buffer that is already held is returned first: alloc() may be called again (a larger request,
    // or once per block) without leaking the slot it held
    hipError_t alloc(size_t n)
    {
        if (p) {
            pool_free(p);
            p = nullptr;
            bytes = 0;
        }
        hipError_t e = pool_alloc(&p, n ? n : 8);
        if (e == hipSuccess) bytes = n ? n : 8; else p = nullptr;
        return e;
    }
    // keeps the buffer when it is large enough already
    hipError_t reserve(size_t n) { return (p && bytes >= n) ? hipSuccess : alloc(n); }
    template <class T>
    T* as() const
    {
        return (T*)p;
    }
};

inline bool isnan_d(double x) { return x != x; }

// the kernels address a launch's cells with 32-bit byte offsets from uniform row bases
constexpr size_t kMaxCellsPerLaunch = (1ull << 29) - 1;

// phases of the spin-up (see k_spin_phase): all but the last finish all but a few cells per
// thousand, the last is the long tail
constexpr int kSpinPhases = 8;
size_t spin_work_layout(size_t ncell, size_t* o_wl_b, size_t* o_meta, size_t* o_ai, size_t* o_cnt,
                        size_t* o_cache = nullptr, size_t* o_pend = nullptr);
// the host part of splash_calendar_create (orbit per day, month bookkeeping, the spin-up's own 365 entries): no device
int calendar_host(int nt, const int* doy, const int* year, const int* month, splash_calendar* cal, char* err, size_t errlen);
void orbit_of_day(int doy, int year, double& dr, double& sd, double& cd);
int spin_phases(const splash_calendar* cal, size_t ncell, const double* sw_in, const double* tc,
                const double* pn, size_t ldf, const double* Tt, double* cellpar, size_t ldp,
                double* state, size_t lds, int* passes, void* work, void* stream, int ph0, int ph1);

// ---- SPLASH_BACKEND_HOST (splash_host_backend.hip): splash.point for ONE cell on the calling CPU thread ----
struct HostCellIn {
    int nt;
    const DayRec* days;       // [nt] calendar table (splash_calendar::host)
    const DayRec* spin_days;  // [365] spin-up table (splash_calendar::spin_host); unused when state0 is given
    const double *sw, *tc, *pn;       // the cell's series, element t at [t * stride]; or (sw == NULL) ...
    const float *sw32, *tc32, *pn32;  // ... the single-precision transport
    size_t stride;
    double lat, elev, slop, asp, resolution, soil[6], au[3];
    bool have_au;
    const double* tt;      // optional: the whole series' snowfall threshold (time chunks), else computed
    const double* state0;  // optional [5] + ai0: continuing a run (no spin-up)
    const double* ai0;
    int monthly, want_passes;
};
struct HostCellOut {
    double* out[SPLASH_NOUT];     // series k at [t_out * ldo], NULL to skip ...
    float* out32[SPLASH_NOUT];    // ... or single precision
    size_t ldo;
    long long* diag_rec;          // the exact limbs of the cell's basin, [nmonth][SPLASH_NDIAG+1][2], or NULL
    double state_end[5], tt, ai;
    int passes1, passes2;
};
int host_cell_run(const HostCellIn& in, HostCellOut& out);
bool host_backend_fma();
void host_mod_spin_up(const double* par, int n, const DayRec* days, const double* sw, const double* tc, const double* pn,
                      const double* snowfall, double* const series[7], int* passes);
void host_snow_threshold(size_t ncell, int nt, const double* tc, const float* tc32, const double* lat, const double* elev,
                         double* tt);
void host_soil_hydro(size_t ncell, const double* lat, const double* elev, const double* slop, const double* asp,
                     const double* resolution, const double* soil, const double* au, double* cellpar);
void host_unswc(size_t ncell, int nt, const double* wn, const double* soil, double uns_depth, double* wn_z, double* wtd,
                double* theta);
void host_unswc_grid(size_t ncell, int nt, const double* wn, const double* soil, double uns_depth, double* w_z,
                     double* wtd, double* se, double* theta_mean);
double host_mod_moist_surf(double depth, double z, double bub_p, double wn, double SAT, double RES, double lambda);
double host_mod_inf_GA(double bub_press, double theta_i, double Ksat, double theta_s, double lambda, double P, double tdur,
                       double slop);
void host_mod_run_all(const double* par, int nt, const DayRec* days, const double* sw, const double* tc, const double* pn,
                      const double* snowfall, const double* st0, double* const dst[11]);

// No C++ exception leaves the C ABI: the callers are C (R's .Call shim, cgo, ctypes), where an exception in flight
// is a terminate() without a message.  Every public entry point that allocates host memory or starts threads runs
// its body under c_guard(); thread bodies that do more than copy run under their own (a throw on a std::thread is
// terminate() as well) and hand the code to their owner.
int exception_code(char* err, size_t errlen, const char* where) noexcept; // call inside a catch (...) block
template <class F>
int c_guard(char* err, size_t errlen, const char* where, F&& body) noexcept
{
    try {
        return body();
    } catch (...) {
        return exception_code(err, errlen, where);
    }
}

// Host threads that are ALWAYS joined.  A std::thread that is destroyed while joinable terminates the process, and
// that is what a plain vector of them does when starting the k-th thread throws (EAGAIN: the container's thread
// limit): here the threads already running finish their share, the destructor joins them, and the exception travels
// on to the entry point's c_guard().
class Threads {
    std::vector<std::thread> pool_;

public:
    Threads() = default;
    Threads(const Threads&) = delete;
    Threads& operator=(const Threads&) = delete;
    ~Threads() { join(); }
    template <class F, class... A>
    void add(F&& fn, A&&... args)
    {
        pool_.emplace_back(std::forward<F>(fn), std::forward<A>(args)...);
    }
    void join() noexcept
    {
        for (auto& t : pool_)
            if (t.joinable()) t.join();
        pool_.clear();
    }
};

} // namespace host
} // namespace splash

struct splash_calendar {
    int nt = 0, nmonth = 0, device = 0;
    std::vector<splash::DayRec> host;
    splash::DayRec* dev = nullptr;
    // spin-up table (nt >= 365 only): the reference spins up on quick_run(i+1, y[1], ...) --
    // day-of-year 1..365 of the FIRST year whatever date the series starts on (src/SPLASH.cpp:1658,
    // 1674; R/splash.point.R:141-152) -- while the rain/snow split of those 365 records keeps the
    // months of their real dates (R/splash.point.R:124-128 runs before the spin-up)
    std::vector<splash::DayRec> spin_host;
    splash::DayRec* spin_dev = nullptr;
};

#define HIP_TRY(expr, err, errlen)                                                          \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess)                                                              \
            return fail(e__ == hipErrorOutOfMemory ? SPLASH_ENOMEM                          \
                        : (e__ == hipErrorNoDevice || e__ == hipErrorInvalidDevice)         \
                            ? SPLASH_ENODEV                                                 \
                            : SPLASH_EHIP,                                                  \
                        err, errlen, "%s: %s", #expr, hipGetErrorString(e__));              \
    } while (0)

#define REQ(cond, msg)                                                   \
    do {                                                                 \
        if (!(cond)) return fail(SPLASH_EINVAL, nullptr, 0, "%s", msg);  \
    } while (0)
