// splash_capi.hip -- the C ABI of librsplash_hip.so (include/splash_hip.h) and the host-side
// pipeline that stands where R/splash.point.R:92-214 and R/splash.grid.R:277-308 stand in the
// reference: calendar tables, stage sequencing, NA-cell masking, blocking over HBM capacity.
// There is no CPU compute path in this file: every entry point that produces results launches the
// HIP kernels, and fails with SPLASH_ENODEV when no device is present.
#include "../../include/splash_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "splash_kernels.hpp"

using namespace splash;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

static int fail(int code, char* err, size_t errlen, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    if (err && errlen) {
        snprintf(err, errlen, "%s", buf);
    }
    return code;
}

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

extern "C" int splash_hip_abi_version(void) { return SPLASH_ABI_VERSION; }

extern "C" int splash_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* splash_hip_last_error(void) { return g_err.c_str(); }

static int need_device(char* err, size_t errlen)
{
    if (splash_hip_device_count() <= 0)
        return fail(SPLASH_ENODEV, err, errlen,
                    "no HIP device available: librsplash_hip has no CPU fallback");
    return 0;
}

// ------------------------------------------------------------------ per-device context
// Lookup table of the float viscosity routine (7.5 MB per device, built on first use by running the
// routine itself for every float of its argument range; never freed).
#include <mutex>
namespace {
struct PoolSlot {
    void* p;
    size_t bytes;
    bool busy;
};
struct DevCtx {
    float* rtab = nullptr;
    // Device buffers of the host-buffer entry points are kept between calls: releasing and
    // re-acquiring ~100 GB costs 4 s per call on this runtime (measured: tools/bench_host_path.py),
    // more than the call's computation.  splash_hip_release_cache() returns them.
    std::vector<PoolSlot> pool;
};
DevCtx g_ctx[64];
// one lock per device: pool bookkeeping, the sizing + allocation phase of a grid worker (so that a
// second job on the same device plans with what the first has left), the viscosity table
std::recursive_mutex g_dev_mu[64];
std::recursive_mutex& device_alloc_mutex(int dev) { return g_dev_mu[(dev >= 0 && dev < 64) ? dev : 0]; }

void pool_release_idle(DevCtx& c)
{
    for (auto it = c.pool.begin(); it != c.pool.end();) {
        if (!it->busy) {
            (void)hipFree(it->p);
            it = c.pool.erase(it);
        } else {
            ++it;
        }
    }
}

// bytes of device `dev` held by idle pooled buffers: free memory as far as a new call is concerned
// (hipMemGetInfo does not count them; a call that sized its blocks from the free figure alone shrank
// with every call after the first multi-block one)
size_t pool_idle_bytes(int dev)
{
    if (dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::recursive_mutex> lk(g_dev_mu[dev]);
    size_t n = 0;
    for (auto& s : g_ctx[dev].pool)
        if (!s.busy) n += s.bytes;
    return n;
}

// all bytes of device `dev` the pool holds, in use or idle
size_t pool_held_bytes(int dev)
{
    if (dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::recursive_mutex> lk(g_dev_mu[dev]);
    size_t n = 0;
    for (auto& s : g_ctx[dev].pool) n += s.bytes;
    return n;
}

hipError_t pool_alloc(void** out, size_t bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || getenv("SPLASH_HIP_NO_POOL")) return hipMalloc(out, bytes);
    std::lock_guard<std::recursive_mutex> lk(g_dev_mu[dev]);
    DevCtx& c = g_ctx[dev];
    PoolSlot* best = nullptr;
    for (auto& s : c.pool)
        if (!s.busy && s.bytes >= bytes && s.bytes <= bytes + bytes / 4 + 4096 &&
            (!best || s.bytes < best->bytes))
            best = &s;
    if (best) {
        best->busy = true;
        *out = best->p;
        return hipSuccess;
    }
    // no slot of that size: idle slots of other sizes are released first when the new buffer would
    // not fit beside them (they were counted as free when the caller sized its blocks)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes + ((size_t)256 << 20) > free_b) pool_release_idle(c);
    e = hipMalloc(out, bytes);
    if (e != hipSuccess) { // make room and try once more
        (void)hipGetLastError();
        pool_release_idle(c);
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) c.pool.push_back(PoolSlot{*out, bytes, true});
    return e;
}

void pool_free(void* p)
{
    if (!p) return;
    for (int d = 0; d < 64; d++) {
        std::lock_guard<std::recursive_mutex> lk(g_dev_mu[d]);
        for (auto& s : g_ctx[d].pool)
            if (s.p == p) {
                s.busy = false;
                return;
            }
    }
    (void)hipFree(p); // not pooled (SPLASH_HIP_NO_POOL)
}

int visc_tables(ViscTables& vt)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return fail(SPLASH_ENODEV, nullptr, 0, "no current HIP device");
    std::lock_guard<std::recursive_mutex> lk(g_dev_mu[dev]);
    DevCtx& c = g_ctx[dev];
    if (!c.rtab) {
        float* r = nullptr;
        hipError_t e = hipMalloc((void**)&r, visc_table_floats_r() * sizeof(float));
        if (e == hipSuccess) e = launch_visc_tables(r, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            if (r) (void)hipFree(r);
            return fail(e == hipErrorOutOfMemory ? SPLASH_ENOMEM : SPLASH_EHIP, nullptr, 0,
                        "viscosity table: %s", hipGetErrorString(e));
        }
        c.rtab = r;
    }
    vt.rtab = c.rtab;
    return SPLASH_OK;
}
} // namespace

extern "C" void splash_hip_release_cache(void)
{
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < 64; d++) {
        std::lock_guard<std::recursive_mutex> lk(g_dev_mu[d]);
        if (g_ctx[d].pool.empty()) continue;
        if (hipSetDevice(d) != hipSuccess) continue;
        pool_release_idle(g_ctx[d]);
    }
    (void)hipSetDevice(cur);
}

// ------------------------------------------------------------------ calendar
// Orbit of one calendar day: the day-static part of SOLAR::calculate_daily_fluxes
// (src/SOLAR.cpp:98-124 with berger_tls :291-350 and julian_day :352-374), evaluated on the host
// with libm once per day instead of once per cell-day.
namespace {

const double PI_ = 3.141592653589793;
const double PIR = PI_ / 180.0;
const double K_E = 0.0167, K_EPS = 23.44, K_OMEGA = 283.0; // Berger (1978), src/global.cpp:80-82

// kept out of line so sin/cos pairs are evaluated by sin() and cos(), like the reference
__attribute__((noinline)) double sind(double x) { return std::sin(x * PIR); }
__attribute__((noinline)) double cosd(double x) { return std::cos(x * PIR); }

int julian_day(int y, int m, int i)
{
    if (m <= 2) {
        y -= 1;
        m += 12;
    }
    int a = y / 100;
    int b = 2 - a + a / 4;
    float jd = (int)(365.25 * (y + 4716)) + (int)(30.6001 * (m + 1)) + i + b - 1524.5; // float, :371
    return (int)jd;
}

void orbit_day(int n, int y, double& dr, double& sd, double& cd)
{
    volatile double e = K_E; // keep libm calls at run time (the reference's constants are extern)
    const int kN = (y == 0) ? 365 : julian_day(y + 1, 1, 1) - julian_day(y, 1, 1);
    const double xee = std::pow(e, 2.0), xec = std::pow(e, 3.0);
    const double xse = std::sqrt(1.0 - xee);
    double xlam = (e / 2.0 + xec / 8.0) * (1.0 + xse) * sind(K_OMEGA);
    xlam -= xee / 4.0 * (0.5 + xse) * sind(2.0 * K_OMEGA);
    xlam += xec / 8.0 * (1.0 / 3.0 + xse) * sind(3.0 * K_OMEGA);
    xlam *= 2.0;
    xlam /= PIR;
    const double dlamm = xlam + (n - 80.0) * (360.0 / kN);
    const double ranm = (dlamm - K_OMEGA) * PIR;
    double ranv = ranm;
    ranv += (2.0 * e - xec / 4.0) * std::sin(ranm);
    ranv += 5.0 / 4.0 * xee * std::sin(2.0 * ranm);
    ranv += 13.0 / 12.0 * xec * std::sin(3.0 * ranm);
    const double anv = ranv / PIR;
    double tls = anv + K_OMEGA;
    if (tls < 0) {
        tls += 360.0;
    } else if (tls > 360) {
        tls -= 360.0;
    }
    double nu = tls - K_OMEGA;
    if (nu < 0) nu += 360.0;
    const double rho = (1.0 - xee) / (1.0 + cosd(nu) * e);
    const double inv = 1.0 / rho; // pow(rho,-1.0)
    dr = inv * inv;               // pow(dr, 2.0)
    volatile double eps = K_EPS;
    double delta = std::asin(sind(tls) * sind(eps)) / PIR;
    sd = sind(delta);
    cd = cosd(delta);
}

} // namespace

// host-only: the day-static orbit terms of one calendar day (no device needed)
extern "C" int splash_orbit_day(int doy, int year, double* dr, double* sd, double* cd)
{
    if (doy < 1 || doy > 366 || !dr || !sd || !cd)
        return fail(SPLASH_EINVAL, nullptr, 0, "orbit_day: bad argument");
    orbit_day(doy, year, *dr, *sd, *cd);
    return SPLASH_OK;
}

struct splash_calendar {
    int nt = 0, nmonth = 0, device = 0;
    std::vector<DayRec> host;
    DayRec* dev = nullptr;
    // spin-up table (nt >= 365 only): the reference spins up on quick_run(i+1, y[1], ...) --
    // day-of-year 1..365 of the FIRST year whatever date the series starts on (src/SPLASH.cpp:1658,
    // 1674; R/splash.point.R:141-152) -- while the rain/snow split of those 365 records keeps the
    // months of their real dates (R/splash.point.R:124-128 runs before the spin-up)
    std::vector<DayRec> spin_host;
    DayRec* spin_dev = nullptr;
};

extern "C" int splash_calendar_create(int nt, const int* doy, const int* year, const int* month,
                                      splash_calendar** out, char* err, size_t errlen)
{
    if (!out || nt <= 0 || !doy || !year || !month)
        return fail(SPLASH_EINVAL, err, errlen, "calendar: nt must be > 0 and doy/year/month given");
    if (int rc = need_device(err, errlen)) return rc;
    auto* cal = new splash_calendar;
    cal->nt = nt;
    cal->host.resize(nt);
    int mid = 0;
    for (int t = 0; t < nt; t++) {
        if (month[t] < 1 || month[t] > 12 || doy[t] < 1 || doy[t] > 366) {
            delete cal;
            return fail(SPLASH_EINVAL, err, errlen, "calendar: bad doy/month at day %d", t);
        }
        if (t > 0 && (month[t] != month[t - 1] || year[t] != year[t - 1])) mid++;
        DayRec& d = cal->host[t];
        orbit_day(doy[t], year[t], d.dr, d.sd, d.cd);
        const int m = month[t];
        d.ttm_s = sind((m + 2) / 1.91);                      // R/splash.point.R:549
        d.inv_trm14 = 1.0 / (1.4 * (13.3 * (0.55 + sind(m + 4)) * 0.6)); // :550, :553
        d.month_id = mid;
        d.doy = doy[t];
        d.flags = 0;
        d.mdays = 0;
    }
    cal->nmonth = mid + 1;
    std::vector<int> cnt(cal->nmonth, 0);
    for (int t = 0; t < nt; t++) cnt[cal->host[t].month_id]++;
    for (int t = 0; t < nt; t++) {
        DayRec& d = cal->host[t];
        d.mdays = cnt[d.month_id];
        if (t == nt - 1 || cal->host[t + 1].month_id != d.month_id) d.flags |= 1;
    }
    if (nt >= 365) {
        cal->spin_host.assign(cal->host.begin(), cal->host.begin() + 365);
        for (int i = 0; i < 365; i++) {
            DayRec& d = cal->spin_host[i];
            orbit_day(i + 1, year[0], d.dr, d.sd, d.cd);
            d.doy = i + 1;
        }
    }
    (void)hipGetDevice(&cal->device);
    const size_t nspin = cal->spin_host.size();
    hipError_t e = hipMalloc((void**)&cal->dev, sizeof(DayRec) * ((size_t)nt + nspin));
    if (e == hipSuccess)
        e = hipMemcpy(cal->dev, cal->host.data(), sizeof(DayRec) * (size_t)nt, hipMemcpyHostToDevice);
    if (e == hipSuccess && nspin) {
        cal->spin_dev = cal->dev + nt;
        e = hipMemcpy(cal->spin_dev, cal->spin_host.data(), sizeof(DayRec) * nspin,
                      hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (cal->dev) (void)hipFree(cal->dev);
        delete cal;
        return fail(SPLASH_EHIP, err, errlen, "calendar upload: %s", hipGetErrorString(e));
    }
    *out = cal;
    return SPLASH_OK;
}

extern "C" void splash_calendar_destroy(splash_calendar* cal)
{
    if (!cal) return;
    if (cal->dev) (void)hipFree(cal->dev);
    delete cal;
}

extern "C" int splash_calendar_nt(const splash_calendar* cal) { return cal ? cal->nt : 0; }
extern "C" int splash_calendar_nmonth(const splash_calendar* cal) { return cal ? cal->nmonth : 0; }

extern "C" int splash_calendar_orbit(const splash_calendar* cal, double* dr, double* sd, double* cd)
{
    if (!cal) return fail(SPLASH_EINVAL, nullptr, 0, "calendar is NULL");
    for (int t = 0; t < cal->nt; t++) {
        if (dr) dr[t] = cal->host[t].dr;
        if (sd) sd[t] = cal->host[t].sd;
        if (cd) cd[t] = cal->host[t].cd;
    }
    return SPLASH_OK;
}

// ------------------------------------------------------------------ device-resident stages
// the kernels address a launch's cells with 32-bit byte offsets from uniform row bases
static const size_t kMaxCellsPerLaunch = (1ull << 29) - 1;

#define REQ(cond, msg)                                                   \
    do {                                                                 \
        if (!(cond)) return fail(SPLASH_EINVAL, nullptr, 0, "%s", msg);  \
    } while (0)

extern "C" int splash_dev_soil_hydro(size_t ncell, const double* lat, const double* elev,
                                     const double* slop, const double* asp,
                                     const double* resolution, const double* soil, const double* au,
                                     double* cellpar, void* stream)
{
    REQ(lat && elev && slop && asp && resolution && soil && cellpar, "soil_hydro: NULL argument");
    if (int rc = need_device(nullptr, 0)) return rc;
    HIP_TRY(launch_soil_hydro(ncell, lat, elev, slop, asp, resolution, soil, au, cellpar,
                              (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
}

extern "C" int splash_dev_snow_threshold(size_t ncell, int nt, const double* tc, size_t ldf,
                                         const double* cellpar, size_t ldp, double* Tt, void* stream)
{
    REQ(tc && cellpar && Tt && nt > 0, "snow_threshold: NULL argument or nt <= 0");
    REQ(ldf >= ncell && ldp >= ncell, "snow_threshold: leading dimension smaller than ncell");
    if (int rc = need_device(nullptr, 0)) return rc;
    HIP_TRY(launch_snow_threshold(ncell, nt, tc, ldf, cellpar, ldp, Tt, (hipStream_t)stream),
            nullptr, 0);
    return SPLASH_OK;
}

// workspace of splash_dev_spin_up: two worklists, job records, aridity index, phase counters, and
// the cache of state-independent day terms: 365 x 19 doubles per slot, 2^19 slots at most (29 GB).
// Phase 0 walks the cells in blocks of that many; the later phases' worklists are far shorter.
static size_t spin_cache_cap(size_t ncell)
{
    size_t lim = (size_t)1 << 19;
    // test / tuning aid: a small capacity makes small grids walk phase 0 in several blocks and gives
    // the later phases jobs beyond the cache (read once: the workspace layout depends on it)
    static const char* env = getenv("SPLASH_HIP_SPIN_CACHE_CAP");
    if (env && atol(env) >= 64) lim = (size_t)atol(env);
    size_t cap = std::min<size_t>(ncell, lim);
    return (cap + 63) & ~(size_t)63;
}

static size_t spin_work_layout(size_t ncell, size_t* o_wl_b, size_t* o_meta, size_t* o_ai,
                               size_t* o_cnt, size_t* o_cache = nullptr, size_t* o_pend = nullptr)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    take(ncell * 4); // wl_a at 0
    size_t wl_b = take(ncell * 4), meta = take(ncell * 4), ai = take(ncell * 8), cnt = take(64);
    size_t pend = take(ncell);
    size_t cache = take(spin_cache_cap(ncell) * spin_cache_bytes_per_slot());
    if (o_pend) *o_pend = pend;
    if (o_wl_b) *o_wl_b = wl_b;
    if (o_meta) *o_meta = meta;
    if (o_ai) *o_ai = ai;
    if (o_cnt) *o_cnt = cnt;
    if (o_cache) *o_cache = cache;
    return off;
}

extern "C" size_t splash_dev_spin_up_work_size(size_t ncell)
{
    return spin_work_layout(ncell, nullptr, nullptr, nullptr, nullptr);
}

// phases [ph0, ph1) of the spin-up (see k_spin_phase): all but the last finish all but a few cells
// per thousand, the last is the long tail
constexpr int kSpinPhases = 8;
static int spin_phases(const splash_calendar* cal, size_t ncell, const double* sw_in,
                       const double* tc, const double* pn, size_t ldf, const double* Tt,
                       double* cellpar, size_t ldp, double* state, size_t lds, int* passes,
                       void* work, void* stream, int ph0, int ph1)
{
    REQ(cal && sw_in && tc && pn && Tt && cellpar && state && work, "spin_up: NULL argument");
    REQ(cal->nt >= 365, "spin_up: needs at least 365 days (R/splash.point.R:141-144)");
    REQ(ldf >= ncell && ldp >= ncell && lds >= ncell, "spin_up: leading dimension < ncell");
    REQ(ncell <= kMaxCellsPerLaunch, "spin_up: more than 2^29-1 cells in one call (32-bit lane byte offsets)");
    if (int rc = need_device(nullptr, 0)) return rc;
    if (ncell == 0) return SPLASH_OK;
    hipStream_t s = (hipStream_t)stream;
    size_t o_wl_b, o_meta, o_ai, o_cnt, o_cache, o_pend;
    spin_work_layout(ncell, &o_wl_b, &o_meta, &o_ai, &o_cnt, &o_cache, &o_pend);
    char* w = (char*)work;
    uint32_t* wl[2] = {(uint32_t*)w, (uint32_t*)(w + o_wl_b)};
    uint32_t* cnt = (uint32_t*)(w + o_cnt);
    if (ph0 == 0) {
        HIP_TRY(hipMemsetAsync(cnt, 0, 64, s), nullptr, 0);
        HIP_TRY(hipMemsetAsync(w + o_pend, 0, ncell, s), nullptr, 0);
    }
    SpinPhaseArgs a{};
    a.days = cal->spin_dev; // doy 1..365 of the first year, real months (see splash_calendar)
    a.n = 365;
    a.ncell = ncell;
    a.sw = sw_in;
    a.tc = tc;
    a.pn = pn;
    a.ldf = ldf;
    a.Tt = Tt;
    a.par = cellpar;
    a.ldp = ldp;
    a.state = state;
    a.lds = lds;
    a.passes = passes;
    a.meta = (int*)(w + o_meta);
    a.ai = (double*)(w + o_ai);
    if (int rc = visc_tables(a.vt)) return rc;
    // pass budgets per phase; the last one covers the 1000-pass worst case of a spin-up
    // (src/SPLASH.cpp:1697).  With `passes` the first spin-up is iterated to its end as well (only
    // to count its passes), otherwise it is a single pass: see k_spin_phase.
    // A wavefront runs as many passes as its slowest lane, so the early phases are short and the
    // worklist is compacted after each (synthetic grid: 20-50 % of the cells need 2 passes of
    // spin-up #2, 35-60 % 3, 6-20 % 4, 2-8 % 5, 3-12 % more): phase 0 = the pass of spin-up #1 + 2
    // passes of #2 for every cell, then 1, 1, 1, 2, 4 and 16 passes for those that are left.
    static const int budget_full[kSpinPhases] = {8, 4, 4, 8, 16, 32, 64, 2100},
                     budget_fast[kSpinPhases] = {3, 1, 1, 1, 2, 4, 16, 1100};
    const int* budget = passes ? budget_full : budget_fast;
    double* const cache_all = (double*)(w + o_cache);
    a.cache_cap = spin_cache_cap(ncell);
    a.ntotal = ncell;
    for (int ph = ph0; ph < ph1; ph++) {
        a.fresh = (ph == 0);
        a.budget = budget[ph];
        a.wl_in = ph == 0 ? nullptr : wl[(ph - 1) & 1];
        a.n_in = ph == 0 ? nullptr : cnt + (ph - 1);
        a.wl_out = wl[ph & 1];
        a.n_out = cnt + ph;
        a.pending = (ph == kSpinPhases - 2) ? (unsigned char*)(w + o_pend) : nullptr;
        // the cache of day terms pays from the second pass of a launch on; phase 0 streams compact
        // records, the short worklists of the later phases keep rich ones (splash_kernels.hip)
        a.cache = budget[ph] >= 2 ? cache_all : nullptr;
        a.rich = ph > 0 ? 1 : 0;
        a.cache_cap = ph > 0 ? spin_cache_rich_cap(spin_cache_cap(ncell)) : spin_cache_cap(ncell);
        if (ph == 0) { // every cell, in blocks the size of the cache
            // per block: one pass of the general kernel (spin-up #1's pass: it stores the block's
            // records), then the other passes of the phase by the read-back kernel, which is the
            // smaller program (see k_spin_phase<true, true>) and does the queueing
            const bool split = a.cache != nullptr && budget[ph] >= 2;
            for (size_t c0 = 0; c0 < ncell; c0 += a.cache_cap) {
                a.cell0 = c0;
                a.ncell = std::min(a.cache_cap, ncell - c0);
                if (split) {
                    a.budget = 1;
                    a.no_queue = 1;
                    HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
                    a.budget = budget[ph] - 1;
                    a.no_queue = 0;
                    a.fresh = 0;
                    a.readback = 1;
                    HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
                    a.readback = 0;
                    a.fresh = 1;
                } else {
                    HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
                }
            }
            a.cell0 = 0;
            a.ncell = ncell;
        } else if (a.cache && budget[ph] > 1) {
            // A phase of several passes as three launches over the SAME worklist (launch positions =
            // cache slots): one pass of the general kernel stores every job's 365 records (it parks
            // the jobs without queueing them); the read-back kernel -- a far smaller program: see
            // k_spin_phase<true, true> -- runs the other passes of the jobs that have a slot and
            // queues what is left; the general kernel does the same for the jobs beyond the cache's
            // capacity, if any (a list longer than 2^19 jobs: grid blocks past the list's end return
            // at once).
            const size_t n_all = a.ncell;
            a.budget = 1;
            a.no_queue = 1;
            HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
            a.no_queue = 0;
            a.budget = budget[ph] - 1;
            a.readback = 1;
            a.ncell = std::min(n_all, a.cache_cap);
            HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
            a.readback = 0;
            if (n_all > a.cache_cap) {
                a.slot0 = a.cache_cap;
                a.ncell = n_all - a.cache_cap;
                double* const cache = a.cache;
                a.cache = nullptr;
                HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
                a.cache = cache;
                a.slot0 = 0;
            }
            a.ncell = n_all;
        } else {
            HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
        }
    }
    return SPLASH_OK;
}

extern "C" int splash_dev_spin_up(const splash_calendar* cal, size_t ncell, const double* sw_in,
                                  const double* tc, const double* pn, size_t ldf, const double* Tt,
                                  double* cellpar, size_t ldp, double* state, size_t lds,
                                  int* passes, void* work, void* stream)
{
    return spin_phases(cal, ncell, sw_in, tc, pn, ldf, Tt, cellpar, ldp, state, lds, passes, work,
                       stream, 0, kSpinPhases);
}

extern "C" int splash_dev_spin_up_begin(const splash_calendar* cal, size_t ncell,
                                        const double* sw_in, const double* tc, const double* pn,
                                        size_t ldf, const double* Tt, double* cellpar, size_t ldp,
                                        double* state, size_t lds, int* passes, void* work,
                                        void* stream)
{
    return spin_phases(cal, ncell, sw_in, tc, pn, ldf, Tt, cellpar, ldp, state, lds, passes, work,
                       stream, 0, kSpinPhases - 1);
}

extern "C" int splash_dev_spin_up_finish(const splash_calendar* cal, size_t ncell,
                                         const double* sw_in, const double* tc, const double* pn,
                                         size_t ldf, const double* Tt, double* cellpar, size_t ldp,
                                         double* state, size_t lds, int* passes, void* work,
                                         void* stream)
{
    return spin_phases(cal, ncell, sw_in, tc, pn, ldf, Tt, cellpar, ldp, state, lds, passes, work,
                       stream, kSpinPhases - 1, kSpinPhases);
}

extern "C" int splash_dev_spin_up_pending(void* work, size_t ncell, const unsigned char** mask,
                                          const uint32_t** cells, const uint32_t** n_cells)
{
    REQ(work, "spin_up_pending: NULL workspace");
    size_t o_wl_b, o_meta, o_ai, o_cnt, o_cache, o_pend;
    spin_work_layout(ncell, &o_wl_b, &o_meta, &o_ai, &o_cnt, &o_cache, &o_pend);
    char* w = (char*)work;
    if (mask) *mask = (const unsigned char*)(w + o_pend);
    static_assert(((kSpinPhases - 2) & 1) == 0, "the phase before the tail writes worklist 0");
    if (cells) *cells = (const uint32_t*)w;
    if (n_cells) *n_cells = (const uint32_t*)(w + o_cnt) + (kSpinPhases - 2);
    return SPLASH_OK;
}

extern "C" int splash_dev_spin_up_pending_count(void* work, size_t ncell, void* stream,
                                                uint32_t* n_pending)
{
    REQ(work && n_pending, "spin_up_pending_count: NULL argument");
    const uint32_t* d_n = nullptr;
    (void)splash_dev_spin_up_pending(work, ncell, nullptr, nullptr, &d_n);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream), nullptr, 0);
    HIP_TRY(hipMemcpy(n_pending, d_n, sizeof(uint32_t), hipMemcpyDeviceToHost), nullptr, 0);
    return SPLASH_OK;
}

extern "C" size_t splash_dev_diag_acc_size(int n_basin, int nmonth)
{
    return (size_t)std::max(n_basin, 1) * (size_t)std::max(nmonth, 0) * SPLASH_DIAG_REC * sizeof(int64_t);
}

extern "C" int splash_dev_diag_finalize(const int64_t* diag_acc, int n_basin, int nmonth,
                                        double* diag, double* counts, void* stream)
{
    REQ(diag_acc && diag && nmonth >= 0, "diag_finalize: NULL argument");
    if (int rc = need_device(nullptr, 0)) return rc;
    HIP_TRY(launch_diag_finalize((const long long*)diag_acc, std::max(n_basin, 1) * nmonth, diag,
                                 counts, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
}

// the same conversion on the host: value = limb0/2^16 + limb1/2^48 (exact up to the final rounding)
extern "C" void splash_diag_finalize_host(const int64_t* acc, int n_basin, int nmonth, double* diag,
                                          double* counts)
{
    const size_t n_rec = (size_t)std::max(n_basin, 1) * (size_t)std::max(nmonth, 0);
    for (size_t i = 0; i < n_rec; i++) {
        const int64_t* r = acc + i * SPLASH_DIAG_REC;
        for (int q = 0; q < SPLASH_NDIAG; q++) {
            const double hi = (double)r[2 * q];
            const double lo = (double)(r[2 * q] - (int64_t)hi);
            diag[i * SPLASH_NDIAG + q] = hi * (1.0 / 65536.0) +
                                         (lo * (1.0 / 65536.0) + (double)r[2 * q + 1] * (1.0 / 281474976710656.0));
        }
        if (counts) {
            counts[2 * i] = (double)r[2 * SPLASH_NDIAG];
            counts[2 * i + 1] = (double)r[2 * SPLASH_NDIAG + 1];
        }
    }
}

extern "C" int splash_dev_run_all(const splash_calendar* cal, int t0, int nt, size_t ncell,
                                  const double* sw_in, const double* tc, const double* pn,
                                  size_t ldf, const double* Tt, const double* cellpar, size_t ldp,
                                  double* state, size_t lds, const splash_dev_out* out, void* stream)
{
    REQ(cal && sw_in && tc && pn && Tt && cellpar && state && out, "run_all: NULL argument");
    REQ(t0 >= 0 && nt >= 0 && t0 + nt <= cal->nt, "run_all: day range outside the calendar");
    REQ(ldf >= ncell && ldp >= ncell && lds >= ncell && out->ldo >= ncell,
        "run_all: leading dimension < ncell");
    REQ(!out->basin || out->diag_acc, "run_all: basin ids given without diag_acc");
    REQ(!out->basin || out->n_basin > 0, "run_all: basin ids given with n_basin <= 0");
    REQ(!(out->monthly && (out->extra[0] || out->extra[1] || out->extra[2])),
        "run_all: the qin_prev/tdrain/snwage series are daily (monthly = 0)");
    REQ(ncell <= kMaxCellsPerLaunch, "run_all: more than 2^29-1 cells in one call (32-bit lane byte offsets)");
    if (int rc = need_device(nullptr, 0)) return rc;
    if (nt == 0 || ncell == 0) return SPLASH_OK;
    RunArgs a{};
    a.days = cal->dev + t0;
    a.nt = nt;
    a.ncell = ncell;
    a.sw = sw_in;
    a.tc = tc;
    a.pn = pn;
    a.snowfall = nullptr;
    a.ldf = ldf;
    a.Tt = Tt;
    a.par = cellpar;
    a.ldp = ldp;
    a.state = state;
    a.lds = lds;
    for (int k = 0; k < SPLASH_NOUT; k++) a.out[k] = out->out[k];
    for (int k = 0; k < SPLASH_NEXTRA; k++) a.extra[k] = out->extra[k];
    a.ldo = out->ldo;
    a.monthly = out->monthly;
    a.month0 = cal->host[t0].month_id;
    a.head_partial = (t0 > 0 && cal->host[t0 - 1].month_id == a.month0) ? 1 : 0;
    a.nmonth = cal->host[t0 + nt - 1].month_id - a.month0 + 1;
    a.diag_acc = (long long*)out->diag_acc;
    a.basin = out->basin;
    a.n_basin = out->basin ? out->n_basin : 1;
    a.nmonth_cal = cal->nmonth;
    a.cells = out->cells;
    a.skip = out->skip;
    if (int rc = visc_tables(a.vt)) return rc;
    HIP_TRY(launch_run_all(a, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
}

extern "C" int splash_dev_synth_statics(uint64_t seed, int H, int W, int64_t cell0, size_t ncell,
                                        int terrain, double* lat, double* elev, double* slop,
                                        double* asp, double* resolution, double* soil, double* au,
                                        void* stream)
{
    REQ(lat && elev && slop && asp && resolution && soil && au && H > 0 && W > 0,
        "synth_statics: NULL argument");
    if (int rc = need_device(nullptr, 0)) return rc;
    HIP_TRY(launch_synth_statics(seed, H, W, cell0, ncell, terrain, lat, elev, slop, asp,
                                 resolution, soil, au, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
}

extern "C" int splash_dev_synth_forcing(uint64_t seed, const splash_calendar* cal, int t0, int nt,
                                        int64_t cell0, size_t ncell, const double* lat,
                                        const double* elev, double* sw_in, double* tc, double* pn,
                                        size_t ldf, void* stream)
{
    REQ(cal && lat && elev && sw_in && tc && pn, "synth_forcing: NULL argument");
    REQ(t0 >= 0 && nt >= 0 && t0 + nt <= cal->nt, "synth_forcing: day range outside the calendar");
    REQ(ldf >= ncell, "synth_forcing: leading dimension < ncell");
    if (int rc = need_device(nullptr, 0)) return rc;
    HIP_TRY(launch_synth_forcing(seed, cal->dev + t0, t0, nt, cell0, ncell, lat, elev, sw_in, tc,
                                 pn, ldf, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
}

// static rasters before the loop: R/splash.grid.R:96-110, R/upslope_area.R:31-33
extern "C" int splash_dev_terrain(int H, int W, const double* elev, double ymax, double yres,
                                  double xres, int na_to_zero, double* slope_deg, double* aspect_deg,
                                  double* lat, double* resolution_m, double* flowdir, void* stream)
{
    REQ(elev && H > 0 && W > 0, "terrain: NULL elevation or empty grid");
    REQ(yres > 0 && xres > 0, "terrain: cell size must be positive (degrees)");
    REQ(slope_deg || aspect_deg || lat || resolution_m || flowdir, "terrain: no output requested");
    if (int rc = need_device(nullptr, 0)) return rc;
    HIP_TRY(launch_terrain(H, W, elev, ymax, yres, xres, na_to_zero, slope_deg, aspect_deg, lat,
                           resolution_m, flowdir, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
}

extern "C" int splash_dev_ncellflow(int H, int W, const double* flowdir, const double* codes9,
                                    int pad, double* ncell, void* stream)
{
    REQ(flowdir && codes9 && ncell && H > 0 && W > 0, "ncellflow: NULL argument or empty grid");
    if (int rc = need_device(nullptr, 0)) return rc;
    HIP_TRY(launch_ncellflow(H, W, flowdir, codes9, pad, ncell, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
}

// sink fill and contributing area (the rest of R/upslope_area.R: SURVEY 8f-1)
extern "C" int splash_dev_sinkfill(int H, int W, const double* elev, double resolution_m,
                                   double degree, double* filled, int* sweeps, void* stream)
{
    REQ(elev && filled && H > 0 && W > 0, "sinkfill: NULL argument or empty grid");
    REQ(elev != filled, "sinkfill: the filled surface needs its own raster");
    REQ(resolution_m > 0 && degree >= 0 && degree < 90, "sinkfill: resolution must be positive, 0 <= degree < 90");
    if (int rc = need_device(nullptr, 0)) return rc;
    HIP_TRY(launch_sinkfill(H, W, elev, resolution_m, degree, filled, sweeps, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
}

extern "C" int splash_dev_flow_accumulation(int H, int W, const double* flowdir, const double* elev,
                                            int tau_codes, const double* resolution_m,
                                            double* ups_ncell, double* ups_area, int* levels,
                                            void* stream)
{
    REQ(flowdir && H > 0 && W > 0, "flow_accumulation: NULL flow direction or empty grid");
    REQ(ups_ncell || ups_area, "flow_accumulation: no output requested");
    REQ(!ups_area || resolution_m, "flow_accumulation: the area needs the resolution raster");
    REQ((size_t)H * W < ((size_t)1 << 32), "flow_accumulation: more than 2^32-1 cells");
    if (int rc = need_device(nullptr, 0)) return rc;
    HIP_TRY(launch_flow_accumulation(H, W, flowdir, elev, tau_codes, resolution_m, ups_ncell, ups_area, levels,
                                     (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
}

extern "C" int splash_dev_mfd_area(int H, int W, const double* elev, double resolution_m,
                                   double* area, int* levels, void* stream)
{
    REQ(elev && area && H > 0 && W > 0, "mfd_area: NULL argument or empty grid");
    REQ(resolution_m > 0, "mfd_area: resolution must be positive");
    REQ((size_t)H * W < ((size_t)1 << 32), "mfd_area: more than 2^32-1 cells");
    if (int rc = need_device(nullptr, 0)) return rc;
    HIP_TRY(launch_mfd_area(H, W, elev, resolution_m, area, levels, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
}

// unSWC epilogue on device-resident rasters (R/unsSWC.R:14-88, R/unsSWC.grid.R)
extern "C" int splash_dev_unswc(size_t ncell, int nt, const double* wn, size_t ldw,
                                const double* soil, double uns_depth, double* wn_z, double* wtd,
                                double* theta, size_t ldo, void* stream)
{
    REQ(wn && soil && (wn_z || wtd || theta), "unswc: NULL argument");
    REQ(ldw >= ncell && ldo >= ncell && nt >= 0, "unswc: leading dimension < ncell");
    REQ(uns_depth > 0, "unswc: uns_depth must be positive");
    if (int rc = need_device(nullptr, 0)) return rc;
    HIP_TRY(launch_unswc(ncell, nt, wn, ldw, soil, uns_depth, wn_z, wtd, theta, ldo,
                         (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
}

// ------------------------------------------------------------------ small RAII helpers
namespace {

struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p) { o.p = nullptr; }
    ~DevBuf() { pool_free(p); }
    hipError_t alloc(size_t bytes) { return pool_alloc(&p, bytes ? bytes : 8); }
    template <class T>
    T* as() const
    {
        return (T*)p;
    }
};

inline bool isnan_d(double x) { return x != x; }

} // namespace

// test hook: SOLAR+EVAP rows (host pointers)
extern "C" int splash_dev_evap_batch(size_t m, const double* lat, const double* elv,
                                     const double* sw, const int* n, const int* y,
                                     const double* sw_in, const double* tc, const double* slop,
                                     const double* asp, const double* snow, const double* nd,
                                     double* out, void* stream)
{
    (void)stream;
    REQ(lat && elv && sw && n && y && sw_in && tc && slop && asp && snow && nd && out,
        "evap_batch: NULL argument");
    if (int rc = need_device(nullptr, 0)) return rc;
    if (m == 0) return SPLASH_OK;
    std::vector<double> orb(3 * m);
    for (size_t i = 0; i < m; i++) orbit_day(n[i], y[i], orb[i], orb[m + i], orb[2 * m + i]);
    const double* hin[9] = {lat, elv, sw, sw_in, tc, slop, asp, snow, nd};
    DevBuf buf[9], dorb, dout;
    for (int k = 0; k < 9; k++) {
        HIP_TRY(buf[k].alloc(m * 8), nullptr, 0);
        HIP_TRY(hipMemcpy(buf[k].p, hin[k], m * 8, hipMemcpyHostToDevice), nullptr, 0);
    }
    HIP_TRY(dorb.alloc(3 * m * 8), nullptr, 0);
    HIP_TRY(hipMemcpy(dorb.p, orb.data(), 3 * m * 8, hipMemcpyHostToDevice), nullptr, 0);
    HIP_TRY(dout.alloc(11 * m * 8), nullptr, 0);
    const double* o = dorb.as<double>();
    ViscTables vt{};
    if (int rc = visc_tables(vt)) return rc;
    HIP_TRY(launch_evap_batch(m, buf[0].as<double>(), buf[1].as<double>(), buf[2].as<double>(), o,
                              o + m, o + 2 * m, buf[3].as<double>(), buf[4].as<double>(),
                              buf[5].as<double>(), buf[6].as<double>(), buf[7].as<double>(),
                              buf[8].as<double>(), dout.as<double>(), vt, nullptr), nullptr, 0);
    HIP_TRY(hipMemcpy(out, dout.p, 11 * m * 8, hipMemcpyDeviceToHost), nullptr, 0);
    return SPLASH_OK;
}

// test hook: device flog/fexp/fpow/facos/fsqrt for n arguments (host pointers); out [5][n]
extern "C" int splash_dev_math_probe(size_t n, const double* x, const double* y, double* out)
{
    REQ(x && y && out, "math_probe: NULL argument");
    if (int rc = need_device(nullptr, 0)) return rc;
    if (n == 0) return SPLASH_OK;
    DevBuf dx, dy, dout;
    HIP_TRY(dx.alloc(n * 8), nullptr, 0);
    HIP_TRY(dy.alloc(n * 8), nullptr, 0);
    HIP_TRY(dout.alloc(7 * n * 8), nullptr, 0);
    HIP_TRY(hipMemcpy(dx.p, x, n * 8, hipMemcpyHostToDevice), nullptr, 0);
    HIP_TRY(hipMemcpy(dy.p, y, n * 8, hipMemcpyHostToDevice), nullptr, 0);
    HIP_TRY(launch_math_probe(n, dx.as<double>(), dy.as<double>(), dout.as<double>(), nullptr),
            nullptr, 0);
    HIP_TRY(hipMemcpy(out, dout.p, 7 * n * 8, hipMemcpyDeviceToHost), nullptr, 0);
    return SPLASH_OK;
}

// test hook: the day loop's own moist_surf()/inf_GA() device functions (host pointers)
extern "C" int splash_dev_infiltration_probe(size_t n, const double* cellpar, const double* wn,
                                             const double* ksat_visc, const double* P,
                                             double* theta_surf, double* infiltration)
{
    REQ(cellpar && wn && ksat_visc && P && theta_surf && infiltration,
        "infiltration_probe: NULL argument");
    if (int rc = need_device(nullptr, 0)) return rc;
    if (n == 0) return SPLASH_OK;
    DevBuf dp, dw, dk, dP, dt, di;
    const double* hin[3] = {wn, ksat_visc, P};
    DevBuf* din[3] = {&dw, &dk, &dP};
    HIP_TRY(dp.alloc(SPLASH_NPAR * n * 8), nullptr, 0);
    HIP_TRY(hipMemcpy(dp.p, cellpar, SPLASH_NPAR * n * 8, hipMemcpyHostToDevice), nullptr, 0);
    for (int k = 0; k < 3; k++) {
        HIP_TRY(din[k]->alloc(n * 8), nullptr, 0);
        HIP_TRY(hipMemcpy(din[k]->p, hin[k], n * 8, hipMemcpyHostToDevice), nullptr, 0);
    }
    HIP_TRY(dt.alloc(n * 8), nullptr, 0);
    HIP_TRY(di.alloc(n * 8), nullptr, 0);
    HIP_TRY(launch_infiltration_probe(n, dp.as<double>(), dw.as<double>(), dk.as<double>(),
                                      dP.as<double>(), dt.as<double>(), di.as<double>(), nullptr),
            nullptr, 0);
    HIP_TRY(hipMemcpy(theta_surf, dt.p, n * 8, hipMemcpyDeviceToHost), nullptr, 0);
    HIP_TRY(hipMemcpy(infiltration, di.p, n * 8, hipMemcpyDeviceToHost), nullptr, 0);
    return SPLASH_OK;
}

// first touch of the output arrays by a few host threads (one write per 4 KiB page): columns
// [c0, c0+ncols) of every row of the [nrows][N] arrays
static void prefault_outputs(double* const* bufs, int nbuf, size_t N, size_t c0, size_t ncols,
                             size_t nrows)
{
    const size_t step = 4096 / sizeof(double);
    if (ncols * nrows < (size_t)(64u << 20) / sizeof(double)) return; // small arrays: not worth threads
    unsigned nthr = std::thread::hardware_concurrency();
    nthr = nthr == 0 ? 4 : (nthr > 8 ? 8 : nthr);
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < nthr; w++)
        pool.emplace_back([=]() {
            for (int k = 0; k < nbuf; k++) {
                if (!bufs[k]) continue;
                for (size_t r = w; r < nrows; r += nthr) {
                    volatile double* p = bufs[k] + r * N + c0;
                    for (size_t i = 0; i < ncols; i += step) p[i] = 0.0;
                    if (ncols) p[ncols - 1] = 0.0;
                }
            }
        });
    for (auto& t : pool) t.join();
}

// ------------------------------------------------------------------ splash.grid / splash.point
namespace {

// What one call of splash_grid_run works on, shared by its per-device workers.
struct GridPlan {
    const splash_grid_in* in;
    const splash_grid_out* out;
    const size_t* idx; // valid (non-NA) cells, ascending
    size_t nv, N;
    bool dense;        // nv == N: no gather/scatter
    int nt, nmonth, nt_out, n_out, monthly, n_basin;
    size_t max_cells;
    bool want_diag;
    double mem_fraction; // share of the device's free memory one worker may plan with
};

inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// fn(i) for i in [0, n) on a few host threads (the gather/scatter loops of masked grids and the
// patching of late cells: single-threaded they ran at ~1 GB/s)
template <class F>
void parallel_rows(size_t n, size_t bytes_per_row, F fn)
{
    unsigned nthr = std::thread::hardware_concurrency();
    nthr = nthr == 0 ? 4 : (nthr > 8 ? 8 : nthr);
    if (n * bytes_per_row < ((size_t)8 << 20) || n < 2 * nthr) nthr = 1;
    if (nthr == 1) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < nthr; w++)
        pool.emplace_back([=]() {
            for (size_t i = w; i < n; i += nthr) fn(i);
        });
    for (auto& t : pool) t.join();
}

struct Stream {
    hipStream_t s = nullptr;
    Stream() = default;
    Stream(const Stream&) = delete;
    Stream(Stream&& o) noexcept : s(o.s) { o.s = nullptr; }
    ~Stream()
    {
        if (s) {
            (void)hipStreamSynchronize(s); // nothing of this call may still run when its buffers go back to the pool
            (void)hipStreamDestroy(s);
        }
    }
    hipError_t create() { return hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
};
struct Event {
    hipEvent_t e = nullptr;
    Event() = default;
    Event(const Event&) = delete;
    Event(Event&& o) noexcept : e(o.e) { o.e = nullptr; }
    ~Event()
    {
        if (e) (void)hipEventDestroy(e);
    }
    hipError_t create() { return hipEventCreateWithFlags(&e, hipEventDisableTiming); }
};

// The stragglers of all blocks of one worker: the jobs the first seven spin-up phases leave behind
// (a few cells per ten thousand; the reference runs them up to 1000 passes, src/SPLASH.cpp:1697) are
// copied here by k_pool_gather as soon as a block's phases are done -- forcing columns, parameters,
// parked job -- so that the block's buffers are free for the next block and the tail, a latency
// chain of over a second whatever the number of cells, runs ONCE per call instead of once per block.
struct Pool {
    size_t cap = 0;
    DevBuf force, par, state, tt, origin, pass, basin, work, out;
    double* sw() const { return force.as<double>(); }
};

// host staging without the value-initialisation of std::vector (zero-filling ~0.5 GB per call is
// 0.1 s of page faults on the calling thread; the gather threads touch the pages anyway)
template <class T>
struct HostBuf {
    std::unique_ptr<T[]> p;
    size_t n = 0;
    void resize(size_t m)
    {
        p.reset(m ? new T[m] : nullptr);
        n = m;
    }
    T* data() const { return p.get(); }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) const { return p[i]; }
};

// One lane of the pipeline: a host thread with its own stream, block buffers and spin-up workspace.
// Lanes take the blocks in turn; while one lane's thread sits in a host<->device copy the other
// lane's kernels have the GPU, so H2D of block k+1, the kernels of block k and D2H of block k-1
// overlap (the SOCK cluster of R/splash.grid.R:359-400 overlapped its workers the same way).
struct Lane {
    DevBuf d_force, d_f32, d_out, d_par, d_state, d_stat, d_tt, d_pass, d_swork, d_basin, d_res, d_late;
    HostBuf<double> stage, stage_out, hstat, hsmall;
    HostBuf<float> stage32;
    std::thread d2h;      // delivers the previous block's results while the lane loads the next one
    int d2h_rc = SPLASH_OK;
    HostBuf<int> hpass;
    HostBuf<int32_t> hbasin;
    int rc = SPLASH_OK;
    char msg[512] = "";
    Event ev_gather;
    Stream st; // declared last = destroyed first: the stream is drained before the buffers above return to the pool
};

struct Worker {
    const GridPlan& P;
    int device;
    size_t v0, v1, nb_max;
    splash_calendar* cal = nullptr;
    bool f32;
    Pool pool;
    DevBuf d_acc;
    std::vector<Lane> lanes;
    std::mutex mu;
    std::condition_variable cv;
    size_t next_block = 0;      // blocks are handed out in order, one host->device load at a time
    bool loading = false;
    size_t blocks_gathered = 0; // blocks whose early phases and gather are enqueued
    int lanes_done = 0;
    std::atomic<bool> abort{false};
    bool trace;
    // what finish_pool brings back for patch_pool
    uint32_t n_pool = 0;
    std::vector<uint64_t> origin;
    std::vector<double> patch, pstate, paux;
    std::vector<int> ppass_h;
    double t_pool[3] = {0, 0, 0};

    Worker(const GridPlan& P_, int dev, size_t a, size_t b) : P(P_), device(dev), v0(a), v1(b), nb_max(0)
    {
        f32 = (P.in->sw_in == nullptr);
        trace = getenv("SPLASH_TRACE") != nullptr;
    }
    ~Worker() { splash_calendar_destroy(cal); }

    size_t nblocks() const { return (v1 - v0 + nb_max - 1) / nb_max; }
    int lane_fail(Lane& L, int code, const char* what, hipError_t e)
    {
        snprintf(L.msg, sizeof L.msg, "%s: %s", what, e == hipSuccess ? g_err.c_str() : hipGetErrorString(e));
        L.rc = code;
        abort = true;
        return code;
    }
    void note(size_t& counter_or_null, bool lane_exit)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (lane_exit) lanes_done++; else counter_or_null++;
        cv.notify_all();
    }
    int run_lane(int li);
    int run_block(Lane& L, size_t k);
    int deliver(Lane& L, size_t b0, size_t nb, splash_dev_out dout);
    int late_fallback(Lane& L, size_t b0, size_t nb, const splash_dev_out& dout, uint32_t n_late);
    int finish_pool(char* err, size_t errlen);
    void patch_pool();
};

#define LANE_TRY(expr)                                                                         \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return lane_fail(L, e__ == hipErrorOutOfMemory ? SPLASH_ENOMEM : SPLASH_EHIP, #expr, e__); \
    } while (0)
#define LANE_RC(expr)                                                 \
    do {                                                              \
        int rc__ = (expr);                                            \
        if (rc__) return lane_fail(L, rc__, #expr, hipSuccess);       \
    } while (0)

int Worker::run_lane(int li)
{
    Lane& L = lanes[li];
    if (hipSetDevice(device) != hipSuccess) {
        lane_fail(L, SPLASH_ENODEV, "hipSetDevice", hipErrorInvalidDevice);
    } else {
        const size_t nblk = nblocks();
        for (;;) {
            size_t k;
            {   // the next block, once nobody else is loading: host->device copies of two lanes at once
                // only share the link (measured: 52 GB/s for one or for two), and lanes that load in
                // lockstep compute in lockstep -- one at a time, copies and kernels interleave
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !loading || abort; });
                if (abort || next_block >= nblk) break;
                k = next_block++;
                loading = true;
            }
            if (run_block(L, k)) break;
        }
        if (L.d2h.joinable()) L.d2h.join();
        if (L.d2h_rc && !L.rc) {
            L.rc = L.d2h_rc;
            snprintf(L.msg, sizeof L.msg, "device->host copy of a block's results failed");
            abort = true;
        }
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        loading = false; // (a lane that failed while loading)
    }
    size_t dummy = 0;
    note(dummy, true);
    return L.rc;
}

// Valid cells [b0, b0+nb) of the worker's range: what clFun does for a row block
// (R/splash.grid.R:277-308).
int Worker::run_block(Lane& L, size_t k)
{
    const splash_grid_in* in = P.in;
    const splash_grid_out* out = P.out;
    const size_t N = P.N;
    const int nt = P.nt, nt_out = P.nt_out;
    hipStream_t s = L.st.s;
    const double t_a = now_s();
    const size_t b0 = v0 + k * nb_max, nb = std::min(nb_max, v1 - b0);
    const size_t* idx = P.idx + b0; // idx[j] = grid cell of the block's j-th cell
    const size_t c0 = idx[0];       // dense: the block is cells [c0, c0+nb)
    double* F = L.d_force.as<double>();
    double* ddst[3] = {F, F + (size_t)nt * nb, F + (size_t)2 * nt * nb};
    double *dsw = ddst[0], *dtc = ddst[1], *dpn = ddst[2];
    // ---- host -> device.  (Pageable sources: the copies return when the source may be reused, so
    // the staging vectors can be refilled at once; on the device they are ordered by the stream.)
    for (int v = 0; v < 3; v++) {
        if (!f32) {
            const double* hsrc = v == 0 ? in->sw_in : (v == 1 ? in->tc : in->pn);
            if (P.dense) {
                LANE_TRY(hipMemcpy2DAsync(ddst[v], nb * 8, hsrc + c0, N * 8, nb * 8, (size_t)nt,
                                          hipMemcpyHostToDevice, s));
            } else {
                double* stg = L.stage.data();
                parallel_rows((size_t)nt, nb * 8, [=](size_t t) {
                    const double* src = hsrc + t * N;
                    double* dst = stg + t * nb;
                    for (size_t j = 0; j < nb; j++) dst[j] = src[idx[j]];
                });
                LANE_TRY(hipMemcpyAsync(ddst[v], stg, (size_t)nt * nb * 8, hipMemcpyHostToDevice, s));
            }
        } else { // single-precision transport, widened on the device (lossless for FLT4S rasters)
            const float* hsrc = v == 0 ? in->sw_in_f32 : (v == 1 ? in->tc_f32 : in->pn_f32);
            if (P.dense) {
                LANE_TRY(hipMemcpy2DAsync(L.d_f32.p, nb * 4, hsrc + c0, N * 4, nb * 4, (size_t)nt,
                                          hipMemcpyHostToDevice, s));
            } else {
                float* stg = L.stage32.data();
                parallel_rows((size_t)nt, nb * 4, [=](size_t t) {
                    const float* src = hsrc + t * N;
                    float* dst = stg + t * nb;
                    for (size_t j = 0; j < nb; j++) dst[j] = src[idx[j]];
                });
                LANE_TRY(hipMemcpyAsync(L.d_f32.p, stg, (size_t)nt * nb * 4, hipMemcpyHostToDevice, s));
            }
            LANE_TRY(launch_widen_f32(L.d_f32.as<float>(), (size_t)nt * nb, ddst[v], s)); // same stream: d_f32 is
                                                                                           // free for the next series
        }
    }
    // statics: lat, elev, slop, asp, resolution, soil[6], au[3]
    {
        double* hs = L.hstat.data();
        const double* st1[5] = {in->lat, in->elev, in->slop, in->asp, in->resolution};
        parallel_rows(14, nb * 8, [=](size_t r) {
            const double* src = r < 5 ? st1[r] : (r < 11 ? in->soil + (r - 5) * N : (in->au ? in->au + (r - 11) * N : nullptr));
            double* dst = hs + r * nb;
            if (!src) return;
            for (size_t j = 0; j < nb; j++) dst[j] = src[idx[j]];
        });
    }
    double* S = L.d_stat.as<double>();
    LANE_TRY(hipMemcpyAsync(S, L.hstat.data(), (size_t)14 * nb * 8, hipMemcpyHostToDevice, s));
    const bool with_basin = !L.hbasin.empty();
    if (with_basin) {
        for (size_t j = 0; j < nb; j++) L.hbasin[j] = in->basin[idx[j]];
        LANE_TRY(hipMemcpyAsync(L.d_basin.p, L.hbasin.data(), nb * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    {   // the copies have left the host buffers: the next lane may load
        std::lock_guard<std::mutex> lk(mu);
        loading = false;
        cv.notify_all();
    }
    const double t_b = now_s();

    // ---- kernels
    double* par = L.d_par.as<double>();
    double* state = L.d_state.as<double>();
    double* Tt = L.d_tt.as<double>();
    LANE_RC(splash_dev_soil_hydro(nb, S, S + nb, S + 2 * nb, S + 3 * nb, S + 4 * nb, S + 5 * nb,
                                  in->au ? S + 11 * nb : nullptr, par, s));
    LANE_RC(splash_dev_snow_threshold(nb, nt, dtc, nb, par, nb, Tt, s));
    int* dpass = out->aux ? L.d_pass.as<int>() : nullptr;
    LANE_RC(splash_dev_spin_up_begin(cal, nb, dsw, dtc, dpn, nb, Tt, par, nb, state, nb, dpass, L.d_swork.p, s));
    const unsigned char* d_mask = nullptr;
    const uint32_t *d_cells = nullptr, *d_ncells = nullptr;
    (void)splash_dev_spin_up_pending(L.d_swork.p, nb, &d_mask, &d_cells, &d_ncells);
    {   // the block's stragglers move to the pool
        size_t o_wl_b, o_meta, o_ai, o_cnt, p_wl_b, p_meta, p_ai, p_cnt;
        spin_work_layout(nb, &o_wl_b, &o_meta, &o_ai, &o_cnt);
        spin_work_layout(pool.cap, &p_wl_b, &p_meta, &p_ai, &p_cnt);
        char* w = (char*)L.d_swork.p;
        char* pw = (char*)pool.work.p;
        PoolGatherArgs g{};
        g.cells = d_cells;
        g.n_cells = d_ncells;
        g.nt = nt;
        g.sw = dsw; g.tc = dtc; g.pn = dpn;
        g.ldf = nb;
        g.par = par; g.ldp = nb;
        g.state = state; g.lds = nb;
        g.Tt = Tt;
        g.meta = (const int*)(w + o_meta);
        g.ai = (const double*)(w + o_ai);
        g.passes = dpass;
        g.ntotal = nb;
        g.basin = with_basin ? L.d_basin.as<int>() : nullptr;
        g.origin0 = (uint64_t)(b0 - v0);
        g.cap = pool.cap;
        g.p_sw = pool.sw();
        g.p_tc = pool.sw() + (size_t)nt * pool.cap;
        g.p_pn = pool.sw() + (size_t)2 * nt * pool.cap;
        g.p_par = pool.par.as<double>();
        g.p_state = pool.state.as<double>();
        g.p_Tt = pool.tt.as<double>();
        g.p_meta = (int*)(pw + p_meta);
        g.p_ai = (double*)(pw + p_ai);
        g.p_passes = out->aux ? pool.pass.as<int>() : nullptr;
        g.p_basin = with_basin ? pool.basin.as<int>() : nullptr;
        g.p_origin = pool.origin.as<uint64_t>();
        g.p_wl = (uint32_t*)pw; // worklist 0: what the tail phase reads
        g.p_n = (uint32_t*)(pw + p_cnt) + (kSpinPhases - 2);
        g.res = L.d_res.as<uint32_t>();
        LANE_TRY(launch_pool_gather(g, s));
        LANE_TRY(hipEventRecord(L.ev_gather.e, s));
    }
    note(blocks_gathered, false); // the worker's thread starts the tail once every block is here
    // the previous block's results must have left d_out before run_all writes it again
    if (L.d2h.joinable()) L.d2h.join();
    if (L.d2h_rc) return lane_fail(L, L.d2h_rc, "device->host copy of the previous block", hipSuccess);
    splash_dev_out dout{};
    int q = 0;
    for (int kk = 0; kk < SPLASH_NOUT; kk++)
        dout.out[kk] = out->out[kk] ? L.d_out.as<double>() + (size_t)(q++) * nt_out * nb : nullptr;
    dout.ldo = nb;
    dout.monthly = P.monthly;
    if (P.want_diag) {
        dout.diag_acc = d_acc.as<int64_t>();
        dout.basin = with_basin ? L.d_basin.as<int32_t>() : nullptr;
        dout.n_basin = P.n_basin;
    }
    dout.skip = d_mask;
    LANE_RC(splash_dev_run_all(cal, 0, nt, nb, dsw, dtc, dpn, nb, Tt, par, nb, state, nb, &dout, s));
    // The kernels of this block are queued; while they run, map the caller's (typically freshly
    // allocated) output arrays: the first touch of tens of GB is otherwise paid page by page inside
    // the device-to-host copies.
    // Only THIS block's columns (the rows' segments [c0, c0 + nb)): the other lane may be delivering a
    // neighbouring block's results at this moment, and a page-touching write outside the block's own
    // columns would race with them.
    if (P.dense) prefault_outputs(out->out, SPLASH_NOUT, N, c0, nb, (size_t)nt_out);
    const double t_c = now_s();
    LANE_TRY(hipStreamSynchronize(s));
    const double t_d = now_s();

    // ---- device -> host: the small arrays now (the next block overwrites them), the result series
    // by a helper thread while this lane loads its next block
    uint32_t res[3] = {0, 0, 0};
    LANE_TRY(hipMemcpy(res, L.d_res.p, sizeof res, hipMemcpyDeviceToHost));
    if (out->state_end) {
        LANE_TRY(hipMemcpy(L.hsmall.data(), state, 5 * nb * 8, hipMemcpyDeviceToHost));
        for (int kk = 0; kk < 5; kk++)
            for (size_t j = 0; j < nb; j++) out->state_end[(size_t)kk * N + idx[j]] = L.hsmall[(size_t)kk * nb + j];
    }
    if (out->aux) {
        LANE_TRY(hipMemcpy(L.hsmall.data(), Tt, nb * 8, hipMemcpyDeviceToHost));
        LANE_TRY(hipMemcpy(L.hsmall.data() + nb, par + (size_t)15 * nb, nb * 8, hipMemcpyDeviceToHost));
        LANE_TRY(hipMemcpy(L.hpass.data(), L.d_pass.p, 2 * nb * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t j = 0; j < nb; j++) {
            out->aux[idx[j]] = L.hsmall[j];
            out->aux[N + idx[j]] = L.hsmall[nb + j];
            out->aux[2 * N + idx[j]] = L.hpass[j];
            out->aux[3 * N + idx[j]] = L.hpass[nb + j];
        }
    }
    const bool in_place = res[1] < res[2]; // the pool cannot take this block's stragglers
    if (in_place) {
        if (int rc = deliver(L, b0, nb, dout)) return lane_fail(L, rc, "device->host copy", hipSuccess);
    } else {
        L.d2h_rc = SPLASH_OK;
        L.d2h = std::thread([this, &L, b0, nb, dout] {
            (void)hipSetDevice(device);
            L.d2h_rc = deliver(L, b0, nb, dout);
        });
    }
    const double t_e = now_s();
    if (in_place) // ... they are finished here, the old way (after the block's own results are out)
        if (int rc = late_fallback(L, b0, nb, dout, res[2])) return rc;
    if (trace)
        fprintf(stderr, "[splash] device %d block %zu (%zu cells): host->device %.3f s, queue+prefault %.3f s, "
                        "kernels (rest) %.3f s, small copies %.3f s (results follow in the background), %u stragglers "
                        "(%u to the pool)%s %.3f s\n",
                device, k, nb, t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d, res[2], res[1],
                in_place ? ", finished in place" : "", now_s() - t_e);
    return SPLASH_OK;
}

// the result series of a block: device -> the caller's arrays
int Worker::deliver(Lane& L, size_t b0, size_t nb, splash_dev_out dout)
{
    const splash_grid_out* out = P.out;
    const size_t N = P.N;
    const int nt_out = P.nt_out;
    const size_t* idx = P.idx + b0;
    const size_t c0 = idx[0];
    const double t0 = now_s();
    for (int kk = 0; kk < SPLASH_NOUT; kk++) {
        if (!out->out[kk]) continue;
        hipError_t e;
        if (P.dense) {
            e = hipMemcpy2D(out->out[kk] + c0, N * 8, dout.out[kk], nb * 8, nb * 8, (size_t)nt_out, hipMemcpyDeviceToHost);
        } else {
            double* stg = L.stage_out.data();
            e = hipMemcpy(stg, dout.out[kk], (size_t)nt_out * nb * 8, hipMemcpyDeviceToHost);
            double* dsth = out->out[kk];
            if (e == hipSuccess)
                parallel_rows((size_t)nt_out, nb * 8, [=](size_t t) {
                    const double* src = stg + t * nb;
                    double* dst = dsth + t * N;
                    for (size_t j = 0; j < nb; j++) dst[idx[j]] = src[j];
                });
        }
        if (e != hipSuccess) return SPLASH_EHIP;
    }
    if (trace)
        fprintf(stderr, "[splash] device %d: results of block at valid cell %zu delivered in %.3f s\n", device, b0, now_s() - t0);
    return SPLASH_OK;
}

// Pool overflow: the block's own tail phase and its late cells in list mode, on the lane's stream
// (what every block did before the pool existed).  Everything of the block is still on the device.
int Worker::late_fallback(Lane& L, size_t b0, size_t nb, const splash_dev_out& dout, uint32_t n_late)
{
    const splash_grid_out* out = P.out;
    const size_t N = P.N;
    const int nt = P.nt, nt_out = P.nt_out, n_out = P.n_out;
    const size_t* idx = P.idx + b0;
    hipStream_t s = L.st.s;
    double* F = L.d_force.as<double>();
    double *dsw = F, *dtc = F + (size_t)nt * nb, *dpn = F + (size_t)2 * nt * nb;
    double* par = L.d_par.as<double>();
    double* state = L.d_state.as<double>();
    double* Tt = L.d_tt.as<double>();
    int* dpass = out->aux ? L.d_pass.as<int>() : nullptr;
    LANE_RC(splash_dev_spin_up_finish(cal, nb, dsw, dtc, dpn, nb, Tt, par, nb, state, nb, dpass, L.d_swork.p, s));
    const uint32_t* d_cells = nullptr;
    (void)splash_dev_spin_up_pending(L.d_swork.p, nb, nullptr, &d_cells, nullptr);
    std::vector<uint32_t> late(n_late);
    LANE_TRY(hipMemcpyAsync(late.data(), d_cells, (size_t)n_late * 4, hipMemcpyDeviceToHost, s));
    const size_t late_cap = std::min<size_t>(n_late, 1 << 16);
    LANE_TRY(L.d_late.alloc((size_t)nt_out * std::max(n_out, 1) * late_cap * 8));
    std::vector<double> patch((size_t)nt_out * late_cap);
    for (size_t k0 = 0; k0 < n_late; k0 += late_cap) {
        const size_t nl = std::min(late_cap, (size_t)n_late - k0);
        splash_dev_out dl = dout;
        int ql = 0;
        for (int k = 0; k < SPLASH_NOUT; k++)
            dl.out[k] = out->out[k] ? L.d_late.as<double>() + (size_t)(ql++) * nt_out * late_cap : nullptr;
        dl.ldo = late_cap;
        dl.skip = nullptr;
        dl.cells = d_cells + k0;
        LANE_RC(splash_dev_run_all(cal, 0, nt, nl, dsw, dtc, dpn, nb, Tt, par, nb, state, nb, &dl, s));
        LANE_TRY(hipStreamSynchronize(s));
        for (int k = 0; k < SPLASH_NOUT; k++) {
            if (!out->out[k]) continue;
            LANE_TRY(hipMemcpy2D(patch.data(), nl * 8, dl.out[k], late_cap * 8, nl * 8, (size_t)nt_out,
                                 hipMemcpyDeviceToHost));
            for (int t = 0; t < nt_out; t++)
                for (size_t j = 0; j < nl; j++)
                    out->out[k][(size_t)t * N + idx[late[k0 + j]]] = patch[(size_t)t * nl + j];
        }
    }
    // state, threshold, aridity index and pass counts of the late cells
    if (out->state_end) {
        LANE_TRY(hipMemcpy(L.hsmall.data(), state, 5 * nb * 8, hipMemcpyDeviceToHost));
        for (uint32_t c : late)
            for (int k = 0; k < 5; k++) out->state_end[(size_t)k * N + idx[c]] = L.hsmall[(size_t)k * nb + c];
    }
    if (out->aux) {
        LANE_TRY(hipMemcpy(L.hsmall.data() + nb, par + (size_t)15 * nb, nb * 8, hipMemcpyDeviceToHost));
        LANE_TRY(hipMemcpy(L.hpass.data(), L.d_pass.p, 2 * nb * sizeof(int), hipMemcpyDeviceToHost));
        for (uint32_t c : late) {
            out->aux[N + idx[c]] = L.hsmall[nb + c];
            out->aux[2 * N + idx[c]] = L.hpass[c];
            out->aux[3 * N + idx[c]] = L.hpass[nb + c];
        }
    }
    return SPLASH_OK;
}

// The tail of the whole call: the last spin-up phase over the pool, then the transient run of its
// cells, patched into the results once the lanes have delivered theirs.
int Worker::finish_pool(char* err, size_t errlen)
{
    const splash_grid_out* out = P.out;
    const size_t cap = pool.cap;
    const int nt = P.nt, nt_out = P.nt_out, n_out = P.n_out;
    Stream st;
    HIP_TRY(st.create(), err, errlen);
    hipStream_t s = st.s;
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return blocks_gathered == nblocks() || lanes_done == (int)lanes.size(); });
    }
    if (abort) return SPLASH_OK; // the lanes report their own failure
    const double t0 = now_s();
    for (auto& L : lanes) HIP_TRY(hipStreamWaitEvent(s, L.ev_gather.e, 0), err, errlen);
    size_t p_cnt;
    spin_work_layout(cap, nullptr, nullptr, nullptr, &p_cnt);
    HIP_TRY(hipMemcpyAsync(&n_pool, (const uint32_t*)((char*)pool.work.p + p_cnt) + (kSpinPhases - 2),
                           sizeof n_pool, hipMemcpyDeviceToHost, s), err, errlen);
    HIP_TRY(hipStreamSynchronize(s), err, errlen); // every block's early phases and gather are done
    const double t1 = now_s();
    origin.assign(n_pool, 0);
    if (n_pool) {
        double *psw = pool.sw(), *ptc = psw + (size_t)nt * cap, *ppn = psw + (size_t)2 * nt * cap;
        int* ppass = out->aux ? pool.pass.as<int>() : nullptr;
        int rc = spin_phases(cal, cap, psw, ptc, ppn, cap, pool.tt.as<double>(), pool.par.as<double>(), cap,
                             pool.state.as<double>(), cap, ppass, pool.work.p, s, kSpinPhases - 1, kSpinPhases);
        if (rc) return fail(rc, err, errlen, "%s", g_err.c_str());
        splash_dev_out dout{};
        int q = 0;
        for (int k = 0; k < SPLASH_NOUT; k++)
            dout.out[k] = out->out[k] ? pool.out.as<double>() + (size_t)(q++) * nt_out * cap : nullptr;
        dout.ldo = cap;
        dout.monthly = P.monthly;
        if (P.want_diag) {
            dout.diag_acc = d_acc.as<int64_t>();
            dout.basin = P.in->basin ? pool.basin.as<int32_t>() : nullptr;
            dout.n_basin = P.n_basin;
        }
        rc = splash_dev_run_all(cal, 0, nt, n_pool, psw, ptc, ppn, cap, pool.tt.as<double>(), pool.par.as<double>(),
                                cap, pool.state.as<double>(), cap, &dout, s);
        if (rc) return fail(rc, err, errlen, "%s", g_err.c_str());
        HIP_TRY(hipMemcpyAsync(origin.data(), pool.origin.p, (size_t)n_pool * 8, hipMemcpyDeviceToHost, s), err, errlen);
        patch.resize((size_t)std::max(n_out, 1) * nt_out * n_pool);
        q = 0;
        for (int k = 0; k < SPLASH_NOUT; k++)
            if (out->out[k])
                HIP_TRY(hipMemcpy2DAsync(patch.data() + (size_t)(q++) * nt_out * n_pool, (size_t)n_pool * 8, dout.out[k],
                                         cap * 8, (size_t)n_pool * 8, (size_t)nt_out, hipMemcpyDeviceToHost, s),
                        err, errlen);
        if (out->state_end) {
            pstate.resize((size_t)5 * n_pool);
            HIP_TRY(hipMemcpy2DAsync(pstate.data(), (size_t)n_pool * 8, pool.state.p, cap * 8, (size_t)n_pool * 8, 5,
                                     hipMemcpyDeviceToHost, s), err, errlen);
        }
        if (out->aux) {
            paux.resize((size_t)n_pool);
            ppass_h.resize((size_t)2 * n_pool);
            HIP_TRY(hipMemcpyAsync(paux.data(), pool.par.as<double>() + (size_t)15 * cap, (size_t)n_pool * 8,
                                   hipMemcpyDeviceToHost, s), err, errlen);
            HIP_TRY(hipMemcpy2DAsync(ppass_h.data(), (size_t)n_pool * 4, pool.pass.p, cap * 4, (size_t)n_pool * 4, 2,
                                     hipMemcpyDeviceToHost, s), err, errlen);
        }
        HIP_TRY(hipStreamSynchronize(s), err, errlen);
    }
    t_pool[0] = t1 - t0;
    t_pool[1] = now_s() - t1;
    return SPLASH_OK;
}

// after the lanes have delivered their blocks: the pool's cells overwrite their columns
void Worker::patch_pool()
{
    if (!n_pool) return;
    const double t0 = now_s();
    const splash_grid_out* out = P.out;
    const size_t N = P.N, np = n_pool;
    const int nt_out = P.nt_out;
    const size_t* idx = P.idx + v0;
    const uint64_t* org = origin.data();
    int q = 0;
    for (int k = 0; k < SPLASH_NOUT; k++) {
        if (!out->out[k]) continue;
        const double* src = patch.data() + (size_t)(q++) * nt_out * np;
        double* dst = out->out[k];
        parallel_rows((size_t)nt_out, np * 16, [=](size_t t) {
            for (size_t j = 0; j < np; j++) dst[t * N + idx[org[j]]] = src[t * np + j];
        });
    }
    for (size_t j = 0; j < np; j++) {
        const size_t c = idx[org[j]];
        if (out->state_end)
            for (int k = 0; k < 5; k++) out->state_end[(size_t)k * N + c] = pstate[(size_t)k * np + j];
        if (out->aux) {
            out->aux[N + c] = paux[j];
            out->aux[2 * N + c] = ppass_h[j];
            out->aux[3 * N + c] = ppass_h[np + j];
        }
    }
    t_pool[2] = now_s() - t0;
}

// Valid cells [v0, v1) on `device`, in HBM-sized blocks dealt to the lanes.  `share` = workers using
// this device (they split its free memory).  acc: this worker's host copy of the exact diagnostic
// limbs (summed by the caller).
int grid_worker(const GridPlan& P, int device, int share, size_t v0, size_t v1, int64_t* acc,
                char* err, size_t errlen)
{
    const splash_grid_in* in = P.in;
    const int nt = P.nt, nt_out = P.nt_out, n_out = P.n_out;
    HIP_TRY(hipSetDevice(device), err, errlen);
    if (v1 <= v0) return SPLASH_OK;
    Worker W(P, device, v0, v1);
    if (int rc = splash_calendar_create(nt, in->doy, in->year, in->month, &W.cal, err, errlen)) return rc;
    const bool f32 = W.f32;
    const double t_w0 = now_s();
    const size_t want = v1 - v0;
    int n_lanes = 2;
    if (const char* e = getenv("SPLASH_HIP_LANES")) n_lanes = std::max(1, std::min(4, atoi(e)));
    const size_t acc_bytes = splash_dev_diag_acc_size(P.n_basin, P.nmonth);
    {
        // Sizing and allocation happen under the device's allocation lock: a second job on the same
        // device (splash_grid_submit) plans with what this one has left.
        std::lock_guard<std::recursive_mutex> alloc_lock(device_alloc_mutex(device));
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b), err, errlen);
        if (const char* e = getenv("SPLASH_HIP_FAKE_DEVICE_BYTES")) { // test aid: a device of that many bytes, of
            const size_t fake = (size_t)atoll(e), held = pool_held_bytes(device); // which the pool's buffers are taken
            free_b = std::min(free_b, fake > held ? fake - held : 0);
        }
        free_b += pool_idle_bytes(device); // buffers kept from earlier calls are ours to reuse or release
        // the straggler pool: room for 1/128 of the worker's cells (0.08 % of the synthetic grids' cells
        // are stragglers); a block whose stragglers do not fit finishes them in place (late_fallback)
        size_t pcap = std::min<size_t>(std::max<size_t>(want / 128, 16384), (size_t)1 << 18);
        pcap = std::min(pcap, (want + 255) & ~(size_t)255);
        pcap = (pcap + 255) & ~(size_t)255;
        if (const char* e = getenv("SPLASH_HIP_POOL_CAP")) pcap = std::max<size_t>(1, (size_t)atol(e)); // tests: force the overflow path
        const size_t pool_per_slot = (size_t)nt * 3 * 8 + (size_t)nt_out * std::max(n_out, 1) * 8 +
                                     (SPLASH_NPAR + SPLASH_NSTATE + 1) * 8 + 8 + 2 * 4 + 4;
        const double pool_bytes = (double)pcap * (double)pool_per_slot + (double)splash_dev_spin_up_work_size(pcap);
        const size_t per_cell = (size_t)nt * 3 * 8 + (f32 ? (size_t)nt * 4 : 0) + (size_t)nt_out * std::max(n_out, 1) * 8 +
                                (SPLASH_NPAR + SPLASH_NSTATE + 14 + 1) * 8 + 2 * 4 + 4 + 64;
        const double frac = P.mem_fraction > 0 ? P.mem_fraction : 0.85;
        const double budget = (double)free_b * frac / (double)std::max(share, 1) - pool_bytes - (double)acc_bytes;
        // blocks: four per lane once a grid is big enough for the pipeline to pay, of a million cells at
        // least (a block's later spin-up phases are short worklists, i.e. latency: ~30 ms per block)
        size_t nb_max = want;
        if (want > ((size_t)1 << 20)) nb_max = std::max<size_t>((size_t)1 << 20, (want + 4 * n_lanes - 1) / (4 * n_lanes));
        if (P.max_cells) nb_max = std::min(nb_max, P.max_cells);
        auto lanes_for = [&](size_t nb) { return (int)std::min<size_t>((size_t)n_lanes, (want + nb - 1) / nb); };
        auto need = [&](size_t nb) {
            return (double)lanes_for(nb) * ((double)nb * (double)per_cell + (double)splash_dev_spin_up_work_size(nb));
        };
        if (need(nb_max) > budget) { // largest block that fits (bisection; need() grows with nb once the
            size_t lo = 0, hi = nb_max;  // lane count has reached its cap, which it has wherever it matters)
            while (hi - lo > 256) {
                const size_t mid = lo + (hi - lo) / 2;
                if (need(mid) > budget) hi = mid; else lo = mid;
            }
            nb_max = lo;
        }
        if (nb_max == 0) return fail(SPLASH_ENOMEM, err, errlen, "grid_run: not enough device memory");
        nb_max = std::min(nb_max, kMaxCellsPerLaunch);
        if (nb_max > 256) nb_max -= nb_max % 256;
        W.nb_max = nb_max;
        n_lanes = lanes_for(nb_max);
        W.lanes = std::vector<Lane>(n_lanes);

        Pool& pl = W.pool;
        pl.cap = pcap;
        HIP_TRY(pl.force.alloc((size_t)nt * 3 * pcap * 8), err, errlen);
        HIP_TRY(pl.par.alloc((size_t)SPLASH_NPAR * pcap * 8), err, errlen);
        HIP_TRY(pl.state.alloc((size_t)SPLASH_NSTATE * pcap * 8), err, errlen);
        HIP_TRY(pl.tt.alloc(pcap * 8), err, errlen);
        HIP_TRY(pl.origin.alloc(pcap * 8), err, errlen);
        HIP_TRY(pl.pass.alloc(2 * pcap * sizeof(int)), err, errlen);
        HIP_TRY(pl.basin.alloc(pcap * sizeof(int32_t)), err, errlen);
        HIP_TRY(pl.work.alloc(splash_dev_spin_up_work_size(pcap)), err, errlen);
        HIP_TRY(pl.out.alloc((size_t)nt_out * std::max(n_out, 1) * pcap * 8), err, errlen);
        {   // the pool's counters start at zero (what phase 0 does for a block's workspace)
            size_t p_cnt;
            spin_work_layout(pcap, nullptr, nullptr, nullptr, &p_cnt);
            HIP_TRY(hipMemset((char*)pl.work.p + p_cnt, 0, 64), err, errlen);
        }
        if (P.want_diag) {
            HIP_TRY(W.d_acc.alloc(acc_bytes), err, errlen);
            HIP_TRY(hipMemset(W.d_acc.p, 0, acc_bytes), err, errlen);
        }
        for (Lane& L : W.lanes) {
            HIP_TRY(L.st.create(), err, errlen);
            HIP_TRY(L.ev_gather.create(), err, errlen);
            HIP_TRY(L.d_force.alloc((size_t)nt * 3 * nb_max * 8), err, errlen);
            if (f32) HIP_TRY(L.d_f32.alloc((size_t)nt * nb_max * 4), err, errlen);
            HIP_TRY(L.d_out.alloc((size_t)nt_out * std::max(n_out, 1) * nb_max * 8), err, errlen);
            HIP_TRY(L.d_par.alloc((size_t)SPLASH_NPAR * nb_max * 8), err, errlen);
            HIP_TRY(L.d_state.alloc((size_t)SPLASH_NSTATE * nb_max * 8), err, errlen);
            HIP_TRY(L.d_stat.alloc((size_t)14 * nb_max * 8), err, errlen);
            HIP_TRY(L.d_tt.alloc(nb_max * 8), err, errlen);
            HIP_TRY(L.d_pass.alloc(2 * nb_max * sizeof(int)), err, errlen);
            HIP_TRY(L.d_swork.alloc(splash_dev_spin_up_work_size(nb_max)), err, errlen);
            HIP_TRY(L.d_res.alloc(64), err, errlen);
            if (P.want_diag && in->basin) HIP_TRY(L.d_basin.alloc(nb_max * sizeof(int32_t)), err, errlen);
            L.hstat.resize((size_t)14 * nb_max);
            if (P.out->state_end || P.out->aux) L.hsmall.resize(5 * nb_max);
            if (P.out->aux) L.hpass.resize(2 * nb_max);
            if (in->basin && P.want_diag) L.hbasin.resize(nb_max);
            if (!P.dense) {
                if (!f32) L.stage.resize((size_t)nt * nb_max);
                L.stage_out.resize((size_t)nt_out * nb_max);
                if (f32) L.stage32.resize((size_t)nt * nb_max);
            }
        }
    }
    if (W.trace)
        fprintf(stderr, "[splash] device %d: %zu cells in %zu blocks of <= %zu on %zu lanes, pool of %zu slots; calendar + "
                        "allocations %.3f s\n", device, want, W.nblocks(), W.nb_max, W.lanes.size(), W.pool.cap, now_s() - t_w0);

    std::vector<std::thread> threads;
    for (size_t l = 0; l < W.lanes.size(); l++) threads.emplace_back([&W, l] { W.run_lane((int)l); });
    const int rc_pool = W.finish_pool(err, errlen); // waits for the lanes' last gathers, then runs the tail
    for (auto& t : threads) t.join();
    for (Lane& L : W.lanes)
        if (L.rc) return fail(L.rc, err, errlen, "%s", L.msg);
    if (rc_pool) return rc_pool;
    W.patch_pool();
    if (W.trace)
        fprintf(stderr, "[splash] device %d: %u stragglers in the pool: waited %.3f s for the last block's early phases, "
                        "tail + their transient run + copy %.3f s, patch %.3f s\n", device, W.n_pool, W.t_pool[0],
                W.t_pool[1], W.t_pool[2]);
    if (P.want_diag) HIP_TRY(hipMemcpy(acc, W.d_acc.p, acc_bytes, hipMemcpyDeviceToHost), err, errlen);
    return SPLASH_OK;
}

} // namespace

extern "C" int splash_grid_run(const splash_grid_in* in, const splash_opts* opt,
                               const splash_grid_out* out, char* err, size_t errlen)
{
    if (!in || !out) return fail(SPLASH_EINVAL, err, errlen, "grid_run: NULL in/out");
    const double t_enter = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const bool f64 = in->sw_in && in->tc && in->pn;
    const bool f32 = !in->sw_in && !in->tc && !in->pn && in->sw_in_f32 && in->tc_f32 && in->pn_f32;
    if (!in->doy || !in->year || !in->month || !(f64 || f32) || !in->lat || !in->elev ||
        !in->slop || !in->asp || !in->resolution || !in->soil)
        return fail(SPLASH_EINVAL, err, errlen,
                    "grid_run: NULL input array (forcing: all three series as double, or all "
                    "three as float with the double pointers NULL)");
    if (in->nt < 365)
        return fail(SPLASH_EINVAL, err, errlen,
                    "grid_run: %d days given, the spin-up needs the first 365 "
                    "(R/splash.point.R:141-144)", in->nt);
    if (in->basin && in->n_basin <= 0)
        return fail(SPLASH_EINVAL, err, errlen, "grid_run: basin ids given with n_basin <= 0");
    if (int rc = need_device(err, errlen)) return rc;

    // devices: one worker per entry of the list (the same ordinal may repeat)
    std::vector<int> devs;
    if (opt && opt->n_devices > 1) {
        for (int k = 0; k < opt->n_devices; k++) devs.push_back(opt->devices ? opt->devices[k] : k);
    } else {
        devs.push_back(opt ? ((opt->n_devices == 1 && opt->devices) ? opt->devices[0] : opt->device) : 0);
    }
    const int ndev_present = splash_hip_device_count();
    for (int d : devs)
        if (d < 0 || d >= ndev_present)
            return fail(SPLASH_EINVAL, err, errlen, "grid_run: device %d not present (%d visible)", d,
                        ndev_present);

    GridPlan P{};
    P.in = in;
    P.out = out;
    P.N = in->ncell;
    P.nt = in->nt;
    P.monthly = opt ? opt->monthly_out : 0;
    P.max_cells = opt ? opt->max_cells_per_launch : 0;
    P.mem_fraction = (opt && opt->mem_fraction > 0 && opt->mem_fraction <= 1) ? opt->mem_fraction : 0.0;
    P.n_basin = in->basin ? in->n_basin : 1;
    P.want_diag = out->diag || out->diag_acc;
    // month count without touching a device: same rule as splash_calendar_create
    {
        int mid = 0;
        for (int t = 0; t < in->nt; t++) {
            if (in->month[t] < 1 || in->month[t] > 12 || in->doy[t] < 1 || in->doy[t] > 366)
                return fail(SPLASH_EINVAL, err, errlen, "calendar: bad doy/month at day %d", t);
            if (t > 0 && (in->month[t] != in->month[t - 1] || in->year[t] != in->year[t - 1])) mid++;
        }
        P.nmonth = mid + 1;
    }
    P.nt_out = P.monthly ? P.nmonth : P.nt;
    const size_t N = P.N;
    const double NaN = std::numeric_limits<double>::quiet_NaN();

    // NA cells (R: every comparison false, all outputs NA): mask them out, emit NaN.  A cell is NA
    // when a static or its first forcing record is NaN (ocean cells of a masked raster); NaNs
    // later in a series propagate through the arithmetic like R's NA (see the header).
    std::vector<size_t> idx;
    {
        auto is_bad = [&](size_t c) {
            bool bad = isnan_d(in->lat[c]) || isnan_d(in->elev[c]) || isnan_d(in->slop[c]) ||
                       isnan_d(in->asp[c]) || isnan_d(in->resolution[c]);
            if (f64)
                bad = bad || isnan_d(in->sw_in[c]) || isnan_d(in->tc[c]) || isnan_d(in->pn[c]);
            else
                bad = bad || in->sw_in_f32[c] != in->sw_in_f32[c] || in->tc_f32[c] != in->tc_f32[c] ||
                      in->pn_f32[c] != in->pn_f32[c];
            for (int k = 0; k < 6 && !bad; k++)
                if (k != 4 && isnan_d(in->soil[(size_t)k * N + c])) bad = true; // bd NA is legal
            if (in->au)
                for (int k = 0; k < 3 && !bad; k++)
                    if (isnan_d(in->au[(size_t)k * N + c])) bad = true;
            return bad;
        };
        // a few threads over contiguous ranges, lists concatenated in order (16.7 M cells: 0.12 s on one)
        unsigned nthr = std::thread::hardware_concurrency();
        nthr = nthr == 0 ? 4 : (nthr > 8 ? 8 : nthr);
        if (N < ((size_t)1 << 20)) nthr = 1;
        std::vector<std::vector<size_t>> part(nthr);
        auto scan = [&](unsigned w) {
            const size_t a0 = N * w / nthr, a1 = N * (w + 1) / nthr;
            part[w].reserve(a1 - a0);
            for (size_t c = a0; c < a1; c++)
                if (!is_bad(c)) part[w].push_back(c);
        };
        if (nthr == 1) {
            scan(0);
            idx.swap(part[0]);
        } else {
            std::vector<std::thread> pool;
            for (unsigned w = 0; w < nthr; w++) pool.emplace_back(scan, w);
            for (auto& t : pool) t.join();
            size_t total = 0;
            for (auto& v : part) total += v.size();
            idx.reserve(total);
            for (auto& v : part) idx.insert(idx.end(), v.begin(), v.end());
        }
    }
    P.idx = idx.data();
    P.nv = idx.size();
    P.dense = (P.nv == N);
    if (!P.dense) {
        for (int k = 0; k < SPLASH_NOUT; k++)
            if (out->out[k]) std::fill(out->out[k], out->out[k] + (size_t)P.nt_out * N, NaN);
        if (out->state_end) std::fill(out->state_end, out->state_end + 5 * N, NaN);
        if (out->aux) std::fill(out->aux, out->aux + 4 * N, NaN);
    }
    const size_t n_rec = (size_t)P.n_basin * P.nmonth;
    if (out->diag) std::fill(out->diag, out->diag + n_rec * SPLASH_NDIAG, 0.0);
    if (out->diag_acc) std::fill(out->diag_acc, out->diag_acc + n_rec * SPLASH_DIAG_REC, (int64_t)0);
    if (P.nv == 0) return SPLASH_OK;
    for (int k = 0; k < SPLASH_NOUT; k++) P.n_out += out->out[k] ? 1 : 0;

    const double t_scan = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    // contiguous ranges of equal valid-cell count, one per worker (cells are independent: the
    // split changes nothing in the results, totals included -- they are exact integer sums)
    const int nw = (int)devs.size();
    std::vector<std::vector<int64_t>> acc(nw);
    std::vector<int> rcs(nw, 0);
    std::vector<std::string> msgs(nw);
    auto run = [&](int w) {
        const size_t v0 = P.nv * (size_t)w / nw, v1 = P.nv * (size_t)(w + 1) / nw;
        int share = 0;
        for (int d : devs) share += (d == devs[w]);
        if (P.want_diag) acc[w].assign(n_rec * SPLASH_DIAG_REC, 0);
        char buf[512] = "";
        rcs[w] = grid_worker(P, devs[w], share, v0, v1, acc[w].data(), buf, sizeof buf);
        msgs[w] = buf;
    };
    // Large result arrays (daily outputs: tens of GB) are page-locked while the first blocks load and
    // compute: device->host copies into registered memory run beside the host->device copies of the
    // next block (measured, tools/ubench/pcie_probe.py: 90 GB/s in both directions against 52 GB/s
    // shared from pageable memory) and registering costs 9 ms per GB on a helper thread.  A failed
    // registration (limits, memory that is already registered) just leaves that array pageable.
    struct Registered {
        std::vector<void*> done;
        std::thread th;
        std::atomic<bool> stop{false};
        ~Registered()
        {
            stop = true;
            if (th.joinable()) th.join();
            for (void* p : done) (void)hipHostUnregister(p);
        }
    } reg;
    {
        const size_t bytes_each = (size_t)P.nt_out * N * 8;
        size_t total = 0;
        for (int k = 0; k < SPLASH_NOUT; k++) total += out->out[k] ? bytes_each : 0;
        static const bool off = getenv("SPLASH_HIP_NO_REGISTER") != nullptr;
        if (!off && P.dense && total >= ((size_t)8 << 30)) {
            const int dev0 = devs[0];
            // the forcing too (both directions must be page-locked for the copies to overlap), first:
            // its copies start at once
            const size_t fbytes = (size_t)P.nt * N * (f64 ? 8 : 4);
            void* fptr[3] = {f64 ? (void*)in->sw_in : (void*)in->sw_in_f32, f64 ? (void*)in->tc : (void*)in->tc_f32,
                             f64 ? (void*)in->pn : (void*)in->pn_f32};
            reg.th = std::thread([&reg, out, bytes_each, dev0, fbytes, fptr] {
                (void)hipSetDevice(dev0);
                auto lock = [&](void* p, size_t n) {
                    if (hipHostRegister(p, n, hipHostRegisterPortable) == hipSuccess)
                        reg.done.push_back(p);
                    else
                        (void)hipGetLastError();
                };
                for (int v = 0; v < 3 && !reg.stop; v++) lock(fptr[v], fbytes);
                for (int k = 0; k < SPLASH_NOUT && !reg.stop; k++)
                    if (out->out[k]) lock(out->out[k], bytes_each);
            });
        }
    }
    if (nw == 1) {
        run(0);
    } else {
        std::vector<std::thread> pool;
        for (int w = 0; w < nw; w++) pool.emplace_back(run, w);
        for (auto& t : pool) t.join();
    }
    for (int w = 0; w < nw; w++)
        if (rcs[w]) return fail(rcs[w], err, errlen, "device %d: %s", devs[w], msgs[w].c_str());
    if (P.want_diag) {
        std::vector<int64_t> total(n_rec * SPLASH_DIAG_REC, 0);
        for (int w = 0; w < nw; w++)
            for (size_t i = 0; i < total.size(); i++) total[i] += acc[w][i];
        if (out->diag_acc) std::copy(total.begin(), total.end(), out->diag_acc);
        if (out->diag) splash_diag_finalize_host(total.data(), P.n_basin, P.nmonth, out->diag, nullptr);
    }
    if (getenv("SPLASH_TRACE")) {
        const double t_end = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        fprintf(stderr, "[splash] grid_run: NA scan %.3f s, workers %.3f s, total %.3f s\n", t_scan - t_enter,
                t_end - t_scan, t_end - t_enter);
    }
    return SPLASH_OK;
}

struct splash_grid_job {
    splash_grid_in in;
    splash_opts opt;
    splash_grid_out out;
    std::vector<int> devices;
    bool has_opt = false;
    std::thread th;
    int rc = SPLASH_OK;
    char msg[512] = "";
};

extern "C" int splash_grid_submit(const splash_grid_in* in, const splash_opts* opt, const splash_grid_out* out,
                                  splash_grid_job** job, char* err, size_t errlen)
{
    if (!in || !out || !job) return fail(SPLASH_EINVAL, err, errlen, "grid_submit: NULL in/out/job");
    if (int rc = need_device(err, errlen)) return rc;
    auto* j = new splash_grid_job;
    j->in = *in;
    j->out = *out;
    if (opt) {
        j->opt = *opt;
        j->has_opt = true;
        if (opt->devices && opt->n_devices > 0) { // the list is copied: the caller's may be a temporary
            j->devices.assign(opt->devices, opt->devices + opt->n_devices);
            j->opt.devices = j->devices.data();
        }
    }
    try {
        j->th = std::thread([j] { j->rc = splash_grid_run(&j->in, j->has_opt ? &j->opt : nullptr, &j->out, j->msg, sizeof j->msg); });
    } catch (...) {
        delete j;
        return fail(SPLASH_ENOMEM, err, errlen, "grid_submit: could not start a host thread");
    }
    *job = j;
    return SPLASH_OK;
}

extern "C" int splash_grid_wait(splash_grid_job* job, char* err, size_t errlen)
{
    if (!job) return fail(SPLASH_EINVAL, err, errlen, "grid_wait: NULL job");
    if (job->th.joinable()) job->th.join();
    const int rc = job->rc;
    if (rc) (void)fail(rc, err, errlen, "%s", job->msg);
    delete job;
    return rc;
}

extern "C" int splash_point_run(int nt, const int* doy, const int* year, const int* month,
                                const double* sw_in, const double* tc, const double* pn, double lat,
                                double elev, double slop, double asp, const double* soil_data,
                                const double* au, int n_au, double resolution, int monthly_out,
                                double* const outp[SPLASH_NOUT], char* err, size_t errlen)
{
    if (!soil_data || !outp) return fail(SPLASH_EINVAL, err, errlen, "point_run: NULL argument");
    if (n_au != 0 && n_au != 1 && n_au != 3)
        return fail(SPLASH_EINVAL, err, errlen, "point_run: Au must have 1 or 3 values");
    // length(Au)==1 => ncellin = ncellout = 3 (R/splash.point.R:106-110)
    double au3[3] = {0.0, 3.0, 3.0};
    if (au && n_au >= 1) au3[0] = au[0];
    if (au && n_au == 3) {
        au3[1] = au[1];
        au3[2] = au[2];
    }
    splash_grid_in in{};
    in.ncell = 1;
    in.nt = nt;
    in.doy = doy;
    in.year = year;
    in.month = month;
    in.sw_in = sw_in;
    in.tc = tc;
    in.pn = pn;
    in.lat = &lat;
    in.elev = &elev;
    in.slop = &slop;
    in.asp = &asp;
    in.resolution = &resolution;
    in.soil = soil_data;
    in.au = au3;
    splash_opts opt{};
    opt.monthly_out = monthly_out;
    splash_grid_out o{};
    for (int k = 0; k < SPLASH_NOUT; k++) o.out[k] = outp[k];
    return splash_grid_run(&in, &opt, &o, err, errlen);
}

// unSWC for host buffers: wn [nt][ncell], soil [6][ncell] -> wn_z, wtd, theta [nt][ncell]
extern "C" int splash_unswc_run(size_t ncell, int nt, const double* wn, const double* soil,
                                double uns_depth, double* wn_z, double* wtd, double* theta,
                                char* err, size_t errlen)
{
    if (!wn || !soil) return fail(SPLASH_EINVAL, err, errlen, "unswc_run: NULL input");
    if (int rc = need_device(err, errlen)) return rc;
    if (ncell == 0 || nt <= 0) return SPLASH_OK;
    const size_t n = ncell * (size_t)nt;
    DevBuf dw, ds, dz, dt_, dh;
    HIP_TRY(dw.alloc(n * 8), err, errlen);
    HIP_TRY(ds.alloc(6 * ncell * 8), err, errlen);
    HIP_TRY(hipMemcpy(dw.p, wn, n * 8, hipMemcpyHostToDevice), err, errlen);
    HIP_TRY(hipMemcpy(ds.p, soil, 6 * ncell * 8, hipMemcpyHostToDevice), err, errlen);
    if (wn_z) HIP_TRY(dz.alloc(n * 8), err, errlen);
    if (wtd) HIP_TRY(dt_.alloc(n * 8), err, errlen);
    if (theta) HIP_TRY(dh.alloc(n * 8), err, errlen);
    int rc = splash_dev_unswc(ncell, nt, dw.as<double>(), ncell, ds.as<double>(), uns_depth,
                              wn_z ? dz.as<double>() : nullptr, wtd ? dt_.as<double>() : nullptr,
                              theta ? dh.as<double>() : nullptr, ncell, nullptr);
    if (rc) return fail(rc, err, errlen, "%s", g_err.c_str());
    HIP_TRY(hipDeviceSynchronize(), err, errlen);
    if (wn_z) HIP_TRY(hipMemcpy(wn_z, dz.p, n * 8, hipMemcpyDeviceToHost), err, errlen);
    if (wtd) HIP_TRY(hipMemcpy(wtd, dt_.p, n * 8, hipMemcpyDeviceToHost), err, errlen);
    if (theta) HIP_TRY(hipMemcpy(theta, dh.p, n * 8, hipMemcpyDeviceToHost), err, errlen);
    return SPLASH_OK;
}

// static rasters on host buffers: R/splash.grid.R:96-110 + upslope_area() (R/upslope_area.R:12-58)
extern "C" int splash_static_rasters_run(int H, int W, const double* elev, double ymax, double yres,
                                         double xres, int na_to_zero, double degree,
                                         const splash_static_out* out, char* err, size_t errlen)
{
    if (!elev || !out || H <= 0 || W <= 0)
        return fail(SPLASH_EINVAL, err, errlen, "static_rasters_run: NULL argument or empty grid");
    if (int rc = need_device(err, errlen)) return rc;
    const size_t n = (size_t)H * W, nb = n * sizeof(double);
    const bool graph = out->filled || out->flowdir || out->ups_ncell || out->ups_area || out->ncellin ||
                       out->ncellout || out->ups_area_mfd;
    DevBuf de, dsl, das, dla, dre, dfi, dfd, dcn, dar, dni, dno, dcodes, dmf;
    HIP_TRY(de.alloc(nb), err, errlen);
    HIP_TRY(hipMemcpy(de.p, elev, nb, hipMemcpyHostToDevice), err, errlen);
    if (out->slope) HIP_TRY(dsl.alloc(nb), err, errlen);
    if (out->aspect) HIP_TRY(das.alloc(nb), err, errlen);
    if (out->lat) HIP_TRY(dla.alloc(nb), err, errlen);
    HIP_TRY(dre.alloc(nb), err, errlen); // always: the fill uses its mean, the area its square
    int rc = splash_dev_terrain(H, W, de.as<double>(), ymax, yres, xres, na_to_zero,
                                out->slope ? dsl.as<double>() : nullptr, out->aspect ? das.as<double>() : nullptr,
                                out->lat ? dla.as<double>() : nullptr, dre.as<double>(), nullptr, nullptr);
    if (rc) return fail(rc, err, errlen, "%s", g_err.c_str());
    HIP_TRY(hipDeviceSynchronize(), err, errlen);
    std::vector<double> res(graph || out->resolution ? n : 0);
    if (!res.empty()) HIP_TRY(hipMemcpy(res.data(), dre.p, nb, hipMemcpyDeviceToHost), err, errlen);
    if (out->resolution) memcpy(out->resolution, res.data(), nb);
    if (graph) {
        // cellStats(resolution, stat = 'mean', na.rm = TRUE), R/upslope_area.R:14: R's mean is a
        // long-double sum followed by one long-double refinement pass
        long double sum = 0.0L;
        size_t cnt = 0;
        for (size_t i = 0; i < n; i++)
            if (res[i] == res[i]) {
                sum += res[i];
                cnt++;
            }
        if (cnt == 0) return fail(SPLASH_EINVAL, err, errlen, "static_rasters_run: the elevation is all NA");
        long double mean = sum / (long double)cnt, corr = 0.0L;
        for (size_t i = 0; i < n; i++)
            if (res[i] == res[i]) corr += (long double)res[i] - mean;
        const double mean_res = (double)(mean + corr / (long double)cnt);
        HIP_TRY(dfi.alloc(nb), err, errlen);
        HIP_TRY(dfd.alloc(nb), err, errlen);
        rc = splash_dev_sinkfill(H, W, de.as<double>(), mean_res, degree, dfi.as<double>(), nullptr, nullptr);
        if (!rc) rc = splash_dev_terrain(H, W, dfi.as<double>(), ymax, yres, xres, 0, nullptr, nullptr, nullptr,
                                         nullptr, dfd.as<double>(), nullptr);
        if (!rc && (out->ups_ncell || out->ups_area)) {
            HIP_TRY(dcn.alloc(nb), err, errlen);
            HIP_TRY(dar.alloc(nb), err, errlen);
            rc = splash_dev_flow_accumulation(H, W, dfd.as<double>(), dfi.as<double>(), 0, dre.as<double>(),
                                              dcn.as<double>(), dar.as<double>(), nullptr, nullptr);
        }
        if (!rc && out->ups_area_mfd) {
            HIP_TRY(dmf.alloc(nb), err, errlen);
            rc = splash_dev_mfd_area(H, W, dfi.as<double>(), mean_res, dmf.as<double>(), nullptr, nullptr);
        }
        if (!rc && (out->ncellin || out->ncellout)) {
            static const double in_top[9] = {2, 1, 128, 4, 0, 64, 8, 16, 32}; // R/upslope_area.R:142
            double codes[18];
            for (int q = 0; q < 9; q++) {
                codes[q] = in_top[q];
                codes[9 + q] = in_top[8 - q]; // :144: rev()
            }
            HIP_TRY(dcodes.alloc(sizeof(codes)), err, errlen);
            HIP_TRY(hipMemcpy(dcodes.p, codes, sizeof(codes), hipMemcpyHostToDevice), err, errlen);
            if (out->ncellin) {
                HIP_TRY(dni.alloc(nb), err, errlen);
                rc = splash_dev_ncellflow(H, W, dfd.as<double>(), dcodes.as<double>(), 0, dni.as<double>(), nullptr);
            }
            if (!rc && out->ncellout) {
                HIP_TRY(dno.alloc(nb), err, errlen);
                rc = splash_dev_ncellflow(H, W, dfd.as<double>(), dcodes.as<double>() + 9, 0, dno.as<double>(), nullptr);
            }
        }
        if (rc) return fail(rc, err, errlen, "%s", g_err.c_str());
    }
    HIP_TRY(hipDeviceSynchronize(), err, errlen);
    struct { double* h; DevBuf* d; } back[] = {{out->slope, &dsl}, {out->aspect, &das}, {out->lat, &dla},
                                               {out->filled, &dfi}, {out->flowdir, &dfd}, {out->ups_ncell, &dcn},
                                               {out->ups_area, &dar}, {out->ncellin, &dni}, {out->ncellout, &dno},
                                               {out->ups_area_mfd, &dmf}};
    for (auto& b : back)
        if (b.h) HIP_TRY(hipMemcpy(b.h, b.d->p, nb, hipMemcpyDeviceToHost), err, errlen);
    return SPLASH_OK;
}

// ------------------------------------------------------------------ Rcpp-module mirrors
namespace {

// cellpar of one cell from the module's arguments (soil_info as R/splash.point.R:110-114 builds it;
// asp arrives already shifted by -180, R/splash.point.R:131)
void par_from_soil_info(double lat, double elv, double slop, double asp, const double* si,
                        double* par)
{
    par[P_LAT] = lat;
    par[P_ELV] = elv;
    par[P_SLOP] = slop;
    par[P_ASP] = asp;
    for (int k = 0; k < 12; k++) par[P_SAT + k] = si[k];
    par[P_WMAXR] = std::numeric_limits<double>::quiet_NaN(); // sm_lim is not a module output
}

struct Upload {
    DevBuf b;
    int put(const void* h, size_t bytes)
    {
        if (b.alloc(bytes) != hipSuccess) return SPLASH_ENOMEM;
        if (hipMemcpy(b.p, h, bytes, hipMemcpyHostToDevice) != hipSuccess) return SPLASH_EHIP;
        return 0;
    }
};

} // namespace

// SPLASH::spin_up (src/SPLASH.cpp:1594-1749; R call at R/splash.point.R:148,152)
extern "C" int splash_mod_spin_up(double lat, double elv, int n, int y, const double* sw_in,
                                  const double* tair, const double* pn, double slop, double asp,
                                  const double* snowfall, const double* soil_info, double* sm,
                                  double* snow, double* qin, double* tdrain, double* ro,
                                  double* snwage, double* pet, int* passes)
{
    REQ(n > 0 && sw_in && tair && pn && snowfall && soil_info, "mod_spin_up: bad argument");
    if (int rc = need_device(nullptr, 0)) return rc;
    std::vector<DayRec> days(n);
    for (int i = 0; i < n; i++) {
        DayRec& d = days[i];
        orbit_day(i + 1, y, d.dr, d.sd, d.cd); // quick_run((i+1), y, ...) :1658
        d.ttm_s = 0.0;
        d.inv_trm14 = 1.0;
        d.month_id = 0;
        d.mdays = n;
        d.flags = (i == n - 1);
        d.doy = i + 1;
    }
    double par[SPLASH_NPAR];
    par_from_soil_info(lat, elv, slop, asp, soil_info, par);
    Upload ud, up, usw, utc, upn, usf;
    int rc = 0;
    if ((rc = ud.put(days.data(), sizeof(DayRec) * n)) || (rc = up.put(par, sizeof par)) ||
        (rc = usw.put(sw_in, 8 * (size_t)n)) || (rc = utc.put(tair, 8 * (size_t)n)) ||
        (rc = upn.put(pn, 8 * (size_t)n)) || (rc = usf.put(snowfall, 8 * (size_t)n)))
        return fail(rc, nullptr, 0, "mod_spin_up: upload failed");
    DevBuf dser, dptr, dstate, dpass;
    HIP_TRY(dser.alloc(7 * 8 * (size_t)n), nullptr, 0);
    HIP_TRY(hipMemset(dser.p, 0, 7 * 8 * (size_t)n), nullptr, 0);
    double* hptr[7];
    for (int k = 0; k < 7; k++) hptr[k] = dser.as<double>() + (size_t)k * n;
    HIP_TRY(dptr.alloc(sizeof hptr), nullptr, 0);
    HIP_TRY(hipMemcpy(dptr.p, hptr, sizeof hptr, hipMemcpyHostToDevice), nullptr, 0);
    HIP_TRY(dstate.alloc(5 * 8), nullptr, 0);
    HIP_TRY(dpass.alloc(2 * sizeof(int)), nullptr, 0);
    SpinArgs a{};
    a.days = ud.b.as<DayRec>();
    a.n = n;
    a.ncell = 1;
    a.sw = usw.b.as<double>();
    a.tc = utc.b.as<double>();
    a.pn = upn.b.as<double>();
    a.ldf = 1;
    a.Tt = nullptr;
    a.par = up.b.as<double>();
    a.ldp = 1;
    a.state = dstate.as<double>();
    a.lds = 1;
    a.passes = dpass.as<int>();
    HIP_TRY(launch_spin_up_mod(a, dptr.as<double*>(), 1, usf.b.as<double>(), nullptr), nullptr, 0);
    HIP_TRY(hipDeviceSynchronize(), nullptr, 0);
    double* dst[7] = {sm, snow, qin, tdrain, ro, snwage, pet}; // names of :1748
    for (int k = 0; k < 7; k++)
        if (dst[k])
            HIP_TRY(hipMemcpy(dst[k], hptr[k], 8 * (size_t)n, hipMemcpyDeviceToHost), nullptr, 0);
    if (passes) HIP_TRY(hipMemcpy(passes, dpass.p, sizeof(int), hipMemcpyDeviceToHost), nullptr, 0);
    return SPLASH_OK;
}

// SPLASH::run_all (src/SPLASH.cpp:1833-1916; R call at R/splash.point.R:158-172)
extern "C" int splash_mod_run_all(double lat, double elv, int nt, const int* doys, const int* yrs,
                                  const double* sw_in, const double* tair, const double* pn,
                                  double wn_last, double slop, double asp, double snow_last,
                                  const double* snowfall, const double* soil_info, double qin_last,
                                  double td_last, double nds_last, double* wn, double* ro,
                                  double* pet, double* aet, double* snow, double* cond,
                                  double* bflow, double* netr, double* qin_prev, double* tdrain,
                                  double* snwage)
{
    REQ(nt > 0 && doys && yrs && sw_in && tair && pn && snowfall && soil_info,
        "mod_run_all: bad argument");
    if (int rc = need_device(nullptr, 0)) return rc;
    std::vector<DayRec> days(nt);
    for (int i = 0; i < nt; i++) {
        DayRec& d = days[i];
        orbit_day(doys[i], yrs[i], d.dr, d.sd, d.cd);
        d.ttm_s = 0.0;
        d.inv_trm14 = 1.0;
        d.month_id = 0;
        d.mdays = nt;
        d.flags = (i == nt - 1);
        d.doy = doys[i];
    }
    double par[SPLASH_NPAR];
    par_from_soil_info(lat, elv, slop, asp, soil_info, par);
    const double st0[5] = {wn_last, snow_last, qin_last, td_last, nds_last};
    Upload ud, up, usw, utc, upn, usf, ust;
    int rc = 0;
    if ((rc = ud.put(days.data(), sizeof(DayRec) * nt)) || (rc = up.put(par, sizeof par)) ||
        (rc = usw.put(sw_in, 8 * (size_t)nt)) || (rc = utc.put(tair, 8 * (size_t)nt)) ||
        (rc = upn.put(pn, 8 * (size_t)nt)) || (rc = usf.put(snowfall, 8 * (size_t)nt)) ||
        (rc = ust.put(st0, sizeof st0)))
        return fail(rc, nullptr, 0, "mod_run_all: upload failed");
    DevBuf dser;
    HIP_TRY(dser.alloc(11 * 8 * (size_t)nt), nullptr, 0);
    RunArgs a{};
    a.days = ud.b.as<DayRec>();
    a.nt = nt;
    a.ncell = 1;
    a.sw = usw.b.as<double>();
    a.tc = utc.b.as<double>();
    a.pn = upn.b.as<double>();
    a.snowfall = usf.b.as<double>();
    a.ldf = 1;
    a.Tt = nullptr;
    a.par = up.b.as<double>();
    a.ldp = 1;
    a.state = ust.b.as<double>();
    a.lds = 1;
    for (int k = 0; k < 8; k++) a.out[k] = dser.as<double>() + (size_t)k * nt;
    a.out[8] = nullptr;
    for (int k = 0; k < 3; k++) a.extra[k] = dser.as<double>() + (size_t)(8 + k) * nt;
    a.ldo = 1;
    HIP_TRY(launch_run_all_mod(a, nullptr), nullptr, 0);
    HIP_TRY(hipDeviceSynchronize(), nullptr, 0);
    double* dst[11] = {wn, ro, pet, aet, snow, cond, bflow, netr, qin_prev, tdrain, snwage};
    for (int k = 0; k < 11; k++)
        if (dst[k])
            HIP_TRY(hipMemcpy(dst[k], dser.as<double>() + (size_t)k * nt, 8 * (size_t)nt,
                              hipMemcpyDeviceToHost), nullptr, 0);
    return SPLASH_OK;
}

// SPLASH::run_one_year (src/SPLASH.cpp:1751-1830): day i+1 of year y for i < n, starting from the
// LAST elements of the state vectors (wrap-around read at i == 0, :1794-1799); n_end = length of
// the input vectors.  Same 11 series as run_all (:1827-1829), each [n].
extern "C" int splash_mod_run_one_year(double lat, double elv, int n, int y, int n_end,
                                       const double* sw_in, const double* tair, const double* pn,
                                       const double* wn_vec, double slop, double asp,
                                       const double* snow, const double* snowfall,
                                       const double* soil_info, const double* qin_vec,
                                       const double* td_vec, const double* nds, double* wn_o,
                                       double* ro, double* pet, double* aet, double* snow_o,
                                       double* cond, double* bflow, double* netr, double* qin_prev,
                                       double* tdrain, double* snwage)
{
    REQ(n > 0 && n_end >= n && wn_vec && snow && qin_vec && td_vec && nds,
        "mod_run_one_year: bad argument");
    std::vector<int> doys(n), yrs(n, y);
    for (int i = 0; i < n; i++) doys[i] = i + 1;
    return splash_mod_run_all(lat, elv, n, doys.data(), yrs.data(), sw_in, tair, pn,
                              wn_vec[n_end - 1], slop, asp, snow[n_end - 1], snowfall, soil_info,
                              qin_vec[n_end - 1], td_vec[n_end - 1], nds[n_end - 1], wn_o, ro, pet,
                              aet, snow_o, cond, bflow, netr, qin_prev, tdrain, snwage);
}

extern "C" int splash_mod_moist_surf(double depth, double z, double bub_p, double wn, double SAT,
                                     double RES, double lambda, double* theta)
{
    REQ(theta, "mod_moist_surf: NULL result");
    if (int rc = need_device(nullptr, 0)) return rc;
    DevBuf d;
    HIP_TRY(d.alloc(8), nullptr, 0);
    HIP_TRY(launch_moist_surf(depth, z, bub_p, wn, SAT, RES, lambda, d.as<double>(), nullptr),
            nullptr, 0);
    HIP_TRY(hipMemcpy(theta, d.p, 8, hipMemcpyDeviceToHost), nullptr, 0);
    return SPLASH_OK;
}

extern "C" int splash_mod_inf_GA(double bub_press, double theta_i, double Ksat, double theta_s,
                                 double lambda, double P, double tdur, double slop,
                                 double* infiltration)
{
    REQ(infiltration, "mod_inf_GA: NULL result");
    if (int rc = need_device(nullptr, 0)) return rc;
    DevBuf d;
    HIP_TRY(d.alloc(8), nullptr, 0);
    HIP_TRY(launch_inf_GA(bub_press, theta_i, Ksat, theta_s, lambda, P, tdur, slop, d.as<double>(),
                          nullptr), nullptr, 0);
    HIP_TRY(hipMemcpy(infiltration, d.p, 8, hipMemcpyDeviceToHost), nullptr, 0);
    return SPLASH_OK;
}
