// splash_capi.hip -- the C ABI of librsplash_hip.so (include/splash_hip.h) and the host-side
// pipeline that stands where R/splash.point.R:92-214 and R/splash.grid.R:277-308 stand in the
// reference: calendar tables, stage sequencing, NA-cell masking, blocking over HBM capacity.
// There is no CPU compute path in this file: every entry point that produces results launches the
// HIP kernels, and fails with SPLASH_ENODEV when no device is present.
#include "splash_host.hpp"

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

using namespace splash;
using namespace splash::host;

// ------------------------------------------------------------------ errors
namespace splash {
namespace host {
thread_local std::string g_err;

int fail(int code, char* err, size_t errlen, const char* fmt, ...)
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

int exception_code(char* err, size_t errlen, const char* where) noexcept
{
    // (fail() assigns a std::string: under memory pressure that can throw again -- then only the code is reported)
    try {
        try {
            throw;
        } catch (const std::bad_alloc&) {
            return fail(SPLASH_ENOMEM, err, errlen, "%s: out of host memory", where);
        } catch (const std::exception& e) {
            return fail(SPLASH_EHIP, err, errlen, "%s: %s", where, e.what());
        } catch (...) {
            return fail(SPLASH_EHIP, err, errlen, "%s: unknown C++ exception", where);
        }
    } catch (...) {
        if (err && errlen) snprintf(err, errlen, "%s: out of host memory", where);
        return SPLASH_ENOMEM;
    }
}
} // namespace host
} // namespace splash

extern "C" int splash_hip_abi_version(void) { return SPLASH_ABI_VERSION; }
extern "C" int splash_hip_version(void) { return SPLASH_ABI_VERSION; } // the name SURVEY.md 8(b) gives it

extern "C" int splash_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* splash_hip_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------ environment knobs
namespace {
std::mutex g_knobs_mu;
splash::host::Knobs g_knobs;
bool g_knobs_read = false;
thread_local const splash::host::Knobs* t_knobs = nullptr; // a call's own copy (KnobScope), or NULL

size_t env_size(const char* name, size_t dflt)
{
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    char* end = nullptr;
    const unsigned long long v = strtoull(e, &end, 10);
    return end == e ? dflt : (size_t)v;
}

void knobs_read_locked()
{
    splash::host::Knobs k;
    k.no_pool = getenv("SPLASH_HIP_NO_POOL") != nullptr;
    if (g_knobs_read) {
        k.spin_cache_cap = g_knobs.spin_cache_cap; // frozen
    } else if (const char* e = getenv("SPLASH_HIP_SPIN_CACHE_CAP")) {
        k.spin_cache_cap = atol(e);
    }
    if (const char* e = getenv("SPLASH_DBG_ACTIVE_LANES")) k.dbg_active_lanes = atoi(e);
    k.no_nd_skip = getenv("SPLASH_HIP_NO_ND_SKIP") != nullptr;
    k.no_block_phase1 = getenv("SPLASH_HIP_NO_BLOCK_PHASE1") != nullptr;
    k.no_solo_tail = getenv("SPLASH_HIP_NO_SOLO_TAIL") != nullptr;
    if (const char* e = getenv("SPLASH_HIP_TAIL_STAGES")) {
        int b0 = 0, b1 = 0;
        if (sscanf(e, "%d,%d", &b0, &b1) == 2) {
            k.tail_stage[0] = std::max(0, std::min(400, b0));
            k.tail_stage[1] = std::max(0, std::min(400, b1));
        }
    }
    if (const char* e = getenv("SPLASH_HIP_TAIL_LDS_PAD")) k.tail_lds_pad = atoi(e);
    if (const char* e = getenv("SPLASH_HIP_PREFAULT")) {
        if (!strcmp(e, "touch")) k.prefault_mode = splash::host::PREFAULT_TOUCH;
        else if (!strcmp(e, "off") || !strcmp(e, "0")) k.prefault_mode = splash::host::PREFAULT_OFF;
    }
    k.prefault_min_bytes = env_size("SPLASH_HIP_PREFAULT_MIN_BYTES", k.prefault_min_bytes);
    k.prefault_delay_ms = env_size("SPLASH_HIP_PREFAULT_DELAY_MS", 0);
    k.test_prefault_whole_range = getenv("SPLASH_HIP_TEST_PREFAULT_WHOLE_RANGE") != nullptr;
    if (const char* e = getenv("SPLASH_HIP_TEST_THROW")) k.test_throw = atoi(e);
    k.trace = getenv("SPLASH_TRACE") != nullptr;
    if (const char* e = getenv("SPLASH_HIP_LANES")) k.lanes = std::max(1, std::min(4, atoi(e)));
    if (const char* e = getenv("SPLASH_HIP_FAKE_DEVICE_BYTES")) {
        k.fake_device = true;
        k.fake_device_bytes = (size_t)atoll(e);
    }
    if (const char* e = getenv("SPLASH_HIP_POOL_CAP")) k.pool_cap = std::max<size_t>(1, (size_t)atol(e));
    k.blocks_per_lane = std::max<size_t>(1, env_size("SPLASH_HIP_BLOCKS_PER_LANE", 4));
    k.min_block_cells = std::max<size_t>(1, env_size("SPLASH_HIP_MIN_BLOCK_CELLS", (size_t)1 << 20));
    k.no_register = getenv("SPLASH_HIP_NO_REGISTER") != nullptr;
    k.register_min_bytes = env_size("SPLASH_HIP_REGISTER_MIN_BYTES", (size_t)8 << 30);
    g_knobs = k;
    g_knobs_read = true;
}
} // namespace

void splash::host::knobs_refresh()
{
    if (t_knobs) return; // a thread working for a call in flight: the call keeps its knobs, and never calls getenv
    std::lock_guard<std::mutex> lk(g_knobs_mu);
    knobs_read_locked();
}

splash::host::KnobScope::KnobScope(const Knobs& k) : mine_(k), prev_(t_knobs) { t_knobs = &mine_; }
splash::host::KnobScope::~KnobScope() { t_knobs = prev_; }

splash::host::Knobs splash::host::knobs()
{
    if (t_knobs) return *t_knobs;
    std::lock_guard<std::mutex> lk(g_knobs_mu);
    if (!g_knobs_read) knobs_read_locked(); // (a library-internal thread never gets here first: every
                                            //  entry point that starts one has refreshed)
    return g_knobs;
}

int splash::host::enter_device(char* err, size_t errlen)
{
    knobs_refresh();
    return need_device(err, errlen);
}

int splash::host::need_device(char* err, size_t errlen)
{
    if (splash_hip_device_count() <= 0)
        return fail(SPLASH_ENODEV, err, errlen,
                    "no HIP device available: librsplash_hip has no CPU fallback");
    return 0;
}

// ------------------------------------------------------------------ per-device context
// Lookup table of the float viscosity routine (7.5 MB per device, built on first use by running the
// routine itself for every float of its argument range; never freed).
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
} // namespace

namespace splash {
namespace host {
std::recursive_mutex& device_alloc_mutex(int dev) { return g_dev_mu[(dev >= 0 && dev < 64) ? dev : 0]; }

static void pool_release_idle(DevCtx& c)
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
    if (dev < 0 || dev >= 64 || knobs().no_pool) return hipMalloc(out, bytes);
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
} // namespace host
} // namespace splash

extern "C" int splash_hip_cache_stats(int device, size_t* held_bytes, size_t* idle_bytes)
try {
    if (device < 0 || device >= 64) return fail(SPLASH_EINVAL, nullptr, 0, "cache_stats: device %d out of range", device);
    if (held_bytes) *held_bytes = pool_held_bytes(device);
    if (idle_bytes) *idle_bytes = pool_idle_bytes(device);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_hip_cache_stats");
}

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
try {
    if (doy < 1 || doy > 366 || !dr || !sd || !cd)
        return fail(SPLASH_EINVAL, nullptr, 0, "orbit_day: bad argument");
    orbit_day(doy, year, *dr, *sd, *cd);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_orbit_day");
}

void splash::host::orbit_of_day(int doy, int year, double& dr, double& sd, double& cd) { orbit_day(doy, year, dr, sd, cd); }

int splash::host::calendar_host(int nt, const int* doy, const int* year, const int* month, splash_calendar* cal,
                                char* err, size_t errlen)
{
    if (nt <= 0 || !doy || !year || !month)
        return fail(SPLASH_EINVAL, err, errlen, "calendar: nt must be > 0 and doy/year/month given");
    cal->nt = nt;
    cal->host.resize(nt);
    int mid = 0;
    for (int t = 0; t < nt; t++) {
        if (month[t] < 1 || month[t] > 12 || doy[t] < 1 || doy[t] > 366)
            return fail(SPLASH_EINVAL, err, errlen, "calendar: bad doy/month at day %d", t);
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
    return SPLASH_OK;
}

extern "C" int splash_calendar_create(int nt, const int* doy, const int* year, const int* month,
                                      splash_calendar** out, char* err, size_t errlen)
try {
    if (!out || nt <= 0 || !doy || !year || !month)
        return fail(SPLASH_EINVAL, err, errlen, "calendar: nt must be > 0 and doy/year/month given");
    if (int rc = enter_device(err, errlen)) return rc;
    auto* cal = new splash_calendar;
    if (int rc = calendar_host(nt, doy, year, month, cal, err, errlen)) {
        delete cal;
        return rc;
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
} catch (...) {
    return exception_code(err, errlen, "splash_calendar_create");
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
try {
    if (!cal) return fail(SPLASH_EINVAL, nullptr, 0, "calendar is NULL");
    for (int t = 0; t < cal->nt; t++) {
        if (dr) dr[t] = cal->host[t].dr;
        if (sd) sd[t] = cal->host[t].sd;
        if (cd) cd[t] = cal->host[t].cd;
    }
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_calendar_orbit");
}

// ------------------------------------------------------------------ device-resident stages
extern "C" int splash_dev_soil_hydro(size_t ncell, const double* lat, const double* elev,
                                     const double* slop, const double* asp,
                                     const double* resolution, const double* soil, const double* au,
                                     double* cellpar, void* stream)
try {
    REQ(lat && elev && slop && asp && resolution && soil && cellpar, "soil_hydro: NULL argument");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_soil_hydro(ncell, lat, elev, slop, asp, resolution, soil, au, cellpar,
                              (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_soil_hydro");
}

extern "C" int splash_dev_snow_threshold(size_t ncell, int nt, const double* tc, size_t ldf,
                                         const double* cellpar, size_t ldp, double* Tt, void* stream)
try {
    REQ(tc && cellpar && Tt && nt > 0, "snow_threshold: NULL argument or nt <= 0");
    REQ(ldf >= ncell && ldp >= ncell, "snow_threshold: leading dimension smaller than ncell");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_snow_threshold(ncell, nt, tc, ldf, cellpar, ldp, Tt, (hipStream_t)stream),
            nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_snow_threshold");
}

// workspace of splash_dev_spin_up: three worklists, job records, aridity index, phase counters, and
// the cache of state-independent day terms: 365 x kNRecCompact (15) doubles = 43 800 B per slot, 2^19
// slots at most (23 GB).  Phase 0 walks the cells in blocks of that many with compact records; the
// later phases lay RICH records (20 words, job by job) over the same bytes: spin_cache_rich_cap() =
// 15/20 of the compact capacity rounded down to whole wavefronts (393 216 slots at 2^19), with no
// slack behind the last slot -- which is why k_spin_phase never forms an address from a slot beyond
// the capacity (cs_ld) and why a spread launch deals only the positions below it.
constexpr size_t kSpinSnapshots = 1024;
static size_t spin_cache_cap(size_t ncell)
{
    size_t lim = (size_t)1 << 19;
    // test / tuning aid: a small capacity makes small grids walk phase 0 in several blocks and gives
    // the later phases jobs beyond the cache (read once per process: the workspace layout depends on it)
    const long env = knobs().spin_cache_cap;
    if (env >= 64) lim = (size_t)env;
    size_t cap = std::min<size_t>(ncell, lim);
    return (cap + 63) & ~(size_t)63;
}

static size_t g_spin_wl_c_offset(size_t ncell); // third worklist (the tail's cascade), see below

size_t splash::host::spin_work_layout(size_t ncell, size_t* o_wl_b, size_t* o_meta, size_t* o_ai,
                                      size_t* o_cnt, size_t* o_cache, size_t* o_pend)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    take(ncell * 4); // wl_a at 0
    size_t wl_b = take(ncell * 4), meta = take(ncell * 4), ai = take(ncell * 8), cnt = take(64 + 4 * kSpinSnapshots); // 16 counters + list positions per block of phase 0
    size_t pend = take(ncell);
    take(ncell * 4); // wl_c: a third list for the tail's cascade (g_spin_wl_c_offset)
    size_t cache = take(spin_cache_cap(ncell) * spin_cache_bytes_per_slot());
    if (o_pend) *o_pend = pend;
    if (o_wl_b) *o_wl_b = wl_b;
    if (o_meta) *o_meta = meta;
    if (o_ai) *o_ai = ai;
    if (o_cnt) *o_cnt = cnt;
    if (o_cache) *o_cache = cache;
    return off;
}

static size_t g_spin_wl_c_offset(size_t ncell)
{
    size_t o_pend;
    spin_work_layout(ncell, nullptr, nullptr, nullptr, nullptr, nullptr, &o_pend);
    return o_pend + ((ncell + 255) & ~(size_t)255);
}

extern "C" size_t splash_dev_spin_up_work_size(size_t ncell)
{
    return spin_work_layout(ncell, nullptr, nullptr, nullptr, nullptr);
}

// phases [ph0, ph1) of the spin-up (see k_spin_phase)
int splash::host::spin_phases(const splash_calendar* cal, size_t ncell, const double* sw_in,
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
    const Knobs kn = knobs(); // (the snapshot of the call's entry: no getenv on lane or pool threads)
    a.dbg_active_lanes = kn.dbg_active_lanes; // (read by measurement builds only)
    a.lds_pad = kn.tail_lds_pad;
    a.no_nd_skip = kn.no_nd_skip ? 1 : 0; // tests: the detector of round 3 (same final bits, more passes)
    bool phase1_done_per_block = false;
    for (int ph = ph0; ph < ph1; ph++) {
        if (ph == 1 && phase1_done_per_block) continue;
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
            // Phase 1 -- one more pass for what phase 0 leaves, two thirds of the cells -- is run block by block
            // too, right behind the block's phase 0, while the block's records are still in the cache: a read-back
            // pass (1 360 instructions per day step) instead of a pass of the general kernel, which forms the day
            // terms again (1 900).  The block's survivors are the entries phase 0 has just appended to its list,
            // from a position only the device knows (snapshot of the counter -> first_dev); a job's cache slot is
            // its cell's position in the block.  Same bits (the records are what the general kernel would form).
            // (measured, 16.7 M cells: 1 150 -> 1 131 ms for phases 0-6 -- less than the instruction counts promise:
            // the survivors' records are read at 65 % density and the pass is bound by those reads.)
            const bool no_p1 = kn.no_block_phase1; // A/B: the round-3 order
            const bool block_p1 = split && !no_p1 && ph1 > 1 && (ncell + a.cache_cap - 1) / a.cache_cap <= kSpinSnapshots;
            uint32_t* const snap = cnt + 16;
            size_t blk = 0;
            for (size_t c0 = 0; c0 < ncell; c0 += a.cache_cap, blk++) {
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
                    if (block_p1) HIP_TRY(hipMemcpyAsync(snap + blk, cnt, sizeof(uint32_t), hipMemcpyDeviceToDevice, s), nullptr, 0);
                    HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
                    if (block_p1) { // phase 1 of this block's survivors: wl[0][snap, cnt[0]) -> wl[1]
                        a.budget = budget[1];
                        a.wl_in = wl[0];
                        a.n_in = cnt;
                        a.first_dev = snap + blk;
                        a.slot_by_cell = 1;
                        a.wl_out = wl[1];
                        a.n_out = cnt + 1;
                        HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
                        a.wl_in = nullptr;
                        a.n_in = nullptr;
                        a.first_dev = nullptr;
                        a.slot_by_cell = 0;
                        a.wl_out = wl[0];
                        a.n_out = cnt;
                    }
                    a.readback = 0;
                    a.fresh = 1;
                } else {
                    HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
                }
            }
            a.cell0 = 0;
            a.ncell = ncell;
            phase1_done_per_block = block_p1;
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
            const bool no_solo = kn.no_solo_tail; // A/B and tests: the round-3 tail
            if (ph == kSpinPhases - 1 && !no_solo) {
                // The tail as a cascade of read-back launches over shrinking lists, every one of them SPREAD: the
                // jobs dealt over 1 024 wavefronts, ceil(n / 1024) to each, the other lanes repeating them
                // (k_spin_phase, "Spread": a wavefront executes the union of its jobs' branches; ONE job runs a
                // pass of the year in about half the time of 64 packed together).  The cycle detector ends most
                // jobs within the first hundred passes, so: 32 passes (8 837 jobs of the benchmark grid: 9 to a
                // wavefront), 64 passes (2 319: 3 to a wavefront), the rest (881: one each).  A stage queues its
                // survivors by their position in the tail's list, which stays their cache slot (wl_slots).
                // SPLASH_HIP_TAIL_STAGES="a,b" sets the first two budgets (experiments).
                int stage_budget[3] = {32, 64, 0};
                if (kn.tail_stage[0] >= 0) {
                    stage_budget[0] = kn.tail_stage[0];
                    stage_budget[1] = kn.tail_stage[1];
                }
                stage_budget[2] = budget[ph] - 1 - stage_budget[0] - stage_budget[1];
                uint32_t* const lists[3] = {nullptr, wl[ph & 1], (uint32_t*)(w + g_spin_wl_c_offset(ncell))};
                uint32_t* const counts[3] = {cnt + (ph - 1), cnt + ph, cnt + ph + 1};
                HIP_TRY(hipMemsetAsync(cnt + ph, 0, 2 * sizeof(uint32_t), s), nullptr, 0);
                a.pending = nullptr;
                a.spread = 1;
                int prev = 0; // the stage whose output the next one reads
                for (int st = 0; st < 3; st++) {
                    if (stage_budget[st] <= 0) continue;
                    const bool last = st == 2;
                    a.budget = stage_budget[st];
                    a.wl_slots = lists[prev];
                    a.n_in = counts[prev];
                    a.queue_slots = last ? 0 : 1;
                    a.no_queue = last ? 1 : 0; // (the budgets cover the 1000-pass cap: nothing is left after the last stage)
                    a.wl_out = last ? wl[ph & 1] : lists[st + 1];
                    a.n_out = last ? cnt + ph : counts[st + 1];
                    HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
                    prev = st + 1;
                }
                a.spread = 0;
                a.wl_slots = nullptr;
                a.queue_slots = 0;
                a.no_queue = 0;
                a.n_in = cnt + (ph - 1);
                a.wl_out = wl[ph & 1];
                a.n_out = cnt + ph;
            } else {
                HIP_TRY(launch_spin_phase(a, s), nullptr, 0);
            }
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
try {
    knobs_refresh(); // the caller's thread (a no-op on a lane or pool thread: KnobScope); spin_phases() itself never refreshes
    return spin_phases(cal, ncell, sw_in, tc, pn, ldf, Tt, cellpar, ldp, state, lds, passes, work,
                       stream, 0, kSpinPhases);
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_spin_up");
}

extern "C" int splash_dev_spin_up_begin(const splash_calendar* cal, size_t ncell,
                                        const double* sw_in, const double* tc, const double* pn,
                                        size_t ldf, const double* Tt, double* cellpar, size_t ldp,
                                        double* state, size_t lds, int* passes, void* work,
                                        void* stream)
try {
    knobs_refresh(); // the caller's thread (a no-op on a lane or pool thread: KnobScope); spin_phases() itself never refreshes
    return spin_phases(cal, ncell, sw_in, tc, pn, ldf, Tt, cellpar, ldp, state, lds, passes, work,
                       stream, 0, kSpinPhases - 1);
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_spin_up_begin");
}

extern "C" int splash_dev_spin_up_finish(const splash_calendar* cal, size_t ncell,
                                         const double* sw_in, const double* tc, const double* pn,
                                         size_t ldf, const double* Tt, double* cellpar, size_t ldp,
                                         double* state, size_t lds, int* passes, void* work,
                                         void* stream)
try {
    knobs_refresh(); // the caller's thread (a no-op on a lane or pool thread: KnobScope); spin_phases() itself never refreshes
    return spin_phases(cal, ncell, sw_in, tc, pn, ldf, Tt, cellpar, ldp, state, lds, passes, work,
                       stream, kSpinPhases - 1, kSpinPhases);
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_spin_up_finish");
}

extern "C" int splash_dev_spin_up_pending(void* work, size_t ncell, const unsigned char** mask,
                                          const uint32_t** cells, const uint32_t** n_cells)
try {
    REQ(work, "spin_up_pending: NULL workspace");
    size_t o_wl_b, o_meta, o_ai, o_cnt, o_cache, o_pend;
    spin_work_layout(ncell, &o_wl_b, &o_meta, &o_ai, &o_cnt, &o_cache, &o_pend);
    char* w = (char*)work;
    if (mask) *mask = (const unsigned char*)(w + o_pend);
    static_assert(((kSpinPhases - 2) & 1) == 0, "the phase before the tail writes worklist 0");
    if (cells) *cells = (const uint32_t*)w;
    if (n_cells) *n_cells = (const uint32_t*)(w + o_cnt) + (kSpinPhases - 2);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_spin_up_pending");
}

extern "C" int splash_dev_spin_up_pending_count(void* work, size_t ncell, void* stream,
                                                uint32_t* n_pending)
try {
    REQ(work && n_pending, "spin_up_pending_count: NULL argument");
    const uint32_t* d_n = nullptr;
    (void)splash_dev_spin_up_pending(work, ncell, nullptr, nullptr, &d_n);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream), nullptr, 0);
    HIP_TRY(hipMemcpy(n_pending, d_n, sizeof(uint32_t), hipMemcpyDeviceToHost), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_spin_up_pending_count");
}

extern "C" size_t splash_dev_diag_acc_size(int n_basin, int nmonth)
{
    return (size_t)std::max(n_basin, 1) * (size_t)std::max(nmonth, 0) * SPLASH_DIAG_REC * sizeof(int64_t);
}

extern "C" int splash_dev_diag_finalize(const int64_t* diag_acc, int n_basin, int nmonth,
                                        double* diag, double* counts, void* stream)
try {
    REQ(diag_acc && diag && nmonth >= 0, "diag_finalize: NULL argument");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_diag_finalize((const long long*)diag_acc, std::max(n_basin, 1) * nmonth, diag,
                                 counts, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_diag_finalize");
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
try {
    REQ(cal && sw_in && tc && pn && Tt && cellpar && state && out, "run_all: NULL argument");
    REQ(t0 >= 0 && nt >= 0 && t0 + nt <= cal->nt, "run_all: day range outside the calendar");
    REQ(ldf >= ncell && ldp >= ncell && lds >= ncell && out->ldo >= ncell,
        "run_all: leading dimension < ncell");
    REQ(!out->basin || out->diag_acc, "run_all: basin ids given without diag_acc");
    REQ(!out->basin || out->n_basin > 0, "run_all: basin ids given with n_basin <= 0");
    REQ(!(out->monthly && (out->extra[0] || out->extra[1] || out->extra[2])),
        "run_all: the qin_prev/tdrain/snwage series are daily (monthly = 0)");
    REQ(ncell <= kMaxCellsPerLaunch, "run_all: more than 2^29-1 cells in one call (32-bit lane byte offsets)");
    if (int rc = enter_device(nullptr, 0)) return rc;
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
    // the last month of this call goes on in the calendar after the call's last day (a later call completes it)
    a.tail_continues = (t0 + nt < cal->nt && cal->host[t0 + nt].month_id == cal->host[t0 + nt - 1].month_id) ? 1 : 0;
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
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_run_all");
}

extern "C" int splash_dev_synth_statics(uint64_t seed, int H, int W, int64_t cell0, size_t ncell,
                                        int terrain, double* lat, double* elev, double* slop,
                                        double* asp, double* resolution, double* soil, double* au,
                                        void* stream)
try {
    REQ(lat && elev && slop && asp && resolution && soil && au && H > 0 && W > 0,
        "synth_statics: NULL argument");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_synth_statics(seed, H, W, cell0, ncell, terrain, lat, elev, slop, asp,
                                 resolution, soil, au, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_synth_statics");
}

extern "C" int splash_dev_synth_forcing(uint64_t seed, const splash_calendar* cal, int t0, int nt,
                                        int64_t cell0, size_t ncell, const double* lat,
                                        const double* elev, double* sw_in, double* tc, double* pn,
                                        size_t ldf, void* stream)
try {
    REQ(cal && lat && elev && sw_in && tc && pn, "synth_forcing: NULL argument");
    REQ(t0 >= 0 && nt >= 0 && t0 + nt <= cal->nt, "synth_forcing: day range outside the calendar");
    REQ(ldf >= ncell, "synth_forcing: leading dimension < ncell");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_synth_forcing(seed, cal->dev + t0, t0, nt, cell0, ncell, lat, elev, sw_in, tc,
                                 pn, ldf, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_synth_forcing");
}

// static rasters before the loop: R/splash.grid.R:96-110, R/upslope_area.R:31-33
extern "C" int splash_dev_terrain(int H, int W, const double* elev, double ymax, double yres,
                                  double xres, int na_to_zero, double* slope_deg, double* aspect_deg,
                                  double* lat, double* resolution_m, double* flowdir, void* stream)
try {
    REQ(elev && H > 0 && W > 0, "terrain: NULL elevation or empty grid");
    REQ(yres > 0 && xres > 0, "terrain: cell size must be positive (degrees)");
    REQ(slope_deg || aspect_deg || lat || resolution_m || flowdir, "terrain: no output requested");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_terrain(H, W, elev, ymax, yres, xres, na_to_zero, slope_deg, aspect_deg, lat,
                           resolution_m, flowdir, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_terrain");
}

// the reference's large-grid terrain (gdaldem on the INT2S DEM, R/splash.grid.R:83-84): device rasters
extern "C" int splash_dev_terrain_gdal(int H, int W, const double* elev, double xres, double yres, double scale,
                                       int round_int, double* slope_deg, double* aspect_deg, void* stream)
try {
    REQ(elev && H > 0 && W > 0, "terrain_gdal: NULL elevation or empty grid");
    REQ(yres > 0 && xres > 0 && scale > 0, "terrain_gdal: cell size and scale must be positive");
    REQ(slope_deg || aspect_deg, "terrain_gdal: no output requested");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_terrain_gdal(H, W, elev, xres, yres, scale, round_int, slope_deg, aspect_deg, (hipStream_t)stream), nullptr,
            0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_terrain_gdal");
}

// ... host arrays
extern "C" int splash_terrain_gdal_run(int H, int W, const double* elev, double xres, double yres, double scale,
                                       int round_int, double* slope_deg, double* aspect_deg, char* err, size_t errlen)
try {
    if (!elev || H <= 0 || W <= 0 || !(slope_deg || aspect_deg))
        return fail(SPLASH_EINVAL, err, errlen, "terrain_gdal_run: NULL argument or empty grid");
    if (!(xres > 0 && yres > 0 && scale > 0)) return fail(SPLASH_EINVAL, err, errlen, "terrain_gdal_run: cell size and scale must be positive");
    if (int rc = enter_device(err, errlen)) return rc;
    const size_t nb = (size_t)H * W * 8;
    DevBuf de, ds, da;
    HIP_TRY(de.alloc(nb), err, errlen);
    if (slope_deg) HIP_TRY(ds.alloc(nb), err, errlen);
    if (aspect_deg) HIP_TRY(da.alloc(nb), err, errlen);
    HIP_TRY(hipMemcpy(de.p, elev, nb, hipMemcpyHostToDevice), err, errlen);
    HIP_TRY(launch_terrain_gdal(H, W, de.as<double>(), xres, yres, scale, round_int, slope_deg ? ds.as<double>() : nullptr,
                                aspect_deg ? da.as<double>() : nullptr, nullptr), err, errlen);
    HIP_TRY(hipDeviceSynchronize(), err, errlen);
    if (slope_deg) HIP_TRY(hipMemcpy(slope_deg, ds.p, nb, hipMemcpyDeviceToHost), err, errlen);
    if (aspect_deg) HIP_TRY(hipMemcpy(aspect_deg, da.p, nb, hipMemcpyDeviceToHost), err, errlen);
    return SPLASH_OK;
} catch (...) {
    return exception_code(err, errlen, "splash_terrain_gdal_run");
}

extern "C" int splash_dev_ncellflow(int H, int W, const double* flowdir, const double* codes9,
                                    int pad, double* ncell, void* stream)
try {
    REQ(flowdir && codes9 && ncell && H > 0 && W > 0, "ncellflow: NULL argument or empty grid");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_ncellflow(H, W, flowdir, codes9, pad, ncell, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_ncellflow");
}

// sink fill and contributing area (the rest of R/upslope_area.R: SURVEY 8f-1)
extern "C" int splash_dev_sinkfill(int H, int W, const double* elev, double resolution_m,
                                   double degree, double* filled, int* sweeps, void* stream)
try {
    REQ(elev && filled && H > 0 && W > 0, "sinkfill: NULL argument or empty grid");
    REQ(elev != filled, "sinkfill: the filled surface needs its own raster");
    REQ(resolution_m > 0 && degree >= 0 && degree < 90, "sinkfill: resolution must be positive, 0 <= degree < 90");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_sinkfill(H, W, elev, resolution_m, degree, filled, sweeps, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_sinkfill");
}

extern "C" int splash_dev_flow_accumulation(int H, int W, const double* flowdir, const double* elev,
                                            int tau_codes, const double* resolution_m,
                                            double* ups_ncell, double* ups_area, int* levels,
                                            void* stream)
try {
    REQ(flowdir && H > 0 && W > 0, "flow_accumulation: NULL flow direction or empty grid");
    REQ(ups_ncell || ups_area, "flow_accumulation: no output requested");
    REQ(!ups_area || resolution_m, "flow_accumulation: the area needs the resolution raster");
    REQ((size_t)H * W < ((size_t)1 << 32), "flow_accumulation: more than 2^32-1 cells");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_flow_accumulation(H, W, flowdir, elev, tau_codes, resolution_m, ups_ncell, ups_area, levels,
                                     (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_flow_accumulation");
}

extern "C" int splash_dev_mfd_area(int H, int W, const double* elev, double resolution_m,
                                   double* area, int* levels, void* stream)
try {
    REQ(elev && area && H > 0 && W > 0, "mfd_area: NULL argument or empty grid");
    REQ(resolution_m > 0, "mfd_area: resolution must be positive");
    REQ((size_t)H * W < ((size_t)1 << 32), "mfd_area: more than 2^32-1 cells");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_mfd_area(H, W, elev, resolution_m, area, levels, (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_mfd_area");
}

// unSWC epilogue on device-resident rasters (R/unsSWC.R:14-88, R/unsSWC.grid.R)
extern "C" int splash_dev_unswc(size_t ncell, int nt, const double* wn, size_t ldw,
                                const double* soil, double uns_depth, double* wn_z, double* wtd,
                                double* theta, size_t ldo, void* stream)
try {
    REQ(wn && soil && (wn_z || wtd || theta), "unswc: NULL argument");
    REQ(ldw >= ncell && ldo >= ncell && nt >= 0, "unswc: leading dimension < ncell");
    REQ(uns_depth > 0, "unswc: uns_depth must be positive");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_unswc(ncell, nt, wn, ldw, soil, uns_depth, wn_z, wtd, theta, ldo,
                         (hipStream_t)stream), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_unswc");
}

// unSWC.grid on device-resident rasters (R/unsSWC.grid.R:14-141: the gridded function's own semantics, splash_unswc.hpp)
extern "C" int splash_dev_unswc_grid(size_t ncell, int nt, const double* wn, size_t ldw, const double* soil,
                                     double uns_depth, double* w_z, double* wtd, double* se, double* theta_mean,
                                     size_t ldo, void* stream)
try {
    REQ(wn && soil && (w_z || wtd || se || theta_mean), "unswc_grid: NULL argument");
    REQ(ldw >= ncell && ldo >= ncell && nt >= 0, "unswc_grid: leading dimension < ncell");
    REQ(uns_depth > 0, "unswc_grid: uns_depth must be positive");
    if (int rc = enter_device(nullptr, 0)) return rc;
    HIP_TRY(launch_unswc_grid(ncell, nt, wn, ldw, soil, uns_depth, w_z, wtd, se, theta_mean, ldo, (hipStream_t)stream),
            nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_unswc_grid");
}

// test hook: SOLAR+EVAP rows (host pointers)
extern "C" int splash_dev_evap_batch(size_t m, const double* lat, const double* elv,
                                     const double* sw, const int* n, const int* y,
                                     const double* sw_in, const double* tc, const double* slop,
                                     const double* asp, const double* snow, const double* nd,
                                     double* out, void* stream)
try {
    (void)stream;
    REQ(lat && elv && sw && n && y && sw_in && tc && slop && asp && snow && nd && out,
        "evap_batch: NULL argument");
    if (int rc = enter_device(nullptr, 0)) return rc;
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
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_evap_batch");
}

// test hook: device flog/fexp/fpow/facos/fsqrt for n arguments (host pointers); out [5][n]
extern "C" int splash_dev_math_probe(size_t n, const double* x, const double* y, double* out)
try {
    REQ(x && y && out, "math_probe: NULL argument");
    if (int rc = enter_device(nullptr, 0)) return rc;
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
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_math_probe");
}

// test hook: the day loop's own moist_surf()/inf_GA() device functions (host pointers)
extern "C" int splash_dev_infiltration_probe(size_t n, const double* cellpar, const double* wn,
                                             const double* ksat_visc, const double* P,
                                             double* theta_surf, double* infiltration)
try {
    REQ(cellpar && wn && ksat_visc && P && theta_surf && infiltration,
        "infiltration_probe: NULL argument");
    if (int rc = enter_device(nullptr, 0)) return rc;
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
} catch (...) {
    return exception_code(nullptr, 0, "splash_dev_infiltration_probe");
}

extern "C" int splash_point_run_backend(int backend, int nt, const int* doy, const int* year, const int* month,
                                        const double* sw_in, const double* tc, const double* pn, double lat,
                                        double elev, double slop, double asp, const double* soil_data,
                                        const double* au, int n_au, double resolution, int monthly_out,
                                        double* const outp[SPLASH_NOUT], char* err, size_t errlen)
try {
    if (!soil_data || !outp) return fail(SPLASH_EINVAL, err, errlen, "point_run: NULL argument");
    if (backend != SPLASH_BACKEND_HIP && backend != SPLASH_BACKEND_HOST)
        return fail(SPLASH_EINVAL, err, errlen, "point_run: unknown backend %d", backend);
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
    opt.backend = backend;
    splash_grid_out o{};
    for (int k = 0; k < SPLASH_NOUT; k++) o.out[k] = outp[k];
    return splash_grid_run(&in, &opt, &o, err, errlen);
} catch (...) {
    return exception_code(err, errlen, "splash_point_run_backend");
}

extern "C" int splash_point_run(int nt, const int* doy, const int* year, const int* month,
                                const double* sw_in, const double* tc, const double* pn, double lat,
                                double elev, double slop, double asp, const double* soil_data,
                                const double* au, int n_au, double resolution, int monthly_out,
                                double* const outp[SPLASH_NOUT], char* err, size_t errlen)
try {
    return splash_point_run_backend(SPLASH_BACKEND_HIP, nt, doy, year, month, sw_in, tc, pn, lat, elev, slop, asp,
                                    soil_data, au, n_au, resolution, monthly_out, outp, err, errlen);
} catch (...) {
    return exception_code(err, errlen, "splash_point_run");
}

extern "C" int splash_host_backend_fma(void) { return host_backend_fma() ? 1 : 0; }

// unSWC / unSWC.grid for host buffers: wn [nt][ncell], soil [6][ncell] -> up to four [nt][ncell] arrays (any NULL), in
// column blocks sized to the device's free memory (a brick of 4096 x 4096 x 365 is 49 GB per array).
// grid_form = false: the site function (R/unsSWC.R: wn_z, wtd, theta); true: the gridded one (R/unsSWC.grid.R: w_z, wtd,
// Se, theta_mean) -- splash_unswc.hpp has the two semantics.
static int unswc_host_buffers(bool grid_form, const char* who, int backend, size_t ncell, int nt, const double* wn,
                              const double* soil, double uns_depth, double* const outs[4], char* err, size_t errlen)
{
    if (!wn || !soil) return fail(SPLASH_EINVAL, err, errlen, "%s: NULL input", who);
    if (!(uns_depth > 0)) return fail(SPLASH_EINVAL, err, errlen, "%s: uns_depth must be positive", who);
    if (backend != SPLASH_BACKEND_HIP && backend != SPLASH_BACKEND_HOST)
        return fail(SPLASH_EINVAL, err, errlen, "%s: unknown backend %d", who, backend);
    if (backend == SPLASH_BACKEND_HOST) { // explicit choice (ABI 6): the same functions the kernels run, on this thread
        if (ncell && nt > 0) {
            if (grid_form) host_unswc_grid(ncell, nt, wn, soil, uns_depth, outs[0], outs[1], outs[2], outs[3]);
            else host_unswc(ncell, nt, wn, soil, uns_depth, outs[0], outs[1], outs[2]);
        }
        return SPLASH_OK;
    }
    if (int rc = enter_device(err, errlen)) return rc;
    if (ncell == 0 || nt <= 0) return SPLASH_OK;
    int n_out = 0;
    for (int k = 0; k < 4; k++) n_out += outs[k] != nullptr;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b), err, errlen);
    int cur_dev = 0;
    HIP_TRY(hipGetDevice(&cur_dev), err, errlen);
    free_b += pool_idle_bytes(cur_dev);
    if (knobs().fake_device) free_b = knobs().fake_device_bytes; // SPLASH_HIP_FAKE_DEVICE_BYTES (tests: several column blocks)
    const size_t per_cell = (size_t)nt * 8 * (size_t)(1 + n_out) + 6 * 8;
    size_t nb = std::min<size_t>(ncell, std::max<size_t>(256, (size_t)((double)free_b * 0.5 / (double)per_cell)));
    nb = std::min(nb, kMaxCellsPerLaunch);
    DevBuf d_wn, d_soil, d_out[4];
    HIP_TRY(d_wn.alloc((size_t)nt * nb * 8), err, errlen);
    HIP_TRY(d_soil.alloc(6 * nb * 8), err, errlen);
    for (int k = 0; k < 4; k++)
        if (outs[k]) HIP_TRY(d_out[k].alloc((size_t)nt * nb * 8), err, errlen);
    auto dp = [&](int k) { return outs[k] ? d_out[k].as<double>() : nullptr; };
    for (size_t c0 = 0; c0 < ncell; c0 += nb) {
        const size_t n = std::min(nb, ncell - c0);
        HIP_TRY(hipMemcpy2D(d_wn.p, n * 8, wn + c0, ncell * 8, n * 8, (size_t)nt, hipMemcpyHostToDevice), err, errlen);
        HIP_TRY(hipMemcpy2D(d_soil.p, n * 8, soil + c0, ncell * 8, n * 8, 6, hipMemcpyHostToDevice), err, errlen);
        if (grid_form)
            HIP_TRY(launch_unswc_grid(n, nt, d_wn.as<double>(), n, d_soil.as<double>(), uns_depth, dp(0), dp(1), dp(2), dp(3),
                                      n, nullptr), err, errlen);
        else
            HIP_TRY(launch_unswc(n, nt, d_wn.as<double>(), n, d_soil.as<double>(), uns_depth, dp(0), dp(1), dp(2), n, nullptr),
                    err, errlen);
        HIP_TRY(hipDeviceSynchronize(), err, errlen);
        for (int k = 0; k < 4; k++)
            if (outs[k])
                HIP_TRY(hipMemcpy2D(outs[k] + c0, ncell * 8, d_out[k].p, n * 8, n * 8, (size_t)nt, hipMemcpyDeviceToHost), err,
                        errlen);
    }
    return SPLASH_OK;
}

extern "C" int splash_unswc_run_backend(int backend, size_t ncell, int nt, const double* wn, const double* soil,
                                        double uns_depth, double* wn_z, double* wtd, double* theta,
                                        char* err, size_t errlen)
try {
    double* const outs[4] = {wn_z, wtd, theta, nullptr};
    return unswc_host_buffers(false, "unswc_run", backend, ncell, nt, wn, soil, uns_depth, outs, err, errlen);
} catch (...) {
    return exception_code(err, errlen, "splash_unswc_run");
}

extern "C" int splash_unswc_run(size_t ncell, int nt, const double* wn, const double* soil, double uns_depth, double* wn_z,
                                double* wtd, double* theta, char* err, size_t errlen)
{
    return splash_unswc_run_backend(SPLASH_BACKEND_HIP, ncell, nt, wn, soil, uns_depth, wn_z, wtd, theta, err, errlen);
}

extern "C" int splash_unswc_grid_run(int backend, size_t ncell, int nt, const double* wn, const double* soil,
                                     double uns_depth, double* w_z, double* wtd, double* se, double* theta_mean,
                                     char* err, size_t errlen)
try {
    double* const outs[4] = {w_z, wtd, se, theta_mean};
    return unswc_host_buffers(true, "unswc_grid_run", backend, ncell, nt, wn, soil, uns_depth, outs, err, errlen);
} catch (...) {
    return exception_code(err, errlen, "splash_unswc_grid_run");
}

// K2 on host buffers (ABI 5): the snowfall threshold of a tc series, for callers that process a long
// series in time chunks -- the threshold of the whole series is the maximum of the chunks' (R/splash.point.R:120-122).
extern "C" int splash_snow_threshold_run_backend(int backend, size_t ncell, int nt, const double* tc, const float* tc_f32,
                                                 const double* lat, const double* elev, double* tt, char* err,
                                                 size_t errlen)
try {
    if (!lat || !elev || !tt || ((tc != nullptr) == (tc_f32 != nullptr)) || nt < 1)
        return fail(SPLASH_EINVAL, err, errlen,
                    "snow_threshold_run: NULL argument, nt < 1, or not exactly one of tc / tc_f32");
    if (backend != SPLASH_BACKEND_HIP && backend != SPLASH_BACKEND_HOST)
        return fail(SPLASH_EINVAL, err, errlen, "snow_threshold_run: unknown backend %d", backend);
    if (backend == SPLASH_BACKEND_HOST) { // explicit choice (ABI 6): a series in time chunks without a GPU
        host_snow_threshold(ncell, nt, tc, tc_f32, lat, elev, tt);
        return SPLASH_OK;
    }
    if (int rc = enter_device(err, errlen)) return rc;
    if (ncell == 0) return SPLASH_OK;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b), err, errlen);
    int cur_dev = 0;
    HIP_TRY(hipGetDevice(&cur_dev), err, errlen);
    free_b += pool_idle_bytes(cur_dev);
    const size_t per_cell = (size_t)nt * (tc ? 8 : 12) + 3 * 8;
    size_t nb = std::min<size_t>(ncell, std::max<size_t>(256, (size_t)((double)free_b * 0.5 / (double)per_cell)));
    nb = std::min(nb, kMaxCellsPerLaunch);
    DevBuf d_tc, d_f32, d_par, d_tt;
    HIP_TRY(d_tc.alloc((size_t)nt * nb * 8), err, errlen);
    if (tc_f32) HIP_TRY(d_f32.alloc((size_t)nt * nb * 4), err, errlen);
    HIP_TRY(d_par.alloc(2 * nb * 8), err, errlen); // rows P_LAT, P_ELV: all the kernel reads
    HIP_TRY(d_tt.alloc(nb * 8), err, errlen);
    static_assert(P_LAT == 0 && P_ELV == 1, "snow_threshold_run hands the kernel a two-row parameter block");
    for (size_t c0 = 0; c0 < ncell; c0 += nb) {
        const size_t n = std::min(nb, ncell - c0);
        if (tc) {
            HIP_TRY(hipMemcpy2D(d_tc.p, n * 8, tc + c0, ncell * 8, n * 8, (size_t)nt, hipMemcpyHostToDevice), err, errlen);
        } else {
            HIP_TRY(hipMemcpy2D(d_f32.p, n * 4, tc_f32 + c0, ncell * 4, n * 4, (size_t)nt, hipMemcpyHostToDevice), err, errlen);
            HIP_TRY(launch_widen_f32(d_f32.as<float>(), (size_t)nt * n, d_tc.as<double>(), nullptr), err, errlen);
        }
        HIP_TRY(hipMemcpy(d_par.as<double>(), lat + c0, n * 8, hipMemcpyHostToDevice), err, errlen);
        HIP_TRY(hipMemcpy(d_par.as<double>() + n, elev + c0, n * 8, hipMemcpyHostToDevice), err, errlen);
        HIP_TRY(launch_snow_threshold(n, nt, d_tc.as<double>(), n, d_par.as<double>(), n, d_tt.as<double>(), nullptr),
                err, errlen);
        HIP_TRY(hipMemcpy(tt + c0, d_tt.p, n * 8, hipMemcpyDeviceToHost), err, errlen);
    }
    return SPLASH_OK;
} catch (...) {
    return exception_code(err, errlen, "splash_snow_threshold_run");
}

extern "C" int splash_snow_threshold_run(size_t ncell, int nt, const double* tc, const float* tc_f32, const double* lat,
                                         const double* elev, double* tt, char* err, size_t errlen)
{
    return splash_snow_threshold_run_backend(SPLASH_BACKEND_HIP, ncell, nt, tc, tc_f32, lat, elev, tt, err, errlen);
}

// static rasters on host buffers: R/splash.grid.R:96-110 + upslope_area() (R/upslope_area.R:12-58)
extern "C" int splash_static_rasters_run(int H, int W, const double* elev, double ymax, double yres,
                                         double xres, int na_to_zero, double degree,
                                         const splash_static_out* out, char* err, size_t errlen)
try {
    if (!elev || !out || H <= 0 || W <= 0)
        return fail(SPLASH_EINVAL, err, errlen, "static_rasters_run: NULL argument or empty grid");
    if (int rc = enter_device(err, errlen)) return rc;
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
} catch (...) {
    return exception_code(err, errlen, "splash_static_rasters_run");
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

// backend of the calling thread's splash_mod_* calls (ABI 6)
static thread_local int t_mod_backend = SPLASH_BACKEND_HIP;
extern "C" int splash_mod_backend(int backend)
{
    if (backend != SPLASH_BACKEND_HIP && backend != SPLASH_BACKEND_HOST)
        return fail(SPLASH_EINVAL, nullptr, 0, "mod_backend: unknown backend %d", backend);
    const int prev = t_mod_backend;
    t_mod_backend = backend;
    return prev;
}

// SPLASH::spin_up (src/SPLASH.cpp:1594-1749; R call at R/splash.point.R:148,152)
extern "C" int splash_mod_spin_up(double lat, double elv, int n, int y, const double* sw_in,
                                  const double* tair, const double* pn, double slop, double asp,
                                  const double* snowfall, const double* soil_info, double* sm,
                                  double* snow, double* qin, double* tdrain, double* ro,
                                  double* snwage, double* pet, int* passes)
try {
    REQ(n > 0 && sw_in && tair && pn && snowfall && soil_info, "mod_spin_up: bad argument");
    const bool on_host = t_mod_backend == SPLASH_BACKEND_HOST;
    if (!on_host)
        if (int rc = enter_device(nullptr, 0)) return rc;
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
    if (on_host) { // the same device functions, compiled for the host (splash_host_backend.hip)
        double* series[7] = {sm, snow, qin, tdrain, ro, snwage, pet}; // names of :1748
        if (pet) std::fill(pet, pet + n, 0.0);
        host_mod_spin_up(par, n, days.data(), sw_in, tair, pn, snowfall, series, passes);
        return SPLASH_OK;
    }
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
} catch (...) {
    return exception_code(nullptr, 0, "splash_mod_spin_up");
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
try {
    REQ(nt > 0 && doys && yrs && sw_in && tair && pn && snowfall && soil_info,
        "mod_run_all: bad argument");
    const bool on_host = t_mod_backend == SPLASH_BACKEND_HOST;
    if (!on_host)
        if (int rc = enter_device(nullptr, 0)) return rc;
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
    if (on_host) {
        double* dsth[11] = {wn, ro, pet, aet, snow, cond, bflow, netr, qin_prev, tdrain, snwage};
        host_mod_run_all(par, nt, days.data(), sw_in, tair, pn, snowfall, st0, dsth);
        return SPLASH_OK;
    }
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
} catch (...) {
    return exception_code(nullptr, 0, "splash_mod_run_all");
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
try {
    REQ(n > 0 && n_end >= n && wn_vec && snow && qin_vec && td_vec && nds,
        "mod_run_one_year: bad argument");
    std::vector<int> doys(n), yrs(n, y);
    for (int i = 0; i < n; i++) doys[i] = i + 1;
    return splash_mod_run_all(lat, elv, n, doys.data(), yrs.data(), sw_in, tair, pn,
                              wn_vec[n_end - 1], slop, asp, snow[n_end - 1], snowfall, soil_info,
                              qin_vec[n_end - 1], td_vec[n_end - 1], nds[n_end - 1], wn_o, ro, pet,
                              aet, snow_o, cond, bflow, netr, qin_prev, tdrain, snwage);
} catch (...) {
    return exception_code(nullptr, 0, "splash_mod_run_one_year");
}

extern "C" int splash_mod_moist_surf(double depth, double z, double bub_p, double wn, double SAT,
                                     double RES, double lambda, double* theta)
try {
    REQ(theta, "mod_moist_surf: NULL result");
    if (t_mod_backend == SPLASH_BACKEND_HOST) {
        *theta = host_mod_moist_surf(depth, z, bub_p, wn, SAT, RES, lambda);
        return SPLASH_OK;
    }
    if (int rc = enter_device(nullptr, 0)) return rc;
    DevBuf d;
    HIP_TRY(d.alloc(8), nullptr, 0);
    HIP_TRY(launch_moist_surf(depth, z, bub_p, wn, SAT, RES, lambda, d.as<double>(), nullptr),
            nullptr, 0);
    HIP_TRY(hipMemcpy(theta, d.p, 8, hipMemcpyDeviceToHost), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_mod_moist_surf");
}

extern "C" int splash_mod_inf_GA(double bub_press, double theta_i, double Ksat, double theta_s,
                                 double lambda, double P, double tdur, double slop,
                                 double* infiltration)
try {
    REQ(infiltration, "mod_inf_GA: NULL result");
    if (t_mod_backend == SPLASH_BACKEND_HOST) {
        *infiltration = host_mod_inf_GA(bub_press, theta_i, Ksat, theta_s, lambda, P, tdur, slop);
        return SPLASH_OK;
    }
    if (int rc = enter_device(nullptr, 0)) return rc;
    DevBuf d;
    HIP_TRY(d.alloc(8), nullptr, 0);
    HIP_TRY(launch_inf_GA(bub_press, theta_i, Ksat, theta_s, lambda, P, tdur, slop, d.as<double>(),
                          nullptr), nullptr, 0);
    HIP_TRY(hipMemcpy(infiltration, d.p, 8, hipMemcpyDeviceToHost), nullptr, 0);
    return SPLASH_OK;
} catch (...) {
    return exception_code(nullptr, 0, "splash_mod_inf_GA");
}

// ------------------------------------------------------------------ test hooks named in SURVEY.md 8(b)
// ONE day step of one cell = SPLASH::run_one_day (src/SPLASH.cpp:920-1588): splash_mod_run_all with nt = 1, on the
// calling thread's module backend (splash_mod_backend).  state_in: wn, snow, qin, td, nd of the day before;
// out[11]: wn, ro, pet, aet, snow, cond, bflow, netr, qin, tdrain, snwage.
extern "C" int splash_day_step(double lat, double elv, int n, int y, double sw_in, double tc, double pn, double snowfall,
                               double slop, double asp, const double* soil_info, const double* state_in, double* out)
{
    if (!soil_info || !state_in || !out) return fail(SPLASH_EINVAL, nullptr, 0, "day_step: NULL argument");
    return splash_mod_run_all(lat, elv, 1, &n, &y, &sw_in, &tc, &pn, state_in[0], slop, asp, state_in[1], &snowfall,
                              soil_info, state_in[2], state_in[3], state_in[4], out + 0, out + 1, out + 2, out + 3, out + 4,
                              out + 5, out + 6, out + 7, out + 8, out + 9, out + 10);
}

// soil_hydro + the soil_info vector of R/splash.point.R:96-115, 232-416 for host arrays: cellpar [SPLASH_NPAR][ncell]
// (K1 alone), on the calling thread's module backend.  au [3][ncell] or NULL (Au = 0, 3 cells in, 3 out: :106-110).
extern "C" int splash_soil_hydro(size_t ncell, const double* lat, const double* elev, const double* slop, const double* asp,
                                 const double* resolution, const double* soil, const double* au, double* cellpar, char* err,
                                 size_t errlen)
try {
    if (!lat || !elev || !slop || !asp || !resolution || !soil || !cellpar)
        return fail(SPLASH_EINVAL, err, errlen, "soil_hydro: NULL argument");
    if (ncell == 0) return SPLASH_OK;
    if (t_mod_backend == SPLASH_BACKEND_HOST) {
        host_soil_hydro(ncell, lat, elev, slop, asp, resolution, soil, au, cellpar);
        return SPLASH_OK;
    }
    if (int rc = enter_device(err, errlen)) return rc;
    DevBuf d_in, d_par;
    const size_t nin = 5 + 6 + (au ? 3 : 0);
    HIP_TRY(d_in.alloc(nin * ncell * 8), err, errlen);
    HIP_TRY(d_par.alloc((size_t)SPLASH_NPAR * ncell * 8), err, errlen);
    double* p = d_in.as<double>();
    const double* src[5] = {lat, elev, slop, asp, resolution};
    for (int k = 0; k < 5; k++) HIP_TRY(hipMemcpy(p + (size_t)k * ncell, src[k], ncell * 8, hipMemcpyHostToDevice), err, errlen);
    HIP_TRY(hipMemcpy(p + 5 * ncell, soil, 6 * ncell * 8, hipMemcpyHostToDevice), err, errlen);
    if (au) HIP_TRY(hipMemcpy(p + 11 * ncell, au, 3 * ncell * 8, hipMemcpyHostToDevice), err, errlen);
    HIP_TRY(launch_soil_hydro(ncell, p, p + ncell, p + 2 * ncell, p + 3 * ncell, p + 4 * ncell, p + 5 * ncell,
                              au ? p + 11 * ncell : nullptr, d_par.as<double>(), nullptr), err, errlen);
    HIP_TRY(hipMemcpy(cellpar, d_par.p, (size_t)SPLASH_NPAR * ncell * 8, hipMemcpyDeviceToHost), err, errlen);
    return SPLASH_OK;
} catch (...) {
    return exception_code(err, errlen, "splash_soil_hydro");
}
