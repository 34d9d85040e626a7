// splash_kernels.hip -- gfx950 kernels of the gridded SPLASH loop and their launchers.
//
// Launch shape everywhere: one lane = one grid cell, 256-lane workgroups (4 wavefronts), grid =
// ceil(ncell/256).  Rasters are time-major [nt][ld] so that the 64 lanes of a wavefront read 512
// contiguous bytes per series per day; the per-day table is read through scalar loads (uniform
// index).  Lanes past ncell stay alive (clamped index, masked stores) so wave-level votes and
// reductions always see full waves.
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "splash_kernels.hpp"

#include "splash_device.hpp"

namespace splash {

static constexpr int kBlock = 256;


// minimum waves per SIMD requested from the register allocator for the two day-loop kernels
#ifndef SPLASH_RUNALL_WPE
#define SPLASH_RUNALL_WPE 2
#endif

// ================================================================ K1 soil_hydro ===============
// R/splash.point.R:232-416 (soil_hydro) + :96-115 (soil_info) + :131 (asp-180)
__global__ __launch_bounds__(kBlock) void k_soil_hydro(
    size_t ncell, const double* __restrict__ lat, const double* __restrict__ elev,
    const double* __restrict__ slop, const double* __restrict__ asp,
    const double* __restrict__ resolution, const double* __restrict__ soil,
    const double* __restrict__ au, double* __restrict__ par)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= ncell) return;
    double col[P_NPAR];
    soil_hydro_cell(lat[i], elev[i], slop[i], asp[i], resolution[i], soil[i], soil[ncell + i], soil[2 * ncell + i],
                    soil[3 * ncell + i], soil[4 * ncell + i], soil[5 * ncell + i],
                    au ? au[i] : 0.0, au ? au[ncell + i] : 3.0, au ? au[2 * ncell + i] : 3.0, col);
#pragma unroll
    for (int k = 0; k < P_NPAR; k++) par[(size_t)k * ncell + i] = col[k];
}

// ================================================================ K2 snow threshold ===========
// Tt = max(tc[p_snow >= 0.5]) over the whole series, R/splash.point.R:120-122
__global__ __launch_bounds__(kBlock) void k_snow_threshold(size_t ncell, int nt,
                                                          const double* __restrict__ tc,
                                                          size_t ldf,
                                                          const double* __restrict__ par,
                                                          size_t ldp, double* __restrict__ Tt)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= ncell) return;
    const double e1 = par[P_ELV * ldp + i] * 0.0004596581, e2 = fabs(par[P_LAT * ldp + i]) * 0.0110592101;
    double best = -__builtin_inf();
    bool gap = false;
#pragma unroll 4
    for (int t = 0; t < nt; t++) {
        double v = __builtin_nontemporal_load(&tc[(size_t)t * ldf + i]);
        // p_snow >= 0.5  <=>  the logistic's exponent <= 0 (see rain_snow_split)
        const double x = snow_logit(v, e1, e2);
        if (x <= kSnowLogitMax && v > best) best = v;
        gap |= (x != x); // a missing tc makes R's max() NA for the whole series
    }
    Tt[i] = gap ? __builtin_nan("") : best;
}

// ================================================================ forcing access ==============
// One lane's view of the forcing rasters.  The row base (array + t*ld) is uniform and stays in
// scalar registers; the lane adds a 32-bit byte offset, so a load is one global_load with an SGPR
// base and no 64-bit vector address arithmetic (callers keep ncell*8 below 4 GiB per launch).
// The snowfall threshold of a lane's cell, for kernels that keep it in LDS rather than in a register
// (k_run_all: it is needed on possible-snow days only, the day loop is short of registers, and a
// register the compiler spills comes back through scratch memory, i.e. through the vector-memory
// counter: the reload would wait for every outstanding load, the prefetched forcing included).
__shared__ double s_cell_Tt[kBlock];

struct Forcing {
    const double *sw, *tc, *pn, *snowfall; // snowfall != NULL => explicit rain/snow series
    size_t ld;
    double Tt, e1, e2; // snowfall threshold; elev*0.0004596581, |lat|*0.0110592101 (snow_logit)
    unsigned off; // cell * sizeof(double)
    static __device__ __forceinline__ double at(const double* row, unsigned off_bytes)
    {
        // non-temporal: the forcing streams through once per pass and must not push the viscosity
        // table (7.5 MB, gathered by every lane every day) out of the 4 MB L2 of an XCD -- with
        // default-policy loads the gathers added 0.76 GB of fabric reads per 512x4096x365 launch
        // (FETCH_SIZE, tools/pmc_fetch.sh), with non-temporal ones 0.09 GB
        return __builtin_nontemporal_load(
            reinterpret_cast<const double*>(reinterpret_cast<const char*>(row) + off_bytes));
    }
    __device__ __forceinline__ void raw(int t, double& a, double& b, double& c_) const
    {
        const size_t r = (size_t)t * ld;
        a = at(sw + r, off);
        b = at(tc + r, off);
        c_ = at(pn + r, off);
    }
    // the day's record of the explicit snowfall series (0 when the split is computed): requested
    // together with the forcing, a day ahead, so that no load is issued -- and none has to be
    // waited for -- in the first half of a day
    __device__ __forceinline__ double raw_snowfall(int t) const
    {
        return snowfall ? at(snowfall + (size_t)t * ld, off) : 0.0;
    }
    template <bool TT_IN_LDS = false>
    __device__ __forceinline__ void split(const DayRec& d, double tcv, double pnv, double snfv,
                                          double& rain, double& snowf) const
    {
        if (snowfall) {
            rain = pnv;
            snowf = snfv;
        } else {
            DayConst dc{d.dr, d.sd, d.cd, d.ttm_s, d.inv_trm14};
            if (TT_IN_LDS)
                rain_snow_split(dc, tcv, pnv, [] {
                    // the index passes through an asm statement, which stays inside the snow branch:
                    // the read cannot be hoisted into a register (a volatile read would be a flat load)
                    unsigned i = threadIdx.x;
                    asm volatile("" : "+v"(i));
                    return s_cell_Tt[i];
                }, e1, e2, rain, snowf);
            else
                rain_snow_split(dc, tcv, pnv, [this] { return Tt; }, e1, e2, rain, snowf);
        }
    }
    // the same with the snowfall record read on the spot (spin-up, module mirrors)
    __device__ __forceinline__ void split(const DayRec& d, int t, double tcv, double pnv,
                                          double& rain, double& snowf) const
    {
        split(d, tcv, pnv, raw_snowfall(t), rain, snowf);
    }
};

// The per-day table is read-only for the whole launch: going through the constant address space
// lets the (uniform) loads be scalar loads into SGPRs instead of 64-lane vector loads followed by
// v_readfirstlane.
typedef const __attribute__((address_space(4))) DayRec* DayTable;
__device__ __forceinline__ DayTable day_table(const DayRec* p)
{
    return (DayTable)(unsigned long long)p;
}
__device__ __forceinline__ DayRec day_rec(DayTable tab, int t)
{
    DayRec d;
    d.dr = tab[t].dr;
    d.sd = tab[t].sd;
    d.cd = tab[t].cd;
    d.ttm_s = tab[t].ttm_s;
    d.inv_trm14 = tab[t].inv_trm14;
    d.month_id = tab[t].month_id;
    d.mdays = tab[t].mdays;
    d.flags = tab[t].flags;
    d.doy = tab[t].doy;
    return d;
}

__device__ __forceinline__ DayConst day_const(const DayRec& d)
{
    return DayConst{d.dr, d.sd, d.cd, d.ttm_s, d.inv_trm14};
}

__device__ __forceinline__ bool same_bits(double a, double b)
{
    return __double_as_longlong(a) == __double_as_longlong(b); // NaN-safe (sticky NaN drainage time)
}

#ifdef SPL_DIFF_STATS
// measurement build (tools/diff_stats.py): |end_sm - start_sm| of the first eight passes of spin-up #2, per cell
__device__ float* g_diff_dbg = nullptr;
extern "C" int splash_dbg_set_diff(float* p)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_diff_dbg), &p, sizeof p);
}
#endif

#ifdef SPL_CHECK_SLOTS
// checking build (librsplash_hip_check.so; tests/test_gpu_host_path.py::test_read_back_never_addresses_a_slot_beyond_the_cache):
// every access to the spin-up's record cache reports its slot.  [0] wavefront-level accesses, [1] lane accesses
// whose slot lies beyond the capacity the launch was given -- must stay 0.
__device__ unsigned long long g_slot_check[2];
__device__ __forceinline__ void slot_check(size_t slot, size_t cap)
{
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_slot_check[0], 1ull);
    if (slot >= cap) atomicAdd(&g_slot_check[1], 1ull);
}
extern "C" int splash_dbg_slot_check(unsigned long long out[2], int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slot_check), 2 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long z[2] = {0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_slot_check), z, sizeof z);
    }
    return (int)e;
}
#define SPL_SLOT_CHECK(slot, cap) slot_check(slot, cap)
#else
#define SPL_SLOT_CHECK(slot, cap) ((void)0)
#endif

// ================================================================ K4 spin-up ===================
// Grid spin-up: both spin-ups of splash.point (R/splash.point.R:141-152) as a resumable per-cell
// job.  The number of passes is data dependent (2..1000, src/SPLASH.cpp:1697): most cells need
// 2-5, a few per thousand never meet the 1 mm criterion and run all 1000.  A wavefront costs as
// much as its slowest lane, so the work is done in phases: every phase gives each job a pass
// budget, finished cells write their state, unfinished ones park their job (stage, pass count,
// day-1 state) and append their cell id to a worklist; the next phase runs on the compacted
// list, so the long-running cells end up packed into few dense waves.
//   job stage 1: spin-up #1 (cellout as given); its first pass also yields the aridity index
//   job stage 2: spin-up #2 with cellout := AI (R/splash.point.R:150)
// One cached record: the state-independent terms of a cell-day (DayTerms) + the day's rain and
// snowfall, 15 words; the cache is [365][kNRec][cap] doubles, slot = position of the job in the
// worklist, so that the 64 lanes of a wavefront read 512 contiguous bytes per field.
// Two layouts.  Compact (15 words = 120 B): what phase 0 streams for every cell of the grid -- its
// fill pass is bound by the bytes it stores.  Rich (19 words): the compact record plus rv and the
// three numbers visc_terms() derives, for the later phases -- short worklists and, at the end, lone
// wavefronts whose every instruction is latency: there the twenty instructions (one reciprocal) and
// the scalar load saved per day step are worth far more than the bytes (measured on the tail:
// 1 272 ms with compact records, 1 160 ms with rich ones).
// Compact records are laid out field by field ([day][field][slot]: the 64 lanes of a wavefront read
// 512 contiguous bytes per field -- phase 0 is a stream over every cell).  Rich records are laid out
// record by record ([day][slot][20 words], the 20th is padding): a lane reads its 160 contiguous bytes
// with ten 16-byte loads off ONE address instead of nineteen 8-byte loads off nineteen -- 57 fewer
// instructions per day step (the address arithmetic was 34 scalar and 17 vector instructions), which
// is what counts where a lone wavefront pays for every instruction; coalescing is no concern there.
constexpr int kNRecCompact = kNTermWords + 2;
constexpr int kNRecRich = kNRecCompact + 4; // (words that carry data; kRichStride in memory)
static_assert(kNRecRich <= 20, "rich record");
constexpr int kRichStride = 20; // words per rich record in memory (16-byte aligned pairs)
// Round 4: rich records are laid out JOB by job, [slot][day][20]: a job's year is 58 KB of consecutive memory (a
// page or two), where day by day ([day][slot][20], round 3) the next day's record lay cap x 160 B = 63 MB further
// on, a new page every day.  (It made no measurable difference to the tail -- a day step leaves a load 1.6 us --
// but it is the layout a job alone on its wavefront wants, and the one a prefetcher can follow.)
constexpr size_t kRichDays = 365;
struct DayRecord {
    DayTerms T;
    ViscTerms V;
    double rain, snf;
};
template <bool RICH>
__device__ __forceinline__ void rec_store(double* __restrict__ base, size_t cap, int day, size_t slot,
                                          const DayRecord& r)
{
    // the viscosity is a float: its 32 bits and the melt flag share one word, exactly
    SPL_SLOT_CHECK(slot, cap);
    const unsigned long long vw = (unsigned long long)__float_as_uint((float)r.T.visc) |
                                  ((unsigned long long)(r.T.warm ? 1u : 0u) << 32);
    const double v[kRichStride] = {r.T.ru, r.T.ruv, r.T.hs, r.T.sin_hs, r.T.xrw, r.T.rnl, r.T.s, r.T.g,
                                   r.T.econ, r.T.inv_econ, r.T.eetc, r.T.pw, __longlong_as_double((long long)vw),
                                   r.rain, r.snf, r.T.rv, r.V.ipk, r.V.kv, r.V.ikv24, 0.0};
    if constexpr (RICH) {
        double2* p = reinterpret_cast<double2*>(base + (slot * kRichDays + (size_t)day) * kRichStride);
#pragma unroll
        for (int k = 0; k < kRichStride / 2; k++) p[k] = double2{v[2 * k], v[2 * k + 1]};
    } else {
        double* p = base + ((size_t)day * kNRecCompact) * cap + slot;
#pragma unroll
        for (int k = 0; k < kNRecCompact; k++) p[(size_t)k * cap] = v[k];
    }
}
// Loading is split from decoding: the words of tomorrow's record are requested a day ahead and must
// not be touched before tomorrow (anything computed from them at the point of the load -- the
// viscosity's conversion, rv, visc_terms() -- makes the wavefront wait for the load right there, a
// full memory round trip per day: +17 % on the tail when this was one function).
template <bool RICH>
struct RawRecord {
    double v[RICH ? kRichStride : kNRecCompact];
};
template <bool RICH>
__device__ __forceinline__ RawRecord<RICH> rec_load(const double* __restrict__ base, size_t cap, int day, size_t slot)
{
    RawRecord<RICH> w;
    SPL_SLOT_CHECK(slot, cap);
    if constexpr (RICH) {
        const double2* p = reinterpret_cast<const double2*>(base + (slot * kRichDays + (size_t)day) * kRichStride);
#pragma unroll
        for (int k = 0; k < kRichStride / 2; k++) {
            const double2 x = p[k];
            w.v[2 * k] = x.x;
            w.v[2 * k + 1] = x.y;
        }
    } else {
        const double* p = base + ((size_t)day * kNRecCompact) * cap + slot;
#pragma unroll
        for (int k = 0; k < kNRecCompact; k++) w.v[k] = p[(size_t)k * cap];
    }
    return w;
}
// Compact: rv = cos(delta) * kb of the record's day (one multiplication of a per-day scalar and a cell
// constant, the same one day_terms() does, pinned like the loaded terms) and visc_terms() are redone.
template <bool RICH>
__device__ __forceinline__ DayRecord rec_decode(const CellConst& c, const RawRecord<RICH>& w, double cd)
{
    const double* v = w.v;
    DayRecord r;
    r.T.ru = v[0]; r.T.ruv = v[1]; r.T.hs = v[2]; r.T.sin_hs = v[3]; r.T.xrw = v[4]; r.T.rnl = v[5];
    r.T.s = v[6]; r.T.g = v[7]; r.T.econ = v[8]; r.T.inv_econ = v[9]; r.T.eetc = v[10]; r.T.pw = v[11];
    r.rain = v[13];
    r.snf = v[14];
    if constexpr (!RICH) {
        const unsigned long long vw = (unsigned long long)__double_as_longlong(v[12]);
        r.T.visc = (double)__uint_as_float((unsigned)vw);
        r.T.warm = ((vw >> 32) & 1ull) != 0;
    }
    if constexpr (RICH) {
        r.T.visc = 0.0;            // (its three derived terms are in the record)
        r.T.warm = (v[16] != 0.0); // ipk = 1/(pw*kfus) > 0 exactly when tc >= 3
        r.T.rv = v[15];
        r.V.ipk = v[16];
        r.V.kv = v[17];
        r.V.ikv24 = v[18];
    } else {
        double rv = cd * c.kb;
        asm volatile("" : "+v"(rv));
        r.T.rv = rv;
        r.V = visc_terms(c, r.T);
    }
    return r;
}

// CACHED: the first pass of the launch stores every day's record, the later ones read them back and
// only evaluate the state-dependent update -- the same bits (terms_fence), about two thirds of the
// instructions, and no forcing reads.  Used by phase 0 (every cell, in blocks the size of the cache:
// the pass of spin-up #1 fills it, the passes of spin-up #2 read it) and by the later phases; jobs
// beyond the cache's capacity run the plain way.
// (A single day loop 0, 1, ..., 364, 0, ... with the pass-end decision after every day 0 -- one
// inlined copy of the day step instead of four -- was tried: the turn a lane needs when it starts
// spin-up #2 in the middle of that loop cost more than the smaller code saved: 2.2-4.4 s vs 1.35 s
// for 2.1 M cells.)
// READBACK: a kernel of its own for the long tail -- jobs whose 365 records are in the cache already
// (a preceding one-pass launch of the general kernel stored them): no forcing, no day terms, no
// viscosity tables, i.e. a fraction of the code and of the live values of the general kernel.  The
// tail is ~1000 passes x 365 sequential day steps of lone wavefronts, every instruction of which is
// latency; compiled together with the general loop its day step carried ~100 v_readlane (scalar
// registers spilled to vector lanes) and ~300 s_mov per day that the smaller kernel does not need.
template <bool CACHED, bool READBACK = false, bool RICH = false>
__global__ __launch_bounds__(kBlock, SPLASH_RUNALL_WPE) void k_spin_phase(SpinPhaseArgs a)
{
#if defined(SPL_TAIL_STATS) && defined(SPL_DIFF_STATS)
    const unsigned long long t_wave_start = wall_clock64();
#endif
    size_t nin = a.n_in ? (size_t)*a.n_in : a.ncell;
    // A spread launch over the tail's own list (no wl_slots: list position == cache slot) deals only the
    // positions the host sized the launch for, a.ncell = min(list bound, cache capacity): positions beyond
    // the cache have no records here and belong to the general-kernel launch that follows (spin_phases).
    if (READBACK && a.spread && !a.wl_slots && nin > a.ncell) nin = a.ncell;
    // Launch position -> job.  Packed (default): 64 jobs to a wavefront.  Spread (the tail's stages, READBACK
    // only): the jobs are dealt over 1024 wavefronts -- one per SIMD of the chip --, k = ceil(n / 1024) to each.
    // A wavefront executes the union of the branches of its lanes: a job that has the wavefront to itself skips
    // the rare branches (ponding, deep column, second transmittance, drainage time, snow) on the days IT does not
    // take them and runs a pass of the year in half the time of a job packed with 63 others (measured: 530
    // against 1 035 ms for the 971 passes of the longest chains, tools/solo_tail.py, profiles/r04_spinup.md).
    size_t slot;
    bool valid, writer = true;
    if (READBACK && a.spread) {
        size_t k = (nin + 1023) / 1024;
        k = k < 1 ? 1 : (k > 64 ? 64 : k);
        const size_t wave = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
        if ((size_t)blockIdx.x * (kBlock / 64) * k >= nin) return; // whole workgroup past the list
        // The other lanes of the wavefront REPEAT its k jobs (lane % k) -- same data, same branches, nothing
        // added to what the wavefront executes -- instead of leaving: a wavefront with one active lane ran the
        // same job in 600 ms on some CUs and in up to 1 590 ms on others, with all 64 lanes active in 576-580 ms
        // on every CU; the threshold is 16 active lanes (tools/solo_tail.py SOLO_SAME / ACTIVE_LANES; no
        // micro-benchmark of ours reproduces it: tools/ubench/cu_variance*).  Only the first k lanes store (`writer`).
        slot = wave * k + ((threadIdx.x & 63) % k);
        if (slot >= nin) slot = wave * k;
        valid = slot < nin;
        writer = (threadIdx.x & 63) < k && wave * k + (threadIdx.x & 63) < nin;
    } else {
        // (first_dev: the launch covers the list from a position known on the device only -- the survivors a
        // block of phase 0 has just appended, see spin_phases)
        const size_t first = a.slot0 + (a.first_dev ? (size_t)*a.first_dev : 0);
        if ((size_t)blockIdx.x * kBlock + first >= nin) return; // whole workgroup past the list
        slot = (size_t)blockIdx.x * kBlock + threadIdx.x + first;
        valid = slot < nin;
    }
    lds_tables_init();
    if (READBACK && a.spread && !valid) return; // (a wavefront past the list; after the tables' barrier)
#if defined(SPL_TAIL_STATS) && defined(SPL_DIFF_STATS)
    if (READBACK && a.spread && a.dbg_active_lanes > 0 && (int)(threadIdx.x & 63) >= a.dbg_active_lanes) return; // measurement: how many lanes must stay?
#endif
    // The tail is the critical path of a call and a handful of wavefronts; whatever shares a SIMD with them (the
    // transient run of the finished cells, launched beside it) is throughput work: the tail's wavefronts issue first.
    // (Only the spread launches: the packed read-back launches of phases 1-6 are throughput work themselves --
    // up to 4e5 jobs -- and must not out-prioritise another lane's wavefronts on a shared SIMD.)
    if (READBACK && RICH && a.spread) __builtin_amdgcn_s_setprio(3);
    // the list may hold positions of ANOTHER list (wl_slots: the survivors of the tail's first stage keep
    // the cache slot = position they had there)
    const size_t sl0 = valid ? slot : nin - 1;
    const size_t sl = a.wl_slots ? (size_t)a.wl_slots[sl0] : sl0;
    const size_t ci = a.wl_in ? (size_t)a.wl_in[sl] : a.cell0 + sl;
    // cache slot: the job's position in the launch, or (slot_by_cell) its cell's position in the block of
    // phase 0 whose records are still in the cache
    const size_t cs = a.slot_by_cell ? ci - a.cell0 : sl;
    const bool cached = CACHED && a.cache != nullptr && cs < a.cache_cap;
    // No address is ever formed from a slot the cache does not have: a read-back lane without a slot is `done`
    // from the start but steps along with its wavefront, and with records laid out job by job its loads would
    // land (cs - cache_cap) x 58 KB past the workspace.  Such lanes read slot 0's records (and store nothing).
    const size_t cs_ld = cached ? cs : 0;

    s_cell_Tt[threadIdx.x] = a.Tt[ci]; // own slot: no barrier needed
    Forcing f{a.sw, a.tc, a.pn, nullptr, a.ldf, 0.0,
              a.par[P_ELV * a.ldp + ci] * 0.0004596581, fabs(a.par[P_LAT * a.ldp + ci]) * 0.0110592101,
              (unsigned)(ci * sizeof(double))};
    const DayTable days = day_table(a.days);
    const DayRec d0 = day_rec(days, 0);

    int stage, count;
    CellConst c;
    DayOut o;
    double pet_sum = 0.0, p_sum = 0.0;
    bool finished_before = false; // a job of this list that an earlier launch of the phase completed
    // "dry year": none of the 365 records has snowfall.  The snow age nd is then never reset and, once it
    // has passed 64 days, never influences anything (see the cycle detector).  Established by a pass that
    // forms the records (any_snf), carried by the parked job (bit 24 of meta), read by the read-back kernel.
    bool dry = false, any_snf = false;
    if (!READBACK && a.fresh) {
        stage = 1;
        count = 1;
        cell_setup(c, a.par, a.ldp, ci, a.par[P_CELLOUT * a.ldp + ci], a.vt);
    } else {
        int m = a.meta[ci];
        if (m < 0) {
            finished_before = true;
            m = (2 << 16) | 1;
        }
        stage = (m >> 16) & 0xFF;
        count = m & 0xFFFF;
        dry = ((m >> 24) & 1) != 0;
        cell_setup(c, a.par, a.ldp, ci, stage == 1 ? a.par[P_CELLOUT * a.ldp + ci] : a.ai[ci], a.vt);
    }
    // The job's start-of-pass state (the state after day 1 of the current pass) is only needed
    // when a pass ends: it waits in LDS, like the checkpoint of the cycle detector below, so that
    // the day loop has the registers to itself (CellConst alone is 45 doubles).
    __shared__ double s_e[5][kBlock];
    __shared__ double s_chk[5][kBlock];
    __shared__ int s_chk_count[kBlock];
    s_chk_count[threadIdx.x] = -1;
    // day 1 of the year is evaluated at the end of every pass (the closing step, :1674 == day 1 of
    // the next pass): from the cache when there is one, from the forcing otherwise
    auto day_one = [&](State& st_, DayOut& o_, double* p_day) {
        DayRecord r;
        if (READBACK || (CACHED && cached && p_day == nullptr)) {
            r = rec_decode<RICH>(c, rec_load<RICH>(a.cache, a.cache_cap, 0, cs_ld), d0.cd);
        } else if constexpr (!READBACK) {
            double sw0, tc0, pn0;
            f.raw(0, sw0, tc0, pn0);
            f.split<true>(d0, tc0, pn0, 0.0, r.rain, r.snf);
            asm volatile("" : "+v"(r.rain), "+v"(r.snf)); // finished doubles, like the terms below
            any_snf = any_snf || (r.snf != 0.0);          // (true for NaN as well)
            day_terms(c, day_const(d0), day_prefetch(c, tc0), sw0, tc0, r.T);
            terms_fence(r.T);
            r.V = visc_terms(c, r.T);
            if (CACHED && cached) rec_store<RICH>(a.cache, a.cache_cap, 0, cs, r);
        }
        if (p_day) *p_day = r.rain + r.snf;
        day_update(c, r.T, r.V, r.rain, r.snf, st_, o_);
    };
    State cur;
    double e_wn;
    if (!READBACK && a.fresh) {
        cur = State{c.RES, 0.0, 0.0, 0.0, 0.0}; // wn_vec(n,RES), others 0, read at [n-1] (:1633-1648)
        double p0;
        day_one(cur, o, &p0);
        // sum(initial_AI$pet, na.rm = T) / sum(Pinit, na.rm = T), R/splash.point.R:150
        pet_sum = (o.pet == o.pet) ? o.pet : 0.0;
        p_sum = (p0 == p0) ? p0 : 0.0;
    } else {
        cur = State{a.state[ci], a.state[a.lds + ci], a.state[2 * a.lds + ci],
                    a.state[3 * a.lds + ci], a.state[4 * a.lds + ci]};
        if (!READBACK && CACHED && cached) { // fill day 1's record (the closing steps of this launch read it)
            State tmp = cur;
            double p0;
            day_one(tmp, o, &p0);
        }
    }
    auto park_e = [&](const State& st_) {
        s_e[0][threadIdx.x] = st_.wn;
        s_e[1][threadIdx.x] = st_.snow;
        s_e[2][threadIdx.x] = st_.qin;
        s_e[3][threadIdx.x] = st_.td;
        s_e[4][threadIdx.x] = st_.nd;
    };
    park_e(cur);
    e_wn = cur.wn;
    // Exact-cycle detection (Brent): the day-1 state of a pass determines everything after it, so
    // if it ever repeats bit for bit (period lam), the passes up to the 1000-pass cap
    // (src/SPLASH.cpp:1697) only go round that cycle -- the stopping rule cannot fire inside it, or
    // it would have fired on the first traversal -- and the state the reference ends with is the
    // one (1000 - count) mod lam passes ahead.  Most cells that never meet the 1 mm criterion sit in
    // an exact 2- or 3-cycle after a few dozen passes.  The checkpoint lives in LDS (one compare
    // per pass); it restarts with every phase, moving to the current state whenever the number of
    // passes since the phase began is a power of two.
    bool done = !valid || finished_before || (READBACK && !cached);
    for (int it = 0; it < a.budget; it++) {
        if (READBACK && it == 0 && __all(done)) break; // a wavefront with nothing to do here (every job beyond the cache)
        const bool acc = !READBACK && a.fresh && it == 0; // first pass of spin-up #1 (uniform)
        // (Finished lanes keep stepping along with the wavefront.  Masking them out of the pass -- so that
        // only the branches of the jobs still running are executed -- was measured in round 3: the
        // tail's longest job takes those branches itself, and the extra execution mask held across the
        // day loop cost 76 more scalar instructions per day step: 1 174 -> 1 235 ms.)
        // cur = state after day 1 of this pass (== the parked start-of-pass state)
        if (READBACK || (CACHED && cached && it > 0)) {
            // read-back pass: next day's record is requested a day ahead
            RawRecord<RICH> w = rec_load<RICH>(a.cache, a.cache_cap, 1, cs_ld);
            for (int i = 1; i < a.n; i++) {
                const DayRecord r = rec_decode<RICH>(c, w, RICH ? 0.0 : days[i].cd);
                if (i + 1 < a.n) w = rec_load<RICH>(a.cache, a.cache_cap, i + 1, cs_ld); // tomorrow's, untouched until tomorrow
                day_update(c, r.T, r.V, r.rain, r.snf, cur, o);
            }
        } else if constexpr (!READBACK) {
            // order of the day's vector-memory traffic as in k_run_all: gathers at the top; once
            // the terms are finished, tomorrow's forcing (and the record's stores); everything is
            // waited for at the end of the day, half a day old
            double swv, tcv, pnv;
            f.raw(1, swv, tcv, pnv);
            asm volatile("" : "+v"(swv), "+v"(tcv), "+v"(pnv) : : "memory");
            for (int i = 1; i < a.n; i++) {
                const DayPre pre = day_prefetch(c, tcv);
                const DayRec d = day_rec(days, i);
                DayRecord r;
                f.split<true>(d, tcv, pnv, 0.0, r.rain, r.snf);
                // rain and snowfall are finished doubles before the update uses them, as when they
                // are read back from the cache: the same bits from either kernel variant and from
                // any split of the passes over launches
                asm volatile("" : "+v"(r.rain), "+v"(r.snf));
                any_snf = any_snf || (r.snf != 0.0);
                day_terms(c, day_const(d), pre, swv, tcv, r.T);
                terms_fence(r.T);
                r.V = visc_terms(c, r.T);
                asm volatile("" : : : "memory");
                const int in_ = i + 1 < a.n ? i + 1 : i; // the last day reads its own row again
                double swn, tcn, pnn;
                f.raw(in_, swn, tcn, pnn);
                if (CACHED && cached) rec_store<RICH>(a.cache, a.cache_cap, i, cs, r);
                day_update(c, r.T, r.V, r.rain, r.snf, cur, o);
                if (acc) { // pet_vec is only filled in the first pass (:1667), Pinit = rain + snow
                    const double pday = r.rain + r.snf;
                    pet_sum += (o.pet == o.pet) ? o.pet : 0.0; // na.rm = T
                    p_sum += (pday == pday) ? pday : 0.0;
                }
                asm volatile("" : "+v"(swn), "+v"(tcn), "+v"(pnn) : : "memory");
                swv = swn;
                tcv = tcn;
                pnv = pnn;
            }
        }
        State nxt = cur; // the closing day-1 step (:1674/:1729) == day 1 of the next pass
        day_one(nxt, o, nullptr);
        if (!READBACK && it == 0) dry = !any_snf; // this launch's first pass has formed all 365 records
        double diff_sm = nxt.wn - e_wn; // end_sm - start_sm, :1678-1688
        if (diff_sm < 0) diff_sm = e_wn - nxt.wn;
#ifdef SPL_DIFF_STATS
        if (!done && stage == 2 && count <= 8 && g_diff_dbg) g_diff_dbg[(size_t)(count - 1) * a.ntotal + ci] = (float)diff_sm;
#endif
        if (!done) {
            if (acc) a.ai[ci] = pet_sum / p_sum; // R/splash.point.R:150
            // :1697.  Spin-up #1 only feeds the aridity index, and that uses the PET of its first
            // pass alone (pet_vec is written in pass 0 only, :1667 vs :1718-1723; the states it
            // returns are overwritten by spin-up #2, R/splash.point.R:148-152): unless its pass
            // count is asked for, it stops after that pass.
            const bool cont = (diff_sm > 1.0) && (count < 1000) && (stage == 2 || a.passes);
            if (cont) {
                count++;
                const int cc = s_chk_count[threadIdx.x];
                const bool same4 = cc >= 0 && same_bits(nxt.wn, s_chk[0][threadIdx.x]) &&
                                   same_bits(nxt.snow, s_chk[1][threadIdx.x]) &&
                                   same_bits(nxt.qin, s_chk[2][threadIdx.x]) &&
                                   same_bits(nxt.td, s_chk[3][threadIdx.x]);
                const double nd_chk = s_chk[4][threadIdx.x];
                const bool same_nd = same_bits(nxt.nd, nd_chk);
                // The snow age nd (days since the last snowfall, src/SPLASH.cpp:1225-1229) is part of the state
                // but, in a dry year, not of the dynamics: without snowfall it is never reset, grows by one per
                // day step, and is read in one place only -- the snow albedo 0.5563 + 0.4437 exp(-0.895189 nd),
                // src/SOLAR.cpp:208 -- where from nd = 64 on the second term (< 6e-26) is below half an ulp of the
                // first: the sum IS 0.5563 whatever nd.  So if wn, snow, qin and td repeat bit for bit and nd was
                // >= 64 at the checkpoint, the same days follow for good; nd after k skipped passes is nd + 365 k,
                // exactly (integers far below 2^53).  Round 4: at low latitudes 0.1 % of the cells sat in such
                // 2-cycles and ran ~1000 passes each because nd never repeats (profiles/r04_spinup.md).
                const bool same = same4 && (same_nd || (dry && nd_chk >= 64.0 && !a.no_nd_skip));
                if (same) {
                    const int left = (1000 - count) % (count - cc);
                    if (!same_nd) nxt.nd += (double)a.n * (double)((1000 - count) - left); // one per skipped day step
                    count = 1000 - left;
                    s_chk_count[threadIdx.x] = -1; // the remaining < lam passes run plainly
                } else if (((it + 1) & it) == 0) { // it + 1 = passes in this phase: 1, 2, 4, 8, ...
                    s_chk[0][threadIdx.x] = nxt.wn;
                    s_chk[1][threadIdx.x] = nxt.snow;
                    s_chk[2][threadIdx.x] = nxt.qin;
                    s_chk[3][threadIdx.x] = nxt.td;
                    s_chk[4][threadIdx.x] = nxt.nd;
                    s_chk_count[threadIdx.x] = count;
                }
                cur = nxt;
                park_e(cur);
                e_wn = cur.wn;
            } else if (stage == 1) {
                if (a.passes) a.passes[ci] = count;
                const double AI = a.ai[ci];
                a.par[P_CELLOUT * a.ldp + ci] = AI; // soil_info[12] <- AI (1-based): ncellout
                cell_setup(c, a.par, a.ldp, ci, AI, a.vt); // (the day terms do not depend on cellout)
                stage = 2;
                count = 1;
                s_chk_count[threadIdx.x] = -1;
                cur = State{c.RES, 0.0, 0.0, 0.0, 0.0};
                day_one(cur, o, nullptr);
                park_e(cur);
                e_wn = cur.wn;
            } else {
                if (writer) {
                    a.state[ci] = cur.wn; // initial$sm[365] etc., R/splash.point.R:164-172
                    a.state[a.lds + ci] = cur.snow;
                    a.state[2 * a.lds + ci] = cur.qin;
                    a.state[3 * a.lds + ci] = cur.td;
                    a.state[4 * a.lds + ci] = cur.nd;
                    if (a.passes) a.passes[a.ntotal + ci] = count;
                }
#ifdef SPL_TAIL_STATS
#ifdef SPL_DIFF_STATS
                if (READBACK && g_diff_dbg) { // when (ms since the wavefront started) and where the job ended
                    g_diff_dbg[(size_t)6 * a.ntotal + ci] = (float)((double)(wall_clock64() - t_wave_start) * 1e-5);
                    g_diff_dbg[(size_t)3 * a.ntotal + ci] = (float)((double)(t_wave_start & 0xFFFFFFFull) * 1e-5); // start, ms mod 2 684
                    unsigned hwid;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                    unsigned xcc;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                    g_diff_dbg[(size_t)7 * a.ntotal + ci] = (float)((hwid & 0xFFFF) | ((xcc & 0xF) << 16));
                    g_diff_dbg[(size_t)4 * a.ntotal + slot] = (float)((double)(wall_clock64() - t_wave_start) * 1e-5); // by list position
                    g_diff_dbg[(size_t)5 * a.ntotal + slot] = (float)((hwid & 0xFFFF) | ((xcc & 0xF) << 16));
                }
#endif
                a.meta[ci] = -(it + 2) - (READBACK ? 100000 : 0); // executed iterations of this launch
#else
                a.meta[ci] = -1; // completed (a later launch over the same list leaves it alone)
#endif
                done = true;
            }
        } else {
            cur = nxt; // finished lanes just keep stepping (their results are already stored)
        }
        if (__all(done)) break;
    }
    if (!done && valid && writer && !finished_before && !(READBACK && !cached)) { // park the job (its start-of-pass
                                                                        // state), queue the cell for the next phase
        a.state[ci] = s_e[0][threadIdx.x];
        a.state[a.lds + ci] = s_e[1][threadIdx.x];
        a.state[2 * a.lds + ci] = s_e[2][threadIdx.x];
        a.state[3 * a.lds + ci] = s_e[3][threadIdx.x];
        a.state[4 * a.lds + ci] = s_e[4][threadIdx.x];
        a.meta[ci] = (stage << 16) | count | (dry ? (1 << 24) : 0);
        if (!a.no_queue) {
            if (a.pending) a.pending[ci] = 1;
            const uint32_t k = atomicAdd(a.n_out, 1u);
            a.wl_out[k] = a.queue_slots ? (uint32_t)sl : (uint32_t)ci; // (the tail's first stage queues positions: see wl_slots)
        }
    }
}

// Module mirror of SPLASH::spin_up (src/SPLASH.cpp:1594-1749): ONE spin-up of ONE cell with an
// explicit snowfall series, written literally like the reference (a 365-day pass, then the
// closing day-1 step that tests convergence), returning the series of the last pass.
// series [7][n]: sm, snow, qin, tdrain, ro, snwage, pet (pet from the first pass only, :1667).
__global__ __launch_bounds__(64) void k_spin_up_mod(SpinArgs a, double* const* series, size_t sld,
                                                   const double* snowfall)
{
    lds_tables_init();
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const size_t ci = 0;
    Forcing f{a.sw, a.tc, a.pn, snowfall, a.ldf, 0.0, 0.0, 0.0, (unsigned)(ci * sizeof(double))};
    CellConst c;
    cell_setup(c, a.par, a.ldp, ci, a.par[P_CELLOUT * a.ldp + ci]);
    State w{c.RES, 0.0, 0.0, 0.0, 0.0}; // wn_vec(n,RES), others 0: read at [n-1] on day 1
    int count = 1;
    for (;;) {
        State cur = w;
        DayOut o;
        double start_sm = 0.0;
        for (int i = 0; i < a.n; i++) {
            double swv, tcv, pnv, rain, snf;
            f.raw(i, swv, tcv, pnv);
            f.split(a.days[i], i, tcv, pnv, rain, snf);
            day_step(c, day_const(a.days[i]), day_prefetch(c, tcv), swv, tcv, rain, snf, cur, o);
            if (i == 0) start_sm = cur.wn;
            const size_t k = (size_t)i * sld;
            series[0][k] = cur.wn;
            series[1][k] = cur.snow;
            series[2][k] = cur.qin;
            series[3][k] = cur.td;
            series[4][k] = o.ro;
            series[5][k] = cur.nd;
            if (count == 1) series[6][k] = o.pet;
        }
        State nxt = cur; // :1674 / :1729
        double sw0, tc0, pn0, rain0, snf0;
        f.raw(0, sw0, tc0, pn0);
        f.split(a.days[0], 0, tc0, pn0, rain0, snf0);
        day_step(c, day_const(a.days[0]), day_prefetch(c, tc0), sw0, tc0, rain0, snf0, nxt, o);
        double diff_sm = nxt.wn - start_sm; // :1678-1688
        if (diff_sm < 0) diff_sm = start_sm - nxt.wn;
        if (!((diff_sm > 1.0) && (count < 1000))) break; // :1697
        w = cur;
        count++;
    }
    if (a.passes) a.passes[0] = count;
}

// ================================================================ straggler pool ===============
// One workgroup reserves the block's slots in the pool; then one workgroup per job copies its columns.
__global__ __launch_bounds__(64) void k_pool_reserve(PoolGatherArgs a)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t n = *a.n_cells;
    // all or nothing: a block whose stragglers do not all fit keeps them (the caller finishes them in
    // place).  The counter only ever moves by a reservation that fits (compare-and-swap), so the slots
    // handed out are contiguous below it whatever the number of lanes reserving at once (an add that is
    // taken back on overflow is not: with three lanes a reservation made between another's add and its
    // subtraction could end up above the counter).
    uint32_t base = *(volatile uint32_t*)a.p_n, fit = 0;
    for (;;) {
        if ((size_t)base + n > a.cap) break;
        const uint32_t seen = atomicCAS(a.p_n, base, base + n);
        if (seen == base) {
            fit = n;
            break;
        }
        base = seen;
    }
    a.res[0] = base;
    a.res[1] = fit;
    a.res[2] = n;
}

__global__ __launch_bounds__(kBlock) void k_pool_gather(PoolGatherArgs a)
{
    const uint32_t fit = a.res[1], base = a.res[0];
    for (uint32_t j = blockIdx.x; j < fit; j += gridDim.x) {
        const size_t c = a.cells[j], slot = (size_t)base + j;
        for (int t = threadIdx.x; t < a.nt; t += kBlock) {
            const size_t k = (size_t)t * a.ldf + c, q = (size_t)t * a.cap + slot;
            a.p_sw[q] = a.sw[k];
            a.p_tc[q] = a.tc[k];
            a.p_pn[q] = a.pn[k];
        }
        if (threadIdx.x < P_NPAR) a.p_par[(size_t)threadIdx.x * a.cap + slot] = a.par[(size_t)threadIdx.x * a.ldp + c];
        if (threadIdx.x >= 32 && threadIdx.x < 37)
            a.p_state[(size_t)(threadIdx.x - 32) * a.cap + slot] = a.state[(size_t)(threadIdx.x - 32) * a.lds + c];
        if (threadIdx.x == 64) {
            a.p_Tt[slot] = a.Tt[c];
            a.p_meta[slot] = a.meta[c];
            a.p_ai[slot] = a.ai[c];
            a.p_origin[slot] = a.origin0 + c;
            a.p_wl[slot] = (uint32_t)slot;
            if (a.p_passes) {
                a.p_passes[slot] = a.passes ? a.passes[c] : 0;
                a.p_passes[a.cap + slot] = a.passes ? a.passes[a.ntotal + c] : 0;
            }
            if (a.p_basin) a.p_basin[slot] = a.basin ? a.basin[c] : 0;
        }
    }
}

// ================================================================ K5 run_all ===================
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Sum of a 32-bit integer over the 64 lanes with DPP adds (no LDS traffic, 6 VALU + 1 readlane;
// the total is returned to every lane): two quad permutes and two row rotations leave every lane
// of a 16-lane row with its row's total, row_bcast:15 / :31 carry the row totals to lane 63.
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);  // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);  // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}

// Sum of a 64-bit integer with |v| < 2^44 over the 64 lanes: three 16-bit digits (the top one
// signed), each summed in 32 bits without overflow, recombined.
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
    const int d0 = (int)(v & 0xFFFF), d1 = (int)((v >> 16) & 0xFFFF), d2 = (int)(v >> 32);
    return (long long)wave_sum_i32(d0) + ((long long)wave_sum_i32(d1) << 16) +
           ((long long)wave_sum_i32(d2) << 32);
}

// K7: add one cell's monthly figures to the (basin, month) record of exact fixed-point limbs
// (include/splash_hip.h).  v = a/2^16 + r with a = rint(v*2^16) and r = v - a/2^16 EXACT (scaling
// by a power of two and the subtraction of two nearby doubles are exact), then b = rint(r*2^48):
// the only rounding is below 2^-49 mm per addend.  Integer addition is associative, so the totals
// do not depend on which lane, wave, block, launch or GPU adds first.  Wavefronts whose cells
// share one basin (the usual case: basins are spatially coherent) reduce in registers and issue
// one atomic per limb; mixed wavefronts add lane by lane.
constexpr int kDiagRec = (SPLASH_NDIAG_K + 1) * 2;
constexpr int kDiagLdsMonths = 16; // months of a launch that the per-workgroup accumulators cover

__device__ __forceinline__ void diag_add(long long* __restrict__ acc, unsigned long long* s_acc,
                                         int wg_basin, const int* __restrict__ basin, int n_basin,
                                         int nmonth_cal, int month_id, int month_rel, size_t cell,
                                         bool valid, const double (*w)[kBlock], bool month_end)
{
    // w: the lane's six monthly figures in LDS (w[q][threadIdx.x]).  One quantity at a time, in a
    // rolled loop: this runs twelve times a year, and unrolled its 14 limbs would claim 28 of the
    // day loop's 256 registers at the point where it is inlined.
    // Where a whole workgroup lies in one basin (wg_basin >= 0: the usual case, decided once per
    // launch) the wave sums go to per-workgroup accumulators in LDS (s_acc[month][limb]) that are
    // flushed to the global record once, at the end of the launch: 32 768 wavefronts adding to the
    // same fourteen addresses every month end would otherwise queue up in the L2 atomic units.
    const int b = basin ? basin[cell] : 0;
    const bool in = valid && (unsigned)b < (unsigned)n_basin;
    const int b0 = __builtin_amdgcn_readfirstlane(in ? b : -1);
    // uniform: every participating lane has the first lane's basin (lanes that do not take part
    // carry zeros and may be summed along)
    const bool uniform = __all(!in || b == b0) && b0 >= 0;
    const bool to_lds = wg_basin >= 0 && month_rel < kDiagLdsMonths;
    unsigned long long* rec =
        (unsigned long long*)acc + ((size_t)(uniform ? b0 : (in ? b : 0)) * nmonth_cal + month_id) * kDiagRec;
    const bool lane0 = (threadIdx.x & 63) == 0;
    long long skipped = 0;
#pragma unroll 1
    for (int q = 0; q <= SPLASH_NDIAG_K; q++) {
        long long l0, l1;
        if (q < SPLASH_NDIAG_K) {
            const double v = w[q][threadIdx.x];
            const bool ok = in && (fabs(v) < 1048576.0); // false for NaN/inf too
            const double a = rint(v * 65536.0);
            const double r = v - a * (1.0 / 65536.0);
            l0 = ok ? (long long)a : 0;
            l1 = ok ? (long long)rint(r * 281474976710656.0) : 0;
            skipped += (in && !ok) ? 1 : 0;
        } else {
            l0 = (in && month_end) ? 1 : 0;
            l1 = skipped;
        }
        if (to_lds || uniform) {
            l0 = wave_sum_i64(l0);
            l1 = wave_sum_i64(l1);
            if (lane0) {
                unsigned long long* dst = to_lds ? s_acc + month_rel * kDiagRec : rec;
                if (l0 != 0) atomicAdd(dst + 2 * q, (unsigned long long)l0);
                if (l1 != 0) atomicAdd(dst + 2 * q + 1, (unsigned long long)l1);
            }
        } else if (in) {
            if (l0 != 0) atomicAdd(rec + 2 * q, (unsigned long long)l0);
            if (l1 != 0) atomicAdd(rec + 2 * q + 1, (unsigned long long)l1);
        }
    }
}

// limbs -> doubles, one thread per (basin, month) record
__global__ __launch_bounds__(kBlock) void k_diag_finalize(const long long* __restrict__ acc,
                                                         int n_rec, double* __restrict__ diag,
                                                         double* __restrict__ counts)
{
    constexpr int REC = (SPLASH_NDIAG_K + 1) * 2;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_rec) return;
    const long long* r = acc + (size_t)i * REC;
    for (int q = 0; q < SPLASH_NDIAG_K; q++) {
        const double hi = (double)r[2 * q];               // rounds beyond 2^53
        const double lo = (double)(r[2 * q] - (long long)hi); // exact remainder
        diag[(size_t)i * SPLASH_NDIAG_K + q] =
            hi * (1.0 / 65536.0) + (lo * (1.0 / 65536.0) + (double)r[2 * q + 1] * (1.0 / 281474976710656.0));
    }
    if (counts) {
        counts[2 * (size_t)i] = (double)r[2 * SPLASH_NDIAG_K];
        counts[2 * (size_t)i + 1] = (double)r[2 * SPLASH_NDIAG_K + 1];
    }
}

// SPLASH::run_all, src/SPLASH.cpp:1833-1916 (+ sm_lim and monthly aggregation of
// R/splash.point.R:197-214, + monthly grid totals for the water-balance diagnostics).
// outputs in order: wn, ro, pet, aet, snow, cond, bflow, netr, sm_lim
constexpr unsigned kOutMaskFive = 0x1Fu;

// SUBSET: the launch works on a list of cells (a.cells) and/or leaves out masked ones (a.skip) -- the
// two ways splash_grid_run hides the spin-up's tail behind the transient run.  A variant of its own
// (general output set), so that the plain launches carry no test for either and profiles list the
// two kinds of launch under different names.
template <bool MONTHLY, bool DIAG, unsigned OUTMASK, bool SUBSET = false>
__global__ __launch_bounds__(kBlock, SPLASH_RUNALL_WPE) void k_run_all(RunArgs a)
{
    lds_tables_init();
#ifdef SPL_SITES
    if (threadIdx.x == 0) s_site_on = 1;
    __syncthreads();
#endif
    size_t cell = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const bool in_range = cell < a.ncell;
    const size_t sl = in_range ? cell : a.ncell - 1;        // position in this launch
    const size_t ci = (SUBSET && a.cells) ? (size_t)a.cells[sl] : sl; // the cell it works on
    const size_t co = sl;                                             // the output column it writes
    // valid: the lane's results count (state, totals, monthly stores); a cell that is still
    // spinning up is computed like any other (from whatever state it has) but counts for nothing
    const bool valid = in_range && !(SUBSET && a.skip && a.skip[ci]);
    s_cell_Tt[threadIdx.x] = a.Tt ? a.Tt[ci] : 0.0; // own slot: no barrier needed
    Forcing f{a.sw, a.tc, a.pn, a.snowfall, a.ldf, 0.0,
              a.par[P_ELV * a.ldp + ci] * 0.0004596581, fabs(a.par[P_LAT * a.ldp + ci]) * 0.0110592101,
              (unsigned)(ci * sizeof(double))};
    const unsigned ooff = (unsigned)(co * sizeof(double));
    const DayTable days = day_table(a.days);
    CellConst c;
    cell_setup(c, a.par, a.ldp, ci, a.par[P_CELLOUT * a.ldp + ci], a.vt);
    const double wmax_r = a.par[P_WMAXR * a.ldp + ci];
    const double lim_den = wmax_r - c.RES;
    State st{a.state[ci], a.state[a.lds + ci], a.state[2 * a.lds + ci], a.state[3 * a.lds + ci],
             a.state[4 * a.lds + ci]};
    if (SUBSET && in_range && !valid)           // a cell still spinning up on another stream: its state is being
        st = State{c.RES, 0.0, 0.0, 0.0, 0.0}; // written there; the lane runs on a fixed one, its results count for
                                               // nothing (lanes past the last cell keep mirroring that cell)

    // which outputs are requested: known at compile time for the two common sets (OUTMASK != 0),
    // otherwise tested at run time (nine uniform booleans that the compiler keeps as 64-bit lane
    // masks, most of them spilled: ~4 issue slots per output and day)
    unsigned omask = OUTMASK;
    if (OUTMASK == 0) {
#pragma unroll
        for (int q = 0; q < 9; q++) omask |= (a.out[q] != nullptr ? 1u : 0u) << q;
    }

    double acc[9]; // monthly accumulators in output order
    unsigned nan_days = 0u; // NA days of wn / snow / sm_lim in the current month (MONTHLY): see the month's end
    // monthly flux totals of the water balance (precip, aet, ro, bflow) accumulate in LDS, one
    // slot per lane: the day loop is register bound (256 VGPRs, spilling), and four running sums
    // kept in registers cost more in spills than four LDS adds per day do
    __shared__ double s_dg[DIAG ? SPLASH_NDIAG_K : 1][DIAG ? kBlock : 1];
    __shared__ unsigned long long s_acc[DIAG ? kDiagLdsMonths * kDiagRec : 1];
    __shared__ int s_wgb[2];
    int wg_basin = -1; // the one basin of this workgroup's cells, or -1 if they are mixed
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = 0.0;
    if constexpr (DIAG) {
#pragma unroll
        for (int k = 0; k < 4; k++) s_dg[k][threadIdx.x] = 0.0;
        for (int k = threadIdx.x; k < kDiagLdsMonths * kDiagRec; k += kBlock) s_acc[k] = 0ull;
        if (threadIdx.x == 0) {
            s_wgb[0] = 0x7fffffff;
            s_wgb[1] = -0x7fffffff;
        }
        __syncthreads();
        const int b = a.basin ? a.basin[ci] : 0;
        if (valid) {
            atomicMin(&s_wgb[0], b);
            atomicMax(&s_wgb[1], b);
        }
        __syncthreads();
        if (s_wgb[0] == s_wgb[1] && (unsigned)s_wgb[0] < (unsigned)a.n_basin) wg_basin = s_wgb[0];
    }

    double swv, tcv, pnv, sfv;
    f.raw(0, swv, tcv, pnv);
    sfv = f.raw_snowfall(0);
    asm volatile("" : "+v"(swv), "+v"(tcv), "+v"(pnv), "+v"(sfv) : : "memory"); // landed before the loop
    for (int t = 0; t < a.nt; t++) {
        // Vector-memory traffic of a day, in issue order (loads and stores share ONE counter on
        // gfx950 and may complete out of order with respect to each other, so a wait for a load
        // while stores are outstanding is a wait for everything):
        //   top      gathers of today's viscosity (Infinity Cache / L2, ~1 us)
        //   mid-day  wait for the gathers (nothing else is outstanding but yesterday's stores,
        //            half a day old), THEN request tomorrow's forcing (HBM)
        //   end      wait for the forcing (half a day old), THEN today's stores
        // so that every wait finds its data long since arrived and the next day starts without one.
        const DayRec d = day_rec(days, t);
        const DayPre pre = day_prefetch(c, tcv);
        double rain, snf;
        f.split<true>(d, tcv, pnv, sfv, rain, snf);
        DayOut o;
        DayTerms T;
        day_terms(c, day_const(d), pre, swv, tcv, T);
        asm volatile("" : "+v"(T.visc) : : "memory"); // the viscosity (gathers) is finished here
        const int tn = t + 1 < a.nt ? t + 1 : t;    // the last day reads its own row again
        double swn, tcn, pnn;
        f.raw(tn, swn, tcn, pnn);
        double sfn = f.raw_snowfall(tn);
        day_update(c, T, visc_terms(c, T), rain, snf, st, o);
        asm volatile("" : "+v"(swn), "+v"(tcn), "+v"(pnn), "+v"(sfn) : : "memory"); // forcing has landed

        double lim = 0.0;
        if (omask & (1u << 8)) { // R/splash.point.R:197-200 (only when sm_lim is requested)
            lim = (st.wn - c.RES) / lim_den;
            if (lim < 0) lim = 0.0;
            if (lim > 1) lim = 1.0;
        }
        const double v[9] = {st.wn, o.ro, o.pet, o.aet, st.snow, o.cond, o.bflow, o.netr, lim};

        if (!MONTHLY) {
            // lanes past the last cell carry a copy of the last cell and store the same values to
            // the same addresses, so the stores need no per-lane predicate (only the uniform test
            // for a requested output)
            const size_t row = (size_t)t * a.ldo;
#pragma unroll
            for (int q = 0; q < 9; q++)
                if (omask & (1u << q))
                    __builtin_nontemporal_store(
                        v[q], reinterpret_cast<double*>(reinterpret_cast<char*>(a.out[q] + row) + ooff));
            if (OUTMASK == 0) { // general variant only: the state series run_all also returns
                const double x[3] = {st.qin, st.td, st.nd}; // qin_prev, tdrain, snwage (:1913-1915)
#pragma unroll
                for (int q = 0; q < 3; q++)
                    if (a.extra[q])
                        __builtin_nontemporal_store(
                            x[q], reinterpret_cast<double*>(reinterpret_cast<char*>(a.extra[q] + row) + ooff));
            }
        } else {
            // fastmatch::ctapply(x, ind.mont, FUN = mean | sum, na.rm = T), R/splash.point.R:210-211: a NA day (a gap in
            // the forcing and everything after it, SURVEY trap 3) is left out of its month -- the mean is over the
            // days that have a value, the sum over them, an all-NA month is NaN (mean) or 0 (sum)
#pragma unroll
            for (int q = 0; q < 9; q++) acc[q] += (v[q] == v[q]) ? v[q] : 0.0;
            // NA days of the three means (wn, snow, sm_lim) in one register: a byte each (a month has <= 31)
            nan_days += (v[0] != v[0] ? 1u : 0u) | (v[4] != v[4] ? 0x100u : 0u) | (v[8] != v[8] ? 0x10000u : 0u);
        }
#ifndef SPL_DIAG_NODAY
        if (DIAG) {
            // ds_add_f64 without return: fire and forget, no LDS round trip on the critical path
            // (each slot has one writer, so the "atomic" is only a way to name the instruction)
            const double inc[4] = {rain + snf, o.aet, o.ro, o.bflow};
#pragma unroll
            for (int k = 0; k < 4; k++)
                (void)__hip_atomic_fetch_add(&s_dg[k][threadIdx.x], inc[k], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#endif
        const bool month_end = (d.flags & 1) != 0;
        if (month_end || t == a.nt - 1) {
            const int m = d.month_id - a.month0;
            if (MONTHLY) {
                // mean of storages (wn, snow, sm_lim), sum of fluxes (R/splash.point.R:210-211);
                // a month split over two calls is completed by adding to the stored partial
                const bool partial_head = (m == 0) && a.head_partial;
                const bool partial_tail = !month_end; // the call ends inside the month (the series may go on: tail_continues)
                size_t k = (size_t)m * a.ldo + co;
#pragma unroll
                for (int q = 0; q < 9; q++) {
                    if ((omask & (1u << q)) && (SUBSET ? valid : in_range)) { // (a skipped cell's months are written
                                                                               //  by the later call in list mode)
                        double x = acc[q];
                        if (q == 0 || q == 4 || q == 8) {
                            // mean over the days that have a value: mdays - NA days (0 of 0: NaN, as mean(numeric(0)))
                            const int na = (int)((nan_days >> (q == 0 ? 0 : (q == 4 ? 8 : 16))) & 0xFFu);
                            const double inv_n = 1.0 / (double)(d.mdays - na);
                            if (!partial_head) {
                                x = acc[q] * inv_n;
                                // a month that a later call completes: that call adds its days to this slot and knows
                                // mdays but not this part's NA days -- with any, the month's mean cannot be completed
                                // from one stored number: NaN (never the case for chunks that end with a month, as
                                // splash.grid's year chunks do; without NA days: acc/mdays as before)
                                if (partial_tail && na != 0 && a.tail_continues) x = u2d(0x7FF8000000000000ull);
                            } else {
                                // completing the month: stored = (first part's sum) / mdays
                                const double prev = a.out[q][k];
                                x = na == 0 ? acc[q] * inv_n + prev : (prev * (double)d.mdays + acc[q]) * inv_n;
                                if (partial_tail && na != 0 && a.tail_continues) x = u2d(0x7FF8000000000000ull); // (a third part follows)
                            }
                        } else if (partial_head) {
                            x += a.out[q][k];
                        }
                        a.out[q][k] = x;
                    }
                    acc[q] = 0.0;
                }
                nan_days = 0u;
            }
#ifndef SPL_DIAG_NOMONTH
            if constexpr (DIAG) {
                // flux totals of the (part of the) month inside this call; the storages wn and
                // snow and the cell-month count only at the month's real last day
                s_dg[4][threadIdx.x] = month_end ? st.wn : 0.0;
                s_dg[5][threadIdx.x] = month_end ? st.snow : 0.0;
                diag_add(a.diag_acc, s_acc, wg_basin, a.basin, a.n_basin, a.nmonth_cal, d.month_id,
                         d.month_id - a.month0, ci, valid, s_dg, month_end);
#pragma unroll
                for (int q = 0; q < 4; q++) s_dg[q][threadIdx.x] = 0.0;
            }
#endif
        }
        swv = swn;
        tcv = tcn;
        pnv = pnn;
        sfv = sfn;
    }
    if (valid) {
        a.state[ci] = st.wn;
        a.state[a.lds + ci] = st.snow;
        a.state[2 * a.lds + ci] = st.qin;
        a.state[3 * a.lds + ci] = st.td;
        a.state[4 * a.lds + ci] = st.nd;
    }
    if constexpr (DIAG) { // flush the workgroup's accumulators: one global atomic per limb
        __syncthreads();
        if (wg_basin >= 0) {
            const int nm = a.nmonth < kDiagLdsMonths ? a.nmonth : kDiagLdsMonths;
            for (int k = threadIdx.x; k < nm * kDiagRec; k += kBlock) {
                const unsigned long long v = s_acc[k];
                if (v != 0ull)
                    atomicAdd((unsigned long long*)a.diag_acc +
                                  ((size_t)wg_basin * a.nmonth_cal + a.month0 + k / kDiagRec) * kDiagRec +
                                  k % kDiagRec,
                              v);
            }
        }
    }
}

// module mirror of run_all: daily series of all 11 outputs for a handful of cells
__global__ __launch_bounds__(64) void k_run_all_mod(RunArgs a)
{
    lds_tables_init();
    size_t cell = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (cell >= a.ncell) return;
    Forcing f{a.sw, a.tc, a.pn, a.snowfall, a.ldf, 0.0, 0.0, 0.0, (unsigned)(cell * sizeof(double))};
    CellConst c;
    cell_setup(c, a.par, a.ldp, cell, a.par[P_CELLOUT * a.ldp + cell]);
    State st{a.state[cell], a.state[a.lds + cell], a.state[2 * a.lds + cell],
             a.state[3 * a.lds + cell], a.state[4 * a.lds + cell]};
    for (int t = 0; t < a.nt; t++) {
        const DayRec d = a.days[t];
        double swv, tcv, pnv, rain, snf;
        f.raw(t, swv, tcv, pnv);
        f.split(d, t, tcv, pnv, rain, snf);
        DayOut o;
        day_step(c, day_const(d), day_prefetch(c, tcv), swv, tcv, rain, snf, st, o);
        size_t k = (size_t)t * a.ldo + cell;
        const double v[8] = {st.wn, o.ro, o.pet, o.aet, st.snow, o.cond, o.bflow, o.netr};
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (a.out[q]) a.out[q][k] = v[q];
        if (a.extra[0]) a.extra[0][k] = st.qin;
        if (a.extra[1]) a.extra[1][k] = st.td;
        if (a.extra[2]) a.extra[2][k] = st.nd;
    }
}

// ================================================================ viscosity tables ============
// every float of the argument range through the routine's own code (splash_device.hpp)
__global__ __launch_bounds__(kBlock) void k_visc_tables(float* __restrict__ rtab)
{
    lds_tables_init();
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < kViscNR) {
        float c[6];
        visc_rbar_part(__uint_as_float(kViscR0 + i), c);
        float* o = rtab + (size_t)i * 8;
#pragma unroll
        for (int k = 0; k < 6; k++) o[k] = c[k];
        o[6] = o[7] = 0.0f;
    }
}

// ================================================================ test hooks ===================
// SOLAR+EVAP for independent rows (src/EVAP.cpp:82-264)
__global__ __launch_bounds__(kBlock) void k_evap_batch(size_t m, const double* lat,
                                                      const double* elv, const double* sw,
                                                      const double* dr, const double* sd,
                                                      const double* cd, const double* sw_in,
                                                      const double* tc, const double* slop,
                                                      const double* asp, const double* snow,
                                                      const double* nd, double* out, ViscTables vt)
{
    lds_tables_init();
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    CellConst c;
    c.vt = vt;
    double sl = sin(lat[i] * kPir), cl = cos(lat[i] * kPir);
    double ss = sin(slop[i] * kPir), cs = cos(slop[i] * kPir);
    double sa = sin(asp[i] * kPir), ca = cos(asp[i] * kPir);
    c.ka = cl * ss * ca - sl * cs;
    c.kb = cl * cs + sl * ss * ca;
    c.kc = ss * sa;
    c.tau_o = (kC + kD) * (1.0 + (2.67e-5) * elv[i]);
    c.tau_a = c.tau_o * 0.1898;
    c.tau_binv = 1.0 / (c.tau_o * (1 - 0.1898));
    double ep = (kG * kMa) / (kR * kL);
    c.patm = pow((1.0 - elv[i] * kL / kTo), ep) * kPo;
    c.patm_f = (float)c.patm;
    c.visc0 = (double)viscosity_h2o(0.0f, c.patm_f);
    DayConst d{dr[i], sd[i], cd[i], 0.0, 1.0};
    Etr e;
    evap_day(c, d, day_prefetch(c, tc[i]), sw[i], sw_in[i], tc[i], snow[i], nd[i], e);
    double* o = out + i * 11;
    o[0] = e.cond;
    o[1] = e.pet; // eet == pet (src/EVAP.cpp:140)
    o[2] = e.pet;
    o[3] = e.aet;
    o[4] = e.snowmelt;
    o[5] = e.sublimation;
    o[6] = e.econ;
    o[7] = e.pw;
    o[8] = e.rn_d;
    o[9] = e.visc;
    o[10] = e.pet_max;
}

// flog / fexp / fpow / facos on the device (the host build of the same header is checked against
// libm by tests/csrc/test_math.cpp; this probe checks the device build)
__global__ __launch_bounds__(kBlock) void k_math_probe(size_t n, const double* x, const double* y,
                                                      double* out)
{
    lds_tables_init();
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    out[i] = flog(x[i]);
    out[n + i] = fexp(x[i]);
    out[2 * n + i] = fpow(x[i], y[i]);
    out[3 * n + i] = facos(x[i]);
    out[4 * n + i] = fsqrt(x[i]);
    out[5 * n + i] = fatan(x[i]);
    out[6 * n + i] = fatan2(x[i], y[i]);
}

// The day loop's own forms of moist_surf() and inf_GA() (splash_device.hpp: shared log/exp algebra,
// reciprocal-based divisions, surface moisture only inside the ponding branch) for n independent
// (cell, wn, Ksat_visc, P) rows -- the code run_one_day's restatement actually executes, as opposed
// to the literal module mirrors below.
__global__ __launch_bounds__(kBlock) void k_infiltration_probe(size_t n, const double* par,
                                                              const double* wn, const double* ksat,
                                                              const double* P, double* theta_surf,
                                                              double* infi)
{
    lds_tables_init();
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    CellConst c;
    cell_setup(c, par, n, i, par[P_CELLOUT * n + i]);
    theta_surf[i] = moist_surf(c, wn[i]);
    infi[i] = inf_GA(c, wn[i], ksat[i], P[i]);
}

// general-argument forms of the two exported helper methods (src/SPLASH.cpp:1921-1961, 1963-2023): the literal
// statements live in splash_device.hpp (moist_surf_literal, inf_GA_literal), shared with the host backend
__global__ void k_moist_surf(double depth, double z, double bub_p, double wn, double SAT,
                             double RES, double lambda, double* out)
{
    lds_tables_init();
    *out = moist_surf_literal(depth, z, bub_p, wn, SAT, RES, lambda);
}

__global__ void k_inf_GA(double bub_press, double theta_i, double Ksat, double theta_s,
                         double lambda, double P, double tdur, double slop, double* out)
{
    lds_tables_init();
    *out = inf_GA_literal(bub_press, theta_i, Ksat, theta_s, lambda, P, tdur, slop);
}

// single-precision transport of forcing rasters (FLT4S in the reference): widen on the device
__global__ __launch_bounds__(kBlock) void k_widen_f32(const float* __restrict__ src, size_t n,
                                                     double* __restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
        dst[i] = (double)__builtin_nontemporal_load(&src[i]);
}

// single-precision transport of result series (opt-in: splash_grid_out.out_f32): round to nearest even
// on the device, two values per lane (8-byte stores); what the reference's FLT4S bricks hold
__global__ __launch_bounds__(kBlock) void k_narrow_f32(const double* __restrict__ src, size_t n,
                                                      float* __restrict__ dst)
{
    const size_t n2 = n / 2;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (size_t)gridDim.x * kBlock) {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const d2_t v = __builtin_nontemporal_load((const d2_t*)src + i);
        float2 f;
        f.x = (float)v.x;
        f.y = (float)v.y;
        ((float2*)dst)[i] = f;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = (float)src[n - 1];
}

// ================================================================ launchers ===================
static inline unsigned nblk(size_t n, int b) { return (unsigned)((n + b - 1) / b); }

hipError_t launch_soil_hydro(size_t ncell, const double* lat, const double* elev,
                             const double* slop, const double* asp, const double* resolution,
                             const double* soil, const double* au, double* par, hipStream_t s)
{
    if (ncell == 0) return hipSuccess;
    hipLaunchKernelGGL(k_soil_hydro, dim3(nblk(ncell, kBlock)), dim3(kBlock), 0, s, ncell, lat,
                       elev, slop, asp, resolution, soil, au, par);
    return hipGetLastError();
}

hipError_t launch_snow_threshold(size_t ncell, int nt, const double* tc, size_t ldf,
                                 const double* par, size_t ldp, double* Tt, hipStream_t s)
{
    if (ncell == 0) return hipSuccess;
    hipLaunchKernelGGL(k_snow_threshold, dim3(nblk(ncell, kBlock)), dim3(kBlock), 0, s, ncell, nt,
                       tc, ldf, par, ldp, Tt);
    return hipGetLastError();
}

hipError_t launch_spin_phase(const SpinPhaseArgs& a, hipStream_t s)
{
    if (a.ncell == 0) return hipSuccess;
    if (a.spread && !a.readback) return hipErrorInvalidValue;
    // spread: 1024 wavefronts (256 workgroups) take ceil(n/1024) jobs each; more only for lists beyond 65 536 jobs
    const unsigned g_rb = a.spread ? std::max(nblk(a.ncell, kBlock), 256u) : nblk(a.ncell, kBlock);
    if (a.readback && a.rich) {
        // Spread launches ask for more than half of a CU's LDS (160 KB), so that no two of their workgroups can be
        // resident on one CU: beside other work (the transient run of the finished cells, whose workgroups come
        // and go every few ms) the dispatcher otherwise puts a few of them on CUs that already hold one, two of
        // the tail's wavefronts then share a SIMD, and each runs its passes 1.6x slower -- the whole tail waits
        // for them (0.68 -> 0.85-0.9 s, run by run; tools/overlap_where.py).  The pad is never touched; a
        // workgroup of k_run_all (26 KB) still fits beside one of these.  (The grid is sized for the longest
        // possible list; the workgroups past the 256 that hold jobs return at once.  Lists beyond 65 536 jobs --
        // more than 256 busy workgroups -- run one per CU instead of two, and the tails of two calls in flight on
        // one device take their CUs in turn instead of sharing them at 1.6x the time each.)
        unsigned pad = 0;
        if (a.spread) {
            // (the attribute belongs to the function ON A DEVICE: asked for once per device of the process)
            static std::atomic<int> granted[64]; // 0: not asked yet, > 0: bytes, -1: refused or switched off
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
            int g = dev < 0 ? -1 : granted[dev].load(std::memory_order_acquire);
            if (g == 0) {
                const int want = a.lds_pad;                        // static 25.9 KB + 56 KB > 80 KB (A/B: 0 switches it off)
                g = want > 0 && hipFuncSetAttribute((const void*)k_spin_phase<true, true, true>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess
                        ? want : -1;
                (void)hipGetLastError();
                granted[dev].store(g, std::memory_order_release);
            }
            pad = g > 0 ? (unsigned)g : 0u;
        }
        hipLaunchKernelGGL((k_spin_phase<true, true, true>), dim3(g_rb), dim3(kBlock), pad, s, a);
    } else if (a.readback)
        hipLaunchKernelGGL((k_spin_phase<true, true, false>), dim3(g_rb), dim3(kBlock), 0, s, a);
    else if (a.cache && a.rich)
        hipLaunchKernelGGL((k_spin_phase<true, false, true>), dim3(nblk(a.ncell, kBlock)), dim3(kBlock), 0, s, a);
    else if (a.cache)
        hipLaunchKernelGGL(k_spin_phase<true>, dim3(nblk(a.ncell, kBlock)), dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL(k_spin_phase<false>, dim3(nblk(a.ncell, kBlock)), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_pool_gather(const PoolGatherArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(k_pool_reserve, dim3(1), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_pool_gather, dim3(2048), dim3(kBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_spin_up_mod(const SpinArgs& a, double* const* series_dev, size_t sld,
                              const double* snowfall, hipStream_t s)
{
    if (a.ncell == 0) return hipSuccess;
    hipLaunchKernelGGL(k_spin_up_mod, dim3(nblk(a.ncell, 64)), dim3(64), 0, s, a, series_dev, sld,
                       snowfall);
    return hipGetLastError();
}

#ifdef SPL_SITES
static void site_dump()
{
    static unsigned long long h[2048][2];
    if (hipDeviceSynchronize() != hipSuccess) return;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(splash::g_site_hits), sizeof h) != hipSuccess) return;
    std::fprintf(stderr, "SITE ---- cumulative\n");
    for (int i = 0; i < 2048; i++)
        if (h[i][0])
            std::fprintf(stderr, "SITE splash_device.hpp:%d lanes %llu waves %llu\n", i, h[i][0], h[i][1]);
}
#endif

hipError_t launch_run_all(const RunArgs& a, hipStream_t s)
{
#ifdef SPL_SITES
    struct Dump { ~Dump() { site_dump(); } } dump_after_launch;
#endif
    if (a.ncell == 0 || a.nt == 0) return hipSuccess;
    dim3 g(nblk(a.ncell, kBlock)), b(kBlock);
    const bool diag = a.diag_acc != nullptr;
    unsigned mask = 0;
    for (int q = 0; q < 9; q++) mask |= (a.out[q] != nullptr ? 1u : 0u) << q;
    if (a.extra[0] || a.extra[1] || a.extra[2]) mask = 0x200u; // only the general variant stores them
    if (a.cells || a.skip) {
        if (a.monthly) {
            if (diag)
                hipLaunchKernelGGL((k_run_all<true, true, 0, true>), g, b, 0, s, a);
            else
                hipLaunchKernelGGL((k_run_all<true, false, 0, true>), g, b, 0, s, a);
        } else {
            if (diag)
                hipLaunchKernelGGL((k_run_all<false, true, 0, true>), g, b, 0, s, a);
            else
                hipLaunchKernelGGL((k_run_all<false, false, 0, true>), g, b, 0, s, a);
        }
    } else if (a.monthly) {
        if (diag)
            hipLaunchKernelGGL((k_run_all<true, true, 0>), g, b, 0, s, a);
        else
            hipLaunchKernelGGL((k_run_all<true, false, 0>), g, b, 0, s, a);
    } else if (mask == kOutMaskFive) { // wn, ro, pet, aet, snow: the BASELINE output set
        if (diag)
            hipLaunchKernelGGL((k_run_all<false, true, kOutMaskFive>), g, b, 0, s, a);
        else
            hipLaunchKernelGGL((k_run_all<false, false, kOutMaskFive>), g, b, 0, s, a);
    } else if (mask == 0x1FFu) { // all nine (splash.grid)
        if (diag)
            hipLaunchKernelGGL((k_run_all<false, true, 0x1FFu>), g, b, 0, s, a);
        else
            hipLaunchKernelGGL((k_run_all<false, false, 0x1FFu>), g, b, 0, s, a);
    } else {
        if (diag)
            hipLaunchKernelGGL((k_run_all<false, true, 0>), g, b, 0, s, a);
        else
            hipLaunchKernelGGL((k_run_all<false, false, 0>), g, b, 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_run_all_mod(const RunArgs& a, hipStream_t s)
{
    if (a.ncell == 0 || a.nt == 0) return hipSuccess;
    hipLaunchKernelGGL(k_run_all_mod, dim3(nblk(a.ncell, 64)), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_diag_finalize(const long long* acc, int n_rec, double* diag, double* counts,
                                hipStream_t s)
{
    if (n_rec <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_diag_finalize, dim3(nblk((size_t)n_rec, kBlock)), dim3(kBlock), 0, s, acc,
                       n_rec, diag, counts);
    return hipGetLastError();
}

size_t spin_cache_bytes_per_slot() { return (size_t)365 * kNRecCompact * sizeof(double); }
size_t spin_cache_rich_cap(size_t cap) { return (cap * kNRecCompact / kRichStride) & ~(size_t)63; }
size_t visc_table_floats_r() { return (size_t)kViscNR * 8; }
hipError_t launch_visc_tables(float* rtab, hipStream_t s)
{
    hipLaunchKernelGGL(k_visc_tables, dim3(nblk(kViscNR, kBlock)), dim3(kBlock), 0, s, rtab);
    return hipGetLastError();
}

hipError_t launch_widen_f32(const float* src, size_t n, double* dst, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const size_t nb = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_widen_f32, dim3((unsigned)(nb < 65536 ? nb : 65536)), dim3(kBlock), 0, s, src,
                       n, dst);
    return hipGetLastError();
}

hipError_t launch_narrow_f32(const double* src, size_t n, float* dst, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const size_t nb = (n / 2 + kBlock - 1) / kBlock + 1;
    hipLaunchKernelGGL(k_narrow_f32, dim3((unsigned)(nb < 65536 ? nb : 65536)), dim3(kBlock), 0, s, src,
                       n, dst);
    return hipGetLastError();
}

hipError_t launch_evap_batch(size_t m, const double* lat, const double* elv, const double* sw,
                             const double* dr, const double* sd, const double* cd,
                             const double* sw_in, const double* tc, const double* slop,
                             const double* asp, const double* snow, const double* nd, double* out,
                             ViscTables vt, hipStream_t s)
{
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_evap_batch, dim3(nblk(m, kBlock)), dim3(kBlock), 0, s, m, lat, elv, sw, dr,
                       sd, cd, sw_in, tc, slop, asp, snow, nd, out, vt);
    return hipGetLastError();
}

hipError_t launch_math_probe(size_t n, const double* x, const double* y, double* out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_math_probe, dim3(nblk(n, kBlock)), dim3(kBlock), 0, s, n, x, y, out);
    return hipGetLastError();
}

hipError_t launch_infiltration_probe(size_t n, const double* par, const double* wn,
                                     const double* ksat, const double* P, double* theta_surf,
                                     double* infi, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_infiltration_probe, dim3(nblk(n, kBlock)), dim3(kBlock), 0, s, n, par, wn,
                       ksat, P, theta_surf, infi);
    return hipGetLastError();
}

hipError_t launch_moist_surf(double depth, double z, double bub_p, double wn, double SAT,
                             double RES, double lambda, double* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_moist_surf, dim3(1), dim3(1), 0, s, depth, z, bub_p, wn, SAT, RES, lambda,
                       out);
    return hipGetLastError();
}

hipError_t launch_inf_GA(double bub_press, double theta_i, double Ksat, double theta_s,
                         double lambda, double P, double tdur, double slop, double* out,
                         hipStream_t s)
{
    hipLaunchKernelGGL(k_inf_GA, dim3(1), dim3(1), 0, s, bub_press, theta_i, Ksat, theta_s, lambda,
                       P, tdur, slop, out);
    return hipGetLastError();
}

} // namespace splash
