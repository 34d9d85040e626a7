// splash_grid_host.hip -- splash_grid_run / _submit / _wait: the host-buffer entry point of the C ABI
// (include/splash_hip.h), i.e. what clFun and the receive loop of the reference do for a row block
// (R/splash.grid.R:277-308, 359-400): NA-cell masking, blocking over HBM capacity, a pipeline of lanes
// (host thread + stream + buffers) whose transfers and kernels overlap, one straggler pool and one
// spin-up tail per call, delivery into the caller's arrays.  No CPU compute path: every block goes
// through the HIP kernels (splash_dev_* stages of splash_capi.hip).
#include "splash_host.hpp"

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <stdexcept>
#include <thread>

using namespace splash;
using namespace splash::host;

// (Environment knobs: sampled ONCE per call, on the caller's thread, when the call enters the library -- tests flip
// them inside one process.  The snapshot travels with the call: GridPlan::kn, installed by a KnobScope on every
// thread that works for it (the job thread of splash_grid_submit, per-device workers, lanes, delivery helpers), so
// that the public splash_dev_* stages a lane calls find the call's knobs instead of reading the environment again
// -- round 6, ADVICE r05: round 5's snapshot was re-read by those stages on the lanes' threads.  No thread of this
// file calls getenv, and a job in flight keeps the knobs it was submitted with.)

// Mapping the caller's (typically freshly allocated) result arrays before the device->host copies
// reach them: columns [c0, c0+ncols) of every row of the [nrows][N] arrays of `esz`-byte elements, on a
// few host threads.
//   populate (default)  madvise(MADV_POPULATE_WRITE): the kernel maps the pages writable and leaves
//                       their contents alone -- nothing a concurrent delivery wrote can be lost;
//   touch               one store of 0 per 4 KiB page (kernels without MADV_POPULATE_WRITE fall back to
//                       it).  It MODIFIES the range, so the range must be the calling block's own
//                       columns only: the first version touched the whole worker range from the first
//                       lane to arrive and could zero values the other lane's helper thread had just
//                       delivered (tests/test_gpu_host_path.py::test_prefault_touches_only_its_own_columns);
//   off                 nothing.
// SPLASH_HIP_PREFAULT selects the mode, SPLASH_HIP_PREFAULT_MIN_BYTES the block size (bytes of one
// block's share of the arrays) from which it pays: 64 MiB.

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static void prefault_outputs(const Knobs& kn, void* const* bufs, const int* esz, int nbuf, size_t N, size_t c0,
                             size_t ncols, size_t nrows, size_t block_bytes)
{
    const int mode = kn.prefault_mode;
    if (mode == PREFAULT_OFF || ncols == 0 || nrows == 0) return;
    size_t bytes = 0;
    for (int k = 0; k < nbuf; k++) bytes += bufs[k] ? ncols * nrows * (size_t)esz[k] : 0;
    if (block_bytes < kn.prefault_min_bytes) return; // small: not worth threads (the whole block's bytes decide)
    if (const size_t ms = kn.prefault_delay_ms) // test aid: let the other lanes deliver first
        std::this_thread::sleep_for(std::chrono::milliseconds(ms));
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    unsigned nthr = std::thread::hardware_concurrency();
    nthr = nthr == 0 ? 4 : (nthr > 8 ? 8 : nthr);
    if (bytes < ((size_t)8 << 20)) nthr = 1;
    auto work = [=](unsigned w) {
        for (int k = 0; k < nbuf; k++) {
            if (!bufs[k]) continue;
            const size_t e = (size_t)esz[k];
            for (size_t r = w; r < nrows; r += nthr) {
                char* seg = (char*)bufs[k] + (r * N + c0) * e;
                const size_t len = ncols * e;
                bool done = false;
                if (mode == PREFAULT_POPULATE) {
                    // whole pages inside... and around: a page shared with a neighbouring block is only
                    // mapped, never written, so rounding outwards is harmless
                    const uintptr_t a0 = (uintptr_t)seg & ~(uintptr_t)(page - 1);
                    const uintptr_t a1 = ((uintptr_t)seg + len + page - 1) & ~(uintptr_t)(page - 1);
                    done = madvise((void*)a0, a1 - a0, MADV_POPULATE_WRITE) == 0;
                }
                if (!done) { // touch: this block's own elements only
                    for (size_t i = 0; i < len; i += page) *(volatile char*)(seg + i) = 0;
                    *(volatile char*)(seg + len - 1) = 0;
                }
            }
        }
    };
    if (nthr == 1) {
        work(0);
        return;
    }
    Threads pool;
    for (unsigned w = 0; w < nthr; w++) pool.add(work, w);
    pool.join();
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
    // where series k goes: the caller's array and its element size (8: out[k]; 4: out_f32[k], rounded to
    // single precision on the device -- what the reference's FLT4S bricks hold, R/splash.grid.R:186-216);
    // dst[k] == NULL: not wanted
    void* dst[SPLASH_NOUT];
    int esz[SPLASH_NOUT];
    int n_out32;       // series delivered in single precision
    size_t max_cells;
    bool want_diag;
    double mem_fraction; // share of the device's free memory one worker may plan with
    bool resume;         // (ABI 5) in->state0: no spin-up, run_all starts from the caller's state
    Knobs kn;            // the environment knobs of the call: sampled once, on the caller's thread, when the call
                         // entered the library; every thread that works for the call installs them (KnobScope)
};

// maximal runs of consecutive grid cells in idx[0, nb) (idx ascending): fn(j0, len) -- the dense fast paths copy
// columns [idx[j0], idx[j0] + len) of the caller's arrays with one strided copy per run.  One run when the block
// is one contiguous range (a one-device call on a grid without NA cells); with several devices a worker's list
// is a round-robin deal of chunks, and a block holds a few of them.
template <class F>
void for_runs(const size_t* idx, size_t nb, F fn)
{
    if (nb == 0) return;
    if (idx[nb - 1] - idx[0] == nb - 1) { // ascending and gap-free
        fn((size_t)0, nb);
        return;
    }
    size_t j0 = 0;
    for (size_t j = 1; j <= nb; j++)
        if (j == nb || idx[j] != idx[j - 1] + 1) {
            fn(j0, j - j0);
            j0 = j;
        }
}

inline void put_value(const GridPlan& P, int k, size_t at, double v)
{
    if (P.esz[k] == 8) ((double*)P.dst[k])[at] = v; else ((float*)P.dst[k])[at] = (float)v;
}

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
    Threads pool;
    for (unsigned w = 0; w < nthr; w++)
        pool.add([=]() {
            for (size_t i = w; i < n; i += nthr) fn(i);
        });
    pool.join();
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
    // for the straggler pool's tail: a few wavefronts on the critical path of the call, dispatched ahead of the lanes' work
    hipError_t create_high_priority()
    {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return create();
        return hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
    }
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
    DevBuf d_force, d_f32, d_out, d_o32, d_par, d_state, d_stat, d_tt, d_pass, d_swork, d_basin, d_res, d_late;
    HostBuf<double> stage, hstat, hsmall;
    HostBuf<char> stage_out;
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
    const size_t* widx; // the worker's valid cells (grid cell of its j-th cell), ascending
    size_t nwv, nb_max; // how many; cells per block
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

    Worker(const GridPlan& P_, int dev, const size_t* cells, size_t ncells)
        : P(P_), device(dev), widx(cells), nwv(ncells), nb_max(0)
    {
        f32 = (P.in->sw_in == nullptr);
        trace = P.kn.trace;
    }
    ~Worker() { splash_calendar_destroy(cal); }

    size_t nblocks() const { return (nwv + nb_max - 1) / nb_max; }
    // slots of a lane's buffer for late cells finished in place (late_fallback): part of the sizing
    static size_t late_slots(size_t nb) { return std::min<size_t>(nb, (size_t)1 << 14); }
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
    KnobScope knobs_of_the_call(P.kn); // this thread calls public stages (splash_dev_*): no getenv here
    Lane& L = lanes[li];
    if (hipSetDevice(device) != hipSuccess) {
        lane_fail(L, SPLASH_ENODEV, "hipSetDevice", hipErrorInvalidDevice);
    } else {
        try { // (a throw on a lane's thread would be terminate(): it becomes the lane's error instead)
            const size_t nblk = nblocks();
            if (P.kn.test_throw == 2) throw std::runtime_error("injected on a lane's thread");
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
        } catch (...) {
            lane_fail(L, exception_code(nullptr, 0, "a lane's host side"), "lane", hipSuccess);
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
    const size_t b0 = k * nb_max, nb = std::min(nb_max, nwv - b0); // cells [b0, b0+nb) of the worker's list
    const size_t* idx = widx + b0;  // idx[j] = grid cell of the block's j-th cell
    // dense: the block is a few runs of consecutive grid cells (one, when one device works on the grid)
    std::vector<std::pair<size_t, size_t>> runs;
    if (P.dense) for_runs(idx, nb, [&](size_t j0, size_t len) { runs.emplace_back(j0, len); });
    double* F = L.d_force.as<double>();
    double* ddst[3] = {F, F + (size_t)nt * nb, F + (size_t)2 * nt * nb};
    double *dsw = ddst[0], *dtc = ddst[1], *dpn = ddst[2];
    // ---- host -> device.  (Pageable sources: the copies return when the source may be reused, so
    // the staging vectors can be refilled at once; on the device they are ordered by the stream.)
    for (int v = 0; v < 3; v++) {
        if (!f32) {
            const double* hsrc = v == 0 ? in->sw_in : (v == 1 ? in->tc : in->pn);
            if (P.dense) {
                for (const auto& r : runs)
                    LANE_TRY(hipMemcpy2DAsync(ddst[v] + r.first, nb * 8, hsrc + idx[r.first], N * 8, r.second * 8,
                                              (size_t)nt, hipMemcpyHostToDevice, s));
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
                for (const auto& r : runs)
                    LANE_TRY(hipMemcpy2DAsync(L.d_f32.as<float>() + r.first, nb * 4, hsrc + idx[r.first], N * 4,
                                              r.second * 4, (size_t)nt, hipMemcpyHostToDevice, s));
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
    // (hsmall holds 7 rows: the three small uploads of a block -- tt, state0, ai0 -- have DISJOINT regions, so that
    //  none is refilled while a copy out of it may still be in flight, whatever the runtime does with a pageable
    //  source)
    if (in->tt) { // the threshold of the WHOLE series, from the caller (time chunks: include/splash_hip.h)
        for (size_t j = 0; j < nb; j++) L.hsmall[j] = in->tt[idx[j]];
        LANE_TRY(hipMemcpyAsync(Tt, L.hsmall.data(), nb * 8, hipMemcpyHostToDevice, s));
    } else {
        LANE_RC(splash_dev_snow_threshold(nb, nt, dtc, nb, par, nb, Tt, s));
    }
    int* dpass = out->aux ? L.d_pass.as<int>() : nullptr;
    const unsigned char* d_mask = nullptr;
    const uint32_t *d_cells = nullptr, *d_ncells = nullptr;
    if (P.resume) {
        // continuing a run: the caller's state and the aridity index of the run's spin-up stand where the
        // spin-up's results would (cellout := AI, R/splash.point.R:150); nothing is left for the pool
        double* const h_state = L.hsmall.data() + nb_max, * const h_ai = L.hsmall.data() + 6 * nb_max;
        for (int kk = 0; kk < 5; kk++)
            for (size_t j = 0; j < nb; j++) h_state[(size_t)kk * nb + j] = in->state0[(size_t)kk * N + idx[j]];
        LANE_TRY(hipMemcpyAsync(state, h_state, 5 * nb * 8, hipMemcpyHostToDevice, s));
        for (size_t j = 0; j < nb; j++) h_ai[j] = in->ai0[idx[j]];
        LANE_TRY(hipMemcpyAsync(par + (size_t)P_CELLOUT * nb, h_ai, nb * 8, hipMemcpyHostToDevice, s));
        LANE_TRY(hipMemsetAsync(L.d_res.p, 0, 64, s));
        if (dpass) LANE_TRY(hipMemsetAsync(dpass, 0, 2 * nb * sizeof(int), s));
        LANE_TRY(hipEventRecord(L.ev_gather.e, s));
    } else {
    LANE_RC(splash_dev_spin_up_begin(cal, nb, dsw, dtc, dpn, nb, Tt, par, nb, state, nb, dpass, L.d_swork.p, s));
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
        g.origin0 = (uint64_t)b0; // position in the worker's list
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
    }
    note(blocks_gathered, false); // the worker's thread starts the tail once every block is here
    // the previous block's results must have left d_out before run_all writes it again
    if (L.d2h.joinable()) L.d2h.join();
    if (L.d2h_rc) return lane_fail(L, L.d2h_rc, "device->host copy of the previous block", hipSuccess);
    splash_dev_out dout{};
    int q = 0;
    for (int kk = 0; kk < SPLASH_NOUT; kk++)
        dout.out[kk] = P.dst[kk] ? L.d_out.as<double>() + (size_t)(q++) * nt_out * nb : nullptr;
    dout.ldo = nb;
    dout.monthly = P.monthly;
    if (P.want_diag) {
        dout.diag_acc = d_acc.as<int64_t>();
        dout.basin = with_basin ? L.d_basin.as<int32_t>() : nullptr;
        dout.n_basin = P.n_basin;
    }
    dout.skip = d_mask;
    LANE_RC(splash_dev_run_all(cal, 0, nt, nb, dsw, dtc, dpn, nb, Tt, par, nb, state, nb, &dout, s));
    // single-precision delivery: the block's series rounded on the device, same layout in floats
    if (P.n_out32)
        LANE_TRY(launch_narrow_f32(L.d_out.as<double>(), (size_t)P.n_out * nt_out * nb, L.d_o32.as<float>(), s));
    // The kernels of this block are queued; while they run, map the caller's (typically freshly
    // allocated) output arrays: the first touch of tens of GB is otherwise paid page by page inside
    // the device-to-host copies.
    // Only THIS block's columns (the rows' segments [c0, c0 + nb)): the other lane may be delivering a
    // neighbouring block's results at this moment, and a page-touching write outside the block's own
    // columns would race with them (see prefault_outputs).
    if (P.dense) {
        size_t blk_bytes = 0; // the block's share of the result arrays decides whether mapping pays, not a run's
        for (int kk = 0; kk < SPLASH_NOUT; kk++) blk_bytes += P.dst[kk] ? nb * (size_t)nt_out * (size_t)P.esz[kk] : 0;
        if (P.kn.test_prefault_whole_range) // fault injection for the regression test ONLY: the range of the
            prefault_outputs(P.kn, P.dst, P.esz, SPLASH_NOUT, N, widx[0], widx[nwv - 1] - widx[0] + 1, (size_t)nt_out, // first version
                             (size_t)-1);
        else
            for (const auto& r : runs)
                prefault_outputs(P.kn, P.dst, P.esz, SPLASH_NOUT, N, idx[r.first], r.second, (size_t)nt_out, blk_bytes);
    }
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
            KnobScope knobs_of_the_call(P.kn);
            (void)hipSetDevice(device);
            try {
                L.d2h_rc = deliver(L, b0, nb, dout);
            } catch (...) {
                L.d2h_rc = exception_code(nullptr, 0, "delivery of a block's results");
            }
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
    const size_t N = P.N;
    const int nt_out = P.nt_out;
    const size_t* idx = widx + b0;
    const double t0 = now_s();
    for (int kk = 0; kk < SPLASH_NOUT; kk++) {
        if (!P.dst[kk]) continue;
        const size_t e8 = (size_t)P.esz[kk];
        // a single-precision series sits in d_o32 where its double-precision original sits in d_out
        const char* dsrc = e8 == 8 ? (const char*)dout.out[kk]
                                   : (const char*)L.d_o32.p + (size_t)(dout.out[kk] - L.d_out.as<double>()) * 4;
        char* const dsth = (char*)P.dst[kk];
        hipError_t e = hipSuccess;
        if (P.dense) {
            for_runs(idx, nb, [&](size_t j0, size_t len) {
                if (e == hipSuccess)
                    e = hipMemcpy2D(dsth + idx[j0] * e8, N * e8, dsrc + j0 * e8, nb * e8, len * e8, (size_t)nt_out,
                                    hipMemcpyDeviceToHost);
            });
        } else {
            char* stg = L.stage_out.data();
            e = hipMemcpy(stg, dsrc, (size_t)nt_out * nb * e8, hipMemcpyDeviceToHost);
            if (e == hipSuccess) {
                if (e8 == 8)
                    parallel_rows((size_t)nt_out, nb * 8, [=](size_t t) {
                        const double* src = (const double*)stg + t * nb;
                        double* dst = (double*)dsth + t * N;
                        for (size_t j = 0; j < nb; j++) dst[idx[j]] = src[j];
                    });
                else
                    parallel_rows((size_t)nt_out, nb * 4, [=](size_t t) {
                        const float* src = (const float*)stg + t * nb;
                        float* dst = (float*)dsth + t * N;
                        for (size_t j = 0; j < nb; j++) dst[idx[j]] = src[j];
                    });
            }
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
    const size_t* idx = widx + b0;
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
    // the buffer the sizing reserved for this (late_slots(); allocated on first use, kept by the lane)
    const size_t late_cap = std::min<size_t>(n_late, late_slots(nb_max));
    LANE_TRY(L.d_late.reserve((size_t)nt_out * std::max(n_out, 1) * late_slots(nb_max) * 8));
    std::vector<double> late_vals((size_t)nt_out * late_cap); // (not the worker's `patch`: that one belongs to the pool)
    for (size_t k0 = 0; k0 < n_late; k0 += late_cap) {
        const size_t nl = std::min(late_cap, (size_t)n_late - k0);
        splash_dev_out dl = dout;
        int ql = 0;
        for (int k = 0; k < SPLASH_NOUT; k++)
            dl.out[k] = P.dst[k] ? L.d_late.as<double>() + (size_t)(ql++) * nt_out * late_cap : nullptr;
        dl.ldo = late_cap;
        dl.skip = nullptr;
        dl.cells = d_cells + k0;
        LANE_RC(splash_dev_run_all(cal, 0, nt, nl, dsw, dtc, dpn, nb, Tt, par, nb, state, nb, &dl, s));
        LANE_TRY(hipStreamSynchronize(s));
        for (int k = 0; k < SPLASH_NOUT; k++) {
            if (!P.dst[k]) continue;
            LANE_TRY(hipMemcpy2D(late_vals.data(), nl * 8, dl.out[k], late_cap * 8, nl * 8, (size_t)nt_out,
                                 hipMemcpyDeviceToHost));
            for (int t = 0; t < nt_out; t++)
                for (size_t j = 0; j < nl; j++)
                    put_value(P, k, (size_t)t * N + idx[late[k0 + j]], late_vals[(size_t)t * nl + j]);
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
    HIP_TRY(st.create_high_priority(), err, errlen);
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
            dout.out[k] = P.dst[k] ? pool.out.as<double>() + (size_t)(q++) * nt_out * cap : nullptr;
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
            if (P.dst[k])
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
    const size_t* idx = widx;
    const uint64_t* org = origin.data();
    int q = 0;
    for (int k = 0; k < SPLASH_NOUT; k++) {
        if (!P.dst[k]) continue;
        const double* src = patch.data() + (size_t)(q++) * nt_out * np;
        if (P.esz[k] == 8) {
            double* dst = (double*)P.dst[k];
            parallel_rows((size_t)nt_out, np * 16, [=](size_t t) {
                for (size_t j = 0; j < np; j++) dst[t * N + idx[org[j]]] = src[t * np + j];
            });
        } else { // rounded here as the device rounds the other cells: to nearest even
            float* dst = (float*)P.dst[k];
            parallel_rows((size_t)nt_out, np * 16, [=](size_t t) {
                for (size_t j = 0; j < np; j++) dst[t * N + idx[org[j]]] = (float)src[t * np + j];
            });
        }
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

// The valid cells cells[0, ncells) (grid cell ids, ascending) on `device`, in HBM-sized blocks dealt to the lanes.
// `share` = workers using this device (they split its free memory).  acc: this worker's host copy of the exact
// diagnostic limbs (summed by the caller).
int grid_worker(const GridPlan& P, int device, int share, const size_t* cells, size_t ncells, int64_t* acc,
                char* err, size_t errlen)
{
    const splash_grid_in* in = P.in;
    const int nt = P.nt, nt_out = P.nt_out, n_out = P.n_out;
    HIP_TRY(hipSetDevice(device), err, errlen);
    if (ncells == 0) return SPLASH_OK;
    Worker W(P, device, cells, ncells);
    if (int rc = splash_calendar_create(nt, in->doy, in->year, in->month, &W.cal, err, errlen)) return rc;
    const bool f32 = W.f32;
    const double t_w0 = now_s();
    const size_t want = ncells;
    // Lanes: two keep the link and the GPU busy when the results are the bulk of the traffic (daily outputs in
    // double precision: device->host is the bottleneck and a third lane only adds blocks); a call whose results are
    // light -- monthly outputs, or daily ones in single precision -- is bound by its kernels, and a third lane
    // fills the gaps between them (*measured*, whole grid monthly 2.56 -> 2.47 s, 2048x4096 x 5 daily f32 results
    // 1.86 -> 1.73 s; daily f64 results 2.52 -> 2.62 s, hence the rule).  SPLASH_HIP_LANES overrides.
    size_t out_bytes_cell = 0;
    for (int k = 0; k < SPLASH_NOUT; k++) out_bytes_cell += P.dst[k] ? (size_t)nt_out * (size_t)P.esz[k] : 0;
    const size_t in_bytes_cell = (size_t)nt * 3 * (f32 ? 4 : 8);
    int n_lanes = out_bytes_cell <= 2 * in_bytes_cell ? 3 : 2;
    const Knobs& kn = P.kn;
    if (kn.lanes > 0) n_lanes = kn.lanes;
    const size_t acc_bytes = splash_dev_diag_acc_size(P.n_basin, P.nmonth);
    {
        // Sizing and allocation happen under the device's allocation lock: a second job on the same
        // device (splash_grid_submit) plans with what this one has left.
        std::lock_guard<std::recursive_mutex> alloc_lock(device_alloc_mutex(device));
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b), err, errlen);
        if (kn.fake_device) { // test aid: a device of that many bytes, of
            const size_t fake = kn.fake_device_bytes, held = pool_held_bytes(device); // which the pool's buffers are taken
            free_b = std::min(free_b, fake > held ? fake - held : 0);
        }
        free_b += pool_idle_bytes(device); // buffers kept from earlier calls are ours to reuse or release
        // the straggler pool: room for 1/128 of the worker's cells (0.08 % of the synthetic grids' cells
        // are stragglers); a block whose stragglers do not fit finishes them in place (late_fallback)
        size_t pcap = std::min<size_t>(std::max<size_t>(want / 128, 16384), (size_t)1 << 18);
        pcap = std::min(pcap, (want + 255) & ~(size_t)255);
        pcap = (pcap + 255) & ~(size_t)255;
        if (kn.pool_cap) pcap = kn.pool_cap; // tests: force the overflow path
        const bool spin = !P.resume; // a continued run has no spin-up: no workspace, a token pool
        if (!spin) pcap = 256;
        const size_t pool_per_slot = (size_t)nt * 3 * 8 + (size_t)nt_out * std::max(n_out, 1) * 8 +
                                     (SPLASH_NPAR + SPLASH_NSTATE + 1) * 8 + 8 + 2 * 4 + 4;
        const double pool_bytes = (double)pcap * (double)pool_per_slot + (double)splash_dev_spin_up_work_size(pcap);
        const size_t per_cell = (size_t)nt * 3 * 8 + (f32 ? (size_t)nt * 4 : 0) + (size_t)nt_out * std::max(n_out, 1) * 8 +
                                (P.n_out32 ? (size_t)nt_out * n_out * 4 : 0) +
                                (SPLASH_NPAR + SPLASH_NSTATE + 14 + 1) * 8 + 2 * 4 + 4 + 64;
        const double frac = P.mem_fraction > 0 ? P.mem_fraction : 0.85;
        const double budget = (double)free_b * frac / (double)std::max(share, 1) - pool_bytes - (double)acc_bytes;
        // blocks: four per lane once a grid is big enough for the pipeline to pay, of a million cells at
        // least (a block's later spin-up phases are short worklists, i.e. latency: ~30 ms per block)
        // (SPLASH_HIP_BLOCKS_PER_LANE / SPLASH_HIP_MIN_BLOCK_CELLS: tests run the many-blocks plan on small grids)
        size_t nb_max = want;
        const size_t per_lane = kn.blocks_per_lane;
        const size_t min_block = kn.min_block_cells;
        if (want > min_block) nb_max = std::max<size_t>(min_block, (want + per_lane * n_lanes - 1) / (per_lane * n_lanes));
        if (P.max_cells) nb_max = std::min(nb_max, P.max_cells);
        auto lanes_for = [&](size_t nb) { return (int)std::min<size_t>((size_t)n_lanes, (want + nb - 1) / nb); };
        auto need = [&](size_t nb) {
            return (double)lanes_for(nb) * ((double)nb * (double)per_cell +
                                            (spin ? (double)splash_dev_spin_up_work_size(nb) +
                                                        (double)Worker::late_slots(nb) * nt_out * std::max(n_out, 1) * 8
                                                  : 0.0));
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
        for (Lane& L : W.lanes) {
            HIP_TRY(L.st.create(), err, errlen);
            HIP_TRY(L.ev_gather.create(), err, errlen);
            HIP_TRY(L.d_force.alloc((size_t)nt * 3 * nb_max * 8), err, errlen);
            if (f32) HIP_TRY(L.d_f32.alloc((size_t)nt * nb_max * 4), err, errlen);
            HIP_TRY(L.d_out.alloc((size_t)nt_out * std::max(n_out, 1) * nb_max * 8), err, errlen);
            if (P.n_out32) HIP_TRY(L.d_o32.alloc((size_t)nt_out * n_out * nb_max * 4), err, errlen);
            HIP_TRY(L.d_par.alloc((size_t)SPLASH_NPAR * nb_max * 8), err, errlen);
            HIP_TRY(L.d_state.alloc((size_t)SPLASH_NSTATE * nb_max * 8), err, errlen);
            HIP_TRY(L.d_stat.alloc((size_t)14 * nb_max * 8), err, errlen);
            HIP_TRY(L.d_tt.alloc(nb_max * 8), err, errlen);
            HIP_TRY(L.d_pass.alloc(2 * nb_max * sizeof(int)), err, errlen);
            HIP_TRY(L.d_swork.alloc(spin ? splash_dev_spin_up_work_size(nb_max) : 64), err, errlen);
            HIP_TRY(L.d_res.alloc(64), err, errlen);
            if (P.want_diag && in->basin) HIP_TRY(L.d_basin.alloc(nb_max * sizeof(int32_t)), err, errlen);
            L.hstat.resize((size_t)14 * nb_max);
            if (P.out->state_end || P.out->aux || in->tt || P.resume) L.hsmall.resize(7 * nb_max);
            if (P.out->aux) L.hpass.resize(2 * nb_max);
            if (in->basin && P.want_diag) L.hbasin.resize(nb_max);
            if (!P.dense) {
                if (!f32) L.stage.resize((size_t)nt * nb_max);
                L.stage_out.resize((size_t)nt_out * nb_max * 8);
                if (f32) L.stage32.resize((size_t)nt * nb_max);
            }
        }
        {   // the pool's counters and the totals start at zero (what phase 0 does for a block's workspace).
            // On a lane's stream, and waited for: the lanes' streams do not synchronise with the null
            // stream (hipStreamNonBlocking), and a memset is not ordered against them by itself.
            hipStream_t s0 = W.lanes[0].st.s;
            size_t p_cnt;
            spin_work_layout(pcap, nullptr, nullptr, nullptr, &p_cnt);
            HIP_TRY(hipMemsetAsync((char*)pl.work.p + p_cnt, 0, 64, s0), err, errlen);
            if (P.want_diag) {
                HIP_TRY(W.d_acc.alloc(acc_bytes), err, errlen);
                HIP_TRY(hipMemsetAsync(W.d_acc.p, 0, acc_bytes, s0), err, errlen);
            }
            HIP_TRY(hipStreamSynchronize(s0), err, errlen);
        }
    }
    if (W.trace)
        fprintf(stderr, "[splash] device %d: %zu cells in %zu blocks of <= %zu on %zu lanes, pool of %zu slots; calendar + "
                        "allocations %.3f s\n", device, want, W.nblocks(), W.nb_max, W.lanes.size(), W.pool.cap, now_s() - t_w0);

    Threads threads; // (lanes take blocks off a shared counter: should one fail to start, the others finish the list)
    for (size_t l = 0; l < W.lanes.size(); l++) threads.add([&W, l] { W.run_lane((int)l); });
    const int rc_pool = W.finish_pool(err, errlen); // waits for the lanes' last gathers, then runs the tail
    threads.join();
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

// SPLASH_BACKEND_HOST: every valid cell through host_cell_run (splash_host_backend.hip: the product's own device
// functions compiled for the host), a few cells per host thread.  For one site or a handful of cells, where a CPU
// core steps through the chain of day steps faster than a lone wavefront; no device is touched.
static int grid_on_host(const GridPlan& P, char* err, size_t errlen)
{
    const splash_grid_in* in = P.in;
    const splash_grid_out* out = P.out;
    const size_t N = P.N;
    splash_calendar cal;
    if (int rc = calendar_host(P.nt, in->doy, in->year, in->month, &cal, err, errlen)) return rc;
    const size_t n_rec = (size_t)P.n_basin * P.nmonth;
    std::vector<int64_t> acc(P.want_diag ? n_rec * SPLASH_DIAG_REC : 0, 0);
    static_assert(sizeof(long long) == sizeof(int64_t), "limbs");
    std::atomic<size_t> next{0};
    std::atomic<int> rc_all{SPLASH_OK};
    auto work = [&]() {
        KnobScope knobs_of_the_call(P.kn);
        for (;;) {
            const size_t j = next.fetch_add(1);
            if (j >= P.nv || rc_all.load() != SPLASH_OK) return; // (host_cell_run reports ENOMEM itself: no throw here)
            const size_t c = P.idx[j];
            HostCellIn ci{};
            ci.nt = P.nt;
            ci.days = cal.host.data();
            ci.spin_days = cal.spin_host.empty() ? nullptr : cal.spin_host.data();
            ci.stride = N;
            if (in->sw_in) {
                ci.sw = in->sw_in + c;
                ci.tc = in->tc + c;
                ci.pn = in->pn + c;
            } else {
                ci.sw32 = in->sw_in_f32 + c;
                ci.tc32 = in->tc_f32 + c;
                ci.pn32 = in->pn_f32 + c;
            }
            ci.lat = in->lat[c];
            ci.elev = in->elev[c];
            ci.slop = in->slop[c];
            ci.asp = in->asp[c];
            ci.resolution = in->resolution[c];
            for (int k = 0; k < 6; k++) ci.soil[k] = in->soil[(size_t)k * N + c];
            ci.have_au = in->au != nullptr;
            for (int k = 0; k < 3; k++) ci.au[k] = in->au ? in->au[(size_t)k * N + c] : 0.0;
            double tt1 = 0.0, st5[5], ai1 = 0.0;
            if (in->tt) {
                tt1 = in->tt[c];
                ci.tt = &tt1;
            }
            if (in->state0) {
                for (int k = 0; k < 5; k++) st5[k] = in->state0[(size_t)k * N + c];
                ai1 = in->ai0[c];
                ci.state0 = st5;
                ci.ai0 = &ai1;
            }
            ci.monthly = P.monthly;
            ci.want_passes = out->aux != nullptr;
            HostCellOut co{};
            for (int k = 0; k < SPLASH_NOUT; k++) {
                if (out->out[k]) co.out[k] = out->out[k] + c;
                else if (out->out_f32[k]) co.out32[k] = out->out_f32[k] + c;
            }
            co.ldo = N;
            if (P.want_diag) {
                const int b = in->basin ? in->basin[c] : 0;
                if ((unsigned)b < (unsigned)P.n_basin) co.diag_rec = (long long*)acc.data() + (size_t)b * P.nmonth * SPLASH_DIAG_REC;
            }
            if (int rc = host_cell_run(ci, co)) {
                rc_all.store(rc);
                return;
            }
            if (out->state_end)
                for (int k = 0; k < 5; k++) out->state_end[(size_t)k * N + c] = co.state_end[k];
            if (out->aux) {
                out->aux[c] = co.tt;
                out->aux[N + c] = co.ai;
                out->aux[2 * N + c] = co.passes1;
                out->aux[3 * N + c] = co.passes2;
            }
        }
    };
    unsigned nthr = std::thread::hardware_concurrency();
    nthr = nthr == 0 ? 4 : (nthr > 16 ? 16 : nthr);
    nthr = (unsigned)std::min<size_t>(nthr, P.nv);
    if (nthr <= 1) {
        work();
    } else {
        Threads pool;
        for (unsigned w = 0; w < nthr; w++) pool.add(work);
        pool.join();
    }
    if (rc_all.load() != SPLASH_OK) return fail(rc_all.load(), err, errlen, "grid_run (host backend): a cell failed");
    if (P.want_diag) {
        if (out->diag_acc) std::copy(acc.begin(), acc.end(), out->diag_acc);
        if (out->diag) splash_diag_finalize_host(acc.data(), P.n_basin, P.nmonth, out->diag, nullptr);
    }
    return SPLASH_OK;
}

// the call itself, on whatever thread runs it (the caller's, or the job thread of splash_grid_submit:
// the environment knobs were sampled by the entry point, on the caller's thread)
static int grid_run_impl(const splash_grid_in* in, const splash_opts* opt, const splash_grid_out* out, char* err,
                         size_t errlen, const Knobs& kn_call) noexcept
try {
    KnobScope knobs_of_the_call(kn_call); // (the job thread of splash_grid_submit: no getenv, the submitter's knobs)
    if (!in || !out) return fail(SPLASH_EINVAL, err, errlen, "grid_run: NULL in/out");
    const double t_enter = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const bool f64 = in->sw_in && in->tc && in->pn;
    const bool f32 = !in->sw_in && !in->tc && !in->pn && in->sw_in_f32 && in->tc_f32 && in->pn_f32;
    if (!in->doy || !in->year || !in->month || !(f64 || f32) || !in->lat || !in->elev ||
        !in->slop || !in->asp || !in->resolution || !in->soil)
        return fail(SPLASH_EINVAL, err, errlen,
                    "grid_run: NULL input array (forcing: all three series as double, or all "
                    "three as float with the double pointers NULL)");
    if ((in->state0 != nullptr) != (in->ai0 != nullptr) || (in->state0 && !in->tt))
        return fail(SPLASH_EINVAL, err, errlen,
                    "grid_run: continuing a run needs state0, ai0 and tt together (include/splash_hip.h, ABI 5)");
    if (in->state0 ? in->nt < 1 : in->nt < 365)
        return fail(SPLASH_EINVAL, err, errlen,
                    "grid_run: %d days given, the spin-up needs the first 365 "
                    "(R/splash.point.R:141-144)", in->nt);
    if (in->basin && in->n_basin <= 0)
        return fail(SPLASH_EINVAL, err, errlen, "grid_run: basin ids given with n_basin <= 0");
    for (int k = 0; k < SPLASH_NOUT; k++)
        if (out->out[k] && out->out_f32[k])
            return fail(SPLASH_EINVAL, err, errlen, "grid_run: series %d asked for in double AND single precision "
                        "(out[k] and out_f32[k]): one array per series", k);
    const int backend = opt ? opt->backend : SPLASH_BACKEND_HIP;
    if (backend != SPLASH_BACKEND_HIP && backend != SPLASH_BACKEND_HOST)
        return fail(SPLASH_EINVAL, err, errlen, "grid_run: unknown backend %d", backend);
    const bool on_host = backend == SPLASH_BACKEND_HOST; // an explicit choice (ABI 6), never a fallback:
    if (!on_host)                                        // the default backend needs a HIP device
        if (int rc = need_device(err, errlen)) return rc;

    // devices: one worker per entry of the list (the same ordinal may repeat)
    std::vector<int> devs;
    if (opt && opt->n_devices > 1) {
        for (int k = 0; k < opt->n_devices; k++) devs.push_back(opt->devices ? opt->devices[k] : k);
    } else {
        devs.push_back(opt ? ((opt->n_devices == 1 && opt->devices) ? opt->devices[0] : opt->device) : 0);
    }
    const int ndev_present = on_host ? 0 : splash_hip_device_count();
    for (int d : devs)
        if (!on_host && (d < 0 || d >= ndev_present))
            return fail(SPLASH_EINVAL, err, errlen, "grid_run: device %d not present (%d visible)", d,
                        ndev_present);

    GridPlan P{};
    P.kn = kn_call;
    P.in = in;
    P.out = out;
    P.N = in->ncell;
    P.nt = in->nt;
    P.monthly = opt ? opt->monthly_out : 0;
    P.resume = in->state0 != nullptr;
    P.max_cells = opt ? opt->max_cells_per_launch : 0;
    P.mem_fraction = (opt && opt->mem_fraction > 0 && opt->mem_fraction <= 1) ? opt->mem_fraction : 0.0;
    P.n_basin = in->basin ? in->n_basin : 1;
    P.want_diag = out->diag || out->diag_acc;
    for (int k = 0; k < SPLASH_NOUT; k++) {
        P.dst[k] = out->out[k] ? (void*)out->out[k] : (void*)out->out_f32[k];
        P.esz[k] = out->out_f32[k] ? 4 : 8;
        P.n_out += P.dst[k] ? 1 : 0;
        P.n_out32 += out->out_f32[k] ? 1 : 0;
    }
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
            // a continued run: a cell that was NA in the chunk before (NaN column water) stays NA
            // (td alone may be NaN in a valid cell: the sticky drainage time of flat terrain)
            if (in->state0 && isnan_d(in->state0[c])) bad = true;
            return bad;
        };
        // a few threads over contiguous ranges, lists concatenated in order (16.7 M cells: 0.12 s on one)
        unsigned nthr = std::thread::hardware_concurrency();
        nthr = nthr == 0 ? 4 : (nthr > 8 ? 8 : nthr);
        if (N < ((size_t)1 << 20)) nthr = 1;
        std::vector<std::vector<size_t>> part(nthr);
        for (unsigned w = 0; w < nthr; w++) // (here, not on the helpers' threads: an allocation may throw)
            part[w].reserve(N * (w + 1) / nthr - N * w / nthr);
        auto scan = [&](unsigned w) {
            const size_t a0 = N * w / nthr, a1 = N * (w + 1) / nthr;
            for (size_t c = a0; c < a1; c++)
                if (!is_bad(c)) part[w].push_back(c);
        };
        if (nthr == 1) {
            scan(0);
            idx.swap(part[0]);
        } else {
            Threads pool;
            for (unsigned w = 0; w < nthr; w++) pool.add(scan, w);
            pool.join();
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
            else if (out->out_f32[k]) std::fill(out->out_f32[k], out->out_f32[k] + (size_t)P.nt_out * N, (float)NaN);
        if (out->state_end) std::fill(out->state_end, out->state_end + 5 * N, NaN);
        if (out->aux) std::fill(out->aux, out->aux + 4 * N, NaN);
    }
    const size_t n_rec = (size_t)P.n_basin * P.nmonth;
    if (out->diag) std::fill(out->diag, out->diag + n_rec * SPLASH_NDIAG, 0.0);
    if (out->diag_acc) std::fill(out->diag_acc, out->diag_acc + n_rec * SPLASH_DIAG_REC, (int64_t)0);
    if (P.nv == 0) return SPLASH_OK;
    if (P.kn.test_throw == 1) throw std::bad_alloc(); // fault injection for tests/: no exception leaves the C ABI
    if (on_host) return grid_on_host(P, err, errlen);

    const double t_scan = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    // One worker per listed device.  Round 6: the valid cells are cut into CHUNKS that are dealt round-robin
    // (worker w owns chunks w, w + nw, ...), not into one contiguous range per worker: cells are in raster order, so
    // a contiguous range is a latitude band, the day step costs ~5 % more at low latitudes and the spin-up's pass
    // counts vary with climate -- the equatorial worker would finish last (what bench.py's ranks avoid the same way:
    // rsplash_amd/dist.py::row_blocks_round_robin).  About sixteen chunks per worker, 256 cells at least, 2^17 at
    // most; a worker's list keeps the raster order, so its blocks stay a few runs of consecutive cells (for_runs).
    // Cells are independent: the split changes nothing in the results, totals included -- exact integer sums.
    const int nw = (int)devs.size();
    std::vector<std::vector<size_t>> wcells(nw > 1 ? nw : 0);
    if (nw > 1) {
        size_t chunk = (P.nv + (size_t)nw * 16 - 1) / ((size_t)nw * 16);
        chunk = std::min<size_t>(std::max<size_t>(chunk, 1), (size_t)1 << 17);
        chunk = (chunk + 255) & ~(size_t)255;
        for (int w = 0; w < nw; w++) wcells[w].reserve(P.nv / nw + chunk);
        size_t j = 0;
        for (size_t v0 = 0; v0 < P.nv; v0 += chunk, j++) {
            std::vector<size_t>& dst = wcells[j % (size_t)nw];
            dst.insert(dst.end(), P.idx + v0, P.idx + std::min(P.nv, v0 + chunk));
        }
    }
    std::vector<std::vector<int64_t>> acc(nw);
    std::vector<int> rcs(nw, 0);
    std::vector<std::string> msgs(nw);
    for (auto& m : msgs) m.reserve(512); // (so that a worker's thread need not allocate to report)
    auto run = [&](int w) {
        KnobScope knobs_of_the_call(P.kn);
        char buf[512] = "";
        try {
            int share = 0;
            for (int d : devs) share += (d == devs[w]);
            if (P.kn.test_throw == 3) throw std::runtime_error("injected on a worker's thread");
            if (P.want_diag) acc[w].assign(n_rec * SPLASH_DIAG_REC, 0);
            rcs[w] = nw > 1 ? grid_worker(P, devs[w], share, wcells[w].data(), wcells[w].size(), acc[w].data(), buf, sizeof buf)
                            : grid_worker(P, devs[w], share, P.idx, P.nv, acc[w].data(), buf, sizeof buf);
        } catch (...) { // (on a worker's own thread a throw would be terminate())
            rcs[w] = exception_code(buf, sizeof buf, "a device's worker");
        }
        msgs[w].assign(buf);
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
        const size_t elems_each = (size_t)P.nt_out * N;
        size_t total = 0;
        for (int k = 0; k < SPLASH_NOUT; k++) total += P.dst[k] ? elems_each * (size_t)P.esz[k] : 0;
        // SPLASH_HIP_REGISTER_MIN_BYTES: result bytes from which registering pays (tests set 0 to run the
        // path on small grids); SPLASH_HIP_NO_REGISTER=1 disables it
        const Knobs& kn = P.kn;
        if (!kn.no_register && P.dense && total >= kn.register_min_bytes) {
            const int dev0 = devs[0];
            // the forcing too (both directions must be page-locked for the copies to overlap), first:
            // its copies start at once
            const size_t fbytes = (size_t)P.nt * N * (f64 ? 8 : 4);
            void* fptr[3] = {f64 ? (void*)in->sw_in : (void*)in->sw_in_f32, f64 ? (void*)in->tc : (void*)in->tc_f32,
                             f64 ? (void*)in->pn : (void*)in->pn_f32};
            const GridPlan* plan = &P;
            reg.done.reserve(3 + SPLASH_NOUT); // (the helper thread never allocates)
            reg.th = std::thread([&reg, plan, elems_each, dev0, fbytes, fptr] {
                (void)hipSetDevice(dev0);
                auto lock = [&](void* p, size_t n) {
                    if (hipHostRegister(p, n, hipHostRegisterPortable) == hipSuccess)
                        reg.done.push_back(p);
                    else
                        (void)hipGetLastError();
                };
                for (int v = 0; v < 3 && !reg.stop; v++) lock(fptr[v], fbytes);
                for (int k = 0; k < SPLASH_NOUT && !reg.stop; k++)
                    if (plan->dst[k]) lock(plan->dst[k], elems_each * (size_t)plan->esz[k]);
            });
        }
    }
    if (nw == 1) {
        run(0);
    } else {
        Threads pool;
        for (int w = 0; w < nw; w++) pool.add(run, w);
        pool.join();
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
    if (P.kn.trace) {
        const double t_end = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        fprintf(stderr, "[splash] grid_run: NA scan %.3f s, workers %.3f s, total %.3f s\n", t_scan - t_enter,
                t_end - t_scan, t_end - t_enter);
    }
    return SPLASH_OK;
} catch (...) { // (this is also the body of splash_grid_submit's job thread)
    return exception_code(err, errlen, "grid_run");
}

extern "C" int splash_grid_run(const splash_grid_in* in, const splash_opts* opt,
                               const splash_grid_out* out, char* err, size_t errlen)
try {
    knobs_refresh();
    return grid_run_impl(in, opt, out, err, errlen, knobs());
} catch (...) {
    return exception_code(err, errlen, "splash_grid_run");
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
    Knobs kn; // the submitter's knobs at submit time: the job keeps them whatever the environment does meanwhile
};

extern "C" int splash_grid_submit(const splash_grid_in* in, const splash_opts* opt, const splash_grid_out* out,
                                  splash_grid_job** job, char* err, size_t errlen)
try {
    if (!in || !out || !job) return fail(SPLASH_EINVAL, err, errlen, "grid_submit: NULL in/out/job");
    if (opt && opt->backend == SPLASH_BACKEND_HOST) {
        knobs_refresh(); // (the host backend touches no device: nothing to refuse at submit time)
    } else if (int rc = enter_device(err, errlen)) {
        return rc;
    }
    std::unique_ptr<splash_grid_job> j(new splash_grid_job);
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
    j->kn = knobs();
    splash_grid_job* jp = j.get();
    // (the job thread: grid_run_impl catches what its body throws, so nothing is thrown on this thread)
    j->th = std::thread([jp] { jp->rc = grid_run_impl(&jp->in, jp->has_opt ? &jp->opt : nullptr, &jp->out, jp->msg, sizeof jp->msg, jp->kn); });
    *job = j.release();
    return SPLASH_OK;
} catch (...) {
    return exception_code(err, errlen, "splash_grid_submit");
}

extern "C" int splash_grid_wait(splash_grid_job* job, char* err, size_t errlen)
try {
    if (!job) return fail(SPLASH_EINVAL, err, errlen, "grid_wait: NULL job");
    if (job->th.joinable()) job->th.join();
    const int rc = job->rc;
    if (rc) (void)fail(rc, err, errlen, "%s", job->msg);
    delete job;
    return rc;
} catch (...) {
    return exception_code(err, errlen, "splash_grid_wait");
}

