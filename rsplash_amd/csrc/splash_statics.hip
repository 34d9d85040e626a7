// splash_statics.hip -- the kernels either side of the daily loop and the synthetic inputs:
//   unSWC epilogue (f-4), terrain slope/aspect/latitude/resolution and ncellflow (f-2), sink fill, D8 flow
//   accumulation and the multiple-flow-direction area (f-1), and the counter-based generator of the synthetic
//   statics and forcing (bench.py and the parity tests generate the same values on the CPU).
// The loop itself -- soil_hydro, snow threshold, spin-up, run_all, the module mirrors -- is splash_kernels.hip.
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "splash_kernels.hpp"
#include "splash_host.hpp"

#include "splash_device.hpp"
#include "splash_unswc.hpp"

namespace splash {

static constexpr int kBlock = 256;
static constexpr int kStripRows = 16; // rows a lane walks in the static-raster stencil kernels
static inline unsigned nblk(size_t n, int b) { return (unsigned)((n + b - 1) / b); }

// ================================================================ unSWC ==========
// R/unsSWC.R:14-88 as a pointwise epilogue over a [nt][ld] raster of column water content wn:
// water held between the surface and uns_depth (analytical integral of the Brooks-Corey profile,
// :41-43), water-table depth and mean volumetric content of that layer.  One lane per cell; the
// soil-only terms (soil_hydro with fgravel = 0, :55) are derived once per lane.
// HBM traffic: 8 B read + 24 B written per cell-step (32 B algorithmic).
__global__ __launch_bounds__(kBlock) void k_unswc(size_t ncell, int nt,
                                                 const double* __restrict__ wn, size_t ldw,
                                                 const double* __restrict__ soil,
                                                 double uns_depth, double* __restrict__ wn_z,
                                                 double* __restrict__ wtd_o,
                                                 double* __restrict__ theta_o, size_t ldo)
{
    lds_tables_init();
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= ncell) return;
    UnswcCell u;
    unswc_setup(soil[i], soil[ncell + i], soil[2 * ncell + i], soil[3 * ncell + i], soil[4 * ncell + i], soil[5 * ncell + i],
                uns_depth, u);
    // the next step's column water is requested a step ahead and waited for before this step's
    // stores are issued (loads and stores share one counter: see k_run_all)
    double w_next = nt > 0 ? __builtin_nontemporal_load(&wn[i]) : 0.0;
    asm volatile("" : "+v"(w_next) : : "memory");
    for (int t = 0; t < nt; t++) {
        const double w = w_next;
        w_next = __builtin_nontemporal_load(&wn[(size_t)(t + 1 < nt ? t + 1 : t) * ldw + i]);
        double v_wn_z, v_wtd, v_theta;
        unswc_step(u, w, v_wn_z, v_wtd, v_theta);
        asm volatile("" : "+v"(w_next) : : "memory"); // the next step's load has landed
        const size_t k = (size_t)t * ldo + i;
        if (theta_o) __builtin_nontemporal_store(v_theta, &theta_o[k]); // :80
        if (wn_z) __builtin_nontemporal_store(v_wn_z, &wn_z[k]);       // :82
        if (wtd_o) __builtin_nontemporal_store(v_wtd, &wtd_o[k]);
    }
}

// unSWC.grid (R/unsSWC.grid.R:14-141): the gridded function's own semantics (unswc_grid_step) and its four rasters.
// HBM traffic: 8 B read + 32 B written per cell-step with all four wanted.
__global__ __launch_bounds__(kBlock) void k_unswc_grid(size_t ncell, int nt, const double* __restrict__ wn, size_t ldw,
                                                      const double* __restrict__ soil, double uns_depth,
                                                      double* __restrict__ w_z, double* __restrict__ wtd_o,
                                                      double* __restrict__ se_o, double* __restrict__ theta_o, size_t ldo)
{
    lds_tables_init();
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= ncell) return;
    UnswcCell u;
    unswc_grid_setup(soil[i], soil[ncell + i], soil[2 * ncell + i], soil[3 * ncell + i], soil[4 * ncell + i],
                     soil[5 * ncell + i], uns_depth, u);
    double w_next = nt > 0 ? __builtin_nontemporal_load(&wn[i]) : 0.0;
    asm volatile("" : "+v"(w_next) : : "memory");
    for (int t = 0; t < nt; t++) {
        const double w = w_next;
        w_next = __builtin_nontemporal_load(&wn[(size_t)(t + 1 < nt ? t + 1 : t) * ldw + i]);
        double v_wz, v_wtd, v_se, v_theta;
        unswc_grid_step(u, w, v_wz, v_wtd, v_se, v_theta);
        asm volatile("" : "+v"(w_next) : : "memory"); // the next step's load has landed (one counter for loads and stores)
        const size_t k = (size_t)t * ldo + i;
        if (w_z) __builtin_nontemporal_store(v_wz, &w_z[k]);
        if (wtd_o) __builtin_nontemporal_store(v_wtd, &wtd_o[k]);
        if (se_o) __builtin_nontemporal_store(v_se, &se_o[k]);
        if (theta_o) __builtin_nontemporal_store(v_theta, &theta_o[k]);
    }
}

// ================================================================ synthetic inputs =============
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double synth_u(uint64_t seed, uint64_t stream, uint64_t cell,
                                          uint64_t day)
{
    uint64_t h = mix64(mix64(mix64(seed ^ stream) ^ cell) ^ day);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}
enum { ST_ELEV = 1, ST_SLOP, ST_ASP, ST_AU, ST_NIN, ST_SAND, ST_CLAY, ST_OM, ST_GRAVEL, ST_BD,
       ST_DEPTH, ST_DEPTH2, ST_TC = 20, ST_SW, ST_ROCC, ST_RAMT };

__global__ __launch_bounds__(kBlock) void k_synth_statics(uint64_t seed, int H, int W,
                                                         int64_t cell0, size_t ncell, int terrain,
                                                         double* lat, double* elev, double* slop,
                                                         double* asp, double* resolution,
                                                         double* soil, double* au)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= ncell) return;
    uint64_t c = (uint64_t)(cell0 + (int64_t)i);
    int64_t r = (cell0 + (int64_t)i) / W;
    lat[i] = 60.0 - 120.0 * ((double)r + 0.5) / (double)H;
    resolution[i] = 250.0;
    const double Ai = 62500.0;
    if (terrain) {
        elev[i] = 200.0 + 2800.0 * synth_u(seed, ST_ELEV, c, 0);
        double u = synth_u(seed, ST_SLOP, c, 0);
        slop[i] = 30.0 * u * u;
        asp[i] = 360.0 * synth_u(seed, ST_ASP, c, 0);
        au[i] = Ai * exp2(12.0 * synth_u(seed, ST_AU, c, 0));
        au[ncell + i] = 1.0 + floor(3.0 * synth_u(seed, ST_NIN, c, 0));
        au[2 * ncell + i] = 3.0;
    } else {
        elev[i] = 500.0;
        slop[i] = 0.0;
        asp[i] = 0.0;
        au[i] = 0.0;
        au[ncell + i] = 3.0;
        au[2 * ncell + i] = 3.0;
    }
    double sand = 15.0 + 55.0 * synth_u(seed, ST_SAND, c, 0);
    double cmax = 95.0 - sand;
    if (cmax > 60.0) cmax = 60.0;
    double clay = 8.0 + (cmax - 8.0) * synth_u(seed, ST_CLAY, c, 0);
    double depth = 0.6 + 1.4 * synth_u(seed, ST_DEPTH, c, 0);
    if (synth_u(seed, ST_DEPTH2, c, 0) < 0.1) depth = 2.0;
    soil[i] = sand;
    soil[ncell + i] = clay;
    soil[2 * ncell + i] = 0.3 + 7.7 * synth_u(seed, ST_OM, c, 0);
    soil[3 * ncell + i] = 40.0 * synth_u(seed, ST_GRAVEL, c, 0);
    soil[4 * ncell + i] = 0.9 + 0.7 * synth_u(seed, ST_BD, c, 0);
    soil[5 * ncell + i] = depth;
}

__global__ __launch_bounds__(kBlock) void k_synth_forcing(uint64_t seed,
                                                         const DayRec* __restrict__ days, int t0,
                                                         int nt, int64_t cell0, size_t ncell,
                                                         const double* __restrict__ lat,
                                                         const double* __restrict__ elev,
                                                         double* sw_in, double* tc, double* pn,
                                                         size_t ldf)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= ncell) return;
    const double TWO_PI = 6.283185307179586;
    const uint64_t c = (uint64_t)(cell0 + (int64_t)i);
    const double la = lat[i], el = elev[i];
    const double base = 25.0 * cos(la * kPir) - 5.0;
    const double amp = (5.0 + 0.25 * fabs(la));
    const double shift = 15.0 + (la < 0.0 ? 182.0 : 0.0);
    for (int t = blockIdx.y; t < nt; t += gridDim.y) {
        const uint64_t day = (uint64_t)(t0 + t);
        const double n = (double)days[t].doy;
        size_t k = (size_t)t * ldf + i;
        double v = base - amp * cos(TWO_PI * (n - shift) / 365.0) - 0.0065 * el +
                   3.0 * (synth_u(seed, ST_TC, c, day) - 0.5) * 2.0;
        tc[k] = (double)(float)v;
        double decl = 23.44 * sin(TWO_PI * (n - 80.0) / 365.0);
        double s = 180.0 * cos((la - decl) * kPir) * (0.55 + 0.45 * synth_u(seed, ST_SW, c, day)) +
                   10.0;
        if (s < 5.0) s = 5.0;
        if (s > 420.0) s = 420.0;
        sw_in[k] = (double)(float)s;
        double p = 0.0;
        if (synth_u(seed, ST_ROCC, c, day) < 0.3)
            p = -6.0 * log(1.0 - synth_u(seed, ST_RAMT, c, day));
        pn[k] = (double)(float)p;
    }
}

// ================================================================ sink fill + D8 accumulation ===
// The rest of R/upslope_area.R: sink filling (:15-16 `topmodel::sinkfill(dem, res, 1)`,
// :68 TauDEM pitremove) and the contributing area (:21 `topidx()$area`, :74 TauDEM aread8).  Both
// are third-party codes outside the reference tree; what stands here are the published algorithms
// (parity unpinned): Planchon & Darboux (2002) for the fill, plain D8 accumulation for the area.
//
// Sink fill: W = z on outlets (grid edge, next to NA), +inf elsewhere; relax
//   W(c) <- max(z(c), min over the 8 neighbours n of W(n) + eps(n)),  eps = tan(degree)*res (*sqrt 2
// on diagonals), until nothing changes.  The relaxation is monotone (every W only ever decreases,
// and rounding keeps a <= b => a + eps <= b + eps), so the fixed point -- for every cell the
// smallest over all paths to an outlet of the sum formed outlet-inwards -- does not depend on the
// order of the updates: sweeps are done in place, any number of them per launch, and the result is
// bit-identical to a serial evaluation in any sweep order.
__global__ __launch_bounds__(kBlock) void k_sinkfill_init(int H, int W, const double* __restrict__ z,
                                                         double* __restrict__ w)
{
    const int c = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (c >= W) return;
    for (int r = (int)blockIdx.y; r < H; r += (int)gridDim.y) {
        const size_t i = (size_t)r * W + c;
        const double zc = z[i];
        bool outlet = r == 0 || r == H - 1 || c == 0 || c == W - 1;
        if (!outlet) {
#pragma unroll
            for (int q = 0; q < 9; q++) {
                const double v = z[(size_t)(r + q / 3 - 1) * W + (c + q % 3 - 1)];
                outlet |= (v != v);
            }
        }
        w[i] = (zc != zc) ? zc : (outlet ? zc : __builtin_inf());
    }
}

__global__ __launch_bounds__(kBlock) void k_sinkfill_sweep(int H, int W, const double* __restrict__ z,
                                                          double* w, double eps, double eps_d,
                                                          int* __restrict__ changed)
{
    const int c = (int)(blockIdx.x * kBlock + threadIdx.x);
    bool any = false;
    if (c > 0 && c < W - 1) {
        for (int r = (int)blockIdx.y * kStripRows; r < H; r += (int)gridDim.y * kStripRows) {
            const int r1 = r + kStripRows < H ? r + kStripRows : H;
            for (int rr = r < 1 ? 1 : r; rr < r1 && rr < H - 1; rr++) {
                const size_t i = (size_t)rr * W + c;
                const double zc = z[i];
                // relaxed agent-scope accesses: other workgroups update w while this one reads it
                const double wc = __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(wc > zc)) continue; // NA, outlet, or already down at the terrain
                double m = __builtin_inf();
#pragma unroll
                for (int q = 0; q < 9; q++) {
                    if (q == 4) continue;
                    const double v = __hip_atomic_load(w + (size_t)(rr + q / 3 - 1) * W + (c + q % 3 - 1),
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const double cand = v + ((q & 1) ? eps : eps_d); // q odd: N, W, E, S
                    m = (cand < m) ? cand : m;                       // NaN neighbours never win
                }
                const double nw = (m > zc) ? m : zc;
                if (nw < wc) {
                    __hip_atomic_store(w + i, nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    any = true;
                }
            }
        }
    }
    if (__any(any) && (threadIdx.x & 63) == 0) *changed = 1;
}

// D8 accumulation: every cell sends the number of cells draining through it (itself included) to
// the one neighbour its flow code names.  Kahn's order: a cell is ready when all cells that drain
// into it have delivered.  Counts are integers, so the sums do not depend on the order of the
// atomic adds.  Cells on a closed loop of codes (a pit the fill left: raster's flowdir names the
// smallest rise when there is no drop) are never ready: they keep what their tributaries delivered
// and pass nothing on.  codes: raster (1 E, 2 SE, 4 S, 8 SW, 16 W, 32 NW, 64 N, 128 NE) or TauDEM
// (1 E, 2 NE, 3 N, 4 NW, 5 W, 6 SW, 7 S, 8 SE).
__device__ __forceinline__ long long d8_target(double code, int tau, int r, int c, int H, int W)
{
    if (!(code >= 1.0)) return -1; // NA, 0 (no direction)
    int q;                         // 0 E, 1 SE, 2 S, 3 SW, 4 W, 5 NW, 6 N, 7 NE
    const int k = (int)code;
    if ((double)k != code) return -1;
    if (tau) {
        if (k > 8) return -1;
        q = (9 - k) & 7; // 1 E -> 0, 2 NE -> 7, 3 N -> 6, ..., 8 SE -> 1
    } else {
        if (k > 128 || (k & (k - 1)) != 0) return -1;
        q = __ffs(k) - 1;
    }
    const int dr = (q >= 1 && q <= 3) ? 1 : ((q >= 5) ? -1 : 0);
    const int dc = (q == 0 || q == 1 || q == 7) ? 1 : ((q >= 3 && q <= 5) ? -1 : 0);
    const int rr = r + dr, cc = c + dc;
    if (rr < 0 || rr >= H || cc < 0 || cc >= W) return -1;
    return (long long)rr * W + cc;
}

__global__ __launch_bounds__(kBlock) void k_fa_init(int H, int W, const double* __restrict__ fd,
                                                   const double* __restrict__ elev, int tau,
                                                   unsigned* __restrict__ count, int* __restrict__ indeg)
{
    const int c = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (c >= W) return;
    for (int r = (int)blockIdx.y; r < H; r += (int)gridDim.y) {
        const size_t i = (size_t)r * W + c;
        const double code = fd[i];
        const bool valid = elev ? (elev[i] == elev[i]) : (code == code);
        int n = 0;
#pragma unroll
        for (int q = 0; q < 9; q++) {
            if (q == 4) continue;
            const int rr = r + q / 3 - 1, cc = c + q % 3 - 1;
            if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
            const size_t j = (size_t)rr * W + cc;
            const bool jvalid = elev ? (elev[j] == elev[j]) : (fd[j] == fd[j]);
            if (jvalid && d8_target(fd[j], tau, rr, cc, H, W) == (long long)i) n++;
        }
        count[i] = valid ? 1u : 0u;
        indeg[i] = valid ? n : -1; // -1: not a cell
    }
}

__global__ __launch_bounds__(kBlock) void k_fa_seed(size_t n, const int* __restrict__ indeg,
                                                   unsigned* __restrict__ queue, unsigned* __restrict__ qn)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n && indeg[i] == 0) queue[atomicAdd(qn, 1u)] = (unsigned)i;
}

__device__ __forceinline__ void fa_deliver(unsigned i, int H, int W, const double* __restrict__ fd,
                                           const double* __restrict__ elev, int tau, unsigned* count,
                                           int* indeg, unsigned* qout, unsigned* qout_n)
{
    const int r = (int)(i / (unsigned)W), c = (int)(i % (unsigned)W);
    const long long t = d8_target(fd[i], tau, r, c, H, W);
    if (t < 0) return;
    if (__hip_atomic_load(indeg + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) return; // flows off the land
    const unsigned mine = __hip_atomic_load(count + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicAdd(count + t, mine);
    if (atomicSub(indeg + t, 1) == 1) qout[atomicAdd(qout_n, 1u)] = (unsigned)t; // last delivery: ready
}

__global__ __launch_bounds__(kBlock) void k_fa_level(int H, int W, const double* __restrict__ fd,
                                                    const double* __restrict__ elev, int tau,
                                                    unsigned* count, int* indeg,
                                                    const unsigned* __restrict__ qin, unsigned nin,
                                                    unsigned* qout, unsigned* qout_n)
{
    const unsigned k = blockIdx.x * kBlock + threadIdx.x;
    if (k < nin) fa_deliver(qin[k], H, W, fd, elev, tau, count, indeg, qout, qout_n);
}

// The long thin end of the order (river trunks: a frontier never grows, every cell frees at most the
// one cell it drains to): ONE workgroup walks the remaining levels without returning to the host.
// Every pass empties the current queue; the loop ends when a pass frees nothing.
constexpr int kFaTailBlock = 1024;
__global__ __launch_bounds__(kFaTailBlock) void k_fa_tail(int H, int W, const double* __restrict__ fd,
                                                         const double* __restrict__ elev, int tau,
                                                         unsigned* count, int* indeg, unsigned* qa,
                                                         unsigned na, unsigned* qb, unsigned* qn)
{
    __shared__ unsigned s_n;
    unsigned* qin = qa;
    unsigned* qout = qb;
    unsigned nin = na;
    while (nin > 0) { // uniform: nin is read from LDS by every thread after a barrier
        if (threadIdx.x == 0) *qn = 0;
        __syncthreads();
        for (unsigned k = threadIdx.x; k < nin; k += kFaTailBlock)
            fa_deliver(__hip_atomic_load(qin + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), H, W, fd, elev,
                       tau, count, indeg, qout, qn);
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_n = __hip_atomic_load(qn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        nin = s_n;
        unsigned* t = qin;
        qin = qout;
        qout = t;
        __syncthreads();
    }
}

// Multiple-flow-direction contributing area (Quinn et al. 1991), what `topmodel::topidx()$area`
// computes for R/upslope_area.R:21 (third-party code outside the reference tree: the published
// algorithm, parity unpinned).  Every cell passes its accumulated area to ALL lower neighbours, in
// proportion to tan(beta) * contour length (0.5 dx across an edge, 0.354 dx across a corner); a cell
// starts with its own dx^2.  Order: a cell is ready once every higher neighbour has been settled; it
// then PULLS from those neighbours in a fixed order (E, SE, S, SW, W, NW, N, NE), so the sums are the
// same doubles whatever the launch geometry -- no floating-point atomics.  wsum[n] = the weight
// total of cell n's downslope directions (0: a pit or an outlet: its area stays put).
__device__ __forceinline__ double mfd_weight(double zhi, double zlo, int q, double res)
{
#pragma clang fp contract(off) // the same roundings as a plain C evaluation, whatever the caller fuses
    // weight of the direction q (0 E, 1 SE, ... 7 NE) from a cell at zhi to its neighbour at zlo
    const bool diag = (q & 1) != 0;
    const double dist = diag ? res * 1.4142135623730951 : res;
    const double contour = (diag ? 0.354 : 0.5) * res;
    return ((zhi - zlo) / dist) * contour;
}
__device__ __forceinline__ void d8_offset(int q, int& dr, int& dc)
{
    dr = (q >= 1 && q <= 3) ? 1 : ((q >= 5) ? -1 : 0);
    dc = (q == 0 || q == 1 || q == 7) ? 1 : ((q >= 3 && q <= 5) ? -1 : 0);
}

__global__ __launch_bounds__(kBlock) void k_mfd_init(int H, int W, const double* __restrict__ z, double res,
                                                    double* __restrict__ wsum, int* __restrict__ nhigher,
                                                    double* __restrict__ area)
{
#pragma clang fp contract(off)
    const int c = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (c >= W) return;
    for (int r = (int)blockIdx.y; r < H; r += (int)gridDim.y) {
        const size_t i = (size_t)r * W + c;
        const double zc = z[i];
        double ws = 0.0;
        int nh = 0;
        if (zc == zc) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                int dr, dc;
                d8_offset(q, dr, dc);
                const int rr = r + dr, cc = c + dc;
                if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
                const double zn = z[(size_t)rr * W + cc];
                if (zn < zc) ws += mfd_weight(zc, zn, q, res);
                if (zn > zc) nh++;
            }
        }
        wsum[i] = ws;
        nhigher[i] = (zc == zc) ? nh : -1;
        area[i] = (zc == zc) ? res * res : __builtin_nan("");
    }
}

__device__ __forceinline__ void mfd_settle(unsigned i, int H, int W, const double* __restrict__ z, double res,
                                           const double* __restrict__ wsum, int* nhigher, double* area,
                                           unsigned* qout, unsigned* qout_n)
{
#pragma clang fp contract(off)
    const int r = (int)(i / (unsigned)W), c = (int)(i % (unsigned)W);
    const double zc = z[i];
    double a = res * res;
#pragma unroll
    for (int q = 0; q < 8; q++) { // pull from the higher neighbours, fixed order
        int dr, dc;
        d8_offset(q, dr, dc);
        const int rr = r + dr, cc = c + dc;
        if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
        const size_t j = (size_t)rr * W + cc;
        const double zn = z[j];
        if (zn > zc) {
            const double an = __hip_atomic_load(area + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the neighbour's direction towards this cell is the opposite one: (q + 4) & 7
            a += an * (mfd_weight(zn, zc, (q + 4) & 7, res) / wsum[j]);
        }
    }
    __hip_atomic_store(area + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
#pragma unroll
    for (int q = 0; q < 8; q++) { // one higher neighbour fewer to wait for, for every lower neighbour
        int dr, dc;
        d8_offset(q, dr, dc);
        const int rr = r + dr, cc = c + dc;
        if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
        const size_t j = (size_t)rr * W + cc;
        if (z[j] < zc && atomicSub(nhigher + j, 1) == 1) qout[atomicAdd(qout_n, 1u)] = (unsigned)j;
    }
}

__global__ __launch_bounds__(kBlock) void k_mfd_level(int H, int W, const double* __restrict__ z, double res,
                                                     const double* __restrict__ wsum, int* nhigher,
                                                     double* area, const unsigned* __restrict__ qin,
                                                     unsigned nin, unsigned* qout, unsigned* qout_n)
{
    const unsigned k = blockIdx.x * kBlock + threadIdx.x;
    if (k < nin) mfd_settle(qin[k], H, W, z, res, wsum, nhigher, area, qout, qout_n);
}

// counts -> doubles (NA where there is no cell) and, optionally, the reference's contributing area
// (R/upslope_area.R:85-96): fix_coast -- a cell with an elevation but no count gets 1 -- and
// ups_ncell * cell area, here count * resolution^2 in m2.
__global__ __launch_bounds__(kBlock) void k_fa_finish(size_t n, const unsigned* __restrict__ count,
                                                     const int* __restrict__ indeg,
                                                     const double* __restrict__ resolution,
                                                     double* __restrict__ ups_ncell,
                                                     double* __restrict__ ups_area)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const bool cell = indeg[i] != -1;
    const double k = cell ? (double)count[i] : __builtin_nan("");
    if (ups_ncell) ups_ncell[i] = k;
    if (ups_area) {
        const double res = resolution[i];
        ups_area[i] = k * (res * res);
    }
}

hipError_t launch_sinkfill(int H, int W, const double* elev, double resolution_m, double degree,
                           double* filled, int* sweeps_out, hipStream_t s)
{
    const size_t n = (size_t)H * W;
    if (n == 0) return hipSuccess;
    int* changed = nullptr;
    hipError_t e = hipMallocAsync((void**)&changed, sizeof(int), s);
    if (e != hipSuccess) return e;
    const dim3 g1(nblk((size_t)W, kBlock), (unsigned)(H < 65535 ? H : 65535));
    const unsigned strips = nblk((size_t)H, kStripRows);
    const dim3 g2(nblk((size_t)W, kBlock), strips < 65535 ? strips : 65535);
    hipLaunchKernelGGL(k_sinkfill_init, g1, dim3(kBlock), 0, s, H, W, elev, filled);
    const double eps = tan(degree * kPir) * resolution_m, eps_d = eps * sqrt(2.0);
    int sweeps = 0, h_changed = 1;
    const int batch = 16;
    // a fill level travels at least one cell per sweep: (H + W) * 8 sweeps bound any DEM's longest
    // fill path many times over; the loop normally ends much earlier, on a batch that changed nothing
    const long max_sweeps = 8L * ((long)H + W) + 64;
    while (h_changed && sweeps < max_sweeps) {
        e = hipMemsetAsync(changed, 0, sizeof(int), s);
        if (e != hipSuccess) break;
        for (int k = 0; k < batch; k++)
            hipLaunchKernelGGL(k_sinkfill_sweep, g2, dim3(kBlock), 0, s, H, W, elev, filled, eps, eps_d, changed);
        sweeps += batch;
        e = hipMemcpyAsync(&h_changed, changed, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) break;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) break;
    }
    if (sweeps_out) *sweeps_out = sweeps;
    (void)hipFreeAsync(changed, s);
    if (e == hipSuccess && h_changed) e = hipErrorNotReady; // still relaxing at the bound: never seen
    return e != hipSuccess ? e : hipGetLastError();
}

hipError_t launch_flow_accumulation(int H, int W, const double* fd, const double* elev, int tau,
                                    const double* resolution, double* ups_ncell, double* ups_area,
                                    int* levels_out, hipStream_t s)
{
    const size_t n = (size_t)H * W;
    if (n == 0) return hipSuccess;
    // stream-ordered scratch: count, in-degree, two queues (a cell enters a queue once), 2 counters
    char* w = nullptr;
    hipError_t e = hipMallocAsync((void**)&w, n * 16 + 64, s);
    if (e != hipSuccess) return e;
    unsigned* count = (unsigned*)w;
    int* indeg = (int*)(w + n * 4);
    unsigned* q[2] = {(unsigned*)(w + n * 8), (unsigned*)(w + n * 12)};
    unsigned* qn = (unsigned*)(w + n * 16);
    const dim3 g1(nblk((size_t)W, kBlock), (unsigned)(H < 65535 ? H : 65535));
    hipLaunchKernelGGL(k_fa_init, g1, dim3(kBlock), 0, s, H, W, fd, elev, tau, count, indeg);
    e = hipMemsetAsync(qn, 0, 64, s);
    hipLaunchKernelGGL(k_fa_seed, dim3(nblk(n, kBlock)), dim3(kBlock), 0, s, n, indeg, q[0], qn);
    int cur = 0, levels = 0;
    while (e == hipSuccess) {
        unsigned nin = 0;
        e = hipMemcpyAsync(&nin, qn + cur, sizeof(unsigned), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess || nin == 0) break;
        levels++;
        if (nin <= 16384u) { // the frontier never grows: one workgroup finishes the order
            hipLaunchKernelGGL(k_fa_tail, dim3(1), dim3(kFaTailBlock), 0, s, H, W, fd, elev, tau, count, indeg,
                               q[cur], nin, q[cur ^ 1], qn + (cur ^ 1));
            break;
        }
        e = hipMemsetAsync(qn + (cur ^ 1), 0, sizeof(unsigned), s);
        hipLaunchKernelGGL(k_fa_level, dim3(nblk((size_t)nin, kBlock)), dim3(kBlock), 0, s, H, W, fd, elev, tau,
                           count, indeg, q[cur], nin, q[cur ^ 1], qn + (cur ^ 1));
        cur ^= 1;
    }
    hipLaunchKernelGGL(k_fa_finish, dim3(nblk(n, kBlock)), dim3(kBlock), 0, s, n, count, indeg, resolution,
                       ups_ncell, ups_area);
    if (levels_out) *levels_out = levels;
    const hipError_t e2 = hipGetLastError();
    (void)hipFreeAsync(w, s);
    return e != hipSuccess ? e : e2;
}

hipError_t launch_mfd_area(int H, int W, const double* elev, double resolution_m, double* area,
                           int* levels_out, hipStream_t s)
{
    const size_t n = (size_t)H * W;
    if (n == 0) return hipSuccess;
    // stream-ordered scratch: weight totals (f64), higher-neighbour counts, two queues, 2 counters
    char* w = nullptr;
    hipError_t e = hipMallocAsync((void**)&w, n * 20 + 64, s);
    if (e != hipSuccess) return e;
    double* wsum = (double*)w;
    int* nhigher = (int*)(w + n * 8);
    unsigned* q[2] = {(unsigned*)(w + n * 12), (unsigned*)(w + n * 16)};
    unsigned* qn = (unsigned*)(w + n * 20);
    const dim3 g1(nblk((size_t)W, kBlock), (unsigned)(H < 65535 ? H : 65535));
    hipLaunchKernelGGL(k_mfd_init, g1, dim3(kBlock), 0, s, H, W, elev, resolution_m, wsum, nhigher, area);
    e = hipMemsetAsync(qn, 0, 64, s);
    hipLaunchKernelGGL(k_fa_seed, dim3(nblk(n, kBlock)), dim3(kBlock), 0, s, n, nhigher, q[0], qn); // count 0: peaks
    int cur = 0, levels = 0;
    while (e == hipSuccess) {
        unsigned nin = 0;
        e = hipMemcpyAsync(&nin, qn + cur, sizeof(unsigned), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess || nin == 0) break;
        levels++;
        e = hipMemsetAsync(qn + (cur ^ 1), 0, sizeof(unsigned), s);
        hipLaunchKernelGGL(k_mfd_level, dim3(nblk((size_t)nin, kBlock)), dim3(kBlock), 0, s, H, W, elev,
                           resolution_m, wsum, nhigher, area, q[cur], nin, q[cur ^ 1], qn + (cur ^ 1));
        cur ^= 1;
    }
    if (levels_out) *levels_out = levels;
    const hipError_t e2 = hipGetLastError();
    (void)hipFreeAsync(w, s);
    return e != hipSuccess ? e : e2;
}

hipError_t launch_unswc_grid(size_t ncell, int nt, const double* wn, size_t ldw, const double* soil, double uns_depth,
                             double* w_z, double* wtd, double* se, double* theta_mean, size_t ldo, hipStream_t s)
{
    if (ncell == 0 || nt == 0) return hipSuccess;
    hipLaunchKernelGGL(k_unswc_grid, dim3(nblk(ncell, kBlock)), dim3(kBlock), 0, s, ncell, nt, wn, ldw, soil, uns_depth,
                       w_z, wtd, se, theta_mean, ldo);
    return hipGetLastError();
}

hipError_t launch_unswc(size_t ncell, int nt, const double* wn, size_t ldw, const double* soil,
                        double uns_depth, double* wn_z, double* wtd, double* theta, size_t ldo,
                        hipStream_t s)
{
    if (ncell == 0 || nt == 0) return hipSuccess;
    hipLaunchKernelGGL(k_unswc, dim3(nblk(ncell, kBlock)), dim3(kBlock), 0, s, ncell, nt, wn, ldw,
                       soil, uns_depth, wn_z, wtd, theta, ldo);
    return hipGetLastError();
}

hipError_t launch_synth_statics(uint64_t seed, int H, int W, int64_t cell0, size_t ncell,
                                int terrain, double* lat, double* elev, double* slop, double* asp,
                                double* resolution, double* soil, double* au, hipStream_t s)
{
    if (ncell == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_statics, dim3(nblk(ncell, kBlock)), dim3(kBlock), 0, s, seed, H, W,
                       cell0, ncell, terrain, lat, elev, slop, asp, resolution, soil, au);
    return hipGetLastError();
}

hipError_t launch_synth_forcing(uint64_t seed, const DayRec* days, int t0, int nt, int64_t cell0,
                                size_t ncell, const double* lat, const double* elev, double* sw_in,
                                double* tc, double* pn, size_t ldf, hipStream_t s)
{
    if (ncell == 0 || nt == 0) return hipSuccess;
    unsigned gy = nt < 64 ? (unsigned)nt : 64u;
    hipLaunchKernelGGL(k_synth_forcing, dim3(nblk(ncell, kBlock), gy), dim3(kBlock), 0, s, seed,
                       days, t0, nt, cell0, ncell, lat, elev, sw_in, tc, pn, ldf);
    return hipGetLastError();
}

// ================================================================ static rasters
// What splash.grid derives from the elevation raster before the loop (R/splash.grid.R:96-110,
// R/upslope_area.R:31-33, 140-165) on a north-up longitude/latitude grid, row-major, NaN = NA:
//   slope, aspect   Horn (1981) 3x3 finite differences as raster::terrain(opt=c('slope','aspect'),
//                   unit='degrees', neighbors=8) defines them: cell spacing in metres from the
//                   row's latitude on a sphere of radius 6378137 m, aspect clockwise from north, flat
//                   cells 90 degrees, NA where a neighbour is missing -- which R/splash.grid.R:107
//                   then turns into 0 for cells that have an elevation (applied here when
//                   na_to_zero is set)
//   lat             cell-centre latitude (R/splash.grid.R:101-103)
//   resolution      sqrt(cell area) in metres (:98: sqrt(raster::area(elev))*1000)
//   flowdir         D8 code of the steepest descent among the 8 neighbours as raster::terrain(opt=
//                   'flowdir') encodes it (1 E, 2 SE, 4 S, 8 SW, 16 W, 32 NW, 64 N, 128 NE; the
//                   smallest rise when every neighbour is higher; ties: first in that order,
//                   where raster draws at random)
// One lane per cell; the 3x3 neighbourhood is read straight from global memory (each row is read
// by three consecutive workgroup rows: L2 hits), 8 B in + up to 40 B out per cell: HBM bound.
// The arithmetic of raster::terrain / raster::area lives in a package that is not vendored in the
// reference: this restates the published algorithm, parity unpinned.
// Everything that depends on the row alone -- latitude, metric cell size, the three D8 distances
// and their reciprocals: a cosine, two square roots and five divisions -- is computed once per ROW
// by k_terrain_rows and read by k_terrain with scalar loads (the row is uniform in a workgroup).
constexpr int kRowTerms = 10;
__global__ __launch_bounds__(kBlock) void k_terrain_rows(int H, double ymax, double yres, double xres,
                                                        double* __restrict__ rows)
{
    const int r = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (r >= H) return;
    const double Rearth = 6378137.0;
    const double lat = ymax - ((double)r + 0.5) * yres;
    const double dy = Rearth * (yres * kPir);
    const double dx = Rearth * cos(lat * kPir) * (xres * kPir);
    const double dd = sqrt(dx * dx + dy * dy);
    double* o = rows + (size_t)r * kRowTerms;
    o[0] = lat;
    o[1] = dx;
    o[2] = dy;
    o[3] = dd;
    o[4] = sqrt(dx * dy);
    o[5] = 1.0 / dx;
    o[6] = 1.0 / dy;
    o[7] = 1.0 / dd;
    o[8] = 1.0 / (8.0 * dx);
    o[9] = 1.0 / (8.0 * dy);
}

// One lane walks a strip of kStripRows rows of its column with the 3x3 window in registers: three
// new elevations per cell instead of nine (the window rows slide down), and kStripRows cells per
// lane to amortise the launch of a wavefront.  Grid: x = column blocks of kBlock, y = row strips.
// Quotients by the row's distances are formed with div_cr() (the correctly rounded a/b from
// y = RN(1/b): bit-identical to the division they replace, so the D8 code -- an argmax over
// quotients -- cannot change).
__device__ __forceinline__ void window_row(const double* __restrict__ p, int H, int W, int r, int c,
                                           double& l, double& m, double& rt)
{
    // the three values of row r at columns c-1, c, c+1; NA outside the grid.  The reads come from
    // clamped positions, so none of them depends on a branch (all in flight together).
    const int rc = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
    const double* q = p + (size_t)rc * W;
    const double a = q[c > 0 ? c - 1 : 0], b = q[c], d = q[c < W - 1 ? c + 1 : W - 1];
    const bool rin = r >= 0 && r < H;
    const double na = __builtin_nan("");
    l = (rin && c > 0) ? a : na;
    m = rin ? b : na;
    rt = (rin && c < W - 1) ? d : na;
}

__global__ __launch_bounds__(kBlock) void k_terrain(int H, int W, const double* __restrict__ elev,
                                                   const double* __restrict__ rows, int na_to_zero,
                                                   double* __restrict__ slope,
                                                   double* __restrict__ aspect,
                                                   double* __restrict__ lat_o,
                                                   double* __restrict__ res_o,
                                                   double* __restrict__ flowdir)
{
    const int cidx = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (cidx >= W) return;
    for (int r0 = (int)blockIdx.y * kStripRows; r0 < H; r0 += (int)gridDim.y * kStripRows) {
    double a, b, c, d, z, f, g, h, k; // a b c / d z f / g h k  (north row first)
    window_row(elev, H, W, r0 - 1, cidx, a, b, c);
    window_row(elev, H, W, r0, cidx, d, z, f);
    window_row(elev, H, W, r0 + 1, cidx, g, h, k);
    const int r1 = r0 + kStripRows < H ? r0 + kStripRows : H;
    for (int r = r0; r < r1; r++) {
    // the row after next is requested now and waited for before this row's stores are issued
    // (loads and stores share one counter: see k_run_all), a whole row of arithmetic later
    double gn, hn, kn;
    window_row(elev, H, W, r + 2, cidx, gn, hn, kn);
    const double* s_row = rows + (size_t)r * kRowTerms;
    const size_t i = (size_t)r * W + cidx;
    const double lat = s_row[0], dx = s_row[1], dy = s_row[2], dd = s_row[3];
    const bool inner = r > 0 && r < H - 1 && cidx > 0 && cidx < W - 1;
    double sl = __builtin_nan(""), as = __builtin_nan(""), fd = __builtin_nan("");
    if (inner && z == z) {
        const double sum8 = a + b + c + d + f + g + h + k; // NaN if any neighbour is NA
        if (sum8 == sum8) {
            const double zx = div_cr((c + 2.0 * f + k) - (a + 2.0 * d + g), 8.0 * dx, s_row[8]); // eastward
            const double zy = div_cr((a + 2.0 * b + c) - (g + 2.0 * h + k), 8.0 * dy, s_row[9]); // northward
            sl = fatan(fsqrt_nn(zx * zx + zy * zy)) * (1.0 / kPir);
            if (zx == 0.0 && zy == 0.0) {
                as = 90.0;
            } else {
                as = fatan2(-zx, -zy) * (1.0 / kPir); // azimuth of the downslope direction
                if (as < 0.0) as += 360.0;
            }
        }
        // D8: largest drop per unit distance (diagonals are sqrt(dx^2+dy^2) away)
        const double nb[8] = {f, k, h, g, d, a, b, c};             // E SE S SW W NW N NE
        const double dist[8] = {dx, dd, dy, dd, dx, dd, dy, dd};
        const double idist[8] = {s_row[5], s_row[7], s_row[6], s_row[7], s_row[5], s_row[7], s_row[6], s_row[7]};
        double best = -__builtin_inf();
        int code = 0;
        bool any = false;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (nb[q] == nb[q]) {
                const double drop = div_cr(z - nb[q], dist[q], idist[q]);
                if (drop > best) {
                    best = drop;
                    code = 1 << q;
                }
                any = true;
            }
        }
        if (any) fd = (double)code;
    }
    if (na_to_zero && z == z) { // terraines[is.na(terraines) & !is.na(elev)] <- 0, R/splash.grid.R:107
        if (sl != sl) sl = 0.0;
        if (as != as) as = 0.0;
    }
    asm volatile("" : "+v"(gn), "+v"(hn), "+v"(kn) : : "memory");
    if (lat_o) __builtin_nontemporal_store((z == z) ? lat : __builtin_nan(""), lat_o + i);
    if (res_o) __builtin_nontemporal_store((z == z) ? s_row[4] : __builtin_nan(""), res_o + i);
    if (slope) __builtin_nontemporal_store(sl, slope + i);
    if (aspect) __builtin_nontemporal_store(as, aspect + i);
    if (flowdir) flowdir[i] = fd; // read again by k_ncellflow: default cache policy
    a = d; b = z; c = f; // the window slides down one row
    d = g; z = h; f = k;
    g = gn; h = hn; k = kn;
    }
    }
}
// The reference's LARGE-GRID terrain (R/splash.grid.R:77-94: ncell > 1e7 or 0.1 < res < 1 degree): `gdaldem slope -s 111120
// -compute_edges` and `gdaldem aspect -zero_for_flat -compute_edges` on the DEM written as 16-bit integers
// (R/upslope_area.R:64).  GDAL is a system tool, not vendored: this restates gdaldem's published algorithm (Horn's
// formula; ONE metres-per-degree scale for both axes at every latitude; aspect from the raw differences as an azimuth,
// flat = 0; -compute_edges: on the first / last line the missing row is 2 * line - next with the column index clamped,
// on the first / last column of the other lines the missing column is 2 * column - next; a missing neighbour takes the
// centre's value, a missing centre is NA) -- parity unpinned.  One lane per cell, the nine
// reads hit the L2; 8 B in + 16 B out per cell.
__device__ __forceinline__ double gd_z(const double* __restrict__ e, int W, int r, int c, int round_int)
{
    const double v = e[(size_t)r * W + c];
    return round_int ? rint(v) : v; // the INT2S file: R's round(), half to even (NaN stays NaN)
}
__device__ __forceinline__ double gd_interpol(double a, double b) { return 2.0 * a - b; } // NaN if either is

__global__ __launch_bounds__(kBlock) void k_terrain_gdal(int H, int W, const double* __restrict__ elev, double xres,
                                                        double yres, double scale, int round_int,
                                                        double* __restrict__ slope, double* __restrict__ aspect)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (size_t)H * W) return;
    const int r = (int)(i / (size_t)W), c = (int)(i % (size_t)W);
    double w[9];
    if (H < 2 || W < 2) {
        const double z = gd_z(elev, W, r, c, round_int);
#pragma unroll
        for (int q = 0; q < 9; q++) w[q] = z;
    } else if (r == 0 || r == H - 1) {
        const int cm = c > 0 ? c - 1 : c, cp = c < W - 1 ? c + 1 : c, rn = r == 0 ? 1 : H - 2;
        const int cc[3] = {cm, c, cp};
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const double mid = gd_z(elev, W, r, cc[q], round_int), nxt = gd_z(elev, W, rn, cc[q], round_int);
            const double ext = gd_interpol(mid, nxt);
            w[3 + q] = mid;
            w[r == 0 ? q : 6 + q] = ext;
            w[r == 0 ? 6 + q : q] = nxt;
        }
    } else {
#pragma unroll
        for (int dr = -1; dr <= 1; dr++) {
            const double m = gd_z(elev, W, r + dr, c, round_int);
            double l, rt;
            if (c == 0) {
                rt = gd_z(elev, W, r + dr, 1, round_int);
                l = gd_interpol(m, rt);
            } else if (c == W - 1) {
                l = gd_z(elev, W, r + dr, W - 2, round_int);
                rt = gd_interpol(m, l);
            } else {
                l = gd_z(elev, W, r + dr, c - 1, round_int);
                rt = gd_z(elev, W, r + dr, c + 1, round_int);
            }
            w[3 * (dr + 1)] = l;
            w[3 * (dr + 1) + 1] = m;
            w[3 * (dr + 1) + 2] = rt;
        }
    }
    double sl = __builtin_nan(""), as = __builtin_nan("");
    if (w[4] == w[4]) {
#pragma unroll
        for (int q = 0; q < 9; q++)
            if (w[q] != w[q]) w[q] = w[4];
        const double ax = ((w[2] + w[5] + w[5] + w[8]) - (w[0] + w[3] + w[3] + w[6]));
        const double ay = ((w[6] + w[7] + w[7] + w[8]) - (w[0] + w[1] + w[1] + w[2]));
        const double dx = -ax / xres, dy = ay / yres;
        sl = atan(sqrt(dx * dx + dy * dy) / (8.0 * scale)) * (1.0 / kPir);
        if (ax == 0.0 && ay == 0.0) {
            as = 0.0; // -zero_for_flat
        } else {
            as = atan2(ay, -ax) * (1.0 / kPir);
            as = as > 90.0 ? 450.0 - as : 90.0 - as;
            if (as == 360.0) as = 0.0;
        }
    }
    if (slope) slope[i] = sl;
    if (aspect) aspect[i] = as;
}

hipError_t launch_terrain_gdal(int H, int W, const double* elev, double xres, double yres, double scale, int round_int,
                               double* slope, double* aspect, hipStream_t s)
{
    if (H <= 0 || W <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_terrain_gdal, dim3(nblk((size_t)H * W, kBlock)), dim3(kBlock), 0, s, H, W, elev, xres, yres, scale,
                       round_int, slope, aspect);
    return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void k_ncellflow(int H, int W, const double* __restrict__ fd,
                                                     const double* __restrict__ codes, int pad,
                                                     double* __restrict__ out)
{
    // strips of kStripRows rows per lane, 3x3 window of flow codes in registers (see k_terrain)
    const int c = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (c >= W) return;
    double code[9];
#pragma unroll
    for (int q = 0; q < 9; q++) code[q] = codes[q];
    for (int r0 = (int)blockIdx.y * kStripRows; r0 < H; r0 += (int)gridDim.y * kStripRows) {
        double v[9];
        window_row(fd, H, W, r0 - 1, c, v[0], v[1], v[2]);
        window_row(fd, H, W, r0, c, v[3], v[4], v[5]);
        const int r1 = r0 + kStripRows < H ? r0 + kStripRows : H;
        for (int r = r0; r < r1; r++) {
            window_row(fd, H, W, r + 1, c, v[6], v[7], v[8]);
            const size_t i = (size_t)r * W + c;
            const bool edge = r == 0 || r == H - 1 || c == 0 || c == W - 1;
            int nna = 0, nmatch = 0;
#pragma unroll
            for (int q = 0; q < 9; q++) {
                nna += (v[q] != v[q]) ? 1 : 0;
                nmatch += (v[q] == code[q]) ? 1 : 0;
            }
            double res = (double)(nmatch == 0 ? 1 : nmatch);
            if (nna == 9 || (edge && !pad)) res = __builtin_nan("");
            __builtin_nontemporal_store(res, out + i);
#pragma unroll
            for (int q = 0; q < 6; q++) v[q] = v[q + 3];
        }
    }
}

hipError_t launch_terrain(int H, int W, const double* elev, double ymax, double yres, double xres,
                          int na_to_zero, double* slope, double* aspect, double* lat, double* res,
                          double* flowdir, hipStream_t s)
{
    const size_t n = (size_t)H * W;
    if (n == 0) return hipSuccess;
    double* rows = nullptr; // stream-ordered scratch: 80 bytes per row
    hipError_t e = hipMallocAsync((void**)&rows, (size_t)H * kRowTerms * sizeof(double), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_terrain_rows, dim3(nblk((size_t)H, kBlock)), dim3(kBlock), 0, s, H, ymax, yres, xres,
                       rows);
    const unsigned strips = nblk((size_t)H, kStripRows);
    hipLaunchKernelGGL(k_terrain, dim3(nblk((size_t)W, kBlock), strips < 65535 ? strips : 65535),
                       dim3(kBlock), 0, s, H, W, elev, rows, na_to_zero, slope, aspect, lat, res, flowdir);
    e = hipGetLastError();
    (void)hipFreeAsync(rows, s);
    return e;
    
}

hipError_t launch_ncellflow(int H, int W, const double* fd, const double* codes, int pad,
                            double* out, hipStream_t s)
{
    const size_t n = (size_t)H * W;
    if (n == 0) return hipSuccess;
    const unsigned strips = nblk((size_t)H, kStripRows);
    hipLaunchKernelGGL(k_ncellflow, dim3(nblk((size_t)W, kBlock), strips < 65535 ? strips : 65535), dim3(kBlock),
                       0, s, H, W, fd, codes, pad, out);
    return hipGetLastError();
}

} // namespace splash
