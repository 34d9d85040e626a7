// splash_device.hpp -- per-cell per-day SPLASH v2.0 flux integrator as gfx950 device functions.
//
// One wavefront lane owns one grid cell and carries its five state scalars (wn, snow, qin, td, nd)
// in registers through the day loop.  Everything that depends only on the cell (soil hydraulics,
// terrain trigonometry, pressure, ...) is computed once per kernel in cell_setup(); everything that
// depends only on the calendar day (orbit) comes from a per-day table read with scalar loads.
//
// Reference being replaced (file:line relative to the rsplash tree):
//   SOLAR::calculate_daily_fluxes  src/SOLAR.cpp:77-267
//   EVAP::calculate_daily_fluxes   src/EVAP.cpp:82-264   (+ water properties :291-515)
//   SPLASH::run_one_day/quick_run  src/SPLASH.cpp:920-1588 / 150-918
//   SPLASH::moist_surf / inf_GA    src/SPLASH.cpp:1921-1961 / 1963-2023
//   soil_hydro, snowfall split     R/splash.point.R:232-416, 120-128, 521-578
//
// Numerics: FP64 throughout, except water viscosity which the reference computes in `float`
// (src/EVAP.cpp:405-462); that routine is emulated operation by operation (contraction off,
// glibc's expf algorithm) because Dunne runoff amplifies a 1-ulp(float) viscosity change to ~1e-5.
// NaN-as-control-flow and std::max/min argument-order semantics of the reference are kept.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "splash_kernels.hpp"
#include "splash_math.hpp"

// The day step's functions compile for the device AND for the host (SPL_DEV = __host__ __device__): the host
// backend of the one-cell entry points (splash_host_backend.hip, SPLASH_BACKEND_HOST; round 6) runs the product's
// own arithmetic -- these very functions -- on a CPU core, where a lone wavefront has nothing to offer a single
// site.  What is device-only inside them is switched on __HIP_DEVICE_COMPILE__: tables in LDS (host: the constant
// arrays themselves), the register pins that keep values finished doubles across kernel variants (host: nothing;
// host and device agree to rounding, not to the bit: another libm in cell_setup, IEEE divisions for the
// reciprocal sequences), the wave-uniform predicate.
#define SPL_DEV __host__ __device__ __forceinline__
#define SPL_DEV_NOINL __host__ __device__ inline
#if defined(__HIP_DEVICE_COMPILE__)
#define SPL_FIN(x) asm volatile("" : "+v"(x))            // x is a finished value in a register from here on
#define SPL_FIN3(x, y, z) asm volatile("" : "+v"(x), "+v"(y), "+v"(z))
#else
#define SPL_FIN(x) ((void)0)
#define SPL_FIN3(x, y, z) ((void)0)
#endif

namespace splash {

// bit casts that work on both sides (the __double_as_longlong family is device-only)
SPL_DEV unsigned long long d2u(double x) { return __builtin_bit_cast(unsigned long long, x); }
SPL_DEV double u2d(unsigned long long x) { return __builtin_bit_cast(double, x); }
SPL_DEV uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
SPL_DEV float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }

// ---- call-site counters: profiling build only (-DSPL_SITES, tools/site_profile.sh) ------------
// How often each transcendental call site / rare branch of the day step runs inside k_run_all, per
// lane and per wavefront; printed to stderr after every run_all launch.  Expands to nothing in the
// product build.
#ifdef SPL_SITES
__device__ unsigned long long g_site_hits[2048][2];
__shared__ int s_site_on;
__device__ __forceinline__ void site_hit(int line)
{
    if (!s_site_on) return;
    atomicAdd(&g_site_hits[line & 2047][0], 1ull);
    const unsigned long long m = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&g_site_hits[line & 2047][1], 1ull);
}
#define flog(x) (site_hit(__LINE__), flog(x))
#define fexp(x) (site_hit(__LINE__), fexp(x))
#define facos(x) (site_hit(__LINE__), facos(x))
#define fpow(x, y) (site_hit(__LINE__), fpow(x, y))
#define SPL_SITE() site_hit(__LINE__)
#else
#define SPL_SITE() ((void)0)
#endif

// ---- src/global.cpp:49-83 ------------------------------------------------------------------
constexpr double kA = 91.86328;
constexpr double kAlbSw = 0.30;
constexpr double kB = 0.2012435;
constexpr double kC = 0.25;
constexpr double kD = 0.50;
constexpr double kKfus = 334000.0;
constexpr double kG = 9.80665;
constexpr double kGsc = 1360.8;
constexpr double kL = 0.0065;
constexpr double kMa = 0.028963;
constexpr double kMv = 0.01802;
constexpr double kPo = 101325.0;
constexpr double kR = 8.31447;
constexpr double kTo = 288.15;
constexpr double kPI = 3.141592653589793;
constexpr double kPir = kPI / 180.0;
constexpr double kFluidity = 35187037.0;

// std::max / std::min as the reference uses them: max(a,b) = (a<b)?b:a, min(a,b) = (b<a)?b:a
#define SPL_MAX(a, b) (((a) < (b)) ? (b) : (a))
#define SPL_MIN(a, b) (((b) < (a)) ? (b) : (a))

struct State {
    double wn, snow, qin, td, nd;
};

struct DayConst { // per-calendar-day terms (host table): orbit, src/SOLAR.cpp:98-124, and the
                  // month-dependent factors of frain_func, R/splash.point.R:549-550
    double dr, sd, cd;  // distance factor, sin(delta), cos(delta)
    double ttm_s;       // dsin((m+2)/1.91)
    double inv_trm14;   // 1 / (1.4 * (13.3*(0.55+dsin(m+4))*0.6))
};

struct DayOut {
    double ro, pet, aet, cond, bflow, netr;
};

// Cell-static terms hoisted out of the day loop (the reference recomputes all of them per day).
struct CellConst {
    // SOLAR (src/SOLAR.cpp:129-131, 170, 197)
    double ka, kb, kc;      // a = sd*ka, b = cd*kb, c = cd*kc
    double tau_o, tau_a, tau_binv; // tau_o, 0.1898*tau_o, 1/(0.8102*tau_o)
    // EVAP (src/EVAP.cpp:110, 120)
    double patm;
    float patm_f;
    double visc0;           // viscosity at tw = 0 (every day with tc < 0)
    ViscTables vt;          // per-device lookup tables of the float viscosity (uniform pointers)
    // SPLASH soil column
    double SAT, RES, Au, Ai, inv_Ai;
    double d1000, inv_d1000; // depth*1000 and its inverse
    double depth;
    double theta_s, inv_theta_s, theta_r, dtheta, inv_dtheta; // dtheta = theta_s - theta_r
    double inv_lambda, e3l1; // 1/lambda, 3*lambda+1
    double lambda;
    double bub, bpr, inv_bpr; // bubbling pressure [mm], [cm], 1 / bpr
    double Wmax, wm_th, inv_wr; // src/SPLASH.cpp:1029, Wmax/(depth*1000), 1/(Wmax-RES)
    double int_perm;        // Ksat/fluidity
    double inv_kv24c;       // 1 / (24 * 3.6 * g * int_perm): 1/(24*Ksat_visc) = (visc/pw) * inv_kv24c
    double tan2_slop, inv_cos2; // tan^2(slop), 1/cos^2(slop)
    double h_f;             // Green-Ampt wetting-front suction (src/SPLASH.cpp:1987)
    double kun_exp;         // 3 + 2/lambda
    double so_cellout;      // sid_oct*cellout
    double tu_all;          // (bub/(3*lambda+1)) * (24*cellout*sid_oct)/(1000*Ai)
    double kb_coef;         // log(Kb) = Ksat_visc*kb_coef   (src/SPLASH.cpp:1291-1326)
    double qt_coef;         // Qt = Ksat_visc*qt_coef*hyd_grad_out (src/SPLASH.cpp:1333-1360)
    bool deep;              // depth >= 2.0
};

// ---- small helpers -------------------------------------------------------------------------
SPL_DEV double dsin_deg(double x) { return sin(x * kPir); }

// src/EVAP.cpp:338-389 (Chen et al. 1977) in Horner form for the FP64 path (the evaluation-order
// faithful form below is only needed where the result is rounded to float)
struct DensityPolys { // temperature polynomials of Chen et al. (1977), src/EVAP.cpp:338-384
    double po1000, ko, ca, cb;
};
SPL_DEV DensityPolys density_polys(double tc)
{
    double po = horner3(-(1.00659e-18), tc, (5.044070e-16), -(1.11663e-13), (1.471150e-11));
    po = horner5(po, tc, -(1.35439e-9), (1.022130e-7), -(9.08659e-6), (6.788260e-5), 0.99983952);
    DensityPolys q;
    q.po1000 = (1.0e3) * po;
    q.ko = horner5((1.035530e-7), tc, -(4.91564e-5), 0.01281, -2.29995, 148.1830, 19652.17);
    q.ca = horner4((8.584e-10), tc, -(7.655e-7), (1.324e-4), (5.223e-4), 3.26138);
    q.cb = horner4((4.3220e-12), tc, -(1.0100e-9), (8.69900e-8), -(5.8948e-6), (7.2061e-5));
    return q;
}
SPL_DEV double density_from_polys(const DensityPolys& q, double p)
{
    const double pbar = (1.0e-5) * p;
    const double num = (q.ko + q.ca * pbar + q.cb * (pbar * pbar)); // ~2e4
    return num * frcp(num - pbar) * q.po1000;
}
SPL_DEV double density_h2o_fast(double tc, double p)
{
    return density_from_polys(density_polys(tc), p);
}

// src/EVAP.cpp:338-389 (Chen et al. 1977).  Contraction is off so that the value later rounded
// to float inside the viscosity routine is the same double the reference rounds.
SPL_DEV double density_h2o(double tc, double p)
{
#pragma clang fp contract(off)
    // products are evaluated left to right exactly as written in the reference
    // (c*tc*tc*... == (((c*tc)*tc)*...)), not as c*(tc^k)
    double po = 0.99983952;
    po += (6.788260e-5) * tc;
    po += -(9.08659e-6) * tc * tc;
    po += (1.022130e-7) * tc * tc * tc;
    po += -(1.35439e-9) * tc * tc * tc * tc;
    po += (1.471150e-11) * tc * tc * tc * tc * tc;
    po += -(1.11663e-13) * tc * tc * tc * tc * tc * tc;
    po += (5.044070e-16) * tc * tc * tc * tc * tc * tc * tc;
    po += -(1.00659e-18) * tc * tc * tc * tc * tc * tc * tc * tc;

    double ko = 19652.17;
    ko += 148.1830 * tc;
    ko += -2.29995 * tc * tc;
    ko += 0.01281 * tc * tc * tc;
    ko += -(4.91564e-5) * tc * tc * tc * tc;
    ko += (1.035530e-7) * tc * tc * tc * tc * tc;

    double ca = 3.26138;
    ca += (5.223e-4) * tc;
    ca += (1.324e-4) * tc * tc;
    ca += -(7.655e-7) * tc * tc * tc;
    ca += (8.584e-10) * tc * tc * tc * tc;

    double cb = (7.2061e-5);
    cb += -(5.8948e-6) * tc;
    cb += (8.69900e-8) * tc * tc;
    cb += -(1.0100e-9) * tc * tc * tc;
    cb += (4.3220e-12) * tc * tc * tc * tc;

    double pbar = (1.0e-5) * p;
    double num = (ko + ca * pbar + cb * (pbar * pbar));
    double pw = num / (num - pbar);
    pw *= (1.0e3) * po;
    return pw;
}

// glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c: N=32 table of 2^(i/32), cubic in double).
// std::exp(float) in the reference resolves to this routine; it is NOT correctly rounded
// (0.06 % of arguments differ from (float)exp((double)x)), so the algorithm itself is restated.
// The table is 2^(i/32) correctly rounded to double, minus (i << 47).
#define SPL_EXP2F_TABLE \
    0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL, \
    0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL, \
    0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL, \
    0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL, \
    0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL, \
    0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL, \
    0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL, \
    0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL
#if !defined(__HIP_DEVICE_COMPILE__)
static const uint64_t kExp2fTabHost[32] = {SPL_EXP2F_TABLE};
#endif
__device__ const uint64_t kExp2fTab[32] = {SPL_EXP2F_TABLE};

// The table is read from LDS, not from global memory: vector-memory loads return in order
// (vmcnt), so a table load in the middle of the day step would make its s_waitcnt also wait for
// the next day's forcing, which the kernels prefetch a whole day ahead so that its HBM (and TLB
// miss) latency stays off the critical path.  LDS reads are counted separately (lgkmcnt).
// Every kernel that reaches expf_glibc or flog calls lds_tables_init() first, with all threads of
// the block (it also fills the logarithm's table, splash_math.hpp).
__shared__ uint64_t s_exp2f_tab[32];

__device__ __forceinline__ void lds_tables_init()
{
#ifdef SPL_SITES
    if (threadIdx.x == 0) s_site_on = 0;
#endif
    math_tables_init();
    if (threadIdx.x < 32) s_exp2f_tab[threadIdx.x] = kExp2fTab[threadIdx.x];
    if (blockDim.x < 32 && threadIdx.x == 0)
        for (int i = (int)blockDim.x; i < 32; i++) s_exp2f_tab[i] = kExp2fTab[i];
    __syncthreads();
}

SPL_DEV float expf_glibc(float x)
{
    constexpr double N = 32.0;
    constexpr double InvLn2N = 0x1.71547652b82fep+0 * N;
    constexpr double Shift = 0x1.8p+52;
    constexpr double C0 = 0x1.c6af84b912394p-5 / N / N / N;
    constexpr double C1 = 0x1.ebfce50fac4f3p-3 / N / N;
    constexpr double C2 = 0x1.62e42ff0c52d6p-1 / N;
    double xd = (double)x;
    // |x| >= 88: overflow/underflow paths of the original are unreachable for viscosity
    // arguments (0 <= x < 8); handled generically for completeness.
    if (!(fabsf(x) < 88.0f)) { // (the short exp of splash_math.hpp: four inlined copies of the
        if (x != x) return x;  //  library routine cost the day loop registers for nothing)
        if (x > 0.0f) return x > 88.72283f ? __builtin_inff() : (float)fexp(xd);
        return x < -103.97208f ? 0.0f : (float)fexp(xd);
    }
    double z = InvLn2N * xd;
    double kd = z + Shift;
    uint64_t ki = (uint64_t)d2u(kd);
    kd -= Shift;
    double r = z - kd;
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t t = s_exp2f_tab[ki & 31];
#else
    uint64_t t = kExp2fTabHost[ki & 31];
#endif
    t += ki << 47;
    double s = u2d(t);
    double zz = C0 * r + C1;
    double r2 = r * r;
    double y = C2 * r + 1.0;
    y = zz * r2 + y;
    y = y * s;
    return (float)y;
}

// x^2..x^6 rounded once from a double-double product chain (error ~1e-32 relative before the
// final rounding), i.e. correctly rounded except in astronomically rare ties.
SPL_DEV void pow_int_cr(double x, double& p2, double& p3, double& p4, double& p5,
                                           double& p6)
{
#pragma clang fp contract(off)
    // (h, l) <- (h, l) * x  with h*x split exactly by FMA
    double h = x * x, l = fma(x, x, -h);
    p2 = h; // x*x is the correctly rounded square already
#define STEP(out)                         \
    {                                     \
        double ph = h * x;                \
        double pl = fma(h, x, -ph);       \
        pl = fma(l, x, pl);               \
        double sh = ph + pl;              \
        l = pl - (sh - ph);               \
        h = sh;                           \
        out = h;                          \
    }
    STEP(p3) STEP(p4) STEP(p5) STEP(p6)
#undef STEP
}

// src/EVAP.cpp:405-462 -- Huber et al. (2009) viscosity in the reference's float/double mix.
// Every rounding point of the C++ original is kept: a `float` variable is a rounding to float;
// an expression containing a double literal is evaluated in double.
//
// The routine factors into three pure functions of FLOAT arguments:
//   visc_tbar_part(tbar)  -> mu0, ctbar            (:416-424, :439)   tbar = (float)((tw+273.15)/647.096)
//   visc_rbar_part(rbar)  -> coef2[0..5]           (:441-448)         rbar = rho/322
//   visc_combine(...)     -> mu                    (:443, :449-459)
// A float has few values in the range rbar can take (2.3e5 for water between 0 and 60 degC), so
// visc_rbar_part -- 21 float-rounded accumulations over correctly rounded integer powers -- is
// TABULATED once per device by running exactly this code for every float of the range
// (k_visc_tables, 7.5 MB): the day loop replaces ~130 instructions by one 32-byte gather, and the
// result is the same bits by construction.  visc_tbar_part has a cheaper evaluation whose
// roundings to float are each checked (visc_tbar_part_fast; a table of it would be 23 MB and miss
// the L2: measured, 1 % faster in run_all, 6 % slower in the latency-bound spin-up tail, and
// 29 GB of extra fabric traffic per launch).  Outside the table's range, and in the (rare) cases
// where a cheaper evaluation cannot be proved to round to the same float, the whole routine is
// evaluated directly (viscosity_h2o).
// (struct ViscTables { rtab [NR][8]: coef2[0..5] + 2 padding } is declared in splash_kernels.hpp;
// index = bits(rbar) - kViscR0)
constexpr uint32_t kViscR0 = 0x40436000u; // rbar = 3.05273...  (rho = 982.98: water at ~60.5 degC)
constexpr uint32_t kViscNR = 0x00039000u; // up to rbar = 3.10840 (rho = 1000.9)

SPL_DEV void visc_tbar_part(float tbar, float& mu0_out, float& ctbar_out)
{
#pragma clang fp contract(off)
    float tbarx = (float)sqrt((double)tbar);                           // :417 pow(tbar,0.5)
    float tbar2 = tbar * tbar;                                         // :418
    float tbar3 = tbar * tbar * tbar;                                  // :419
    float mu0 = (float)(1.67752 + 2.20462 / (double)tbar + 0.6366564 / (double)tbar2 -
                        0.241605 / (double)tbar3);                     // :423
    mu0 = (float)(1e2 * (double)tbarx / (double)mu0);                  // :424
    mu0_out = mu0;
    ctbar_out = (float)((1.0 / (double)tbar) - 1.0);                   // :439
}

SPL_DEV void visc_rbar_part(float rbar, float (&c)[6])
{
#pragma clang fp contract(off)
    // Table 3 of Huber et al. (:427-435); the literals are rounded to float as in the
    // reference's `float h_array[7][6]`.  Zero entries add exactly 0 and are skipped.
    double rb = (double)rbar - 1.0;                                    // pow(rbar - 1.0, j) base
    // Integer powers as the correctly rounded value of the exact power (what glibc's pow returns
    // except within ~0.01 ulp of a rounding tie): a float-rounding of the running sum flips on a
    // 1-ulp(double) difference about once per 1e8 cell-days, which at grid scale is every run.
    double rb2, rb3, rb4, rb5, rb6;
    pow_int_cr(rb, rb2, rb3, rb4, rb5, rb6);
#define H(v) ((double)(float)(v))
#define ACC(c2, h, pw_) c2 = (float)((double)(c2) + H(h) * (pw_))
    float c2;
    c2 = 0.0f;
    ACC(c2, 0.520094, 1.0); ACC(c2, 0.222531, rb); ACC(c2, -0.281378, rb2);
    ACC(c2, 0.161913, rb3); ACC(c2, -0.0325372, rb4);
    c[0] = c2;
    c2 = 0.0f;
    ACC(c2, 0.0850895, 1.0); ACC(c2, 0.999115, rb); ACC(c2, -0.906851, rb2);
    ACC(c2, 0.257399, rb3);
    c[1] = c2;
    c2 = 0.0f;
    ACC(c2, -1.08374, 1.0); ACC(c2, 1.88797, rb); ACC(c2, -0.772479, rb2);
    c[2] = c2;
    c2 = 0.0f;
    ACC(c2, -0.289555, 1.0); ACC(c2, 1.26613, rb); ACC(c2, -0.489837, rb2);
    ACC(c2, 0.0698452, rb4); ACC(c2, -0.00435673, rb6);
    c[3] = c2;
    c2 = 0.0f;
    ACC(c2, -0.257040, rb2); ACC(c2, 0.00872102, rb5);
    c[4] = c2;
    c2 = 0.0f;
    ACC(c2, 0.120573, rb); ACC(c2, -0.000593264, rb6);
    c[5] = c2;
#undef ACC
#undef H
}

SPL_DEV float visc_combine(float rbar, float mu0, float ctbar, const float (&c)[6])
{
#pragma clang fp contract(off)
    const float mu_ast = 1e-6f;
    double cd_ = (double)ctbar;
    double ct2, ct3, ct4, ct5, ct6;
    pow_int_cr(cd_, ct2, ct3, ct4, ct5, ct6);                         // coef1 = (float)pow(ctbar, i), :443
    float mu1 = 0.0f;
    mu1 = mu1 + 1.0f * c[0];                                          // :449, i = 0..5
    mu1 = mu1 + (float)cd_ * c[1];
    mu1 = mu1 + (float)ct2 * c[2];
    mu1 = mu1 + (float)ct3 * c[3];
    mu1 = mu1 + (float)ct4 * c[4];
    mu1 = mu1 + (float)ct5 * c[5];
    mu1 = expf_glibc(rbar * mu1);                                      // :451
    float mu_bar = mu0 * mu1;                                          // :456
    return mu_bar * mu_ast;                                            // :459
}

SPL_DEV float viscosity_h2o(float tc, float p)
{
#pragma clang fp contract(off)
    const float tk_ast = 647.096f;
    const float rho_ast = 322.0f;
    float rho = (float)density_h2o((double)tc, (double)p);            // :413
    float tbar = (float)(((double)tc + 273.15) / (double)tk_ast);      // :416
    float rbar = rho / rho_ast;                                        // :420
    float mu0, ctbar, c[6];
    visc_tbar_part(tbar, mu0, ctbar);
    visc_rbar_part(rbar, c);
    return visc_combine(rbar, mu0, ctbar, c);
}

// (double) -> float with a proof obligation: `v` approximates, to ~1e-15 relative, a real number
// whose correctly rounded float is wanted (the reference rounds a slightly different double
// approximation of the same real).  The two round alike unless v lies within 2^-43 relative of a
// float rounding boundary (a midpoint of two adjacent floats: low 29 mantissa bits = 0x10000000);
// then `risky` is raised and the caller falls back to the reference's own evaluation.
SPL_DEV float to_float_checked(double v, bool& risky)
{
    const uint32_t lo = (uint32_t)d2u(v) & 0x1FFFFFFFu;
    risky |= (lo - (0x10000000u - 512u)) < 1024u;
    return (float)v;
}

// visc_tbar_part() with reciprocal-based quotients and a fast square root, every rounding to float
// checked (to_float_checked): ~60 instructions instead of three IEEE divisions, a correctly
// rounded square root and two more divisions -- or instead of a 23 MB table whose gathers miss L2.
SPL_DEV void visc_tbar_part_fast(float tbar, double t1, double t2, double t3,
                                                    double r, float& mu0_out, float& ctbar_out,
                                                    bool& risky)
{
#pragma clang fp contract(off)
    // t1 = tbar, t2 = (float)(tbar*tbar), t3 = (float)(tbar*tbar*tbar) (:418-419) and
    // r = 1/(t1*t2*t3) come from the caller, which takes r out of a reciprocal it needs anyway:
    // 1/t1, 1/t2, 1/t3 from the one reciprocal of the product (all three ~0.1 .. 0.5)
    const double i1 = r * (t2 * t3), i2 = r * (t1 * t3), i3 = r * (t1 * t2);
    const float den = to_float_checked(1.67752 + 2.20462 * i1 + 0.6366564 * i2 - 0.241605 * i3, risky);
    const float tbarx = to_float_checked(fsqrt_nn(t1), risky);            // :417
    mu0_out = to_float_checked(1e2 * (double)tbarx * frcp((double)den), risky); // :424
    // :439: 1/tbar - 1 is ~1.0..1.4; the subtraction is exact enough (error of i1 is 2e-16 * 2.4)
    ctbar_out = to_float_checked(i1 - 1.0, risky);
}

// The temperature-only part of a cell-day that needs memory: water density (double path) and the
// table entries of the float viscosity.  It is evaluated at the TOP of the day, before the next
// day's forcing is requested, so that the two gathers (L2 / Infinity-Cache hits, ~1 us) are in
// flight during the radiation part of the day step and return before the forcing prefetch
// (vector-memory loads return in order).
struct DayPre {
    double pw;          // density_h2o(tc, patm), src/EVAP.cpp:113
    uint32_t cw[6];     // coef2[0..5] of the viscosity (table), as the raw words the gather returns:
                        // nothing may touch them before day_viscosity(), or the compiler waits for
                        // the gather where it was issued instead of where its result is needed
    float mu0, ctbar;
    float rbar;
    int mode;           // 0: tc < 0 (viscosity at tw = 0 is a cell constant)  1: tables  2: direct
};

SPL_DEV DayPre day_prefetch(const CellConst& c, double tc)
{
#pragma clang fp contract(off)
    DayPre q;
    const DensityPolys dq = density_polys(tc);
#if defined(SPL_ABLATE) && SPL_ABLATE == 5
    q.pw = 1000.0 - 0.1 * tc;
#else
    q.pw = density_from_polys(dq, c.patm);
#endif
    q.mode = 0;
    q.rbar = q.mu0 = q.ctbar = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; k++) q.cw[k] = 0u;
#if !(defined(SPL_ABLATE) && SPL_ABLATE == 1)
    uint32_t ir = 0;
    if (!(tc < 0.0)) { // tw = tc (src/EVAP.cpp:100-105)
        const float tcf = (float)tc;
        // forcing that is not float-representable, or no tables: the routine is evaluated directly
        bool risky = ((double)tcf != tc) || c.vt.rtab == nullptr;
        // density at the float-rounded pressure (:413): same temperature polynomials
        const double pbar = (1.0e-5) * (double)c.patm_f;
        const double num = (dq.ko + dq.ca * pbar + dq.cb * (pbar * pbar));
        const float tbar = to_float_checked(((double)tcf + 273.15) * (1.0 / 647.09600830078125), risky); // :416
        // the density quotient and the three quotients of the tbar part share ONE reciprocal,
        // 1/((num - pbar) * t1*t2*t3): a reciprocal is a quarter-rate seed and two Newton steps
        // (~9 issue slots), splitting it afterwards two multiplies
        const float tbar2 = tbar * tbar;                                   // :418
        const float tbar3 = tbar * tbar * tbar;                            // :419
        const double t1 = (double)tbar, t2 = (double)tbar2, t3 = (double)tbar3;
        const double den_rho = num - pbar, ttt = t1 * t2 * t3;
        const double rr = frcp(den_rho * ttt);
        const float rho = to_float_checked(num * (rr * ttt) * dq.po1000, risky);
        const float rbar = to_float_checked((double)rho * (1.0 / 322.0), risky);          // :420
        ir = f2u(rbar) - kViscR0;
        risky |= (ir >= kViscNR);
        visc_tbar_part_fast(tbar, t1, t2, t3, rr * den_rho, q.mu0, q.ctbar, risky);
        q.mode = risky ? 2 : 1;
        q.rbar = rbar;
        ir = risky ? 0u : ir;
    }
    // The gather is issued by EVERY lane (entry 0 where the table is not used): a load under a
    // lane predicate ends in a join at which the compiler settles all outstanding loads.
    if (c.vt.rtab != nullptr) { // uniform
        const uint4 a = *reinterpret_cast<const uint4*>(c.vt.rtab + (size_t)ir * 8);
        const uint2 b = *reinterpret_cast<const uint2*>(c.vt.rtab + (size_t)ir * 8 + 4);
        q.cw[0] = a.x; q.cw[1] = a.y; q.cw[2] = a.z; q.cw[3] = a.w; q.cw[4] = b.x; q.cw[5] = b.y;
    }
#endif
    return q;
}

// viscosity of the day [Pa s] from the prefetched pieces
SPL_DEV double day_viscosity(const CellConst& c, const DayPre& q, double tc)
{
    double visc = c.visc0;
    if (q.mode == 1) {
        SPL_SITE();
        float cf[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            uint32_t w = q.cw[k];
            SPL_FIN(w); // first touch of the gathered words: HERE, not at the load
            cf[k] = u2f(w);
        }
        visc = (double)visc_combine(q.rbar, q.mu0, q.ctbar, cf);
    } else if (q.mode == 2) {
        SPL_SITE();
        visc = (double)viscosity_h2o((float)tc, c.patm_f);
    }
    return visc;
}

// src/EVAP.cpp:491-515
SPL_DEV double specific_heat(double tc)
{
    double cp;
    if (tc < 0) {
        cp = 1004.5714270;
    } else if (tc > 100) {
        cp = 2031.2260590;
    } else {
        cp = horner5((5.071307038e-10), tc, -(8.830478888e-8), (6.212300300e-6), -(1.631537093e-4),
                     (2.050632750e-3), 1.0045714270);
        cp *= (1.0e3);
    }
    return cp;
}

// ---- cell setup ----------------------------------------------------------------------------
// Loads one cell's raw parameters (cellpar rows) and derives the day-loop constants.
// cellout is passed separately because splash.point overwrites it with the aridity index
// between the two spin-ups (R/splash.point.R:150).  Runs once per kernel per cell, so it uses
// the library pow/exp/log throughout.
SPL_DEV_NOINL void cell_setup(CellConst& c, const double* __restrict__ par, size_t ld,
                                  size_t cell, double cellout, ViscTables vt = ViscTables{nullptr})
{
    c.vt = vt;
    const double lat = par[P_LAT * ld + cell], elv = par[P_ELV * ld + cell];
    const double slop = par[P_SLOP * ld + cell], asp = par[P_ASP * ld + cell];
    const double SAT = par[P_SAT * ld + cell], WP = par[P_WP * ld + cell];
    const double FC = par[P_FC * ld + cell], Ksat = par[P_KSAT * ld + cell];
    const double lambda = par[P_LAMBDA * ld + cell], depth = par[P_DEPTH * ld + cell];
    const double bub = par[P_BUB * ld + cell], RES = par[P_RES * ld + cell];
    const double Au = par[P_AU * ld + cell], Ai = par[P_AI * ld + cell];
    const double cellin = par[P_CELLIN * ld + cell];

    // SOLAR terrain terms, src/SOLAR.cpp:129-131
    double sl = sin(lat * kPir), cl = cos(lat * kPir);
    double ss = sin(slop * kPir), cs = cos(slop * kPir);
    double sa = sin(asp * kPir), ca = cos(asp * kPir);
    c.ka = cl * ss * ca - sl * cs;
    c.kb = cl * cs + sl * ss * ca;
    c.kc = ss * sa;
    c.tau_o = (kC + kD) * (1.0 + (2.67e-5) * elv); // :170
    c.tau_a = c.tau_o * 0.1898;                    // :197
    c.tau_binv = 1.0 / (c.tau_o * (1 - 0.1898));

    // EVAP::elv2pres, src/EVAP.cpp:319-336
    {
        double ep = (kG * kMa) / (kR * kL);
        double pa = (1.0 - elv * kL / kTo);
        c.patm = pow(pa, ep) * kPo;
        c.patm_f = (float)c.patm;
        c.visc0 = (double)viscosity_h2o(0.0f, c.patm_f); // tw = 0 whenever tc < 0 (:100-105)
    }

    c.SAT = SAT;
    c.RES = RES;
    c.Au = Au;
    c.Ai = Ai;
    c.inv_Ai = 1.0 / Ai;
    c.depth = depth;
    c.d1000 = depth * 1000.0;
    c.inv_d1000 = 1.0 / c.d1000;
    c.lambda = lambda;
    c.bub = bub;
    c.bpr = bub / 10; // src/SPLASH.cpp:1940
    c.inv_bpr = 1.0 / c.bpr;
    c.deep = (depth >= 2.0);

    // src/SPLASH.cpp:971-993
    const double KG_o = 1000.0 / (997 * kG);
    c.theta_s = SAT / c.d1000;
    c.inv_theta_s = 1.0 / c.theta_s;
    c.theta_r = RES / c.d1000;
    c.dtheta = c.theta_s - c.theta_r;
    c.inv_dtheta = 1.0 / c.dtheta;
    const double theta_fc = FC / c.d1000;
    const double theta_wp = WP / c.d1000;
    const double theta_q0 = theta_wp + 0.001;
    const double sid_oct = sqrt(Ai / (2.0 * (1 + sqrt(2.0))));
    c.inv_lambda = 1 / lambda;
    c.e3l1 = 3.0 * lambda + 1.0;
    c.kun_exp = 3.0 + (2.0 / lambda);

    // src/SPLASH.cpp:1027-1029
    const double coeff_A = exp(log(33.0) + (1.0 / lambda) * log(theta_fc));
    c.Wmax = pow((coeff_A * KG_o / (depth)), (1.0 / ((1 / lambda) + 1.0))) * (c.d1000);
    c.wm_th = c.Wmax / c.d1000;
    c.inv_wr = 1.0 / (c.Wmax - RES);

    c.int_perm = Ksat / kFluidity; // :1260
    c.inv_kv24c = 1.0 / (24.0 * 3.6 * kG * c.int_perm);
    const double tan_slop = tan(slop * kPir); // :1278
    c.tan2_slop = tan_slop * tan_slop;
    c.inv_cos2 = 1.0 / (cs * cs);  // :2011
    c.h_f = ((2 + 3 * lambda) / (1 + 3 * lambda)) * (bub / 2); // :1987
    const double tu_coef = bub / c.e3l1;
    c.so_cellout = sid_oct * cellout;
    c.tu_all = tu_coef * ((24.0 * cellout * sid_oct) / (1000.0 * Ai));

    // Recession constant: log(Kb) = Ksat_visc * kb_coef, src/SPLASH.cpp:1291-1326
    {
        double psi_q0 = bub / pow(((theta_q0 - c.theta_r) / c.dtheta), c.inv_lambda);
        double wtd_q0 = ((bub - psi_q0) / 1000.0);
        if (wtd_q0 < 0.0 || isnan(wtd_q0)) {
            wtd_q0 = 0.0;
        } else if (wtd_q0 > depth) {
            wtd_q0 = depth;
        }
        double tq0 = tu_coef * (pow((bub / psi_q0), c.e3l1) -
                                pow((bub / (psi_q0 + (wtd_q0 * 1000.0))), c.e3l1));
        double q_q0 = tq0 * tan_slop * ((24.0 * sid_oct * cellout) / (1.0e6));
        double acs_qs = depth * sid_oct * cellout;
        double q_qs = tan_slop * 24.0 * acs_qs / 1000.0;
        c.kb_coef = (q_q0 - q_qs) / ((SAT - WP) * (Ai / 1000.0));
    }
    // Drainage at Wmax: Qt = Ksat_visc*qt_coef*hyd_grad_out, src/SPLASH.cpp:1333-1360
    {
        double th = c.wm_th;
        double psi_m = bub / pow(((th - c.theta_r) / c.dtheta), c.inv_lambda);
        double wtd = ((bub - psi_m) / 1000.0);
        if (wtd < 0.0 || isnan(wtd)) {
            wtd = 0.01;
        } else if (wtd > depth) {
            wtd = depth;
        }
        double acs = (depth - wtd) * cellin * sid_oct;
        double to_uns = tu_coef * (pow((bub / psi_m), c.e3l1) -
                                   pow((bub / (psi_m + (wtd * 1000.0))), c.e3l1));
        double qo_uns = to_uns * ((24.0 * cellin * sid_oct) / (1.0e6));
        double qo_sat = 24.0 * acs / 1000.0;
        c.qt_coef = qo_sat + qo_uns;
    }
}

// ---- SOLAR + EVAP ---------------------------------------------------------------------------
struct Etr { // src/etr.h:44-56 (fields the water balance needs)
    double cond, pet, aet, snowmelt, sublimation, pw, rn_d, visc, econ, pet_max;
};

// hour angle (radians) and its sine from the cosine, with the reference's saturation at
// >= 1 / <= -1 (src/SOLAR.cpp:149-159, 231-240; src/EVAP.cpp:209-218).  sin(acos x) is taken as
// sqrt((1-x)(1+x)) (both factors exact near |x| = 1), which replaces a library sin per angle.
// Branch-free: the argument is saturated with selects (NaN stays NaN, as in the reference's else
// branch) and acos(+-1) = 0 / pi fall out of the same evaluation, so the three-way branch and its
// exec-mask bookkeeping disappear.  (sin(180 deg) is 0 here and 1.2e-16 in the reference.)
SPL_DEV void hour_angle(double cosv, bool at_one_is_zero, double& h,
                                           double& sinh_)
{
    // saturation by max/min: one instruction each instead of compare + two selects.  A NaN cosine
    // comes out as -1 (h = pi) rather than NaN; every caller multiplies the angle with the NaN's
    // own source (rw, rnl, econ: evap_update; a NaN ru/rv is a NaN cell), so the day's results are
    // NaN all the same (test_gaps_in_the_forcing_propagate_like_NA_in_R).
    double x = at_one_is_zero ? cosv : -cosv;
    x = fmin(fmax(x, -1.0), 1.0);
#if defined(SPL_ABLATE) && SPL_ABLATE == 2
    h = 1.0 - x;
    sinh_ = x;
#else
    h = facos(x);
    sinh_ = fsqrt_nn((1.0 - x) * (1.0 + x));
#endif
}

// two hour angles whose cosines saturate at >= 1 -> 0 (hn and hi of evap_update), see hour_angle()
SPL_DEV void hour_angle2(double cos1, double cos2, double& h1, double& sinh1,
                                            double& h2, double& sinh2)
{
    const double x1 = fmin(fmax(cos1, -1.0), 1.0), x2 = fmin(fmax(cos2, -1.0), 1.0);
#if defined(SPL_ABLATE) && SPL_ABLATE == 2
    h1 = 1.0 - x1; sinh1 = x1; h2 = 1.0 - x2; sinh2 = x2;
#else
    facos2(x1, x2, h1, h2);
    sinh1 = fsqrt_nn((1.0 - x1) * (1.0 + x1));
    sinh2 = fsqrt_nn((1.0 - x2) * (1.0 + x2));
#endif
}

// src/SOLAR.cpp:77-267 + src/EVAP.cpp:82-264 for one cell-day, in two parts.
//
// day_terms():  everything that depends only on the cell, the calendar day and the day's forcing
//   -- the slope geometry and the sunset angle, the cloud factor and net longwave, the water
//   properties (saturation slope, latent heat, density, psychrometric constant, viscosity) and
//   what follows from them -- 14 numbers and a flag.  They do not depend on the state (soil water, snow pack),
//   so a spin-up pass after the first finds the same numbers for every day of the year: the
//   tail of the spin-up (cells that run hundreds of passes, alone on their SIMD, paying the full
//   instruction latency of every day step) stores them once and reads them back (k_spin_phase).
// evap_update(): the state-dependent rest (albedo, net radiation and its cross-over angle, the
//   supply-limited evaporation, snowmelt).
// Both callers go through terms_fence(), which pins every term to a register as a finished
// double, so that the compiler cannot contract a multiply of day_terms() into an add of the
// update: the results are bit-identical whether the terms were computed or loaded.
// sw is the supply fraction (src/SPLASH.cpp:1042-1048), snow already includes today's snowfall.
// Angles are kept in radians (the reference converts to degrees and back at every use).
struct DayTerms {
    double ru, rv, ruv;   // slope-corrected geometry, ru/rv (src/SOLAR.cpp:139-140); rv = cos(delta)*kb
    double hs, sin_hs;    // sunset angle on the slope and its sine (:149-159)
    double xrw;           // rw = (1 - albedo) * xrw (:220-224)
    double rnl;           // net longwave (:203)
    double s, g;          // saturation slope, psychrometric constant (src/EVAP.cpp:111, 115)
    double econ, inv_econ, eetc; // :117, its reciprocal, 1e3*s/(lv*pw*(s+0.24g)) (:129)
    double pw;            // water density (:113)
    double visc;          // viscosity: a FLOAT by construction (:405-462), held in a double
    bool warm;            // tc >= 3.0: snow melts (:233)
};
// what the spin-up's cache keeps of them per cell-day: 13 doubles (rv is cos(delta) of the day times a
// cell constant: recomputed) and one word holding the viscosity's 32 bits and the flag
constexpr int kNTermWords = 13;

SPL_DEV void terms_fence(DayTerms& T)
{
#define SPL_PIN(x) SPL_FIN(x)
    SPL_PIN(T.ru); SPL_PIN(T.rv); SPL_PIN(T.ruv); SPL_PIN(T.hs); SPL_PIN(T.sin_hs); SPL_PIN(T.xrw);
    SPL_PIN(T.rnl); SPL_PIN(T.s); SPL_PIN(T.g); SPL_PIN(T.econ); SPL_PIN(T.inv_econ); SPL_PIN(T.eetc);
    SPL_PIN(T.pw); SPL_PIN(T.visc);
#undef SPL_PIN
}

// What the water balance takes from the day's viscosity and density: 1/(pw*kfus) when snow can melt
// (src/EVAP.cpp:233-236), Ksat_visc (src/SPLASH.cpp:1260-1261) and 1/(24*Ksat_visc) (:1281), from ONE
// reciprocal of visc*pw.  Evaluated by the update (not kept among the terms: three doubles of the
// spin-up's cached record for ten instructions), so run_all and every spin-up variant execute the
// same operations on the same two finished numbers.
struct ViscTerms {
    double ipk, kv, ikv24;
};
SPL_DEV ViscTerms visc_terms(const CellConst& c, const DayTerms& T)
{
    ViscTerms v;
    const double r_v = frcp(T.visc * T.pw);
    const double inv_pw = r_v * T.visc;
    v.ipk = T.warm ? inv_pw * (1.0 / kKfus) : 0.0;
    v.kv = c.int_perm * ((T.pw * kG) * (r_v * T.pw)) * 3.6;
    v.ikv24 = (T.visc * inv_pw) * c.inv_kv24c;
    // finished doubles, as when they come back from the tail's cache (a product that is later an
    // addend -- r - Ksat in inf_GA -- could otherwise be contracted into an FMA in one variant only)
    SPL_FIN3(v.ipk, v.kv, v.ikv24);
    return v;
}

SPL_DEV void day_terms(const CellConst& c, const DayConst& d, const DayPre& pre,
                                          double sw_in, double tc, DayTerms& T)
{
    // ----- SOLAR
    const double a = d.sd * c.ka, b = d.cd * c.kb, cc = d.cd * c.kc;
    const double b2c2 = b * b + cc * cc;
    const double dd = b2c2 - a * a;
    const double acc = a * cc;
    const double num = (dd < 0) ? acc : (acc + b * fsqrt_nn(dd < 0 ? 0.0 : dd)); // :133-137
    // 1/b2c2 and 1/rv (rv = b = cos(delta)*kb, an ordinary number for any real cell) from one
    // reciprocal of their product
    const double r_sol = frcp(b2c2 * b);
    const double sinfirst = num * (r_sol * b);
    const double ru = -1 * a + cc * sinfirst;
    const double rv = b;
    const double ruv = ru * (r_sol * b2c2);

    double hs, sin_hs; // :149-159: ruv >= 1 polar day (hs = 180), ruv <= -1 polar night (hs = 0)
    hour_angle(ruv, false, hs, sin_hs);
    const double geo = (ru * hs + rv * sin_hs);
    const double ra_d = (86400.0 / kPI) * d.dr * kGsc * geo; // :164-165

    const double r_in = 86400 * sw_in;
    // r_in/ra_d serves twice: as the transmittivity (:177) and, times dr*Gsc, as the
    // r_in/((86400/pi)*geo) of :223 (ra_d = (86400/pi)*dr*Gsc*geo)
    const double tau_raw = r_in * frcp(ra_d);
    double tau; // :174-178
    if (isnan(ra_d) || r_in == 0 || ra_d < r_in) {
        tau = c.tau_o;
    } else {
        tau = tau_raw; // here ra_d >= r_in > 0
    }
#if defined(SPL_ABLATE) && SPL_ABLATE == 4
    double sf = (tau - c.tau_a) * c.tau_binv;
#else
    double sf = fpow((tau - c.tau_a) * c.tau_binv, (1 / 0.7410)); // :197
#endif
    if (isnan(sf)) {
        sf = 0.0;
    } else if (sf > 1.0) {
        sf = 1.0;
    }
    T.rnl = (0.0883289 + (1.0 - kB) * sf) * (kA + 1.95974 * tc); // :203
    // rw = (1 - alb) * xrw, :220-224
    T.xrw = (((sw_in == 0.0) || (hs == 0.0)) ? tau : tau_raw) * d.dr * kGsc;
    T.ru = ru;
    T.rv = rv;
    T.ruv = ruv;
    T.hs = hs;
    T.sin_hs = sin_hs;

    // ----- EVAP
    // Reciprocals are the dearest ordinary operation of the day step (a quarter-rate seed and two
    // Newton steps, ~9 issue slots), so denominators that are known together share one: the
    // reciprocal of their product, split by multiplications (each quotient 2-3 ulp).
    // tc + 237.3 (sat_slope :299-301), tc + 239.24 and tc + 273.15 (enthalpy_vap :313-315 and its
    // reciprocal in the psychrometric constant :486) are all > 0 for any air temperature
    const double qa = tc + 237.3, u = tc + 273.15, qb = u - 33.91;
    const double r_t = frcp(qa * qb * u);
    const double inv_q = r_t * (qb * u);
    const double s = fexp((tc * 17.269) * inv_q) * (inv_q * inv_q) * ((17.269) * (237.3) * (610.78));
    const double qlv = u * (r_t * (qa * u));     // u/(u - 33.91)
    const double lv = qlv * qlv * 1.91846e6;
    const double qil = qb * (r_t * (qa * qb));   // (u - 33.91)/u
    const double pw = pre.pw;                                         // :113
    const double g = (kMa * specific_heat(tc) * c.patm) * ((qil * qil) * (1.0 / (kMv * 1.91846e6))); // psychro :486
    const double lvpw = lv * pw;
    // econ = s/(lv*pw*(s+g)) (:117), its reciprocal and s/(lv*pw*(s+0.24g)) (:129)
    const double sg1 = s + g, sg2 = s + 0.24 * g;
    const double d1 = lvpw * sg1, d2 = d1 * sg2;
    const double r_e = frcp(d2 * s);
    const double inv_e = r_e * s;
    T.econ = s * sg2 * inv_e;
    T.inv_econ = d1 * (r_e * d2);                                     // lv*pw*(s+g)/s
    T.eetc = (1.0e3) * (s * sg1 * inv_e);                             // :129
    T.s = s;
    T.g = g;
    T.pw = pw;
    T.visc = day_viscosity(c, pre, tc);                               // :100-105, :120 (see visc_terms)
    T.warm = (tc >= 3.0);
}

SPL_DEV void evap_update(const CellConst& c, const DayTerms& T, const ViscTerms& V,
                                            double sw, double snow, double nd, Etr& o)
{
    const double ru = T.ru, rv = T.rv, rnl = T.rnl, s = T.s, g = T.g, econ = T.econ;
    const double sw_s = sw > 1.0 ? 1.0 : sw; // :92
    const double alb_v = kAlbSw - 0.17 * sw_s; // :215
    double alb = alb_v;                        // snow-free: sfc = 0 => alb = alb_v exactly
    if (snow > 0.0 || isnan(snow)) { // :208-219
        const double max_alb_snw = (1.0 - 0.443700) + (0.443700 * fexp(-0.895189 * nd));
        const double sfc = snow * frcp(140.0 + snow);
        alb = alb_v * (1.0 - sfc) + sfc * max_alb_snw;
    }
    const double rw = (1.0 - alb) * T.xrw; // :220-224

    // (1/(rw*rv) keeps its own reciprocal: the cross-over angles hn and hi below are arc cosines of
    // differences formed with it, near 1 for a low sun, where an ulp of the cosine is 1e4 ulp of
    // the angle)
    const double inv_rwrv = frcp(rw * rv);
    const double xn = (rnl - rw * ru) * inv_rwrv; // :231
    const double rx = (3.6e6) * econ;                                 // :145
    const double pet_max = rx * ((rw * (ru + rv)) - rnl);             // :148

    // :157-166  EF = 1/(g/(sw*s) + 1) = sw*s/(g + sw*s)
    const double sws = sw * s;
    double sup = pet_max * (sws * frcp(g + sws)); // g > 0
    if (sup < 0.0 || isnan(sup)) {
        sup = 0.0;
    }
    const double cos_hi = (sup * (T.inv_econ * (1.0 / 3.6e6)) + rnl) * inv_rwrv - T.ruv; // :208
    // xn >= 1: net radiation negative all day (hn = 0); cos_hi >= 1: supply exceeds demand (hi = 0)
    double hn, sin_hn, hi, sin_hi;
    hour_angle2(xn, cos_hi, hn, sin_hn, hi, sin_hi);
    const double rn_d = (hn * (rw * ru - rnl) + rw * rv * sin_hn) * (86400.0 / kPI); // :245-246
    double rnn_d = rw * rv * (T.sin_hs - sin_hn);                                     // :252-255
    rnn_d += rw * ru * (T.hs - hn);
    rnn_d -= rnl * (kPI - hn);
    rnn_d *= (86400.0 / kPI);

    const double cn = (1.0e3) * econ * fabs(rnn_d) * 0.1;             // :124
    const double eet_d = T.eetc * rn_d;                               // :129

    double snowmelt = 0.0, sublimation = 0.0, melt_enrg = 0.0; // :233-249
    if (T.warm) {
        const double m = (rn_d * V.ipk) * 1000.0;
        snowmelt = SPL_MIN(snow, m);
        melt_enrg = (snowmelt * 1.0e-3) * T.pw * kKfus;
    }
    {
        const double AE = rn_d - melt_enrg;
        const double sub_c = (AE * econ) * 1000.0;
        sublimation = SPL_MIN(snowmelt, sub_c);
        melt_enrg += ((sublimation * 1.0e-3) * T.inv_econ);
    }

    double aet_d = sup * hi; // :254-263
    aet_d += rx * rw * rv * (sin_hn - sin_hi);
    aet_d += (rx * rw * ru - rx * rnl) * (hn - hi);
    aet_d *= (24.0 / kPI);
    aet_d -= (melt_enrg * econ * 1000.0);
    if (aet_d < 0.0) {
        aet_d = 0.0;
    }

    o.cond = cn;
    o.pet = eet_d; // pet_d = eet_d, :140
    o.aet = aet_d;
    o.snowmelt = snowmelt;
    o.sublimation = sublimation;
    o.pw = T.pw;
    o.rn_d = rn_d;
    o.visc = T.visc;
    o.econ = econ;
    o.pet_max = pet_max;
}

// SOLAR + EVAP in one go (test hook k_evap_batch: src/EVAP.cpp:82-264 row by row)
SPL_DEV void evap_day(const CellConst& c, const DayConst& d, const DayPre& pre,
                                         double sw, double sw_in, double tc, double snow, double nd,
                                         Etr& o)
{
    DayTerms T;
    day_terms(c, d, pre, sw_in, tc, T);
    terms_fence(T);
    evap_update(c, T, visc_terms(c, T), sw, snow, nd, o);
}

// ---- SPLASH helpers ---------------------------------------------------------------------------
// a / b correctly rounded, given y = RN(1/b) (an IEEE division done once per cell): q = a*y is
// within an ulp, its exact residual r = a - b*q (one FMA) corrects it, and RN(q + r*y) is the
// correctly rounded quotient (Markstein's division step).  Three issue slots.
SPL_DEV double div_cr(double a, double b, double y)
{
    const double q = a * y;
    const double r = fma(-b, q, a);
    return fma(r, y, q);
}

// Theta = (theta - theta_r)/(theta_s - theta_r) for a column holding w mm, with the reference's own
// operations and roundings: theta = w/(depth*1000), then one subtraction and one division
// (src/SPLASH.cpp:1406-1408).  Bit-identical to the reference for a given w, and it has to be:
// Theta is raised to 1/lambda, and the synthetic and real soils reach lambda ~ 1e-4, so an ulp of
// Theta is 1e-12 of the transmittance -- enough, in such a cell, to move the next day's bucket by
// an ulp and start a visibly different trajectory a few hundred days later (the bucket is a
// steep map there: d(bflow)/d(wn) ~ bflow/(lambda * column range)).  A reciprocal-multiply Theta
// agrees to 2-3 ulp only.  As in the reference, Theta is EXACTLY 0 at w == RES and EXACTLY 1 at
// w == SAT (theta_r and theta_s are the same quotients), where its code is discontinuous
// (Theta = 0 puts the water table at the column base, Theta = 1 at the surface).
SPL_DEV double rel_saturation(const CellConst& c, double w)
{
    const double th = div_cr(w, c.d1000, c.inv_d1000);
    return div_cr(th - c.theta_r, c.dtheta, c.inv_dtheta);
}

// src/SPLASH.cpp:1921-1961 with z = 10 cm (call at :1265): Brooks-Corey water content of the
// top 10 cm.  With L = log(Theta):  psi = bpr*Theta^(-1/lambda),  (psi+z)/bpr =
// Theta^(-1/lambda) + z/bpr,  theta = dtheta*((psi+z)/bpr)^(-lambda) + theta_r.
SPL_DEV double moist_surf(const CellConst& c, double wn)
{
    const double L = flog(rel_saturation(c, wn));
    // the potential itself is formed (not just its ratio to bpr) so that it overflows to -inf
    // where the reference's bpr/pow(Theta, 1/lambda) does (1/lambda reaches 1e4: Theta^(1/lambda)
    // underflows for quite ordinary Theta), which sends theta to theta_r there
    const double psi = c.bpr * fexp(-L * c.inv_lambda);
    const double head_ratio = (psi + 10.0) * c.inv_bpr;
    const double theta_bc0 = c.dtheta * fexp(-c.lambda * flog(head_ratio)) + c.theta_r;
    // (if / else-if chains of this file are written as nested conditional expressions over finished values: the
    //  three-way merge of an else-if is not if-converted by the compiler, which then emits execution-mask branches --
    //  s_and_saveexec, s_cbranch_execz, s_or -- around one v_cndmask's worth of work, in every kernel)
    return (wn < c.RES) ? c.theta_r                       // theta_mean < theta_r
                        : (isnan(theta_bc0) ? c.theta_s : theta_bc0);
}

// src/SPLASH.cpp:1963-2023 with tdur = 6 h (call at :1268).  The surface moisture (moist_surf,
// :1265) only enters when the rain rate exceeds the conductivity, so it is evaluated inside that
// branch (two logs and two exps saved on every other cell-day).
SPL_DEV double inf_GA(const CellConst& c, double wn, double Ksat, double P)
{
    const double tdur = 6.0;
    const double r = P * (1.0 / 6.0); // P/tdur
    double I = P;
#if defined(SPL_ABLATE) && SPL_ABLATE == 11
    if (false) {
#else
    if (!(r <= Ksat)) {
#endif
        SPL_SITE();
        const double theta_i = moist_surf(c, wn);
        const double delta_theta = (c.theta_s - theta_i);
        if (delta_theta <= 0.0) {
            I = Ksat * tdur;
        } else {
            const double hd = c.h_f * delta_theta;
            double tp = (Ksat * delta_theta * -1.0 * c.h_f) * frcp(r * (r - Ksat)); // r > Ksat > 0
            if (tp <= 0.0 || isnan(tp)) {
                tp = 0.01;
            }
            const double tp_s = tp * c.inv_cos2;
            I = r * tp_s + (Ksat * (tdur - tp_s) - (hd * flog(1 - (r * tp_s * frcp(hd)))));
        }
    }
    if (I > P) {
        I = P;
    }
    return I;
}

// Unsaturated transmittance of the column at water content sm, scaled to flux density [mm/day]
// and guarded (src/SPLASH.cpp:1406-1431 and 1511-1536), and the saturated cross-section.
//   T_uns = C * (A - B),  A = (bub/psi_m)^e,  B = (bub/(psi_m + wtd*1000))^e,  e = 3*lambda+1
// With L = log(Theta): Theta^(1/lambda) gives psi_m and A = Theta^(e/lambda): one log, two exps.
// B = A * (1+u)^(-e) with u = wtd*1000/psi_m, so A - B = -A * expm1(-e*log1p(u)):
//   * water table inside the column (wtd unclamped): psi_m + wtd*1000 == bub to rounding, B = 1
//     (the reference's base bub/(psi_m+wtd*1000) is 1 +- 1 ulp, i.e. B = 1 +- 2e-16);
//   * clamped (dry column, wtd = depth, or the 0.01 guard): the expm1/log1p form keeps full
//     relative accuracy of the DIFFERENCE however close the two powers are.  That matters
//     although T_uns is then ~1e-30 mm/day: src/SPLASH.cpp:1472 divides by it (Q), so its
//     relative accuracy -- not its size -- sets the drainage time.  (Past |psi_m| ~ 1e11*depth
//     the reference's own difference is rounding noise of fl(psi_m + depth*1000): DESIGN.md
//     "Conditioning"; the parity tests set such cell-days aside.)
SPL_DEV double transmittance_uns(const CellConst& c, double Ksat_visc,
                                                    double sm, double& Acs_out)
{
#if defined(SPL_ABLATE) && SPL_ABLATE == 3
    Acs_out = (c.depth - 0.5 * c.depth) * c.so_cellout;
    return Ksat_visc * 1e-3 * sm;
#endif
    const double Theta = rel_saturation(c, sm);
    const double L = flog<kLogFinite>(Theta); // sm <= SAT: Theta <= 1
    const double P = fexp(-L * c.inv_lambda); // Theta^(-1/lambda) in [1, +inf]
    const double psi_m = c.bub * P;           // bub / Theta^(1/lambda)
    const double wtd0 = ((c.bub - psi_m) * 1.0e-3);
    const bool wtd_low = (wtd0 < 0.0 || isnan(wtd0));
    const bool wtd_high = !wtd_low && (wtd0 > c.depth);
    const bool clamped = wtd_low || wtd_high;
    const double wtd = wtd_low ? 0.01 : (wtd_high ? c.depth : wtd0);
    Acs_out = (c.depth - wtd) * c.so_cellout;
    const double w1000 = wtd * 1000.0;
    // A = Theta^((3*lambda+1)/lambda) = Theta^3 * Theta^(1/lambda) = Theta^3 / P: a reciprocal instead
    // of a second exponential (P = +inf for Theta = 0 gives A = 0, like the reference's pow(-0, e)).
    // (Left as a conditional expression around the reciprocal, execution-mask branch and all: the compiler folds the
    //  subtraction below into it and contracts -- the product's pinned bits are those of fma(Theta^3, 1/P, -1) --,
    //  and a reciprocal pinned before a select costs k_run_all's day loop a spilled register: measured, round 6.)
    const double A = (P < 1.0e300) ? (Theta * Theta * Theta) * frcp(P) : 0.0;
    double diff; // A - B
    if (!clamped) {
        diff = A - 1.0; // bub/(psi_m + wtd*1000) is 1 +- 1 ulp: B = 1 to 1e-16
    } else {
        // u = wtd*1000/psi_m, with the increment taken as the reference forms it: fl(psi_m +
        // wtd*1000) - psi_m.  For |psi_m| >~ 2^53*wtd*1000 the sum rounds back to psi_m, the
        // reference's two powers are then bit-identical, T_uns is exactly 0 and the drainage time
        // of src/SPLASH.cpp:1472 is +inf (sticky): reproduced exactly.  Just below that the
        // increment is a small multiple of ulp(psi_m) -- the same quantisation, though not the same
        // bits, as the reference's.
        const double xs = psi_m + w1000;
        const double u = (xs - psi_m) * frcp(psi_m); // in (-1, 0] for a dry column (NaN if psi_m = -inf:
                                                      // T_uns then falls to the guard below, = 0)
        double l1p, em1;
        if (fabs(u) < 1.0e-5) l1p = u * (1.0 - u * (0.5 - u * (1.0 / 3.0)));
        else l1p = flog<kLogPositive>(1.0 + u); // 1 + u in (0, 2] or NaN
        const double z = -c.e3l1 * l1p;
        if (fabs(z) < 1.0e-5) em1 = z * (1.0 + z * (0.5 + z * (1.0 / 6.0)));
        else em1 = fexp(z) - 1.0;
        diff = -A * em1;
    }
    double T_uns = (Ksat_visc * c.tu_all) * diff;
    if (T_uns < 0.0 || isnan(T_uns)) {
        T_uns = 0.0;
    }
    return T_uns;
}

// ---- the day step: src/SPLASH.cpp:920-1588 (== quick_run :150-918) ----------------------------
// day_update(): the water balance given the day's state-independent terms (computed by day_terms()
// or read back from the spin-up's cache), what follows from the day's viscosity (visc_terms()) and
// the day's rain and snowfall.
SPL_DEV void day_update(const CellConst& c, const DayTerms& K, const ViscTerms& V,
                                           double pn, double snowfall, State& st, DayOut& o)
{
    const double wn = st.wn;
    double snow = st.snow, qin = st.qin, td = st.td, nd = st.nd;

    // :1042-1048
    const double sw0 = (wn - c.RES) * c.inv_wr;
    const double sw = (sw0 < 0.0 || isnan(sw0)) ? 0.0 : ((sw0 > 1.0) ? 1.0 : sw0);
    // :1225-1231
    if (snowfall > 0.0) {
        nd = 0.0;
    } else {
        nd += 1.0;
    }
    snow += snowfall;

    Etr ev;
    evap_update(c, K, V, sw, snow, nd, ev);  // :1235
    snow -= ev.snowmelt;                          // :1240
    const double snowmelt = ev.snowmelt - ev.sublimation; // :1243

    // :1260-1272
    const double Ksat_visc = V.kv;
    const double inflow = pn + ev.cond + snowmelt;
    const double infi = inf_GA(c, wn, Ksat_visc, inflow);
    const double ro_h_arg = inflow - infi;
    const double ro_h = SPL_MAX(ro_h_arg, 0.0);
    double R = infi - ev.aet;

    // :1275-1284
    const double hyd_grad_z = (infi * V.ikv24) - 1.0;
    const double hyd_grad_out = fsqrt_nn(hyd_grad_z * hyd_grad_z + c.tan2_slop);

    // :1291-1326 (cell-static part folded into kb_coef); Kb and log(Kb) keep the reference's
    // exp-then-log rounding.  Kb is only consumed under the two conditions below.
    const double lkb_arg = Ksat_visc * c.kb_coef;

    // :1365-1372
    double q_in_o = 0.0;
    double Kb = 0.0;
    bool have_kb = false;
    if (!((td <= 0.0) || (qin <= 0.0))) {
        Kb = fexp(lkb_arg);
        have_kb = true;
        q_in_o = qin * Kb;
    }
    // :1377-1397
    const double sm0 = wn + q_in_o + R;
    const bool over0 = sm0 > c.SAT;
    const double ro_d0 = sm0 - c.SAT;
    double ro_d = over0 ? ro_d0 : 0.0;
    {
        const double R_less = R - ro_d0;
        R = (over0 && R > 0) ? R_less : R;
    }
    double sm = over0 ? c.SAT : ((sm0 < c.RES) ? c.RES : sm0);

    // :1406-1457 transmittance after recharge
    double Acs_out;
    SPL_SITE();
    const double T_uns1 = transmittance_uns(c, Ksat_visc, sm, Acs_out);
    const double sm1 = sm;
    double T_uns = T_uns1, T_sat;
#if defined(SPL_ABLATE) && SPL_ABLATE == 12
    if (false) {
#else
    if (c.deep) { // depth >= 2.0: vertical drainage corrections :1433-1450
#endif
        SPL_SITE();
        // :984-990 clamp of the start-of-day volumetric content, :1267 theta_m
        // (both quotients with the reference's rounding: the exponent 3 + 2/lambda reaches 1e4, so
        // an ulp of the base is 1e-12 of Kunsat -- see rel_saturation)
        const double theta_i0 = div_cr(wn, c.d1000, c.inv_d1000);
        // (tr + 0.001 is formed here: hoisted out of the day loop the sum would be one more live register,
        // i.e. a spill to scratch memory, whose reload waits on the vector-memory counter)
        double tr = c.theta_r;
        SPL_FIN(tr);
        const double theta_i = (theta_i0 >= c.theta_s) ? c.theta_s - 0.001
                                                       : ((theta_i0 <= c.theta_r) ? tr + 0.001 : theta_i0);
        const double theta_m = SPL_MAX(c.wm_th, theta_i);
        const double Kunsat = Ksat_visc * fpow(div_cr(theta_m, c.theta_s, c.inv_theta_s), c.kun_exp); // :1275
        T_uns += Kunsat * 24.0;
        T_sat = Ksat_visc * 24.0 * ((Acs_out + c.Ai) * c.inv_Ai);
    } else {
        T_sat = Ksat_visc * 24.0 * (Acs_out * c.inv_Ai);
    }
    const double T = (T_sat + T_uns) * hyd_grad_out;

    // :1465-1483
    double t_drain = 0.0;
    double q_in_f = 0.0;
    td -= 1.0;
    if ((R > 0.0) && (sm > c.Wmax)) {
        SPL_SITE();
        // log(Kb) with Kb = exp(lkb_arg) (:1326, :1470): the round trip returns lkb_arg PLUS the rounding of Kb, a
        // relative rho = (fl(e^x) - e^x)/e^x, |rho| <= 1.1e-16 -- relative to log(Kb) that is 1e-16/|x|, 1e-11 for the
        // recession constants of slopes.  Rounds 1-5 dropped it above |x| = 1e-7 ("< 1e-9"): a spin-up on an expanding
        // orbit amplified that 4.6e-13 of the lateral inflow to 2e-5 .. 9e-5 in one cell per million (round 6: cells
        // (2057, 1855), (2192, 523), tests/test_conditioning.py); with the plain round trip the same cells end 2e-8 ..
        // 3e-7 from the reference -- at 1.8 % of k_run_all.  So the rounding is REPRODUCED instead, for |x| < 2^-10
        // (three cells in four of the benchmark grid: the median |x| is 2e-4, the stray cells' 6e-5 .. 1e-4):
        // u = expm1(x) by its series to 1e-21, Kb = fl(1 + u) IS the correctly rounded exponential (glibc's is within
        // 0.51 ulp of it), Kb - 1 is exact, and log(Kb) = x + log1p(rho) = x + ((Kb - 1) - u)/Kb to 1e-16 of itself; ten
        // instructions instead of an exp and a log.  Kb = 1 exactly (x = 0: flat terrain, or |x| < 1e-16) gives
        // log(Kb) = 0 exactly, the sticky-NaN drainage time of SURVEY trap 3, as the round trip does.  From 2^-10 on the
        // rounding is < 1.2e-13 of log(Kb) and the argument itself stands, as before; an underflowing Kb and a NaN go
        // the reference's way.
        const double ax = fabs(lkb_arg);
        double lkb = lkb_arg;
        if (ax < 0.0009765625) {
            const double x = lkb_arg;
            double pk = fma(x, 1.0 / 120.0, 1.0 / 24.0);
            pk = fma(x, pk, 1.0 / 6.0);
            pk = fma(x, pk, 0.5);
            const double u = fma(x * x, pk, x);  // expm1(x) to 1e-21
            const double v = (1.0 + u) - 1.0;    // fl(exp(x)) - 1, exact
            lkb = (v == 0.0) ? 0.0 : fma(v - u, 1.0 - u, x); // x + ((Kb - 1) - expm1(x)) / Kb, 1/Kb = 1 - u to 1e-6
        } else if (!(ax <= 700.0)) {
            if (!have_kb) Kb = fexp(lkb_arg);
            lkb = flog(Kb);
        }
        const double Q = (T * c.Ai) * 1.0e-3; // :1456
        const double Qt = (Ksat_visc * c.qt_coef) * hyd_grad_out; // :1360
        // (R/Q stays an IEEE division: Q = 0 -- a transmittance of exactly zero -- must give the
        // reference's +inf; 1/lkb is NaN for lkb = 0 either way: the sticky NaN of flat terrain)
        t_drain = (-1.0 * flog(1.0 - (lkb * (c.Au * R / Q)))) * frcp(lkb);
        q_in_f = (Qt - c.Au * R * lkb) * c.inv_Ai;
    }
    if (q_in_f < 0.0 || isnan(q_in_f)) {
        q_in_f = 0.0;
    }
    const double td_arg = (td + t_drain) / 2;
    const double tdrain_out = SPL_MAX(td_arg, 0.0);

    // :1488-1506
    {
        const double sm_in = sm + (q_in_f);
        const bool over1 = sm_in > c.SAT;
        const double ro_more = ro_d + (sm_in - c.SAT);
        ro_d = over1 ? ro_more : ro_d;
        sm = over1 ? c.SAT : ((sm_in < c.RES) ? c.RES : sm_in);
    }
    const double ro = ro_d + ro_h;

    // :1511-1547 transmittance after upslope inflow (no vertical-drainage correction here);
    // identical to the first evaluation whenever the inflow left sm unchanged
    double Acs2 = Acs_out, T_uns2 = T_uns1;
#if defined(SPL_ABLATE) && SPL_ABLATE == 13
    if (false) {
#else
    if (sm != sm1) {
#endif
        SPL_SITE();
        T_uns2 = transmittance_uns(c, Ksat_visc, sm, Acs2);
    }
    const double T2 = (Ksat_visc * 24.0 * (Acs2 * c.inv_Ai) + T_uns2) * hyd_grad_out;

    // :1552-1567
    {
        const double sm_out = sm - (T2);
        sm = (sm_out > c.SAT) ? c.SAT : ((sm_out < c.RES) ? c.RES : sm_out);
    }
    // :1574
    const double mx = SPL_MAX(q_in_o, q_in_f);
    const double qin_nday = SPL_MAX(mx, 0.0);

    st.wn = sm;
    st.snow = snow;
    st.qin = qin_nday;
    st.td = tdrain_out;
    st.nd = nd;
    o.ro = ro;
    o.pet = ev.pet;
    o.aet = ev.aet;
    o.cond = ev.cond;
    o.bflow = T2;
    o.netr = ev.rn_d * 1.0e-6; // src/SPLASH.cpp:1907
}

// FENCE = true (the spin-up): the terms are finished doubles before the update uses them, exactly
// as when they are read back from the cache.  FENCE = false (run_all): the compiler may fuse and
// reorder across the seam (3 % faster: the 17 terms need not all be live at once).
template <bool FENCE = false>
SPL_DEV void day_step(const CellConst& c, const DayConst& d, const DayPre& pre,
                                         double sw_in, double tc, double pn, double snowfall,
                                         State& st, DayOut& o)
{
    DayTerms T;
    day_terms(c, d, pre, sw_in, tc, T);
    if (FENCE) terms_fence(T);
    day_update(c, T, visc_terms(c, T), pn, snowfall, st, o);
}

// ---- the two helper methods the module exports, with their general argument lists and written literally like the
// reference (src/SPLASH.cpp:1921-1961, 1963-2023; library pow/log/cos): splash_mod_moist_surf / splash_mod_inf_GA
SPL_DEV_NOINL double moist_surf_literal(double depth, double z, double bub_p, double wn, double SAT, double RES,
                                        double lambda)
{
    double theta_r = RES / (depth * 1000.0);
    double theta_s = SAT / (depth * 1000.0);
    double bubbling_pr = bub_p / 10;
    double theta_mean = (wn) / (depth * 1000.0);
    double water_pot_BC =
        bubbling_pr / pow((((theta_mean - theta_r) / (theta_s - theta_r))), (1 / lambda));
    double total_head_BC = water_pot_BC + z;
    double theta_BC =
        (theta_s - theta_r) * pow((total_head_BC / bubbling_pr), (-1 * lambda)) + theta_r;
    if (theta_mean < theta_r) {
        theta_BC = theta_r;
    } else if (isnan(theta_BC)) {
        theta_BC = theta_s;
    }
    return theta_BC;
}

SPL_DEV_NOINL double inf_GA_literal(double bub_press, double theta_i, double Ksat, double theta_s, double lambda,
                                    double P, double tdur, double slop)
{
    double r = P / tdur;
    double h_f = ((2 + 3 * lambda) / (1 + 3 * lambda)) * (bub_press / 2);
    double delta_head = h_f;
    double delta_theta = (theta_s - theta_i);
    double I = 0.0;
    if (r <= Ksat) {
        I = P;
    } else {
        if (delta_theta <= 0.0) {
            I = Ksat * tdur;
        } else {
            double tp = (Ksat * delta_theta * -1.0 * delta_head) / (r * (r - Ksat));
            if (tp <= 0.0 || isnan(tp)) tp = 0.01;
            double cs = cos(slop * kPir);
            double tp_s = tp / (cs * cs);
            I = r * tp_s + (Ksat * (tdur - tp_s) -
                            (delta_head * delta_theta *
                             log(1 - (r * tp_s / (delta_head * delta_theta)))));
        }
    }
    if (I > P) I = P;
    return I;
}

// ---- soil_hydro + soil_info for one cell: R/splash.point.R:232-416, 96-115, :131 (asp-180) -----------------------
// col[P_NPAR]: the cell's column of the cell-parameter array (k_soil_hydro on the device, the host backend on a CPU)
SPL_DEV_NOINL void soil_hydro_cell(double lat, double elev, double slop, double asp, double res, double sand,
                                   double clay, double OM, double fgravel, double bd, double depth_m,
                                   double au_area, double au_in, double au_out, double* col)
{
    double fsand = sand / 100, fclay = clay / 100, fOM = OM / 100; // :261-264
    fgravel = fgravel / 100;
    const double depth30 = 30; // :268
    double dp = 1 / ((fOM / 1.3) + ((1 - fOM) / 2.65));
    if (isnan(bd)) { // :276
        bd = (1.5 + (dp - 1.5 - 1.10 * (1 - fclay)) * (1 - exp(-0.022 * depth30))) /
             (1 + 6.27 * fOM);
    }
    if (bd < 0.81) bd = 0.81; // :280
    double sat = 1 - (bd / dp); // :285
    double sq_clay = sqrt(fclay); // fclay^0.5
    double fc = (sat / bd) * (0.4760944 + (0.9402962 - 0.4760944) * sq_clay) *
                exp(-1 * (0.05472678 * fsand - 0.01 * fOM) / (sat / bd)); // :287
    double wp_Ball = fc * (0.2018522 + (0.7809203 - 0.2018522) * sq_clay);    // :289
    double wp = -2.464e-05 * sand + 3.650e-03 * clay + 8.680e-03 * OM + 9.393e-03 * bd; // :294
    if (!isnan(wp) && wp >= fc) wp = wp_Ball;                                           // :295
    double coef_B = (log(1500.0) - log(33.0)) / (log(fc) - log(wp)); // :319
    double coef_A = exp(log(33.0) + coef_B * log(fc));               // :320
    double coeff_c = 1000.0 / (997 * 9.80665);                       // :326
    double theta_c = pow((coeff_c * coef_A / 2.0), (1 / (1 + coef_B))); // :327
    double theta_r = (0.0285 + 0.00336 * (clay)) * bd;                  // :340
    if (!isnan(theta_r) && theta_r > wp) theta_r = wp;                  // :341
    sat = sat * (1 - fgravel); // :351-353
    fc = fc * (1 - fgravel);
    wp = wp * (1 - fgravel);
    double coef_lambda = 1 / coef_B;
    double ksat = 857.48454 / (1 + exp(-2.70927 * fsand + 3.62264 * bd + 7.33398 * fclay +
                                       -8.11795 * (sat - fc) + 18.75552 * fOM +
                                       1.03319 * coef_lambda)); // :355-363
    double m33i = 0.278 * fsand + 0.034 * fclay + 0.022 * fOM - 0.018 * (fsand * fOM) -
                  0.027 * (fclay * fOM) - 0.584 * (fsand * fclay) + 0.078; // :369
    double m33 = m33i + (0.636 * m33i - 0.107);                            // :370
    double bub_init = -21.6 * fsand - 27.93 * fclay - 81.97 * m33 + 71.12 * (fsand * m33) +
                      8.29 * (fclay * m33) + 14.05 * (fsand * fclay) + 27.16; // :371
    double bubbling_p = bub_init + (0.02 * (bub_init * bub_init) - 0.113 * bub_init - 0.7); // :372
    bubbling_p = bubbling_p * -101.97162129779;                                             // :374
    if (!isnan(bubbling_p) && bubbling_p > 0) bubbling_p = coef_A * -101.97162129779;        // :379
    double RESv = theta_r * (1 - fgravel); // :410

    col[P_LAT] = lat;
    col[P_ELV] = elev;
    col[P_SLOP] = slop;
    col[P_ASP] = asp - 180; // R/splash.point.R:131
    col[P_SAT] = sat * depth_m * 1000; // :98-101
    col[P_WP] = wp * depth_m * 1000;
    col[P_FC] = fc * depth_m * 1000;
    col[P_KSAT] = ksat;
    col[P_LAMBDA] = 1 / coef_B; // :104
    col[P_DEPTH] = depth_m;
    col[P_BUB] = bubbling_p;
    col[P_RES] = RESv * depth_m * 1000;
    col[P_AU] = au_area;
    col[P_AI] = res * res; // resolution^2
    col[P_CELLIN] = au_in;
    col[P_CELLOUT] = au_out;
    col[P_WMAXR] = theta_c * depth_m * 1000; // :102
}

// ---- rain / snow split: R/splash.point.R:120-128, 521-578 -------------------------------------
// Exponent of the snowfall logistic (:576) in R's own evaluation order,
//   ((-0.4710405934 + 1.0473543991*tc) - elev*0.0004596581) - |lat|*0.0110592101,
// with every product and sum rounded separately (no FMA: R's build has none on x86-64).
// e1 = elev*0.0004596581 and e2 = |lat|*0.0110592101 are the cell-static products.
SPL_DEV double snow_logit(double tc, double e1, double e2)
{
#pragma clang fp contract(off)
    return ((-0.4710405934 + 1.0473543991 * tc) - e1) - e2;
}
// p_snow = 1/(1+exp(x)) >= 0.5 as the reference evaluates it in doubles: exp(x) rounds to 1 for
// x <= 2^-53 and to 1+2^-52 up to 1.5*2^-52; 1+(1+2^-52) ties to 2 and 1/2 = 0.5 passes, the next
// double up gives 0.5-2^-54.  So the test is x <= 1.5*2^-52 (not x <= 0): identical to the
// logistic for a correctly rounded exp (glibc's is, at these arguments), without evaluating it.
// A NaN exponent (missing tc) is neither: R's ifelse() then yields NA (see rain_snow_split).
constexpr double kSnowLogitMax = 0x1.8p-52;

SPL_DEV double snowfall_prob(double tc, double e1, double e2)
{
    return 1 / (1 + fexp(snow_logit(tc, e1, e2)));
}

// Tt is handed over as a callable and fetched only on possible-snow days (k_run_all keeps it in LDS).
template <class TtSource>
SPL_DEV void rain_snow_split(const DayConst& d, double tc, double pn,
                                                const TtSource& Tt_of_cell, double e1, double e2,
                                                double& rain, double& snowfall)
{
    const double x = snow_logit(tc, e1, e2);
    double f_rain = 1.0;
    if (x <= kSnowLogitMax) {
        const double Tt = Tt_of_cell();
        double Ttm = Tt + (Tt * d.ttm_s);
        double q = ((tc - Ttm) * d.inv_trm14); // the divisor is a per-day constant: reciprocal from the host table
        double q2 = q * q, q3 = q2 * q;
        double fr;
        if (tc <= Ttm) {
            fr = 5 * q3 + 6.76 * q2 + 3.19 * q + 0.5;
        } else {
            fr = 5 * q3 - 6.76 * q2 + 3.19 * q + 0.5;
        }
        if (fr < 0) fr = 0; // frain[frain<0] <- 0 leaves NA (Tt = NA: a gap in tc) alone
        if (fr > 1) fr = 1;
        f_rain = fr;
    } else if (x != x) {
        f_rain = x; // ifelse(NA >= 0.5, ., .) is NA
    }
    snowfall = pn * (1 - f_rain);
    rain = pn * f_rain;
}

} // namespace splash
