// splash_math.hpp -- FP64 log / exp / pow / acos tuned for the SPLASH day step on gfx950.
//
// The day step needs ~10 pow() per cell-day in the reference's formulation; the general-purpose
// library pow costs ~230 VALU instructions on gfx950 because it carries a double-double
// logarithm to stay below 1 ulp for every (x, y).  The water-balance terms only need
// |y*log x| <~ 40 at 1e-14 relative, so the pows are rebuilt from one short log (table driven:
// 128 x (1/c, log c), series to r^8, <~ 2 ulp) and one short exp (Cody-Waite reduction by ln2 and
// a degree-13 series, < 1 ulp), and logs are shared between powers of the same base.
//
// The file is plain C++ (frexp/ldexp/rint/fma/sqrt map to single VALU instructions on the
// device), so tests/ compile it with g++ too and check it against libm on the host.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SPL_HD __host__ __device__ __forceinline__
#else
#define SPL_HD inline
#endif

namespace splash {

// p*r + C with the constant C held in a scalar register pair.  For a Horner step with a constant
// that is not an inline literal hipcc emits v_mov_b32 x2 + v_fmac_f64 (the VOP2 form wants C in
// the destination VGPR): three VALU issues instead of one.  Written as the VOP3 form with C as
// the single scalar operand, the constant is materialised on the scalar pipe (s_mov_b32) and the
// step is one v_fma_f64.  Polynomials are a large share of the day step, so this matters.
SPL_HD double hfma(double p, double r, double C)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(r), "s"(C));
    return d;
#else
    return fma(p, r, C);
#endif
}

// Whole Horner chains in one asm statement (hipcc pads every separate asm statement whose result
// feeds the next with an s_nop; inside one statement the dependent v_fma_f64 issue back to back,
// exactly as compiler-generated chains do).  p = (((c0*r + c1)*r + c2)*r + ...).
#if defined(__HIP_DEVICE_COMPILE__)
#define SPL_H1 "v_fma_f64 %0, %1, %2, %3\n\t"
#define SPL_HN(k) "v_fma_f64 %0, %0, %2, %" #k "\n\t"
SPL_HD double horner3(double c0, double r, double c1, double c2, double c3)
{
    double p;
    asm(SPL_H1 SPL_HN(4) SPL_HN(5) : "=&v"(p) : "v"(c0), "v"(r), "s"(c1), "s"(c2), "s"(c3));
    return p;
}
SPL_HD double horner2(double c0, double r, double c1, double c2)
{
    double p;
    asm(SPL_H1 SPL_HN(4) : "=&v"(p) : "v"(c0), "v"(r), "s"(c1), "s"(c2));
    return p;
}
SPL_HD double horner4(double c0, double r, double c1, double c2, double c3, double c4)
{
    double p;
    asm(SPL_H1 SPL_HN(4) SPL_HN(5) SPL_HN(6)
        : "=&v"(p) : "v"(c0), "v"(r), "s"(c1), "s"(c2), "s"(c3), "s"(c4));
    return p;
}
SPL_HD double horner5(double c0, double r, double c1, double c2, double c3, double c4, double c5)
{
    double p;
    asm(SPL_H1 SPL_HN(4) SPL_HN(5) SPL_HN(6) SPL_HN(7)
        : "=&v"(p) : "v"(c0), "v"(r), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5));
    return p;
}
#undef SPL_H1
#undef SPL_HN
#else
SPL_HD double horner2(double c0, double r, double c1, double c2) { return fma(fma(c0, r, c1), r, c2); }
SPL_HD double horner3(double c0, double r, double c1, double c2, double c3)
{
    return fma(fma(fma(c0, r, c1), r, c2), r, c3);
}
SPL_HD double horner4(double c0, double r, double c1, double c2, double c3, double c4)
{
    return fma(fma(fma(fma(c0, r, c1), r, c2), r, c3), r, c4);
}
SPL_HD double horner5(double c0, double r, double c1, double c2, double c3, double c4, double c5)
{
    return fma(fma(fma(fma(fma(c0, r, c1), r, c2), r, c3), r, c4), r, c5);
}
#endif

// Reciprocal for a finite, nonzero, normal-range argument: hardware seed (v_rcp_f64, ~2^-24
// relative) + two Newton steps -> ~1 ulp.  Skips the range scaling and special-case fix-up of a
// full IEEE division (6 instead of ~11 VALU instructions); only for denominators that are known
// to be ordinary numbers.  On the host build it is a plain division.
SPL_HD double frcp(double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double x = __builtin_amdgcn_rcp(b);
    x = fma(fma(-b, x, 1.0), x, x);
    x = fma(fma(-b, x, 1.0), x, x);
    return x;
#else
    return 1.0 / b;
#endif
}

// Square root to ~1 ulp for x >= 0 in the normal range (0 and NaN handled): hardware reciprocal
// square root seed (v_rsq_f64) + two Heron steps, without the range scaling of the IEEE routine.
// NOT correctly rounded: never used where the reference's result is subsequently rounded to
// float (the viscosity routine keeps the library sqrt).
SPL_HD double fsqrt(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rsq(x);
    double y = x * r;
    const double h = 0.5 * r;
    y = fma(fma(-y, y, x), h, y);
    y = fma(fma(-y, y, x), h, y);
    return (x == 0.0) ? x : y;
#else
    return sqrt(x);
#endif
}

// The same for an argument that is known to be >= 0 (or NaN): the seed is taken of max(x, 1e-300),
// so x = 0 gives y = 0 * 1e150 = 0 and stays 0 through the Heron steps -- one instruction instead
// of the compare and two selects that patch rsq(0) = inf above.  NaN stays NaN (y = x * r).
SPL_HD double fsqrt_nn(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_amdgcn_rsq(fmax(x, 1.0e-300));
    double y = x * r;
    const double h = 0.5 * r;
    y = fma(fma(-y, y, x), h, y);
    y = fma(fma(-y, y, x), h, y);
    return y;
#else
    return sqrt(x);
#endif
}

// natural logarithm, table driven (Tang / glibc scheme), <~ 2 ulp; log(0) = -inf, log(x<0) = NaN.
//   x = 2^k * z, z in [0.6875, 1.375);  log x = k*ln2 + log c_i + log1p(z/c_i - 1)
// with (1/c_i, log c_i) from a 128-entry table indexed by the top 7 mantissa bits (offset so that
// the interval is centred on 1: no cancellation between k*ln2 and the rest), r = z/c_i - 1 formed
// exactly enough by one FMA (|r| < 0.0079) and log1p(r) by its series to r^8.  The two intervals
// that touch 1 have c = 1: log(1) = 0 exactly (the water balance relies on it: Kb = 1 on flat
// terrain) and full relative accuracy around 1 (near saturation log(Theta) -> 0).
// Against the division-based fdlibm form this saves the reciprocal (a quarter-rate seed and two
// Newton steps) and half of the polynomial: ~20 of ~56 issue slots per call, and the day step
// spends a sixth of its time in log().  On the device the table is read from LDS (filled by
// math_tables_init(), which every kernel that reaches flog calls first).
// Branch-free: special values are patched by selects.
#include "splash_log_table.hpp"
#if defined(__HIPCC__)
__device__ const double kLogTabDev[256] = {SPL_LOG_TABLE_VALUES};
__shared__ __attribute__((aligned(16))) double s_log_tab[256];
__device__ __forceinline__ void math_tables_init()
{
    for (unsigned i = threadIdx.x; i < 256; i += blockDim.x) s_log_tab[i] = kLogTabDev[i];
}
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
static const double kLogTabHost[256] = {SPL_LOG_TABLE_VALUES};
#endif

// kLogFull: every special value patched (log(+inf) = +inf, log(0) = -inf, log(x<0) = log(NaN) = NaN).
// kLogFinite: the caller's argument is never +inf (a relative saturation, <= 1): that patch is left out.
// kLogPositive: the argument is a positive finite number or NaN (1 + u of the transmittance): only
//   NaN is patched, in the high word.
enum LogDomain { kLogFull = 0, kLogFinite = 1, kLogPositive = 2 };
template <int DOMAIN = kLogFull>
SPL_HD double flog(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    int e;
    const double m = frexp(x, &e); // [0.5, 1); handles subnormal x
    const unsigned long long bm = __builtin_bit_cast(unsigned long long, m);
    const unsigned hi = (unsigned)(bm >> 32);
    const int tmp = (int)(hi - 0x3FE60000u);
    const int i = (tmp >> 13) & 127;
    const int kk = tmp >> 20; // -1 for m < 0.6875 (then z = 2m), else 0
    const unsigned zhi = hi - ((unsigned)tmp & 0xFFF00000u);
    const double z = __builtin_bit_cast(double, ((unsigned long long)zhi << 32) | (bm & 0xFFFFFFFFull));
#if defined(__HIP_DEVICE_COMPILE__)
    // one 16-byte LDS read for the pair: logc is not needed until the end, but a second read would
    // cost a second round trip to LDS (the wait is for ALL outstanding LDS reads)
    typedef double vec2d __attribute__((ext_vector_type(2)));
    const vec2d ent = reinterpret_cast<const vec2d*>(s_log_tab)[i];
    const double invc = ent.x, logc = ent.y;
#else
    const double invc = kLogTabHost[2 * i], logc = kLogTabHost[2 * i + 1];
#endif
    const double r = fma(z, invc, -1.0);
    const double kd = (double)(e + kk);
    const double r2 = r * r;
    double p = horner5(-1.0 / 8.0, r, 1.0 / 7.0, -1.0 / 6.0, 1.0 / 5.0, -1.0 / 4.0, 1.0 / 3.0);
    p = fma(p, r, -0.5); // log1p(r) = r + r^2 * p
    const double hi_ = fma(kd, ln2_hi, logc);
    const double lo_ = fma(kd, ln2_lo, r2 * p);
    double res = (hi_ + r) + lo_;
    if (DOMAIN == kLogPositive) {
        const unsigned long long rb = __builtin_bit_cast(unsigned long long, res);
        const unsigned rh = (x != x) ? 0x7FF80000u : (unsigned)(rb >> 32);
        return __builtin_bit_cast(double, ((unsigned long long)rh << 32) | (rb & 0xFFFFFFFFull));
    }
    // (The compiler sinks the last additions into the x > 0 arm of the patch below and emits an execution-mask
    //  branch around them at every call site.  Pinning `res` before the select removes those branches -- measured,
    //  round 6: within the run-to-run noise of the spin-up's tail -- but changes which multiply the compiler fuses
    //  downstream: the product's pinned bits moved.  Left alone.)
    if (DOMAIN == kLogFull) res = (x < HUGE_VAL) ? res : x; // +inf (NaN falls through to the next line)
    res = (x > 0.0) ? res : ((x == 0.0) ? -HUGE_VAL : __builtin_nan("")); // 0 -> -inf, <0/NaN -> NaN
    return res;
}

// exponential, < 1 ulp; exp(-inf) = 0, overflow -> +inf, NaN -> NaN.  Branch-free.
SPL_HD double fexp(double x)
{
#if defined(SPL_ABLATE) && SPL_ABLATE == 8
    return 1.0 + x * 1e-3;
#endif
    const double log2e = 1.44269504088896338700e+00;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double xc = fmin(fmax(x, -746.0), 710.0); // ldexp saturates to 0 / inf at the ends
    const double k = rint(xc * log2e);
    const double r = fma(-k, ln2_lo, fma(-k, ln2_hi, xc)); // |r| <= ln2/2 (+ tiny)
    // exp(r) by its series to degree 13: remainder (ln2/2)^14/14! = 4e-18
    double p = horner5(1.0 / 6227020800.0, r, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0,
                       1.0 / 362880.0, 1.0 / 40320.0);
    p = horner5(p, r, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double y = ldexp(p, (int)k);
#if defined(__HIP_DEVICE_COMPILE__)
    // NaN in, NaN out (the clamp above turned it into -746): patched in the high word alone --
    // a compare and ONE select instead of two
    const unsigned long long yb = __builtin_bit_cast(unsigned long long, y);
    const unsigned hi = (x != x) ? 0x7FF80000u : (unsigned)(yb >> 32);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (yb & 0xFFFFFFFFull));
#else
    return (x != x) ? x : y;
#endif
}

// x^y for the bases the water balance uses (x >= 0 or NaN-producing x < 0), via exp(y*log x);
// relative error ~ (1 + |y*log x|) ulp.  pow(0, y>0) = 0, pow(x<0, y) = NaN as in libm for
// non-integer y (the reference relies on that NaN, src/SPLASH.cpp:1951-1955).
SPL_HD double fpow(double x, double y) { return fexp(y * flog<>(x)); }

// arc cosine in radians, |x| <= 1 (callers guard the range), < 2.5 ulp.
// fdlibm e_acos.c scheme (rational approximation R(z) of (asin(t)-t)/t^3 on t^2 = z <= 0.25,
// reflection elsewhere), written branch-free: both ranges share one evaluation of R with
// z = x^2 (|x| <= 0.5) or z = (1-|x|)/2, so a wavefront whose lanes fall in different ranges
// does not run two copies of the polynomial.
SPL_HD double facos(double x)
{
#if defined(SPL_ABLATE) && SPL_ABLATE == 9
    return 1.5707963267948966 - x;
#endif
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double pi = 3.14159265358979311600e+00;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const double ax = fabs(x);
    const bool small = ax <= 0.5;
    const double z = small ? x * x : (1.0 - ax) * 0.5;
    const double p = z * horner5(pS5, z, pS4, pS3, pS2, pS1, pS0);
    const double q = fma(z, horner3(qS4, z, qS3, qS2, qS1), 1.0);
    const double r = p * frcp(q); // q in [0.3, 1]
    // |x| <= 0.5: pi/2 - (x + x*r)
    const double a_small = pio2_hi - (x - (pio2_lo - x * r));
    // |x| > 0.5: 2*asin(sqrt(z)), reflected for x < 0
    const double s = fsqrt_nn(small ? 0.25 : z);
    const double v = 2.0 * (s + r * s);
    const double a_large = (x < 0.0) ? pi - (v - 2.0 * pio2_lo) : v;
    return small ? a_small : a_large;
}

// Two arc cosines at once: the same evaluation as facos() for each, with the two rational
// approximations sharing ONE reciprocal, 1/(q1*q2) (q in [0.3, 1]) -- the net-radiation and the
// supply/demand cross-over angles of a day are needed together.
SPL_HD void facos2(double x1, double x2, double& a1, double& a2)
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double pi = 3.14159265358979311600e+00;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const double ax1 = fabs(x1), ax2 = fabs(x2);
    const bool small1 = ax1 <= 0.5, small2 = ax2 <= 0.5;
    const double z1 = small1 ? x1 * x1 : (1.0 - ax1) * 0.5;
    const double z2 = small2 ? x2 * x2 : (1.0 - ax2) * 0.5;
    const double p1 = z1 * horner5(pS5, z1, pS4, pS3, pS2, pS1, pS0);
    const double p2 = z2 * horner5(pS5, z2, pS4, pS3, pS2, pS1, pS0);
    const double q1 = fma(z1, horner3(qS4, z1, qS3, qS2, qS1), 1.0);
    const double q2 = fma(z2, horner3(qS4, z2, qS3, qS2, qS1), 1.0);
    const double rq = frcp(q1 * q2);
    const double r1 = p1 * (rq * q2), r2 = p2 * (rq * q1);
    const double s1 = fsqrt_nn(small1 ? 0.25 : z1), s2 = fsqrt_nn(small2 ? 0.25 : z2);
    const double as1 = pio2_hi - (x1 - (pio2_lo - x1 * r1)), as2 = pio2_hi - (x2 - (pio2_lo - x2 * r2));
    const double v1 = 2.0 * (s1 + r1 * s1), v2 = 2.0 * (s2 + r2 * s2);
    const double al1 = (x1 < 0.0) ? pi - (v1 - 2.0 * pio2_lo) : v1;
    const double al2 = (x2 < 0.0) ? pi - (v2 - 2.0 * pio2_lo) : v2;
    a1 = small1 ? as1 : al1;
    a2 = small2 ? as2 : al2;
}

// arc tangent, fdlibm s_atan.c scheme (argument reduction at 7/16, 11/16, 19/16, 39/16 with
// atan(0.5), atan(1), atan(1.5), atan(inf) in two parts; odd/even polynomial in the reduced
// argument), branch-free: the reduction's numerator, denominator and the two table parts are
// selected, one reciprocal serves every range.  < 1.5 ulp.  NaN -> NaN, +-inf -> +-pi/2.
SPL_HD double fatan(double x)
{
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    const double ax = fabs(x);
    // range  num        den        hi                          lo
    // <7/16  ax         1          0                           0
    // <11/16 2ax-1      2+ax       atan(0.5)
    // <19/16 ax-1       ax+1       atan(1)
    // <39/16 ax-1.5     1+1.5ax    atan(1.5)
    // else   -1         ax         atan(inf)
    const bool r0 = ax < 0.4375, r1 = ax < 0.6875, r2 = ax < 1.1875, r3 = ax < 2.4375;
    const double num = r0 ? ax : (r1 ? 2.0 * ax - 1.0 : (r2 ? ax - 1.0 : (r3 ? ax - 1.5 : -1.0)));
    const double den = r0 ? 1.0 : (r1 ? 2.0 + ax : (r2 ? ax + 1.0 : (r3 ? 1.0 + 1.5 * ax : ax)));
    const double hi = r0 ? 0.0 : (r1 ? 4.63647609000806093515e-01 : (r2 ? 7.85398163397448278999e-01
                                  : (r3 ? 9.82793723247329054082e-01 : 1.57079632679489655800e+00)));
    const double lo = r0 ? 0.0 : (r1 ? 2.26987774529616870924e-17 : (r2 ? 3.06161699786838301793e-17
                                  : (r3 ? 1.39033110312309984516e-17 : 6.12323399573676603587e-17)));
    // den >= 1 except in the last range, where it is ax >= 2.4375 (+inf gives t = -0): ordinary
    const double rd = frcp(den < 1.0e300 ? den : 1.0e300);
    double t = num * rd;
    t = fma(fma(-den, t, num), rd, t); // one correction step: the quotient to ~0.5 ulp
    t = (ax == HUGE_VAL) ? -0.0 : t;
    const double z = t * t, w = z * z;
    const double s1 = z * horner5(aT10, w, aT8, aT6, aT4, aT2, aT0);
    const double s2 = w * horner4(aT9, w, aT7, aT5, aT3, aT1);
    const double small = t - t * (s1 + s2);
    const double big = hi - ((t * (s1 + s2) - lo) - t);
    const double r = r0 ? small : big;
    return (x != x) ? x : ((x < 0.0) ? -r : r);
}

// atan2(y, x) with fdlibm's quadrant rules (e_atan2.c) for finite arguments that are not both zero
// (the caller's case); y/x is an IEEE division, so x = 0 gives atan(inf) = pi/2.
SPL_HD double fatan2(double y, double x)
{
    const double pi = 3.1415926535897931160e+00, pi_lo = 1.2246467991473531772e-16;
    const double z = fatan(fabs(y / x));
    const double q = (x < 0.0) ? pi - (z - pi_lo) : z; // quadrants II/III mirror about pi/2
    return (y < 0.0) ? -q : q;
}

} // namespace splash
