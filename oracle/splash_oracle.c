/*
 * splash_oracle.c -- CPU restatement of the rsplash hot path.  TEST INFRASTRUCTURE ONLY
 * (see splash_oracle.h for who may use it and for the pin status of each part).
 *
 * Written from the reference's equations, following its evaluation order statement by
 * statement (no hoisting, no algebraic merging) so that, built with the pinned flags
 * (gcc -std=c11 -O2, no -march, no fast-math => no FMA contraction on x86-64), it
 * reproduces the reference C++ built with R's default flags (g++ -O2).
 *
 * Citations are file:line relative to /root/reference/.
 */
#include "splash_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- evidence builds only (tools/conditioning_evidence.py; never the parity oracle) ----------
 * -DORC_PERTURB_LIBM wraps the libm calls of the FP64 path so that every result is moved by one
 * ulp up or down (or left alone) by a hash of its arguments: "another correct libm", which is what
 * any second implementation of the reference amounts to (the GPU's math library, glibc on another
 * CPU).  Results that every implementation returns exactly (x == 0, integer and 0.5 powers,
 * log(1), ...) are left alone, and so is the float viscosity routine together with the density
 * polynomial it rounds to float (ORC_EXACT_BEGIN/END): those are reproduced bit for bit by the
 * product and are not a source of disagreement. */
#ifdef ORC_PERTURB_LIBM
static inline double orc_pert(double r, double x, double y, unsigned salt)
{
    union { double d; uint64_t u; } a = {x}, b = {y};
    uint64_t z = a.u * 0x9E3779B97F4A7C15ULL + (b.u ^ ((uint64_t)salt << 32));
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
#ifndef ORC_PERTURB_ULPS
#define ORC_PERTURB_ULPS 1 /* amplitude; 4 models a library of a few ulp (e.g. a 2.5-ulp acos) */
#endif
    if (!(r == r) || r == 0.0 || isinf(r) || x == 0.0) return r;
    double dir = ((z & 3) == 1) ? INFINITY : (((z & 3) == 2) ? -INFINITY : 0.0);
    if (dir == 0.0) return r;
    int steps = 1 + (int)((z >> 8) % ORC_PERTURB_ULPS);
    for (int i = 0; i < steps; i++) r = nextafter(r, dir);
    return r;
}
static inline double p_pow(double x, double y)
{
    double r = pow(x, y);
    if (x == 1.0 || y == 0.5 || (y == floor(y) && fabs(y) <= 8.0)) return r;
    return orc_pert(r, x, y, 1);
}
static inline double p_exp(double x) { return orc_pert(exp(x), x, 0.0, 2); }
static inline double p_log(double x) { return x == 1.0 ? 0.0 : orc_pert(log(x), x, 0.0, 3); }
static inline double p_sin(double x) { return orc_pert(sin(x), x, 0.0, 4); }
static inline double p_cos(double x) { return orc_pert(cos(x), x, 0.0, 5); }
static inline double p_tan(double x) { return orc_pert(tan(x), x, 0.0, 6); }
static inline double p_acos(double x) { return (x == 1.0) ? 0.0 : orc_pert(acos(x), x, 0.0, 7); }
static inline double p_asin(double x) { return orc_pert(asin(x), x, 0.0, 8); }
#define pow p_pow
#define exp p_exp
#define log p_log
#define sin p_sin
#define cos p_cos
#define tan p_tan
#define acos p_acos
#define asin p_asin
#define ORC_EXACT_BEGIN _Pragma("push_macro(\"pow\")") _Pragma("push_macro(\"exp\")")
#define ORC_EXACT_END _Pragma("pop_macro(\"pow\")") _Pragma("pop_macro(\"exp\")")
#else
#define ORC_EXACT_BEGIN
#define ORC_EXACT_END
#endif
/* the float viscosity routine and the density it rounds to float keep the pinned arithmetic in
 * every build (no FMA contraction even under -ffp-contract=fast -march=native) */
#define ORC_PINNED __attribute__((optimize("fp-contract=off")))

/* std::max / std::min semantics (NaN-asymmetric; src/SPLASH.cpp:1267,1270,1483,1574;
 * src/EVAP.cpp:234,247): max(a,b) = (a<b)?b:a ; min(a,b) = (b<a)?b:a */
#define STDMAX(a, b) (((a) < (b)) ? (b) : (a))
#define STDMIN(a, b) (((b) < (a)) ? (b) : (a))

/* src/global.cpp:49-83.  The reference declares these `extern const double` in global.h and
 * defines them in global.cpp, so SOLAR.cpp/EVAP.cpp/SPLASH.cpp cannot constant-fold libm calls
 * on them (pow(e,2.0), dsin(omega), ... run at run time).  They are therefore non-const,
 * externally visible objects here too: in a -fPIC shared object gcc may not fold them either. */
double G_A = 91.86328;
double G_alb_sw = 0.30;
double G_b = 0.2012435;
double G_c = 0.25;
double G_d = 0.50;
double G_kfus = 334000;
double G_G = 9.80665;
double G_Gsc = 1360.8;
double G_L = 0.0065;
double G_Ma = 0.028963;
double G_Mv = 0.01802;
double G_Po = 101325;
double G_R = 8.31447;
double G_To = 288.15;
double G_PI = 3.141592653589793;
double G_pir = (3.141592653589793 / 180.0);
double G_fluidity = 35187037;
double G_e = 0.0167;
double G_eps = 23.44;
double G_omega = 283.0;

/* Out-of-line like the reference's member functions: if gcc inlines them it merges sin(x)/cos(x)
 * pairs into glibc sincos(), whose results differ from sin()/cos() in the last bit now and then;
 * the reference build calls sin and cos separately (objdump -T shows no sincos import). */
#define NOINLINE __attribute__((noinline))
static NOINLINE double dcos(double x) { return cos(x * G_pir); } /* src/SOLAR.cpp:269-278 */
static NOINLINE double dsin(double x) { return sin(x * G_pir); } /* src/SOLAR.cpp:280-289 */
static NOINLINE double dtan(double x) { return tan(x * G_pir); } /* src/SPLASH.cpp:139-148 */

/* ------------------------------------------------------------------ SOLAR */

/* slope sunset angle hs [degrees] of the last orc_solar call on this thread (src/SOLAR.cpp:149-159):
 * the conditioning monitor of the grazing-sun class (DESIGN.md "Conditioning") */
static __thread double tl_last_hs = 0.0;
double orc_last_hs(void) { return tl_last_hs; }

/* src/SOLAR.cpp:352-374 (note the float jd) */
int orc_julian_day(int y, int m, int i)
{
    if (m <= 2.0) {
        y -= 1.0;
        m += 12.0;
    }
    int a = (int)(y / 100);
    int b = 2 - a + (int)(a / 4);
    float jd = (int)(365.25 * (y + 4716)) + (int)(30.6001 * (m + 1)) + i + b - 1524.5;
    int jde = (int)jd;
    return jde;
}

/* src/SOLAR.cpp:291-350 */
static void berger_tls(int n, int kN, double* my_nu_out, double* my_tls_out)
{
    double xee = pow(G_e, 2.0);
    double xec = pow(G_e, 3.0);
    double xse = sqrt(1.0 - xee);

    double xlam = (G_e / 2.0 + xec / 8.0) * (1.0 + xse) * dsin(G_omega);
    xlam -= xee / 4.0 * (0.5 + xse) * dsin(2.0 * G_omega);
    xlam += xec / 8.0 * (1.0 / 3.0 + xse) * dsin(3.0 * G_omega);
    xlam *= 2.0;
    xlam /= G_pir;

    double dlamm = xlam + (n - 80.0) * (360.0 / kN);
    double anm = (dlamm - G_omega);
    double ranm = anm * G_pir;

    double ranv = ranm;
    ranv += (2.0 * G_e - xec / 4.0) * sin(ranm);
    ranv += 5.0 / 4.0 * xee * sin(2.0 * ranm);
    ranv += 13.0 / 12.0 * xec * sin(3.0 * ranm);
    double anv = ranv / G_pir;

    double my_tls = (anv + G_omega);
    if (my_tls < 0) {
        my_tls += 360.0;
    } else if (my_tls > 360) {
        my_tls -= 360.0;
    }
    double my_nu = (my_tls - G_omega);
    if (my_nu < 0) {
        my_nu += 360.0;
    }
    *my_nu_out = my_nu;
    *my_tls_out = my_tls;
}

/* src/SOLAR.cpp:98-124 */
void orc_orbit(int n, int y, double* kN_out, double* nu_out, double* lam_out, double* dr_out,
               double* delta_out)
{
    int kN;
    if (y == 0) {
        kN = 365;
    } else {
        kN = orc_julian_day((y + 1), 1, 1) - orc_julian_day(y, 1, 1);
    }
    double nu, lam;
    berger_tls(n, kN, &nu, &lam);
    double kee = pow(G_e, 2.0);
    double rho = (1.0 - kee) / (1.0 + dcos(nu) * G_e);
    double dr = pow(rho, -1.0);
    dr = pow(dr, 2.0);
    double delta = dsin(lam) * dsin(G_eps);
    delta = asin(delta);
    delta /= G_pir;
    if (kN_out) *kN_out = kN;
    if (nu_out) *nu_out = nu;
    if (lam_out) *lam_out = lam;
    if (dr_out) *dr_out = dr;
    if (delta_out) *delta_out = delta;
}

/* src/SOLAR.cpp:77-267 */
void orc_solar(double lat, double elv, int n, int y, double sw_in, double tc, double slop,
               double asp, double snow, double nd, double sw, orc_srad* out)
{
    if (sw > 1.0) { /* :92 */
        sw = 1.0;
    }
    double dr, delta;
    orc_orbit(n, y, NULL, NULL, NULL, &dr, &delta);

    /* :129-140 */
    double a = dsin(delta) * dcos(lat) * dsin(slop) * dcos(asp) - dsin(delta) * dsin(lat) * dcos(slop);
    double b = dcos(delta) * dcos(lat) * dcos(slop) + dcos(delta) * dsin(lat) * dsin(slop) * dcos(asp);
    double c = dcos(delta) * dsin(slop) * dsin(asp);
    double d = pow(b, 2.0) + pow(c, 2.0) - pow(a, 2.0);
    double sinfirst;
    if (d < 0) {
        sinfirst = (a * c) / (pow(b, 2.0) + pow(c, 2.0));
    } else {
        sinfirst = (a * c + b * pow(d, 0.5)) / (pow(b, 2.0) + pow(c, 2.0));
    }
    double ru = -1 * a + c * sinfirst;
    double rv = b;

    /* :149-159 */
    double hs;
    if ((ru / rv) >= 1.0) {
        hs = 180.0;
    } else if ((ru / rv) <= -1.0) {
        hs = 0.0;
    } else {
        hs = -1.0 * (ru / rv);
        hs = acos(hs);
        hs /= G_pir;
    }

    tl_last_hs = hs;

    /* :164-165 */
    double ra_d = (86400.0 / G_PI) * dr * G_Gsc;
    ra_d *= (ru * hs * G_pir + rv * dsin(hs));

    /* :170-178 */
    double tau_o = (G_c + G_d) * (1.0 + (2.67e-5) * elv);
    double r_in = 86400 * sw_in;
    double tau;
    if (isnan(ra_d) || r_in == 0 || ra_d < r_in) {
        tau = tau_o;
    } else {
        tau = r_in / (ra_d);
    }

    /* :196-203 */
    double sf = 0.0;
    sf = pow(((tau - tau_o * 0.1898) / (tau_o * (1 - 0.1898))), (1 / 0.7410));
    if (isnan(sf)) {
        sf = 0.0;
    } else if (sf > 1.0) {
        sf = 1.0;
    }
    double rnl = (0.0883289 + (1.0 - G_b) * sf) * (G_A + 1.95974 * tc);

    /* :208-224 */
    double max_alb_snw = (1.0 - 0.443700) + (0.443700 * exp(-0.895189 * nd));
    double sfc = snow / (140.0 + snow);
    double alb_v = G_alb_sw - 0.17 * sw;
    double alb = alb_v * (1.0 - sfc) + sfc * max_alb_snw;
    double rw;
    if ((sw_in == 0.0) || (hs == 0.0)) {
        rw = (1.0 - alb) * tau * dr * G_Gsc;
    } else {
        rw = (1.0 - alb) * (r_in) / ((86400.0 / G_PI) * (ru * G_pir * hs + rv * dsin(hs)));
    }

    /* :231-240 */
    double hn;
    if ((rnl - rw * ru) / (rw * rv) >= 1.0) {
        hn = 0;
    } else if ((rnl - rw * ru) / (rw * rv) <= -1.0) {
        hn = 180.0;
    } else {
        hn = acos((rnl - rw * ru) / (rw * rv));
        hn /= G_pir;
    }

    /* :245-246 */
    double rn_d = G_pir * hn * (rw * ru - rnl) + rw * rv * dsin(hn);
    rn_d *= (86400.0 / G_PI);

    /* :252-255 */
    double rnn_d = rw * rv * (dsin(hs) - dsin(hn));
    rnn_d += rw * ru * (hs - hn) * G_pir;
    rnn_d -= rnl * (G_PI - hn * G_pir);
    rnn_d *= (86400.0 / G_PI);

    /* :262 (abs resolves to std::abs(double)) */
    double ts = 1.0691 * tc + 2.7302 * tau - 0.035 * fabs(lat) + 2.334;

    out->ru = ru;
    out->rv = rv;
    out->rw = rw;
    out->rnl = rnl;
    out->hn = hn;
    out->rn_d = rn_d;
    out->rnn_d = rnn_d;
    out->ts = ts;
}

/* ------------------------------------------------------------------ EVAP */

/* src/EVAP.cpp:291-303 */
static double sat_slope(double tc)
{
    double s = exp((tc * 17.269) / (tc + 237.3));
    s /= pow((tc + 237.3), 2.0);
    s *= (17.269) * (237.3) * (610.78);
    return s;
}

/* src/EVAP.cpp:305-317 */
static double enthalpy_vap(double tc)
{
    double lv = (tc + 273.15) / (tc + 273.15 - 33.91);
    lv = pow(lv, 2.0);
    lv *= 1.91846e6;
    return lv;
}

/* src/EVAP.cpp:319-336 */
double orc_elv2pres(double z)
{
    double ep = (G_G * G_Ma) / (G_R * G_L);
    double pa = (1.0 - z * G_L / G_To);
    pa = pow(pa, ep);
    pa *= G_Po;
    return pa;
}

/* src/EVAP.cpp:338-389 */
ORC_EXACT_BEGIN
#ifdef ORC_PERTURB_LIBM
#undef pow
#undef exp
#endif
ORC_PINNED double orc_density_h2o(double tc, double p)
{
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

    double pw = (ko + ca * pbar + cb * pow(pbar, 2.0));
    pw /= (ko + ca * pbar + cb * pow(pbar, 2.0) - pbar);
    pw *= (1.0e3) * po;
    return pw;
}

/* src/EVAP.cpp:405-462.  The reference computes this in `float` with C++'s usual
 * promotions; every cast below is one the C++ compiler inserts implicitly. */
ORC_PINNED float orc_viscosity_h2o(float tc, float p)
{
    float tk_ast = 647.096;
    float rho_ast = 322.0;
    float mu_ast = 1e-6;

    float rho = (float)orc_density_h2o((double)tc, (double)p); /* :413 */

    float tbar = (float)(((double)tc + 273.15) / (double)tk_ast); /* :416 */
    float tbarx = (float)pow((double)tbar, 0.5);                  /* :417 */
    float tbar2 = tbar * tbar;                                    /* :418 */
    float tbar3 = tbar * tbar * tbar;                             /* :419 */
    float rbar = rho / rho_ast;                                   /* :420 */

    /* :423-424 (double expression, rounded to float on assignment) */
    float mu0 = (float)(1.67752 + 2.20462 / (double)tbar + 0.6366564 / (double)tbar2 -
                        0.241605 / (double)tbar3);
    mu0 = (float)(1e2 * (double)tbarx / (double)mu0);

    /* :427-435 */
    float h_array[7][6] = {
        {0.520094, 0.0850895, -1.08374, -0.289555, 0.0, 0.0},
        {0.222531, 0.999115, 1.88797, 1.26613, 0.0, 0.120573},
        {-0.281378, -0.906851, -0.772479, -0.489837, -0.257040, 0.0},
        {0.161913, 0.257399, 0.0, 0.0, 0.0, 0.0},
        {-0.0325372, 0.0, 0.0, 0.0698452, 0.0, 0.0},
        {0.0, 0.0, 0.0, 0.0, 0.00872102, 0.0},
        {0.0, 0.0, 0.0, -0.00435673, 0.0, -0.000593264}};

    float mu1 = 0.0;
    float ctbar = (float)((1.0 / (double)tbar) - 1.0); /* :439 */
    for (int i = 0; i < 6; ++i) {
        float coef1 = (float)pow((double)ctbar, (double)i); /* :443 std::pow(float,int)->double */
        float coef2 = 0.0;
        for (int j = 0; j < 7; ++j) {
            /* :447 float + float*double -> double, rounded to float on assignment */
            coef2 = (float)((double)coef2 +
                            (double)h_array[j][i] * pow((double)rbar - 1.0, (double)j));
        }
        mu1 = mu1 + coef1 * coef2; /* :449 float arithmetic */
    }
    mu1 = expf(rbar * mu1); /* :451 std::exp(float) overload */

    float mu_bar = mu0 * mu1;
    float mu = mu_bar * mu_ast;
    return mu;
}
ORC_EXACT_END

/* src/EVAP.cpp:491-515 */
static double specific_heat(double tc)
{
    double cp;
    if (tc < 0) {
        cp = 1004.5714270;
    } else if (tc > 100) {
        cp = 2031.2260590;
    } else {
        cp = 1.0045714270;
        cp += (2.050632750e-3) * tc;
        cp += -(1.631537093e-4) * tc * tc;
        cp += (6.212300300e-6) * tc * tc * tc;
        cp += -(8.830478888e-8) * tc * tc * tc * tc;
        cp += (5.071307038e-10) * tc * tc * tc * tc * tc;
        cp *= (1.0e3);
    }
    return cp;
}

/* src/EVAP.cpp:465-489 */
static double psychro(double tc, double p)
{
    double cp = specific_heat(tc);
    double lv = enthalpy_vap(tc);
    double ps = (G_Ma * cp * p) / (G_Mv * lv);
    return ps;
}

/* src/EVAP.cpp:82-264 */
void orc_evap(double lat, double elv, double sw, int n, int y, double sw_in, double tc,
              double slop, double asp, double snow, double nd, orc_etr* out)
{
    orc_srad sr;
    orc_solar(lat, elv, n, y, sw_in, tc, slop, asp, snow, nd, sw, &sr);
    double ru = sr.ru, rv = sr.rv, rw = sr.rw, rnl = sr.rnl, hn = sr.hn;
    double rn_d = sr.rn_d, rnn_d = sr.rnn_d;

    double tw; /* :100-105 */
    if (tc < 0.0) {
        tw = 0.0;
    } else {
        tw = tc;
    }

    /* :110-120 */
    double patm = orc_elv2pres(elv);
    double s = sat_slope(tc);
    double lv = enthalpy_vap(tc);
    double pw = orc_density_h2o(tc, patm);
    double g = psychro(tc, patm);
    double econ = s / (lv * pw * (s + g));
    double visc = orc_viscosity_h2o((float)tw, (float)patm);

    double cn = (1.0e3) * econ * fabs(rnn_d) * 0.1;                 /* :124 */
    double eet_d = (1.0e3) * (s / (lv * pw * (s + 0.24 * g))) * rn_d; /* :129 */
    double pet_d = eet_d;                                            /* :140 */
    double rx = (3.6e6) * econ;                                      /* :145 */
    double pet_max = rx * ((rw * (ru + rv)) - rnl);                  /* :148 */

    /* :157-166 */
    double B_r = g / (sw * s);
    double EF = 1 / (B_r + 1.0);
    sw = pet_max * EF;
    if (sw < 0.0 || isnan(sw)) {
        sw = 0.0;
    }

    /* :208-218 */
    double cos_hi = sw / (rw * rv * rx) + rnl / (rw * rv) - ru / rv;
    double hi;
    if (cos_hi >= 1.0) {
        hi = 0.0;
    } else if (cos_hi <= -1.0) {
        hi = 180.0;
    } else {
        hi = acos(cos_hi);
        hi /= G_pir;
    }

    /* :233-249 */
    double snowmelt;
    if (tc >= 3.0) {
        double m = (rn_d / (pw * G_kfus)) * 1000.0;
        snowmelt = STDMIN(snow, m);
    } else {
        snowmelt = 0.0;
    }
    double melt_enrg = (snowmelt / 1000) * pw * G_kfus;
    double AE = rn_d - melt_enrg;
    double sub_c = (AE * econ) * 1000.0;
    double sublimation = STDMIN(snowmelt, sub_c);
    melt_enrg += ((sublimation / 1000.0) / econ);

    /* :254-263 */
    double aet_d = sw * hi * G_pir;
    aet_d += rx * rw * rv * (dsin(hn) - dsin(hi));
    aet_d += (rx * rw * ru - rx * rnl) * (hn - hi) * G_pir;
    aet_d *= (24.0 / G_PI);
    aet_d -= (melt_enrg * econ * 1000.0);
    if (aet_d < 0.0) {
        aet_d = 0.0;
    }

    out->cond = cn;
    out->eet = eet_d;
    out->pet = pet_d;
    out->aet = aet_d;
    out->snowmelt = snowmelt;
    out->sublimation = sublimation;
    out->econ = econ;
    out->pw = pw;
    out->rn_d = rn_d;
    out->visc = visc;
    out->pet_max = pet_max;
}

void orc_solar_batch(int m, const double* lat, const double* elv, const int* n, const int* y,
                     const double* sw_in, const double* tc, const double* slop, const double* asp,
                     const double* snow, const double* nd, const double* sw, double* out)
{
    for (int i = 0; i < m; i++) {
        orc_srad s;
        orc_solar(lat[i], elv[i], n[i], y[i], sw_in[i], tc[i], slop[i], asp[i], snow[i], nd[i],
                  sw[i], &s);
        memcpy(out + (size_t)i * 8, &s, sizeof(s));
    }
}

void orc_evap_batch(int m, const double* lat, const double* elv, const double* sw, const int* n,
                    const int* y, const double* sw_in, const double* tc, const double* slop,
                    const double* asp, const double* snow, const double* nd, double* out)
{
    for (int i = 0; i < m; i++) {
        orc_etr e;
        orc_evap(lat[i], elv[i], sw[i], n[i], y[i], sw_in[i], tc[i], slop[i], asp[i], snow[i],
                 nd[i], &e);
        memcpy(out + (size_t)i * 11, &e, sizeof(e));
    }
}

/* ------------------------------------------------------------------ SPLASH */

/* src/SPLASH.cpp:1921-1961 */
double orc_moist_surf(double depth, double z, double bub_p, double wn, double SAT, double RES,
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

/* src/SPLASH.cpp:1963-2023 */
double orc_inf_GA(double bub_press, double theta_i, double Ksat, double theta_s, double lambda,
                  double P, double tdur, double slop)
{
    double r = P / tdur;
    double h_f = ((2 + 3 * lambda) / (1 + 3 * lambda)) * (bub_press / 2);
    double delta_head = h_f;
    double delta_theta = (theta_s - theta_i);
    double I = 0.0;
    double tp = 0.1;
    double tp_s = 0.1;
    if (r <= Ksat) {
        I = P;
    } else {
        if (delta_theta <= 0.0) {
            tp = 0.0;
            I = Ksat * tdur;
        } else {
            tp = (Ksat * delta_theta * -1.0 * delta_head) / (r * (r - Ksat));
            if (tp <= 0.0 || isnan(tp)) {
                tp = 0.01;
            }
            tp_s = tp / pow(dcos(slop), 2.0);
            I = r * tp_s + (Ksat * (tdur - tp_s) -
                            (delta_head * delta_theta *
                             log(1 - (r * tp_s / (delta_head * delta_theta)))));
        }
    }
    if (I > P) {
        I = P;
    }
    return I;
}

/* src/SPLASH.cpp:2025-2050 */
double orc_get_wtd(double wn, double depth, double bub_press, double theta_s, double theta_r,
                   double lambda)
{
    double theta_i = (wn) / (depth * 1000.0);
    if (theta_i >= theta_s) {
        theta_i = theta_s - 0.001;
    } else if (theta_i <= theta_r) {
        theta_i = theta_r + 0.001;
    }
    double psi_m = bub_press / pow((((theta_i - theta_r) / (theta_s - theta_r))), (1 / lambda));
    double wtd = ((bub_press - psi_m) / 1000.0);
    if (wtd < 0.0 || isnan(wtd)) {
        wtd = 0.01;
    } else if (wtd > depth) {
        wtd = depth;
    }
    return wtd;
}

/* intermediates of the last orc_run_one_day call on this thread (debugging aid; thread-local so
 * that the OpenMP drivers do not fight over one cache line) */
__thread double orc_debug[16];
double* orc_debug_ptr(void) { return orc_debug; }

/* Conditioning bookkeeping (not part of the restatement): number of day steps, on the calling
 * thread, in which the drainage time of src/SPLASH.cpp:1472 was computed from a transmittance
 * that is rounding noise.  With the water table clamped to the column base, the two powers at
 * :1422 are bub/psi_m and bub/fl(psi_m + depth*1000) raised to 3*lambda+1: their bases differ
 * through the ROUNDING of that sum.  Once |psi_m| >~ 2^53*depth*1000 the sum rounds back to psi_m,
 * both powers are bit-identical, T_uns = 0 exactly and t_drain = +inf (sticky): deterministic, and
 * any faithful implementation reproduces it.  In the band below that (increment = a few ulp of
 * psi_m, here 2^42..2^54.5 times depth*1000) T_uns, hence Q = T*Ai/1000, is set by which way
 * those few ulp fall, and so is t_drain = -log(1 - log(Kb)*Au*R/Q)/log(Kb).  No two libm
 * implementations agree there: those day steps are counted. */
static __thread long tl_fragile_events = 0;
long orc_fragile_events(int reset)
{
    long v = tl_fragile_events;
    if (reset) tl_fragile_events = 0;
    return v;
}
/* The critical end of that band: the increment fl(psi_m + depth*1000) - psi_m is an ulp or two
 * of psi_m.  Whether the two powers then come out equal (T_uns = 0 exactly, Q = 0, drainage time
 * +inf, which is STICKY: src/SPLASH.cpp:1468, :1483 keep it infinite for the rest of the run and
 * the upslope inflow of :1367-1371 never switches off) or one ulp apart (a finite drainage time of
 * a few thousand days that halves every day) depends on the last bit of pow(): the reference built
 * against another correct libm takes the other branch and ends the year millimetres apart
 * (tools/conditioning_evidence.py, profiles/r02_conditioning.md).  Counted separately, by a
 * local sensitivity test: psi_m itself is the quotient of a pow() and carries its last-bit
 * uncertainty, so the two bases bub/psi_m and bub/(psi_m + depth*1000) are formed for psi_m and
 * its two neighbouring doubles; the day is decided when all three give identical bases (T_uns = 0
 * in any implementation) or bases at least 3 ulp apart in all three (T_uns > 0 in any
 * implementation), critical otherwise. */
static int64_t ulp_distance(double a, double b)
{
    int64_t ia, ib;
    memcpy(&ia, &a, 8);
    memcpy(&ib, &b, 8);
    if (ia < 0) ia = INT64_MIN - ia;
    if (ib < 0) ib = INT64_MIN - ib;
    return ia > ib ? ia - ib : ib - ia;
}
static int zero_or_not_is_undecided(double bub_press, double psi_m, double w)
{
    int all_zero = 1, all_far = 1;
    double cand[3] = {psi_m, nextafter(psi_m, INFINITY), nextafter(psi_m, -INFINITY)};
    for (int i = 0; i < 3; i++) {
        int64_t d = ulp_distance(bub_press / cand[i], bub_press / (cand[i] + w));
        if (d != 0) all_zero = 0;
        if (d < 3) all_far = 0;
    }
    return !(all_zero || all_far);
}
static __thread long tl_critical_events = 0;
long orc_critical_events(int reset)
{
    long v = tl_critical_events;
    if (reset) tl_critical_events = 0;
    return v;
}

/* src/SPLASH.cpp:920-1588 (live code only; quick_run :150-918 has the same arithmetic and
 * additionally returns pet).  Dead locals (psi_m/wtd/w_z at :999-1022, max_sw :1040, To :1358,
 * T_qs :1321, Q_uns/Q_sat) are not restated. */
void orc_run_one_day(double lat, double elv, int n, int y, double wn, double sw_in, double tc,
                     double pn, double slop, double asp, double snow, double snowfall,
                     const double* soil_info, double qin, double td, double nd, orc_smr* dsoil,
                     orc_etr* dvap_out)
{
    /* :946-968 */
    double SAT = soil_info[0];
    double WP = soil_info[1];
    double FC = soil_info[2];
    double Ksat = soil_info[3];
    double lambda = soil_info[4];
    double depth = soil_info[5];
    double bub_press = soil_info[6];
    double RES = soil_info[7];
    double Au = soil_info[8];
    double Ai = soil_info[9];
    double cellin = soil_info[10];
    double cellout = soil_info[11];
    double KG_o = 1000.0 / (997 * G_G); /* :971 */

    /* :978-990 */
    double theta_s = SAT / (depth * 1000.0);
    double theta_r = RES / (depth * 1000.0);
    double theta_fc = FC / (depth * 1000.0);
    double theta_wp = WP / (depth * 1000.0);
    double theta_q0 = theta_wp + 0.001;
    double theta_i = (wn) / (depth * 1000.0);
    if (theta_i >= theta_s) {
        theta_i = theta_s - 0.001;
    } else if (theta_i <= theta_r) {
        theta_i = theta_r + 0.001;
    }

    double sid_oct = sqrt(Ai / (2.0 * (1 + sqrt(2.0)))); /* :993 */

    /* :1027-1029 */
    double coeff_A = exp(log(33.0) + (1.0 / lambda) * log(theta_fc));
    double Wmax = pow((coeff_A * KG_o / (depth)), (1.0 / ((1 / lambda) + 1.0))) * (depth * 1000.0);

    /* :1042-1048 */
    double sw = ((wn - RES) / (Wmax - RES));
    if (sw < 0.0 || isnan(sw)) {
        sw = 0.0;
    } else if (sw > 1.0) {
        sw = 1.0;
    }

    /* :1225-1231 */
    if (snowfall > 0.0) {
        nd = 0.0;
    } else {
        nd += 1.0;
    }
    snow += snowfall;

    /* :1235-1243 */
    orc_etr dvap;
    orc_evap(lat, elv, sw, n, y, sw_in, tc, slop, asp, snow, nd, &dvap);
    double pw = dvap.pw;
    snow -= dvap.snowmelt;
    double visc = dvap.visc;
    double snowmelt = dvap.snowmelt - dvap.sublimation;

    /* :1260-1272 */
    double int_perm = Ksat / G_fluidity;
    double Ksat_visc = int_perm * ((pw * G_G) / visc) * 3.6;
    double inflow = pn + dvap.cond + snowmelt;
    double surf_moist = orc_moist_surf(depth, 10.0, bub_press, wn, SAT, RES, lambda);
    double wm_th = Wmax / (depth * 1000.0);
    double theta_m = STDMAX(wm_th, theta_i);
    double infi = orc_inf_GA(bub_press, surf_moist, Ksat_visc, theta_s, lambda, inflow, 6.0, slop);
    double ro_h_arg = inflow - infi;
    double ro_h = STDMAX(ro_h_arg, 0.0);
    double R = infi - dvap.aet;

    /* :1275-1284 */
    double Kunsat = Ksat_visc * pow((theta_m / theta_s), (3.0 + (2.0 / lambda)));
    double hyd_grad_in = dtan(slop);
    double hyd_grad_out = dtan(slop);
    double hyd_grad_z = (infi / (Ksat_visc * 24)) - 1.0;
    if (depth >= 0.0) {
        hyd_grad_out = sqrt(pow((hyd_grad_z), 2) + pow(hyd_grad_in, 2));
    }

    /* :1291-1305 */
    double psi_q0 = bub_press / pow((((theta_q0 - theta_r) / (theta_s - theta_r))), (1 / lambda));
    double wtd_q0 = ((bub_press - psi_q0) / 1000.0);
    if (wtd_q0 < 0.0 || isnan(wtd_q0)) {
        wtd_q0 = 0.0;
    } else if (wtd_q0 > depth) {
        wtd_q0 = depth;
    }
    double T_q0 = (Ksat_visc * bub_press / (3.0 * lambda + 1.0)) *
                  (pow((bub_press / psi_q0), (3.0 * lambda + 1.0)) -
                   pow((bub_press / (psi_q0 + (wtd_q0 * 1000.0))), (3.0 * lambda + 1.0)));
    double Q_q0 = T_q0 * hyd_grad_in * ((24.0 * sid_oct * cellout) / (1.0e6));

    /* :1319-1326 */
    double Acs_out_qs = (depth)*sid_oct * cellout;
    double Q_qs = (hyd_grad_in * Ksat_visc * 24.0 * (Acs_out_qs) / 1000.0);
    double Kb = exp((Q_q0 - Q_qs) / ((SAT - WP) * (Ai / 1000.0)));

    /* :1333-1360 */
    Wmax = pow((coeff_A * KG_o / (depth)), (1.0 / ((1 / lambda) + 1.0))) * (depth * 1000.0);
    theta_i = (Wmax) / (depth * 1000.0);
    double psi_m = bub_press / pow((((theta_i - theta_r) / (theta_s - theta_r))), (1 / lambda));
    double wtd = ((bub_press - psi_m) / 1000.0);
    if (wtd < 0.0 || isnan(wtd)) {
        wtd = 0.01;
    } else if (wtd > depth) {
        wtd = depth;
    }
    double Acs_out = (depth - wtd) * cellin * sid_oct;
    double To_uns = (Ksat_visc * bub_press / (3.0 * lambda + 1.0)) *
                    (pow((bub_press / psi_m), (3.0 * lambda + 1.0)) -
                     pow((bub_press / (psi_m + (wtd * 1000.0))), (3.0 * lambda + 1.0)));
    double Qo_uns = To_uns * ((24.0 * cellin * sid_oct) / (1.0e6));
    double Qo_sat = Ksat_visc * 24.0 * (Acs_out) / 1000.0;
    double Qt = (Qo_sat + Qo_uns) * hyd_grad_out;

    /* :1365-1372 */
    double q_in_o = 0.0;
    if ((td <= 0.0) || (qin <= 0.0)) {
        q_in_o = 0.0;
    } else {
        q_in_o = qin * Kb;
    }

    /* :1377-1397 */
    double sm = wn + q_in_o + R;
    double ro_d = 0.0;
    if (sm > SAT) {
        ro_d = (sm - SAT);
        sm = SAT;
        if (R > 0) {
            R -= ro_d;
        }
    } else if (sm < RES) {
        sm = RES;
        ro_d = 0.0;
    }

    /* :1406-1457 */
    theta_i = (sm) / (depth * 1000.0);
    psi_m = bub_press / pow((((theta_i - theta_r) / (theta_s - theta_r))), (1 / lambda));
    wtd = ((bub_press - psi_m) / 1000.0);
    if (wtd < 0.0 || isnan(wtd)) {
        wtd = 0.01;
    } else if (wtd > depth) {
        wtd = depth;
    }
    Acs_out = (depth - wtd) * sid_oct * cellout;
    double T_uns = (Ksat_visc * bub_press / (3.0 * lambda + 1.0)) *
                   (pow((bub_press / psi_m), (3.0 * lambda + 1.0)) -
                    pow((bub_press / (psi_m + (wtd * 1000.0))), (3.0 * lambda + 1.0)));
    T_uns *= ((24.0 * cellout * sid_oct) / (1000.0 * Ai));
    if (T_uns < 0.0 || isnan(T_uns)) {
        T_uns = 0.0;
    }
    if (depth >= 2.0) {
        T_uns += Kunsat * 24.0;
    }
    double T_sat = 0.0;
    if (depth >= 2.0) {
        T_sat = Ksat_visc * 24.0 * ((Acs_out + Ai) / Ai);
    } else {
        T_sat = Ksat_visc * 24.0 * (Acs_out / Ai);
    }
    double T = (T_sat + T_uns) * hyd_grad_out;
    double Q = (T * Ai) / 1000;

    /* :1465-1483 */
    double t_drain = 0.0;
    double tdrain_out = 0.0;
    double q_in_f = 0.0;
    td -= 1.0;
    if ((R > 0.0) && (sm > Wmax)) {
        /* the difference of the two powers of :1422 is ~e*u of either one, u = wtd*1000/psi_m, and
         * carries the rounding of both: a relative noise of 1.6e-16/(e*u), which passes 2e-6 (a fifth
         * of the tolerance, e <= 4) at |psi_m| = 2e10 * wtd*1000 -- the drainage time divides by it */
        if (Au > 0.0 && wtd >= depth && fabs(psi_m) > 2.0e10 * (wtd * 1000.0) &&
            fabs(psi_m) < 2.6e16 * (wtd * 1000.0))
            tl_fragile_events++;
        if (Au > 0.0 && wtd >= depth && isfinite(psi_m) &&
            zero_or_not_is_undecided(bub_press, psi_m, wtd * 1000.0))
            tl_critical_events++;
        t_drain = -1.0 * log(1.0 - (log(Kb) * (Au * R / Q))) / log(Kb);
        q_in_f = (Qt - Au * R * log(Kb)) / Ai;
    }
    if (q_in_f < 0.0 || isnan(q_in_f)) {
        q_in_f = 0.0;
    }
    double td_arg = (td + t_drain) / 2;
    tdrain_out = STDMAX(td_arg, 0.0);

    /* :1488-1506 */
    sm += (q_in_f);
    if (sm > SAT) {
        ro_d += (sm - SAT);
        sm = SAT;
    } else if (sm < RES) {
        sm = RES;
    }
    double ro = ro_d + ro_h;

    /* :1511-1547 */
    theta_i = (sm) / (depth * 1000.0);
    psi_m = bub_press / pow((((theta_i - theta_r) / (theta_s - theta_r))), (1 / lambda));
    wtd = ((bub_press - psi_m) / 1000.0);
    if (wtd < 0.0 || isnan(wtd)) {
        wtd = 0.01;
    } else if (wtd > depth) {
        wtd = depth;
    }
    Acs_out = (depth - wtd) * sid_oct * cellout;
    T_uns = (Ksat_visc * bub_press / (3.0 * lambda + 1.0)) *
            (pow((bub_press / psi_m), (3.0 * lambda + 1.0)) -
             pow((bub_press / (psi_m + (wtd * 1000.0))), (3.0 * lambda + 1.0)));
    T_uns *= ((24.0 * cellout * sid_oct) / (1000.0 * Ai));
    if (T_uns < 0.0 || isnan(T_uns)) {
        T_uns = 0.0;
    }
    T_sat = Ksat_visc * 24.0 * (Acs_out / Ai);
    T = (T_sat + T_uns) * hyd_grad_out;

    /* :1552-1567 */
    sm -= (T);
    if (sm > SAT) {
        sm = SAT;
    } else if (sm < RES) {
        sm = RES;
    }

    /* :1573-1583 */
    double mx = STDMAX(q_in_o, q_in_f);
    double qin_nday = STDMAX(mx, 0.0);

    orc_debug[0] = Kb; orc_debug[1] = Q; orc_debug[2] = T; orc_debug[3] = t_drain;
    orc_debug[4] = q_in_f; orc_debug[5] = q_in_o; orc_debug[6] = R; orc_debug[7] = Ksat_visc;
    orc_debug[8] = hyd_grad_out; orc_debug[9] = Qt; orc_debug[10] = infi; orc_debug[11] = Wmax;
    orc_debug[12] = surf_moist; orc_debug[13] = T_uns; orc_debug[14] = Acs_out; orc_debug[15] = Kb;
    dsoil->sm = sm;
    dsoil->ro = ro;
    dsoil->swe = snow;
    dsoil->tdr = tdrain_out;
    dsoil->sqout = qin_nday;
    dsoil->bflow = T;
    dsoil->nd = nd;
    dsoil->pet = dvap.pet; /* quick_run :910 (dsm.pet = pet_d) */
    if (dvap_out) *dvap_out = dvap;
}

/* src/SPLASH.cpp:1594-1749 */
/* analysis hook (tools/tail_periods.py): when set, every pass of the NEXT spin-ups records the state
 * after its day 1 -- (sm, swe, qin, tdrain, nds) -- as row `pass` of buf [cap][5] */
static __thread double* g_spin_trace = NULL;
static __thread int g_spin_trace_cap = 0;
void orc_set_spin_trace(double* buf, int cap)
{
    g_spin_trace = buf;
    g_spin_trace_cap = cap;
}

int orc_spin_up(double lat, double elv, int n, int y, const double* sw_in, const double* tair,
                const double* pn, double slop, double asp, const double* snowfall,
                const double* soil_info, double* sm_o, double* snow_o, double* qin_o,
                double* tdrain_o, double* ro_o, double* snwage_o, double* pet_o)
{
    double RES = soil_info[7];
    double* wn_vec = (double*)malloc(sizeof(double) * (size_t)n * 7);
    double* ro_vec = wn_vec + n;
    double* snow_vec = ro_vec + n;
    double* tdrain_vec = snow_vec + n;
    double* qin_prev_vec = tdrain_vec + n;
    double* nds_prev_vec = qin_prev_vec + n;
    double* pet_vec = nds_prev_vec + n;
    for (int i = 0; i < n; i++) {
        wn_vec[i] = RES; /* :1633 */
        ro_vec[i] = snow_vec[i] = tdrain_vec[i] = qin_prev_vec[i] = nds_prev_vec[i] = 0.0;
        pet_vec[i] = 0.0;
    }
    orc_smr dsm;
    double wn, snow, qin, td, nd;

    /* :1641-1669 */
    for (int i = 0; i < n; i++) {
        int p = (i == 0) ? (n - 1) : (i - 1);
        wn = wn_vec[p];
        snow = snow_vec[p];
        qin = qin_prev_vec[p];
        td = tdrain_vec[p];
        nd = nds_prev_vec[p];
        orc_run_one_day(lat, elv, (i + 1), y, wn, sw_in[i], tair[i], pn[i], slop, asp, snow,
                        snowfall[i], soil_info, qin, td, nd, &dsm, NULL);
        wn_vec[i] = dsm.sm;
        ro_vec[i] = dsm.ro;
        snow_vec[i] = dsm.swe;
        tdrain_vec[i] = dsm.tdr;
        qin_prev_vec[i] = dsm.sqout;
        nds_prev_vec[i] = dsm.nd;
        pet_vec[i] = dsm.pet;
    }

    /* :1672-1688 (diff_swe, diff_wtd never gate the loop) */
    double start_sm = wn_vec[0];
    orc_run_one_day(lat, elv, 1, y, wn_vec[n - 1], sw_in[0], tair[0], pn[0], slop, asp,
                    snow_vec[n - 1], snowfall[0], soil_info, qin_prev_vec[n - 1],
                    tdrain_vec[n - 1], nds_prev_vec[n - 1], &dsm, NULL);
    double end_sm = dsm.sm;
    double diff_sm = (end_sm - start_sm);
    if (diff_sm < 0) {
        diff_sm = (start_sm - end_sm);
    }

    /* :1696-1743 */
    int spin_count = 1;
    while ((diff_sm > 1.0) && (spin_count < 1000)) {
        for (int i = 0; i < n; i++) {
            int p = (i == 0) ? (n - 1) : (i - 1);
            wn = wn_vec[p];
            snow = snow_vec[p];
            td = tdrain_vec[p];
            qin = qin_prev_vec[p];
            nd = nds_prev_vec[p];
            orc_run_one_day(lat, elv, (i + 1), y, wn, sw_in[i], tair[i], pn[i], slop, asp, snow,
                            snowfall[i], soil_info, qin, td, nd, &dsm, NULL);
            wn_vec[i] = dsm.sm;
            ro_vec[i] = dsm.ro;
            snow_vec[i] = dsm.swe;
            tdrain_vec[i] = dsm.tdr;
            qin_prev_vec[i] = dsm.sqout;
            nds_prev_vec[i] = dsm.nd;
        }
        start_sm = wn_vec[0];
        if (g_spin_trace && spin_count < g_spin_trace_cap) {
            double* t5 = g_spin_trace + 5 * (size_t)spin_count;
            t5[0] = wn_vec[0];
            t5[1] = snow_vec[0];
            t5[2] = qin_prev_vec[0];
            t5[3] = tdrain_vec[0];
            t5[4] = nds_prev_vec[0];
        }
        orc_run_one_day(lat, elv, 1, y, wn_vec[n - 1], sw_in[0], tair[0], pn[0], slop, asp,
                        snow_vec[n - 1], snowfall[0], soil_info, qin_prev_vec[n - 1],
                        tdrain_vec[n - 1], nds_prev_vec[n - 1], &dsm, NULL);
        end_sm = dsm.sm;
        diff_sm = (end_sm - start_sm);
        if (diff_sm < 0) {
            diff_sm = (start_sm - end_sm);
        }
        spin_count++;
    }

    size_t nb = sizeof(double) * (size_t)n;
    if (sm_o) memcpy(sm_o, wn_vec, nb);
    if (snow_o) memcpy(snow_o, snow_vec, nb);
    if (qin_o) memcpy(qin_o, qin_prev_vec, nb);
    if (tdrain_o) memcpy(tdrain_o, tdrain_vec, nb);
    if (ro_o) memcpy(ro_o, ro_vec, nb);
    if (snwage_o) memcpy(snwage_o, nds_prev_vec, nb);
    if (pet_o) memcpy(pet_o, pet_vec, nb);
    free(wn_vec);
    return spin_count;
}

/* src/SPLASH.cpp:1833-1916; series are strided so the grid driver can use [nt][ncell] */
static void run_all_strided(double lat, double elv, int nt, const int* doys, const int* yrs,
                            const double* sw_in, const double* tair, const double* pn,
                            size_t fstride, double wn_last, double slop, double asp,
                            double snow_last, const double* snowfall, const double* soil_info,
                            double qin_last, double td_last, double nds_last, double* const o[11],
                            size_t ostride, double* state_end)
{
    double wn = wn_last, swe = snow_last, qin = qin_last, td = td_last, nd = nds_last;
    orc_smr ds;
    orc_etr dv;
    for (int i = 0; i < nt; i++) {
        orc_run_one_day(lat, elv, doys[i], yrs[i], wn, sw_in[i * fstride], tair[i * fstride],
                        pn[i], slop, asp, swe, snowfall[i], soil_info, qin, td, nd, &ds, &dv);
        size_t k = (size_t)i * ostride;
        if (o[0]) o[0][k] = ds.sm;
        if (o[1]) o[1][k] = ds.ro;
        if (o[2]) o[2][k] = dv.pet;
        if (o[3]) o[3][k] = dv.aet;
        if (o[4]) o[4][k] = ds.swe;
        if (o[5]) o[5][k] = dv.cond;
        if (o[6]) o[6][k] = ds.bflow;
        if (o[7]) o[7][k] = dv.rn_d / 1e6; /* :1907 */
        if (o[8]) o[8][k] = ds.sqout;
        if (o[9]) o[9][k] = ds.tdr;
        if (o[10]) o[10][k] = ds.nd;
        wn = ds.sm;
        swe = ds.swe;
        qin = ds.sqout;
        td = ds.tdr;
        nd = ds.nd;
    }
    if (state_end) {
        state_end[0] = wn;
        state_end[1] = swe;
        state_end[2] = qin;
        state_end[3] = td;
        state_end[4] = nd;
    }
}

void orc_run_all(double lat, double elv, int nt, const int* doys, const int* yrs,
                 const double* sw_in, const double* tair, const double* pn, double wn_last,
                 double slop, double asp, double snow_last, const double* snowfall,
                 const double* soil_info, double qin_last, double td_last, double nds_last,
                 double* wn, double* ro, double* pet, double* aet, double* snow, double* cond,
                 double* bflow, double* netr, double* qin_prev, double* tdrain, double* snwage)
{
    double* const o[11] = {wn, ro, pet, aet, snow, cond, bflow, netr, qin_prev, tdrain, snwage};
    run_all_strided(lat, elv, nt, doys, yrs, sw_in, tair, pn, 1, wn_last, slop, asp, snow_last,
                    snowfall, soil_info, qin_last, td_last, nds_last, o, 1, NULL);
}

/* ------------------------------------------------------------------ R prelude */

/* R/splash.point.R:232-416 (only the fields the hot path consumes; VG alpha/n :387-398 unused).
 * R's `^` is R_pow(): x^2 -> x*x, anything else -> libm pow(). */
void orc_soil_hydro(double sand, double clay, double OM, double fgravel, double bd, orc_soil* out)
{
    double fsand = sand / 100; /* :261-264 */
    double fclay = clay / 100;
    double fOM = OM / 100;
    fgravel = fgravel / 100;
    double depth = 30; /* :268-269 */
    double dp = 1 / ((fOM / 1.3) + ((1 - fOM) / 2.65));
    if (isnan(bd)) { /* :276 */
        bd = (1.5 + (dp - 1.5 - 1.10 * (1 - fclay)) * (1 - exp(-0.022 * depth))) / (1 + 6.27 * fOM);
    }
    if (bd < 0.81) bd = 0.81; /* :280 */
    double sat = 1 - (bd / dp); /* :285 */
    /* :287 */
    double fc = (sat / bd) * (0.4760944 + (0.9402962 - 0.4760944) * pow(fclay, 0.5)) *
                exp(-1 * (0.05472678 * fsand - 0.01 * fOM) / (sat / bd));
    /* :289 */
    double wp_Ball = fc * (0.2018522 + (0.7809203 - 0.2018522) * pow(fclay, 0.5));
    /* :294-295 */
    double wp = -2.464e-05 * sand + 3.650e-03 * clay + 8.680e-03 * OM + 9.393e-03 * bd;
    if (!isnan(wp) && wp >= fc) wp = wp_Ball;
    /* :319-321 */
    double coef_B = (log(1500) - log(33)) / (log(fc) - log(wp));
    double coef_A = exp(log(33) + coef_B * log(fc));
    /* :326-327 */
    double coeff_c = 1000.0 / (997 * 9.80665);
    double theta_c = pow((coeff_c * coef_A / 2.0), (1 / (1 + coef_B)));
    /* :340-341 */
    double theta_r = (0.0285 + 0.00336 * (clay)) * bd;
    if (!isnan(theta_r) && theta_r > wp) theta_r = wp;
    /* :351-353 */
    sat = sat * (1 - fgravel);
    fc = fc * (1 - fgravel);
    wp = wp * (1 - fgravel);
    /* :355-363 */
    double coef_lambda = 1 / coef_B;
    double ksat = 857.48454 / (1 + exp(-2.70927 * fsand + 3.62264 * bd + 7.33398 * fclay +
                                       -8.11795 * (sat - fc) + 18.75552 * fOM +
                                       1.03319 * coef_lambda));
    /* :369-380 */
    double moist_fvol33init = 0.278 * fsand + 0.034 * fclay + 0.022 * fOM -
                              0.018 * (fsand * fOM) - 0.027 * (fclay * fOM) -
                              0.584 * (fsand * fclay) + 0.078;
    double moist_fvol33 = moist_fvol33init + (0.636 * moist_fvol33init - 0.107);
    double bub_init = -21.6 * fsand - 27.93 * fclay - 81.97 * moist_fvol33 +
                      71.12 * (fsand * moist_fvol33) + 8.29 * (fclay * moist_fvol33) +
                      14.05 * (fsand * fclay) + 27.16;
    double bubbling_p = bub_init + (0.02 * (bub_init * bub_init) - 0.113 * bub_init - 0.7);
    bubbling_p = bubbling_p * -101.97162129779;
    if (!isnan(bubbling_p) && bubbling_p > 0) bubbling_p = coef_A * -101.97162129779;

    out->SAT = sat;
    out->FC = fc;
    out->WP = wp;
    out->bd = bd;
    out->AWC = (fc - wp);
    out->Ksat = ksat;
    out->A = coef_A;
    out->B = coef_B;
    out->theta_c = theta_c;
    out->RES = theta_r * (1 - fgravel); /* :410 */
    out->bubbling_p = bubbling_p;
}

/* R/splash.point.R:96-115 */
void orc_soil_info(const double* soil_data, const double* au, double resolution,
                   double* soil_info, double* wmax_r)
{
    orc_soil s;
    orc_soil_hydro(soil_data[0], soil_data[1], soil_data[2], soil_data[3], soil_data[4], &s);
    double depth = soil_data[5];
    soil_info[0] = s.SAT * depth * 1000;
    soil_info[1] = s.WP * depth * 1000;
    soil_info[2] = s.FC * depth * 1000;
    soil_info[3] = s.Ksat;
    soil_info[4] = 1 / s.B;
    soil_info[5] = depth;
    soil_info[6] = s.bubbling_p;
    soil_info[7] = s.RES * depth * 1000;
    soil_info[8] = au ? au[0] : 0.0;
    soil_info[9] = resolution * resolution;
    soil_info[10] = au ? au[1] : 3.0; /* :108-109 / :112-113 */
    soil_info[11] = au ? au[2] : 3.0;
    if (wmax_r) *wmax_r = s.theta_c * depth * 1000; /* :102 */
}

/* R/splash.point.R:576 */
double orc_snowfall_prob(double tc, double lat, double elev)
{
    return 1 / (1 + exp(-0.4710405934 + 1.0473543991 * tc - elev * 0.0004596581 -
                        fabs(lat) * 0.0110592101));
}

/* R/splash.point.R:547-555 for one day, month m in 1..12 */
static double frain_one(double tc, double Tt, double Tr, int m)
{
    double Ttm = Tt + (Tt * dsin((m + 2) / 1.91));
    double Trm = Tr * (0.55 + dsin(m + 4)) * 0.6;
    double x = ((tc - Ttm) / (1.4 * Trm));
    double frain;
    if (tc <= Ttm) {
        frain = 5 * pow(x, 3) + 6.76 * (x * x) + 3.19 * x + 0.5;
    } else {
        frain = 5 * pow(x, 3) - 6.76 * (x * x) + 3.19 * x + 0.5;
    }
    if (frain < 0) frain = 0;
    if (frain > 1) frain = 1;
    return frain;
}

static double snow_threshold(int nt, const double* tc, size_t stride, double lat, double elev)
{
    /* R/splash.point.R:120-122: Tt = max(tc[p_snow>=0.5]); max of nothing is -Inf.  A missing
     * temperature makes p_snow NA, the NA logical subscript selects an NA element and max() (no
     * na.rm) returns NA: one gap in tc makes Tt NA for the whole series. */
    double Tt = -INFINITY;
    for (int t = 0; t < nt; t++) {
        double v = tc[(size_t)t * stride];
        double p = orc_snowfall_prob(v, lat, elev);
        if (isnan(p)) return NAN;
        if (p >= 0.5 && v > Tt) Tt = v;
    }
    return Tt;
}

static void rain_snow_one(double tc, double pn, int month, double lat, double elev, double Tt,
                          double* rain, double* snowfall)
{
    /* R/splash.point.R:124-128 */
    double p = orc_snowfall_prob(tc, lat, elev);
    /* ifelse(NA, ., .) is NA; frain[frain<0] <- 0 leaves NA entries alone */
    double f_rain = isnan(p) ? NAN : ((p >= 0.5) ? frain_one(tc, Tt, 13.3, month) : 1.0);
    *snowfall = pn * (1 - f_rain);
    *rain = pn * f_rain;
}

double orc_snow_partition(int nt, const double* tc, const double* pn, const int* month,
                          double lat, double elev, double* rain, double* snowfall)
{
    double Tt = snow_threshold(nt, tc, 1, lat, elev);
    for (int t = 0; t < nt; t++) {
        rain_snow_one(tc[t], pn[t], month[t], lat, elev, Tt, &rain[t], &snowfall[t]);
    }
    return Tt;
}

/* ------------------------------------------------------------------ unSWC */

/* R/unsSWC.R:14-88 for one site: water held between the surface and uns_depth [m], water-table
 * depth and mean volumetric content of that layer, from the column total wn [mm].
 * cond (optional): |w_uns_0| / |w_uns|, the cancellation factor of the analytical integral
 * (:41-43), large where the result is rounding noise. */
void orc_unswc(const double* soil_data, double uns_depth, int nt, const double* wn, size_t stride,
               double* wn_z, double* wtd_o, double* theta_o, double* cond, size_t ostride)
{
    orc_soil si;
    orc_soil_hydro(soil_data[0], soil_data[1], soil_data[2], 0.0, soil_data[4], &si); /* :55 */
    double theta_s = si.SAT, theta_r = si.RES;                                       /* :59-60 */
    double lambda = 1 / si.B;                                                        /* :62 */
    double bub_press = si.bubbling_p;
    double depth = soil_data[5];
    for (int t = 0; t < nt; t++) {
        double w = wn[(size_t)t * stride] * (100 + soil_data[3]) / 100; /* :56 */
        double theta_i = w / (depth * 1000);                            /* :64-66 */
        if (theta_i >= theta_s) theta_i = theta_s - 0.0001;
        if (theta_i <= theta_r) theta_i = theta_r + 0.0001;
        double psi_m = bub_press / pow((((theta_i - theta_r) / (theta_s - theta_r))), (1 / lambda));
        /* UnsWater, :17-50 */
        double wtdini = (bub_press - psi_m) / 1000;
        double wtd = (wtdini > depth) ? depth : ((wtdini < 0) ? 0 : wtdini);
        double z_uns = (wtd <= uns_depth) ? wtd * 1000 : uns_depth * 1000;
        double w_uns_z = theta_r * z_uns + (((psi_m + z_uns) * (theta_r - theta_s) *
                                             pow((bub_press / (psi_m + z_uns)), lambda)) /
                                            (lambda - 1));
        double w_uns_0 = theta_r * 0 + (((psi_m + 0) * (theta_r - theta_s) *
                                         pow((bub_press / (psi_m + 0)), lambda)) /
                                        (lambda - 1));
        double w_uns = w_uns_z - w_uns_0;
        double sat_swc = (wtd <= uns_depth) ? theta_s * (uns_depth - wtd) * 1000 : 0;
        double swc_tot = (w_uns + sat_swc);
        if (swc_tot <= 0) swc_tot = theta_r * (uns_depth * 1000); /* :74 */
        double uns_theta = (swc_tot / (uns_depth * 1000));        /* :80 */
        swc_tot = swc_tot * (1 - (soil_data[3] / 100));           /* :82 */
        size_t k = (size_t)t * ostride;
        if (wn_z) wn_z[k] = swc_tot;
        if (wtd_o) wtd_o[k] = wtd;
        if (theta_o) theta_o[k] = uns_theta;
        if (cond) cond[k] = fabs(w_uns_0) / fabs(w_uns);
    }
}

/* R/unsSWC.grid.R:14-141 for one cell of the rasters: NOT unSWC applied cell by cell -- the gridded function leaves the
 * stoniness correction out (:82-83: `fgravel = soil_data[[4]]*0`, the `(1-soil_data[[4]]/100)` factor commented out),
 * has no patch for a non-positive integral (R/unsSWC.R:74 has), hands UnsWater the sensor depth as its `depth`
 * argument (:115: `overlay(psi_m, z_uns, theta_r, theta_s, bub_press, lambda, uns_depth, fun = UnsWater)` against
 * the signature (psi_m, uns_depth, theta_r, theta_s, bub_press, lambda, depth), :29, with z_uns <- uns_depth, :71), computes
 * the water table separately against the soil depth (calcwtd, :98-106) and returns the effective saturation of the
 * layer (calc_Se, :120-125).  Outputs = its four files: theta_mean (:87), wtd (:108), swc_top (:115), Se_top (:126).
 * cond (optional): |w_uns_0| / |w_uns|, the cancellation factor of the integral. */
void orc_unswc_grid(const double* soil_data, double uns_depth, int nt, const double* wn, size_t stride,
                    double* w_z_o, double* wtd_o, double* se_o, double* theta_mean_o, double* cond, size_t ostride)
{
    orc_soil si;
    orc_soil_hydro(soil_data[0], soil_data[1], soil_data[2], soil_data[3] * 0, soil_data[4], &si); /* :65 */
    double theta_s = si.SAT, theta_r = si.RES;                                                     /* :67-68 */
    double z_uns_arg = uns_depth;                                                                  /* :69 */
    double lambda = 1 / si.B;                                                                      /* :70 */
    double bub_press = si.bubbling_p;                                                              /* :71 */
    double totdepth = soil_data[5];
    for (int t = 0; t < nt; t++) {
        double w = wn[(size_t)t * stride];
        /* calc_thetai, :79-86 */
        double theta_o = (w / (totdepth * 1000));
        double theta_i = (theta_o >= theta_s) ? theta_s - 0.0001 : ((theta_o <= theta_r) ? theta_r + 0.0001 : theta_o);
        if (isnan(theta_o) || isnan(theta_s) || isnan(theta_r)) theta_i = NAN; /* ifelse(NA, ...) is NA */
        /* :92 */
        double psi_m = bub_press / pow((((theta_i - theta_r) / (theta_s - theta_r))), (1 / lambda));
        /* calcwtd, :98-106 */
        double wtdini = (bub_press - psi_m) / 1000;
        double wtd = (wtdini > totdepth) ? totdepth : ((wtdini < 0) ? 0 : wtdini);
        if (isnan(wtdini)) wtd = NAN;
        /* UnsWater(psi_m, uns_depth = z_uns_arg, theta_r, theta_s, bub_press, lambda, depth = uns_depth), :29-63 */
        double depth = uns_depth, ud = z_uns_arg;
        double wtdini2 = (bub_press - psi_m) / 1000;
        double wtd2 = (wtdini2 > depth) ? depth : ((wtdini2 < 0) ? 0 : wtdini2);
        double z_uns = (wtd2 <= ud) ? wtd2 * 1000 : ud * 1000;
        double w_uns_z = theta_r * z_uns + (((psi_m + z_uns) * (theta_r - theta_s) *
                                             pow((bub_press / (psi_m + z_uns)), lambda)) /
                                            (lambda - 1));
        double w_uns_0 = theta_r * 0 + (((psi_m + 0) * (theta_r - theta_s) *
                                         pow((bub_press / (psi_m + 0)), lambda)) /
                                        (lambda - 1));
        double w_uns = w_uns_z - w_uns_0;
        double sat_swc = (wtd2 <= ud) ? theta_s * (ud - wtd2) * 1000 : 0;
        double swc_tot = w_uns + sat_swc;
        if (isnan(wtdini2)) swc_tot = NAN;
        /* calc_Se, :120-125 */
        double theta_i_top = swc_tot / (uns_depth * 1000);
        double Se = theta_i_top / theta_s;
        Se = (Se > 1) ? 1 : ((Se < 0) ? 0 : Se);
        if (isnan(theta_i_top)) Se = NAN;
        size_t k = (size_t)t * ostride;
        if (w_z_o) w_z_o[k] = swc_tot;
        if (wtd_o) wtd_o[k] = wtd;
        if (se_o) se_o[k] = Se;
        if (theta_mean_o) theta_mean_o[k] = theta_i;
        if (cond) cond[k] = fabs(w_uns_0) / fabs(w_uns);
    }
}

/* ------------------------------------------------------------------ drivers */

static const double kNaN = NAN;

/* R/splash.point.R:29-225 */
void orc_point(int nt, const int* doy, const int* year, const int* month, const int* month_id,
               int nmonth, const double* sw_in, const double* tc, const double* pn,
               size_t stride, double lat, double elev, double slop, double asp,
               const double* soil_data, const double* au, double resolution,
               const orc_opts* opt, double* const out[9], size_t out_stride, orc_cell_aux* aux)
{
    int monthly = opt ? opt->monthly_out : 0;
    double soil_info[12], wmax_r;
    orc_fragile_events(1);
    orc_critical_events(1);
    orc_soil_info(soil_data, au, resolution, soil_info, &wmax_r); /* :96-115 */

    /* :120-128 */
    double* rain = (double*)malloc(sizeof(double) * (size_t)nt * 2);
    double* snowfall = rain + nt;
    double Tt = snow_threshold(nt, tc, stride, lat, elev);
    for (int t = 0; t < nt; t++) {
        rain_snow_one(tc[(size_t)t * stride], pn[(size_t)t * stride], month[t], lat, elev, Tt,
                      &rain[t], &snowfall[t]);
    }
    double asp_c = asp - 180; /* :131 */

    /* :141-152: first 365 elements, n=365, y=y[1] */
    int ns = 365;
    double* f365 = (double*)malloc(sizeof(double) * (size_t)ns * 9);
    double* sw_av = f365;
    double* tc_av = f365 + ns;
    double* sm = f365 + 2 * ns;
    double* snw = f365 + 3 * ns;
    double* qn = f365 + 4 * ns;
    double* tdv = f365 + 5 * ns;
    double* age = f365 + 6 * ns;
    double* petv = f365 + 7 * ns;
    for (int t = 0; t < ns; t++) {
        sw_av[t] = sw_in[(size_t)t * stride];
        tc_av[t] = tc[(size_t)t * stride];
    }
    int p1 = orc_spin_up(lat, elev, ns, year[0], sw_av, tc_av, rain, slop, asp_c, snowfall,
                         soil_info, NULL, NULL, NULL, NULL, NULL, NULL, petv);
    double spet = 0.0, sp = 0.0; /* :147-150 (R sum() accumulates in long double) */
    {
        long double a = 0.0L, b = 0.0L;
        for (int t = 0; t < ns; t++) {
            if (!isnan(petv[t])) a += petv[t];
            double P = rain[t] + snowfall[t];
            if (!isnan(P)) b += P;
        }
        spet = (double)a;
        sp = (double)b;
    }
    double AI = spet / sp;
    soil_info[11] = AI; /* R soil_info[12] <- AI overwrites ncellout (1-based) */
    int p2 = orc_spin_up(lat, elev, ns, year[0], sw_av, tc_av, rain, slop, asp_c, snowfall,
                         soil_info, sm, snw, qn, tdv, NULL, age, NULL);

    /* :158-172 */
    double* series = (double*)malloc(sizeof(double) * (size_t)nt * 9);
    double* o11[11];
    for (int k = 0; k < 8; k++) o11[k] = series + (size_t)k * nt;
    o11[8] = o11[9] = o11[10] = NULL;
    double state_end[5];
    run_all_strided(lat, elev, nt, doy, year, sw_in, tc, rain, stride, sm[ns - 1], slop, asp_c,
                    snw[ns - 1], snowfall, soil_info, qn[ns - 1], tdv[ns - 1], age[ns - 1],
                    (double* const*)o11, 1, state_end);

    /* :197-201 */
    double RES = soil_info[7];
    double* sm_lim = series + (size_t)8 * nt;
    for (int t = 0; t < nt; t++) {
        double v = (series[t] - RES) / (wmax_r - RES);
        if (v < 0) v = 0.0;
        if (v > 1) v = 1.0;
        sm_lim[t] = v;
    }

    if (aux) {
        memcpy(aux->soil_info, soil_info, sizeof(soil_info));
        aux->wmax_r = wmax_r;
        aux->Tt = Tt;
        aux->AI = AI;
        aux->passes1 = p1;
        aux->passes2 = p2;
        aux->state0[0] = sm[ns - 1];
        aux->state0[1] = snw[ns - 1];
        aux->state0[2] = qn[ns - 1];
        aux->state0[3] = tdv[ns - 1];
        aux->state0[4] = age[ns - 1];
        memcpy(aux->state1, state_end, sizeof(state_end));
        aux->fragile = (int)orc_fragile_events(0);
        aux->critical = (int)orc_critical_events(0);
    }

    if (!monthly) {
        for (int k = 0; k < 9; k++) {
            if (!out[k]) continue;
            for (int t = 0; t < nt; t++) out[k][(size_t)t * out_stride] = series[(size_t)k * nt + t];
        }
    } else {
        /* :207-214: mean for wn(0), snow(4), sm_lim(8); sum for the rest (na.rm=T).
         * R's mean() is a long-double two-pass mean; sum() a long-double sum. */
        for (int k = 0; k < 9; k++) {
            if (!out[k]) continue;
            int is_mean = (k == 0 || k == 4 || k == 8);
            const double* x = series + (size_t)k * nt;
            int t = 0;
            for (int m = 0; m < nmonth; m++) {
                int t0 = t;
                long double s = 0.0L;
                int cnt = 0;
                while (t < nt && month_id[t] == m) {
                    if (!isnan(x[t])) {
                        s += x[t];
                        cnt++;
                    }
                    t++;
                }
                double v;
                if (is_mean) {
                    if (cnt == 0) {
                        v = kNaN;
                    } else {
                        long double mu = s / cnt, r = 0.0L;
                        for (int q = t0; q < t; q++)
                            if (!isnan(x[q])) r += (x[q] - mu);
                        mu += r / cnt;
                        v = (double)mu;
                    }
                } else {
                    v = (double)s;
                }
                out[k][(size_t)m * out_stride] = v;
            }
        }
    }
    free(series);
    free(f365);
    free(rain);
}

/* threads a parallel region asked for n really gets (what the baseline's loops run on) */
int orc_threads_obtained(int n)
{
    int got = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(n > 0 ? n : omp_get_max_threads())
    {
#pragma omp single
        got = omp_get_num_threads();
    }
#else
    (void)n;
#endif
    return got;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_grid(int ncell, int nt, const int* doy, const int* year, const int* month,
              const int* month_id, int nmonth, const double* sw_in, const double* tc,
              const double* pn, const double* lat, const double* elev, const double* slop,
              const double* asp, const double* soil, const double* au, const double* resolution,
              const orc_opts* opt, double* const out[9], orc_cell_aux* aux)
{
    int nthreads = (opt && opt->n_threads > 0) ? opt->n_threads : orc_max_threads();
    int nt_out = (opt && opt->monthly_out) ? nmonth : nt;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
#endif
    for (int c = 0; c < ncell; c++) {
        double sd[6], a3[3];
        int bad = 0;
        for (int k = 0; k < 6; k++) {
            sd[k] = soil[(size_t)k * ncell + c];
            if (k != 4 && isnan(sd[k])) bad = 1; /* bd NA is legal (R/splash.point.R:276) */
        }
        if (au)
            for (int k = 0; k < 3; k++) a3[k] = au[(size_t)k * ncell + c];
        if (isnan(lat[c]) || isnan(elev[c]) || isnan(slop[c]) || isnan(asp[c]) ||
            isnan(sw_in[c]) || isnan(tc[c]) || isnan(pn[c]))
            bad = 1;
        double* o9[9];
        for (int k = 0; k < 9; k++) o9[k] = out[k] ? out[k] + c : NULL;
        if (bad) {
            for (int k = 0; k < 9; k++)
                if (o9[k])
                    for (int t = 0; t < nt_out; t++) o9[k][(size_t)t * ncell] = kNaN;
            if (aux) memset(&aux[c], 0, sizeof(orc_cell_aux));
            continue;
        }
        orc_point(nt, doy, year, month, month_id, nmonth, sw_in + c, tc + c, pn + c,
                  (size_t)ncell, lat[c], elev[c], slop[c], asp[c], sd, au ? a3 : NULL,
                  resolution[c], opt, (double* const*)o9, (size_t)ncell, aux ? &aux[c] : NULL);
    }
}

void orc_grid_transient(int ncell, int nt, const int* doy, const int* year, const int* month,
                        const double* sw_in, const double* tc, const double* pn,
                        const double* lat, const double* elev, const double* slop,
                        const double* asp, const double* soil_info, const double* Tt,
                        double* state, double* const out5[5], int n_threads, int* fragile)
{
    int nthreads = n_threads > 0 ? n_threads : orc_max_threads();
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
#endif
    for (int c = 0; c < ncell; c++) {
        double si[12];
        for (int k = 0; k < 12; k++) si[k] = soil_info[(size_t)k * ncell + c];
        double wn = state[c], swe = state[(size_t)ncell + c], qin = state[2 * (size_t)ncell + c];
        double td = state[3 * (size_t)ncell + c], nd = state[4 * (size_t)ncell + c];
        double asp_c = asp[c]; /* already asp-180 (R/splash.point.R:131) */
        orc_fragile_events(1);
        orc_smr ds;
        orc_etr dv;
        for (int t = 0; t < nt; t++) {
            size_t k = (size_t)t * ncell + c;
            double rain, snf;
            rain_snow_one(tc[k], pn[k], month[t], lat[c], elev[c], Tt[c], &rain, &snf);
            orc_run_one_day(lat[c], elev[c], doy[t], year[t], wn, sw_in[k], tc[k], rain, slop[c],
                            asp_c, swe, snf, si, qin, td, nd, &ds, &dv);
            if (out5[0]) out5[0][k] = ds.sm;
            if (out5[1]) out5[1][k] = ds.swe;
            if (out5[2]) out5[2][k] = ds.ro;
            if (out5[3]) out5[3][k] = dv.aet;
            if (out5[4]) out5[4][k] = dv.pet;
            wn = ds.sm;
            swe = ds.swe;
            qin = ds.sqout;
            td = ds.tdr;
            nd = ds.nd;
        }
        state[c] = wn;
        state[(size_t)ncell + c] = swe;
        state[2 * (size_t)ncell + c] = qin;
        state[3 * (size_t)ncell + c] = td;
        state[4 * (size_t)ncell + c] = nd;
        if (fragile) fragile[c] = (int)orc_fragile_events(0);
    }
}


/* Transient loop with the conditioning monitors the parity tests need (TEST INFRASTRUCTURE).
 *   out11   wn, ro, pet, aet, snow, cond, bflow, netr, qin_prev, tdrain, snwage : each [nt][ncell]
 *           or NULL (the 11 series of SPLASH::run_all, src/SPLASH.cpp:1913-1915)
 *   prev    NULL: the state is carried from day to day (a trajectory, as run_all does).
 *           non-NULL [5][nt][ncell] (wn, snow, qin, td, nd AFTER day t, e.g. from another
 *           implementation): day t starts from prev[.][t-1] (day 0 from `state`), i.e. every
 *           cell-day is an independent one-day transition ("one-step" parity: a difference on
 *           one day cannot propagate into the following days).
 *   events  optional [nt][ncell] bit mask of the cell-days on which the reference's own result is
 *           rounding noise (DESIGN.md "Conditioning"):
 *             bit0 noise-level drainage denominator (src/SPLASH.cpp:1422, :1472)
 *             bit1 sun grazing the slope: 0 < hs < hs_lim_deg (src/SOLAR.cpp:149-165, :223)
 *             bit2 the bucket's overflow residual decides the drainage branch: 0 < |R| < 1e-10
 *                  after `R -= ro_d` (src/SPLASH.cpp:1386-1390, :1470)
 *             bit3 the critical end of bit0: zero-or-one-ulp transmittance, sticky infinite
 *                  drainage time or not (see orc_critical_events)
 *   hs_out  optional [nt][ncell] float: hs in degrees (evidence tables)                       */
void orc_grid_transient2(int ncell, int nt, const int* doy, const int* year, const int* month,
                         const double* sw_in, const double* tc, const double* pn,
                         const double* lat, const double* elev, const double* slop,
                         const double* asp, const double* soil_info, const double* Tt,
                         double* state, const double* prev, double* const out11[11],
                         unsigned char* events, double hs_lim_deg, float* hs_out, int n_threads)
{
    int nthreads = n_threads > 0 ? n_threads : orc_max_threads();
    (void)nthreads;
    const size_t N = (size_t)ncell, plane = (size_t)nt * N;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
#endif
    for (int c = 0; c < ncell; c++) {
        double si[12];
        for (int k = 0; k < 12; k++) si[k] = soil_info[(size_t)k * N + c];
        double wn = state[c], swe = state[N + c], qin = state[2 * N + c];
        double td = state[3 * N + c], nd = state[4 * N + c];
        orc_smr ds;
        orc_etr dv;
        for (int t = 0; t < nt; t++) {
            size_t k = (size_t)t * N + c;
            if (prev && t > 0) {
                size_t kp = k - N;
                wn = prev[kp];
                swe = prev[plane + kp];
                qin = prev[2 * plane + kp];
                td = prev[3 * plane + kp];
                nd = prev[4 * plane + kp];
            }
            double rain, snf;
            rain_snow_one(tc[k], pn[k], month[t], lat[c], elev[c], Tt[c], &rain, &snf);
            orc_fragile_events(1);
            orc_critical_events(1);
            orc_run_one_day(lat[c], elev[c], doy[t], year[t], wn, sw_in[k], tc[k], rain, slop[c],
                            asp[c] /* already asp-180 */, swe, snf, si, qin, td, nd, &ds, &dv);
            if (events) {
                unsigned char e = 0;
                double hs = orc_last_hs(), R = orc_debug[6];
                if (orc_fragile_events(0) > 0) e |= 1;
                if (orc_critical_events(0) > 0) e |= 8;
                if (hs > 0.0 && hs < hs_lim_deg) e |= 2;
                if (R != 0.0 && fabs(R) < 1e-10) e |= 4;
                events[k] = e;
            }
            if (hs_out) hs_out[k] = (float)orc_last_hs();
            if (out11[0]) out11[0][k] = ds.sm;
            if (out11[1]) out11[1][k] = ds.ro;
            if (out11[2]) out11[2][k] = dv.pet;
            if (out11[3]) out11[3][k] = dv.aet;
            if (out11[4]) out11[4][k] = ds.swe;
            if (out11[5]) out11[5][k] = dv.cond;
            if (out11[6]) out11[6][k] = ds.bflow;
            if (out11[7]) out11[7][k] = dv.rn_d / 1e6;
            if (out11[8]) out11[8][k] = ds.sqout;
            if (out11[9]) out11[9][k] = ds.tdr;
            if (out11[10]) out11[10][k] = ds.nd;
            wn = ds.sm;
            swe = ds.swe;
            qin = ds.sqout;
            td = ds.tdr;
            nd = ds.nd;
        }
        state[c] = wn;
        state[N + c] = swe;
        state[2 * N + c] = qin;
        state[3 * N + c] = td;
        state[4 * N + c] = nd;
    }
}

/* ------------------------------------------------------------------ static rasters
 * PARITY UNPINNED: raster::terrain, raster::area and raster::focal are not part of the reference
 * tree (DESCRIPTION:10 imports them) and no R runs here.  orc_terrain restates the published
 * algorithms those functions document (Horn 1981 for slope/aspect on a lon/lat grid with metric
 * cell spacing per row, D8 steepest descent with raster's power-of-two codes); orc_ncellflow
 * restates R/upslope_area.R:140-165, which IS reference code, assuming raster::focal hands the
 * window to `fun` in row order. */
void orc_terrain(int H, int W, const double* elev, double ymax, double yres, double xres,
                 int na_to_zero, double* slope, double* aspect, double* lat_o, double* res_o,
                 double* flowdir)
{
    const double Rearth = 6378137.0;
    for (int r = 0; r < H; r++) {
        const double lat = ymax - ((double)r + 0.5) * yres;
        const double dy = Rearth * (yres * G_pir);
        const double dx = Rearth * cos(lat * G_pir) * (xres * G_pir);
        for (int c = 0; c < W; c++) {
            size_t i = (size_t)r * W + c;
            double z = elev[i];
            if (lat_o) lat_o[i] = isnan(z) ? NAN : lat;
            if (res_o) res_o[i] = isnan(z) ? NAN : sqrt(dx * dy);
            double sl = NAN, as = NAN, fd = NAN;
            if (r > 0 && r < H - 1 && c > 0 && c < W - 1 && !isnan(z)) {
                double a = elev[i - W - 1], b = elev[i - W], cc = elev[i - W + 1];
                double d = elev[i - 1], f = elev[i + 1];
                double g = elev[i + W - 1], h = elev[i + W], k = elev[i + W + 1];
                double sum8 = a + b + cc + d + f + g + h + k;
                if (!isnan(sum8)) {
                    double zx = ((cc + 2.0 * f + k) - (a + 2.0 * d + g)) / (8.0 * dx);
                    double zy = ((a + 2.0 * b + cc) - (g + 2.0 * h + k)) / (8.0 * dy);
                    sl = atan(sqrt(zx * zx + zy * zy)) / G_pir;
                    if (zx == 0.0 && zy == 0.0) {
                        as = 90.0;
                    } else {
                        as = atan2(-zx, -zy) / G_pir;
                        if (as < 0.0) as += 360.0;
                    }
                }
                double dd = sqrt(dx * dx + dy * dy);
                double nb[8] = {f, k, h, g, d, a, b, cc};
                double dist[8] = {dx, dd, dy, dd, dx, dd, dy, dd};
                double best = -INFINITY;
                int code = 0, any = 0;
                for (int q = 0; q < 8; q++) {
                    if (!isnan(nb[q])) {
                        double drop = (z - nb[q]) / dist[q];
                        if (drop > best) {
                            best = drop;
                            code = 1 << q;
                        }
                        any = 1;
                    }
                }
                if (any) fd = (double)code;
            }
            if (na_to_zero && !isnan(z)) { /* R/splash.grid.R:107 */
                if (isnan(sl)) sl = 0.0;
                if (isnan(as)) as = 0.0;
            }
            if (slope) slope[i] = sl;
            if (aspect) aspect[i] = as;
            if (flowdir) flowdir[i] = fd;
        }
    }
}

/* The LARGE-GRID terrain of the reference (R/splash.grid.R:77-94, ncell > 1e7 or 0.1 < res < 1 degree): the DEM is written
 * as 16-bit integers (`writeRaster(dem, "rawdem.tif", datatype = 'INT2S')`, R/upslope_area.R:64) and
 *     gdaldem slope  -s 111120 -compute_edges rawdem.tif slope_deg.tif
 *     gdaldem aspect -zero_for_flat -compute_edges rawdem.tif aspect_deg.tif          (R/splash.grid.R:83-84)
 * PARITY UNPINNED: GDAL is a system tool of unknown version, absent from tree and image.  This restates gdaldem's
 * published algorithm (gdaldem_lib.cpp, Horn's formula, the default): window 0 1 2 / 3 4 5 / 6 7 8 from the north-west;
 *   slope  = atan( sqrt(dx^2 + dy^2) / (8 * scale) ),  dx = ((w0+2w3+w6) - (w2+2w5+w8)) / ewres, dy = ((w6+2w7+w8) - (w0+2w1+w2)) / nsres
 *            -- ONE scale for both axes at every latitude (-s 111120: metres per degree), resolutions in degrees;
 *   aspect = atan2(dy', -dx') with dx' = (w2+2w5+w8) - (w0+2w3+w6), dy' = (w6+2w7+w8) - (w0+2w1+w2) (no cell sizes),
 *            as an azimuth (> 90: 450 - a, else 90 - a; 360 -> 0); flat (dx' = dy' = 0): 0 with -zero_for_flat;
 *   -compute_edges: on the first / last LINE the missing row is 2 * line - next line with the column index clamped,
 *            on the first / last COLUMN of the other lines the missing column is 2 * column - next column; a
 *            missing (nodata) neighbour takes the centre's value; a missing centre is nodata.
 * round_int != 0: elevations rounded to whole metres first (the INT2S file; R's round(): half to even). */
static double gd_interpol(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : 2 * a - b; }
void orc_terrain_gdal(int H, int W, const double* elev, double xres, double yres, double scale, int round_int,
                      double* slope, double* aspect)
{
    double* z = (double*)malloc(sizeof(double) * (size_t)H * W);
    for (size_t i = 0; i < (size_t)H * W; i++) z[i] = round_int ? nearbyint(elev[i]) : elev[i];
#define Z(r, c) z[(size_t)(r) * W + (c)]
    for (int r = 0; r < H; r++) {
        for (int c = 0; c < W; c++) {
            size_t i = (size_t)r * W + c;
            double w[9];
            int cm = c > 0 ? c - 1 : c, cp = c < W - 1 ? c + 1 : c; /* clamped (first / last line) */
            if (H < 2 || W < 2) {
                for (int q = 0; q < 9; q++) w[q] = Z(r, c);
            } else if (r == 0 || r == H - 1) {
                int rn = r == 0 ? 1 : H - 2; /* the line next to the edge */
                double mid[3] = {Z(r, cm), Z(r, c), Z(r, cp)}, nxt[3] = {Z(rn, cm), Z(rn, c), Z(rn, cp)}, ext[3];
                for (int q = 0; q < 3; q++) ext[q] = gd_interpol(mid[q], nxt[q]);
                for (int q = 0; q < 3; q++) {
                    w[3 + q] = mid[q];
                    w[r == 0 ? q : 6 + q] = ext[q];
                    w[r == 0 ? 6 + q : q] = nxt[q];
                }
            } else {
                for (int dr = -1; dr <= 1; dr++) {
                    double l, m = Z(r + dr, c), rt;
                    if (c == 0) {
                        rt = Z(r + dr, 1);
                        l = gd_interpol(m, rt);
                    } else if (c == W - 1) {
                        l = Z(r + dr, W - 2);
                        rt = gd_interpol(m, l);
                    } else {
                        l = Z(r + dr, c - 1);
                        rt = Z(r + dr, c + 1);
                    }
                    w[3 * (dr + 1)] = l;
                    w[3 * (dr + 1) + 1] = m;
                    w[3 * (dr + 1) + 2] = rt;
                }
            }
            double sl = NAN, as = NAN;
            if (!isnan(w[4])) {
                for (int q = 0; q < 9; q++)
                    if (isnan(w[q])) w[q] = w[4];
                double dx = ((w[0] + w[3] + w[3] + w[6]) - (w[2] + w[5] + w[5] + w[8])) / xres;
                double dy = ((w[6] + w[7] + w[7] + w[8]) - (w[0] + w[1] + w[1] + w[2])) / yres;
                sl = atan(sqrt(dx * dx + dy * dy) / (8 * scale)) / G_pir;
                double ax = ((w[2] + w[5] + w[5] + w[8]) - (w[0] + w[3] + w[3] + w[6]));
                double ay = ((w[6] + w[7] + w[7] + w[8]) - (w[0] + w[1] + w[1] + w[2]));
                if (ax == 0 && ay == 0) {
                    as = 0.0; /* -zero_for_flat */
                } else {
                    as = atan2(ay, -ax) / G_pir;
                    as = as > 90.0 ? 450.0 - as : 90.0 - as;
                    if (as == 360.0) as = 0.0;
                }
            }
            if (slope) slope[i] = sl;
            if (aspect) aspect[i] = as;
        }
    }
#undef Z
    free(z);
}

/* R/upslope_area.R:140-165 (compare() under raster::focal with a 3x3 window of ones) */
void orc_ncellflow(int H, int W, const double* fd, const double* codes9, int pad, double* out)
{
    for (int r = 0; r < H; r++) {
        for (int c = 0; c < W; c++) {
            size_t i = (size_t)r * W + c;
            if ((r == 0 || r == H - 1 || c == 0 || c == W - 1) && !pad) {
                out[i] = NAN;
                continue;
            }
            int nna = 0, nmatch = 0;
            for (int q = 0; q < 9; q++) {
                int rr = r + q / 3 - 1, cc = c + q % 3 - 1;
                double v = NAN;
                if (rr >= 0 && rr < H && cc >= 0 && cc < W) v = fd[(size_t)rr * W + cc];
                if (isnan(v)) {
                    nna++; /* length(x[is.na(x)]), :150 */
                } else if (v == codes9[q]) {
                    nmatch++; /* comp <- x[!is.na(x)] == flow[!is.na(x)], :153-154 */
                }
            }
            if (nmatch == 0) nmatch = 1; /* nmatch[nmatch==0] <- 1, :157 */
            out[i] = (nna == 9) ? NAN : (double)nmatch;
        }
    }
}

/* Sink filling and D8 contributing area: the remaining steps of R/upslope_area.R (:15-16
 * topmodel::sinkfill, :21 topidx()$area; :68-74 TauDEM pitremove/aread8; :85-96 fix_coast and
 * ups_ncell * cell area).  The third-party codes are not in the reference tree (parity unpinned):
 * these are plain serial statements of the published algorithms the product implements --
 * Planchon & Darboux (2002) relaxation, Kahn-ordered D8 accumulation -- written independently of
 * the kernels (raster-order sweeps; an explicit FIFO). */
void orc_sinkfill(int H, int W, const double* z, double resolution_m, double degree, double* w)
{
    const double eps = tan(degree * G_pir) * resolution_m, eps_d = eps * sqrt(2.0);
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) {
            size_t i = (size_t)r * W + c;
            int outlet = (r == 0 || r == H - 1 || c == 0 || c == W - 1);
            if (!outlet)
                for (int q = 0; q < 9; q++)
                    if (isnan(z[(size_t)(r + q / 3 - 1) * W + (c + q % 3 - 1)])) outlet = 1;
            w[i] = isnan(z[i]) ? z[i] : (outlet ? z[i] : INFINITY);
        }
    int changed = 1;
    while (changed) {
        changed = 0;
        for (int pass = 0; pass < 2; pass++) /* down-right, then up-left */
            for (int rr = 1; rr < H - 1; rr++)
                for (int cc = 1; cc < W - 1; cc++) {
                    int r = pass ? H - 1 - rr : rr, c = pass ? W - 1 - cc : cc;
                    size_t i = (size_t)r * W + c;
                    if (!(w[i] > z[i])) continue;
                    double m = INFINITY;
                    for (int q = 0; q < 9; q++) {
                        if (q == 4) continue;
                        double cand = w[(size_t)(r + q / 3 - 1) * W + (c + q % 3 - 1)] + ((q & 1) ? eps : eps_d);
                        if (cand < m) m = cand;
                    }
                    double nw = (m > z[i]) ? m : z[i];
                    if (nw < w[i]) {
                        w[i] = nw;
                        changed = 1;
                    }
                }
    }
}

static long long orc_d8_target(double code, int tau, int r, int c, int H, int W)
{
    static const int DR[8] = {0, 1, 1, 1, 0, -1, -1, -1}, DC[8] = {1, 1, 0, -1, -1, -1, 0, 1}; /* E SE S SW W NW N NE */
    if (!(code >= 1.0) || code != floor(code)) return -1;
    int k = (int)code, q = -1;
    if (tau) {
        static const int TAU[9] = {-1, 0, 7, 6, 5, 4, 3, 2, 1}; /* 1 E, 2 NE, 3 N, 4 NW, 5 W, 6 SW, 7 S, 8 SE */
        if (k > 8) return -1;
        q = TAU[k];
    } else {
        for (int b = 0; b < 8; b++)
            if (k == (1 << b)) q = b;
        if (q < 0) return -1;
    }
    int rr = r + DR[q], cc = c + DC[q];
    if (rr < 0 || rr >= H || cc < 0 || cc >= W) return -1;
    return (long long)rr * W + cc;
}

void orc_flow_accumulation(int H, int W, const double* fd, const double* elev, int tau,
                           const double* resolution, double* ups_ncell, double* ups_area)
{
    size_t n = (size_t)H * W;
    long long* tgt = (long long*)malloc(n * sizeof(long long));
    int* indeg = (int*)calloc(n, sizeof(int));
    double* cnt = (double*)malloc(n * sizeof(double));
    size_t* fifo = (size_t*)malloc(n * sizeof(size_t));
    for (size_t i = 0; i < n; i++) {
        int cell = elev ? !isnan(elev[i]) : !isnan(fd[i]);
        cnt[i] = cell ? 1.0 : NAN;
        tgt[i] = cell ? orc_d8_target(fd[i], tau, (int)(i / W), (int)(i % W), H, W) : -1;
    }
    for (size_t i = 0; i < n; i++)
        if (tgt[i] >= 0 && !isnan(cnt[tgt[i]])) indeg[tgt[i]]++;
    size_t head = 0, tail = 0;
    for (size_t i = 0; i < n; i++)
        if (!isnan(cnt[i]) && indeg[i] == 0) fifo[tail++] = i;
    while (head < tail) {
        size_t i = fifo[head++];
        long long t = tgt[i];
        if (t < 0 || isnan(cnt[t])) continue;
        cnt[t] += cnt[i];
        if (--indeg[t] == 0) fifo[tail++] = (size_t)t;
    }
    for (size_t i = 0; i < n; i++) {
        if (ups_ncell) ups_ncell[i] = cnt[i];
        if (ups_area) ups_area[i] = cnt[i] * (resolution[i] * resolution[i]); /* x*y*1e6 with y in km2 */
    }
    free(tgt);
    free(indeg);
    free(cnt);
    free(fifo);
}

/* Multiple-flow-direction contributing area (Quinn et al. 1991; topmodel::topidx()$area of
 * R/upslope_area.R:21 is third-party code outside the reference tree: parity unpinned).  Serial
 * statement: cells in order of descending elevation; each takes its own res^2 plus, from every higher
 * neighbour (E, SE, S, SW, W, NW, N, NE), that neighbour's area times its weight towards this cell
 * over the neighbour's weight total. */
static double orc_mfd_weight(double zhi, double zlo, int q, double res)
{
    double dist = (q & 1) ? res * 1.4142135623730951 : res;
    double contour = ((q & 1) ? 0.354 : 0.5) * res;
    return ((zhi - zlo) / dist) * contour;
}
static const double* g_sort_z;
static int orc_cmp_desc(const void* a, const void* b)
{
    double za = g_sort_z[*(const size_t*)a], zb = g_sort_z[*(const size_t*)b];
    return (za < zb) - (za > zb);
}
void orc_mfd_area(int H, int W, const double* z, double res, double* area)
{
    static const int DR[8] = {0, 1, 1, 1, 0, -1, -1, -1}, DC[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    size_t n = (size_t)H * W, m = 0;
    double* wsum = (double*)calloc(n, sizeof(double));
    size_t* order = (size_t*)malloc(n * sizeof(size_t));
    for (size_t i = 0; i < n; i++) {
        area[i] = NAN;
        if (isnan(z[i])) continue;
        order[m++] = i;
        int r = (int)(i / W), c = (int)(i % W);
        for (int q = 0; q < 8; q++) {
            int rr = r + DR[q], cc = c + DC[q];
            if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
            double zn = z[(size_t)rr * W + cc];
            if (zn < z[i]) wsum[i] += orc_mfd_weight(z[i], zn, q, res);
        }
    }
    g_sort_z = z;
    qsort(order, m, sizeof(size_t), orc_cmp_desc);
    for (size_t k = 0; k < m; k++) {
        size_t i = order[k];
        int r = (int)(i / W), c = (int)(i % W);
        double a = res * res;
        for (int q = 0; q < 8; q++) {
            int rr = r + DR[q], cc = c + DC[q];
            if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
            size_t j = (size_t)rr * W + cc;
            if (z[j] > z[i]) a += area[j] * (orc_mfd_weight(z[j], z[i], (q + 4) & 7, res) / wsum[j]);
        }
        area[i] = a;
    }
    free(wsum);
    free(order);
}

/* ------------------------------------------------------------------ synthetic inputs */

static inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

double orc_synth_u(uint64_t seed, uint64_t stream, uint64_t cell, uint64_t day)
{
    uint64_t h = mix64(mix64(mix64(seed ^ stream) ^ cell) ^ day);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

enum {
    ST_ELEV = 1, ST_SLOP, ST_ASP, ST_AU, ST_NIN, ST_SAND, ST_CLAY, ST_OM, ST_GRAVEL, ST_BD,
    ST_DEPTH, ST_DEPTH2, ST_TC = 20, ST_SW, ST_ROCC, ST_RAMT
};

void orc_synth_statics(uint64_t seed, int H, int W, int64_t cell0, int ncell, int terrain,
                       double* lat, double* elev, double* slop, double* asp, double* resolution,
                       double* soil, double* au)
{
    (void)W;
    for (int i = 0; i < ncell; i++) {
        uint64_t c = (uint64_t)(cell0 + i);
        int64_t r = (cell0 + i) / W;
        lat[i] = 60.0 - 120.0 * ((double)r + 0.5) / (double)H;
        resolution[i] = 250.0;
        double Ai = 62500.0;
        if (terrain) {
            double u;
            elev[i] = 200.0 + 2800.0 * orc_synth_u(seed, ST_ELEV, c, 0);
            u = orc_synth_u(seed, ST_SLOP, c, 0);
            slop[i] = 30.0 * u * u;
            asp[i] = 360.0 * orc_synth_u(seed, ST_ASP, c, 0);
            au[i] = Ai * exp2(12.0 * orc_synth_u(seed, ST_AU, c, 0));
            au[(size_t)ncell + i] = 1.0 + floor(3.0 * orc_synth_u(seed, ST_NIN, c, 0));
            au[2 * (size_t)ncell + i] = 3.0;
        } else {
            elev[i] = 500.0;
            slop[i] = 0.0;
            asp[i] = 0.0;
            au[i] = 0.0;
            au[(size_t)ncell + i] = 3.0;
            au[2 * (size_t)ncell + i] = 3.0;
        }
        double sand = 15.0 + 55.0 * orc_synth_u(seed, ST_SAND, c, 0);
        double cmax = 95.0 - sand;
        if (cmax > 60.0) cmax = 60.0;
        double clay = 8.0 + (cmax - 8.0) * orc_synth_u(seed, ST_CLAY, c, 0);
        double om = 0.3 + 7.7 * orc_synth_u(seed, ST_OM, c, 0);
        double gravel = 40.0 * orc_synth_u(seed, ST_GRAVEL, c, 0);
        double bd = 0.9 + 0.7 * orc_synth_u(seed, ST_BD, c, 0);
        double depth = 0.6 + 1.4 * orc_synth_u(seed, ST_DEPTH, c, 0);
        if (orc_synth_u(seed, ST_DEPTH2, c, 0) < 0.1) depth = 2.0;
        soil[i] = sand;
        soil[(size_t)ncell + i] = clay;
        soil[2 * (size_t)ncell + i] = om;
        soil[3 * (size_t)ncell + i] = gravel;
        soil[4 * (size_t)ncell + i] = bd;
        soil[5 * (size_t)ncell + i] = depth;
    }
}

void orc_synth_forcing(uint64_t seed, int H, int W, int64_t cell0, int ncell, int t0, int nt,
                       const int* doy, const double* lat, const double* elev, double* sw_in,
                       double* tc, double* pn)
{
    (void)H;
    (void)W;
    const double TWO_PI = 6.283185307179586;
    for (int t = 0; t < nt; t++) {
        uint64_t day = (uint64_t)(t0 + t);
        double n = (double)doy[t];
        for (int i = 0; i < ncell; i++) {
            uint64_t c = (uint64_t)(cell0 + i);
            size_t k = (size_t)t * ncell + i;
            double la = lat[i];
            double ph = n - 15.0 - (la < 0.0 ? 182.0 : 0.0);
            double v = 25.0 * cos(la * G_pir) - 5.0 -
                       (5.0 + 0.25 * fabs(la)) * cos(TWO_PI * ph / 365.0) - 0.0065 * elev[i] +
                       3.0 * (orc_synth_u(seed, ST_TC, c, day) - 0.5) * 2.0;
            tc[k] = (double)(float)v;
            double decl = 23.44 * sin(TWO_PI * (n - 80.0) / 365.0);
            double s = 180.0 * cos((la - decl) * G_pir) *
                           (0.55 + 0.45 * orc_synth_u(seed, ST_SW, c, day)) +
                       10.0;
            if (s < 5.0) s = 5.0;
            if (s > 420.0) s = 420.0;
            sw_in[k] = (double)(float)s;
            double p = 0.0;
            if (orc_synth_u(seed, ST_ROCC, c, day) < 0.3)
                p = -6.0 * log(1.0 - orc_synth_u(seed, ST_RAMT, c, day));
            pn[k] = (double)(float)p;
        }
    }
}

static int is_leap(int y) { return (y % 4 == 0 && y % 100 != 0) || (y % 400 == 0); }

int orc_calendar(int year0, int nt, int* doy, int* year, int* month, int* month_id)
{
    static const int mdays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    int y = year0, m = 1, d = 1, n = 1, mid = 0;
    for (int t = 0; t < nt; t++) {
        doy[t] = n;
        year[t] = y;
        month[t] = m;
        month_id[t] = mid;
        int dim = mdays[m - 1] + ((m == 2 && is_leap(y)) ? 1 : 0);
        d++;
        n++;
        if (d > dim) {
            d = 1;
            m++;
            mid++;
            if (m > 12) {
                m = 1;
                y++;
                n = 1;
            }
        }
    }
    return nt > 0 ? month_id[nt - 1] + 1 : 0;
}
