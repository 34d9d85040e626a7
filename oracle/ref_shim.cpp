// ref_shim.cpp -- extern "C" doorway into the REAL reference classes (TEST INFRASTRUCTURE ONLY).
//
// Compiled together with /root/reference/src/{global,SOLAR,EVAP}.cpp (where they lie; never
// copied) into oracle/_ref/libsplash_ref.so by oracle/Makefile.  Those three files build from
// their own sources.  src/SPLASH.cpp is NOT part of this build: it includes <Rcpp.h>, which the
// image lacks, and a reference build may not be made by writing stand-in headers -- so the
// water-balance step has no executable reference here (DESIGN.md "Oracle").
//
// Only the public API of the reference classes is used (src/SOLAR.h:86-94, src/EVAP.h:99-108).
#include "EVAP.h"
#include "SOLAR.h"

extern "C" {

// SOLAR::calculate_daily_fluxes + get_vals (src/SOLAR.cpp:77-267, 376-393)
void ref_solar(double lat, double elv, int n, int y, double sw_in, double tc, double slop,
               double asp, double snow, double nd, double sw, double* out8)
{
    SOLAR s(lat, elv);
    s.calculate_daily_fluxes(n, y, sw_in, tc, slop, asp, snow, nd, sw);
    srad v = s.get_vals();
    out8[0] = v.ru;
    out8[1] = v.rv;
    out8[2] = v.rw;
    out8[3] = v.rnl;
    out8[4] = v.hn;
    out8[5] = v.rn_d;
    out8[6] = v.rnn_d;
    out8[7] = v.ts;
}

// EVAP::calculate_daily_fluxes + get_vals (src/EVAP.cpp:82-264, 517-537)
void ref_evap(double lat, double elv, double sw, int n, int y, double sw_in, double tc,
              double slop, double asp, double snow, double nd, double* out11)
{
    EVAP e(lat, elv);
    e.calculate_daily_fluxes(sw, n, y, sw_in, tc, slop, asp, snow, nd);
    etr v = e.get_vals();
    out11[0] = v.cond;
    out11[1] = v.eet;
    out11[2] = v.pet;
    out11[3] = v.aet;
    out11[4] = v.snowmelt;
    out11[5] = v.sublimation;
    out11[6] = v.econ;
    out11[7] = v.pw;
    out11[8] = v.rn_d;
    out11[9] = v.visc;
    out11[10] = v.pet_max;
}

// batched form (same call per row) so python can sweep millions of cases quickly
void ref_evap_batch(int m, const double* lat, const double* elv, const double* sw, const int* n,
                    const int* y, const double* sw_in, const double* tc, const double* slop,
                    const double* asp, const double* snow, const double* nd, double* out /*[m][11]*/)
{
    for (int i = 0; i < m; i++)
        ref_evap(lat[i], elv[i], sw[i], n[i], y[i], sw_in[i], tc[i], slop[i], asp[i], snow[i],
                 nd[i], out + (long)i * 11);
}

void ref_solar_batch(int m, const double* lat, const double* elv, const int* n, const int* y,
                     const double* sw_in, const double* tc, const double* slop, const double* asp,
                     const double* snow, const double* nd, const double* sw, double* out /*[m][8]*/)
{
    for (int i = 0; i < m; i++)
        ref_solar(lat[i], elv[i], n[i], y[i], sw_in[i], tc[i], slop[i], asp[i], snow[i], nd[i],
                  sw[i], out + (long)i * 8);
}
}
