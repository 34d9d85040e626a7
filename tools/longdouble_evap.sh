#!/bin/bash
# SOLAR + EVAP of the oracle (oracle/splash_oracle.c, the statements of src/SOLAR.cpp and
# src/EVAP.cpp) evaluated in long double on the rows of the known-answer file that decide the
# bounds of tests/test_gpu_parity.py::test_evap_vs_reference_kat -- how far the reference's own
# doubles are from a 64-bit-mantissa evaluation of the same formulas (conditioning, not error).
#   bash tools/longdouble_evap.sh        (CPU only; prints aet, pet, rn_d per row)
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
awk '/^void orc_evap_batch/{exit} {print}' oracle/splash_oracle.c | sed "s|#include \"splash_oracle.h\"|#include \"$PWD/oracle/splash_oracle.h\"|" > $T/part.c
cat > $T/drv.c <<'C'
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <float.h>
#include <omp.h>
#include <math.h>
typedef long double ld_t;
#define double ld_t
#define sin sinl
#define cos cosl
#define tan tanl
#define acos acosl
#define asin asinl
#define atan atanl
#define atan2 atan2l
#define pow powl
#define exp expl
#define log logl
#define sqrt sqrtl
#define fabs fabsl
#define floor floorl
#define ceil ceill
#define fmod fmodl
#define fmax fmaxl
#define fmin fminl
#include "part.c"
#undef double
int main(void)
{
    /* lat, elv, sw, n, y, sw_in, tc, slop, asp, snow, nd: rows 4431, 5655, 2 more of ref_evap_kat.npz */
    long double r[][11] = {
        {63.43765976927031L, 1368.4299728919991L, 0.06092354624931668L, 344, 1999, 84.6809382640499L, -30.702655512495745L, 30.81773106663432L, -95.93399871329409L, 0.0L, 20.0L},
        {65.83613376801185L, 313.83161554518426L, 0.936186891597624L, 219, 2001, 243.5465693494308L, 36.72932330827068L, 57.16616603141703L, -123.90721773317476L, 0.0L, 38.0L},
        {-55.67998987200062L, 1947.4900034609072L, 0.2455303202285296L, 290, 2100, 84.24056031765168L, -6.297809864019328L, 50.49496337817509L, -39.10142854705529L, 0.0L, 23.0L},
        {69.57550403888797L, 4297.873247155524L, 0.11126566921804515L, 344, 0, 256.30882670791925L, -35.072742962084845L, 20.703283325896628L, -85.22199160170393L, 339.95729482008386L, 36.0L}};
    for (int i = 0; i < 4; i++) {
        orc_etr e;
        orc_evap(r[i][0], r[i][1], r[i][2], (int)r[i][3], (int)r[i][4], r[i][5], r[i][6], r[i][7], r[i][8], r[i][9], r[i][10], &e);
        printf("aet %.18Lg pet %.18Lg rn_d %.18Lg\n", e.aet, e.pet, e.rn_d);
    }
    return 0;
}
C
gcc -O1 -std=gnu11 -fopenmp -I$T -o $T/drv $T/drv.c -lm
$T/drv
rm -rf $T
