/* r_mock.c -- a MOCK of the part of R's C API that r/splash_shim.c uses, and a driver that calls the shim's
 * .Call entry points through its own registration table.  TEST INFRASTRUCTURE: there is no R in this image, so the shim
 * had only ever been syntax-checked; with this file it is COMPILED AND EXECUTED (tests/test_abi.py::
 * test_r_shim_runs_against_a_mock_r_api): SEXP unpacking, dimension checks, result allocation, PROTECT balance,
 * registration arities, the external-pointer handle of submit/wait, the error path.  What it cannot show is that real R
 * behaves like this mock -- the mock follows "Writing R Extensions" 5.9 / 5.4 / 6 for the ~30 functions declared in
 * tests/csrc/r_decls/ (our own declarations; nothing of R or of the reference is copied or stood in for).
 * The calls use SPLASH_BACKEND_HOST (devices = -1 / backend = 1), so the whole thing runs without a GPU.
 *
 *   r_mock in.bin out.bin
 * in.bin  (doubles): nt, ncell, nt_cut, then site: lat, elev, slop, asp, resolution, au, soil[6], doy[nt], year[nt],
 *          month[nt], sw[nt], tc[nt], pn[nt] (the point), then the grid: stat[5][ncell], soil[6][ncell], au[3][ncell],
 *          tt[ncell], sw[nt][ncell], tc[nt][ncell], pn[nt][ncell]
 * out.bin (doubles): point 9 x nt | grid_run 9 x nt x ncell | chunks 9 x nt x ncell (two chunks concatenated in time) |
 *          state_end of the second chunk 5 x ncell
 * exit code 0 = every check of the driver held (PROTECT balance, submit/wait == run bitwise, errors raised where due). */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#include "splash_hip.h"

#define LGLSXP 10
#define EXTPTRSXP 22
#define NILSXP 0

struct SEXPREC {
    SEXPTYPE type;
    R_xlen_t n;
    void* data;          /* int* / double* / SEXP* */
    SEXP dim;            /* the "dim" attribute or NULL */
    const char** names;  /* mkNamed */
    void* ext;           /* external pointer: address, protected slot, finalizer */
    SEXP prot;
    R_CFinalizer_t fin;
};

static struct SEXPREC nil_rec = {NILSXP, 0, NULL, NULL, NULL, NULL, NULL, NULL};
static struct SEXPREC dimsym_rec = {NILSXP, 0, NULL, NULL, NULL, NULL, NULL, NULL};
SEXP R_NilValue = &nil_rec;
SEXP R_DimSymbol = &dimsym_rec;

static int g_protect = 0, g_protect_max = 0;
static jmp_buf g_on_error;
static int g_error_armed = 0;
static char g_error_msg[1024];

static SEXP new_sexp(SEXPTYPE type, R_xlen_t n)
{
    SEXP s = (SEXP)calloc(1, sizeof(struct SEXPREC));
    s->type = type;
    s->n = n;
    size_t es = type == REALSXP ? sizeof(double) : (type == VECSXP ? sizeof(SEXP) : sizeof(int));
    s->data = calloc((size_t)(n > 0 ? n : 1), es);
    if (type == VECSXP)
        for (R_xlen_t i = 0; i < n; i++) ((SEXP*)s->data)[i] = R_NilValue;
    return s;
}

void* R_chk_calloc(size_t nelem, size_t elsize) { return calloc(nelem, elsize); }
void R_chk_free(void* p) { free(p); }
SEXP Rf_getAttrib(SEXP x, SEXP name) { return (name == R_DimSymbol && x->dim) ? x->dim : R_NilValue; }
Rboolean Rf_isNull(SEXP s) { return s == R_NilValue || s->type == NILSXP ? TRUE : FALSE; }
Rboolean Rf_isInteger(SEXP s) { return s->type == INTSXP ? TRUE : FALSE; }
Rboolean Rf_isReal(SEXP s) { return s->type == REALSXP ? TRUE : FALSE; }
int Rf_asLogical(SEXP x) { return x->type == REALSXP ? ((double*)x->data)[0] != 0.0 : ((int*)x->data)[0] != 0; }
int Rf_asInteger(SEXP x) { return x->type == REALSXP ? (int)((double*)x->data)[0] : ((int*)x->data)[0]; }
double Rf_asReal(SEXP x) { return x->type == REALSXP ? ((double*)x->data)[0] : (double)((int*)x->data)[0]; }
int LENGTH(SEXP x) { return (int)x->n; }
R_xlen_t XLENGTH(SEXP x) { return x->n; }
int* INTEGER(SEXP x)
{
    if (x->type != INTSXP && x->type != LGLSXP) { fprintf(stderr, "mock: INTEGER() of a non-integer\n"); exit(4); }
    return (int*)x->data;
}
double* REAL(SEXP x)
{
    if (x->type != REALSXP) { fprintf(stderr, "mock: REAL() of a non-numeric\n"); exit(4); }
    return (double*)x->data;
}
SEXP Rf_mkNamed(SEXPTYPE type, const char** names)
{
    R_xlen_t n = 0;
    while (names[n][0]) n++;
    SEXP s = new_sexp(type, n);
    s->names = names;
    return s;
}
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n) { return new_sexp(type, n); }
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol)
{
    SEXP s = new_sexp(type, (R_xlen_t)nrow * ncol);
    s->dim = new_sexp(INTSXP, 2);
    ((int*)s->dim->data)[0] = nrow;
    ((int*)s->dim->data)[1] = ncol;
    return s;
}
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v)
{
    if (x->type != VECSXP || i < 0 || i >= x->n) { fprintf(stderr, "mock: SET_VECTOR_ELT out of range\n"); exit(4); }
    ((SEXP*)x->data)[i] = v;
    return v;
}
SEXP VECTOR_ELT(SEXP x, R_xlen_t i)
{
    if (x->type != VECSXP || i < 0 || i >= x->n) { fprintf(stderr, "mock: VECTOR_ELT out of range\n"); exit(4); }
    return ((SEXP*)x->data)[i];
}
SEXP Rf_protect(SEXP s)
{
    if (++g_protect > g_protect_max) g_protect_max = g_protect;
    return s;
}
void Rf_unprotect(int n)
{
    g_protect -= n;
    if (g_protect < 0) { fprintf(stderr, "mock: UNPROTECT below the stack\n"); exit(4); }
}
void Rf_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error_msg, sizeof g_error_msg, fmt, ap);
    va_end(ap);
    if (!g_error_armed) { fprintf(stderr, "mock: unexpected R error: %s\n", g_error_msg); exit(3); }
    g_protect = 0; /* R unwinds the protection stack on an error */
    longjmp(g_on_error, 1);
}
SEXP R_MakeExternalPtr(void* p, SEXP tag, SEXP prot)
{
    (void)tag;
    SEXP s = new_sexp(EXTPTRSXP, 0);
    s->ext = p;
    s->prot = prot;
    return s;
}
void* R_ExternalPtrAddr(SEXP s) { return s->ext; }
SEXP R_ExternalPtrProtected(SEXP s) { return s->prot; }
void R_ClearExternalPtr(SEXP s) { s->ext = NULL; }
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit) { (void)onexit; s->fin = fun; }

static const R_CallMethodDef* g_calls = NULL;
int R_registerRoutines(DllInfo* info, const R_CMethodDef* const c, const R_CallMethodDef* const call,
                       const R_FortranMethodDef* const f, const R_ExternalMethodDef* const e)
{
    (void)info; (void)c; (void)f; (void)e;
    g_calls = call;
    return 1;
}
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value) { (void)info; return value; }

/* ---- the driver: .Call(name, ...) through the registration table -------------------------------------------- */
void R_init_rsplash_hip(DllInfo* dll);

static SEXP dot_call(const char* name, int nargs, SEXP* a)
{
    const R_CallMethodDef* d = g_calls;
    for (; d && d->name; d++)
        if (!strcmp(d->name, name)) break;
    if (!d || !d->name) { fprintf(stderr, "mock: %s is not registered\n", name); exit(5); }
    if (d->numArgs != nargs) { fprintf(stderr, "mock: %s registered with %d arguments, called with %d\n", name, d->numArgs, nargs); exit(5); }
    typedef SEXP (*F1)(SEXP);
    typedef SEXP (*F3)(SEXP, SEXP, SEXP);
    typedef SEXP (*F4)(SEXP, SEXP, SEXP, SEXP);
    typedef SEXP (*F11)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
    typedef SEXP (*F12)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
    typedef SEXP (*F14)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
    typedef SEXP (*F15)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
    const int before = g_protect;
    SEXP r = R_NilValue;
    switch (nargs) {
    case 1: r = ((F1)d->fun)(a[0]); break;
    case 3: r = ((F3)d->fun)(a[0], a[1], a[2]); break;
    case 4: r = ((F4)d->fun)(a[0], a[1], a[2], a[3]); break;
    case 11: r = ((F11)d->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10]); break;
    case 12: r = ((F12)d->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11]); break;
    case 14: r = ((F14)d->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]); break;
    case 15: r = ((F15)d->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14]); break;
    default: fprintf(stderr, "mock: no trampoline for %d arguments\n", nargs); exit(5);
    }
    if (g_protect != before) { fprintf(stderr, "mock: %s left the protection stack at %d (was %d)\n", name, g_protect, before); exit(6); }
    return r;
}

static SEXP real_vec(const double* v, R_xlen_t n)
{
    SEXP s = new_sexp(REALSXP, n);
    memcpy(s->data, v, (size_t)n * sizeof(double));
    return s;
}
static SEXP real_mat(const double* v, int nrow, int ncol)
{
    SEXP s = Rf_allocMatrix(REALSXP, nrow, ncol);
    memcpy(s->data, v, (size_t)nrow * ncol * sizeof(double));
    return s;
}
static SEXP int_vec_from(const double* v, R_xlen_t n)
{
    SEXP s = new_sexp(INTSXP, n);
    for (R_xlen_t i = 0; i < n; i++) ((int*)s->data)[i] = (int)v[i];
    return s;
}
static SEXP scalar(SEXPTYPE t, double v)
{
    SEXP s = new_sexp(t, 1);
    if (t == REALSXP) ((double*)s->data)[0] = v; else ((int*)s->data)[0] = (int)v;
    return s;
}
/* expects the call to raise an R error whose message contains `needle` */
static int expect_error(const char* name, int nargs, SEXP* a, const char* needle)
{
    g_error_armed = 1;
    if (setjmp(g_on_error) == 0) {
        (void)dot_call(name, nargs, a);
        g_error_armed = 0;
        fprintf(stderr, "mock: %s did not raise an error (expected one with \"%s\")\n", name, needle);
        return 1;
    }
    g_error_armed = 0;
    if (!strstr(g_error_msg, needle)) {
        fprintf(stderr, "mock: %s raised \"%s\", expected one with \"%s\"\n", name, g_error_msg, needle);
        return 1;
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const size_t nd = (size_t)ftell(f) / sizeof(double);
    fseek(f, 0, SEEK_SET);
    double* in = (double*)malloc(nd * sizeof(double));
    if (fread(in, sizeof(double), nd, f) != nd) return 2;
    fclose(f);
    const int nt = (int)in[0], ncell = (int)in[1], cut = (int)in[2];
    const double* p = in + 3;
    const double *site = p, *soil1 = p + 6;
    p += 12;
    const double *doy = p, *year = p + nt, *month = p + 2 * nt, *sw1 = p + 3 * nt, *tc1 = p + 4 * nt, *pn1 = p + 5 * nt;
    p += 6 * (size_t)nt;
    const double *stat = p, *soilg = p + 5 * (size_t)ncell, *aug = p + 11 * (size_t)ncell, *tt = p + 14 * (size_t)ncell;
    p += 15 * (size_t)ncell;
    const size_t ng = (size_t)nt * ncell;
    const double *swg = p, *tcg = p + ng, *png = p + 2 * ng;
    if ((size_t)(p + 3 * ng - in) != nd) { fprintf(stderr, "mock: input size mismatch\n"); return 2; }

    R_init_rsplash_hip(NULL);
    int bad = 0;
    FILE* o = fopen(argv[2], "wb");

    /* 1. splash.point on the host backend: .Call("C_splash_point_run", ..., monthly_out, backend) */
    SEXP a[15] = {real_vec(sw1, nt), real_vec(tc1, nt), real_vec(pn1, nt), scalar(REALSXP, site[0]), scalar(REALSXP, site[1]),
                  scalar(REALSXP, site[2]), scalar(REALSXP, site[3]), real_vec(soil1, 6), scalar(REALSXP, site[5]),
                  scalar(REALSXP, site[4]), int_vec_from(doy, nt), int_vec_from(year, nt), int_vec_from(month, nt),
                  scalar(LGLSXP, 0), scalar(INTSXP, 1)};
    SEXP r = dot_call("C_splash_point_run", 15, a);
    if (LENGTH(r) != 9 || strcmp(r->names[0], "wn") || strcmp(r->names[8], "sm_lim")) { fprintf(stderr, "mock: point result list\n"); bad++; }
    for (int k = 0; k < 9; k++) fwrite(REAL(VECTOR_ELT(r, k)), sizeof(double), (size_t)nt, o);
    /* a soil vector of the wrong length is an R error, not a crash */
    SEXP a_bad[15];
    memcpy(a_bad, a, sizeof a);
    a_bad[7] = real_vec(soil1, 5);
    bad += expect_error("C_splash_point_run", 15, a_bad, "soil_data must have 6 values");
    /* the default backend without a device is an R error that says so */
    SEXP a_hip[15];
    memcpy(a_hip, a, sizeof a);
    a_hip[14] = scalar(INTSXP, 0);
    if (splash_hip_device_count() == 0) bad += expect_error("C_splash_point_run", 15, a_hip, "no HIP device");

    /* 2. a row block through C_splash_grid_run; devices = -1L selects the host backend */
    SEXP g[12] = {real_mat(swg, ncell, nt), real_mat(tcg, ncell, nt), real_mat(png, ncell, nt), real_mat(stat, ncell, 5),
                  real_mat(soilg, ncell, 6), real_mat(aug, ncell, 3), int_vec_from(doy, nt), int_vec_from(year, nt),
                  int_vec_from(month, nt), scalar(LGLSXP, 0), scalar(INTSXP, -1), scalar(REALSXP, 0.0)};
    SEXP rg = dot_call("C_splash_grid_run", 11, g);
    for (int k = 0; k < 9; k++) {
        SEXP m = VECTOR_ELT(rg, k);
        SEXP dim = Rf_getAttrib(m, R_DimSymbol);
        if (INTEGER(dim)[0] != ncell || INTEGER(dim)[1] != nt) { fprintf(stderr, "mock: grid result dims\n"); bad++; }
        fwrite(REAL(m), sizeof(double), ng, o);
    }
    /* 3. the same block in two halves: submit returns a handle at once, wait returns the list; bitwise the one-piece call;
     *    collecting twice is an R error; the finalizer of a collected handle frees it */
    SEXP h = dot_call("C_splash_grid_submit", 12, g);
    if (h->type != EXTPTRSXP || !h->fin) { fprintf(stderr, "mock: submit did not return a finalizable external pointer\n"); bad++; }
    SEXP rs = dot_call("C_splash_grid_wait", 1, &h);
    for (int k = 0; k < 9; k++)
        if (memcmp(REAL(VECTOR_ELT(rs, k)), REAL(VECTOR_ELT(rg, k)), ng * sizeof(double))) { fprintf(stderr, "mock: submit/wait differs from run in series %d\n", k); bad++; }
    bad += expect_error("C_splash_grid_wait", 1, &h, "collected already");
    h->fin(h);
    if (R_ExternalPtrAddr(h) != NULL) { fprintf(stderr, "mock: finalizer left the handle\n"); bad++; }
    /* a handle dropped without wait(): the finalizer waits for the job */
    SEXP h2 = dot_call("C_splash_grid_submit", 12, g);
    h2->fin(h2);
    /* wrong matrix dimensions are R errors */
    SEXP g_bad[12];
    memcpy(g_bad, g, sizeof g);
    g_bad[3] = real_mat(stat, ncell, 4);
    bad += expect_error("C_splash_grid_run", 11, g_bad, "stat must be a [ncell x 5] matrix");

    /* 4. the series in two time chunks (ABI 5): C_splash_grid_chunk with tt, then state0 / ai0 of the chunk before */
    const size_t n1 = (size_t)cut * ncell, n2 = (size_t)(nt - cut) * ncell;
    SEXP c1[14] = {real_mat(swg, ncell, cut), real_mat(tcg, ncell, cut), real_mat(png, ncell, cut), g[3], g[4], g[5],
                   int_vec_from(doy, cut), int_vec_from(year, cut), int_vec_from(month, cut), scalar(LGLSXP, 0),
                   scalar(INTSXP, -1), real_vec(tt, ncell), R_NilValue, R_NilValue};
    SEXP r1 = dot_call("C_splash_grid_chunk", 14, c1);
    if (LENGTH(r1) != 11 || strcmp(r1->names[9], "state_end") || strcmp(r1->names[10], "aux")) { fprintf(stderr, "mock: chunk result list\n"); bad++; }
    SEXP ai0 = real_vec(REAL(VECTOR_ELT(r1, 10)) + ncell, ncell); /* aux[, 2]: column-major [ncell x 4] */
    SEXP c2[14] = {real_mat(swg + n1, ncell, nt - cut), real_mat(tcg + n1, ncell, nt - cut), real_mat(png + n1, ncell, nt - cut),
                   g[3], g[4], g[5], int_vec_from(doy + cut, nt - cut), int_vec_from(year + cut, nt - cut),
                   int_vec_from(month + cut, nt - cut), scalar(LGLSXP, 0), scalar(INTSXP, -1), real_vec(tt, ncell),
                   VECTOR_ELT(r1, 9), ai0};
    SEXP r2 = dot_call("C_splash_grid_chunk", 14, c2);
    for (int k = 0; k < 9; k++) {
        fwrite(REAL(VECTOR_ELT(r1, k)), sizeof(double), n1, o);
        fwrite(REAL(VECTOR_ELT(r2, k)), sizeof(double), n2, o);
    }
    fwrite(REAL(VECTOR_ELT(r2, 9)), sizeof(double), (size_t)5 * ncell, o);
    SEXP c_bad[14];
    memcpy(c_bad, c2, sizeof c2);
    c_bad[13] = R_NilValue; /* state0 without ai0 */
    bad += expect_error("C_splash_grid_chunk", 14, c_bad, "state0 and ai0 come together");
    /* 5. unSWC on the host backend (the README example's second step): wn [1 x nt] of step 1, soil [1 x 6], 0.49 m */
    SEXP wn_m = real_mat(REAL(VECTOR_ELT(r, 0)), 1, nt);
    SEXP soil_m = real_mat(soil1, 1, 6);
    SEXP u[4] = {wn_m, soil_m, scalar(REALSXP, 0.49), scalar(INTSXP, 1)};
    SEXP ru = dot_call("C_splash_unswc_run", 4, u);
    if (LENGTH(ru) != 3 || strcmp(ru->names[0], "wn_z") || strcmp(ru->names[2], "theta_i")) { fprintf(stderr, "mock: unswc result list\n"); bad++; }
    for (int k = 0; k < 3; k++) fwrite(REAL(VECTOR_ELT(ru, k)), sizeof(double), (size_t)nt, o);
    SEXP u_bad[4] = {wn_m, real_mat(soil1, 1, 5), u[2], u[3]};
    bad += expect_error("C_splash_unswc_run", 4, u_bad, "soil must be [ncell x 6]");
    /* 6. unSWC.grid on the block's wn brick ([ncell x nt] as getValues() gives it), host backend: its four rasters */
    SEXP ug[4] = {VECTOR_ELT(rg, 0), g[4], scalar(REALSXP, 0.3), scalar(INTSXP, 1)};
    SEXP rug = dot_call("C_splash_unswc_grid_run", 4, ug);
    if (LENGTH(rug) != 4 || strcmp(rug->names[0], "w_z") || strcmp(rug->names[2], "Se") || strcmp(rug->names[3], "theta_mean")) { fprintf(stderr, "mock: unswc_grid result list\n"); bad++; }
    for (int k = 0; k < 4; k++) fwrite(REAL(VECTOR_ELT(rug, k)), sizeof(double), ng, o);
    fclose(o);
    if (g_protect != 0) { fprintf(stderr, "mock: protection stack at %d at exit\n", g_protect); bad++; }
    printf("r_mock: %d problems; deepest protection stack %d\n", bad, g_protect_max);
    return bad ? 1 : 0;
}
