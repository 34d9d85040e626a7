/*
 * splash_shim.c -- the `.Call` doorway an rsplash maintainer adds to the package's src/ so that
 * splash.grid()/splash.point() run on librsplash_hip.so (include/splash_hip.h).
 *
 * It replaces, for the gridded path, the Rcpp module registration of the reference
 * (src/SPLASH.cpp:12-21, src/RcppExports.cpp:9-25): instead of one R->C++ crossing per cell and
 * method (R/splash.point.R:92,148,152,158), one crossing per row block.
 *
 * There is no R in this repository's image: the file is syntax-checked against declarations-only headers of R's
 * public C API and, since round 6, COMPILED AND EXECUTED against a mock of that API (tests/csrc/r_mock/r_mock.c,
 * tests/test_abi.py::test_r_shim_runs_against_a_mock_r_api) -- never against R itself.  It is deliberately thin:
 * unpack SEXPs, allocate results, call the C ABI, raise an R error on a non-zero code.  Everything with logic in
 * it lives behind the C ABI, which the test-suite exercises.
 *
 * Build inside the package:   PKG_LIBS = -L<dir> -lrsplash_hip   (src/Makevars)
 */
#include <string.h>

#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#include "splash_hip.h"

/* sw_in, tc, pn: numeric matrices [ncell x nt] exactly as raster::getValues(brick, row, nrows)
 * returns them (R/splash.grid.R:278-280): column-major => time-major [nt][ncell], zero-copy.
 * stat: numeric matrix [ncell x 5] = lat, elev, slop, asp, resolution; soil [ncell x 6];
 * au [ncell x 3] or NULL; doy/year/month: integer(nt); monthly: logical(1).
 * Fills in/opt/out for one block and returns the (protected) list of result matrices. */
static SEXP grid_call_setup(SEXP sw_in, SEXP tc, SEXP pn, SEXP stat, SEXP soil, SEXP au, SEXP doy,
                            SEXP year, SEXP month, SEXP monthly, SEXP devices, splash_grid_in* in,
                            splash_opts* opt, splash_grid_out* out)
{
    SEXP dim = getAttrib(sw_in, R_DimSymbol);
    if (isNull(dim) || LENGTH(dim) != 2) error("sw_in must be a [ncell x nt] matrix");
    const size_t ncell = (size_t)INTEGER(dim)[0];
    const int nt = INTEGER(dim)[1];
    if (LENGTH(doy) != nt || LENGTH(year) != nt || LENGTH(month) != nt)
        error("doy/year/month must have one entry per day");
    if ((size_t)XLENGTH(tc) != ncell * nt || (size_t)XLENGTH(pn) != ncell * nt)
        error("tc and pn must have the dimensions of sw_in");

    if ((size_t)XLENGTH(stat) != 5 * ncell) error("stat must be a [ncell x 5] matrix");
    if ((size_t)XLENGTH(soil) != 6 * ncell) error("soil must be a [ncell x 6] matrix");
    if (!isNull(au) && (size_t)XLENGTH(au) != 3 * ncell) error("Au must be a [ncell x 3] matrix");
    if (!isInteger(devices) || LENGTH(devices) < 1) error("devices must be integer(>= 1)");

    memset(in, 0, sizeof *in);
    in->ncell = ncell;
    in->nt = nt;
    in->doy = INTEGER(doy);
    in->year = INTEGER(year);
    in->month = INTEGER(month);
    in->sw_in = REAL(sw_in);
    in->tc = REAL(tc);
    in->pn = REAL(pn);
    const double* st = REAL(stat); /* column-major [ncell x 5] == [5][ncell] */
    in->lat = st;
    in->elev = st + ncell;
    in->slop = st + 2 * ncell;
    in->asp = st + 3 * ncell;
    in->resolution = st + 4 * ncell;
    in->soil = REAL(soil);          /* [ncell x 6] == [6][ncell] */
    in->au = isNull(au) ? NULL : REAL(au);

    memset(opt, 0, sizeof *opt);
    opt->monthly_out = asLogical(monthly);
    if (INTEGER(devices)[0] < 0) {
        /* devices = -1L: no device -- SPLASH_BACKEND_HOST, the library's own day step compiled for the host, for a
         * block of a few cells (ABI 6; the default backend without a HIP device is an error, never a fallback) */
        opt->backend = SPLASH_BACKEND_HOST;
    } else {
        opt->device = INTEGER(devices)[0];
        opt->n_devices = LENGTH(devices);   /* > 1: the block is split over these GPUs inside the call */
        opt->devices = INTEGER(devices);
    }

    /* number of output steps: months present in the calendar, or days */
    int nmonth = 1;
    for (int t = 1; t < nt; t++)
        if (in->month[t] != in->month[t - 1] || in->year[t] != in->year[t - 1]) nmonth++;
    const int nt_out = opt->monthly_out ? nmonth : nt;

    static const char* names[] = {"wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr",
                                  "sm_lim", ""};
    SEXP res = PROTECT(mkNamed(VECSXP, names));
    memset(out, 0, sizeof *out); /* no state_end / diag / aux */
    for (int k = 0; k < SPLASH_NOUT; k++) {
        SEXP m = PROTECT(allocMatrix(REALSXP, (int)ncell, nt_out)); /* [ncell x nt_out] */
        out->out[k] = REAL(m);
        SET_VECTOR_ELT(res, k, m);
        UNPROTECT(1);
    }
    return res; /* still protected: the caller unprotects */
}

SEXP C_splash_grid_run(SEXP sw_in, SEXP tc, SEXP pn, SEXP stat, SEXP soil, SEXP au, SEXP doy,
                       SEXP year, SEXP month, SEXP monthly, SEXP devices)
{
    splash_grid_in in;
    splash_opts opt;
    splash_grid_out out;
    SEXP res = grid_call_setup(sw_in, tc, pn, stat, soil, au, doy, year, month, monthly, devices, &in, &opt, &out);
    char err[512];
    int rc = splash_grid_run(&in, &opt, &out, err, sizeof err);
    UNPROTECT(1);
    if (rc != SPLASH_OK) error("splash_grid_run failed (%d): %s", rc, err);
    return res;
}

/* The same call in two halves (splash_grid_submit / splash_grid_wait), so that splash.grid's loop
 * over row blocks keeps several blocks in flight, as the reference's receive loop keeps its cluster
 * busy (R/splash.grid.R:359-400).  _submit returns an external pointer; everything the running call
 * reads or writes (the argument matrices and the result list) hangs off that pointer's protected
 * slot, so R's collector leaves it alone until _wait has returned the results.
 * mem_fraction: this call's share of the device's free memory (0 = the library's default). */
typedef struct {
    splash_grid_job* job;
} grid_handle;

static void grid_handle_finalizer(SEXP ptr)
{
    grid_handle* h = (grid_handle*)R_ExternalPtrAddr(ptr);
    if (!h) return;
    if (h->job) { /* dropped without wait(): let the call finish before its buffers can go */
        char err[256];
        (void)splash_grid_wait(h->job, err, sizeof err);
    }
    R_Free(h);
    R_ClearExternalPtr(ptr);
}

SEXP C_splash_grid_submit(SEXP sw_in, SEXP tc, SEXP pn, SEXP stat, SEXP soil, SEXP au, SEXP doy,
                          SEXP year, SEXP month, SEXP monthly, SEXP devices, SEXP mem_fraction)
{
    splash_grid_in in;
    splash_opts opt;
    splash_grid_out out;
    SEXP res = grid_call_setup(sw_in, tc, pn, stat, soil, au, doy, year, month, monthly, devices, &in, &opt, &out);
    opt.mem_fraction = asReal(mem_fraction);
    /* keep-alive list: results + every argument the call reads */
    SEXP keep = PROTECT(allocVector(VECSXP, 11));
    SEXP args[10] = {sw_in, tc, pn, stat, soil, au, doy, year, month, devices};
    SET_VECTOR_ELT(keep, 0, res);
    for (int k = 0; k < 10; k++) SET_VECTOR_ELT(keep, k + 1, args[k]);
    grid_handle* h = R_Calloc(1, grid_handle);
    SEXP ptr = PROTECT(R_MakeExternalPtr(h, R_NilValue, keep));
    R_RegisterCFinalizerEx(ptr, grid_handle_finalizer, TRUE);
    char err[512];
    int rc = splash_grid_submit(&in, &opt, &out, &h->job, err, sizeof err);
    UNPROTECT(3);
    if (rc != SPLASH_OK) error("splash_grid_submit failed (%d): %s", rc, err);
    return ptr;
}

SEXP C_splash_grid_wait(SEXP ptr)
{
    grid_handle* h = (grid_handle*)R_ExternalPtrAddr(ptr);
    if (!h || !h->job) error("splash.grid: this block has been collected already");
    char err[512];
    int rc = splash_grid_wait(h->job, err, sizeof err);
    h->job = NULL;
    SEXP res = VECTOR_ELT(R_ExternalPtrProtected(ptr), 0);
    if (rc != SPLASH_OK) error("splash_grid_run failed (%d): %s", rc, err);
    return res;
}

/* A long series in time chunks (ABI 5: splash_grid_in.tt / state0 / ai0).  The module's run_all takes the
 * state to start from (src/SPLASH.h:93-97); this is that, for a block.  tt: numeric(ncell), the snowfall
 * threshold of the WHOLE series (pmax over C_splash_snow_threshold of every chunk; R/splash.point.R:120-122);
 * state0: NULL for the first chunk (it runs the spin-up), else the [ncell x 5] "state_end" of the chunk
 * before; ai0: NULL, else the aridity index the first chunk returned (column 2 of its "aux").
 * Returns the nine result matrices plus state_end [ncell x 5] and aux [ncell x 4] (Tt, AI, passes1, passes2). */
SEXP C_splash_grid_chunk(SEXP sw_in, SEXP tc, SEXP pn, SEXP stat, SEXP soil, SEXP au, SEXP doy,
                         SEXP year, SEXP month, SEXP monthly, SEXP devices, SEXP tt, SEXP state0, SEXP ai0)
{
    splash_grid_in in;
    splash_opts opt;
    splash_grid_out out;
    SEXP res9 = grid_call_setup(sw_in, tc, pn, stat, soil, au, doy, year, month, monthly, devices, &in, &opt, &out);
    const size_t ncell = in.ncell;
    if (!isReal(tt) || (size_t)XLENGTH(tt) != ncell) error("tt must be numeric(ncell)");
    if (isNull(state0) != isNull(ai0)) error("state0 and ai0 come together");
    if (!isNull(state0) && (!isReal(state0) || (size_t)XLENGTH(state0) != 5 * ncell)) error("state0 must be [ncell x 5]");
    if (!isNull(ai0) && (!isReal(ai0) || (size_t)XLENGTH(ai0) != ncell)) error("ai0 must be numeric(ncell)");
    in.tt = REAL(tt);
    in.state0 = isNull(state0) ? NULL : REAL(state0); /* [ncell x 5] column-major == [5][ncell] */
    in.ai0 = isNull(ai0) ? NULL : REAL(ai0);
    static const char* names[] = {"wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr",
                                  "sm_lim", "state_end", "aux", ""};
    SEXP res = PROTECT(mkNamed(VECSXP, names));
    for (int k = 0; k < SPLASH_NOUT; k++) SET_VECTOR_ELT(res, k, VECTOR_ELT(res9, k));
    SEXP se = PROTECT(allocMatrix(REALSXP, (int)ncell, 5));
    SEXP ax = PROTECT(allocMatrix(REALSXP, (int)ncell, 4));
    SET_VECTOR_ELT(res, SPLASH_NOUT, se);
    SET_VECTOR_ELT(res, SPLASH_NOUT + 1, ax);
    out.state_end = REAL(se);
    out.aux = REAL(ax);
    char err[512];
    int rc = splash_grid_run(&in, &opt, &out, err, sizeof err);
    UNPROTECT(4);
    if (rc != SPLASH_OK) error("splash_grid_run failed (%d): %s", rc, err);
    return res;
}

/* the snowfall threshold of one chunk of tc ([ncell x nt] matrix): numeric(ncell); backend 0 = HIP, 1 = host (ABI 6) */
SEXP C_splash_snow_threshold(SEXP tc, SEXP lat, SEXP elev, SEXP backend)
{
    SEXP dim = getAttrib(tc, R_DimSymbol);
    if (isNull(dim) || LENGTH(dim) != 2 || !isReal(tc)) error("tc must be a numeric [ncell x nt] matrix");
    const size_t ncell = (size_t)INTEGER(dim)[0];
    const int nt = INTEGER(dim)[1];
    if ((size_t)XLENGTH(lat) != ncell || (size_t)XLENGTH(elev) != ncell) error("lat and elev must be numeric(ncell)");
    SEXP res = PROTECT(allocVector(REALSXP, (R_xlen_t)ncell));
    char err[512];
    int rc = splash_snow_threshold_run_backend(asInteger(backend), ncell, nt, REAL(tc), NULL, REAL(lat), REAL(elev), REAL(res),
                                               err, sizeof err);
    UNPROTECT(1);
    if (rc != SPLASH_OK) error("splash_snow_threshold_run failed (%d): %s", rc, err);
    return res;
}

/* splash.point: the same pipeline for one site (numeric vectors of length nt) */
/* backend: 0 = SPLASH_BACKEND_HIP (default), 1 = SPLASH_BACKEND_HOST -- the library's own day step compiled for
 * the host, for a single site (ABI 6; include/splash_hip.h: splash_point_run_backend) */
SEXP C_splash_point_run(SEXP sw_in, SEXP tc, SEXP pn, SEXP lat, SEXP elev, SEXP slop, SEXP asp,
                        SEXP soil_data, SEXP Au, SEXP resolution, SEXP doy, SEXP year, SEXP month,
                        SEXP monthly, SEXP backend)
{
    const int nt = LENGTH(sw_in);
    if (LENGTH(tc) != nt || LENGTH(pn) != nt || LENGTH(doy) != nt) error("series lengths differ");
    if (LENGTH(soil_data) != 6) error("soil_data must have 6 values");
    int nmonth = 1;
    for (int t = 1; t < nt; t++)
        if (INTEGER(month)[t] != INTEGER(month)[t - 1] || INTEGER(year)[t] != INTEGER(year)[t - 1])
            nmonth++;
    const int mo = asLogical(monthly);
    const int nt_out = mo ? nmonth : nt;
    static const char* names[] = {"wn", "ro", "pet", "aet", "snow", "cond", "bflow", "netr",
                                  "sm_lim", ""};
    SEXP res = PROTECT(mkNamed(VECSXP, names));
    double* outp[SPLASH_NOUT];
    for (int k = 0; k < SPLASH_NOUT; k++) {
        SEXP v = PROTECT(allocVector(REALSXP, nt_out));
        outp[k] = REAL(v);
        SET_VECTOR_ELT(res, k, v);
        UNPROTECT(1);
    }
    char err[512];
    int rc = splash_point_run_backend(asInteger(backend), nt, INTEGER(doy), INTEGER(year), INTEGER(month),
                                      REAL(sw_in), REAL(tc), REAL(pn), asReal(lat), asReal(elev), asReal(slop),
                                      asReal(asp), REAL(soil_data), REAL(Au), LENGTH(Au), asReal(resolution), mo,
                                      outp, err, sizeof err);
    if (rc != SPLASH_OK) {
        UNPROTECT(1);
        error("splash_point_run failed (%d): %s", rc, err);
    }
    UNPROTECT(1);
    return res;
}

/* unSWC / unSWC.grid (R/unsSWC.R:14, R/unsSWC.grid.R:14): wn [ncell x nt] matrix (one column per
 * time step as getValues() gives it; one site = [1 x nt]), soil [ncell x 6], uns_depth numeric(1),
 * backend integer(1): SPLASH_BACKEND_HIP (0) or SPLASH_BACKEND_HOST (1, ABI 6: no GPU touched) */
SEXP C_splash_unswc_run(SEXP wn, SEXP soil, SEXP uns_depth, SEXP backend)
{
    SEXP dim = getAttrib(wn, R_DimSymbol);
    if (isNull(dim) || LENGTH(dim) != 2) error("wn must be a [ncell x nt] matrix");
    const size_t ncell = (size_t)INTEGER(dim)[0];
    const int nt = INTEGER(dim)[1];
    if ((size_t)XLENGTH(soil) != 6 * ncell)
        error("soil must be [ncell x 6]: %d cells in wn, %d soil values", (int)ncell, (int)XLENGTH(soil));
    static const char* names[] = {"wn_z", "wtd", "theta_i", ""};
    SEXP res = PROTECT(mkNamed(VECSXP, names));
    double* o[3];
    for (int k = 0; k < 3; k++) {
        SEXP m = PROTECT(allocMatrix(REALSXP, (int)ncell, nt));
        o[k] = REAL(m);
        SET_VECTOR_ELT(res, k, m);
        UNPROTECT(1);
    }
    char err[512];
    int rc = splash_unswc_run_backend(asInteger(backend), ncell, nt, REAL(wn), REAL(soil), asReal(uns_depth), o[0], o[1],
                                      o[2], err, sizeof err);
    UNPROTECT(1);
    if (rc != SPLASH_OK) error("splash_unswc_run failed (%d): %s", rc, err);
    return res;
}

/* unSWC.grid (R/unsSWC.grid.R:14-141): the gridded function's own semantics and its four rasters.
 * wn [ncell x nt] (getValues of the brick), soil [ncell x 6], uns_depth numeric(1), backend integer(1) */
SEXP C_splash_unswc_grid_run(SEXP wn, SEXP soil, SEXP uns_depth, SEXP backend)
{
    SEXP dim = getAttrib(wn, R_DimSymbol);
    if (isNull(dim) || LENGTH(dim) != 2) error("wn must be a [ncell x nt] matrix");
    const size_t ncell = (size_t)INTEGER(dim)[0];
    const int nt = INTEGER(dim)[1];
    if ((size_t)XLENGTH(soil) != 6 * ncell)
        error("soil must be [ncell x 6]: %d cells in wn, %d soil values", (int)ncell, (int)XLENGTH(soil));
    static const char* names[] = {"w_z", "wtd", "Se", "theta_mean", ""};
    SEXP res = PROTECT(mkNamed(VECSXP, names));
    double* o[4];
    for (int k = 0; k < 4; k++) {
        SEXP m = PROTECT(allocMatrix(REALSXP, (int)ncell, nt));
        o[k] = REAL(m);
        SET_VECTOR_ELT(res, k, m);
        UNPROTECT(1);
    }
    char err[512];
    int rc = splash_unswc_grid_run(asInteger(backend), ncell, nt, REAL(wn), REAL(soil), asReal(uns_depth), o[0], o[1], o[2],
                                   o[3], err, sizeof err);
    UNPROTECT(1);
    if (rc != SPLASH_OK) error("splash_unswc_grid_run failed (%d): %s", rc, err);
    return res;
}

/* The terrain of the reference's large-grid branch (gdaldem on the INT2S DEM, R/splash.grid.R:83-84): elev [W x H] as for
 * C_splash_static_rasters, xres / yres in degrees -> list(slope, aspect) of [W x H] matrices */
SEXP C_splash_terrain_gdal(SEXP elev, SEXP xres, SEXP yres, SEXP scale, SEXP round_int)
{
    SEXP dim = getAttrib(elev, R_DimSymbol);
    if (isNull(dim) || LENGTH(dim) != 2 || !isReal(elev)) error("elev must be a numeric [ncol x nrow] matrix");
    const int W = INTEGER(dim)[0], H = INTEGER(dim)[1];
    static const char* names[] = {"slope", "aspect", ""};
    SEXP res = PROTECT(mkNamed(VECSXP, names));
    double* o[2];
    for (int k = 0; k < 2; k++) {
        SEXP m = PROTECT(allocMatrix(REALSXP, W, H));
        o[k] = REAL(m);
        SET_VECTOR_ELT(res, k, m);
        UNPROTECT(1);
    }
    char err[512];
    int rc = splash_terrain_gdal_run(H, W, REAL(elev), asReal(xres), asReal(yres), asReal(scale), asInteger(round_int), o[0],
                                     o[1], err, sizeof err);
    UNPROTECT(1);
    if (rc != SPLASH_OK) error("splash_terrain_gdal_run failed (%d): %s", rc, err);
    return res;
}

/* Static rasters of an elevation matrix (R/splash.grid.R:96-110, R/upslope_area.R:12-58).
 * elev: numeric matrix [W x H], i.e. t(raster::as.matrix(dem)) or the vector raster::getValues(dem)
 * with dim c(ncol, nrow) -- row-major cells, as the C ABI wants them; ymax/yres/xres: northern edge
 * and cell size in degrees.  Returns a named list of [W x H] matrices in the same cell order. */
SEXP C_splash_static_rasters(SEXP elev, SEXP ymax, SEXP yres, SEXP xres, SEXP degree)
{
    SEXP dim = getAttrib(elev, R_DimSymbol);
    if (isNull(dim) || LENGTH(dim) != 2 || !isReal(elev)) error("elev must be a numeric [ncol x nrow] matrix");
    const int W = INTEGER(dim)[0], H = INTEGER(dim)[1];
    static const char* names[] = {"slope", "aspect", "lat", "resolution", "filled", "flowdir",
                                  "ups_ncell", "areacatch_d8", "ncellin", "ncellout", "areacatch", ""};
    SEXP res = PROTECT(mkNamed(VECSXP, names));
    double* p[11];
    for (int k = 0; k < 11; k++) {
        SEXP m = PROTECT(allocMatrix(REALSXP, W, H));
        SET_VECTOR_ELT(res, k, m);
        UNPROTECT(1);
        p[k] = REAL(m);
    }
    /* areacatch = the multiple-flow-direction area, as upslope_area()'s topidx; areacatch_d8 = count x
     * cell area, as upslope_areav2()'s aread8 */
    splash_static_out out = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10]};
    char msg[256] = "";
    const int rc = splash_static_rasters_run(H, W, REAL(elev), asReal(ymax), asReal(yres), asReal(xres),
                                             /* na_to_zero (R/splash.grid.R:107) */ 1, asReal(degree), &out,
                                             msg, sizeof msg);
    UNPROTECT(1);
    if (rc != 0) error("splash_static_rasters_run: %s (code %d)", msg, rc);
    return res;
}

static const R_CallMethodDef CallEntries[] = {
    {"C_splash_grid_run", (DL_FUNC)&C_splash_grid_run, 11},
    {"C_splash_grid_submit", (DL_FUNC)&C_splash_grid_submit, 12},
    {"C_splash_grid_wait", (DL_FUNC)&C_splash_grid_wait, 1},
    {"C_splash_grid_chunk", (DL_FUNC)&C_splash_grid_chunk, 14},
    {"C_splash_snow_threshold", (DL_FUNC)&C_splash_snow_threshold, 4},
    {"C_splash_point_run", (DL_FUNC)&C_splash_point_run, 15},
    {"C_splash_unswc_run", (DL_FUNC)&C_splash_unswc_run, 4},
    {"C_splash_unswc_grid_run", (DL_FUNC)&C_splash_unswc_grid_run, 4},
    {"C_splash_terrain_gdal", (DL_FUNC)&C_splash_terrain_gdal, 5},
    {"C_splash_static_rasters", (DL_FUNC)&C_splash_static_rasters, 5},
    {NULL, NULL, 0}};

/* add these two entries to the package's existing R_init_rsplash (src/RcppExports.cpp:21-24),
 * or use this initialiser if the Rcpp module is dropped */
void R_init_rsplash_hip(DllInfo* dll)
{
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
