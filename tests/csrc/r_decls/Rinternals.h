/* Rinternals.h -- DECLARATIONS ONLY (see R.h in this directory): the SEXP accessors, allocators and
 * protection calls r/splash_shim.c uses, with the remapped short names R offers by default. */
#ifndef SPLASH_TEST_RINTERNALS_H
#define SPLASH_TEST_RINTERNALS_H
#include <stddef.h>
#include "R.h"

typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
#define INTSXP 13
#define REALSXP 14
#define VECSXP 19

extern SEXP R_NilValue;
extern SEXP R_DimSymbol;

SEXP Rf_getAttrib(SEXP x, SEXP name);
Rboolean Rf_isNull(SEXP s);
Rboolean Rf_isInteger(SEXP s);
Rboolean Rf_isReal(SEXP s);
int Rf_asLogical(SEXP x);
int Rf_asInteger(SEXP x);
double Rf_asReal(SEXP x);
int LENGTH(SEXP x);
R_xlen_t XLENGTH(SEXP x);
int* INTEGER(SEXP x);
double* REAL(SEXP x);
SEXP Rf_mkNamed(SEXPTYPE type, const char** names);
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol);
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP Rf_protect(SEXP s);
void Rf_unprotect(int n);
void Rf_error(const char* fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));

typedef void (*R_CFinalizer_t)(SEXP);
SEXP R_MakeExternalPtr(void* p, SEXP tag, SEXP prot);
void* R_ExternalPtrAddr(SEXP s);
SEXP R_ExternalPtrProtected(SEXP s);
void R_ClearExternalPtr(SEXP s);
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit);

#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
#define getAttrib Rf_getAttrib
#define isNull Rf_isNull
#define isInteger Rf_isInteger
#define isReal Rf_isReal
#define asLogical Rf_asLogical
#define asInteger Rf_asInteger
#define asReal Rf_asReal
#define mkNamed Rf_mkNamed
#define allocMatrix Rf_allocMatrix
#define allocVector Rf_allocVector
#define error Rf_error
#endif
