/* R.h -- DECLARATIONS ONLY, written for this repository's syntax check of r/splash_shim.c
 * (tests/test_abi.py::test_r_shim_syntax).  It is not R's header and builds nothing: it declares the
 * part of R's public C API the shim uses (names, arities and argument types as documented in
 * "Writing R Extensions", sections 5.9 "Handling R objects in C" and 6 "The R API"), so that gcc
 * -fsyntax-only catches typos, wrong arities and wrong argument types in the shim.  The shim is
 * syntax-checked, never run, in this image (no R here). */
#ifndef SPLASH_TEST_R_H
#define SPLASH_TEST_R_H
#include <stddef.h>

typedef enum { FALSE = 0, TRUE } Rboolean;
void* R_chk_calloc(size_t nelem, size_t elsize);
void R_chk_free(void* p);
#define R_Calloc(n, t) ((t*)R_chk_calloc((size_t)(n), sizeof(t)))
#define R_Free(p) (R_chk_free((void*)(p)), (p) = NULL)
#endif
