/* R_ext/Rdynload.h -- DECLARATIONS ONLY (see ../R.h): native routine registration
 * ("Writing R Extensions" 5.4 "Registering native routines"). */
#ifndef SPLASH_TEST_RDYNLOAD_H
#define SPLASH_TEST_RDYNLOAD_H
#include "../R.h"

typedef void* (*DL_FUNC)(void);
typedef struct {
    const char* name;
    DL_FUNC fun;
    int numArgs;
} R_CallMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct {
    const char* name;
    DL_FUNC fun;
    int numArgs;
    void* types;
} R_CMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo* info, const R_CMethodDef* const croutines, const R_CallMethodDef* const callRoutines,
                       const R_FortranMethodDef* const fortranRoutines, const R_ExternalMethodDef* const externalRoutines);
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value);
#endif
