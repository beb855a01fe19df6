/* syntax-check shim, see README */
#ifndef PSD_TEST_R_SHIM_RDYNLOAD_H
#define PSD_TEST_R_SHIM_RDYNLOAD_H
#ifdef __cplusplus
extern "C" {
#endif
typedef void *(*DL_FUNC)(void);
typedef unsigned int R_NativePrimitiveArgType;
typedef struct {
  const char *name;
  DL_FUNC fun;
  int numArgs;
  R_NativePrimitiveArgType *types;
} R_CMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *, const R_CMethodDef *, const void *, const void *, const void *);
int R_useDynamicSymbols(DllInfo *, int);
#define FALSE 0
#ifdef __cplusplus
}
#endif
#endif
