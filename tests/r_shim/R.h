/* syntax-check shim, see README */
#ifndef PSD_TEST_R_SHIM_R_H
#define PSD_TEST_R_SHIM_R_H
#include <stddef.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif
void Rprintf(const char *, ...);
void Rf_error(const char *, ...) __attribute__((noreturn));
char *R_alloc(size_t, int);
#ifdef __cplusplus
}
#endif
#endif
