/* syntax-check shim, see README */
#ifndef PSD_TEST_R_SHIM_RINTERNALS_H
#define PSD_TEST_R_SHIM_RINTERNALS_H
#define INTSXP 13
#define STRSXP 16
#ifndef NULL
#define NULL 0
#endif
#endif
