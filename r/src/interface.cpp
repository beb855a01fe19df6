/* interface.cpp -- R glue for libpeaksegdisk_hip.so.
 *
 * Takes the place of /root/reference/src/interface.cpp:10-73 in the PeakSegDisk package when
 * the solver sources (src/PeakSegFPOPLog.cpp, src/funPieceListLog.cpp) are replaced by the
 * MI355X library: the same registered `.C` routine (three STRSXP arguments, element 0 of each
 * used), the same Rf_error texts for every status code, the same R_init_PeakSegDisk.  Two
 * additive routines expose the batch forms (penalty grids, cached directory batches and the
 * resident penalty search); nothing in R/ has to change to keep using the first one.
 *
 * Built by R CMD INSTALL with src/Makevars next to this file (needs R.h; R is not part of the
 * image this repository is developed in, where tests/test_r_glue.py only syntax-checks it).
 */
#include "peaksegdisk_hip.h"

#include <R.h>
#include <R_ext/Rdynload.h>
#include <Rinternals.h>

extern "C" {

static void to_r_console(const char *text) { Rprintf("%s", text); }

/* .C("PeakSegFPOP_interface", bedGraph.file, penalty, db.file) -- R/PeakSegFPOP_file.R:66-71 */
void PeakSegFPOP_interface(char **file_vec, char **pen_vec, char **temp_vec) {
  char *bedGraph = file_vec[0];
  char *penalty = pen_vec[0];
  char *db = temp_vec[0];
  peakseg_hip_set_print(to_r_console); /* "problem: %d items on line %d" (PeakSegFPOPLog.cpp:181) */
  int status = PeakSegFPOP_disk(bedGraph, penalty, db);
  if (status != 0) {
    char msg[4096];
    PeakSegFPOP_status_message(status, bedGraph, penalty, db, msg, sizeof msg);
    if (status >= ERROR_NO_HIP_DEVICE) /* device-side failures carry a detail text */
      Rf_error("%s (%s)", msg, peakseg_hip_last_error());
    Rf_error("%s", msg);
  }
}

/* .C("PeakSegFPOP_batch_interface", files, penalties, dbs, n, status=integer(n)):
 * one parse and upload per distinct file, every dynamic program in one launch */
void PeakSegFPOP_batch_interface(char **files, char **pens, char **dbs, int *n, int *status) {
  peakseg_hip_set_print(to_r_console);
  PeakSegFPOP_disk_batch(*n, files, pens, dbs, status);
}

/* .C("PeakSegFPOP_dir_batch_interface", problem.dirs, penalties, n, status, cached):
 * PeakSegFPOP_dir's cache protocol and _timing.tsv for a batch (R/PeakSegFPOP_dir.R:70-108) */
void PeakSegFPOP_dir_batch_interface(char **dirs, char **pens, int *n, int *status, int *cached) {
  peakseg_hip_set_print(to_r_console);
  PeakSegFPOP_dir_batch(*n, dirs, pens, status, cached);
}

/* .C("PeakSegFPOP_search_interface", problem.dir, peaks.int, verbose, capacity,
 *    penalty=character(capacity), iteration=integer(capacity), n=integer(1), chosen=integer(1)):
 * sequentialSearch_dir's loop with the contig resident (R/sequentialSearch_dir.R:39-99); the
 * caller reads each model's files with PeakSegFPOP_dir (cache hits) to build $others. */
void PeakSegFPOP_search_interface(char **dir, int *peaks_int, int *verbose, int *capacity,
                                  char **penalty_out, int *iteration_out, int *under_out,
                                  int *over_out, int *n_out, int *chosen_out) {
  peakseg_hip_set_print(to_r_console);
  const int cap = *capacity;
  psd_search_row *rows = (psd_search_row *)R_alloc((size_t)cap, sizeof(psd_search_row));
  int n = 0, chosen = -1;
  int status = PeakSegFPOP_sequential_search(dir[0], *peaks_int, *verbose, cap, rows, &n, &chosen);
  if (status == ERROR_SEARCH_TOO_MANY_PEAKS) Rf_error("%s", peakseg_hip_last_error());
  if (status != 0) {
    char msg[4096];
    const char *pen = n < cap ? rows[n].penalty_str : "";
    PeakSegFPOP_status_message(status, dir[0], pen, "", msg, sizeof msg);
    Rf_error("%s", msg);
  }
  for (int k = 0; k < n; k++) {
    /* penalty_out[k] was allocated by R with room for 39 characters (character(capacity) filled
     * with strrep(" ", 39) by the R wrapper) */
    snprintf(penalty_out[k], 40, "%s", rows[k].penalty_str);
    iteration_out[k] = rows[k].iteration;
    under_out[k] = rows[k].under_peaks; /* INT_MIN is R's NA_integer_ */
    over_out[k] = rows[k].over_peaks;
  }
  *n_out = n;
  *chosen_out = chosen + 1; /* 1-based for R */
}

/* .C("PeakSegFPOP_search_batch_interface", problem.dir.vec, n, peaks.int.vec, verbose, capacity,
 *    penalty=character(n*capacity), iteration=, under=, over=integer(n*capacity),
 *    n.models=integer(n), chosen=integer(n), status=integer(n)): sequentialSearch_dir on every
 * directory, the models of one iteration in one launch; directory d's rows are
 * [d*capacity, d*capacity + n.models[d]). */
void PeakSegFPOP_search_batch_interface(char **dirs, int *n_dirs, int *peaks_int, int *verbose,
                                        int *capacity, char **penalty_out, int *iteration_out,
                                        int *under_out, int *over_out, int *n_out,
                                        int *chosen_out, int *status_out) {
  peakseg_hip_set_print(to_r_console);
  const int cap = *capacity, n = *n_dirs;
  psd_search_row *rows =
      (psd_search_row *)R_alloc((size_t)cap * (size_t)n, sizeof(psd_search_row));
  PeakSegFPOP_sequential_search_batch(n, dirs, peaks_int, *verbose, cap, rows, n_out, chosen_out,
                                      status_out);
  for (int d = 0; d < n; d++) {
    for (int k = 0; k < n_out[d]; k++) {
      const size_t i = (size_t)d * (size_t)cap + (size_t)k;
      snprintf(penalty_out[i], 40, "%s", rows[i].penalty_str);
      iteration_out[i] = rows[i].iteration;
      under_out[i] = rows[i].under_peaks; /* INT_MIN is R's NA_integer_ */
      over_out[i] = rows[i].over_peaks;
    }
    chosen_out[d] += 1; /* 1-based for R */
  }
}

static R_NativePrimitiveArgType PeakSegFPOP_types[] = {STRSXP, STRSXP, STRSXP};
static R_NativePrimitiveArgType batch_types[] = {STRSXP, STRSXP, STRSXP, INTSXP, INTSXP};
static R_NativePrimitiveArgType dir_batch_types[] = {STRSXP, STRSXP, INTSXP, INTSXP, INTSXP};
static R_NativePrimitiveArgType search_types[] = {STRSXP, INTSXP, INTSXP, INTSXP, STRSXP,
                                                  INTSXP, INTSXP, INTSXP, INTSXP, INTSXP};

static R_NativePrimitiveArgType search_batch_types[] = {STRSXP, INTSXP, INTSXP, INTSXP,
                                                        INTSXP, STRSXP, INTSXP, INTSXP,
                                                        INTSXP, INTSXP, INTSXP, INTSXP};

static const R_CMethodDef cMethods[] = {
    {"PeakSegFPOP_interface", (DL_FUNC)&PeakSegFPOP_interface, 3, PeakSegFPOP_types},
    {"PeakSegFPOP_batch_interface", (DL_FUNC)&PeakSegFPOP_batch_interface, 5, batch_types},
    {"PeakSegFPOP_dir_batch_interface", (DL_FUNC)&PeakSegFPOP_dir_batch_interface, 5,
     dir_batch_types},
    {"PeakSegFPOP_search_interface", (DL_FUNC)&PeakSegFPOP_search_interface, 10, search_types},
    {"PeakSegFPOP_search_batch_interface", (DL_FUNC)&PeakSegFPOP_search_batch_interface, 12,
     search_batch_types},
    {NULL, NULL, 0, NULL}};

void R_init_PeakSegDisk(DllInfo *info) {
  R_registerRoutines(info, cMethods, NULL, NULL, NULL);
  /* .C only finds registered symbols: R's own argument-type errors
   * (tests/testthat/test-CRAN-cpp-errors.R:172-203) keep coming from this table */
  R_useDynamicSymbols(info, FALSE);
}

} /* extern "C" */
