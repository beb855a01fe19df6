/* peakseg_oracle.h -- CPU oracle for the PeakSegFPOP hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or run anything in this directory; nothing under
 * peaksegdisk_amd/ includes or links it.
 *
 * It is a plain-C restatement (flat arrays instead of std::list, an fopen/fseek store
 * instead of std::fstream) of the reference algorithm in
 *   /root/reference/src/funPieceListLog.cpp   (piece-list algebra)
 *   /root/reference/src/PeakSegFPOPLog.cpp    (solver driver, DiskVector, backtrack, writers)
 * with every function citing the reference lines it follows.
 *
 * Two builds (oracle/Makefile):
 *   liboracle_libm.so  exp/log = the host libm, as the reference uses them. Pinned against
 *                      the reference's own test known-answers and the reference outputs
 *                      recorded in SURVEY.md section 8c (tests/golden/).
 *   liboracle_det.so   exp/log = include/peakseg_detmath.h, the deterministic pair the HIP
 *                      kernels use; the GPU path must match this build bit for bit.
 *
 * Parity pin: the reference itself cannot be built in this image (it needs R's headers,
 * which are absent, and stand-in headers are not allowed), so the oracle is pinned by the
 * reference's test fixtures and recorded outputs only -- see DESIGN.md.
 */
#ifndef PEAKSEG_ORACLE_H
#define PEAKSEG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: /root/reference/src/PeakSegFPOPLog.h:3-13 */
#define ORACLE_ERROR_PENALTY_NOT_FINITE 1
#define ORACLE_ERROR_PENALTY_NEGATIVE 2
#define ORACLE_ERROR_UNABLE_TO_OPEN_BEDGRAPH 3
#define ORACLE_ERROR_NOT_ENOUGH_COLUMNS 4
#define ORACLE_ERROR_NON_INTEGER_DATA 5
#define ORACLE_ERROR_INCONSISTENT_CHROMSTART_CHROMEND 6
#define ORACLE_ERROR_WRITING_COST_FUNCTIONS 7
#define ORACLE_ERROR_WRITING_LOSS_OUTPUT 8
#define ORACLE_ERROR_NO_DATA 9
#define ORACLE_ERROR_PENALTY_NOT_NUMERIC 10
#define ORACLE_ERROR_WRITING_SEGMENTS_OUTPUT 11
/* the reference would std::terminate here (uncaught throw); the oracle reports it */
#define ORACLE_ERROR_REFERENCE_WOULD_THROW 100

/* Same contract as the reference's PeakSegFPOP_disk (PeakSegFPOPLog.cpp:143-463):
 * reads bedGraph, writes <bedGraph>_penalty=<pen>_segments.bed and _loss.tsv, uses db as
 * the cost-function store (same byte layout as the reference's DiskVector). */
int oracle_PeakSegFPOP_disk(const char *bedGraph_file_name, const char *penalty_str,
                            const char *db_file_name);

/* "libm" or "detmath": which exp/log this build uses. */
const char *oracle_math_kind(void);

/* element-wise probes of the math this build uses (tests only) */
void oracle_exp_vec(int n, const double *x, double *y);
void oracle_log_vec(int n, const double *x, double *y);

#ifdef __cplusplus
}
#endif
#endif
