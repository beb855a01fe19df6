/* peaksegdisk_hip.h -- C ABI of libpeaksegdisk_hip.so, the MI355X-native drop-in for
 * PeakSegDisk's PeakSegFPOP hot path.
 *
 * Plain C types only (pointers, sizes, ints, doubles).  Each entry point cites the
 * reference interface it replaces; INTEGRATION.md shows the bindings (R `.C` glue, ctypes).
 */
#ifndef PEAKSEGDISK_HIP_H
#define PEAKSEGDISK_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------- */
/* 1..11: identical to /root/reference/src/PeakSegFPOPLog.h:3-13 */
#define ERROR_PENALTY_NOT_FINITE 1
#define ERROR_PENALTY_NEGATIVE 2
#define ERROR_UNABLE_TO_OPEN_BEDGRAPH 3
#define ERROR_NOT_ENOUGH_COLUMNS 4
#define ERROR_NON_INTEGER_DATA 5
#define ERROR_INCONSISTENT_CHROMSTART_CHROMEND 6
#define ERROR_WRITING_COST_FUNCTIONS 7
#define ERROR_WRITING_LOSS_OUTPUT 8
#define ERROR_NO_DATA 9
#define ERROR_PENALTY_NOT_NUMERIC 10
#define ERROR_WRITING_SEGMENTS_OUTPUT 11
/* additions of this implementation (the reference reports "error code %d" for them) */
#define ERROR_NO_HIP_DEVICE 12     /* no MI355X visible: there is no CPU fallback */
#define ERROR_DEVICE_SOLVER 13     /* kernel reported a failure; see peakseg_hip_last_error() */
#define ERROR_DEVICE_MEMORY 14     /* arena / tables do not fit in HBM (or in PEAKSEG_HIP_MAX_BYTES) */
#define ERROR_SEARCH_ARGUMENTS 15  /* PeakSegFPOP_sequential_search: bad arguments / row capacity */
#define ERROR_SEARCH_TOO_MANY_PEAKS 16 /* peaks.int exceeds the maximum for the data; the text of
                                          R/sequentialSearch_dir.R:57-66 is in peakseg_hip_last_error() */

/* ---- environment ---------------------------------------------------------------------
 * PEAKSEG_HIP_DEVICE            GPU used by the file-level entry points (default 0); one process
 *                               per GPU sets it from its rank
 * PEAKSEG_HIP_MAX_BYTES         cap on the HBM one problem set may hold (suffix K/M/G/T); several
 *                               processes can then share one GPU (R's future workers)
 * PEAKSEG_HIP_PIECES_PER_FUNCTION  arena estimate, pieces per stored cost function (default 7):
 *                               picks the arena's chunk and block sizes; the arena itself grows
 *                               block by block WHILE the kernel runs (a host thread maps ahead of
 *                               what the waves have taken), up to nine tenths of the free HBM or
 *                               PEAKSEG_HIP_MAX_BYTES
 * PEAKSEG_HIP_NO_LIVE_GROWTH=1  map what the estimate asks for at creation and nothing under a
 *                               running kernel: a solve that needs more parks its problems, adds
 *                               blocks and resumes them where they stopped
 * PEAKSEG_HIP_SPILL_CAP / _SPILL_SLOTS  capacity (pieces per list, at most 32767) and initial
 *                               number of slots of the HBM spill pool for functions that outgrow LDS
 * PEAKSEG_HIP_CHECKPOINT=K / PEAKSEG_HIP_NO_CHECKPOINT=1  force / forbid the checkpointed store
 * PEAKSEG_HIP_CKPT_OVERFLOW     initial size (pieces) of the pool that holds checkpoints of functions
 *                               too long for a checkpoint slot (adversarial data)
 * PEAKSEG_HIP_NO_PARK=1         rerun a set that ran out of arena instead of resuming its problems
 * PEAKSEG_HIP_NO_VMM=1          arena blocks from hipMalloc instead of the HIP virtual-memory calls
 *                               (same growth, nothing is ever copied)
 * PEAKSEG_HIP_ARENA_BLOCK_LOG2  tests: log2 of the pieces per arena block (default 19-24)
 * PEAKSEG_HIP_VARIANT=lat|thr|pk  force a build of the forward kernel
 * PEAKSEG_HIP_NO_PACKED=1       the launch planner never picks the packed build (pk)
 * PEAKSEG_HIP_RATES=lat,thr     diagnostic: data points per second per problem the launch planner
 *                               assumes for the latency / throughput build (default: measured by
 *                               this process's earlier solves, else 96000,58000)
 * PEAKSEG_HIP_TIMING=1          phase timings of the file-level calls on stderr */

/* ---- the reference's boundary -------------------------------------------------------- */

/* Replaces `int PeakSegFPOP_disk(char*, char*, char*)`
 * (/root/reference/src/PeakSegFPOPLog.h:15, PeakSegFPOPLog.cpp:143-463): same arguments,
 * same status codes, same validation order, byte-identical
 * <bedGraph>_penalty=<pen>_segments.bed / _loss.tsv files.  The dynamic program runs on the
 * GPU and the cost-function store lives in HBM; db_file_name is still opened (status 7 if
 * that fails, as the reference) and is left as a sparse file of exactly the size the
 * reference's database would have, so callers that report its size keep working. */
int PeakSegFPOP_disk(char *bedGraph_file_name, char *penalty_str, char *db_file_name);

/* Additive batch form of the same call for penalty grids: problem i is
 * (bedGraph_files[i], penalty_strs[i], db_files[i]); every distinct bedGraph file is parsed
 * and uploaded once, all dynamic programs run concurrently (one workgroup each), and the
 * output files are byte-identical to n separate PeakSegFPOP_disk calls.  status_out[i]
 * receives each problem's status; the return value is the first non-zero one (0 if none). */
int PeakSegFPOP_disk_batch(int n_problems, char **bedGraph_files, char **penalty_strs,
                           char **db_files, int *status_out);

/* PeakSegFPOP_dir for a batch of (problem directory, penalty string) pairs
 * (/root/reference/R/PeakSegFPOP_dir.R:64-117 over R/PeakSegFPOP_file.R:57-86): pairs whose
 * coverage.bedGraph_penalty=<pen>_{segments.bed,loss.tsv,timing.tsv} files exist and pass the
 * reference's consistency test are skipped (cached_out[i] = 1); the rest are solved in one
 * PeakSegFPOP_disk_batch (default database name, removed afterwards) and get their _timing.tsv
 * (penalty, megabytes = size of the reference's database / 2^20, seconds = the problem's share
 * of the batch's wall time).  Returns the first non-zero status (0 if none). */
int PeakSegFPOP_dir_batch(int n_problems, char **problem_dirs, char **penalty_strs,
                          int *status_out, int *cached_out);

/* One model visited by the penalty search: a row of sequentialSearch_dir's $others. */
typedef struct {
  char penalty_str[40];  /* paste(penalty): names the result files of this model */
  double penalty;        /* loss.tsv column 1 */
  double total_loss;     /* loss.tsv column 7 */
  int peaks, segments, bases;
  int iteration;         /* 1-based */
  int under_peaks, over_peaks; /* peaks of the bracket when the model was requested; INT_MIN = NA */
  int cached;            /* result files were reused, no dynamic program ran */
} psd_search_row;

/* sequentialSearch_dir(problem.dir, peaks.int) (/root/reference/R/sequentialSearch_dir.R:22-103)
 * next to the solver: the same loop, the same penalties (R's 15-significant-digit paste() of
 * each secant step) and the same files as the reference leaves behind for every model --
 * PeakSegFPOP_dir's cache included -- but coverage.bedGraph is parsed and uploaded once and the
 * arena is reused from one penalty to the next.  rows[0..*n_rows) are the models in the order
 * they were requested (iteration 1: "0" then "Inf"); *chosen_row is the returned model.
 * Status: 0, a solver status, ERROR_SEARCH_TOO_MANY_PEAKS or ERROR_SEARCH_ARGUMENTS. */
int PeakSegFPOP_sequential_search(const char *problem_dir, int peaks_int, int verbose,
                                  int row_capacity, psd_search_row *rows, int *n_rows,
                                  int *chosen_row);

/* The same search over several problem directories at once (additive: the reference runs one
 * R process per directory, /root/reference/R/sequentialSearch_dir.R:34-38 with a `future`
 * plan).  Every directory follows its own search -- the penalties, files and rows are those
 * of PeakSegFPOP_sequential_search on that directory alone -- but the models the searches ask
 * for in the same iteration are computed in one launch (PeakSegFPOP_dir_batch), so a set of
 * contigs is searched in about the time of its longest search.  rows: n_dirs x row_capacity
 * (directory d at rows + d*row_capacity); n_rows, chosen_row, status_out: one per directory
 * (a directory whose search fails keeps its status and the others go on).  Returns the first
 * non-zero status. */
int PeakSegFPOP_sequential_search_batch(int n_dirs, char **problem_dirs, const int *peaks_int,
                                        int verbose, int row_capacity, psd_search_row *rows,
                                        int *n_rows, int *chosen_row, int *status_out);

/* The text the reference's glue passes to Rf_error for a status
 * (/root/reference/src/interface.cpp:16-55); returns buf; empty string for status 0. */
char *PeakSegFPOP_status_message(int status, const char *bedGraph, const char *penalty,
                                 const char *db, char *buf, size_t buf_len);

/* Where "problem: %d items on line %d" goes (the reference Rprintf()s it,
 * PeakSegFPOPLog.cpp:181).  NULL restores the default (stdout). */
void peakseg_hip_set_print(void (*print)(const char *text));

/* ---- device-resident problem sets (penalty x contig grids, benchmarking) ------------ */

typedef struct psd_problem_set psd_problem_set;

typedef struct {
  int status;             /* 0, or ERROR_DEVICE_* */
  int kernel_status;      /* PST_* detail from the kernel */
  int n_segments;
  int n_peaks;
  int n_equality_constraints;
  int max_intervals;
  unsigned long long total_intervals;
  double best_cost;       /* mean penalized cost (loss.tsv column 6) */
  int n_serial_env;       /* diagnostics: min-envelope calls replayed sequentially */
  int step_reached;
  int spill_steps;        /* data points processed with the piece lists spilled to HBM */
} psd_result;

int peakseg_hip_device_count(void);
/* shader clock of a device in kHz, 0 when unknown */
int peakseg_hip_device_clock_khz(int device);
/* text of the calling thread's last FAILURE (statuses >= 12 carry their detail here) */
const char *peakseg_hip_last_error(void);
/* Something worth knowing about the calling thread's last problem-set solve although it
 * succeeded (e.g. the mixed launch's wait for its latency-build workgroups ran into its bound);
 * "" when there is nothing.  Never part of an error message. */
const char *peakseg_hip_last_warning(void);

/* Upload contigs (count = 4th bedGraph column, weight = chromEnd-chromStart) and the
 * problem list to HBM and allocate the arena (arena_pieces = 0: sized automatically and
 * grown on demand).  One problem = one (contig, penalty) dynamic program. */
int peakseg_hip_problem_set_create(int device, int n_contigs, const int *contig_n_bins,
                                   const int *const *contig_count,
                                   const int *const *contig_weight, int n_problems,
                                   const int *problem_contig, const double *problem_penalty,
                                   unsigned long long arena_pieces, psd_problem_set **out);

/* Run forward DP + backtrack for every problem of the set; inputs are already resident.
 * *forward_ms = duration of the kernel, measured with HIP events on the stream it runs on.
 * Each workgroup decodes its segmentation right after its last data point, inside the same
 * kernel, so there is no separate backtrack launch: *backtrack_ms is always 0. */
int peakseg_hip_problem_set_solve(psd_problem_set *set, float *forward_ms, float *backtrack_ms);

int peakseg_hip_problem_set_result(psd_problem_set *set, int problem, psd_result *out);

/* Segment table of one problem in the reference's output order (last segment first):
 * seg_start[r] = index of the data point whose chromEnd starts segment r (-1: the first
 * chromStart), seg_mean[r] = segment mean.  Returns the number of rows, or -1. */
int peakseg_hip_problem_set_segments(psd_problem_set *set, int problem, int capacity,
                                     int *seg_start, double *seg_mean);

/* Debug/parity aid: write one problem's in-HBM cost-function store in the byte layout of
 * the reference's DiskVector file (PeakSegFPOPLog.cpp:12-34,76-141); chromEnd[] supplies
 * the per-function chromEnd field. */
int peakseg_hip_problem_set_export_db(psd_problem_set *set, int problem, const int *chromEnd,
                                      const char *path);

/* Which build of the forward kernel the last solve used: "lat" (latency build: helper waves,
 * one workgroup per CU; sets of at most one problem per CU), "thr" (throughput build:
 * 4 workgroups per CU, 64-piece lists), "pk" (packed build: 6 workgroups per CU, 40-piece lists;
 * problems whose functions outgrow them go on, from the data point reached, on a wider build),
 * or "lat+thr" / "lat+pk" (a set that oversubscribes the chip with contigs of unequal length: its
 * longest problems on the latency build, the rest packed, concurrently).  The planner takes
 * the build it predicts to finish the set first; same results whichever it is.
 * PEAKSEG_HIP_VARIANT=lat|thr|pk overrides the choice, PEAKSEG_HIP_NO_PACKED=1 excludes pk. */
const char *peakseg_hip_problem_set_kernel_build(psd_problem_set *set);

/* bytes of HBM held by the set (arena + tables) */
unsigned long long peakseg_hip_problem_set_bytes(psd_problem_set *set);

/* 0 when every cost function is kept in HBM (the reference's store, in memory), K > 0 when the
 * set uses the checkpointed store: only a checkpoint every K data points is kept and the
 * decoding recomputes the blocks it walks through (chosen automatically when the full store
 * would not fit; PEAKSEG_HIP_CHECKPOINT=K forces it, PEAKSEG_HIP_NO_CHECKPOINT=1 forbids it). */
int peakseg_hip_problem_set_checkpoint_interval(psd_problem_set *set);

/* bytes of the arena the last solve handed out (whole chunks) */
unsigned long long peakseg_hip_problem_set_arena_bytes_used(psd_problem_set *set);

/* How the last solve went: kernel launches (1 unless a store ran out between launches: blocks
 * are added to the arena and the problems it had parked are resumed, finished problems are never
 * repeated; growth UNDER the kernel does not count, see peakseg_hip_problem_set_arena_stats) and
 * the data points its launches worked through (the sum of the problems' lengths when nothing
 * was repeated). */
int peakseg_hip_problem_set_solve_stats(psd_problem_set *set, int *launches,
                                        unsigned long long *steps_run);

/* The arena of the set (include/../csrc/fpop_types.h): pieces per block, blocks mapped now, and
 * how many of them the last solve mapped WHILE its kernels ran (0 with
 * PEAKSEG_HIP_NO_LIVE_GROWTH=1, with an explicit arena size and for the checkpointed store). */
int peakseg_hip_problem_set_arena_stats(psd_problem_set *set, unsigned long long *block_pieces,
                                        int *blocks, int *blocks_added_live);

/* How often the last solve parked a problem (the arena had run out; the problem was resumed
 * after more had been mapped) and how many pieces of the overflow pool -- where a parked
 * problem keeps functions too long for its park slot -- those parks took. */
int peakseg_hip_problem_set_park_stats(psd_problem_set *set, int *parks,
                                       unsigned long long *overflow_pool_pieces);

/* Pack the segment tables of a solved set at their exact sizes, IN HBM: problem p's rows
 * (rows_out[p] of them, n_problems entries, host) follow problem p-1's.  Returns the total
 * number of rows, -1 on failure.  *start_dev / *mean_dev receive the device addresses of the
 * packed arrays (valid until the set is solved again or destroyed): what the multi-GPU gather
 * hands to RCCL without a host round trip (peaksegdisk_amd/parallel.py);
 * peakseg_hip_problem_set_packed_download copies them to host arrays of that many rows. */
long long peakseg_hip_problem_set_pack_tables(psd_problem_set *set, long long *rows_out,
                                              const int **start_dev, const double **mean_dev);
int peakseg_hip_problem_set_packed_download(psd_problem_set *set, int *start_out, double *mean_out);

/* Shader cycles a problem's workgroup ran in the last launch that worked on it (0 when
 * unknown); with the device's clock, data points / cycles is the problem's rate: the dealing of
 * problems to ranks is sized from it. */
long long peakseg_hip_problem_set_cycles(psd_problem_set *set, int problem);
/* Data points per second one problem advanced at in this process's last clean solves on the
 * latency build (a CU per problem) and on the throughput build (a full chip). */
void peakseg_hip_measured_rates(double *lat_rate, double *thr_rate);

/* Tests: the bound on the number of polls of a wait between the waves of a workgroup, and (in
 * builds with -DPSD_SPIN_STATS, 0 otherwise) the largest poll count a problem's waves saw. */
long long peakseg_hip_spin_limit(void);
int peakseg_hip_problem_set_max_spin(psd_problem_set *set, int problem);

/* Change one problem's penalty in place (the contig stays resident, the arena is reused by the
 * next solve): what the penalty search does between its dynamic programs.  0 or -1. */
int peakseg_hip_problem_set_set_penalty(psd_problem_set *set, int problem, double penalty);

void peakseg_hip_problem_set_destroy(psd_problem_set *set);

/* Tests: parse a bedGraph file the way PeakSegFPOP_disk does (use_fast != 0: byte scanner with
 * sscanf fallback; 0: the reference's sscanf format on every line) and return the status, the
 * number of data lines and a hash of everything parsed. */
int peakseg_hip_parse_probe(const char *path, int use_fast, int *n_lines,
                            unsigned long long *hash);

/* Tests: R's paste() of a double (15 significant digits) as the penalty search and the timing
 * files format numbers; returns the length. */
int peakseg_hip_paste_double(double x, char *buf, size_t buf_len);

/* diagnostic builds (-DPSD_PROFILE): per-wave cycle counters of the forward kernel; -1 in
 * normal builds */
int peakseg_hip_problem_set_profile(psd_problem_set *set, int problem, long long *out);

/* y[i] = exp(x[i]) (op 0) or log(x[i]) (op 1) evaluated on the device with the library's
 * deterministic math (include/peakseg_detmath.h); op 2: y[i] = psd_div(x[i], x[n + i]) (x holds
 * n numerators, then n divisors), op 3: the same quotients by the compiler's own fp64 division
 * sequence, which psd_div repairs; tests compare with the host build. */
int peakseg_hip_math_probe(int op, int n, const double *x, double *y);

#ifdef __cplusplus
}
#endif
#endif
