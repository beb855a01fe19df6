/* fpop_types.h -- what the host driver and every build variant of the kernels share: problem
 * status codes, per-problem results and the kernel argument block.
 */
#ifndef PSD_FPOP_TYPES_H
#define PSD_FPOP_TYPES_H

namespace psd {

constexpr int N_PROF = 24; /* PSD_PROFILE builds: cycle counters per wave */

/* The arena is handed out in chunks of 2^ar_chunk_log2 pieces (one global atomic per chunk, one
 * cursor per wave).  The host picks the chunk size per problem set: 65536 pieces for long
 * contigs, down to 1024 for sets of many short problems, so that the unused tail of each
 * wave's last chunk stays a small fraction of the arena. */
constexpr int ARENA_CHUNK_LOG2_MAX = 16;
constexpr int ARENA_CHUNK_LOG2_MIN = 10;
/* ... and is made of blocks of 2^ar_block_log2 pieces, each a device allocation of its own; the
 * host picks the block size per set (an eighth of its first estimate).  The checkpointed store
 * keeps a wave's whole region in one block and may need larger ones. */
constexpr int ARENA_BLOCK_LOG2_MIN = 19;
constexpr int ARENA_BLOCK_LOG2_MAX = 24;
constexpr int ARENA_BLOCK_LOG2_CKPT_MAX = 30;
constexpr int FN_COUNT_BITS = 24;     /* fn_ref = (arena offset << 24) | piece count */
/* merged-interval table entries pack (piece of f1 << 16) | piece of f2 into an int */
constexpr int SPILL_CAP_MAX = 32767;

/* problem status written by the kernels (0 = ok) */
enum {
  PST_OK = 0,
  PST_LDS_OVERFLOW = 1,  /* a list outgrew LDS_CAP and the spill path is not available */
  PST_ARENA_FULL = 2,    /* host retries with a larger arena */
  PST_REF_THROW = 3,     /* the reference would throw / loop / read a list sentinel */
  PST_BACKTRACK = 4,     /* findMean found no piece (the reference would never return) */
  PST_SPILL_FULL = 5,    /* no free slot in the HBM spill pool: parked; the host resumes it with more slots */
  PST_CKPT_SPILL = 6,    /* (rounds 1-2: a function to checkpoint had outgrown LDS; no longer raised) */
  PST_CKPT_FULL = 7,     /* checkpointed store: the overflow pool for checkpoints of functions with
                            more than ckpt_cap pieces is exhausted: host retries with a larger one */
};

struct ProbResult {
  double best_cost;       /* Minimize() of the last down function (drv:404-406) */
  double best_log_mean;
  double prev_log_mean;
  int prev_seg_end;
  int status;
  int wave_err;           /* WERR_* bits for diagnostics */
  int max_intervals;      /* drv:374-379 */
  unsigned long long total_intervals; /* drv:373 */
  int n_segments;         /* filled by the backtrack kernel */
  int n_equality;         /* drv:411,436 */
  int n_serial_env;       /* min-envelope calls that needed the sequential replay */
  int step_reached;
  int spill_steps;        /* data points processed with the lists in the HBM spill area */
  int parked;             /* PST_ARENA_FULL: the state before data point step_reached is in the
                             problem's park slot, the problem can be resumed there */
  int max_spin;           /* -DPSD_SPIN_STATS builds: largest poll count of a wait between waves */
  long long cycles;       /* shader cycles the workgroup ran (this launch) */
};

struct DeviceArgs {
  /* Device address of a copy of this block.  The kernel receives the block by value (scalar
   * loads from the kernel-argument segment); out-of-line device functions get *self instead of
   * a reference to the by-value copy, which the compiler would have to spill to scratch memory
   * -- and then read back from there in the kernel's loop as well. */
  const DeviceArgs *self;
  int n_problems;
  /* per problem */
  const int *prob_contig;
  const double *prob_penalty;
  const long long *prob_fn_off;  /* into fn_ref: 2*N entries (up: [0,N), down: [N,2N)) */
  const long long *prob_seg_off; /* into seg_start / seg_mean: N+1 entries */
  const int *prob_order;         /* workgroup b solves problem prob_order[b]: longest first */
  ProbResult *result;
  /* per contig */
  const int *contig_n;
  const long long *contig_off; /* into count / weight */
  const double *contig_min_log_mean;
  const double *contig_max_log_mean;
  const int *count;  /* 4th bedGraph column */
  const int *weight; /* chromEnd - chromStart */
  /* arena: the in-HBM cost-function store, made of BLOCKS of 2^ar_block_log2 pieces.  Every
   * block is a device allocation (an address range) of its own, laid out
   *   [max_log_mean f64 x B][prev_log_mean f64 x B][data_i i32 x B]      (20 bytes per piece),
   * and the host can add blocks WHILE THE KERNEL RUNS: memory mapped behind an address range of
   * its own is visible to a running kernel (tools/vmm_block_probe.cpp; mapping behind a range
   * the kernel already uses waits for the kernel, tools/vmm_overlap_probe.cpp).  A piece offset
   * is (block << ar_block_log2) | index; chunks and function records never straddle a block.
   *   ar_block  device table of block base addresses: filled by the host for the blocks that
   *             exist at launch, by the wave that first takes a chunk of a later block for those
   *             (from ar_live), so that the decoding finds every block of its own problem there
   *   ar_cap    pieces mapped when the kernel was launched
   *   ar_live   pinned host memory (nullptr: the arena cannot grow during the launch):
   *             [0] pieces mapped by now, [1] != 0: no more will come, [2 + b] base of block b
   *   ar_used   pinned host word: high-water mark of the pieces handed out (the host maps ahead
   *             of it)
   * A wave that still finds no room parks its problem (its two live functions go to the
   * problem's park slot) and the host relaunches those problems only, from the data point they
   * had reached, after adding blocks. */
  char **ar_block;
  int ar_block_log2;
  unsigned long long ar_cap;
  const unsigned long long *ar_live;
  unsigned long long *ar_used;
  /* per problem: the data point to resume at after a regrowth (0: from the start); nullptr when
   * the set has no park slots */
  const int *prob_resume;
  int ar_chunk_log2;
  unsigned long long *ar_next_chunk;
  unsigned long long *fn_ref;
  /* segment tables, in backtrack order */
  int *seg_start;   /* data index whose chromEnd starts the segment; -1 = first_chromStart */
  double *seg_mean; /* exp(best_log_mean) */
  long long *prof;  /* PSD_PROFILE builds: per (problem, wave) cycle counters, else NULL */
  /* spill pool for functions with more than LDS_CAP pieces: spill_slots slots of 48*spill_cap
   * doubles (6 lists x 6 fields + 2 waves x 6 scratch arrays) and 12*spill_cap ints; a problem
   * takes a slot (spill_next, one atomic) the first time one of its functions outgrows LDS */
  double *spill_f64;
  int *spill_i32;
  int spill_cap;
  int spill_slots;
  int *spill_next;
  /* Checkpointed store (ckpt_interval = 0: every function is stored,
   * as the reference's DiskVector does).  With ckpt_interval = K > 0 the forward pass keeps no
   * per-step records, only the two live functions after every K-th data point; the decoding
   * recomputes, block of K data points by block, the records it walks through, into a private
   * region of the arena (ckpt_region pieces per chain).  fn_ref then holds 2 (K+1) entries per
   * problem.  Per checkpoint: 12 ckpt_cap + 4 doubles and 2 ckpt_cap + 2 ints.  A function with
   * more than ckpt_cap pieces (adversarial data, lists in the HBM spill area) is checkpointed into
   * the overflow pool instead -- n pieces taken with one atomic, their offset kept in the slot's
   * header -- so that checkpoints never limit the size of a function. */
  int ckpt_interval;
  int ckpt_cap;
  unsigned long long ckpt_region;
  const long long *prob_ckpt_off; /* first checkpoint slot of each problem */
  double *ckpt_f64;
  int *ckpt_i32;
  double *ckpt_ovf_f64;            /* overflow pool: 6 doubles + 1 int per piece */
  int *ckpt_ovf_i32;
  unsigned long long ckpt_ovf_cap; /* pieces */
  unsigned long long *ckpt_ovf_next;
  /* mixed launch: host-visible count of latency-build workgroups that have started (the host
   * holds the packed part back until they all have their CUs); nullptr otherwise */
  int *started;
};

}  // namespace psd
#endif
