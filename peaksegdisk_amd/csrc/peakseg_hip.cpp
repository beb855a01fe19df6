/* peakseg_hip.cpp -- host driver and C ABI of libpeaksegdisk_hip.so.
 *
 * Host-side counterpart of the reference's solver driver
 * (/root/reference/src/PeakSegFPOPLog.cpp:143-463, "drv"): bedGraph parsing and validation,
 * the trivial one-segment branch, upload, kernel launches, and the two output files.  The
 * dynamic program itself only exists as HIP kernels (fpop_kernels.h): when no GPU is
 * visible the DP branch fails with ERROR_NO_HIP_DEVICE -- there is no CPU fallback.
 *
 * Compiled with: hipcc -x hip --offload-arch=gfx950 -ffp-contract=off
 * (tests/emu builds the same file with g++ -DPSD_EMU against the SIMT emulator).
 */
#include "../../include/peaksegdisk_hip.h"

/* Two builds of the same kernel source:
 *   lat  latency build: 128 pieces per LDS list, a helper wave per chain, and the piece-list
 *        operations inlined into the kernel's loop (4 waves per workgroup, all registers of
 *        the SIMDs: 1 workgroup per CU).  Fastest per problem; used while every problem of
 *        the set gets a CU of its own (the 64-penalty grid of one contig runs here).
 *   thr  throughput build: 64 pieces per LDS list, no helper waves, operations out of line
 *        (2 waves per workgroup, 4 workgroups per CU).  ~12% slower per problem, four times
 *        the problems per CU; used for sets that oversubscribe the chip (many contigs x many
 *        penalties).  (Registers for 3 waves per SIMD with 56-piece lists -- 5 workgroups per
 *        CU, -DPSD_THR_LDS_CAP=56 -DPSD_THR_WAVES_PER_EU=3 -- gain 7.5 % on 6144 equal problems
 *        and lose 7 % on the 24-contig x 64-penalty shape, whose end is set by its longest
 *        problems: profiles/r02/ab_thr_occupancy.log.)
 * Both produce identical results. */
#define PSD_VARIANT lat
#define PSD_LDS_CAP 128
#define PSD_MATH_VK 1
#ifndef PSD_NO_HELPER_WAVES /* -DPSD_NO_HELPER_WAVES: A/B builds (tools/ab_libs.py) */
#define PSD_HELPER_WAVES 1
#endif
#ifndef PSD_NO_FLAG_BARRIER /* -DPSD_NO_FLAG_BARRIER: A/B builds */
#define PSD_FLAG_BARRIER 1
#endif
#include "fpop_kernels.h"
#undef PSD_VARIANT
#undef PSD_LDS_CAP
#undef PSD_HELPER_WAVES
#undef PSD_FLAG_BARRIER
#undef PSD_MATH_VK
#define PSD_VARIANT thr
#ifndef PSD_THR_LDS_CAP /* A/B builds: -DPSD_THR_LDS_CAP=56 -DPSD_THR_WAVES_PER_EU=3 */
#define PSD_THR_LDS_CAP 64
#endif
#ifndef PSD_THR_WAVES_PER_EU
#define PSD_THR_WAVES_PER_EU 2
#endif
#define PSD_LDS_CAP PSD_THR_LDS_CAP
#define PSD_KERNEL_WAVES_PER_EU PSD_THR_WAVES_PER_EU
#ifdef PSD_THR_FLAG_BARRIER /* A/B: measured slower than the workgroup barrier */
#define PSD_FLAG_BARRIER 1
#endif
#if !defined(PSD_CALL_LDS_OPS) && !defined(PSD_THR_INLINE_OPS) /* A/B: -DPSD_THR_INLINE_OPS */
#define PSD_CALL_LDS_OPS 1
#define PSD_CALL_LDS_OPS_THR_ONLY 1
#endif
#include "fpop_kernels.h"
#undef PSD_VARIANT
#undef PSD_LDS_CAP
#undef PSD_KERNEL_WAVES_PER_EU
#ifdef PSD_CALL_LDS_OPS_THR_ONLY
#undef PSD_CALL_LDS_OPS
#undef PSD_CALL_LDS_OPS_THR_ONLY
#endif

/*   pk   packed build (round 4): 40 pieces per LDS list, registers for three waves per SIMD
 *        (six workgroups per CU), operations out of line.  A problem on a SIMD shared three ways
 *        advances more slowly than on the throughput build, but a CU holds half as many again:
 *        +18 % on sets of many problems of similar length, -10 % where a set ends with its
 *        longest packed problems (profiles/r04/ab_thr_occupancy_*.log) -- the planner in
 *        peakseg_hip_problem_set_solve picks it when it predicts the earlier end.  A function that
 *        outgrows 40 pieces does not go to the (ten times slower) HBM path here: the problem is
 *        parked and resumed on the throughput build, whose lists hold 64. */
#define PSD_VARIANT pk
#define PSD_LDS_CAP 40
#define PSD_KERNEL_WAVES_PER_EU 3
#define PSD_CALL_LDS_OPS 1
#define PSD_PARK_ON_LDS_OVERFLOW 1
#include "fpop_kernels.h"
#undef PSD_VARIANT
#undef PSD_LDS_CAP
#undef PSD_KERNEL_WAVES_PER_EU
#undef PSD_CALL_LDS_OPS
#undef PSD_PARK_ON_LDS_OVERFLOW

#include <errno.h>
#include <algorithm>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <map>
#include <thread>
#include <string>
#include <vector>

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace {

thread_local std::string g_last_error;
void (*g_print)(const char *) = nullptr;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

/* Things a caller may want to know about a call that SUCCEEDED (a wait that ran into its bound,
 * a fallback taken): kept apart from the error text, which describes failures only. */
thread_local std::string g_last_warning;
void set_warning(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_warning = buf;
  if (getenv("PEAKSEG_HIP_TIMING")) fprintf(stderr, "peakseg_hip warning: %s\n", buf);
}

void emit_text(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (g_print) {
    g_print(buf);
  } else {
    fputs(buf, stdout);
  }
}

#define HIP_TRY(expr)                                                               \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                         \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return e_ == hipErrorOutOfMemory ? ERROR_DEVICE_MEMORY : ERROR_DEVICE_SOLVER; \
    }                                                                               \
  } while (0)

/* ---- bedGraph input (drv:160-209) ---------------------------------------------------- */

struct Coverage {
  std::vector<int> chromEnd, count, weight;
  std::string chrom; /* the last line's first column (drv:166,178) */
  int first_chromStart = -1;
  double cum_weight = 0.0, cum_weighted_count = 0.0;
  double min_log_mean = INFINITY, max_log_mean = -INFINITY;
  int n() const { return (int)count.size(); }
};

/* One bedGraph line the way the reference's sscanf("%s %d %d %d%s") sees it
 * (drv:175-178).  Returns the item count sscanf would return (-1 for an empty line). */
static int scan_line_sscanf(const char *line, char *chrom, int *chromStart, int *chromEnd,
                            int *coverage, char *extra) {
  /* the reference reads "%s" into char[100] buffers (drv:166-167); bounded here */
  return sscanf(line, "%99s %d %d %d%99s\n", chrom, chromStart, chromEnd, coverage, extra);
}

static inline bool is_ws(unsigned char c) {
  return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r';
}

/* Fast path for the lines real files consist of: token, three decimal integers of at most
 * nine digits (optional '-' / '+'), nothing else.  Returns false for anything it is not sure
 * about; the caller then lets sscanf decide, so behaviour is the reference's by
 * construction (tests/test_cabi_cpu.py fuzzes fast vs sscanf-only). */
static bool scan_line_fast(const char *p, const char *end, char *chrom, int *v) {
  while (p < end && is_ws((unsigned char)*p)) p++;
  const char *tok = p;
  while (p < end && !is_ws((unsigned char)*p)) p++;
  size_t len = (size_t)(p - tok);
  if (len == 0 || len > 99) return false;
  memcpy(chrom, tok, len);
  chrom[len] = 0;
  for (int k = 0; k < 3; k++) {
    const char *q = p;
    while (p < end && is_ws((unsigned char)*p)) p++;
    if (p == q) return false; /* the integer must be separated from the previous field */
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) {
      neg = *p == '-';
      p++;
    }
    const char *d = p;
    int x = 0;
    while (p < end && *p >= '0' && *p <= '9') {
      x = x * 10 + (*p - '0');
      p++;
    }
    if (p == d || p - d > 9) return false;
    v[k] = neg ? -x : x;
  }
  while (p < end && is_ws((unsigned char)*p)) p++;
  return p == end; /* a fifth field: sscanf reports it */
}

/* Pass 1 of the reference (drv:173-205): same per-line conversions, same checks in the same
 * order; the whole file is read once and shared by every penalty of a batch.
 * use_fast = false forces the sscanf-only path (tests). */
int read_bedGraph_impl(const char *path, Coverage &cv, bool use_fast) {
  FILE *f = fopen(path, "rb");
  if (!f) return ERROR_UNABLE_TO_OPEN_BEDGRAPH;
  std::string buf;
  {
    char chunk[1 << 16];
    size_t got;
    while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) buf.append(chunk, got);
    fclose(f);
  }
  int chromStart = 0, chromEnd = 0, coverage = 0, items, line_i = 0;
  char chrom[100];
  char extra[100] = "";
  int prev_chromEnd = -1;
  int status = 0;
  std::string tmp;
  const char *p = buf.data(), *file_end = buf.data() + buf.size();
  while (p < file_end) { /* std::getline: up to '\n', the last line may lack it */
    const char *nl = (const char *)memchr(p, '\n', (size_t)(file_end - p));
    const char *line_end = nl ? nl : file_end;
    line_i++;
    int v[3];
    if (use_fast && scan_line_fast(p, line_end, chrom, v)) {
      chromStart = v[0];
      chromEnd = v[1];
      coverage = v[2];
      items = 4;
    } else {
      tmp.assign(p, (size_t)(line_end - p));
      /* an embedded NUL ends the line for sscanf, as line.c_str() does in the reference */
      items = scan_line_sscanf(tmp.c_str(), chrom, &chromStart, &chromEnd, &coverage, extra);
    }
    p = nl ? nl + 1 : file_end;
    if (items < 4) {
      emit_text("problem: %d items on line %d\n", items, line_i);
      status = ERROR_NOT_ENOUGH_COLUMNS;
      break;
    }
    if (0 < strlen(extra)) {
      status = ERROR_NON_INTEGER_DATA;
      break;
    }
    double weight = chromEnd - chromStart;
    cv.cum_weight += weight;
    cv.cum_weighted_count += weight * coverage;
    if (line_i == 1) {
      cv.first_chromStart = chromStart;
    } else if (chromStart != prev_chromEnd) {
      status = ERROR_INCONSISTENT_CHROMSTART_CHROMEND;
      break;
    }
    prev_chromEnd = chromEnd;
    double log_data = psd_log((double)coverage);
    if (log_data < cv.min_log_mean) cv.min_log_mean = log_data;
    if (cv.max_log_mean < log_data) cv.max_log_mean = log_data;
    cv.chromEnd.push_back(chromEnd);
    cv.count.push_back(coverage);
    cv.weight.push_back(chromEnd - chromStart);
  }
  if (status) return status;
  if (line_i == 0) return ERROR_NO_DATA;
  cv.chrom = chrom;
  return 0;
}

int read_bedGraph(const char *path, Coverage &cv) { return read_bedGraph_impl(path, cv, true); }

/* penalty string handling of drv:145-159 */
int parse_penalty(const char *s, bool &is_Inf, double &penalty) {
  is_Inf = strcmp(s, "Inf") == 0;
  char *end;
  errno = 0;
  penalty = strtod(s, &end);
  if (end == s) return ERROR_PENALTY_NOT_NUMERIC;
  if (is_Inf) return 0;
  if (!std::isfinite(penalty)) return ERROR_PENALTY_NOT_FINITE;
  if (penalty < 0) return ERROR_PENALTY_NEGATIVE;
  return 0;
}

}  // namespace

/* ---- device problem set --------------------------------------------------------------- */

struct psd_problem_set {
  int device = 0;
  int n_contigs = 0, n_problems = 0;
  std::vector<int> contig_n;
  std::vector<long long> contig_off;
  std::vector<int> prob_contig;
  std::vector<double> prob_penalty;
  std::vector<long long> prob_fn_off, prob_seg_off;
  long long total_bins = 0, fn_total = 0, seg_total = 0;
  unsigned long long arena_pieces = 0;
  bool arena_auto = true;
  unsigned long long max_bytes = 0;   /* PEAKSEG_HIP_MAX_BYTES (0 = no cap besides free HBM) */
  unsigned long long arena_used = 0;  /* pieces handed out by the last solve */
  long long dp_bins = 0;
  int spill_slots = 0;
  int ckpt_interval = 0;                 /* 0 = full store */
  unsigned long long ckpt_pieces_per_fn = 0; /* region sizing of the checkpointed store */
  psd::DeviceArgs d{};
  hipStream_t stream = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  std::vector<psd::ProbResult> results;
  bool solved = false;
  int n_cu = 0;            /* compute units of the device */
  bool throughput = false; /* which kernel build the last solve used */
  bool packed = false;     /* ... the packed build (pk) for the part that is not on the latency build */
  int widened = 0;         /* problems the packed build handed to the throughput build (a function
                              outgrew its 40-piece lists) */
  int n_lat_mixed = 0;     /* mixed launch: this many (longest) problems ran on the latency build */
  std::vector<int> order;  /* problems, longest contig first */
  bool can_park = false;   /* the set has a park slot per problem (full store) */
  /* The arena: blocks of 2^ar_block_log2 pieces, each a device allocation of its own (an
   * address range reserved, created, mapped and given access by the virtual-memory API; plain
   * hipMalloc without it).  Blocks can be added while a kernel runs (fpop_types.h), which is
   * what grow_arena_live() does from a second host thread during a solve. */
  struct ArenaBlock {
    void *base = nullptr;
    bool vmm = false;
#ifndef PSD_EMU
    hipMemGenericAllocationHandle_t handle{};
#endif
  };
  std::vector<ArenaBlock> arena_blocks;
  bool arena_vmm = false;
  size_t table_cap = 0;              /* entries of the two block tables */
  size_t table_synced = 0;           /* blocks whose address is in the device table */
  unsigned long long *h_live = nullptr; /* pinned: [0] pieces mapped, [1] final, [2 + b] bases */
  unsigned long long *h_used = nullptr; /* pinned: pieces handed out (kernel -> host) */
  bool live_growth = false;          /* this set's arena grows while its kernel runs */
  unsigned long long first_estimate = 0; /* pieces the set was first sized for */
  unsigned long long live_blocks_added = 0; /* blocks the last solve added under its kernels */
  int *d_resume = nullptr; /* device copy of resume_t */
  int *d_order_sub = nullptr; /* launch order of a relaunch: the unfinished problems */
  std::vector<int> resume_t;  /* per problem: data point to resume at (0: from the start) */
  int launches = 0;                     /* kernel launches of the last solve */
  unsigned long long steps_run = 0;     /* data points the last solve's launches worked through */
  hipStream_t stream2 = nullptr;
  hipEvent_t ev2 = nullptr;
  int *started = nullptr; /* pinned host word: latency-build workgroups of a mixed launch */
  bool mixed_wait_timed_out = false; /* the wait for them ran into its bound once: recorded, not repeated */
  /* the segment tables packed at their exact sizes (peakseg_hip_problem_set_pack_tables) */
  int *d_pack_start = nullptr;
  double *d_pack_mean = nullptr;
  long long *d_pack_rows = nullptr; /* per problem: first packed row, row count, source offset */
  long long pack_capacity = 0, pack_total = -1;
  int parks = 0;                          /* problems parked by the last solve's launches */
  unsigned long long park_pool_pieces = 0; /* ... and what they took from the overflow pool */
  std::vector<void *> allocs;
  unsigned long long bytes = 0;
};

namespace {

template <class T>
int dev_alloc(psd_problem_set *s, T **p, size_t n) {
  void *q = nullptr;
  size_t bytes = (n ? n : 1) * sizeof(T);
  hipError_t e = hipMalloc(&q, bytes);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return ERROR_DEVICE_MEMORY;
  }
  s->allocs.push_back(q);
  s->bytes += bytes;
  *p = (T *)q;
  return 0;
}

template <class T>
int dev_upload(psd_problem_set *s, const T **p, const std::vector<T> &v) {
  T *q = nullptr;
  int st = dev_alloc(s, &q, v.size());
  if (st) return st;
  if (!v.empty()) HIP_TRY(hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *p = q;
  return 0;
}

void forget_alloc(psd_problem_set *s, void *q) {
  for (size_t i = 0; i < s->allocs.size(); i++) {
    if (s->allocs[i] == q) {
      s->allocs.erase(s->allocs.begin() + (long)i);
      break;
    }
  }
}

unsigned long long env_bytes(const char *name) {
  const char *e = getenv(name);
  if (!e || !*e) return 0;
  char *end = nullptr;
  double v = strtod(e, &end);
  if (end == e || !(v > 0)) return 0;
  switch (*end) { /* optional K/M/G/T suffix */
    case 'k': case 'K': v *= 1024.0; break;
    case 'm': case 'M': v *= 1024.0 * 1024.0; break;
    case 'g': case 'G': v *= 1024.0 * 1024.0 * 1024.0; break;
    case 't': case 'T': v *= 1024.0 * 1024.0 * 1024.0 * 1024.0; break;
    default: break;
  }
  return (unsigned long long)v;
}

/* How many arena pieces may still be allocated: what the device has free (a tenth is left for
 * other processes sharing the GPU -- R's future workers are separate processes on one device,
 * SURVEY.md section 8b "Threading") and what PEAKSEG_HIP_MAX_BYTES leaves of this set's budget. */
unsigned long long arena_fit(psd_problem_set *s) {
  unsigned long long fit = ~0ull;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) fit = (unsigned long long)(free_b * 0.9) / 20ull;
  if (s->max_bytes) {
    unsigned long long room = s->max_bytes > s->bytes ? (s->max_bytes - s->bytes) / 20ull : 0ull;
    if (room < fit) fit = room;
  }
  return fit;
}


size_t arena_block_bytes(const psd_problem_set *s) { return (size_t)20 << s->d.ar_block_log2; }
unsigned long long arena_mapped(const psd_problem_set *s) {
  return (unsigned long long)s->arena_blocks.size() << s->d.ar_block_log2;
}

/* release the arena: every block, and the block tables */
void free_arena(psd_problem_set *s) {
  const size_t bytes = arena_block_bytes(s);
  for (auto &b : s->arena_blocks) {
#ifndef PSD_EMU
    if (b.vmm) {
      (void)hipMemUnmap(b.base, bytes);
      (void)hipMemRelease(b.handle);
      (void)hipMemAddressFree(b.base, bytes);
    } else
#endif
    {
      (void)hipFree(b.base);
    }
    s->bytes -= bytes;
  }
  s->arena_blocks.clear();
  if (s->d.ar_block) {
    forget_alloc(s, s->d.ar_block);
    (void)hipFree(s->d.ar_block);
    s->bytes -= s->table_cap * sizeof(char *);
    s->d.ar_block = nullptr;
  }
  if (s->h_live) (void)hipHostFree(s->h_live);
  if (s->h_used) (void)hipHostFree(s->h_used);
  s->h_live = s->h_used = nullptr;
  s->table_cap = s->table_synced = 0;
  s->d.ar_cap = 0;
  s->arena_pieces = 0;
}

/* One more block.  Safe while a kernel of this set runs: nothing the kernel uses is touched,
 * the block is published through the pinned words (address first, then the capacity). */
int arena_add_block(psd_problem_set *s) {
  if (s->arena_blocks.size() >= s->table_cap) {
    set_error("cost-function arena: block table of %zu entries is full", s->table_cap);
    return ERROR_DEVICE_MEMORY;
  }
  const size_t bytes = arena_block_bytes(s);
  psd_problem_set::ArenaBlock b;
  const char *what = "hipMalloc";
  hipError_t e = hipSuccess;
#ifndef PSD_EMU
  if (s->arena_vmm) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = s->device;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = s->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    b.vmm = true;
    what = "hipMemAddressReserve";
    e = hipMemAddressReserve(&b.base, bytes, (size_t)2 << 20, nullptr, 0);
    if (e == hipSuccess) {
      what = "hipMemCreate";
      e = hipMemCreate(&b.handle, bytes, &prop, 0);
      if (e == hipSuccess) {
        what = "hipMemMap";
        e = hipMemMap(b.base, bytes, 0, b.handle, 0);
        if (e == hipSuccess) {
          what = "hipMemSetAccess";
          e = hipMemSetAccess(b.base, bytes, &acc, 1);
          if (e != hipSuccess) (void)hipMemUnmap(b.base, bytes);
        }
        if (e != hipSuccess) (void)hipMemRelease(b.handle);
      }
      if (e != hipSuccess) (void)hipMemAddressFree(b.base, bytes);
    }
  } else
#endif
  {
    e = hipMalloc(&b.base, bytes);
  }
  if (e != hipSuccess) {
    set_error("cost-function arena: block %zu of %zu bytes failed in %s: %s", s->arena_blocks.size(),
              bytes, what, hipGetErrorString(e));
    (void)hipGetLastError();
    return ERROR_DEVICE_MEMORY;
  }
  s->arena_blocks.push_back(b);
  s->bytes += bytes;
  const size_t k = s->arena_blocks.size() - 1;
  __atomic_store_n(&s->h_live[2 + k], (unsigned long long)(uintptr_t)b.base, __ATOMIC_RELAXED);
  __atomic_store_n(&s->h_live[0], arena_mapped(s), __ATOMIC_RELEASE);
  return 0;
}

/* between launches: the device table learns the blocks added since, DeviceArgs the capacity */
int arena_sync_table(psd_problem_set *s) {
  const size_t n = s->arena_blocks.size();
  if (n > s->table_synced) {
    std::vector<char *> bases;
    for (size_t k = s->table_synced; k < n; k++) bases.push_back((char *)s->arena_blocks[k].base);
    HIP_TRY(hipMemcpy(s->d.ar_block + s->table_synced, bases.data(), bases.size() * sizeof(char *),
                      hipMemcpyHostToDevice));
    s->table_synced = n;
  }
  s->d.ar_cap = arena_mapped(s);
  s->arena_pieces = s->d.ar_cap;
  return 0;
}

/* Give the arena at least `pieces` pieces in all, never more than `limit` (0: no limit): the
 * first allocation, or growth between launches.  Growth adds blocks: no record moves, so a
 * solve that ran out of room is resumed, not repeated. */
int alloc_arena(psd_problem_set *s, unsigned long long pieces, unsigned long long limit = 0,
                bool starter_only = false) {
  const bool first = s->arena_blocks.empty() && s->d.ar_block == nullptr;
  const unsigned long long n_waves = 2ull * (unsigned long long)s->n_problems;
  if (first) {
    /* chunk size: about a sixteenth of what one wave will store, within [2^10, 2^16] pieces */
    int lg = psd::ARENA_CHUNK_LOG2_MIN;
    while (lg < psd::ARENA_CHUNK_LOG2_MAX && (pieces / n_waves) >> (lg + 5)) lg++;
    s->d.ar_chunk_log2 = lg;
    /* block size: about an eighth of the first estimate, within [2^19, 2^24] pieces (10 MB to
     * 336 MB; 2^19 pieces make the int array 2 MiB, the granularity the virtual-memory calls
     * accept on ROCm 7.2); the checkpointed store needs a wave's region inside one block */
    int blg = psd::ARENA_BLOCK_LOG2_MIN;
    while (blg < psd::ARENA_BLOCK_LOG2_MAX && (pieces >> (blg + 3))) blg++;
    bool small_blocks = false;
    if (const char *e = getenv("PEAKSEG_HIP_ARENA_BLOCK_LOG2")) {
      /* tests: blocks smaller than the virtual-memory granularity come from hipMalloc */
      const int v = atoi(e);
      if (v >= lg && v >= 10 && v <= psd::ARENA_BLOCK_LOG2_MAX) {
        blg = v;
        small_blocks = v < psd::ARENA_BLOCK_LOG2_MIN;
      }
    }
    (void)small_blocks;
    while (s->ckpt_interval > 0 && blg < psd::ARENA_BLOCK_LOG2_CKPT_MAX && (1ull << blg) < s->d.ckpt_region)
      blg++;
    s->d.ar_block_log2 = blg;
#ifndef PSD_EMU
    int vmm = 0;
    s->arena_vmm = !getenv("PEAKSEG_HIP_NO_VMM") && !small_blocks &&
                   hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported,
                                         s->device) == hipSuccess && vmm != 0;
#endif
    /* tables for every block the device's memory could hold */
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) total_b = (size_t)288 << 30;
    s->table_cap = (size_t)(((unsigned long long)total_b / 20ull) >> blg) + 8;
    int st = dev_alloc(s, &s->d.ar_block, s->table_cap);
    if (st) return st;
    hipError_t e = hipMemset(s->d.ar_block, 0, s->table_cap * sizeof(char *));
    if (e == hipSuccess)
      e = hipHostMalloc((void **)&s->h_live, (2 + s->table_cap) * sizeof(unsigned long long),
                        hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess)
      e = hipHostMalloc((void **)&s->h_used, 64, hipHostMallocCoherent | hipHostMallocMapped);
    if (e != hipSuccess) {
      set_error("cost-function arena: block tables: %s", hipGetErrorString(e));
      return ERROR_DEVICE_MEMORY;
    }
    memset(s->h_live, 0, (2 + s->table_cap) * sizeof(unsigned long long));
    s->h_live[1] = 1; /* final until a solve says otherwise */
    *s->h_used = 0;
    s->d.ar_live = s->h_live;
    s->d.ar_used = s->h_used;
    s->table_synced = 0;
  }
  const int blg = s->d.ar_block_log2;
  const unsigned long long chunk = 1ull << s->d.ar_chunk_log2;
  if (s->ckpt_interval > 0) {
    /* whole regions per block (the kernel's addressing, forward_body): the caller asks for
     * region x 2 x problems, the blocks hold a whole number of regions each */
    if ((1ull << blg) < s->d.ckpt_region) {
      set_error("checkpointed store: a region of %llu pieces exceeds the largest arena block (%llu)",
                s->d.ckpt_region, 1ull << blg);
      return ERROR_DEVICE_MEMORY;
    }
    const unsigned long long per_block = (1ull << blg) / s->d.ckpt_region;
    pieces = ((n_waves + per_block - 1) / per_block) << blg;
    if (limit && pieces > limit) {
      /* The kernel indexes region (2 p + chain) without looking at ar_cap: an arena clipped to
       * what fits would be written beyond its end.  The regions either fit or the set does not. */
      set_error("checkpointed store: %llu pieces per region x %llu regions (%llu bytes) do not "
                "fit (free HBM / PEAKSEG_HIP_MAX_BYTES)", s->d.ckpt_region, n_waves, pieces * 20ull);
      return ERROR_DEVICE_MEMORY;
    }
  } else {
    /* at least two chunks per wave so that nobody starves at start-up */
    const unsigned long long min_pieces = chunk * 2ull * n_waves;
    /* (live growth: just that and two blocks of headroom; the rest comes under the kernel) */
    if (starter_only) pieces = min_pieces + (2ull << blg);
    if (pieces < min_pieces) pieces = min_pieces;
  }
  unsigned long long want_blocks = (pieces + (1ull << blg) - 1) >> blg;
  if (limit) {
    /* a limit counts whole blocks, rounded DOWN: the cap is a hard bound for processes that
     * share a GPU (one block at least: without it there is no arena) */
    unsigned long long most = limit >> blg;
    if (most < 1) most = 1;
    if (want_blocks > most) want_blocks = most;
  }
  int st = 0;
  const size_t before = s->arena_blocks.size();
  while (s->arena_blocks.size() < want_blocks) {
    st = arena_add_block(s);
#ifndef PSD_EMU
    if (st && first && s->arena_vmm && s->arena_blocks.empty()) {
      /* the virtual-memory calls refuse on this system: plain allocations from here on */
      s->arena_vmm = false;
      continue;
    }
#endif
    if (st) break;
  }
  if (st && s->arena_blocks.size() == before) return st; /* keep what could be added otherwise */
  return arena_sync_table(s);
}

void free_spill(psd_problem_set *s) {
  void *ptrs[2] = {s->d.spill_f64, s->d.spill_i32};
  for (void *q : ptrs) {
    if (!q) continue;
    for (size_t i = 0; i < s->allocs.size(); i++) {
      if (s->allocs[i] == q) {
        s->allocs.erase(s->allocs.begin() + (long)i);
        break;
      }
    }
    (void)hipFree(q);
  }
  s->bytes -= (unsigned long long)s->spill_slots * (unsigned long long)s->d.spill_cap * (48ull * 8 + 12ull * 4);
  s->d.spill_f64 = nullptr;
  s->d.spill_i32 = nullptr;
  s->spill_slots = 0;
  s->d.spill_slots = 0;
}

/* Pool of HBM spill slots for problems whose functions outgrow LDS (adversarial data): 432
 * bytes per piece of capacity per slot; a problem takes a slot on its first overflow. */
int alloc_spill(psd_problem_set *s, int slots) {
  if (slots > s->n_problems) slots = s->n_problems;
  if (slots < 1) slots = 1;
  const size_t cap = (size_t)s->d.spill_cap;
  int st;
  if (cap == 0) return 0;
  if ((st = dev_alloc(s, &s->d.spill_f64, (size_t)slots * 48 * cap)) ||
      (st = dev_alloc(s, &s->d.spill_i32, (size_t)slots * 12 * cap)))
    return st;
  s->spill_slots = slots;
  s->d.spill_slots = slots;
  return 0;
}

void free_ckpt_overflow(psd_problem_set *s) {
  void *ptrs[2] = {s->d.ckpt_ovf_f64, s->d.ckpt_ovf_i32};
  for (void *q : ptrs) {
    if (!q) continue;
    for (size_t i = 0; i < s->allocs.size(); i++) {
      if (s->allocs[i] == q) {
        s->allocs.erase(s->allocs.begin() + (long)i);
        break;
      }
    }
    (void)hipFree(q);
  }
  s->bytes -= s->d.ckpt_ovf_cap * 52ull;
  s->d.ckpt_ovf_f64 = nullptr;
  s->d.ckpt_ovf_i32 = nullptr;
  s->d.ckpt_ovf_cap = 0;
}

/* Overflow pool of the checkpointed store: checkpoints of functions too long for a slot. */
int alloc_ckpt_overflow(psd_problem_set *s, unsigned long long pieces) {
  if (pieces < 1024) pieces = 1024;
  int st;
  if ((st = dev_alloc(s, &s->d.ckpt_ovf_f64, (size_t)pieces * 6)) ||
      (st = dev_alloc(s, &s->d.ckpt_ovf_i32, (size_t)pieces)))
    return st;
  s->d.ckpt_ovf_cap = pieces;
  return 0;
}

/* Full store: a larger overflow pool that KEEPS what parked problems have in it (they read it
 * back when they are resumed). */
int grow_ckpt_overflow_keep(psd_problem_set *s, unsigned long long pieces) {
  double *f64 = nullptr;
  int *i32 = nullptr;
  int st;
  if ((st = dev_alloc(s, &f64, (size_t)pieces * 6)) || (st = dev_alloc(s, &i32, (size_t)pieces))) {
    if (f64) {
      forget_alloc(s, f64);
      (void)hipFree(f64);
      s->bytes -= pieces * 48ull;
    }
    return st;
  }
  const size_t old = (size_t)s->d.ckpt_ovf_cap;
  /* a function of n pieces at offset off: 6 n doubles from 6 off, n ints from off -- offsets
   * are positions, not sizes, so the arrays are copied as they are */
  HIP_TRY(hipMemcpy(f64, s->d.ckpt_ovf_f64, old * 6 * sizeof(double), hipMemcpyDeviceToDevice));
  HIP_TRY(hipMemcpy(i32, s->d.ckpt_ovf_i32, old * sizeof(int), hipMemcpyDeviceToDevice));
  /* (device-to-device copies may return before they have run, and the set's streams do not wait
   * for the null stream) */
  HIP_TRY(hipStreamSynchronize((hipStream_t) nullptr));
  free_ckpt_overflow(s);
  s->d.ckpt_ovf_f64 = f64;
  s->d.ckpt_ovf_i32 = i32;
  s->d.ckpt_ovf_cap = pieces;
  return 0;
}

int env_device() {
  /* which GPU the file-level entry points use: PEAKSEG_HIP_DEVICE (one process per GPU sets it
   * from its rank); default 0 */
  if (const char *e = getenv("PEAKSEG_HIP_DEVICE")) {
    int d = atoi(e);
    if (d >= 0) return d;
  }
  return 0;
}

}  // namespace

extern "C" int peakseg_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char *peakseg_hip_last_error(void) { return g_last_error.c_str(); }
extern "C" const char *peakseg_hip_last_warning(void) { return g_last_warning.c_str(); }

/* shader clock of a device in kHz (0 when unknown): bench.py turns kernel time into cycles per
 * data point with it */
extern "C" int peakseg_hip_device_clock_khz(int device) {
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, device) != hipSuccess) return 0;
  return khz;
}

extern "C" void peakseg_hip_set_print(void (*print)(const char *)) { g_print = print; }

extern "C" void peakseg_hip_problem_set_destroy(psd_problem_set *s) {
  if (!s) return;
  free_arena(s);
  for (void *q : s->allocs) (void)hipFree(q);
  for (auto &e : s->ev)
    if (e) (void)hipEventDestroy(e);
  if (s->ev2) (void)hipEventDestroy(s->ev2);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  if (s->stream2) (void)hipStreamDestroy(s->stream2);
  if (s->started) (void)hipHostFree(s->started);
  delete s;
}

extern "C" const char *peakseg_hip_problem_set_kernel_build(psd_problem_set *s) {
  if (s->throughput && s->n_lat_mixed > 0) return s->packed ? "lat+pk" : "lat+thr";
  return s->throughput ? (s->packed ? "pk" : "thr") : "lat";
}

extern "C" int peakseg_hip_problem_set_solve_stats(psd_problem_set *s, int *launches,
                                                   unsigned long long *steps_run) {
  if (!s) return -1;
  if (launches) *launches = s->launches;
  if (steps_run) *steps_run = s->steps_run;
  return 0;
}

extern "C" int peakseg_hip_problem_set_arena_stats(psd_problem_set *s,
                                                   unsigned long long *block_pieces, int *blocks,
                                                   int *blocks_added_live) {
  if (!s) return -1;
  if (block_pieces) *block_pieces = 1ull << s->d.ar_block_log2;
  if (blocks) *blocks = (int)s->arena_blocks.size();
  if (blocks_added_live) *blocks_added_live = (int)s->live_blocks_added;
  return 0;
}

extern "C" unsigned long long peakseg_hip_problem_set_bytes(psd_problem_set *s) {
  return s ? s->bytes : 0;
}

extern "C" unsigned long long peakseg_hip_problem_set_arena_bytes_used(psd_problem_set *s) {
  return s ? s->arena_used * 20ull : 0;
}

extern "C" int peakseg_hip_problem_set_set_penalty(psd_problem_set *s, int p, double penalty) {
  if (!s || p < 0 || p >= s->n_problems) return -1;
  if (hipSetDevice(s->device) != hipSuccess) return -1;
  s->prob_penalty[(size_t)p] = penalty;
  if (hipMemcpyAsync(const_cast<double *>(s->d.prob_penalty) + p, &penalty, sizeof(double),
                     hipMemcpyHostToDevice, s->stream) != hipSuccess ||
      hipStreamSynchronize(s->stream) != hipSuccess) { /* (in order with the set's launches) */
    set_error("penalty upload failed");
    return -1;
  }
  s->solved = false;
  return 0;
}

extern "C" int peakseg_hip_problem_set_create(int device, int n_contigs, const int *contig_n_bins,
                                              const int *const *contig_count,
                                              const int *const *contig_weight, int n_problems,
                                              const int *problem_contig,
                                              const double *problem_penalty,
                                              unsigned long long arena_pieces,
                                              psd_problem_set **out) {
  *out = nullptr;
  if (peakseg_hip_device_count() <= device) {
    set_error("no HIP device %d visible (this library has no CPU fallback)", device);
    return ERROR_NO_HIP_DEVICE;
  }
  if (n_contigs <= 0 || n_problems <= 0) {
    set_error("empty problem set");
    return ERROR_DEVICE_SOLVER;
  }
  HIP_TRY(hipSetDevice(device));
  /* PEAKSEG_HIP_TIMING=1: where the creation of a set spends its time, on stderr */
  const bool timing = getenv("PEAKSEG_HIP_TIMING") != nullptr;
  auto t_mark = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "peakseg_hip timing: create: %-26s %8.3f s\n", what,
            std::chrono::duration<double>(now - t_mark).count());
    t_mark = now;
  };
  psd_problem_set *s = new psd_problem_set();
  s->device = device;
  s->n_contigs = n_contigs;
  s->n_problems = n_problems;
  std::vector<int> count, weight;
  std::vector<double> min_lm(n_contigs), max_lm(n_contigs);
  long long off = 0;
  for (int c = 0; c < n_contigs; c++) {
    int n = contig_n_bins[c];
    if (n <= 0 || n >= (1 << 30)) {
      set_error("contig %d has %d bins", c, n);
      peakseg_hip_problem_set_destroy(s);
      return ERROR_DEVICE_SOLVER;
    }
    s->contig_n.push_back(n);
    s->contig_off.push_back(off);
    off += n;
    count.insert(count.end(), contig_count[c], contig_count[c] + n);
    weight.insert(weight.end(), contig_weight[c], contig_weight[c] + n);
    double mn = INFINITY, mx = -INFINITY;
    long long width_sum = 0;
    for (int i = 0; i < n; i++) { /* drv:198-204 */
      double log_data = psd_log((double)contig_count[c][i]);
      if (log_data < mn) mn = log_data;
      if (mx < log_data) mx = log_data;
      width_sum += contig_weight[c][i];
    }
    /* (the kernels divide by cumulated widths with the hardware's division, which is the IEEE
     * quotient for whole-number divisors below 2^48, peakseg_detmath.h; chromosome coordinates
     * are 32-bit, so this never triggers on a bedGraph file) */
    if (width_sum >= (1ll << 48)) {
      set_error("contig %d: the bin widths sum to 2^48 or more", c);
      peakseg_hip_problem_set_destroy(s);
      return ERROR_DEVICE_SOLVER;
    }
    min_lm[c] = mn;
    max_lm[c] = mx;
  }
  s->total_bins = off;
  lap("gather contigs, log range");
  long long dp_bins = 0;
  for (int p = 0; p < n_problems; p++) {
    int c = problem_contig[p];
    if (c < 0 || c >= n_contigs) {
      set_error("problem %d names contig %d", p, c);
      peakseg_hip_problem_set_destroy(s);
      return ERROR_DEVICE_SOLVER;
    }
    dp_bins += s->contig_n[(size_t)c];
  }
  /* Store mode.  Full store: every cost function's backtrack record stays in HBM, as the
   * reference keeps them on disk (about 24 + 40 P bytes per data point and penalty).
   * Checkpointed store: only a checkpoint every K data points and the
   * records of one block of K at a time; the decoding recomputes the blocks it walks through.
   * Chosen when the full store would not fit (free HBM / PEAKSEG_HIP_MAX_BYTES), or forced
   * with PEAKSEG_HIP_CHECKPOINT=K. */
  /* pieces per stored function the arena is first sized for.  Typical coverage data needs 2-14
   * (the 1e6 x 64 grid: 5.2 on average); an estimate that proves too small costs one more
   * launch, not a repeated solve (the arena grows in place, parked problems go on), so the
   * default no longer has to be generous: 7 instead of rounds 1-2's 16. */
  double per_fn = 7.0;
  bool per_fn_given = false;
  if (const char *e = getenv("PEAKSEG_HIP_PIECES_PER_FUNCTION")) {
    double v = atof(e);
    if (v >= 1.0) {
      per_fn = v;
      per_fn_given = true;
    }
  }
  s->max_bytes = env_bytes("PEAKSEG_HIP_MAX_BYTES");
  int K = 0;
  if (const char *e = getenv("PEAKSEG_HIP_CHECKPOINT")) K = atoi(e);
  if (K == 0 && arena_pieces == 0 && !getenv("PEAKSEG_HIP_NO_CHECKPOINT")) {
    /* what the full store needs at a typical 8 pieces per function, with the tables */
    const double need = (double)dp_bins * (2.0 * 8.0 * 20.0 + 16.0 + 12.0);
    size_t free_b = 0, total_b = 0;
    double room = 1e30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) room = (double)free_b * 0.9;
    if (s->max_bytes && (double)s->max_bytes < room) room = (double)s->max_bytes;
    if (need > room) K = 2048;
  }
  if (K < 0) K = 0;
  if (K > 0 && K < 16) K = 16;
  s->ckpt_interval = K;
  long long fn_off = 0, seg_off = 0, ckpt_off = 0;
  std::vector<long long> prob_ckpt_off;
  /* Full store: one park slot per problem (13 KB), so that a solve that runs out of arena is
   * resumed after the arena has grown instead of repeated (PEAKSEG_HIP_NO_PARK=1: as rounds
   * 1-2, rerun the set; sets of more than 16384 problems do without, too). */
  const bool park = K == 0 && n_problems <= 16384 && !getenv("PEAKSEG_HIP_NO_PARK");
  s->can_park = park;
  for (int p = 0; p < n_problems; p++) {
    int c = problem_contig[p];
    const long long n = s->contig_n[(size_t)c];
    s->prob_contig.push_back(c);
    s->prob_penalty.push_back(problem_penalty[p]);
    s->prob_fn_off.push_back(fn_off);
    s->prob_seg_off.push_back(seg_off);
    prob_ckpt_off.push_back(K > 0 ? ckpt_off : (park ? (long long)p : 0ll));
    fn_off += K > 0 ? 2ll * (K + 1) : 2ll * n;
    seg_off += n + 1;
    if (K > 0) ckpt_off += (n - 1) / K;
  }
  s->fn_total = fn_off;
  s->seg_total = seg_off;
  s->dp_bins = dp_bins;
  /* workgroups are dispatched in index order: start the longest problems first so that a
   * set of unequal contigs does not end with one long problem running alone */
  std::vector<int> order((size_t)n_problems);
  for (int p = 0; p < n_problems; p++) order[(size_t)p] = p;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
    return s->contig_n[(size_t)s->prob_contig[(size_t)x]] >
           s->contig_n[(size_t)s->prob_contig[(size_t)y]];
  });
  int st = 0;
  psd::DeviceArgs &d = s->d;
  d.n_problems = n_problems;
  if ((st = dev_upload(s, &d.prob_contig, s->prob_contig)) ||
      (st = dev_upload(s, &d.prob_penalty, s->prob_penalty)) ||
      (st = dev_upload(s, &d.prob_fn_off, s->prob_fn_off)) ||
      (st = dev_upload(s, &d.prob_seg_off, s->prob_seg_off)) ||
      (st = dev_upload(s, &d.prob_order, order)) ||
      (st = dev_upload(s, &d.contig_n, s->contig_n)) ||
      (st = dev_upload(s, &d.contig_off, s->contig_off)) ||
      (st = dev_upload(s, &d.contig_min_log_mean, min_lm)) ||
      (st = dev_upload(s, &d.contig_max_log_mean, max_lm)) ||
      (st = dev_upload(s, &d.count, count)) || (st = dev_upload(s, &d.weight, weight)) ||
      (st = dev_alloc(s, &d.result, (size_t)n_problems)) ||
      (st = dev_alloc(s, &d.ar_next_chunk, (size_t)1)) ||
      (st = dev_alloc(s, &d.spill_next, (size_t)1)) ||
      (st = dev_alloc(s, const_cast<psd::DeviceArgs **>(&d.self), (size_t)1)) ||
      (st = dev_alloc(s, &d.fn_ref, (size_t)fn_off)) ||
      (st = dev_alloc(s, &d.seg_start, (size_t)seg_off)) ||
      (st = dev_alloc(s, &d.seg_mean, (size_t)seg_off)) ||
      (st = dev_upload(s, &d.prob_ckpt_off, prob_ckpt_off))) {
    peakseg_hip_problem_set_destroy(s);
    return st;
  }
  lap("upload, tables");
  d.ckpt_interval = K;
  d.ckpt_cap = psd::lat::LDS_CAP;
  d.ckpt_region = 0;
  d.ckpt_f64 = nullptr;
  d.ckpt_i32 = nullptr;
  d.ckpt_ovf_f64 = nullptr;
  d.ckpt_ovf_i32 = nullptr;
  d.ckpt_ovf_cap = 0;
  d.ckpt_ovf_next = nullptr;
  d.prob_resume = nullptr;
  s->resume_t.assign((size_t)n_problems, 0);
  if ((st = dev_alloc(s, &s->d_order_sub, (size_t)n_problems))) {
    peakseg_hip_problem_set_destroy(s);
    return st;
  }
  if (park) {
    if ((st = dev_alloc(s, &s->d_resume, (size_t)n_problems))) {
      peakseg_hip_problem_set_destroy(s);
      return st;
    }
    HIP_TRY(hipMemset(s->d_resume, 0, sizeof(int) * (size_t)n_problems));
    d.prob_resume = s->d_resume;
  }
  if (K > 0 || park) {
    const size_t cap = (size_t)d.ckpt_cap;
    const size_t slots = K > 0 ? (size_t)(ckpt_off > 0 ? ckpt_off : 1) : (size_t)n_problems;
    if ((st = dev_alloc(s, &d.ckpt_f64, slots * (6 + 12 * cap))) ||
        (st = dev_alloc(s, &d.ckpt_i32, slots * (8 + 2 * cap))) ||
        (st = dev_alloc(s, &d.ckpt_ovf_next, (size_t)1)) ||
        /* checkpoints of functions with more than ckpt_cap pieces (adversarial data):
         * PEAKSEG_HIP_CKPT_OVERFLOW pieces to start with (default 2^18 = 13 MB), four times as
         * many and a rerun whenever that proves too small */
        (st = alloc_ckpt_overflow(s, env_bytes("PEAKSEG_HIP_CKPT_OVERFLOW")
                                         ? env_bytes("PEAKSEG_HIP_CKPT_OVERFLOW")
                                         : (K > 0 ? (1ull << 18) : (1ull << 16))))) {
      peakseg_hip_problem_set_destroy(s);
      return st;
    }
  }
  /* spill pool for functions that outgrow LDS (adversarial data): PEAKSEG_HIP_SPILL_CAP pieces
   * per list (default 16384, at most 32767: the interval table packs two indices into an int;
   * 0 disables spilling), PEAKSEG_HIP_SPILL_SLOTS slots to start with (default 16; the pool is
   * grown and the set rerun when more problems spill at once) */
  {
    int cap = 16384;
    if (const char *e = getenv("PEAKSEG_HIP_SPILL_CAP")) cap = atoi(e);
    if (cap > psd::SPILL_CAP_MAX) cap = psd::SPILL_CAP_MAX;
    if (cap <= psd::lat::LDS_CAP) cap = 0;
    d.spill_cap = cap;
    d.spill_f64 = nullptr;
    d.spill_i32 = nullptr;
    int slots = 16;
    if (const char *e = getenv("PEAKSEG_HIP_SPILL_SLOTS")) slots = atoi(e);
    if ((st = alloc_spill(s, slots))) {
      peakseg_hip_problem_set_destroy(s);
      return st;
    }
  }
#ifdef PSD_PROFILE
  if ((st = dev_alloc(s, &d.prof, (size_t)n_problems * 2 * psd::N_PROF))) {
    peakseg_hip_problem_set_destroy(s);
    return st;
  }
#else
  d.prof = nullptr;
#endif
  /* arena: the reference's store holds 2 functions per data point with, on typical coverage
   * data, 2-14 pieces each.  Sized from that estimate
   * (PEAKSEG_HIP_PIECES_PER_FUNCTION, default 7: growth is cheap, memory is not), never beyond
   * nine tenths of what is free on the device or what PEAKSEG_HIP_MAX_BYTES allows; solve()
   * maps more and resumes the parked problems if one reports PST_ARENA_FULL. */
  s->arena_auto = arena_pieces == 0;
  unsigned long long want = arena_pieces;
  unsigned long long first_limit = 0;
  if (K > 0) {
    /* one region per chain and problem: the records of K + 1 data points */
    /* The checkpointed store cannot park: a block whose records outgrow the region during the
     * decoding's recomputation costs the problem a second solve from its first data point
     * (measured with 14 per function: 42 of the 1536 problems of the scaled config 4, all at
     * large penalties, the set's time doubled).  So this estimate stays generous -- 32 per
     * function, the longest functions of the 1e6-1e7 grids have 25-27 -- and costs little:
     * regions are (K + 1) functions per chain, not the whole contig. */
    s->ckpt_pieces_per_fn = per_fn_given ? (unsigned long long)(per_fn * 2.0) : 32ull;
    if (s->ckpt_pieces_per_fn < 8) s->ckpt_pieces_per_fn = 8;
    d.ckpt_region = (unsigned long long)(K + 1) * s->ckpt_pieces_per_fn;
    want = d.ckpt_region * 2ull * (unsigned long long)n_problems;
    s->arena_auto = true;
  } else if (s->arena_auto) {
    want = (unsigned long long)((double)dp_bins * 2.0 * per_fn);
    unsigned long long fit = arena_fit(s);
    if (want > fit) want = fit;
    first_limit = fit;
  }
  if (s->max_bytes && s->bytes + want * 20ull > s->max_bytes) {
    set_error("problem set needs %llu bytes, PEAKSEG_HIP_MAX_BYTES allows %llu",
              s->bytes + want * 20ull, s->max_bytes);
    peakseg_hip_problem_set_destroy(s);
    return ERROR_DEVICE_MEMORY;
  }
  /* Full store sized by the library: the arena grows WHILE the kernel runs (the host maps blocks
   * ahead of what the waves have taken, solve()), so only the first few blocks are mapped here
   * -- getting memory costs 13-35 ms per GB on a GPU whose memory has been used before, and
   * that time now passes under the kernel instead of in front of it.  The estimate still picks
   * the chunk and block sizes.  PEAKSEG_HIP_NO_LIVE_GROWTH=1: everything the estimate asks for
   * is mapped here, and a solve that needs more parks, grows and resumes (as round 3 did). */
  s->live_growth = K == 0 && s->arena_auto && !getenv("PEAKSEG_HIP_NO_LIVE_GROWTH");
  s->first_estimate = want;
  lap("park slots, pools");
  if ((st = alloc_arena(s, want, first_limit, s->live_growth))) {
    peakseg_hip_problem_set_destroy(s);
    return st;
  }
  lap("first arena blocks");
  if (hipDeviceGetAttribute(&s->n_cu, hipDeviceAttributeMultiprocessorCount, device) !=
          hipSuccess ||
      s->n_cu <= 0)
    s->n_cu = 256;
  s->order = order;
  /* Streams that do NOT synchronise with the null stream: the virtual-memory calls that add an
   * arena block while a kernel runs wait for every stream the null stream waits for -- with
   * blocking streams the host would wait for the kernel that waits for the host (measured:
   * waves stalled for seconds until their bound, tools/vmm_block_probe.cpp cases F and G).
   * Nothing here relies on the null stream's implicit ordering: copies are either enqueued
   * on these streams or synchronous and issued after hipStreamSynchronize. */
#ifdef PSD_EMU
  hipError_t e = hipStreamCreate(&s->stream);
  if (e == hipSuccess) e = hipStreamCreate(&s->stream2);
#else
  hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  if (e == hipSuccess) {
    /* the latency-build part of a mixed launch must get its CUs before the packed part does */
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    e = hipStreamCreateWithPriority(&s->stream2, hipStreamNonBlocking, hi);
  }
#endif
  if (e == hipSuccess) e = hipEventCreate(&s->ev2);
  if (e == hipSuccess)
    e = hipHostMalloc((void **)&s->started, sizeof(int), hipHostMallocCoherent | hipHostMallocMapped);
  for (auto &ev : s->ev)
    if (e == hipSuccess) e = hipEventCreate(&ev);
  if (e != hipSuccess) {
    set_error("stream/event creation failed: %s", hipGetErrorString(e));
    peakseg_hip_problem_set_destroy(s);
    return ERROR_DEVICE_SOLVER;
  }
  s->results.resize((size_t)n_problems);
  /* the set's streams do not synchronise with the null stream (hipStreamNonBlocking, so that
   * arena blocks can be mapped under a running kernel): whatever the creation put on the null
   * stream -- the memsets of the tables above may return before they have run -- is complete
   * before a solve launches anything */
  if ((e = hipStreamSynchronize((hipStream_t) nullptr)) != hipSuccess) {
    set_error("creating the problem set: %s", hipGetErrorString(e));
    peakseg_hip_problem_set_destroy(s);
    return ERROR_DEVICE_SOLVER;
  }
  lap("streams, events");
  *out = s;
  return 0;
}

namespace {
/* While the kernels of a solve run: a second host thread maps arena blocks AHEAD of what the
 * waves have taken (ar_used, a pinned word the waves add to whenever they take chunks), up to
 * what the device and PEAKSEG_HIP_MAX_BYTES allow.  A wave that needs a block that is not there
 * yet waits for it (arena_take); when no more can come the thread says so and the wave parks its
 * problem.  Mapping a block is 13-35 ms per GB where the memory has been used before: at the
 * 1.3 GB/s the 64-penalty grid stores, or the 14 GB/s of a full chip, the thread keeps ahead. */
struct LiveGrower {
  psd_problem_set *s = nullptr;
  unsigned long long limit = 0; /* pieces the arena may reach */
  std::atomic<bool> stop{false};
  std::thread th;
  unsigned long long added = 0;

  void start(psd_problem_set *set, unsigned long long limit_pieces) {
    s = set;
    limit = limit_pieces;
    __atomic_store_n(&s->h_live[0], arena_mapped(s), __ATOMIC_RELAXED);
    __atomic_store_n(&s->h_live[1], 0ull, __ATOMIC_RELEASE);
    th = std::thread([this]() { run(); });
  }
  void run() {
    (void)hipSetDevice(s->device);
    const unsigned long long B = 1ull << s->d.ar_block_log2;
    bool final = false;
    while (!stop.load(std::memory_order_acquire)) {
      bool progressed = false;
      for (;;) {
        const unsigned long long used = __atomic_load_n(s->h_used, __ATOMIC_ACQUIRE);
        unsigned long long ahead = used / 8ull;
        if (ahead < 2ull * B) ahead = 2ull * B;
        if (final || arena_mapped(s) >= used + ahead) break;
        if (arena_mapped(s) + B > limit || arena_add_block(s) != 0) {
          final = true; /* the capacity published so far is all there will be */
          __atomic_store_n(&s->h_live[1], 1ull, __ATOMIC_RELEASE);
          break;
        }
        added++;
        progressed = true;
      }
      if (!progressed) std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
  }
  void finish() {
    if (!th.joinable()) return;
    stop.store(true, std::memory_order_release);
    th.join();
    __atomic_store_n(&s->h_live[1], 1ull, __ATOMIC_RELEASE);
  }
  ~LiveGrower() { finish(); }
};
}  // namespace

/* what the mixed-launch planner assumes a problem advances at, refreshed by every solve that
 * ran on one build alone with every problem resident from the start (a clean measurement).
 * Defaults: an MI355X on contigs of 1e5-1e6 bins -- 96 k data points/s on the latency build;
 * 63 k on the throughput build with the chip full, 75 k with a CU to itself
 * (profiles/r04/thr_rate_long_contigs.log; rounds 2-3 assumed 27 k, the round-2 build's rate),
 * taken a little low: a problem the planner leaves on the throughput build must not end after
 * the longest one on the latency build. */
static std::atomic<double> g_lat_rate{96e3}, g_thr_rate{58e3};
/* a problem on the packed build (a SIMD shared three ways) against one on the throughput build
 * (two ways): 6144 equal problems ran 18.2 % faster six to a CU than four to a CU */
constexpr double PK_RATE_OF_THR = 1.182 * 4.0 / 6.0;
/* The planner cannot know how long a set's functions get.  When the packed build had to hand
 * more than one problem in twenty to the wider builds (each of them waited for the end of the
 * first launch before it went on), this process's later sets -- more of the same data, as a
 * rule -- are planned without it. */
static std::atomic<int> g_pk_handed_over_many{0};

extern "C" int peakseg_hip_problem_set_solve(psd_problem_set *s, float *forward_ms,
                                             float *backtrack_ms) {
  HIP_TRY(hipSetDevice(s->device));
  g_last_warning.clear();
  /* the latency build wants a CU per problem: beyond that, problems would queue behind each
   * other and a build that packs several problems on a CU (throughput: 4, packed: 6) finishes
   * the set sooner.  PEAKSEG_HIP_VARIANT=lat|thr|pk overrides (tests, A/B runs). */
  s->throughput = s->n_problems > s->n_cu;
  s->packed = false;
  s->widened = 0;
  bool forced = false;
  if (const char *e = getenv("PEAKSEG_HIP_VARIANT")) {
    if (!strcmp(e, "lat")) s->throughput = false, forced = true;
    if (!strcmp(e, "thr")) s->throughput = true, forced = true;
    if (!strcmp(e, "pk")) s->throughput = true, s->packed = true, forced = true;
  }
  /* the packed build hands functions of more than 40 pieces to the throughput build through
   * the park slots; without them (checkpointed store, very large sets) it is not used */
  if (s->packed && !s->can_park) s->packed = false;
  /* Mixed launch for sets of unequal contigs that oversubscribe the chip: a problem on the
   * throughput build advances about 60 k data points per second, on the latency build (a CU of
   * its own) about 96 k, so the longest problems would decide when the set ends.  The L longest
   * problems go to the latency build -- launched first, on a stream of its own, one CU each --
   * and the rest is packed on what is left; L minimises the later of the two predicted ends.
   * (Equal contigs: L = 0.)  The same prediction chooses between the throughput and the packed
   * build for the rest: six problems per CU at 0.79 of the speed each (measured on 6144 equal
   * problems: +18 %; on 24 unequal contigs x 64 penalties, which end with their longest packed
   * problems: -10 %, profiles/r04/ab_thr_occupancy_*.log). */
  s->n_lat_mixed = 0;
  if (s->throughput && !forced) {
    /* data points per second of one problem on either build: measured by this process's own
     * earlier solves when there were any (g_lat_rate / g_thr_rate below), else the figures of
     * an MI355X at 2.4 GHz */
    double lat_rate = g_lat_rate.load(), thr_rate = g_thr_rate.load();
    if (const char *e = getenv("PEAKSEG_HIP_RATES")) { /* diagnostic: "lat,thr" data points per s */
      double a = 0.0, b = 0.0;
      if (sscanf(e, "%lf,%lf", &a, &b) == 2 && a > 0.0 && b > 0.0) lat_rate = a, thr_rate = b;
    }
    std::vector<double> len((size_t)s->n_problems);
    double rest = 0.0;
    for (int k = 0; k < s->n_problems; k++) {
      len[(size_t)k] = (double)s->contig_n[(size_t)s->prob_contig[(size_t)s->order[(size_t)k]]];
      rest += len[(size_t)k];
    }
    const int l_max = std::max(0, s->n_cu - 16 < s->n_problems ? s->n_cu - 16 : s->n_problems - 1);
    /* the predicted end of the set with problems [0, l) on the latency build and [l, n) packed
     * per_cu to a CU at `rate` each, minimised over l */
    auto plan = [&](double per_cu, double rate, int &best_l) -> double {
      double best = 1e300, sum_lat = 0.0;
      best_l = 0;
      for (int l = 0; l <= l_max; l++) {
        const double t_lat = l > 0 ? len[0] / lat_rate : 0.0;
        const double cus = (double)(s->n_cu - l);
        const double t_work = (rest - sum_lat) / (cus * per_cu * rate);
        const double t_long = len[(size_t)l] / rate;
        const double t = std::max(t_lat, std::max(t_work, t_long));
        if (t < best * 0.98) { /* prefer fewer latency problems unless it clearly pays */
          best = t;
          best_l = l;
        }
        sum_lat += len[(size_t)l];
      }
      return best;
    };
    int l_thr = 0, l_pk = 0;
    const double t_thr = plan(4.0, thr_rate, l_thr);
    const double t_pk = plan(6.0, thr_rate * PK_RATE_OF_THR, l_pk);
    s->packed = s->can_park && !getenv("PEAKSEG_HIP_NO_PACKED") &&
                !g_pk_handed_over_many.load() && t_pk < 0.97 * t_thr;
    s->n_lat_mixed = s->packed ? l_pk : l_thr;
    if (getenv("PEAKSEG_HIP_TIMING")) {
      const int L = s->n_lat_mixed;
      const double per_cu = s->packed ? 6.0 : 4.0;
      const double rate = s->packed ? thr_rate * PK_RATE_OF_THR : thr_rate;
      double on_lat = 0.0;
      for (int k = 0; k < L; k++) on_lat += len[(size_t)k];
      fprintf(stderr, "peakseg_hip timing: plan: %d problems, %d on the latency build (predicted end "
                      "%.2f s), the rest %s (work %.2f s, longest %.2f s); rates %.0f / %.0f per s; "
                      "all on thr %.2f s, all on pk %.2f s\n", s->n_problems, L,
              L > 0 ? len[0] / lat_rate : 0.0, s->packed ? "pk" : "thr",
              (rest - on_lat) / ((double)(s->n_cu - L) * per_cu * rate), len[(size_t)L] / rate,
              lat_rate, thr_rate, t_thr, t_pk);
    }
  }
  /* Launches.  The first one runs every problem.  When problems come back unfinished for
   * want of room (arena, spill pool, checkpoint overflow pool) the host enlarges what was short
   * and launches THOSE problems again: a problem that ran out of arena was parked by the kernel
   * and goes on at the data point it had reached (the arena grows by a segment, its records
   * stay in place); the others start over.  Finished problems are never computed twice. */
  std::vector<int> todo(s->order); /* launch order: longest contig first */
  std::fill(s->resume_t.begin(), s->resume_t.end(), 0);
  if (s->can_park) HIP_TRY(hipMemsetAsync(s->d_resume, 0, sizeof(int) * (size_t)s->n_problems, s->stream));
  HIP_TRY(hipMemsetAsync(s->d.ar_next_chunk, 0, sizeof(unsigned long long), s->stream));
  __atomic_store_n(s->h_used, 0ull, __ATOMIC_RELEASE); /* follows ar_next_chunk */
  s->launches = 0;
  s->steps_run = 0;
  s->live_blocks_added = 0;
  s->pack_total = -1;
  s->parks = 0;
  s->park_pool_pieces = 0;
  float total_ms = 0.f;
  hipEvent_t ev_packed_end = nullptr; /* (diagnostic, PEAKSEG_HIP_TIMING) */
  int arena_rounds = 0;               /* launches of this solve that ran out of arena */
  for (int attempt = 0;; attempt++) {
    if (s->ckpt_interval > 0) {
      const unsigned long long B = 1ull << s->d.ar_block_log2;
      const unsigned long long per_block = s->d.ckpt_region ? B / s->d.ckpt_region : 0ull;
      const unsigned long long n_regions = 2ull * (unsigned long long)s->n_problems;
      if (per_block == 0 || s->arena_blocks.size() < (n_regions + per_block - 1) / per_block) {
        set_error("checkpointed store: arena of %llu pieces is smaller than its %d regions of %llu",
                  s->d.ar_cap, 2 * s->n_problems, s->d.ckpt_region);
        return ERROR_DEVICE_MEMORY;
      }
    }
    const int n_todo = (int)todo.size();
    const bool relaunch = attempt > 0;
    /* the arena may grow under this launch's kernels */
    LiveGrower grower;
    const bool live = s->live_growth && s->ckpt_interval == 0 && s->arena_auto;
    s->d.ar_live = live ? s->h_live : nullptr;
    s->d.ar_used = s->h_used;
    psd::DeviceArgs d_run = s->d;
    if (relaunch) {
      /* only the unfinished problems, in their original order */
      HIP_TRY(hipMemcpyAsync(s->d_order_sub, todo.data(), sizeof(int) * (size_t)n_todo,
                             hipMemcpyHostToDevice, s->stream));
      d_run.prob_order = s->d_order_sub;
      if (s->can_park)
        HIP_TRY(hipMemcpyAsync(s->d_resume, s->resume_t.data(), sizeof(int) * (size_t)s->n_problems,
                               hipMemcpyHostToDevice, s->stream));
      d_run.n_problems = n_todo;
    }
    if (s->ckpt_interval > 0)
      HIP_TRY(hipMemsetAsync(s->d.ar_next_chunk, 0, sizeof(unsigned long long), s->stream));
    HIP_TRY(hipMemsetAsync(s->d.spill_next, 0, sizeof(int), s->stream));
    /* Overflow pool.  Checkpointed store: every relaunched problem starts over and saves its
     * checkpoints again, so the pool starts empty.  Full store: the pool holds the functions of
     * PARKED problems (those longer than a park slot) from the launch that parked them until the
     * workgroup that resumes them has read them back -- which, in a grid larger than the chip
     * holds resident, can be long after other workgroups of the same launch have parked again:
     * the pool is emptied once per solve, never between its launches. */
    if (s->d.ckpt_ovf_next && (s->ckpt_interval > 0 || !relaunch))
      HIP_TRY(hipMemsetAsync(s->d.ckpt_ovf_next, 0, sizeof(unsigned long long), s->stream));
    HIP_TRY(hipMemcpyAsync(const_cast<psd::DeviceArgs *>(s->d.self), &d_run, sizeof(psd::DeviceArgs),
                           hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipEventRecord(s->ev[0], s->stream));
    if (live) grower.start(s, arena_mapped(s) + arena_fit(s));
    const dim3 grid((unsigned)n_todo);
    /* (a relaunch after the packed build: the problems it parked need the wider lists of the
     * throughput build, or a CU each when they are few) */
    const bool thr_now = relaunch ? ((forced && !s->packed) ? s->throughput : n_todo > s->n_cu)
                                  : s->throughput;
    if (!relaunch && s->throughput && s->n_lat_mixed > 0) {
      /* mixed launch: both kernels index prob_order by their own blockIdx.x */
      const int L = s->n_lat_mixed;
      psd::DeviceArgs d_lat = s->d, d_thr = s->d;
      d_lat.n_problems = L;
      d_thr.n_problems = s->n_problems - L;
      d_thr.prob_order = s->d.prob_order + L;
      HIP_TRY(hipStreamWaitEvent(s->stream2, s->ev[0], 0));
      /* A latency-build workgroup needs every register of a CU: once the packed part has put
       * a workgroup on each CU it would find none free before the packed part has drained,
       * and the two kernels would run one after the other.  So its workgroups report in (one
       * system-scope atomic each, on a pinned host word) and the packed part is launched when
       * all of them have started -- or after half a second, whatever they are waiting for. */
      __atomic_store_n(s->started, 0, __ATOMIC_RELEASE);
      d_lat.started = s->started;
      if (s->ckpt_interval > 0)
        hipLaunchKernelGGL(psd::lat::fpop_forward_ckpt_kernel, dim3((unsigned)L),
                           dim3(psd::lat::FORWARD_THREADS), 0, s->stream2, d_lat);
      else
        hipLaunchKernelGGL(psd::lat::fpop_forward_kernel, dim3((unsigned)L),
                           dim3(psd::lat::FORWARD_THREADS), 0, s->stream2, d_lat);
      HIP_TRY(hipGetLastError());
      (void)hipStreamQuery(s->stream2); /* submit now */
      if (!s->mixed_wait_timed_out) { /* later solves after a time-out do not wait again */
        const auto t0 = std::chrono::steady_clock::now();
        while (__atomic_load_n(s->started, __ATOMIC_ACQUIRE) < L &&
               std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(500))
          std::this_thread::yield();
        if (__atomic_load_n(s->started, __ATOMIC_ACQUIRE) < L) {
          s->mixed_wait_timed_out = true;
          set_warning("mixed launch: %d of %d latency-build workgroups had not started after 0.5 s; "
                      "the packed part was launched anyway",
                      L - __atomic_load_n(s->started, __ATOMIC_ACQUIRE), L);
        }
      }
      if (s->ckpt_interval > 0)
        hipLaunchKernelGGL(psd::thr::fpop_forward_ckpt_kernel, dim3((unsigned)d_thr.n_problems),
                           dim3(psd::thr::FORWARD_THREADS), 0, s->stream, d_thr);
      else if (s->packed)
        hipLaunchKernelGGL(psd::pk::fpop_forward_kernel, dim3((unsigned)d_thr.n_problems),
                           dim3(psd::pk::FORWARD_THREADS), 0, s->stream, d_thr);
      else
        hipLaunchKernelGGL(psd::thr::fpop_forward_kernel, dim3((unsigned)d_thr.n_problems),
                           dim3(psd::thr::FORWARD_THREADS), 0, s->stream, d_thr);
      HIP_TRY(hipGetLastError());
      if (getenv("PEAKSEG_HIP_TIMING") && !ev_packed_end && hipEventCreate(&ev_packed_end) != hipSuccess)
        ev_packed_end = nullptr;
      if (ev_packed_end) HIP_TRY(hipEventRecord(ev_packed_end, s->stream));
      HIP_TRY(hipEventRecord(s->ev2, s->stream2));
      HIP_TRY(hipStreamWaitEvent(s->stream, s->ev2, 0));
    } else if (thr_now) {
      if (s->ckpt_interval > 0)
        hipLaunchKernelGGL(psd::thr::fpop_forward_ckpt_kernel, grid,
                           dim3(psd::thr::FORWARD_THREADS), 0, s->stream, d_run);
      else if (s->packed && !relaunch) /* (a relaunch holds what the packed build handed over) */
        hipLaunchKernelGGL(psd::pk::fpop_forward_kernel, grid, dim3(psd::pk::FORWARD_THREADS),
                           0, s->stream, d_run);
      else
        hipLaunchKernelGGL(psd::thr::fpop_forward_kernel, grid, dim3(psd::thr::FORWARD_THREADS),
                           0, s->stream, d_run);
    } else {
      if (s->ckpt_interval > 0)
        hipLaunchKernelGGL(psd::lat::fpop_forward_ckpt_kernel, grid,
                           dim3(psd::lat::FORWARD_THREADS), 0, s->stream, d_run);
      else
        hipLaunchKernelGGL(psd::lat::fpop_forward_kernel, grid, dim3(psd::lat::FORWARD_THREADS),
                           0, s->stream, d_run);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev[1], s->stream));
    {
      const hipError_t e_sync = hipStreamSynchronize(s->stream);
      grower.finish(); /* (before any return: the thread works on *s) */
      s->live_blocks_added += grower.added;
      HIP_TRY(e_sync);
      int st_tab = arena_sync_table(s);
      if (st_tab) return st_tab;
    }
    float f_ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&f_ms, s->ev[0], s->ev[1]));
    total_ms += f_ms;
    if (ev_packed_end) { /* PEAKSEG_HIP_TIMING: when each part of a mixed launch ended */
      float lat_ms = 0.f, packed_ms = 0.f;
      if (hipEventElapsedTime(&lat_ms, s->ev[0], s->ev2) == hipSuccess &&
          hipEventElapsedTime(&packed_ms, s->ev[0], ev_packed_end) == hipSuccess)
        fprintf(stderr, "peakseg_hip timing: mixed launch: latency part ended at %.2f s, packed part "
                        "at %.2f s\n", lat_ms / 1e3, packed_ms / 1e3);
      (void)hipEventDestroy(ev_packed_end);
      ev_packed_end = nullptr;
    }
    s->launches++;
    if (forward_ms) *forward_ms = total_ms;
    if (backtrack_ms) *backtrack_ms = 0.f; /* decoding happens inside the forward kernel */
    {
      /* results of the problems this launch ran (the others keep theirs) */
      std::vector<psd::ProbResult> all((size_t)s->n_problems);
      HIP_TRY(hipMemcpy(all.data(), s->d.result, sizeof(psd::ProbResult) * (size_t)s->n_problems,
                        hipMemcpyDeviceToHost));
      for (int p : todo) {
        const psd::ProbResult &r = all[(size_t)p];
        const int n = s->contig_n[(size_t)s->prob_contig[(size_t)p]];
        const int reached = r.status == 0 ? n : r.step_reached;
        if (reached > s->resume_t[(size_t)p])
          s->steps_run += (unsigned long long)(reached - s->resume_t[(size_t)p]);
        s->results[(size_t)p] = r;
      }
    }
    {
      unsigned long long chunks = 0;
      HIP_TRY(hipMemcpy(&chunks, s->d.ar_next_chunk, sizeof chunks, hipMemcpyDeviceToHost));
      s->arena_used = chunks << s->d.ar_chunk_log2;
      if (s->arena_used > s->d.ar_cap) s->arena_used = s->d.ar_cap;
    }
    bool arena_full = false, spill_full = false, ckpt_full = false, park_pool_full = false;
    int widen = 0; /* problems the packed build parked because a function outgrew its lists */
    int longest_function = 0;
    std::vector<int> again;
    for (int p : todo) {
      const psd::ProbResult &r = s->results[(size_t)p];
      arena_full = arena_full || r.status == psd::PST_ARENA_FULL;
      /* out of arena beyond data point 0 and not parked although the set has park slots: its
       * functions were too long for a slot and the overflow pool had no room for them */
      park_pool_full = park_pool_full || (r.status == psd::PST_ARENA_FULL && s->can_park &&
                                          !r.parked && r.step_reached > 0);
      spill_full = spill_full || r.status == psd::PST_SPILL_FULL;
      ckpt_full = ckpt_full || r.status == psd::PST_CKPT_FULL;
      if (r.max_intervals > longest_function) longest_function = r.max_intervals;
      const bool to_wider = r.status == psd::PST_LDS_OVERFLOW && r.parked && s->can_park;
      if (r.status == psd::PST_ARENA_FULL || r.status == psd::PST_SPILL_FULL ||
          r.status == psd::PST_CKPT_FULL || to_wider) {
        again.push_back(p);
        if (r.status == psd::PST_ARENA_FULL && r.parked && s->can_park) s->parks++;
        if (to_wider) widen++;
        /* parked: go on where it stopped; anything else starts over */
        s->resume_t[(size_t)p] = ((r.status == psd::PST_ARENA_FULL || to_wider ||
                                   r.status == psd::PST_SPILL_FULL) &&
                                  r.parked && s->can_park)
                                     ? r.step_reached
                                     : 0;
      }
    }
    if (s->ckpt_interval == 0 && s->d.ckpt_ovf_next) { /* the parks' share of the overflow pool */
      unsigned long long used = 0;
      HIP_TRY(hipMemcpy(&used, s->d.ckpt_ovf_next, sizeof used, hipMemcpyDeviceToHost));
      s->park_pool_pieces = used;
    }
    s->widened += widen;
    if (again.empty()) break;
    if (getenv("PEAKSEG_HIP_TIMING")) {
      fprintf(stderr, "peakseg_hip timing: launch %d: %d of %d problems unfinished (arena %d, spill "
                      "pool %d, checkpoint pool %d, to wider lists %d):", s->launches,
              (int)again.size(), n_todo, (int)arena_full, (int)spill_full, (int)ckpt_full, widen);
      for (size_t k = 0; k < again.size() && k < 8; k++)
        fprintf(stderr, " p%d@%d%s", again[k], s->results[(size_t)again[k]].step_reached,
                s->resume_t[(size_t)again[k]] ? "(parked)" : "");
      fprintf(stderr, "\n");
    }
    if (attempt >= 12) {
      set_error("cost-function arena (%llu pieces) / spill pool (%d slots) still too small after "
                "%d relaunches", s->arena_pieces, s->spill_slots, attempt);
      return ERROR_DEVICE_MEMORY;
    }
    if (park_pool_full && s->ckpt_interval == 0) {
      /* such problems start over this time; with four times the pool (what other parked
       * problems keep in it is preserved) the next exhaustion parks them */
      const unsigned long long bigger = s->d.ckpt_ovf_cap * 4ull;
      if (bigger <= arena_fit(s) * 20ull / 52ull) {
        int st = grow_ckpt_overflow_keep(s, bigger);
        if (st) return st;
      }
    }
    if (ckpt_full) {
      unsigned long long bigger = s->d.ckpt_ovf_cap * 4ull;
      free_ckpt_overflow(s);
      const unsigned long long fit = arena_fit(s) * 20ull / 52ull;
      if (bigger > fit) {
        set_error("checkpointed store: an overflow pool of %llu pieces does not fit (free HBM / "
                  "PEAKSEG_HIP_MAX_BYTES)", bigger);
        return ERROR_DEVICE_MEMORY;
      }
      int st = alloc_ckpt_overflow(s, bigger);
      if (st) return st;
    }
    if (spill_full) {
      /* more problems spilled at once than the pool has slots: four times the slots */
      int slots = s->spill_slots * 4;
      if (s->spill_slots >= s->n_problems) {
        set_error("spill pool exhausted with one slot per problem");
        return ERROR_DEVICE_SOLVER;
      }
      free_spill(s);
      int st = alloc_spill(s, slots);
      if (st) return st;
    }
    if (arena_full) {
      if (!s->arena_auto) {
        set_error("cost-function arena of %llu pieces is too small", s->arena_pieces);
        return ERROR_DEVICE_MEMORY;
      }
      if (s->ckpt_interval > 0) {
        /* checkpointed store: a block's records outgrew a wave's region -- twice the region,
         * or at once what the longest function of the forward pass asks for (K + 1 functions of
         * that length always fit then) when that is more.  The regions are scratch for the
         * decoding's recomputation: re-allocated, never kept. */
        const unsigned long long old_ppf = s->ckpt_pieces_per_fn;
        s->ckpt_pieces_per_fn *= 2ull;
        if (s->ckpt_pieces_per_fn < (unsigned long long)longest_function)
          s->ckpt_pieces_per_fn = (unsigned long long)longest_function;
        s->d.ckpt_region = (unsigned long long)(s->ckpt_interval + 1) * s->ckpt_pieces_per_fn;
        const unsigned long long bigger = s->d.ckpt_region * 2ull * (unsigned long long)s->n_problems;
        free_arena(s);
        const unsigned long long fit = arena_fit(s);
        int st = alloc_arena(s, bigger, fit); /* (refuses, never clips, what does not fit) */
        if (st) {
          const std::string why = g_last_error;
          s->ckpt_pieces_per_fn = old_ppf;
          s->d.ckpt_region = (unsigned long long)(s->ckpt_interval + 1) * s->ckpt_pieces_per_fn;
          if (s->arena_blocks.empty() && s->d.ar_block == nullptr)
            (void)alloc_arena(s, s->d.ckpt_region * 2ull * (unsigned long long)s->n_problems, fit);
          g_last_error = why;
          return st;
        }
      } else {
        /* Full store: the arena GROWS by whole blocks (problems come here parked when the live
         * growth could not keep up or was switched off).  How much more: what the unfinished
         * problems' progress says the rest of the set needs (pieces handed out so far x data
         * points left / data points done, x 1.3); at least a quarter of what the arena has. */
        /* (a problem that could not be parked starts over and stores all its records again;
         * those of its first attempt stay where they are, unused) */
        double done = 0.0, rest = 0.0;
        for (int p = 0; p < s->n_problems; p++) {
          const double n = (double)s->contig_n[(size_t)s->prob_contig[(size_t)p]];
          const psd::ProbResult &r = s->results[(size_t)p];
          done += r.status == 0 ? n : (double)r.step_reached;
          if (r.status != 0) rest += n - (double)s->resume_t[(size_t)p];
        }
        /* (functions that get longer with t -- adversarial counts -- need more per data point
         * the further they get: the linear estimate falls short every time, so from the second
         * exhaustion of a solve on the arena at least doubles) */
        arena_rounds++;
        unsigned long long more = arena_rounds >= 2 ? s->arena_pieces : s->arena_pieces / 4ull;
        if (done > 0.0) {
          const double left = (double)(s->d.ar_cap - s->arena_used);
          const double need = (double)s->arena_used / done * rest * 1.3 - left;
          if (need > (double)more) more = (unsigned long long)need;
        }
        /* (at least four chunks for every problem that comes back: a set that ran out with
         * all its problems nearly done estimates less than their next requests take) */
        const unsigned long long fit = arena_fit(s);
        const unsigned long long chunk = 1ull << s->d.ar_chunk_log2;
        const unsigned long long least = chunk * 4ull * (unsigned long long)again.size();
        if (more < least) more = least;
        if (more > fit) more = fit;
        if (more < least) {
          set_error("cost-function arena cannot grow beyond %llu pieces (free HBM / "
                    "PEAKSEG_HIP_MAX_BYTES)", s->arena_pieces);
          return ERROR_DEVICE_MEMORY;
        }
        int st = alloc_arena(s, s->arena_pieces + more, s->arena_pieces + fit);
        if (st) return st;
      }
    }
    todo.swap(again);
  }
  if (s->packed && !forced && (long long)s->widened * 20 > (long long)s->n_problems)
    g_pk_handed_over_many.store(1);
#ifndef PSD_EMU /* (the emulator's timings say nothing about the hardware) */
  if (s->launches == 1 && s->n_lat_mixed == 0 && s->ckpt_interval == 0 && total_ms > 500.f) {
    /* a clean single launch: the longest problem's data points / kernel time is the rate of
     * that build (the throughput build only while every workgroup was resident at once) */
    int longest = 0;
    bool spilled = false;
    for (int p = 0; p < s->n_problems; p++) {
      longest = std::max(longest, s->contig_n[(size_t)s->prob_contig[(size_t)p]]);
      spilled = spilled || s->results[(size_t)p].spill_steps > 0;
    }
    const double rate = (double)longest / ((double)total_ms / 1e3);
    if (!spilled && rate > 1e3 && rate < 1e7) {
      /* latency build: a CU per problem; throughput build: only a chip that was full (four
       * workgroups on nearly every CU) shows the packed rate the planner reasons with */
      if (!s->throughput && s->n_problems <= s->n_cu) g_lat_rate.store(rate);
      if (s->throughput && !s->packed && s->n_problems <= 4 * s->n_cu &&
          s->n_problems >= 7 * s->n_cu / 2)
        g_thr_rate.store(rate);
    }
  }
#endif
  s->solved = true;
  int first = 0;
  for (int p = 0; p < s->n_problems; p++) {
    const psd::ProbResult &r = s->results[(size_t)p];
    if (r.status != 0 && first == 0) {
      set_error("problem %d: kernel status %d (wave error bits %d) at data point %d", p, r.status,
                r.wave_err, r.step_reached);
      first = ERROR_DEVICE_SOLVER;
    }
  }
  return first;
}

extern "C" int peakseg_hip_problem_set_result(psd_problem_set *s, int p, psd_result *out) {
  if (!s || !s->solved || p < 0 || p >= s->n_problems) return -1;
  const psd::ProbResult &r = s->results[(size_t)p];
  out->status = r.status == 0 ? 0 : ERROR_DEVICE_SOLVER;
  out->kernel_status = r.status;
  out->n_segments = r.n_segments;
  out->n_peaks = (r.n_segments - 1) / 2;
  out->n_equality_constraints = r.n_equality;
  out->max_intervals = r.max_intervals;
  out->total_intervals = r.total_intervals;
  out->best_cost = r.best_cost;
  out->n_serial_env = r.n_serial_env;
  out->step_reached = r.step_reached;
  out->spill_steps = r.spill_steps;
  return 0;
}

extern "C" int peakseg_hip_problem_set_segments(psd_problem_set *s, int p, int capacity,
                                                int *seg_start, double *seg_mean) {
  if (!s || !s->solved || p < 0 || p >= s->n_problems) return -1;
  const psd::ProbResult &r = s->results[(size_t)p];
  if (r.status != 0 || r.n_segments > capacity) return -1;
  size_t n = (size_t)r.n_segments;
  long long off = s->prob_seg_off[(size_t)p];
  if (hipMemcpy(seg_start, s->d.seg_start + off, n * sizeof(int), hipMemcpyDeviceToHost) !=
          hipSuccess ||
      hipMemcpy(seg_mean, s->d.seg_mean + off, n * sizeof(double), hipMemcpyDeviceToHost) !=
          hipSuccess) {
    set_error("segment table download failed");
    return -1;
  }
  return r.n_segments;
}

extern "C" int peakseg_hip_problem_set_checkpoint_interval(psd_problem_set *s) {
  return s ? s->ckpt_interval : 0;
}

extern "C" int peakseg_hip_problem_set_export_db(psd_problem_set *s, int p, const int *chromEnd,
                                                 const char *path) {
  if (!s || !s->solved || p < 0 || p >= s->n_problems) return -1;
  if (s->ckpt_interval > 0) return -1; /* checkpointed store: the functions were not all kept */
  int N = s->contig_n[(size_t)s->prob_contig[(size_t)p]];
  std::vector<unsigned long long> ref((size_t)2 * N);
  if (hipMemcpy(ref.data(), s->d.fn_ref + s->prob_fn_off[(size_t)p], ref.size() * 8,
                hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  /* table of 2N std::streampos (16 bytes: offset + zero state), then records in the order
   * the reference writes them: down_0, then (down_t, up_t) for t >= 1 (drv:388-392) */
  std::vector<long long> table((size_t)4 * N, 0);
  long long pos = 32ll * N;
  std::vector<char> body;
  /* the records are read from host copies of the arena's blocks (two at a time: the two chains
   * of a problem write into different chunk runs), not with three small copies per function */
  const int blg = s->d.ar_block_log2;
  const size_t block_bytes = arena_block_bytes(s);
  struct HostBlock {
    size_t blk = (size_t)-1;
    std::vector<char> bytes;
    unsigned long long used = 0;
  } cache[2];
  unsigned long long tick = 0;
  auto host_block = [&](size_t blk) -> const char * {
    for (auto &c : cache)
      if (c.blk == blk) {
        c.used = ++tick;
        return c.bytes.data();
      }
    HostBlock &c = cache[0].used <= cache[1].used ? cache[0] : cache[1];
    if (blk >= s->arena_blocks.size()) return nullptr;
    c.bytes.resize(block_bytes);
    if (hipMemcpy(c.bytes.data(), s->arena_blocks[blk].base, block_bytes, hipMemcpyDeviceToHost) !=
        hipSuccess)
      return nullptr;
    c.blk = blk;
    c.used = ++tick;
    return c.bytes.data();
  };
  for (int t = 0; t < N; t++) {
    for (int which = 0; which < 2; which++) {
      int element = which == 0 ? N + t : t;
      if (which == 1 && t == 0) continue;
      unsigned long long r = ref[(size_t)element];
      unsigned long long off = r >> psd::FN_COUNT_BITS;
      int n = (int)(r & ((1ull << psd::FN_COUNT_BITS) - 1));
      /* the record's block and its three arrays (fpop_types.h) */
      const size_t w = (size_t)(off & ((1ull << blg) - 1ull));
      const char *bb = host_block((size_t)(off >> blg));
      if (!bb || w + (size_t)n > ((size_t)1 << blg)) {
        fclose(f);
        return -1;
      }
      const double *mx = (const double *)bb + w;
      const double *prv = (const double *)(bb + ((size_t)8 << blg)) + w;
      const int *di = (const int *)(bb + ((size_t)16 << blg)) + w;
      table[(size_t)2 * element] = pos;
      int size = 20 * n + 8;
      size_t at = body.size();
      body.resize(at + 4 + (size_t)size);
      char *q = body.data() + at;
      memcpy(q, &size, 4);
      q += 4;
      memcpy(q, &n, 4);
      q += 4;
      memcpy(q, &chromEnd[t], 4);
      q += 4;
      for (int i = 0; i < n; i++) {
        memcpy(q, &mx[(size_t)i], 8);
        q += 8;
        memcpy(q, &di[(size_t)i], 4);
        q += 4;
        memcpy(q, &prv[(size_t)i], 8);
        q += 8;
      }
      pos += 4 + size;
    }
  }
  bool ok = fwrite(table.data(), 8, table.size(), f) == table.size() &&
            fwrite(body.data(), 1, body.size(), f) == body.size();
  ok = fclose(f) == 0 && ok;
  return ok ? 0 : -1;
}

/* Tests: parse a bedGraph file with the fast path (use_fast != 0) or with sscanf only, and
 * report the status, the line count and an FNV-1a hash of everything parsed. */
extern "C" int peakseg_hip_parse_probe(const char *path, int use_fast, int *n_lines,
                                       unsigned long long *hash) {
  Coverage cv;
  int st = read_bedGraph_impl(path, cv, use_fast != 0);
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&h](const void *q, size_t n) {
    const unsigned char *b = (const unsigned char *)q;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  };
  if (st == 0) {
    mix(cv.chromEnd.data(), cv.chromEnd.size() * 4);
    mix(cv.count.data(), cv.count.size() * 4);
    mix(cv.weight.data(), cv.weight.size() * 4);
    mix(cv.chrom.data(), cv.chrom.size());
    mix(&cv.first_chromStart, 4);
    mix(&cv.cum_weight, 8);
    mix(&cv.cum_weighted_count, 8);
    mix(&cv.min_log_mean, 8);
    mix(&cv.max_log_mean, 8);
  }
  if (n_lines) *n_lines = cv.n();
  if (hash) *hash = h;
  return st;
}

/* diagnostic builds (-DPSD_PROFILE) only: per-wave cycle counters of the forward kernel */
extern "C" int peakseg_hip_problem_set_profile(psd_problem_set *s, int p, long long *out) {
#ifdef PSD_PROFILE
  if (!s || !s->solved || p < 0 || p >= s->n_problems) return -1;
  if (hipMemcpy(out, s->d.prof + (size_t)p * 2 * psd::N_PROF, sizeof(long long) * 2 * psd::N_PROF,
                hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return psd::N_PROF;
#else
  (void)s;
  (void)p;
  (void)out;
  return -1;
#endif
}

extern "C" int peakseg_hip_math_probe(int op, int n, const double *x, double *y) {
  if (peakseg_hip_device_count() <= 0) {
    set_error("no HIP device visible (this library has no CPU fallback)");
    return ERROR_NO_HIP_DEVICE;
  }
  if (n <= 0) return 0;
  double *dx = nullptr, *dy = nullptr;
  const size_t n_in = op >= 2 ? (size_t)2 * (size_t)n : (size_t)n; /* (division: two operands) */
  HIP_TRY(hipMalloc(&dx, n_in * 8));
  HIP_TRY(hipMalloc(&dy, (size_t)n * 8));
  HIP_TRY(hipMemcpy(dx, x, n_in * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(psd::lat::math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t) nullptr, op, n, dx, dy);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(y, dy, (size_t)n * 8, hipMemcpyDeviceToHost));
  (void)hipFree(dx);
  (void)hipFree(dy);
  return 0;
}

/* ---- segment tables packed in HBM (the multi-GPU gather's payload) ----------------------- */

/* one workgroup per problem: rows[3 p] = first packed row, rows[3 p + 1] = row count,
 * rows[3 p + 2] = the table's offset in seg_start / seg_mean */
__global__ void pack_tables_kernel(const int *seg_start, const double *seg_mean,
                                   const long long *rows, int *out_start, double *out_mean) {
  const long long to = rows[3 * blockIdx.x], n = rows[3 * blockIdx.x + 1],
                  from = rows[3 * blockIdx.x + 2];
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    out_start[to + i] = seg_start[from + i];
    out_mean[to + i] = seg_mean[from + i];
  }
}

extern "C" long long peakseg_hip_problem_set_pack_tables(psd_problem_set *s, long long *rows_out,
                                                         const int **start_dev,
                                                         const double **mean_dev) {
  if (!s || !s->solved) return -1;
  if (hipSetDevice(s->device) != hipSuccess) return -1;
  std::vector<long long> rows((size_t)3 * (size_t)s->n_problems);
  long long total = 0;
  for (int p = 0; p < s->n_problems; p++) {
    const psd::ProbResult &r = s->results[(size_t)p];
    const long long n = r.status == 0 ? r.n_segments : 0;
    rows[(size_t)3 * p] = total;
    rows[(size_t)3 * p + 1] = n;
    rows[(size_t)3 * p + 2] = s->prob_seg_off[(size_t)p];
    if (rows_out) rows_out[p] = n;
    total += n;
  }
  if (total > s->pack_capacity || !s->d_pack_rows) {
    for (void *q : {(void *)s->d_pack_start, (void *)s->d_pack_mean}) {
      if (!q) continue;
      forget_alloc(s, q);
      (void)hipFree(q);
    }
    s->bytes -= (unsigned long long)s->pack_capacity * 12ull;
    s->d_pack_start = nullptr;
    s->d_pack_mean = nullptr;
    s->pack_capacity = 0;
    if (dev_alloc(s, &s->d_pack_start, (size_t)total) || dev_alloc(s, &s->d_pack_mean, (size_t)total))
      return -1;
    s->pack_capacity = total > 0 ? total : 1;
    if (!s->d_pack_rows && dev_alloc(s, &s->d_pack_rows, rows.size())) return -1;
  }
  if (hipMemcpy(s->d_pack_rows, rows.data(), rows.size() * sizeof(long long),
                hipMemcpyHostToDevice) != hipSuccess)
    return -1;
  hipLaunchKernelGGL(pack_tables_kernel, dim3((unsigned)s->n_problems), dim3(256), 0, s->stream,
                     (const int *)s->d.seg_start, (const double *)s->d.seg_mean,
                     (const long long *)s->d_pack_rows, s->d_pack_start, s->d_pack_mean);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) {
    set_error("packing the segment tables failed");
    return -1;
  }
  s->pack_total = total;
  if (start_dev) *start_dev = s->d_pack_start;
  if (mean_dev) *mean_dev = s->d_pack_mean;
  return total;
}

extern "C" int peakseg_hip_problem_set_packed_download(psd_problem_set *s, int *start_out,
                                                       double *mean_out) {
  if (!s || s->pack_total < 0) return -1;
  const size_t n = (size_t)s->pack_total;
  if (n == 0) return 0;
  if (hipMemcpy(start_out, s->d_pack_start, n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(mean_out, s->d_pack_mean, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
    set_error("download of the packed segment tables failed");
    return -1;
  }
  return 0;
}

extern "C" int peakseg_hip_problem_set_park_stats(psd_problem_set *s, int *parks,
                                                  unsigned long long *overflow_pool_pieces) {
  if (!s) return -1;
  if (parks) *parks = s->parks;
  if (overflow_pool_pieces) *overflow_pool_pieces = s->park_pool_pieces;
  return 0;
}

/* shader cycles the problem's workgroup ran in the last launch that touched it */
extern "C" long long peakseg_hip_problem_set_cycles(psd_problem_set *s, int p) {
  if (!s || !s->solved || p < 0 || p >= s->n_problems) return -1;
  return s->results[(size_t)p].cycles;
}

/* data points per second one problem advanced at in this process's last clean single-build
 * solves (what the mixed-launch planner reasons with); the defaults before any */
extern "C" void peakseg_hip_measured_rates(double *lat_rate, double *thr_rate) {
  if (lat_rate) *lat_rate = g_lat_rate.load();
  if (thr_rate) *thr_rate = g_thr_rate.load();
}

/* the bound on the polls of a wait between waves (tests/test_gpu_round4.py) and, in builds with
 * -DPSD_SPIN_STATS, the largest poll count a problem's waves saw */
extern "C" long long peakseg_hip_spin_limit(void) { return (long long)psd::lat::WAIT_SPIN_LIMIT; }
extern "C" int peakseg_hip_problem_set_max_spin(psd_problem_set *s, int p) {
  if (!s || !s->solved || p < 0 || p >= s->n_problems) return -1;
  return s->results[(size_t)p].max_spin;
}

/* ---- file-level solver (the reference's boundary), directory-level batch with the cache
 *      protocol, resident penalty search ----------------------------------------------------- */
#include "peakseg_files.h"

extern "C" char *PeakSegFPOP_status_message(int status, const char *bedGraph, const char *penalty,
                                            const char *db, char *buf, size_t buf_len) {
  if (!buf || buf_len == 0) return buf;
  buf[0] = 0;
  switch (status) { /* texts of /root/reference/src/interface.cpp:16-55 */
    case 0:
      break;
    case ERROR_PENALTY_NOT_FINITE:
      snprintf(buf, buf_len, "penalty=%s but must be finite", penalty);
      break;
    case ERROR_PENALTY_NEGATIVE:
      snprintf(buf, buf_len, "penalty=%s must be non-negative", penalty);
      break;
    case ERROR_UNABLE_TO_OPEN_BEDGRAPH:
      snprintf(buf, buf_len, "unable to open input file for reading %s", bedGraph);
      break;
    case ERROR_NOT_ENOUGH_COLUMNS:
      snprintf(buf, buf_len, "each line of input data file %s should have exactly four columns",
               bedGraph);
      break;
    case ERROR_NON_INTEGER_DATA:
      snprintf(buf, buf_len, "fourth column of input data file %s should be integer", bedGraph);
      break;
    case ERROR_INCONSISTENT_CHROMSTART_CHROMEND:
      snprintf(buf, buf_len, "there should be no gaps (columns 2-3) in input data file %s",
               bedGraph);
      break;
    case ERROR_WRITING_COST_FUNCTIONS:
      snprintf(buf, buf_len, "unable to write to cost function database file %s", db);
      break;
    case ERROR_WRITING_LOSS_OUTPUT:
      snprintf(buf, buf_len, "unable to write to loss output file %s_penalty=%s_loss.tsv", bedGraph,
               penalty);
      break;
    case ERROR_WRITING_SEGMENTS_OUTPUT:
      snprintf(buf, buf_len, "unable to write to segments output file %s_penalty=%s_segments.bed",
               bedGraph, penalty);
      break;
    case ERROR_NO_DATA:
      snprintf(buf, buf_len, "input file %s contains no data", bedGraph);
      break;
    case ERROR_PENALTY_NOT_NUMERIC:
      snprintf(buf, buf_len,
               "penalty string '%s' is not numeric; it should be convertible to double", penalty);
      break;
    case ERROR_NO_HIP_DEVICE:
      snprintf(buf, buf_len,
               "error code %d: no HIP device (MI355X) is visible and this solver has no CPU path",
               status);
      break;
    default:
      snprintf(buf, buf_len, "error code %d", status);
      break;
  }
  return buf;
}
