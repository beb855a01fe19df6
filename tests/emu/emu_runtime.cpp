/* emu_runtime.cpp -- TEST INFRASTRUCTURE: fiber scheduler behind tests/emu/hip_emu.h. */
#include "hip_emu.h"

#include <sys/mman.h>
#include <time.h>

#include <atomic>
#include <thread>
#include <vector>

extern "C" void psd_emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl psd_emu_switch
.type psd_emu_switch,@function
psd_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size psd_emu_switch,.-psd_emu_switch
)");

namespace emu {

thread_local ThreadCtx *g_cur = nullptr;

namespace {

constexpr size_t STACK_BYTES = 256 * 1024;

struct Wave {
  int nlanes = 0;
  int arrived = 0;
  int gen = 0;
  uint64_t slot[2][64];
};

struct Fiber {
  ThreadCtx ctx;
  void *sp = nullptr;
  char *stack = nullptr;
  bool done = false;
  Wave *wave = nullptr;
};

struct BlockState {
  ~BlockState() {
    for (Fiber &f : fibers)
      if (f.stack) munmap(f.stack, STACK_BYTES);
  }
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int bar_arrived = 0;
  int bar_gen = 0;
  int live = 0;
  const std::function<void()> *body = nullptr;
  void *sched_sp = nullptr;
  Fiber *cur = nullptr;
  bool progress = false;
  unsigned nthreads = 0;
};

thread_local BlockState *g_blk = nullptr;

void yield_to_scheduler() {
  Fiber *f = g_blk->cur;
  psd_emu_switch(&f->sp, g_blk->sched_sp);
}

void fiber_entry() {
  Fiber *f = g_blk->cur;
  (*g_blk->body)();
  f->done = true;
  g_blk->live--;
  g_blk->progress = true;
  /* a finished lane no longer takes part in its wave's collectives */
  f->wave->nlanes--;
  if (f->wave->nlanes > 0 && f->wave->arrived == f->wave->nlanes) {
    fprintf(stderr, "hip_emu: a lane exited while its wave waits in a collective\n");
    abort();
  }
  yield_to_scheduler();
  abort(); /* never resumed */
}

void prepare(Fiber &f) {
  if (!f.stack) {
    f.stack = (char *)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE,
                           MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (f.stack == (char *)MAP_FAILED) abort();
  }
  uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
  void **sp = (void **)top;
  *--sp = nullptr;               /* fake return address of fiber_entry */
  *--sp = (void *)&fiber_entry;  /* popped by `ret` */
  for (int i = 0; i < 6; i++) *--sp = nullptr; /* rbp rbx r12 r13 r14 r15 */
  f.sp = (void *)sp;
  f.done = false;
}

uint64_t collective(uint64_t v) {
  Fiber *f = g_blk->cur;
  Wave *w = f->wave;
  int lane = (int)(f->ctx.threadIdx_.x & 63u);
  int mygen = w->gen;
  int g = mygen & 1;
  w->slot[g][lane] = v;
  w->arrived++;
  g_blk->progress = true;
  if (w->arrived == w->nlanes) {
    w->arrived = 0;
    w->gen++;
  } else {
    while (w->gen == mygen) yield_to_scheduler();
  }
  return (uint64_t)g;
}

}  // namespace

unsigned long long ballot(bool p) {
  Wave *w = g_blk->cur->wave;
  int g = (int)collective(p ? 1 : 0);
  unsigned long long m = 0;
  int base = (int)(g_blk->cur->ctx.threadIdx_.x & ~63u);
  for (int l = 0; l < 64; l++) {
    unsigned t = (unsigned)(base + l);
    if (t < g_blk->nthreads && !g_blk->fibers[t].done && w->slot[g][l]) m |= 1ull << l;
  }
  return m;
}

double shfl_f64(double v, int src) {
  Wave *w = g_blk->cur->wave;
  uint64_t u;
  memcpy(&u, &v, 8);
  int g = (int)collective(u);
  uint64_t r = w->slot[g][src & 63];
  double d;
  memcpy(&d, &r, 8);
  return d;
}

int shfl_i32(int v, int src) {
  Wave *w = g_blk->cur->wave;
  int g = (int)collective((uint64_t)(uint32_t)v);
  return (int)(uint32_t)w->slot[g][src & 63];
}

void wave_sync() { (void)collective(0); }

void yield_fiber() { yield_to_scheduler(); }

void syncthreads() {
  BlockState *b = g_blk;
  int mygen = b->bar_gen;
  b->bar_arrived++;
  b->progress = true;
  if (b->bar_arrived == b->live) {
    b->bar_arrived = 0;
    b->bar_gen++;
  } else {
    while (b->bar_gen == mygen) yield_to_scheduler();
  }
}

/* the blocks [next, grid.x) of a launch, taken one at a time by the calling host thread */
static void run_blocks(dim3 grid, dim3 block, const std::function<void()> &body,
                       std::atomic<unsigned> &next) {
  static thread_local BlockState blk; /* stacks are reused across the launches of a thread */
  unsigned nthreads = block.x;
  if (blk.fibers.size() < nthreads) blk.fibers.resize(nthreads);
  blk.waves.assign((nthreads + 63) / 64, Wave());
  blk.body = &body;
  blk.nthreads = nthreads;
  BlockState *saved_blk = g_blk;
  ThreadCtx *saved_cur = g_cur;
  g_blk = &blk;
  for (unsigned b = next.fetch_add(1); b < grid.x; b = next.fetch_add(1)) {
    for (auto &w : blk.waves) w = Wave();
    for (unsigned t = 0; t < nthreads; t++) {
      Fiber &f = blk.fibers[t];
      prepare(f);
      f.ctx.threadIdx_ = dim3(t);
      f.ctx.blockIdx_ = dim3(b);
      f.ctx.blockDim_ = block;
      f.ctx.gridDim_ = grid;
      f.wave = &blk.waves[t / 64];
      f.wave->nlanes++;
    }
    blk.live = (int)nthreads;
    blk.bar_arrived = 0;
    blk.bar_gen = 0;
    int idle_rounds = 0; /* spin-wait loops (yield_fiber) may legitimately idle a few rounds */
    while (blk.live > 0) {
      blk.progress = false;
      for (unsigned t = 0; t < nthreads; t++) {
        Fiber &f = blk.fibers[t];
        if (f.done) continue;
        blk.cur = &f;
        g_cur = &f.ctx;
        psd_emu_switch(&blk.sched_sp, f.sp);
      }
      if (blk.progress) idle_rounds = 0;
      if (!blk.progress && blk.live > 0 && ++idle_rounds > 64) {
        fprintf(stderr,
                "hip_emu: deadlock in block %u (lanes disagree on a collective or barrier)\n", b);
        abort();
      }
    }
  }
  g_blk = saved_blk;
  g_cur = saved_cur;
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
  /* Blocks are independent (they meet in global memory through atomics only, as on the
   * device): a few host threads take them one at a time.  The calling thread is one of them,
   * so a launch of one block runs exactly as it did when blocks ran one after another. */
  unsigned workers = std::thread::hardware_concurrency();
  if (workers > 8) workers = 8;
  if (const char *e = getenv("PSD_EMU_THREADS")) workers = (unsigned)atoi(e);
  if (workers < 1) workers = 1;
  if (workers > grid.x) workers = grid.x;
  std::atomic<unsigned> next{0};
  std::vector<std::thread> pool;
  for (unsigned w = 1; w < workers; w++)
    pool.emplace_back([&]() { run_blocks(grid, block, body, next); });
  run_blocks(grid, block, body, next);
  for (auto &t : pool) t.join();
}

}  // namespace emu

/* ---- host runtime subset ------------------------------------------------------------ */
static double now_ms() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
hipError_t hipGetDeviceCount(int *n) {
  *n = 1;
  return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) {
  *v = 256; /* an MI355X */
  /* (PSD_EMU_CUS: a smaller device, so that a test of the launch planner -- which build for how
   * many problems per CU -- needs tens of problems, not thousands) */
  if (const char *e = getenv("PSD_EMU_CUS")) {
    const int n = atoi(e);
    if (n > 0) *v = n;
  }
  return hipSuccess;
}
hipError_t hipGetDevice(int *d) {
  *d = 0;
  return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t n) {
  *p = malloc(n ? n : 1);
  if (!*p) return hipErrorOutOfMemory;
  memset(*p, 0xCD, n); /* device memory starts uninitialised */
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
  *p = calloc(1, n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) {
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) {
  memcpy(dst, src, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) {
  memcpy(dst, src, n);
  return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t n) {
  memset(p, v, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t, emu_event *, unsigned) { return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) {
  *s = nullptr;
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) {
  *e = new emu_event{0.0};
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  e->t_ms = now_ms();
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  *ms = (float)(b->t_ms - a->t_ms);
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
  *free_b = (size_t)2 << 30;
  *total_b = (size_t)2 << 30;
  return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }
