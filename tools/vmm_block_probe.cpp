// Diagnostic (round 4): can a RUNNING kernel be given more arena if every block is an address
// range of its own?  Round 3 found that hipMemMap / hipMemSetAccess behind a reserved range
// wait for a running kernel that uses the range's earlier blocks (tools/vmm_overlap_probe.cpp).
// Cases, each against a kernel that polls a table of block base pointers in pinned host memory
// and writes one word into every block it is shown:
//   A  hipMemAddressReserve per block, reserved BEFORE the launch, created + mapped + access set
//      while the kernel runs
//   B  reserve + create + map + access, all while the kernel runs
//   C  plain hipMalloc of a new block while the kernel runs
//   D  hipMallocAsync on a second stream
//   E  as B, but the blocks are mapped by a SECOND host thread while the main thread sits in
//      hipStreamSynchronize on the kernel's stream (does the runtime serialise the two?)
//   F  ONE reserved range, blocks mapped behind it while the kernel runs (round 3's case), but
//      the kernel on a hipStreamNonBlocking stream like A-E
//   G  as B (a range per block), but the kernel on a BLOCKING stream (plain hipStreamCreate)
// (F and G separate the two things that changed between round 3's probe and cases A-E.)
// build: hipcc --offload-arch=gfx950 -O2 tools/vmm_block_probe.cpp -o tools/_build/vmm_block_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                        \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

constexpr int MAX_BLOCKS = 8;
struct Table {
  unsigned long long n_blocks; /* published last, release */
  unsigned long long *base[MAX_BLOCKS];
};

// one wave: until `want` blocks are published (or `limit` ticks of the 100 MHz clock), touch
// block 0; then write one word into every block it saw
__global__ void walker(volatile Table *tab, unsigned long long want, long long limit,
                       unsigned long long *out) {
  long long t0 = wall_clock64();
  unsigned long long seen = 0, spins = 0;
  unsigned long long *b0 = tab->base[0];
  while ((seen = __atomic_load_n((unsigned long long *)&tab->n_blocks, __ATOMIC_ACQUIRE)) < want) {
    b0[threadIdx.x] = spins++;
    if (wall_clock64() - t0 > limit) break;
    __builtin_amdgcn_s_sleep(100);
  }
  if (threadIdx.x == 0) {
    for (unsigned long long b = 0; b < seen; b++) tab->base[b][7] = 1000 + b;
    out[0] = seen;
    out[1] = (unsigned long long)(wall_clock64() - t0);
  }
}

static hipMemAllocationProp g_prop;
static hipMemAccessDesc g_acc;

static int map_block(void *va, size_t bytes, hipMemGenericAllocationHandle_t *h, double *t_create,
                     double *t_map) {
  double t1 = now();
  CHECK(hipMemCreate(h, bytes, &g_prop, 0));
  double t2 = now();
  CHECK(hipMemMap(va, bytes, 0, *h, 0));
  CHECK(hipMemSetAccess(va, bytes, &g_acc, 1));
  double t3 = now();
  *t_create = (t2 - t1) * 1e3;
  *t_map = (t3 - t2) * 1e3;
  return 0;
}

static int run_case(char which, size_t block) {
  const int n_blocks = 5;
  Table *tab = nullptr;
  unsigned long long *out = nullptr;
  CHECK(hipHostMalloc((void **)&tab, sizeof(Table), hipHostMallocMapped | hipHostMallocCoherent));
  CHECK(hipHostMalloc((void **)&out, 64, hipHostMallocMapped | hipHostMallocCoherent));
  memset(tab, 0, sizeof(Table));
  out[0] = out[1] = 0;
  void *va[MAX_BLOCKS] = {nullptr};
  hipMemGenericAllocationHandle_t h[MAX_BLOCKS];
  bool vmm = which == 'A' || which == 'B' || which == 'E' || which == 'F' || which == 'G';
  void *one_range = nullptr;
  double tc, tm;
  if (which == 'F') {
    CHECK(hipMemAddressReserve(&one_range, block * n_blocks, 2 << 20, nullptr, 0));
    for (int b = 0; b < n_blocks; b++) va[b] = (char *)one_range + (size_t)b * block;
    if (map_block(va[0], block, &h[0], &tc, &tm)) return 1;
  } else if (vmm) {
    CHECK(hipMemAddressReserve(&va[0], block, 2 << 20, nullptr, 0));
    if (map_block(va[0], block, &h[0], &tc, &tm)) return 1;
    if (which == 'A')
      for (int b = 1; b < n_blocks; b++) CHECK(hipMemAddressReserve(&va[b], block, 2 << 20, nullptr, 0));
  } else {
    CHECK(hipMalloc(&va[0], block));
  }
  tab->base[0] = (unsigned long long *)va[0];
  __atomic_store_n(&tab->n_blocks, 1ull, __ATOMIC_RELEASE);
  hipStream_t st, st2;
  if (which == 'G') {
    CHECK(hipStreamCreate(&st));
  } else {
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  }
  CHECK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  double t0 = now();
  hipLaunchKernelGGL(walker, dim3(1), dim3(64), 0, st, tab, (unsigned long long)n_blocks,
                     100000000ll * 8, out);  // gives up after 8 s
  CHECK(hipGetLastError());
  (void)hipStreamQuery(st);
  if (which == 'E') {
    std::thread mapper([&]() {
      (void)hipSetDevice(0);
      for (int b = 1; b < n_blocks; b++) {
        double c = 0, m = 0;
        if (hipMemAddressReserve(&va[b], block, 2 << 20, nullptr, 0) != hipSuccess) return;
        if (map_block(va[b], block, &h[b], &c, &m)) return;
        tab->base[b] = (unsigned long long *)va[b];
        __atomic_store_n(&tab->n_blocks, (unsigned long long)(b + 1), __ATOMIC_RELEASE);
        printf("case E block %d (second thread): create %.1f ms, map+access %.1f ms, at %.1f ms\n", b, c, m,
               (now() - t0) * 1e3);
        fflush(stdout);
      }
    });
    CHECK(hipStreamSynchronize(st));
    printf("case E: hipStreamSynchronize returned at %.1f ms\n", (now() - t0) * 1e3);
    mapper.join();
  }
  for (int b = 1; b < n_blocks && which != 'E'; b++) {
    double t1 = now();
    if (vmm) {
      if (which == 'B' || which == 'G') CHECK(hipMemAddressReserve(&va[b], block, 2 << 20, nullptr, 0));
      if (map_block(va[b], block, &h[b], &tc, &tm)) return 1;
    } else if (which == 'C') {
      CHECK(hipMalloc(&va[b], block));
      tc = (now() - t1) * 1e3;
      tm = 0;
    } else {
      CHECK(hipMallocAsync(&va[b], block, st2));
      CHECK(hipStreamSynchronize(st2));
      tc = (now() - t1) * 1e3;
      tm = 0;
    }
    tab->base[b] = (unsigned long long *)va[b];
    __atomic_store_n(&tab->n_blocks, (unsigned long long)(b + 1), __ATOMIC_RELEASE);
    printf("case %c block %d: create %.1f ms, map+access %.1f ms, at %.1f ms the kernel is %s\n", which, b,
           tc, tm, (now() - t0) * 1e3, hipStreamQuery(st) == hipErrorNotReady ? "still running" : "DONE");
    fflush(stdout);
  }
  CHECK(hipStreamSynchronize(st));
  printf("case %c: kernel saw %llu of %d blocks, ran %.1f ms of its clock; host wall %.1f ms\n", which,
         out[0], n_blocks, out[1] / 1e5, (now() - t0) * 1e3);
  int good = 0;
  for (int b = 0; b < n_blocks; b++) {
    unsigned long long v = 0;
    CHECK(hipMemcpy(&v, (char *)va[b] + 7 * 8, 8, hipMemcpyDeviceToHost));
    good += v == 1000ull + (unsigned long long)b;
  }
  printf("case %c: %d of %d blocks hold the kernel's word => %s\n", which, good, n_blocks,
         (good == n_blocks && out[1] / 1e5 < 7000.0) ? "GROWS UNDER A RUNNING KERNEL" : "does NOT");
  fflush(stdout);
  for (int b = 0; b < n_blocks; b++) {
    if (vmm) {
      (void)hipMemUnmap(va[b], block);
      (void)hipMemRelease(h[b]);
      if (which != 'F') (void)hipMemAddressFree(va[b], block);
    } else {
      (void)hipFree(va[b]);
    }
  }
  if (one_range) (void)hipMemAddressFree(one_range, block * n_blocks);
  (void)hipStreamDestroy(st);
  (void)hipStreamDestroy(st2);
  (void)hipHostFree(tab);
  (void)hipHostFree(out);
  return 0;
}

int main(int argc, char **argv) {
  CHECK(hipSetDevice(0));
  g_prop = {};
  g_prop.type = hipMemAllocationTypePinned;
  g_prop.location.type = hipMemLocationTypeDevice;
  g_prop.location.id = 0;
  g_acc = {};
  g_acc.location.type = hipMemLocationTypeDevice;
  g_acc.location.id = 0;
  g_acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t block = (argc > 2 ? (size_t)atoll(argv[2]) : 4ull) << 30;
  const char *cases = argc > 1 ? argv[1] : "ABCDEFG";
  for (const char *c = cases; *c; c++)
    if (run_case(*c, block)) printf("case %c: error\n", *c);
  return 0;
}
