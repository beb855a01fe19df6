// Diagnostic: can the arena be grown (hipMemCreate / Map / SetAccess behind a reserved range)
// while a kernel that uses the range's earlier blocks is running, and does a running kernel see
// a capacity word in pinned host memory change?
// build: hipcc --offload-arch=gfx950 -O2 tools/vmm_overlap_probe.cpp -o tools/_build/vmm_overlap_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

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

// one wave: waits for the capacity word to reach `want` blocks, touching block 0 meanwhile,
// then writes one word into every block; gives up after `limit` clock ticks
__global__ void walker(unsigned long long *base, size_t block_words, volatile unsigned long long *cap,
                       unsigned long long want, long long limit, unsigned long long *out) {
  long long t0 = wall_clock64();
  unsigned long long seen = 0, spins = 0;
  while ((seen = __atomic_load_n((unsigned long long *)cap, __ATOMIC_RELAXED)) < want) {
    base[threadIdx.x] = spins++;
    if (wall_clock64() - t0 > limit) break;
    __builtin_amdgcn_s_sleep(100);
  }
  if (threadIdx.x == 0) {
    for (unsigned long long b = 0; b < seen; b++) base[b * block_words + 7] = 1000 + b;
    out[0] = seen;
    out[1] = (unsigned long long)(wall_clock64() - t0);
  }
}

int main() {
  CHECK(hipSetDevice(0));
  const size_t block = 8ull << 30;  // 8 GiB blocks
  const int n_blocks = 6;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = 0;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  void *va = nullptr;
  CHECK(hipMemAddressReserve(&va, block * n_blocks, 2 << 20, nullptr, 0));
  hipMemGenericAllocationHandle_t h[n_blocks];
  unsigned long long *cap = nullptr, *out = nullptr;
  CHECK(hipHostMalloc((void **)&cap, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CHECK(hipHostMalloc((void **)&out, 64, hipHostMallocMapped | hipHostMallocCoherent));
  double t0 = now();
  CHECK(hipMemCreate(&h[0], block, &prop, 0));
  CHECK(hipMemMap(va, block, 0, h[0], 0));
  CHECK(hipMemSetAccess(va, block, &acc, 1));
  printf("block 0 mapped in %.1f ms\n", (now() - t0) * 1e3);
  *cap = 1;
  out[0] = out[1] = 0;
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  t0 = now();
  hipLaunchKernelGGL(walker, dim3(1), dim3(64), 0, st, (unsigned long long *)va, block / 8, cap,
                     (unsigned long long)n_blocks, 100000000ll * 20, out);  // 100 MHz clock: 20 s
  CHECK(hipGetLastError());
  (void)hipStreamQuery(st);
  for (int b = 1; b < n_blocks; b++) {
    double t1 = now();
    CHECK(hipMemCreate(&h[b], block, &prop, 0));
    double t2 = now();
    CHECK(hipMemMap((char *)va + b * block, block, 0, h[b], 0));
    CHECK(hipMemSetAccess((char *)va + b * block, block, &acc, 1));
    double t3 = now();
    __atomic_store_n(cap, (unsigned long long)(b + 1), __ATOMIC_RELEASE);
    printf("block %d: create %.1f ms, map+access %.1f ms, kernel %s\n", b, (t2 - t1) * 1e3,
           (t3 - t2) * 1e3, hipStreamQuery(st) == hipErrorNotReady ? "still running" : "DONE");
  }
  CHECK(hipStreamSynchronize(st));
  printf("kernel saw %llu blocks after %.1f ms of its clock; host wall %.1f ms\n", out[0],
         out[1] / 1e5, (now() - t0) * 1e3);
  unsigned long long v = 0;
  for (int b = 0; b < n_blocks; b++) {
    CHECK(hipMemcpy(&v, (char *)va + b * block + 7 * 8, 8, hipMemcpyDeviceToHost));
    printf("block %d word: %llu\n", b, v);
  }
  return 0;
}
