// Diagnostic: what the calls behind the growing arena cost at the sizes a big set needs
// (hipMemCreate / Map / SetAccess / Unmap / Release against hipMalloc / hipFree).
// build: hipcc --offload-arch=gfx950 -O2 tools/vmm_time_probe.cpp -o tools/_build/vmm_time_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define T(label, call)                                                   \
  do {                                                                   \
    double t0 = now();                                                   \
    hipError_t e = (call);                                               \
    (void)hipDeviceSynchronize();                                        \
    printf("  %-16s %8.1f ms  %s\n", label, (now() - t0) * 1e3,          \
           e == hipSuccess ? "" : hipGetErrorString(e));                 \
  } while (0)

int main() {
  (void)hipSetDevice(0);
  (void)hipFree(nullptr);
  const size_t gib = 1ull << 30;
  for (size_t bytes : {4 * gib, 32 * gib, 64 * gib}) {
    printf("%zu GiB\n", bytes / gib);
    void *p = nullptr;
    T("hipMalloc", hipMalloc(&p, bytes));
    T("hipMemset", hipMemset(p, 0, bytes));
    T("hipFree", hipFree(p));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void *va = nullptr;
    T("AddressReserve", hipMemAddressReserve(&va, bytes, 2 << 20, nullptr, 0));
    hipMemGenericAllocationHandle_t h;
    T("hipMemCreate", hipMemCreate(&h, bytes, &prop, 0));
    T("hipMemMap", hipMemMap(va, bytes, 0, h, 0));
    T("hipMemSetAccess", hipMemSetAccess(va, bytes, &acc, 1));
    T("hipMemset", hipMemset(va, 0, bytes));
    T("hipMemUnmap", hipMemUnmap(va, bytes));
    T("hipMemRelease", hipMemRelease(h));
    T("AddressFree", hipMemAddressFree(va, bytes));
  }
  return 0;
}
