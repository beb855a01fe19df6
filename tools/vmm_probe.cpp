#include <hip/hip_runtime.h>
#include <cstdio>
static hipMemAllocationProp prop; static hipMemAccessDesc acc;
static const char *try_map(char *base, size_t off, size_t bytes, hipMemGenericAllocationHandle_t *h) {
  hipError_t e = hipMemCreate(h, bytes, &prop, 0); if (e != hipSuccess) return "create";
  e = hipMemMap(base + off, bytes, 0, *h, 0); if (e != hipSuccess) { hipMemRelease(*h); return "map"; }
  e = hipMemSetAccess(base + off, bytes, &acc, 1); if (e != hipSuccess) { hipMemUnmap(base + off, bytes); hipMemRelease(*h); return "setaccess"; }
  return "ok";
}
__global__ void touch(double *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (double)i; }
int main() {
  hipSetDevice(0);
  prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t MB = 1 << 20;
  size_t blocks[] = {12 * MB, 10 * MB, 4 * MB, 36 * MB, 1024 * MB};
  for (int rep = 0; rep < 2; rep++)
  for (size_t S : blocks) {
    void *base[3]; size_t bs[3] = {S, S, S / 2};
    for (int q = 0; q < 3; q++) if (hipMemAddressReserve(&base[q], (size_t)64 << 30, 2 * MB, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
    hipMemGenericAllocationHandle_t h[3][5]; int ok = 1;
    for (int k = 0; k < 5 && ok; k++)
      for (int q = 0; q < 3 && ok; q++) {
        const char *r = try_map((char *)base[q], k * bs[q], bs[q], &h[q][k]);
        if (r[0] != 'o') { printf("block %zu MB: array %d block %d failed in %s\n", S / MB, q, k, r); ok = 0; }
      }
    if (ok) {
      hipLaunchKernelGGL(touch, dim3((unsigned)((5 * S / 8 + 255) / 256)), dim3(256), 0, 0, (double *)base[0], 5 * S / 8);
      hipError_t e = hipDeviceSynchronize();
      printf("block %zu MB x 5 on three ranges: ok, kernel over all five blocks: %s\n", S / MB, hipGetErrorString(e));
      for (int k = 0; k < 5; k++) for (int q = 0; q < 3; q++) { hipMemUnmap((char *)base[q] + k * bs[q], bs[q]); hipMemRelease(h[q][k]); }
    }
    for (int q = 0; q < 3; q++) hipMemAddressFree(base[q], (size_t)64 << 30);
    (void)hipGetLastError();
  }
  return 0;
}
