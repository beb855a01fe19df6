#include <hip/hip_runtime.h>
#include <cstdio>
#define PSD_VARIANT probe
#define PSD_LDS_CAP 64
#define PSD_MATH_VK 1
#include "fpop_pieces.h"
using namespace psd::probe;
__device__ __forceinline__ double exp_poly(double x) {
  const double shift = 0x1.8p52;
  double z = __builtin_fma(x, 0x1.71547652b82fep+0, shift);
  long long ki = (long long)psd_d2u(z) - (long long)0x4338000000000000LL;
  double kd = z - shift;
  double r = __builtin_fma(kd, -0x1.62e42fee00000p-1, x);
  r = __builtin_fma(kd, -0x1.a39ef35793c76p-33, r);
  double p = 1.0 / 6227020800.0;
  p = __builtin_fma(p, r, 1.0 / 479001600.0);
  p = __builtin_fma(p, r, 1.0 / 39916800.0);
  p = __builtin_fma(p, r, 1.0 / 3628800.0);
  p = __builtin_fma(p, r, 1.0 / 362880.0);
  p = __builtin_fma(p, r, 1.0 / 40320.0);
  p = __builtin_fma(p, r, 1.0 / 5040.0);
  p = __builtin_fma(p, r, 1.0 / 720.0);
  p = __builtin_fma(p, r, 1.0 / 120.0);
  p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  double r2 = r * r;
  double t = __builtin_fma(p, r2, r);
  double scale = psd_u2d((uint64_t)(ki + 1023) << 52);
  return __builtin_fma(scale, t, scale);
}
__global__ void probe(double *out, long long *res, int reps, double c) {
  psd_tables_init();
  double a = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < reps; i++) a = d_exp(a * 1e-3 + c);
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) res[0] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < reps; i++) a = exp_poly(a * 1e-3 + c);
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) res[1] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < reps; i++) a = d_log(a * 1e-3 + 2.5);
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) res[2] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < reps; i++) a = a * 1e-3 + c;
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) res[3] = t1 - t0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < reps; i++) a = 1.5 / (a * 1e-3 + 2.5);
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) res[4] = t1 - t0;
  out[threadIdx.x] = a;
}
int main() {
  double *out; long long *res, h[5];
  hipMalloc(&out, 64 * 8); hipMalloc(&res, 5 * 8); hipMemset(out, 0, 64 * 8);
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, res, 2000, 0.37); hipDeviceSynchronize(); }
  hipMemcpy(h, res, 40, hipMemcpyDeviceToHost);
  printf("dependent chain, one wave alone, cycles per call (incl. one fma of glue = %.1f): table exp %.1f, table-free degree-13 exp %.1f, table log %.1f, division %.1f\n",
         h[3] / 2000.0, h[0] / 2000.0, h[1] / 2000.0, h[2] / 2000.0, h[4] / 2000.0);
  return 0;
}
