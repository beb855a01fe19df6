// latency_probe.cpp -- diagnostic: single-wave dependent-chain latencies on gfx950 for the
// operations the FPOP kernels are made of (fp64 fma / div, psd_exp / psd_log, LDS round trip,
// ballot, shuffle).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude
#include <hip/hip_runtime.h>
#include <cstdio>
#include "peakseg_detmath.h"

#define N_ITER 2000

__global__ void probe(double *out, long long *cyc, double seed) {
  __shared__ double lds[256];
  psd_tables_init();
  int lane = threadIdx.x;
  double x = seed + lane * 1e-3;
  long long t0, t1;
  int slot = 0;
#define RUN(NAME, BODY)                         \
  t0 = __builtin_readcyclecounter();            \
  for (int i = 0; i < N_ITER; i++) { BODY; }    \
  t1 = __builtin_readcyclecounter();            \
  if (lane == 0) cyc[slot] = t1 - t0;           \
  slot++;
  RUN(fma, x = __builtin_fma(x, 0.999999, 1e-7))
  RUN(mul, x = x * 1.0000001)
  RUN(add, x = x + 1e-9)
  RUN(div, x = 1.0000001 / x + 0.5)
  RUN(exp, x = psd_exp(x * 1e-3) )
  RUN(log, x = psd_log(x + 1.5))
  RUN(lds, lds[lane] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); x = lds[(lane + 1) & 63] + 1e-9)
  RUN(ballot, x += (double)__popcll(__ballot(x > 0.5)) * 1e-12)
  RUN(shfl, x = __shfl(x, (lane + 1) & 63, 64) + 1e-9)
  RUN(readlane, x = __shfl(x, i & 63, 64) + 1e-9)
  RUN(newton_exp_div, { double e = psd_exp(x * 1e-3); double c = 2.0 * e + 0.5 * x - 3.0; double d = 2.0 * e + 0.5; x = x - c / d * 1e-3; })
  out[lane] = x;
}

int main() {
  double *out;
  long long *cyc, h[16];
  hipMalloc(&out, 64 * 8);
  hipMalloc(&cyc, 16 * 8);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.25);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
  const char *names[] = {"fma_f64", "mul_f64", "add_f64", "div_f64(+add)", "psd_exp(+mul)", "psd_log(+add)",
                         "lds write+sync+read(+add)", "ballot+popc+cvt+fma", "shfl(bpermute)+add",
                         "shfl uniform idx+add", "newton step: exp + ~6 flops + div"};
  for (int i = 0; i < 11; i++) printf("%-36s %8.1f cycles per iteration\n", names[i], (double)h[i] / N_ITER);
  return 0;
}
