// newton_probe.cpp -- diagnostic: cycles per Newton trip of the root finders the FPOP kernels
// use (fpop_pieces.h: get_smaller_root in log-mean space, get_larger_root in mean space), one
// wave alone on its SIMD, every lane solving the same well-conditioned crossing.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Ipeaksegdisk_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#define PSD_VARIANT probe
#define PSD_LDS_CAP 64
#define PSD_MATH_VK 1
#include "fpop_pieces.h"

using namespace psd::probe;

__global__ void probe(double *out, long long *res, double lin, double lg, double con, int reps) {
  psd_tables_init();
  Coef c = {lin + threadIdx.x * 1e-9, lg, con};
  PieceOpt o = piece_opt(c);
  long long t0, t1;
  double acc = 0.0;
  int steps = 0, total = 0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < reps; i++) {
    acc += get_smaller_root(c, o, -1e300, __builtin_nan(""), acc * 1e-300, &steps);
    total += steps;
  }
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { res[0] = t1 - t0; res[1] = total; }
  total = 0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < reps; i++) {
    acc += get_larger_root(c, o, 1e300, __builtin_nan(""), acc * 1e-300, &steps);
    total += steps;
  }
  t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { res[2] = t1 - t0; res[3] = total; }
  out[threadIdx.x] = acc;
}

int main() {
  double *out;
  long long *res, h[4];
  hipMalloc(&out, 64 * 8);
  hipMalloc(&res, 4 * 8);
  // a difference piece like those of the envelope: Linear e^x + Log x + Constant with two roots
  const double cases[3][3] = {{2.0, -6.0, 0.5}, {0.7, -3.0, 1.0}, {5.0, -40.0, 30.0}};
  for (int k = 0; k < 3; k++) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, res, cases[k][0], cases[k][1], cases[k][2], 500);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, res, 4 * 8, hipMemcpyDeviceToHost);
    printf("case %d: smaller root %6.1f cycles/trip (%.1f trips/solve, %.0f cycles/solve); larger root %6.1f cycles/trip (%.1f trips/solve, %.0f cycles/solve)\n",
           k, (double)h[0] / (double)h[1], (double)h[1] / 500.0, (double)h[0] / 500.0,
           (double)h[2] / (double)h[3], (double)h[3] / 500.0, (double)h[2] / 500.0);
  }
  return 0;
}
