// branch_probe.cpp -- diagnostic: what one wave alone pays for the control flow the forward
// kernel is full of: an exec-mask region with its skip branch (s_and_saveexec + s_cbranch_execz),
// a wave-uniform branch on a vector compare (v_cmp + s_cbranch_vccz), against the same
// arithmetic with no branch.  Build: hipcc --offload-arch=gfx950 -O3 tools/branch_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void probe(double *out, long long *res, int k, double thr) {
  const int lane = threadIdx.x;
  double a = out[lane], b = a * 0.5 + 1.0;
  long long t0, t1;
  // (0) plain: 16 x (14 dependent FMAs)
  t0 = __builtin_readcyclecounter();
  for (int r = 0; r < 64; r++) {
    REP16(a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); __asm__ volatile("" : "+v"(a));)
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) res[0] = t1 - t0;
  // (1) the same inside a per-lane if (lanes < k active): exec-mask region + skip branch
  t0 = __builtin_readcyclecounter();
  for (int r = 0; r < 64; r++) {
    REP16(if (lane < k) { a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); } __asm__ volatile("" : "+v"(a));)
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) res[1] = t1 - t0;
  // (2) the same behind a wave-uniform branch on a vector compare of the running value
  t0 = __builtin_readcyclecounter();
  for (int r = 0; r < 64; r++) {
    REP16(if (__builtin_amdgcn_ballot_w64(a < thr) != 0) { a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); a = __builtin_fma(a, b, 1e-3); } __asm__ volatile("" : "+v"(a));)
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) res[2] = t1 - t0;
  // (3) independent FMAs (two chains) for the issue rate of independent f64 work
  double c = b + 0.25;
  t0 = __builtin_readcyclecounter();
  for (int r = 0; r < 64; r++) {
    REP16(a = __builtin_fma(a, b, 1e-3); c = __builtin_fma(c, b, 1e-3); a = __builtin_fma(a, b, 1e-3); c = __builtin_fma(c, b, 1e-3); a = __builtin_fma(a, b, 1e-3); c = __builtin_fma(c, b, 1e-3); a = __builtin_fma(a, b, 1e-3); c = __builtin_fma(c, b, 1e-3); a = __builtin_fma(a, b, 1e-3); c = __builtin_fma(c, b, 1e-3); a = __builtin_fma(a, b, 1e-3); c = __builtin_fma(c, b, 1e-3); a = __builtin_fma(a, b, 1e-3); c = __builtin_fma(c, b, 1e-3); __asm__ volatile("" : "+v"(a), "+v"(c));)
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) res[3] = t1 - t0;
  out[lane] = a + c;
}
int main() {
  double *out; long long *res, h[4];
  hipMalloc(&out, 64 * 8); hipMalloc(&res, 4 * 8);
  hipMemset(out, 0, 64 * 8);
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, res, 64, 1e300); hipDeviceSynchronize(); }
  hipMemcpy(h, res, 32, hipMemcpyDeviceToHost);
  const double n = 64.0 * 16.0;
  printf("per block of 14 f64 FMAs, one wave alone: dependent chain %.1f cycles (%.1f per FMA); in an exec-mask region %.1f (+%.1f); behind ballot + uniform branch %.1f (+%.1f); 14 FMAs on two independent chains %.1f (%.1f per FMA)\n",
         h[0] / n, h[0] / n / 14, h[1] / n, (h[1] - h[0]) / n, h[2] / n, (h[2] - h[0]) / n, h[3] / n, h[3] / n / 14);
  return 0;
}
