/* peakseg_detmath.h -- deterministic fp64 exp / log.
 *
 * The reference's only third-party arithmetic on the hot path is libm exp()/log()
 * (call sites /root/reference/src/funPieceListLog.cpp:38,57,116-126,
 * 203,214,231,960-961,996,1178-1179,1211-1212 and src/PeakSegFPOPLog.cpp:198,228,430,442).
 * glibc's and ROCm ocml's versions differ in last-bit rounding, and every branch of the
 * piece-list algebra compares values produced by them, so the numeric contract of this
 * library pins ONE implementation that yields bit-identical results on the host
 * (gcc/clang, any x86-64) and on gfx950:
 *
 *   - only IEEE-754 binary64 +, -, *, /, fused multiply-add and integer bit operations;
 *   - every fused operation is an explicit psd_fma(); everything else must be compiled
 *     with floating-point contraction OFF (-ffp-contract=off; the pragmas below repeat it);
 *   - round-to-nearest-even mode, subnormals honoured (both true by default on x86-64
 *     and for f64 on gfx950).
 *
 * Accuracy (measured by tests/test_detmath.py against mpmath): < 1 ulp for both.
 * Table-driven (128-entry tables, tools/gen_detmath_tables.py) with short polynomials: the
 * kernels are bound by the length of one wave's dependent instruction stream, and this form
 * needs about half the instructions of a table-free one and no division inside log.
 *
 * Used by: the HIP kernels and C++ host driver in peaksegdisk_amd/csrc/, and by the CPU
 * oracle in oracle/ (which can alternatively be built against libm, -DORACLE_LIBM, to
 * reproduce the reference's own digits).
 */
#ifndef PEAKSEG_DETMATH_H
#define PEAKSEG_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define PSD_HD __host__ __device__
#else
#define PSD_HD
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

#define PSD_INFINITY (__builtin_inf())

PSD_HD static inline double psd_fma(double a, double b, double c) {
  return __builtin_fma(a, b, c);
}

PSD_HD static inline uint64_t psd_d2u(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, sizeof u);
  return u;
}

PSD_HD static inline double psd_u2d(uint64_t u) {
  double x;
  __builtin_memcpy(&x, &u, sizeof x);
  return x;
}

/* a / b, correctly rounded, for the divisions whose divisor can sit just below a power of two.
 * On the host that is the division operator.  On the device it is not quite: the quotient of
 * the compiler's fp64 division sequence (v_div_scale, v_rcp_f64, two Newton steps, v_div_fmas,
 * v_div_fixup) is one unit off when the exact quotient lies within ~2^-50 of a unit of the
 * midpoint of two doubles.  That takes a divisor a few units below a power of two (its
 * significand all ones, the one case the classical proof of this sequence leaves out):
 * 0x1.6666666666663p-1 / 0x1.ffffffffffffbp-1 gives ...666 instead of ...667; 29 of 8 million
 * divisions by 1 - k ulp, k < 64, with numerators a few units off small fractions, against 0 of
 * 8 million with random operands, with divisors 1 + k ulp, or with divisors 2^n -+ k ulp under
 * numerators up to 60 units off (profiles/r04/div_probe.log).  A divisor of 1 - k ulp is what
 * the Linear coefficient of a piece that has seen every weight so far is (normalised weights
 * sum to one less rounding), and -Log / Linear is that piece's optimum: one such quotient made a
 * prev_log_mean and a breakpoint of a 4851-bin contig differ from the CPU's in their last bits
 * (found by tools/knob_soak.py, seed 12; tests/golden/division_near_tie.json).
 * So the quotients by a piece's Linear coefficient (the optimum of a piece, the degenerate
 * crossing, and the envelope's difference pieces, whose Linear is the function piece's own
 * against a constant piece) are repaired, exactly: the residual
 * a - q b of the hardware's q (one fma, exact), the neighbour of q on the side the residual
 * points to, its residual, and the quotient with the smaller one.  (A quotient of two doubles
 * is never exactly a midpoint: no tie; zero, infinite and NaN quotients fail both comparisons
 * and stay.)  Twelve instructions -- at every division of the kernel they cost 13 % -- so the
 * other divisors keep the hardware's sequence: sums of bin widths (a whole number b < 2^48
 * keeps the quotient 1/(2b) of a unit away from every midpoint), and the iterates and slopes of
 * the Newton loops and the coefficients of DIFFERENCES of pieces, arbitrary reals that are
 * within 2^9 units below a power of two once in 2^44 times and then still need the numerator
 * to match. */
/* q: a quotient of a by b that is at most one unit off; returns the correctly rounded one
 * (host and device: tests/test_oracle_golden.py feeds it the IEEE quotient's neighbours) */
PSD_HD static inline double psd_div_repair(double q, double a, double b) {
  const double e = psd_fma(-q, b, a);
  const int above = (e > 0.0) == (b > 0.0); /* the exact quotient is above q */
  const uint64_t uq = psd_d2u(q);
  const double qn = psd_u2d(above == (q > 0.0) ? uq + 1u : uq - 1u);
  const double en = psd_fma(-qn, b, a);
  return __builtin_fabs(en) < __builtin_fabs(e) ? qn : q;
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ static inline double psd_div(double a, double b) { return psd_div_repair(a / b, a, b); }
#else
PSD_HD static inline double psd_div(double a, double b) { return a / b; }
#endif

/* 2^e for e in [-1022, 1023] */
PSD_HD static inline double psd_pow2i(int e) {
  return psd_u2d((uint64_t)(e + 1023) << 52);
}

#define PSD_LN2_HI 0x1.62e42fee00000p-1  /* ln2 truncated to 32 bits: k*LN2_HI is exact */
#define PSD_LN2_LO 0x1.a39ef35793c76p-33
#define PSD_INV_LN2 0x1.71547652b82fep+0

#if defined(__GNUC__) || defined(__clang__)
#define PSD_COLD __attribute__((noinline, cold))
#else
#define PSD_COLD
#endif

/* ---- tables (generated by tools/gen_detmath_tables.py) -------------------------------- */
#include "peakseg_detmath_tables.h"

#define PSD_EXP_N 128 /* exp: 2^(j/128), j = 0..127, as {tail, hi} pairs */
#define PSD_LOG_N 128 /* log: c_j = 1 + j/128, j = 0..128 */

static const double psd_tab_exp[2 * PSD_EXP_N] = {PSD_EXP_TABLE_INIT};
static const double psd_tab_invc[PSD_LOG_N + 1] = {PSD_LOG_INVC_INIT};
static const double psd_tab_logc_hi[PSD_LOG_N + 1] = {PSD_LOG_LOGC_HI_INIT};
static const double psd_tab_logc_lo[PSD_LOG_N + 1] = {PSD_LOG_LOGC_LO_INIT};

#if defined(__HIPCC__) || defined(__HIP__)
/* Device code reads the tables from LDS (a dependent table lookup from LDS costs ~60 cycles,
 * from HBM/L2 several hundred); every kernel that evaluates psd_exp/psd_log calls
 * psd_tables_init() first. */
/* exp: {tail, hi} pairs first (offset 0: the pair is one ds_read2_b64 off the masked index);
 * log: one 32-byte record per table point (one ds_read_b128 + one ds_read_b64, the table base
 * in the instruction's offset field). */
struct alignas(32) psd_lds_log_entry_t {
  double invc, logc_hi, logc_lo, pad;
};
struct psd_lds_tables_t {
  double e[2 * PSD_EXP_N];
  psd_lds_log_entry_t lg[PSD_LOG_N + 1];
};
__shared__ psd_lds_tables_t psd_lds_tables;
__device__ const double psd_dev_exp[2 * PSD_EXP_N] = {PSD_EXP_TABLE_INIT};
__device__ const double psd_dev_invc[PSD_LOG_N + 1] = {PSD_LOG_INVC_INIT};
__device__ const double psd_dev_logc_hi[PSD_LOG_N + 1] = {PSD_LOG_LOGC_HI_INIT};
__device__ const double psd_dev_logc_lo[PSD_LOG_N + 1] = {PSD_LOG_LOGC_LO_INIT};
__device__ static inline void psd_tables_init(void) {
  for (unsigned i = threadIdx.x; i < 2 * PSD_EXP_N; i += blockDim.x)
    psd_lds_tables.e[i] = psd_dev_exp[i];
  for (unsigned i = threadIdx.x; i <= PSD_LOG_N; i += blockDim.x) {
    psd_lds_tables.lg[i].invc = psd_dev_invc[i];
    psd_lds_tables.lg[i].logc_hi = psd_dev_logc_hi[i];
    psd_lds_tables.lg[i].logc_lo = psd_dev_logc_lo[i];
    psd_lds_tables.lg[i].pad = 0.0;
  }
  __syncthreads();
}
#else
static inline void psd_tables_init(void) {}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PSD_T_EXP(i) (psd_lds_tables.e[i])
#define PSD_T_INVC(i) (psd_lds_tables.lg[i].invc)
#define PSD_T_LOGC_HI(i) (psd_lds_tables.lg[i].logc_hi)
#define PSD_T_LOGC_LO(i) (psd_lds_tables.lg[i].logc_lo)
#else
#define PSD_T_EXP(i) (psd_tab_exp[i])
#define PSD_T_INVC(i) (psd_tab_invc[i])
#define PSD_T_LOGC_HI(i) (psd_tab_logc_hi[i])
#define PSD_T_LOGC_LO(i) (psd_tab_logc_lo[i])
#endif

/* ln2/128 split: the high part has 32 significant bits so kd * hi is exact for |kd| < 2^21 */
#define PSD_LN2N_HI 0x1.62e42feep-8
#define PSD_LN2N_LO 0x1.a39ef35793c76p-40
#define PSD_INV_LN2N 0x1.71547652b82fep+7

/* On the device the rare-argument test is made once per wave (a scalar branch on the ballot)
 * instead of an exec-mask branch around the hot path; on the host it is a plain test. */
#if defined(__HIP_DEVICE_COMPILE__)
#define PSD_ANY_LANE(c) (__builtin_amdgcn_ballot_w64(c) != 0)
#else
#define PSD_ANY_LANE(c) (c)
#endif

PSD_HD static PSD_COLD double psd_exp_slow(double x);
PSD_HD static PSD_COLD double psd_log_slow(double x);

/* psd_exp_core, psd_exp, psd_log_core, psd_log */
#define PSD_FN(name) name
#define PSD_K(x) (x)
#include "peakseg_detmath_core.h"
#undef PSD_FN
#undef PSD_K

/* NaN, overflow, underflow and results near the ends of the double range */
PSD_HD static PSD_COLD double psd_exp_slow(double x) {
  if (x != x) return x + x;
  if (x > 0x1.62e42fefa39efp+9) return PSD_INFINITY; /* > log(DBL_MAX) */
  if (x < -0x1.74910d52d3052p+9) return 0.0;         /* < log(2^-1075) */
  long long kq;
  double hi;
  double t = psd_exp_core(x, &kq, &hi);
  int k = (int)(kq >> 7); /* floor */
  double y = psd_fma(hi, t, hi); /* in [1, 2) */
  /* exact scaling first, one more rounding only when the result is subnormal */
  int k2 = k / 2;
  return (y * psd_pow2i(k - k2)) * psd_pow2i(k2);
}

/* NaN, negative, zero, infinite and subnormal arguments */
PSD_HD static PSD_COLD double psd_log_slow(double x) {
  if (x != x) return x + x;
  if (x < 0.0) return (x - x) / 0.0; /* NaN */
  if (x == 0.0) return -PSD_INFINITY;
  if (x == PSD_INFINITY) return x;
  return psd_log_core(psd_d2u(x * 0x1p54), -54); /* subnormal */
}

#if defined(__HIP_DEVICE_COMPILE__)
/* psd_exp_vk, psd_log_vk: constants in vector registers (see peakseg_detmath_core.h) */
/* The constant XORed with a zero the optimiser cannot see through (mbcnt of an empty mask):
 * a pure, speculatable value, so the compiler materialises each constant once per kernel in a
 * vector register pair and keeps it there (an `asm("" : "+v"(k))` barrier did the same but cost
 * a v_mov_b64 copy at every use: 6 of the 35 instructions of an exp, 7 of the 45 of a log). */
__device__ static inline double psd_vk(double k) {
  const uint64_t u = psd_d2u(k);
  const uint32_t z = __builtin_amdgcn_mbcnt_lo(0u, 0u); /* always 0 */
  return psd_u2d(((uint64_t)((uint32_t)(u >> 32) ^ z) << 32) | ((uint32_t)u ^ z));
}
#define PSD_FN(name) name##_vk
#define PSD_K(x) psd_vk(x)
#include "peakseg_detmath_core.h"
#undef PSD_FN
#undef PSD_K
#endif

#endif /* PEAKSEG_DETMATH_H */
