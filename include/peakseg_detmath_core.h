/* peakseg_detmath_core.h -- the bodies of psd_exp / psd_log.
 *
 * NO include guard.  peakseg_detmath.h includes this file with PSD_FN(name) = name and
 * PSD_K(x) = x: the functions every host program and the oracle use.  Device code gets a second
 * set, name##_vk, whose fp64 constants are held in vector registers (PSD_K = psd_vk): as
 * literals they live in scalar register pairs for the whole kernel (24 SGPRs), and the
 * latency build of the forward kernel is short of SGPRs -- every spilled SGPR costs a
 * v_readlane at each use -- while the throughput build is short of VGPRs and keeps the
 * plain set.  Same source text, same operations in the same order: same bits.
 */

/* Core of exp: x = (128 k + j) ln2/128 + r, |r| <= ln2/256,
 * exp(x) = 2^k * T[j] * (1 + p(r)),  p = r + r^2/2 + ... + r^5/120  (truncation 5e-19).
 * Returns t = tail_j + p(r) and hi_j; *kq = 128 k + j. */
PSD_HD static inline double PSD_FN(psd_exp_core)(double x, long long *kq, double *hi) {
  const double shift = PSD_K(0x1.8p52);
  /* one rounding: the low mantissa bits of z are rint(x * 128/ln2) */
  double z = psd_fma(x, PSD_K(PSD_INV_LN2N), shift);
  long long ki = (long long)psd_d2u(z) - (long long)0x4338000000000000LL;
  double kd = z - shift;
  double r = psd_fma(kd, PSD_K(-PSD_LN2N_HI), x);
  r = psd_fma(kd, PSD_K(-PSD_LN2N_LO), r);
  int j = (int)(ki & (PSD_EXP_N - 1));
  double tail = PSD_T_EXP(2 * j);
  *hi = PSD_T_EXP(2 * j + 1);
  *kq = ki;
  double r2 = r * r;
  double a = psd_fma(r, PSD_K(0x1.5555555555555p-3), 0.5); /* 1/2 + r/6 */
  double b = psd_fma(r, PSD_K(0x1.1111111111111p-7), PSD_K(0x1.5555555555555p-5)); /* 1/24 + r/120 */
  double t = tail + r;
  t = psd_fma(r2, a, t);
  return psd_fma(r2 * r2, b, t);
}

/* exp(x); |x| <= 708 keeps the result normal.  The hot path is computed unconditionally (it
 * is harmless for any argument: the table index is masked) and replaced for the rare lanes
 * that need psd_exp_slow. */
PSD_HD static inline double PSD_FN(psd_exp)(double x) {
  long long kq;
  double hi;
  double t = PSD_FN(psd_exp_core)(x, &kq, &hi);
  /* scale = 2^k * hi_j: add k to the exponent field */
  double scale = psd_u2d(psd_d2u(hi) + ((uint64_t)(kq >> 7) << 52));
  double y = psd_fma(scale, t, scale);
  /* The rare-argument test is ONE wave-uniform branch on the device (v_cmp + s_cbranch_vccz);
   * behind it every lane evaluates the slow form and the rare ones keep it: no exec-mask
   * bracket around the common path. */
  const int rare = !(__builtin_fabs(x) <= 708.0);
  if (PSD_ANY_LANE(rare)) {
    const double ys = psd_exp_slow(x);
    y = rare ? ys : y;
  }
  return y;
}

/* log of a positive normal number given its bits; k0 = exponent adjustment.
 * x = 2^k m, m in [1,2); c = 1 + J/128 is the table point nearest to m (J = 0..128),
 * r = m * invc - 1 (|r| <= 2^-8, exact for J = 0 and J = 128),
 * log x = k ln2 + (-log invc) + log1p(r),  log1p(r) = r + r^2 q(r), q of degree 5.
 * k*LN2_HI + logc_hi is exact by construction of the table, the rest is accumulated in a
 * low word.  For x near 1 (J = 0 with k = 0, or J = 128 with k = -1) the table terms cancel
 * exactly and the result keeps full relative accuracy. */
PSD_HD static inline double PSD_FN(psd_log_core)(uint64_t hx, int k0) {
  /* exponent, table index and mantissa all come from the high word (32-bit integer work) */
  const uint32_t top = (uint32_t)(hx >> 32);
  int k = k0 + (int)(top >> 20) - 1023;
  const uint32_t mant_hi = top & 0x000fffffu;
  int J = (int)((mant_hi + 0x00001000u) >> 13);
  double m = psd_u2d(((uint64_t)(mant_hi | 0x3ff00000u) << 32) | (uint32_t)hx);
  double invc = PSD_T_INVC(J);
  double logc_hi = PSD_T_LOGC_HI(J);
  double logc_lo = PSD_T_LOGC_LO(J);
  double r = psd_fma(m, invc, -1.0);
  double kd = (double)k;
  double w = psd_fma(kd, PSD_K(PSD_LN2_HI), logc_hi); /* exact */
  double hi = w + r;
  double lo = (w - hi) + r; /* exact: w == 0 or |w| > |r| */
  lo = lo + psd_fma(kd, PSD_K(PSD_LN2_LO), logc_lo);
  double r2 = r * r;
  double q01 = psd_fma(r, PSD_K(0x1.5555555555555p-2), -0.5);  /* -1/2 + r/3 */
  double q23 = psd_fma(r, PSD_K(0x1.999999999999ap-3), -0.25); /* -1/4 + r/5 */
  double q45 = psd_fma(r, PSD_K(0x1.2492492492492p-3), PSD_K(-0x1.5555555555555p-3)); /* -1/6 + r/7 */
  double q = psd_fma(r2 * r2, q45, psd_fma(r2, q23, q01));
  return psd_fma(r2, q, lo) + hi;
}

PSD_HD static inline double PSD_FN(psd_log)(double x) {
  uint64_t hx = psd_d2u(x);
  double y = PSD_FN(psd_log_core)(hx, 0);
  /* anything but positive, normal, finite: one unsigned compare of the high word */
  const int rare = !((uint32_t)(hx >> 32) - 0x00100000u < 0x7fe00000u);
  if (PSD_ANY_LANE(rare)) {
    const double ys = psd_log_slow(x);
    y = rare ? ys : y;
  }
  return y;
}

/* Two (three) independent evaluations in one basic block.  Each psd_exp / psd_log ends in the
 * branch to its rare-argument path, and a branch is a scheduling barrier: two calls in a row
 * run one after the other although neither needs the other's result.  A wave on its own issues
 * a dependent fp64 instruction every 7-9 cycles but an independent one every 4, so evaluating
 * them interleaved, with ONE rare-argument branch for all, takes little more than one of them.
 * Results are those of the single functions, bit for bit (pure functions of the argument). */
PSD_HD static inline void PSD_FN(psd_exp2)(double x0, double x1, double *y0, double *y1) {
  long long k0, k1;
  double h0, h1;
  double t0 = PSD_FN(psd_exp_core)(x0, &k0, &h0);
  double t1 = PSD_FN(psd_exp_core)(x1, &k1, &h1);
  double s0 = psd_u2d(psd_d2u(h0) + ((uint64_t)(k0 >> 7) << 52));
  double s1 = psd_u2d(psd_d2u(h1) + ((uint64_t)(k1 >> 7) << 52));
  double r0 = psd_fma(s0, t0, s0), r1 = psd_fma(s1, t1, s1);
  const int rare0 = !(__builtin_fabs(x0) <= 708.0), rare1 = !(__builtin_fabs(x1) <= 708.0);
  if (PSD_ANY_LANE(rare0 | rare1)) {
    const double q0 = psd_exp_slow(x0), q1 = psd_exp_slow(x1);
    r0 = rare0 ? q0 : r0;
    r1 = rare1 ? q1 : r1;
  }
  *y0 = r0;
  *y1 = r1;
}

PSD_HD static inline void PSD_FN(psd_log2)(double x0, double x1, double *y0, double *y1) {
  const uint64_t u0 = psd_d2u(x0), u1 = psd_d2u(x1);
  double r0 = PSD_FN(psd_log_core)(u0, 0), r1 = PSD_FN(psd_log_core)(u1, 0);
  const int rare0 = !((uint32_t)(u0 >> 32) - 0x00100000u < 0x7fe00000u);
  const int rare1 = !((uint32_t)(u1 >> 32) - 0x00100000u < 0x7fe00000u);
  if (PSD_ANY_LANE(rare0 | rare1)) {
    const double q0 = psd_log_slow(x0), q1 = psd_log_slow(x1);
    r0 = rare0 ? q0 : r0;
    r1 = rare1 ? q1 : r1;
  }
  *y0 = r0;
  *y1 = r1;
}

/* exp(x0), exp(x1), log(z) */
PSD_HD static inline void PSD_FN(psd_exp2_log)(double x0, double x1, double z, double *y0,
                                               double *y1, double *lz) {
  long long k0, k1;
  double h0, h1;
  double t0 = PSD_FN(psd_exp_core)(x0, &k0, &h0);
  double t1 = PSD_FN(psd_exp_core)(x1, &k1, &h1);
  const uint64_t uz = psd_d2u(z);
  double rz = PSD_FN(psd_log_core)(uz, 0);
  double s0 = psd_u2d(psd_d2u(h0) + ((uint64_t)(k0 >> 7) << 52));
  double s1 = psd_u2d(psd_d2u(h1) + ((uint64_t)(k1 >> 7) << 52));
  double r0 = psd_fma(s0, t0, s0), r1 = psd_fma(s1, t1, s1);
  const int rare0 = !(__builtin_fabs(x0) <= 708.0), rare1 = !(__builtin_fabs(x1) <= 708.0);
  const int rarez = !((uint32_t)(uz >> 32) - 0x00100000u < 0x7fe00000u);
  if (PSD_ANY_LANE(rare0 | rare1 | rarez)) {
    const double q0 = psd_exp_slow(x0), q1 = psd_exp_slow(x1), qz = psd_log_slow(z);
    r0 = rare0 ? q0 : r0;
    r1 = rare1 ? q1 : r1;
    rz = rarez ? qz : rz;
  }
  *y0 = r0;
  *y1 = r1;
  *lz = rz;
}

/* The hot path of psd_exp / psd_log WITHOUT the branch to the rare-argument path: the value is
 * psd_exp(x) / psd_log(x) whenever *rare stays 0 and unspecified (but harmless) otherwise.  For
 * the Newton loops, which evaluate one of them per trip: the loop only records that a rare
 * argument was met; the solve is then redone with the complete functions (never on real data). */
PSD_HD static inline double PSD_FN(psd_exp_nb)(double x, int *rare) {
  long long kq;
  double hi;
  double t = PSD_FN(psd_exp_core)(x, &kq, &hi);
  double scale = psd_u2d(psd_d2u(hi) + ((uint64_t)(kq >> 7) << 52));
  *rare |= !(__builtin_fabs(x) <= 708.0);
  return psd_fma(scale, t, scale);
}
PSD_HD static inline double PSD_FN(psd_log_nb)(double x, int *rare) {
  const uint64_t hx = psd_d2u(x);
  /* A negative argument is not rare: the reference takes the log of negative quotients as a
   * matter of course (funPieceListLog.cpp:563,996: the crossing of degenerate pieces) and
   * lets the NaN fail every comparison.  NaN by a select; the record is for 0, subnormals, Inf
   * and NaN only. */
  const int negative = x < 0.0;
  *rare |= !((uint32_t)(hx >> 32) - 0x00100000u < 0x7fe00000u) && !negative;
  const double y = PSD_FN(psd_log_core)(hx, 0);
  return negative ? psd_u2d(0x7ff8000000000000ULL) : y;
}

/* psd_log_nb for a site whose argument is a quotient that may be anything -- a division by the
 * zero Linear coefficient of a constant piece gives +-Inf or NaN, an exact cancellation 0
 * (funPieceListLog.cpp:563) -- and all of it is ordinary there: the special values by selects
 * (what psd_log_slow returns for them), the record only for positive subnormals. */
PSD_HD static inline double PSD_FN(psd_log_wild_nb)(double x, int *rare) {
  const uint64_t hx = psd_d2u(x);
  const int odd = !((uint32_t)(hx >> 32) - 0x00100000u < 0x7fe00000u);
  double y = PSD_FN(psd_log_core)(hx, 0);
  y = (x < 0.0) ? psd_u2d(0x7ff8000000000000ULL) : y;
  y = (x == 0.0) ? -PSD_INFINITY : y;
  y = (x == PSD_INFINITY) ? x : y;
  y = (x != x) ? x + x : y;
  *rare |= odd && x > 0.0 && x < 0x1p-1022;
  return y;
}
