/* fpop_pieces.h -- per-lane (scalar) arithmetic on one Poisson-loss piece
 *   g(x) = Linear*exp(x) + Log*x + Constant   on a log-mean interval.
 *
 * Device restatement of the reference's PoissonLossPieceLog methods
 * (/root/reference/src/funPieceListLog.cpp, cited per function as fpl:lines).  Operation
 * and operand order are the reference's: they decide the rounding of every fp64 result and
 * therefore every branch of the piece-list algebra.  exp/log are the deterministic pair of
 * include/peakseg_detmath.h; the file is compiled with -ffp-contract=off.
 *
 * NO include guard: compiled once per build variant into namespace psd::PSD_VARIANT (see
 * fpop_wave.h); PSD_MATH_VK selects the exp/log pair with constants in vector registers. */
#include "psd_platform.h"

#ifndef PSD_VARIANT
#error "define PSD_VARIANT before including fpop_pieces.h (see fpop_wave.h)"
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace psd {
namespace PSD_VARIANT {

constexpr double NEWTON_EPSILON_VALUE = 1e-12; /* fpl:9 */
/* The tolerance is compared against in some sixty places of a step.  As a literal every use
 * costs two scalar moves (an fp64 literal cannot be an operand).  Keeping it in a vector
 * register pair like the exp/log constants (-DPSD_EPS_IN_VGPR) measured 0.5% slower
 * (profiles/r02/ab_uniform_args.log): the pair adds to a register file that already spills. */
#undef NEWTON_EPSILON
#if defined(PSD_MATH_VK) && defined(__HIP_DEVICE_COMPILE__) && defined(PSD_EPS_IN_VGPR)
#define NEWTON_EPSILON (psd_vk(NEWTON_EPSILON_VALUE))
#else
#define NEWTON_EPSILON NEWTON_EPSILON_VALUE
#endif
#ifdef PSD_NEWTON_STEPS /* tests only: force the step-cap fallback of the root finders */
constexpr int NEWTON_STEPS = PSD_NEWTON_STEPS;
#else
constexpr int NEWTON_STEPS = 100;        /* fpl:10 */
#endif
constexpr int PREV_NOT_SET = -3;         /* fpl:11 */
#define PSD_INF (__builtin_inf())

struct Coef {
  double Linear, Log, Constant;
};

PSD_D double absd(double x) { return x < 0 ? -x : x; } /* fpl:13 ABS */

/* Leaf math stays inline: an out-of-line exp/log/root makes every caller spill its live
 * doubles to scratch around the call.  Code size is controlled one level up instead (the
 * wave operations of fpop_wave.h are the out-of-line units). */
#if defined(PSD_MATH_VK) && defined(__HIP_DEVICE_COMPILE__)
PSD_D double d_exp(double x) { return psd_exp_vk(x); }
PSD_D double d_log(double x) { return psd_log_vk(x); }
#else
PSD_D double d_exp(double x) { return psd_exp(x); }
PSD_D double d_log(double x) { return psd_log(x); }
#endif

#if defined(PSD_MATH_VK) && defined(__HIP_DEVICE_COMPILE__)
PSD_D double d_exp_nb(double x, int &rare) { return psd_exp_nb_vk(x, &rare); }
PSD_D double d_log_nb(double x, int &rare) { return psd_log_nb_vk(x, &rare); }
PSD_D double d_log_wild_nb(double x, int &rare) { return psd_log_wild_nb_vk(x, &rare); }
#else
PSD_D double d_exp_nb(double x, int &rare) { return psd_exp_nb(x, &rare); }
PSD_D double d_log_nb(double x, int &rare) { return psd_log_nb(x, &rare); }
PSD_D double d_log_wild_nb(double x, int &rare) { return psd_log_wild_nb(x, &rare); }
#endif
#if defined(PSD_MATH_VK) && defined(__HIP_DEVICE_COMPILE__)
PSD_D void d_exp2(double x0, double x1, double &y0, double &y1) { psd_exp2_vk(x0, x1, &y0, &y1); }
PSD_D void d_log2(double x0, double x1, double &y0, double &y1) { psd_log2_vk(x0, x1, &y0, &y1); }
PSD_D void d_exp2_log(double x0, double x1, double z, double &y0, double &y1, double &lz) {
  psd_exp2_log_vk(x0, x1, z, &y0, &y1, &lz);
}
#else
PSD_D void d_exp2(double x0, double x1, double &y0, double &y1) { psd_exp2(x0, x1, &y0, &y1); }
PSD_D void d_log2(double x0, double x1, double &y0, double &y1) { psd_log2(x0, x1, &y0, &y1); }
PSD_D void d_exp2_log(double x0, double x1, double z, double &y0, double &y1, double &lz) {
  psd_exp2_log(x0, x1, z, &y0, &y1, &lz);
}
#endif

/* fpl:192-197 */
/* (psd_div: Linear is 1 - k ulp for a piece that has seen every weight; peakseg_detmath.h) */
PSD_D double argmin_mean(const Coef &c) { return psd_div(-c.Log, c.Linear); }

/* fpl:199-204 */
PSD_D double argmin(const Coef &c) { return d_log(argmin_mean(c)); }

/* fpl:206-222 */
PSD_D double get_cost(const Coef &c, double log_mean) {
  /* the reference skips exp() at -Inf and uses 0; here exp is evaluated at a harmless argument
   * and the product is replaced by 0 with a select (no exec-mask branch around the hot path) */
  const bool at_zero_mean = log_mean == -PSD_INF;
  double e = d_exp(at_zero_mean ? 0.0 : log_mean);
  double linear_term = at_zero_mean ? 0.0 : c.Linear * e;
  double log_term = (c.Log == 0) ? 0.0 : c.Log * log_mean;
  return linear_term + log_term + c.Constant;
}

/* fpl:52-61 */
PSD_D double poisson_loss(const Coef &c, double mean) {
  double loss_without_log_term = c.Linear * mean + c.Constant;
  if (c.Log == 0) {
    return loss_without_log_term;
  }
  double product = d_log(mean) * c.Log;
  return loss_without_log_term + product;
}

/* getCost (fpl:206-222) with exp(log_mean) supplied by the caller.  The reference calls libm
 * again for every evaluation; exp is a pure function of its argument, so re-using a value
 * already computed for the same argument is exact. */
PSD_D double get_cost_e(const Coef &c, double log_mean, double exp_log_mean) {
  double linear_term = (log_mean == -PSD_INF) ? 0.0 : c.Linear * exp_log_mean;
  double log_term = (c.Log == 0) ? 0.0 : c.Log * log_mean;
  return linear_term + log_term + c.Constant;
}

/* How a step evaluates exp / log.  NB = false: the complete functions, each with its branch to
 * the rare-argument path.  NB = true (the inlined fast path of a step): without those branches
 * -- a branch ends a basic block, and a wave on its own can only overlap the ~110 cycles of a
 * table look-up with work of the same block -- and with a record of having met a rare argument;
 * the step then ends with ONE test of that record and is redone by the general path, complete
 * functions and all (never on real data: |log-mean| > 708, means <= 0, NaN).  Removing every
 * such branch is worth 3.8 % of the forward kernel (profiles/r03/ab_rare_branches.log). */
template <bool NB>
struct StepMath {
  int rare;
#ifdef PSD_FORCE_RARE /* tests only: every step (and helper round) takes the redo path */
  PSD_M StepMath() : rare(NB ? 1 : 0) {}
#else
  PSD_M StepMath() : rare(0) {}
#endif
  PSD_M double exp(double x) {
    if (NB) return d_exp_nb(x, rare);
    return d_exp(x);
  }
  PSD_M double log(double x) {
    if (NB) return d_log_nb(x, rare);
    return d_log(x);
  }
  /* log of a quotient that may be +-Inf, NaN or 0 as a matter of course (psd_log_wild_nb) */
  PSD_M double log_wild(double x) {
    if (NB) return d_log_wild_nb(x, rare);
    return d_log(x);
  }
  PSD_M void exp2(double x0, double x1, double &y0, double &y1) {
    if (NB) {
      y0 = d_exp_nb(x0, rare);
      y1 = d_exp_nb(x1, rare);
    } else {
      d_exp2(x0, x1, y0, y1);
    }
  }
  PSD_M void log2(double x0, double x1, double &y0, double &y1) {
    if (NB) {
      y0 = d_log_nb(x0, rare);
      y1 = d_log_nb(x1, rare);
    } else {
      d_log2(x0, x1, y0, y1);
    }
  }
  PSD_M void exp2_log(double x0, double x1, double z, double &y0, double &y1, double &lz) {
    if (NB) {
      y0 = d_exp_nb(x0, rare);
      y1 = d_exp_nb(x1, rare);
      lz = d_log_nb(z, rare);
    } else {
      d_exp2_log(x0, x1, z, y0, y1, lz);
    }
  }
  /* getCost (fpl:206-222), as get_cost() */
  PSD_M double cost(const Coef &c, double log_mean) {
    const bool at_zero_mean = log_mean == -PSD_INF;
    double e = exp(at_zero_mean ? 0.0 : log_mean);
    double linear_term = at_zero_mean ? 0.0 : c.Linear * e;
    double log_term = (c.Log == 0) ? 0.0 : c.Log * log_mean;
    return linear_term + log_term + c.Constant;
  }
  /* the final log of get_larger_root */
  PSD_M int *rare_out() { return NB ? &rare : nullptr; }
};
typedef StepMath<false> MathFull;
typedef StepMath<true> MathFast;

/* The optimum of a piece, shared by has_two_roots / get_smaller_root / get_larger_root /
 * argmin (fpl:37-40,70-71,134-135,203): every one of them starts from the same
 * argmin_mean(), log(argmin_mean()) and the costs there. */
struct PieceOpt {
  double mean;      /* argmin_mean()            = -Log/Linear */
  double log_mean;  /* argmin()                 = log(mean) */
  double cost;      /* getCost(log_mean) */
  double cost2;     /* PoissonLoss(mean)        (uses log(mean) = log_mean) */
};

PSD_D PieceOpt piece_opt(const Coef &c) {
  PieceOpt o;
  o.mean = argmin_mean(c);
  o.log_mean = d_log(o.mean);
  o.cost = get_cost(c, o.log_mean);
  double loss_without_log_term = c.Linear * o.mean + c.Constant; /* fpl:52-61 */
  o.cost2 = (c.Log == 0) ? loss_without_log_term : loss_without_log_term + o.log_mean * c.Log;
  return o;
}

/* fpl:29-50; caller guarantees c.Log != 0 (the reference throws otherwise) */
PSD_D bool has_two_roots(const Coef &c, const PieceOpt &o, double equals) {
  if (0 < c.Linear) {
    return o.cost + NEWTON_EPSILON < equals && o.cost2 + NEWTON_EPSILON < equals;
  }
  return equals + NEWTON_EPSILON < o.cost && equals + NEWTON_EPSILON < o.cost2;
}

/* The reference's Newton loops (fpl:98-124,158-188) also track the closest iterate on each
 * side of the root, but only use that bracket when 100 steps were not enough.  The hot loops
 * below iterate without the bookkeeping (about a fifth of the loop's instructions); if step
 * 100 is ever reached they hand over to newton_with_bracket, which repeats the solve from the
 * start with the bracket -- same iterates, same result. */

/* A solve that ran into the step cap (or met a rare exp/log argument), redone from its start
 * with what the reference keeps for exactly this case (fpl:84-124,145-188): of all iterates,
 * the one whose residual is closest to zero from above and the one closest from below.  At the
 * cap the answer is the better of the last iterate and the midpoint of that bracket.
 * IN_LOG_SPACE: the iterates are log-means (the smaller root; residual by getCost, slope
 * Linear e^x + Log), otherwise means (the larger root; residual by PoissonLoss, slope
 * Linear + Log / m, answer returned as a log-mean).  The same iterates as the hot loops below:
 * same operations, same order.  x_opt / f_opt: the optimum the solve starts next to and the
 * cost there, as the caller's entry point derived them. */
template <bool IN_LOG_SPACE>
PSD_COLD_DEV double newton_with_bracket(Coef c, double x_opt, double f_opt, double level) {
  struct Side { /* the iterate nearest to the root on one side, and its residual */
    double f, x;
  };
  Side above = {PSD_INF, PSD_INF}, below = {-PSD_INF, PSD_INF};
  if (f_opt < 0) { /* (the optimum's cost as it is, not its residual: fpl:90,152) */
    below.f = f_opt;
    below.x = x_opt;
  } else {
    above.f = f_opt;
    above.x = x_opt;
  }
  auto residual = [&](double x, double &slope) -> double {
    if (IN_LOG_SPACE) {
      const double lin = (x == -PSD_INF) ? 0.0 : c.Linear * d_exp(x);
      const double lg = (c.Log == 0) ? 0.0 : c.Log * x;
      slope = lin + c.Log;
      return (lin + lg + c.Constant) - level;
    }
    slope = c.Linear + c.Log / x; /* PoissonDeriv fpl:63-65 */
    return poisson_loss(c, x) - level;
  };
  double x = IN_LOG_SPACE ? x_opt - 1 : x_opt + 1;
  for (int trip = 1;; trip++) {
    double slope;
    const double f = residual(x, slope);
    if (0 < f && f < above.f) {
      above.f = f;
      above.x = x;
    }
    if (below.f < f && f < 0) {
      below.f = f;
      below.x = x;
    }
    if (NEWTON_STEPS <= trip) {
      const double mid = (above.x + below.x) / 2;
      double unused;
      const double f_mid = residual(mid, unused);
      const double best = (absd(f_mid) < absd(f)) ? mid : x;
      return IN_LOG_SPACE ? best : d_log(best);
    }
    if (IN_LOG_SPACE) {
      const double step = f / slope;
      x = x - step;
    } else {
      x = x - f / slope;
    }
    if (!(NEWTON_EPSILON < absd(f))) return IN_LOG_SPACE ? x : d_log(x);
  }
}

/* fpl:69-127: larger root, returned as a log-mean.
 * right_cost = getCost(max_log_mean), supplied by the caller. */
PSD_D double get_larger_root(const Coef &c, const PieceOpt &o, double max_log_mean,
                             double right_cost, double equals, int *steps_out = nullptr,
                             int *rare_out = nullptr) {
  double optimal_mean = o.mean;
  double optimal_cost = o.cost2;
  if ((optimal_cost < right_cost && right_cost < equals) ||
      (optimal_cost > right_cost && right_cost > equals)) {
    return max_log_mean + 1;
  }
  double candidate_root = optimal_mean + 1;
  double candidate_cost;
  int step = 0;
#ifndef PSD_NEWTON_EXIT_IN_LOOP
  /* One way out of the loop: the reference leaves at trip NEWTON_STEPS from the middle of the
   * body (fpl:109-120), whatever the cost there; here the loop simply ends at that trip and the
   * complete function -- which redoes the solve with the bracket bookkeeping -- is called
   * behind it.  Same results; a loop with a single exit and no return inside. */
#ifndef PSD_NEWTON_RARE_BRANCH
  /* ... and no branch to log's rare-argument path inside it: the loop records that it met such
   * an argument (a mean <= 0, a subnormal, Inf, NaN: an iteration gone astray) and the solve is
   * then redone by the complete function, which handles them as the reference's libm does. */
  int rare = 0;
  do {
    const double loss_without_log_term = c.Linear * candidate_root + c.Constant; /* fpl:52-61 */
    const double product = d_log_nb(candidate_root, rare) * c.Log;
    candidate_cost = ((c.Log == 0) ? loss_without_log_term : loss_without_log_term + product) - equals;
    ++step;
    double deriv = c.Linear + c.Log / candidate_root; /* PoissonDeriv fpl:63-65 */
    candidate_root = candidate_root - candidate_cost / deriv;
  } while (NEWTON_EPSILON < absd(candidate_cost) && step < NEWTON_STEPS);
  if (NEWTON_STEPS <= step || rare != 0)
    return newton_with_bracket<false>(c, optimal_mean, optimal_cost, equals);
#else
  do {
    candidate_cost = poisson_loss(c, candidate_root) - equals;
    ++step;
    double deriv = c.Linear + c.Log / candidate_root; /* PoissonDeriv fpl:63-65 */
    candidate_root = candidate_root - candidate_cost / deriv;
  } while (NEWTON_EPSILON < absd(candidate_cost) && step < NEWTON_STEPS);
  if (NEWTON_STEPS <= step) return newton_with_bracket<false>(c, optimal_mean, optimal_cost, equals);
#endif
#else
  do {
    candidate_cost = poisson_loss(c, candidate_root) - equals;
    if (NEWTON_STEPS <= ++step) return newton_with_bracket<false>(c, optimal_mean, optimal_cost, equals);
    double deriv = c.Linear + c.Log / candidate_root; /* PoissonDeriv fpl:63-65 */
    candidate_root = candidate_root - candidate_cost / deriv;
  } while (NEWTON_EPSILON < absd(candidate_cost));
#endif
  if (steps_out) *steps_out = step;
  if (rare_out) return d_log_nb(candidate_root, *rare_out); /* (see StepMath) */
  return d_log(candidate_root);
}

/* fpl:129-190: smaller root (a log-mean).
 * left_cost = getCost(min_log_mean), supplied by the caller. */
PSD_D double get_smaller_root(const Coef &c, const PieceOpt &o, double min_log_mean,
                              double left_cost, double equals, int *steps_out = nullptr) {
  double optimal_log_mean = o.log_mean;
  double optimal_cost = o.cost;
  if ((equals < left_cost && left_cost < optimal_cost) ||
      (equals > left_cost && left_cost > optimal_cost)) {
    return min_log_mean - 1;
  }
  double candidate_root = optimal_log_mean - 1;
  double candidate_cost;
  int step = 0;
#ifndef PSD_NEWTON_RARE_BRANCH
  int rare = 0; /* an exp argument beyond +-708 or NaN was met: redo with the complete function */
#endif
  do {
    /* getCost and getDeriv (fpl:206-234) evaluate the same Linear*exp(x): once here */
    const bool at_zero_mean = candidate_root == -PSD_INF;
#ifndef PSD_NEWTON_RARE_BRANCH
    double e = d_exp_nb(at_zero_mean ? 0.0 : candidate_root, rare);
#else
    double e = d_exp(at_zero_mean ? 0.0 : candidate_root);
#endif
    double linear_term = at_zero_mean ? 0.0 : c.Linear * e;
    double log_term = (c.Log == 0) ? 0.0 : c.Log * candidate_root;
    candidate_cost = (linear_term + log_term + c.Constant) - equals;
#ifdef PSD_NEWTON_EXIT_IN_LOOP
    if (NEWTON_STEPS <= ++step)
      return newton_with_bracket<true>(c, optimal_log_mean, optimal_cost, equals);
#else
    ++step; /* the step cap ends the loop; see get_larger_root */
#endif
    double deriv = linear_term + c.Log;
    double offset = candidate_cost / deriv;
    candidate_root = candidate_root - offset;
#ifdef PSD_NEWTON_EXIT_IN_LOOP
  } while (NEWTON_EPSILON < absd(candidate_cost));
#else
  } while (NEWTON_EPSILON < absd(candidate_cost) && step < NEWTON_STEPS);
#ifndef PSD_NEWTON_RARE_BRANCH
  if (NEWTON_STEPS <= step || rare != 0)
#else
  if (NEWTON_STEPS <= step)
#endif
    return newton_with_bracket<true>(c, optimal_log_mean, optimal_cost, equals);
#endif
  if (steps_out) *steps_out = step;
  return candidate_root;
}

/* fpl:862-868 */
PSD_D bool same_funs(const Coef &a, const Coef &b) {
  return a.Linear == b.Linear && a.Log == b.Log &&
         absd(a.Constant - b.Constant) < NEWTON_EPSILON;
}

}  // namespace PSD_VARIANT
}  // namespace psd
