/* fpop_pieces.h -- per-lane (scalar) arithmetic on one Poisson-loss piece
 *   g(x) = Linear*exp(x) + Log*x + Constant   on a log-mean interval.
 *
 * Device restatement of the reference's PoissonLossPieceLog methods
 * (/root/reference/src/funPieceListLog.cpp, cited per function as fpl:lines).  Operation
 * and operand order are the reference's: they decide the rounding of every fp64 result and
 * therefore every branch of the piece-list algebra.  exp/log are the deterministic pair of
 * include/peakseg_detmath.h; the file is compiled with -ffp-contract=off.
 */
#ifndef PSD_FPOP_PIECES_H
#define PSD_FPOP_PIECES_H

#include "psd_platform.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace psd {

constexpr double NEWTON_EPSILON = 1e-12; /* fpl:9 */
constexpr int NEWTON_STEPS = 100;        /* fpl:10 */
constexpr int PREV_NOT_SET = -3;         /* fpl:11 */
#define PSD_INF (__builtin_inf())

struct Coef {
  double Linear, Log, Constant;
};

PSD_D double absd(double x) { return x < 0 ? -x : x; } /* fpl:13 ABS */

/* Leaf math stays inline: an out-of-line exp/log/root makes every caller spill its live
 * doubles to scratch around the call.  Code size is controlled one level up instead (the
 * wave operations of fpop_wave.h are the out-of-line units). */
PSD_D double d_exp(double x) { return psd_exp(x); }
PSD_D double d_log(double x) { return psd_log(x); }

/* fpl:192-197 */
PSD_D double argmin_mean(const Coef &c) { return -c.Log / c.Linear; }

/* fpl:199-204 */
PSD_D double argmin(const Coef &c) { return d_log(argmin_mean(c)); }

/* fpl:206-222 */
PSD_D double get_cost(const Coef &c, double log_mean) {
  double linear_term, log_term;
  if (log_mean == -PSD_INF) {
    linear_term = 0.0;
  } else {
    linear_term = c.Linear * d_exp(log_mean);
  }
  if (c.Log == 0) {
    log_term = 0.0;
  } else {
    log_term = c.Log * log_mean;
  }
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

/* fpl:29-50; caller guarantees c.Log != 0 (the reference throws otherwise) */
PSD_D bool has_two_roots(const Coef &c, double equals) {
  double optimal_mean = argmin_mean(c);
  double optimal_log_mean = d_log(optimal_mean);
  double optimal_cost = get_cost(c, optimal_log_mean);
  double optimal_cost2 = poisson_loss(c, optimal_mean);
  if (0 < c.Linear) {
    return optimal_cost + NEWTON_EPSILON < equals && optimal_cost2 + NEWTON_EPSILON < equals;
  }
  return equals + NEWTON_EPSILON < optimal_cost && equals + NEWTON_EPSILON < optimal_cost2;
}

/* fpl:69-127: Newton in mean space from argmin_mean+1; returns the log of the root. */
PSD_D double get_larger_root(const Coef &c, double max_log_mean, double equals) {
  double optimal_mean = argmin_mean(c);
  double optimal_cost = poisson_loss(c, optimal_mean);
  double right_cost = get_cost(c, max_log_mean);
  if ((optimal_cost < right_cost && right_cost < equals) ||
      (optimal_cost > right_cost && right_cost > equals)) {
    return max_log_mean + 1;
  }
  double candidate_root = optimal_mean + 1;
  double candidate_cost, deriv;
  double closest_positive_cost = PSD_INF, closest_positive_mean = PSD_INF;
  double closest_negative_cost = -PSD_INF, closest_negative_mean = PSD_INF;
  if (optimal_cost < 0) {
    closest_negative_cost = optimal_cost;
    closest_negative_mean = optimal_mean;
  } else {
    closest_positive_cost = optimal_cost;
    closest_positive_mean = optimal_mean;
  }
  int step = 0;
  do {
    candidate_cost = poisson_loss(c, candidate_root) - equals;
    if (0 < candidate_cost && candidate_cost < closest_positive_cost) {
      closest_positive_cost = candidate_cost;
      closest_positive_mean = candidate_root;
    }
    if (closest_negative_cost < candidate_cost && candidate_cost < 0) {
      closest_negative_cost = candidate_cost;
      closest_negative_mean = candidate_root;
    }
    if (NEWTON_STEPS <= ++step) {
      double between_closest = (closest_positive_mean + closest_negative_mean) / 2;
      double between_cost = poisson_loss(c, between_closest) - equals;
      if (absd(between_cost) < absd(candidate_cost)) {
        return d_log(between_closest);
      } else {
        return d_log(candidate_root);
      }
    }
    deriv = c.Linear + c.Log / candidate_root; /* PoissonDeriv fpl:63-65 */
    candidate_root = candidate_root - candidate_cost / deriv;
  } while (NEWTON_EPSILON < absd(candidate_cost));
  return d_log(candidate_root);
}

/* fpl:129-190: Newton in log-mean space from argmin-1. */
PSD_D double get_smaller_root(const Coef &c, double min_log_mean, double equals) {
  double optimal_log_mean = argmin(c);
  double optimal_cost = get_cost(c, optimal_log_mean);
  double left_cost = get_cost(c, min_log_mean);
  if ((equals < left_cost && left_cost < optimal_cost) ||
      (equals > left_cost && left_cost > optimal_cost)) {
    return min_log_mean - 1;
  }
  double candidate_root = optimal_log_mean - 1;
  double candidate_cost, deriv;
  double closest_positive_cost = PSD_INF, closest_positive_log_mean = PSD_INF;
  double closest_negative_cost = -PSD_INF, closest_negative_log_mean = PSD_INF;
  if (optimal_cost < 0) {
    closest_negative_cost = optimal_cost;
    closest_negative_log_mean = optimal_log_mean;
  } else {
    closest_positive_cost = optimal_cost;
    closest_positive_log_mean = optimal_log_mean;
  }
  int step = 0;
  do {
    /* getCost and getDeriv (fpl:206-234) evaluate the same Linear*exp(x): once here */
    double linear_term;
    if (candidate_root == -PSD_INF) {
      linear_term = 0.0;
    } else {
      linear_term = c.Linear * d_exp(candidate_root);
    }
    double log_term = (c.Log == 0) ? 0.0 : c.Log * candidate_root;
    candidate_cost = (linear_term + log_term + c.Constant) - equals;
    if (0 < candidate_cost && candidate_cost < closest_positive_cost) {
      closest_positive_cost = candidate_cost;
      closest_positive_log_mean = candidate_root;
    }
    if (closest_negative_cost < candidate_cost && candidate_cost < 0) {
      closest_negative_cost = candidate_cost;
      closest_negative_log_mean = candidate_root;
    }
    if (NEWTON_STEPS <= ++step) {
      double between_closest = (closest_positive_log_mean + closest_negative_log_mean) / 2;
      double between_cost = get_cost(c, between_closest) - equals;
      if (absd(between_cost) < absd(candidate_cost)) {
        return between_closest;
      } else {
        return candidate_root;
      }
    }
    deriv = linear_term + c.Log;
    double offset = candidate_cost / deriv;
    candidate_root = candidate_root - offset;
  } while (NEWTON_EPSILON < absd(candidate_cost));
  return candidate_root;
}

/* fpl:862-868 */
PSD_D bool same_funs(const Coef &a, const Coef &b) {
  return a.Linear == b.Linear && a.Log == b.Log &&
         absd(a.Constant - b.Constant) < NEWTON_EPSILON;
}

}  // namespace psd
#endif
