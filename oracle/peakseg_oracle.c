/* peakseg_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see peakseg_oracle.h).
 *
 * Plain-C restatement of the reference's constrained functional-pruning solver.
 * "ref:" comments cite /root/reference/src/<file>:<lines>; fpl = funPieceListLog.cpp,
 * drv = PeakSegFPOPLog.cpp.  Control flow, operation order and operand order follow the
 * reference exactly (they decide the rounding of every fp64 result); data structures do
 * not (flat arrays, indices instead of std::list iterators).
 */
#define _GNU_SOURCE
#include "peakseg_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_LIBM
#define O_EXP(x) exp(x)
#define O_LOG(x) log(x)
#define O_MATH_KIND "libm"
#else
#include "peakseg_detmath.h"
#define O_EXP(x) psd_exp(x)
#define O_LOG(x) psd_log(x)
#define O_MATH_KIND "detmath"
#endif

#define NEWTON_EPSILON 1e-12 /* ref: fpl:9 */
#ifdef ORACLE_NEWTON_STEPS /* tests only: force the step-cap fallback of the root finders */
#define NEWTON_STEPS ORACLE_NEWTON_STEPS
#else
#define NEWTON_STEPS 100     /* ref: fpl:10 */
#endif
#define PREV_NOT_SET (-3)    /* ref: fpl:11 */
#define ABS(x) ((x) < 0 ? -(x) : (x)) /* ref: fpl:13 */

const char *oracle_math_kind(void) { return O_MATH_KIND; }
void oracle_exp_vec(int n, const double *x, double *y) {
  for (int i = 0; i < n; i++) y[i] = O_EXP(x[i]);
}
void oracle_log_vec(int n, const double *x, double *y) {
  for (int i = 0; i < n; i++) y[i] = O_LOG(x[i]);
}
#ifdef PEAKSEG_DETMATH_H
/* x: n quotients, then n numerators, then n divisors; y[i] = psd_div_repair(q, a, b): the repair
 * the device applies to its own division (include/peakseg_detmath.h), run on the host so that a
 * test can hand it quotients that are one unit off */
void oracle_div_repair_vec(int n, const double *x, double *y) {
  for (int i = 0; i < n; i++) y[i] = psd_div_repair(x[i], x[n + i], x[2 * n + i]);
}
#endif

/* ref: funPieceListLog.h:11-34 */
typedef struct {
  double Linear, Log, Constant;
  double min_log_mean, max_log_mean;
  int data_i;
  double prev_log_mean;
} piece_t;

/* ref: funPieceListLog.h:38-41 (piece_list + chromEnd) */
typedef struct {
  piece_t *p;
  int n, cap;
  int chromEnd;
} pwfun_t;

/* Set where the reference would throw / dereference a list sentinel / never terminate. */
static __thread const char *would_throw = NULL;

static void fun_reserve(pwfun_t *f, int cap) {
  if (f->cap < cap) {
    f->cap = cap * 2;
    f->p = (piece_t *)realloc(f->p, sizeof(piece_t) * (size_t)f->cap);
    if (!f->p) abort();
  }
}

static void fun_emplace_back(pwfun_t *f, double li, double lo, double co, double m, double M,
                             int i, double prev) {
  fun_reserve(f, f->n + 1);
  piece_t *q = &f->p[f->n++];
  q->Linear = li;
  q->Log = lo;
  q->Constant = co;
  q->min_log_mean = m;
  q->max_log_mean = M;
  q->data_i = i;
  q->prev_log_mean = prev;
}

static void fun_copy(pwfun_t *dst, const pwfun_t *src) {
  fun_reserve(dst, src->n);
  memcpy(dst->p, src->p, sizeof(piece_t) * (size_t)src->n);
  dst->n = src->n;
  dst->chromEnd = src->chromEnd;
}

/* ---- single-piece arithmetic ------------------------------------------------------- */

/* ref: fpl:192-197 */
static double piece_argmin_mean(const piece_t *q) { return -q->Log / q->Linear; }

/* ref: fpl:199-204 */
static double piece_argmin(const piece_t *q) { return O_LOG(piece_argmin_mean(q)); }

/* ref: fpl:206-222 */
static double piece_getCost(const piece_t *q, double log_mean) {
  double linear_term, log_term;
  if (log_mean == -INFINITY) {
    linear_term = 0.0;
  } else {
    linear_term = q->Linear * O_EXP(log_mean);
  }
  if (q->Log == 0) {
    log_term = 0.0;
  } else {
    log_term = q->Log * log_mean;
  }
  return linear_term + log_term + q->Constant;
}

/* ref: fpl:224-234 */
static double piece_getDeriv(const piece_t *q, double log_mean) {
  double linear_term;
  if (log_mean == -INFINITY) {
    linear_term = 0.0;
  } else {
    linear_term = q->Linear * O_EXP(log_mean);
  }
  return linear_term + q->Log;
}

/* ref: fpl:52-61 */
static double piece_PoissonLoss(const piece_t *q, double mean) {
  double loss_without_log_term = q->Linear * mean + q->Constant;
  if (q->Log == 0) {
    return loss_without_log_term;
  }
  double log_mean_only = O_LOG(mean);
  double log_coef_only = q->Log;
  double product = log_mean_only * log_coef_only;
  return loss_without_log_term + product;
}

/* ref: fpl:63-65 */
static double piece_PoissonDeriv(const piece_t *q, double mean) {
  return q->Linear + q->Log / mean;
}

/* ref: fpl:29-50 */
static int piece_has_two_roots(const piece_t *q, double equals) {
  if (q->Log == 0) {
    would_throw = "has_two_roots on degenerate linear piece (fpl:34)";
    return 0;
  }
  double optimal_mean = piece_argmin_mean(q);
  double optimal_log_mean = O_LOG(optimal_mean);
  double optimal_cost = piece_getCost(q, optimal_log_mean);
  double optimal_cost2 = piece_PoissonLoss(q, optimal_mean);
  if (0 < q->Linear) {
    return optimal_cost + NEWTON_EPSILON < equals && optimal_cost2 + NEWTON_EPSILON < equals;
  }
  return equals + NEWTON_EPSILON < optimal_cost && equals + NEWTON_EPSILON < optimal_cost2;
}

/* ref: fpl:69-127 (Newton in mean space, returns log of the root) */
static double piece_get_larger_root(const piece_t *q, double equals) {
  double optimal_mean = piece_argmin_mean(q);
  double optimal_cost = piece_PoissonLoss(q, optimal_mean);
  double right_cost = piece_getCost(q, q->max_log_mean);
  if ((optimal_cost < right_cost && right_cost < equals) ||
      (optimal_cost > right_cost && right_cost > equals)) {
    return q->max_log_mean + 1;
  }
  double candidate_root = optimal_mean + 1;
  double candidate_cost, possibly_outside, deriv;
  double closest_positive_cost = INFINITY, closest_positive_mean = INFINITY;
  double closest_negative_cost = -INFINITY, closest_negative_mean = INFINITY;
  if (optimal_cost < 0) {
    closest_negative_cost = optimal_cost;
    closest_negative_mean = optimal_mean;
  } else {
    closest_positive_cost = optimal_cost;
    closest_positive_mean = optimal_mean;
  }
  int step = 0;
  do {
    candidate_cost = piece_PoissonLoss(q, candidate_root) - equals;
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
      double between_cost = piece_PoissonLoss(q, between_closest) - equals;
      if (ABS(between_cost) < ABS(candidate_cost)) {
        return O_LOG(between_closest);
      } else {
        return O_LOG(candidate_root);
      }
    }
    deriv = piece_PoissonDeriv(q, candidate_root);
    possibly_outside = candidate_root - candidate_cost / deriv;
    candidate_root = possibly_outside;
  } while (NEWTON_EPSILON < ABS(candidate_cost));
  return O_LOG(candidate_root);
}

/* ref: fpl:129-190 (Newton in log-mean space) */
static double piece_get_smaller_root(const piece_t *q, double equals) {
  double optimal_log_mean = piece_argmin(q);
  double optimal_cost = piece_getCost(q, optimal_log_mean);
  double left_cost = piece_getCost(q, q->min_log_mean);
  if ((equals < left_cost && left_cost < optimal_cost) ||
      (equals > left_cost && left_cost > optimal_cost)) {
    return q->min_log_mean - 1;
  }
  double candidate_root = optimal_log_mean - 1;
  double candidate_cost, possibly_outside, deriv;
  double closest_positive_cost = INFINITY, closest_positive_log_mean = INFINITY;
  double closest_negative_cost = -INFINITY, closest_negative_log_mean = INFINITY;
  if (optimal_cost < 0) {
    closest_negative_cost = optimal_cost;
    closest_negative_log_mean = optimal_log_mean;
  } else {
    closest_positive_cost = optimal_cost;
    closest_positive_log_mean = optimal_log_mean;
  }
  int step = 0;
  double offset;
  do {
    candidate_cost = piece_getCost(q, candidate_root) - equals;
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
      double between_cost = piece_getCost(q, between_closest) - equals;
      if (ABS(between_cost) < ABS(candidate_cost)) {
        return between_closest;
      } else {
        return candidate_root;
      }
    }
    deriv = piece_getDeriv(q, candidate_root);
    offset = candidate_cost / deriv;
    possibly_outside = candidate_root - offset;
    candidate_root = possibly_outside;
  } while (NEWTON_EPSILON < ABS(candidate_cost));
  return candidate_root;
}

/* ---- piecewise-function operations ------------------------------------------------- */

/* ref: fpl:236-437.  out(x) = min over y <= x of in(y). */
static void fun_set_to_min_less_of(pwfun_t *out, const pwfun_t *input) {
  out->n = 0;
  const piece_t *in = input->p;
  int n = input->n;
  int i = 0;
  double prev_min_cost = INFINITY;
  double prev_min_log_mean = in[0].min_log_mean;
  double prev_best_log_mean = INFINITY;
  while (i != n) {
    const piece_t *it = &in[i];
    double left_cost = piece_getCost(it, it->min_log_mean);
    double right_cost = piece_getCost(it, it->max_log_mean);
    if (prev_min_cost == INFINITY) {
      /* look for a minimum achieved in this interval. */
      double next_left_cost = INFINITY;
      int next_i = i + 1;
      if (it->Log == 0) {
        /* degenerate linear piece: increasing or numerically constant (fpl:256-308) */
        double right_left_diff = right_cost - left_cost;
        int right_left_equal = right_left_diff < NEWTON_EPSILON;
        int next_cost_more_than_left;
        if (next_i == n) {
          next_cost_more_than_left = 1;
        } else {
          next_left_cost = piece_getCost(&in[next_i], in[next_i].min_log_mean);
          double next_left_diff = next_left_cost - left_cost;
          next_cost_more_than_left = NEWTON_EPSILON < next_left_diff;
        }
        if (next_cost_more_than_left && !right_left_equal) {
          prev_min_cost = left_cost;
          prev_best_log_mean = it->min_log_mean;
        } else {
          fun_emplace_back(out, it->Linear, it->Log, it->Constant, prev_min_log_mean,
                           it->max_log_mean, PREV_NOT_SET, INFINITY);
          prev_min_log_mean = it->max_log_mean;
        }
      } else { /* fpl:309-366 */
        double mu = piece_argmin(it);
        double mu_cost = piece_getCost(it, mu);
        int next_ok;
        if (next_i == n) {
          next_ok = 1;
        } else {
          next_left_cost = piece_getCost(&in[next_i], in[next_i].min_log_mean);
          next_ok = NEWTON_EPSILON < next_left_cost - mu_cost;
        }
        int cost_ok = NEWTON_EPSILON < right_cost - mu_cost && next_ok;
        if (mu <= it->min_log_mean && cost_ok) {
          /* minimum at or before the left end: increasing piece, start a constant. */
          prev_min_cost = piece_getCost(it, it->min_log_mean);
          prev_best_log_mean = it->min_log_mean;
        } else if (mu < it->max_log_mean && cost_ok) {
          /* minimum inside: keep the decreasing part, start a constant at mu. */
          if (prev_min_log_mean < mu) {
            fun_emplace_back(out, it->Linear, it->Log, it->Constant, prev_min_log_mean, mu,
                             PREV_NOT_SET, INFINITY);
          }
          prev_min_log_mean = mu;
          prev_best_log_mean = mu;
          prev_min_cost = mu_cost;
        } else {
          /* minimum after the interval: decreasing piece, keep as is. */
          fun_emplace_back(out, it->Linear, it->Log, it->Constant, prev_min_log_mean,
                           it->max_log_mean, PREV_NOT_SET, INFINITY);
          prev_min_log_mean = it->max_log_mean;
        }
      }
    } else { /* prev_min_cost is finite: look for where the constant is crossed (fpl:367-422) */
      if (it->Log == 0) {
        if (it->Linear < 0) {
          would_throw = "decreasing degenerate linear piece in min-less (fpl:380)";
          return;
        }
      } else {
        if (piece_has_two_roots(it, prev_min_cost)) {
          double mu = piece_get_smaller_root(it, prev_min_cost);
          if (it->min_log_mean < mu && mu < it->max_log_mean) {
            fun_emplace_back(out, 0, 0, prev_min_cost, prev_min_log_mean, mu, PREV_NOT_SET,
                             prev_best_log_mean);
            prev_min_cost = INFINITY;
            prev_min_log_mean = mu;
            i--; /* revisit this piece in search mode */
          }
        }
        if (right_cost <= prev_min_cost + NEWTON_EPSILON && prev_min_cost < INFINITY) {
          /* constant ends exactly/numerically on the right end. */
          fun_emplace_back(out, 0, 0, prev_min_cost, prev_min_log_mean, it->max_log_mean,
                           PREV_NOT_SET, prev_best_log_mean);
          prev_min_cost = INFINITY;
          prev_min_log_mean = it->max_log_mean;
        }
      }
    }
    i++;
  }
  if (prev_min_cost < INFINITY) {
    /* ending on a constant piece (fpl:429-436) */
    fun_emplace_back(out, 0, 0, prev_min_cost, prev_min_log_mean, in[n - 1].max_log_mean,
                     PREV_NOT_SET, prev_best_log_mean);
  }
}

/* emplace_front is emulated by filling a scratch array from its end. */
typedef struct {
  piece_t *p;
  int cap, head; /* pieces live in p[head..cap) */
} frontbuf_t;

static void front_emplace(frontbuf_t *b, double li, double lo, double co, double m, double M,
                          int i, double prev) {
  if (b->head == 0) {
    would_throw = "oracle internal: min-more scratch exhausted";
    return;
  }
  piece_t *q = &b->p[--b->head];
  q->Linear = li;
  q->Log = lo;
  q->Constant = co;
  q->min_log_mean = m;
  q->max_log_mean = M;
  q->data_i = i;
  q->prev_log_mean = prev;
}

/* ref: fpl:439-616.  out(x) = min over y >= x of in(y). */
static void fun_set_to_min_more_of(pwfun_t *out, const pwfun_t *input) {
  const piece_t *in = input->p;
  int n = input->n;
  static __thread frontbuf_t fb = {NULL, 0, 0};
  int need = 2 * n + 4;
  if (fb.cap < need) {
    fb.cap = need * 2;
    fb.p = (piece_t *)realloc(fb.p, sizeof(piece_t) * (size_t)fb.cap);
    if (!fb.p) abort();
  }
  fb.head = fb.cap;
  int i = n;
  double prev_min_cost = INFINITY;
  double prev_max_log_mean = in[n - 1].max_log_mean;
  double prev_best_log_mean = INFINITY;
  while (i != 0) {
    i--;
    const piece_t *it = &in[i];
    if (prev_min_cost == INFINITY) {
      if (it->Log == 0) {
        /* degenerate linear: increasing or numerically constant, store (fpl:458-467) */
        front_emplace(&fb, it->Linear, it->Log, it->Constant, it->min_log_mean,
                      prev_max_log_mean, PREV_NOT_SET, INFINITY);
        prev_max_log_mean = it->min_log_mean;
      } else { /* fpl:468-548 */
        double mu = piece_argmin(it);
        double mu_cost = piece_getCost(it, mu);
        int prev_ok;
        if (i == 0) {
          prev_ok = 1;
        } else {
          const piece_t *prev_it = &in[i - 1];
          double prev_cost_right = piece_getCost(prev_it, prev_it->max_log_mean);
          prev_ok = NEWTON_EPSILON < prev_cost_right - mu_cost;
        }
        double this_cost_left = piece_getCost(it, it->min_log_mean);
        if (it->max_log_mean <= mu) {
          double this_cost_right = piece_getCost(it, it->max_log_mean);
          double this_cost_diff = this_cost_left - this_cost_right;
          if (NEWTON_EPSILON < this_cost_diff) {
            /* decreasing piece: start a constant from the right end. */
            prev_min_cost = this_cost_right;
            prev_best_log_mean = it->max_log_mean;
          } else {
            /* numerically constant: store it. */
            front_emplace(&fb, it->Linear, it->Log, it->Constant, it->min_log_mean,
                          prev_max_log_mean, PREV_NOT_SET, INFINITY);
            prev_max_log_mean = it->min_log_mean;
          }
        } else if (it->min_log_mean < mu && NEWTON_EPSILON < this_cost_left - mu_cost &&
                   prev_ok) {
          /* minimum inside: keep the increasing part, start a constant at mu. */
          if (mu < prev_max_log_mean) {
            front_emplace(&fb, it->Linear, it->Log, it->Constant, mu, prev_max_log_mean,
                          PREV_NOT_SET, INFINITY);
          }
          prev_max_log_mean = mu;
          prev_best_log_mean = mu;
          prev_min_cost = mu_cost;
        } else {
          /* minimum before the interval: increasing piece, keep as is. */
          front_emplace(&fb, it->Linear, it->Log, it->Constant, it->min_log_mean,
                        prev_max_log_mean, PREV_NOT_SET, INFINITY);
          prev_max_log_mean = it->min_log_mean;
        }
      }
    } else { /* prev_min_cost finite: where is the constant crossed? (fpl:549-602) */
      double left_cost = piece_getCost(it, it->min_log_mean);
      double mu = INFINITY;
      if (it->Log == 0) {
        /* degenerate linear: one intersection point. */
        mu = O_LOG((prev_min_cost - it->Constant) / it->Linear);
      } else {
        if (piece_has_two_roots(it, prev_min_cost)) {
          mu = piece_get_larger_root(it, prev_min_cost);
        }
      }
      if (it->min_log_mean < mu && mu < it->max_log_mean) {
        front_emplace(&fb, 0, 0, prev_min_cost, mu, prev_max_log_mean, PREV_NOT_SET,
                      prev_best_log_mean);
        prev_min_cost = INFINITY;
        prev_max_log_mean = mu;
        i++; /* revisit this piece in search mode */
      } else if (left_cost <= prev_min_cost + NEWTON_EPSILON) {
        /* constant ends exactly/numerically on the left end. */
        front_emplace(&fb, 0, 0, prev_min_cost, it->min_log_mean, prev_max_log_mean,
                      PREV_NOT_SET, prev_best_log_mean);
        prev_min_cost = INFINITY;
        prev_max_log_mean = it->min_log_mean;
      }
    }
    if (would_throw) return;
  }
  if (prev_min_cost < INFINITY) {
    /* ending on a constant piece (fpl:608-615); `it` is the first piece here. */
    front_emplace(&fb, 0, 0, prev_min_cost, in[0].min_log_mean, prev_max_log_mean,
                  PREV_NOT_SET, prev_best_log_mean);
  }
  int m = fb.cap - fb.head;
  fun_reserve(out, m);
  memcpy(out->p, fb.p + fb.head, sizeof(piece_t) * (size_t)m);
  out->n = m;
}

/* ref: fpl:618-625 */
static void fun_add(pwfun_t *f, double Linear, double Log, double Constant) {
  for (int i = 0; i < f->n; i++) {
    f->p[i].Linear += Linear;
    f->p[i].Log += Log;
    f->p[i].Constant += Constant;
  }
}

/* ref: fpl:627-634 */
static void fun_multiply(pwfun_t *f, double x) {
  for (int i = 0; i < f->n; i++) {
    f->p[i].Linear *= x;
    f->p[i].Log *= x;
    f->p[i].Constant *= x;
  }
}

/* ref: fpl:636-641 */
static void fun_set_prev_seg_end(pwfun_t *f, int prev_seg_end) {
  for (int i = 0; i < f->n; i++) f->p[i].data_i = prev_seg_end;
}

/* ref: fpl:643-653 */
static void fun_findMean(const pwfun_t *f, double log_mean, int *seg_end, double *prev_log_mean) {
  for (int i = 0; i < f->n; i++) {
    const piece_t *it = &f->p[i];
    if (it->min_log_mean <= log_mean && log_mean <= it->max_log_mean) {
      *seg_end = it->data_i;
      *prev_log_mean = it->prev_log_mean;
      return;
    }
  }
}

/* ref: fpl:689-712 */
static void fun_Minimize(const pwfun_t *f, double *best_cost, double *best_log_mean, int *data_i,
                         double *prev_log_mean) {
  double candidate_cost, candidate_log_mean;
  *best_cost = INFINITY;
  for (int i = 0; i < f->n; i++) {
    const piece_t *it = &f->p[i];
    candidate_log_mean = piece_argmin(it);
    if (candidate_log_mean < it->min_log_mean) {
      candidate_log_mean = it->min_log_mean;
    } else if (it->max_log_mean < candidate_log_mean) {
      candidate_log_mean = it->max_log_mean;
    }
    candidate_cost = piece_getCost(it, candidate_log_mean);
    if (candidate_cost < *best_cost) {
      *best_cost = candidate_cost;
      *best_log_mean = candidate_log_mean;
      *data_i = it->data_i;
      *prev_log_mean = it->prev_log_mean;
    }
  }
}

/* ref: fpl:862-868 */
static int sameFuns(const piece_t *a, const piece_t *b) {
  return a->Linear == b->Linear && a->Log == b->Log &&
         ABS(a->Constant - b->Constant) < NEWTON_EPSILON;
}

/* ref: fpl:1261-1285 */
static void fun_push_piece(pwfun_t *out, const piece_t *it, double min_log_mean,
                           double max_log_mean) {
  if (max_log_mean <= min_log_mean) {
    return;
  }
  if (out->n) {
    piece_t *last = &out->p[out->n - 1];
    if (sameFuns(last, it) && it->prev_log_mean == last->prev_log_mean &&
        it->data_i == last->data_i) {
      last->max_log_mean = max_log_mean;
      return;
    }
  }
  fun_emplace_back(out, it->Linear, it->Log, it->Constant, min_log_mean, max_log_mean,
                   it->data_i, it->prev_log_mean);
}

/* A neighbour index outside [0,n) is the std::list sentinel in the reference: its fields
 * are garbage there, so the oracle refuses to guess. */
static const piece_t *nb(const pwfun_t *f, int i) {
  if (i < 0 || i >= f->n) {
    would_throw = "push_min_pieces would read a std::list sentinel (fpl:878-931)";
    return &f->p[0];
  }
  return &f->p[i];
}

/* ref: fpl:870-1259.  Emit the pointwise minimum of fun1[i1], fun2[i2] on the overlap of
 * their intervals (1 to 3 pieces). */
static void fun_push_min_pieces(pwfun_t *out, const pwfun_t *fun1, const pwfun_t *fun2, int i1,
                                int i2) {
  const piece_t *it1 = &fun1->p[i1], *it2 = &fun2->p[i2];
  int same_at_left;
  double last_min_log_mean;
  if (it1->min_log_mean < it2->min_log_mean) {
    same_at_left = sameFuns(nb(fun2, i2 - 1), it1);
    last_min_log_mean = it2->min_log_mean;
  } else {
    last_min_log_mean = it1->min_log_mean;
    if (it2->min_log_mean < it1->min_log_mean) {
      same_at_left = sameFuns(nb(fun1, i1 - 1), it2);
    } else {
      if (i1 == 0 && i2 == 0) {
        same_at_left = 0;
      } else {
        same_at_left = sameFuns(nb(fun1, i1 - 1), nb(fun2, i2 - 1));
      }
    }
  }
  int same_at_right;
  double first_max_log_mean;
  if (it1->max_log_mean < it2->max_log_mean) {
    same_at_right = sameFuns(nb(fun1, i1 + 1), it2);
    first_max_log_mean = it1->max_log_mean;
  } else {
    first_max_log_mean = it2->max_log_mean;
    if (it2->max_log_mean < it1->max_log_mean) {
      same_at_right = sameFuns(it1, nb(fun2, i2 + 1));
    } else {
      if (i1 + 1 == fun1->n && i2 + 1 == fun2->n) {
        same_at_right = 0;
      } else {
        same_at_right = sameFuns(nb(fun1, i1 + 1), nb(fun2, i2 + 1));
      }
    }
  }
  if (would_throw) return;
  if (last_min_log_mean == first_max_log_mean) {
    return; /* interval of size 0 (fpl:933-944) */
  }
  if (sameFuns(it1, it2)) {
    fun_push_piece(out, it1, last_min_log_mean, first_max_log_mean);
    return;
  }
  piece_t diff_piece;
  diff_piece.Linear = it1->Linear - it2->Linear;
  diff_piece.Log = it1->Log - it2->Log;
  diff_piece.Constant = it1->Constant - it2->Constant;
  diff_piece.min_log_mean = last_min_log_mean;
  diff_piece.max_log_mean = first_max_log_mean;
  diff_piece.data_i = -5;
  diff_piece.prev_log_mean = 0.0; /* `false` in the reference (fpl:957) */
  /* midpoint evaluated in mean space, robust to -Inf (fpl:958-961) */
  double mid_mean = (O_EXP(first_max_log_mean) + O_EXP(last_min_log_mean)) / 2;
  double cost_diff_mid = piece_getCost(&diff_piece, O_LOG(mid_mean));
  if (same_at_left && same_at_right) {
    if (cost_diff_mid < 0) {
      fun_push_piece(out, it1, last_min_log_mean, first_max_log_mean);
    } else {
      fun_push_piece(out, it2, last_min_log_mean, first_max_log_mean);
    }
    return;
  }
  if (diff_piece.Log == 0) { /* no root finding needed (fpl:973-1019) */
    if (diff_piece.Linear == 0) {
      if (diff_piece.Constant < 0) {
        fun_push_piece(out, it1, last_min_log_mean, first_max_log_mean);
      } else {
        fun_push_piece(out, it2, last_min_log_mean, first_max_log_mean);
      }
      return;
    }
    if (diff_piece.Constant == 0) {
      if (diff_piece.Linear < 0) {
        fun_push_piece(out, it1, last_min_log_mean, first_max_log_mean);
      } else {
        fun_push_piece(out, it2, last_min_log_mean, first_max_log_mean);
      }
      return;
    }
    double log_mean_at_equal_cost = O_LOG(-diff_piece.Constant / diff_piece.Linear);
    if (last_min_log_mean < log_mean_at_equal_cost &&
        log_mean_at_equal_cost < first_max_log_mean) {
      if (0 < diff_piece.Linear) {
        fun_push_piece(out, it1, last_min_log_mean, log_mean_at_equal_cost);
        fun_push_piece(out, it2, log_mean_at_equal_cost, first_max_log_mean);
      } else {
        fun_push_piece(out, it2, last_min_log_mean, log_mean_at_equal_cost);
        fun_push_piece(out, it1, log_mean_at_equal_cost, first_max_log_mean);
      }
      return;
    }
    if (cost_diff_mid < 0) {
      fun_push_piece(out, it1, last_min_log_mean, first_max_log_mean);
    } else {
      fun_push_piece(out, it2, last_min_log_mean, first_max_log_mean);
    }
    return;
  }
  double cost_diff_left = piece_getCost(&diff_piece, last_min_log_mean);
  double cost_diff_right = piece_getCost(&diff_piece, first_max_log_mean);
  int two_roots = piece_has_two_roots(&diff_piece, 0.0);
  double smaller_log_mean = INFINITY, larger_log_mean = INFINITY;
  if (two_roots) {
    smaller_log_mean = piece_get_smaller_root(&diff_piece, 0.0);
    larger_log_mean = piece_get_larger_root(&diff_piece, 0.0);
  }
  if (same_at_right) { /* fpl:1029-1093 */
    if (two_roots) {
      double log_mean_at_crossing = smaller_log_mean;
      /* log_mean_between_zeros / cost_between_zeros (fpl:1035-1036) are only printed */
      double log_mean_at_optimum = piece_argmin(&diff_piece);
      if (last_min_log_mean < log_mean_at_crossing &&
          log_mean_at_crossing < log_mean_at_optimum &&
          log_mean_at_optimum < first_max_log_mean) {
        if (cost_diff_left < 0) {
          fun_push_piece(out, it1, last_min_log_mean, log_mean_at_crossing);
          fun_push_piece(out, it2, log_mean_at_crossing, first_max_log_mean);
        } else {
          fun_push_piece(out, it2, last_min_log_mean, log_mean_at_crossing);
          fun_push_piece(out, it1, log_mean_at_crossing, first_max_log_mean);
        }
        return;
      }
      int it1_smaller_at_mean0 = 0 < diff_piece.Log;
      if (log_mean_at_crossing < last_min_log_mean) {
        if (it1_smaller_at_mean0) {
          fun_push_piece(out, it2, last_min_log_mean, first_max_log_mean);
        } else {
          fun_push_piece(out, it1, last_min_log_mean, first_max_log_mean);
        }
      } else {
        if (it1_smaller_at_mean0) {
          fun_push_piece(out, it1, last_min_log_mean, first_max_log_mean);
        } else {
          fun_push_piece(out, it2, last_min_log_mean, first_max_log_mean);
        }
      }
      return;
    }
    if (cost_diff_mid < 0) {
      fun_push_piece(out, it1, last_min_log_mean, first_max_log_mean);
    } else {
      fun_push_piece(out, it2, last_min_log_mean, first_max_log_mean);
    }
    return;
  }
  if (same_at_left) { /* fpl:1094-1123 */
    if (two_roots) {
      double log_mean_at_crossing = larger_log_mean;
      double log_mean_at_optimum = piece_argmin(&diff_piece);
      if (last_min_log_mean < log_mean_at_optimum &&
          log_mean_at_optimum < log_mean_at_crossing &&
          log_mean_at_crossing < first_max_log_mean) {
        if (cost_diff_right < 0) {
          fun_push_piece(out, it2, last_min_log_mean, log_mean_at_crossing);
          fun_push_piece(out, it1, log_mean_at_crossing, first_max_log_mean);
        } else {
          fun_push_piece(out, it1, last_min_log_mean, log_mean_at_crossing);
          fun_push_piece(out, it2, log_mean_at_crossing, first_max_log_mean);
        }
        return;
      }
    }
    if (cost_diff_mid < 0) {
      fun_push_piece(out, it1, last_min_log_mean, first_max_log_mean);
    } else {
      fun_push_piece(out, it2, last_min_log_mean, first_max_log_mean);
    }
    return;
  }
  /* equal on neither side: 0, 1 or 2 crossings inside (fpl:1124-1258) */
  double first_log_mean = INFINITY, second_log_mean = INFINITY;
  if (two_roots) {
    int larger_inside =
        last_min_log_mean < larger_log_mean && larger_log_mean < first_max_log_mean;
    int smaller_inside = last_min_log_mean < smaller_log_mean &&
                         0 < O_EXP(smaller_log_mean) && smaller_log_mean < first_max_log_mean;
    if (larger_inside) {
      if (smaller_inside && smaller_log_mean < larger_log_mean) {
        first_log_mean = smaller_log_mean;
        second_log_mean = larger_log_mean;
      } else {
        first_log_mean = larger_log_mean;
      }
    } else {
      if (smaller_inside) {
        first_log_mean = smaller_log_mean;
      }
    }
  }
  if (second_log_mean != INFINITY) {
    int it1_larger_before;
    if (second_log_mean - first_log_mean < first_log_mean - last_min_log_mean) {
      double before_mean = (O_EXP(last_min_log_mean) + O_EXP(first_log_mean)) / 2;
      double cost_diff_before = piece_getCost(&diff_piece, O_LOG(before_mean));
      if (cost_diff_before < 0) {
        it1_larger_before = 1;
      } else {
        it1_larger_before = 0;
      }
    } else {
      double log_mean_between = (first_log_mean + second_log_mean) / 2;
      double cost_diff_between = piece_getCost(&diff_piece, log_mean_between);
      if (cost_diff_between < 0) {
        it1_larger_before = 0;
      } else {
        it1_larger_before = 1;
      }
    }
    if (it1_larger_before) {
      fun_push_piece(out, it1, last_min_log_mean, first_log_mean);
      fun_push_piece(out, it2, first_log_mean, second_log_mean);
      fun_push_piece(out, it1, second_log_mean, first_max_log_mean);
    } else {
      fun_push_piece(out, it2, last_min_log_mean, first_log_mean);
      fun_push_piece(out, it1, first_log_mean, second_log_mean);
      fun_push_piece(out, it2, second_log_mean, first_max_log_mean);
    }
  } else if (first_log_mean != INFINITY) {
    double before_mean = (O_EXP(last_min_log_mean) + O_EXP(first_log_mean)) / 2;
    double cost_diff_before = piece_getCost(&diff_piece, O_LOG(before_mean));
    double after_mean = (first_max_log_mean + first_log_mean) / 2; /* a log-mean (fpl:1216) */
    double cost_diff_after = piece_getCost(&diff_piece, after_mean);
    if (cost_diff_before < 0) {
      if (cost_diff_after < 0) {
        fun_push_piece(out, it1, last_min_log_mean, first_max_log_mean);
      } else {
        fun_push_piece(out, it1, last_min_log_mean, first_log_mean);
        fun_push_piece(out, it2, first_log_mean, first_max_log_mean);
      }
    } else {
      if (cost_diff_after < 0) {
        fun_push_piece(out, it2, last_min_log_mean, first_log_mean);
        fun_push_piece(out, it1, first_log_mean, first_max_log_mean);
      } else {
        fun_push_piece(out, it2, last_min_log_mean, first_max_log_mean);
      }
    }
  } else {
    double cost_diff;
    if (ABS(cost_diff_mid) < NEWTON_EPSILON) {
      cost_diff = cost_diff_right;
    } else {
      cost_diff = cost_diff_mid;
    }
    if (cost_diff < 0) {
      fun_push_piece(out, it1, last_min_log_mean, first_max_log_mean);
    } else {
      fun_push_piece(out, it2, last_min_log_mean, first_max_log_mean);
    }
  }
}

/* ref: fpl:832-860 */
static void fun_set_to_min_env_of(pwfun_t *out, const pwfun_t *fun1, const pwfun_t *fun2) {
  int i1 = 0, i2 = 0;
  out->n = 0;
  while (i1 != fun1->n && i2 != fun2->n) {
    fun_push_min_pieces(out, fun1, fun2, i1, i2);
    if (would_throw) return;
    if (out->n == 0) {
      would_throw = "min-env produced no piece: reference reads back() of an empty list";
      return;
    }
    double last_max_log_mean = out->p[out->n - 1].max_log_mean;
    int moved = 0;
    if (fun1->p[i1].max_log_mean == last_max_log_mean) {
      i1++;
      moved = 1;
    }
    if (fun2->p[i2].max_log_mean == last_max_log_mean) {
      i2++;
      moved = 1;
    }
    if (!moved) {
      would_throw = "min-env made no progress: the reference would loop forever";
      return;
    }
  }
}

/* ---- DiskVector: the reference's on-disk cost-function store (drv:12-56,76-141) ------ */

typedef struct {
  FILE *db;
  int n_entries;
  int failed;
} diskvec_t;

/* std::streampos is 16 bytes on this ABI: 8-byte offset + 8-byte (zero) mbstate_t. */
typedef struct {
  int64_t off;
  int64_t state;
} streampos_t;

static int dv_write_bytes(diskvec_t *dv, const void *p, size_t size) {
  if (!dv->db || fwrite(p, 1, size, dv->db) != size) {
    dv->failed = 1;
    return -1;
  }
  return 0;
}

/* ref: drv:81-90 */
static int dv_init(diskvec_t *dv, const char *filename, int N) {
  dv->n_entries = N;
  dv->failed = 0;
  dv->db = fopen(filename, "w+b");
  streampos_t beginning = {0, 0};
  for (int i = 0; i < N; i++) {
    if (dv_write_bytes(dv, &beginning, sizeof beginning)) return -1;
  }
  return 0;
}

/* ref: drv:121-140.  Record = int size, then int n_pieces, int chromEnd,
 * n x {double max_log_mean, int data_i, double prev_log_mean} packed (drv:12-34). */
static int dv_write(diskvec_t *dv, int element, const pwfun_t *fun) {
  streampos_t pos;
  if (fseeko(dv->db, (off_t)sizeof(streampos_t) * element, SEEK_SET)) return -1;
  if (fread(&pos, sizeof pos, 1, dv->db) != 1) return -1;
  if (pos.off != 0) {
    would_throw = "AlreadyWrittenException (drv:124)";
    return -1;
  }
  if (fseeko(dv->db, 0, SEEK_END)) return -1;
  pos.off = ftello(dv->db);
  pos.state = 0;
  int size = (int)((2 * sizeof(double) + sizeof(int)) * (size_t)fun->n + sizeof(int) * 2);
  if (dv_write_bytes(dv, &size, sizeof size)) return -1;
  char *buffer = (char *)malloc((size_t)size), *p = buffer;
  int n_pieces = fun->n;
  memcpy(p, &n_pieces, sizeof(int));
  p += sizeof(int);
  memcpy(p, &fun->chromEnd, sizeof(int));
  p += sizeof(int);
  for (int i = 0; i < fun->n; i++) {
    memcpy(p, &fun->p[i].max_log_mean, sizeof(double));
    p += sizeof(double);
    memcpy(p, &fun->p[i].data_i, sizeof(int));
    p += sizeof(int);
    memcpy(p, &fun->p[i].prev_log_mean, sizeof(double));
    p += sizeof(double);
  }
  int rc = dv_write_bytes(dv, buffer, (size_t)size);
  free(buffer);
  if (rc) return -1;
  if (fseeko(dv->db, (off_t)sizeof(streampos_t) * element, SEEK_SET)) return -1;
  if (dv_write_bytes(dv, &pos, sizeof pos)) return -1;
  return 0;
}

/* ref: drv:106-120 + PiecewiseFunRestore drv:36-56 (min_log_mean = previous max, first -Inf;
 * coefficients are not stored). */
static int dv_read(diskvec_t *dv, int element, pwfun_t *fun) {
  streampos_t pos;
  if (fseeko(dv->db, (off_t)sizeof(streampos_t) * element, SEEK_SET)) return -1;
  if (fread(&pos, sizeof pos, 1, dv->db) != 1) return -1;
  if (pos.off == 0) {
    would_throw = "UndefinedReadException (drv:109)";
    return -1;
  }
  if (fseeko(dv->db, (off_t)pos.off, SEEK_SET)) return -1;
  int size;
  if (fread(&size, sizeof size, 1, dv->db) != 1) return -1;
  char *buffer = (char *)malloc((size_t)size), *p = buffer;
  if (fread(buffer, 1, (size_t)size, dv->db) != (size_t)size) {
    free(buffer);
    return -1;
  }
  int n_pieces;
  memcpy(&n_pieces, p, sizeof(int));
  p += sizeof(int);
  memcpy(&fun->chromEnd, p, sizeof(int));
  p += sizeof(int);
  fun->n = 0;
  double min_log_mean = -INFINITY;
  for (int i = 0; i < n_pieces; i++) {
    double max_log_mean, prev_log_mean;
    int data_i;
    memcpy(&max_log_mean, p, sizeof(double));
    p += sizeof(double);
    memcpy(&data_i, p, sizeof(int));
    p += sizeof(int);
    memcpy(&prev_log_mean, p, sizeof(double));
    p += sizeof(double);
    fun_emplace_back(fun, 0, 0, 0, min_log_mean, max_log_mean, data_i, prev_log_mean);
    min_log_mean = max_log_mean;
  }
  free(buffer);
  return 0;
}

/* ---- solver driver (drv:143-463) ------------------------------------------------------ */

typedef struct {
  pwfun_t up_cost, down_cost, up_cost_prev, down_cost_prev, min_prev_cost;
  diskvec_t dv;
  FILE *bedGraph_file, *segments_file, *loss_file;
  char *line;
} solver_state_t;

static void solver_cleanup(solver_state_t *s) {
  free(s->up_cost.p);
  free(s->down_cost.p);
  free(s->up_cost_prev.p);
  free(s->down_cost_prev.p);
  free(s->min_prev_cost.p);
  if (s->dv.db) fclose(s->dv.db);
  if (s->bedGraph_file) fclose(s->bedGraph_file);
  if (s->segments_file) fclose(s->segments_file);
  if (s->loss_file) fclose(s->loss_file);
  free(s->line);
}

/* std::stod semantics for the penalty string (drv:147-151): leading whitespace skipped,
 * trailing garbage ignored, no conversion -> invalid_argument -> code 10.  (Out-of-range
 * throws std::out_of_range in the reference, which is not caught there.) */
static int parse_penalty(const char *s, double *out) {
  char *end;
  *out = strtod(s, &end);
  return end == s ? -1 : 0;
}

int oracle_PeakSegFPOP_disk(const char *bedGraph_file_name, const char *penalty_str,
                            const char *db_file_name) {
  would_throw = NULL;
  int penalty_is_Inf = strcmp(penalty_str, "Inf") == 0;
  double penalty;
  if (parse_penalty(penalty_str, &penalty)) {
    return ORACLE_ERROR_PENALTY_NOT_NUMERIC;
  }
  if (penalty_is_Inf) {
    /* special case below */
  } else if (!isfinite(penalty)) {
    return ORACLE_ERROR_PENALTY_NOT_FINITE;
  } else if (penalty < 0) {
    return ORACLE_ERROR_PENALTY_NEGATIVE;
  }
  solver_state_t S;
  memset(&S, 0, sizeof S);
  S.bedGraph_file = fopen(bedGraph_file_name, "r");
  if (!S.bedGraph_file) {
    return ORACLE_ERROR_UNABLE_TO_OPEN_BEDGRAPH;
  }
  size_t line_cap = 0;
  int chromStart, chromEnd = 0, coverage, items, line_i = 0;
  char chrom[100];
  char extra[100] = "";
  double cum_weight_i = 0.0, cum_weight_prev_i = -1.0, cum_weighted_count = 0.0;
  double min_log_mean = INFINITY, max_log_mean = -INFINITY, log_data;
  int data_i = 0;
  double weight;
  int first_chromStart = -1, prev_chromEnd = -1;
  int status = 0;
  /* pass 1 (drv:173-205) */
  while (getline(&S.line, &line_cap, S.bedGraph_file) != -1) {
    line_i++;
    items = sscanf(S.line, "%s %d %d %d%s\n", chrom, &chromStart, &chromEnd, &coverage, extra);
    if (items < 4) {
      printf("problem: %d items on line %d\n", items, line_i);
      status = ORACLE_ERROR_NOT_ENOUGH_COLUMNS;
      goto done;
    }
    if (0 < strlen(extra)) {
      status = ORACLE_ERROR_NON_INTEGER_DATA;
      goto done;
    }
    weight = chromEnd - chromStart;
    cum_weight_i += weight;
    cum_weighted_count += weight * coverage;
    if (line_i == 1) {
      first_chromStart = chromStart;
    } else {
      if (chromStart != prev_chromEnd) {
        status = ORACLE_ERROR_INCONSISTENT_CHROMSTART_CHROMEND;
        goto done;
      }
    }
    prev_chromEnd = chromEnd;
    log_data = O_LOG((double)coverage);
    if (log_data < min_log_mean) {
      min_log_mean = log_data;
    }
    if (max_log_mean < log_data) {
      max_log_mean = log_data;
    }
  }
  int data_count = line_i;
  if (data_count == 0) {
    status = ORACLE_ERROR_NO_DATA;
    goto done;
  }
  double best_cost, best_log_mean, prev_log_mean;
  {
    /* output files are opened (truncated) before the trivial/DP split (drv:212-223) */
    size_t L = strlen(bedGraph_file_name) + strlen(penalty_str) + 64;
    char *name = (char *)malloc(L);
    snprintf(name, L, "%s_penalty=%s_loss.tsv", bedGraph_file_name, penalty_str);
    S.loss_file = fopen(name, "w");
    snprintf(name, L, "%s_penalty=%s_segments.bed", bedGraph_file_name, penalty_str);
    S.segments_file = fopen(name, "w");
    free(name);
  }
  int loss_failed = S.loss_file == NULL, segments_failed = S.segments_file == NULL;
#define SEG(...)                                                       \
  do {                                                                 \
    if (S.segments_file && fprintf(S.segments_file, __VA_ARGS__) < 0) \
      segments_failed = 1;                                             \
  } while (0)
#define LOSS(...)                                              \
  do {                                                         \
    if (S.loss_file && fprintf(S.loss_file, __VA_ARGS__) < 0) \
      loss_failed = 1;                                         \
  } while (0)
  if (penalty_is_Inf || min_log_mean == max_log_mean) {
    /* trivial one-segment model (drv:224-243) */
    if (cum_weighted_count != 0) {
      best_cost = cum_weighted_count * (1 - O_LOG(cum_weighted_count) + O_LOG(cum_weight_i));
    } else {
      best_cost = 0;
    }
    SEG("%s\t%d\t%d\tbackground\t%g\n", chrom, first_chromStart, chromEnd,
        cum_weighted_count / cum_weight_i);
    LOSS("%s\t%d\t%d\t%d\t%d\t%.20g\t%.20g\t%d\t%d\t%d\n", penalty_str, 1, 0, (int)cum_weight_i,
         data_count, best_cost / cum_weight_i, best_cost, 0, 0, 0);
  } else {
    /* dynamic programming (drv:244-455) */
    rewind(S.bedGraph_file);
    if (dv_init(&S.dv, db_file_name, data_count * 2)) {
      status = ORACLE_ERROR_WRITING_COST_FUNCTIONS;
      goto done;
    }
    cum_weight_i = 0;
    double total_intervals = 0.0, max_intervals = 0.0;
    while (getline(&S.line, &line_cap, S.bedGraph_file) != -1) {
      items = sscanf(S.line, "%*s\t%d\t%d\t%d\n", &chromStart, &chromEnd, &coverage);
      weight = chromEnd - chromStart;
      cum_weight_i += weight;
      if (data_i == 0) {
        /* C^down_1(m) = gamma_1(m)/w_1 (drv:266-270) */
        fun_emplace_back(&S.down_cost, 1.0, (double)-coverage, 0.0, min_log_mean, max_log_mean,
                         -1, -5.0);
      } else {
        /* up cost: may come from the previous down cost (drv:273-321) */
        fun_set_to_min_less_of(&S.min_prev_cost, &S.down_cost_prev);
        fun_set_prev_seg_end(&S.min_prev_cost, data_i - 1);
        fun_add(&S.min_prev_cost, 0.0, 0.0, penalty / cum_weight_prev_i);
        if (data_i == 1) {
          fun_copy(&S.up_cost, &S.min_prev_cost);
        } else {
          fun_set_to_min_env_of(&S.up_cost, &S.min_prev_cost, &S.up_cost_prev);
        }
        fun_multiply(&S.up_cost, cum_weight_prev_i);
        fun_add(&S.up_cost, weight, -coverage * weight, 0.0);
        fun_multiply(&S.up_cost, 1 / cum_weight_i);
        /* down cost: may come from the previous up cost, no penalty (drv:324-370) */
        if (data_i == 1) {
          fun_copy(&S.down_cost, &S.down_cost_prev);
        } else {
          fun_set_to_min_more_of(&S.min_prev_cost, &S.up_cost_prev);
          fun_set_prev_seg_end(&S.min_prev_cost, data_i - 1);
          fun_set_to_min_env_of(&S.down_cost, &S.min_prev_cost, &S.down_cost_prev);
        }
        fun_multiply(&S.down_cost, cum_weight_prev_i);
        fun_add(&S.down_cost, weight, -coverage * weight, 0.0);
        fun_multiply(&S.down_cost, 1 / cum_weight_i);
        if (would_throw) {
          fprintf(stderr, "oracle: data_i=%d: %s\n", data_i, would_throw);
          status = ORACLE_ERROR_REFERENCE_WOULD_THROW;
          goto done;
        }
      }
      cum_weight_prev_i = cum_weight_i;
      total_intervals += S.up_cost.n + S.down_cost.n;
      if (max_intervals < S.up_cost.n) {
        max_intervals = S.up_cost.n;
      }
      if (max_intervals < S.down_cost.n) {
        max_intervals = S.down_cost.n;
      }
      fun_copy(&S.up_cost_prev, &S.up_cost);
      fun_copy(&S.down_cost_prev, &S.down_cost);
      S.up_cost.chromEnd = chromEnd;
      S.down_cost.chromEnd = chromEnd;
      if (dv_write(&S.dv, data_i + data_count, &S.down_cost) ||
          (0 < data_i && dv_write(&S.dv, data_i, &S.up_cost))) {
        if (would_throw) {
          fprintf(stderr, "oracle: data_i=%d: %s\n", data_i, would_throw);
          status = ORACLE_ERROR_REFERENCE_WOULD_THROW;
        } else {
          status = ORACLE_ERROR_WRITING_COST_FUNCTIONS;
        }
        goto done;
      }
      data_i++;
    }
    /* decoding / backtrack (drv:399-442) */
    int prev_seg_end = 0;
    int prev_seg_offset = 0;
    fun_Minimize(&S.down_cost, &best_cost, &best_log_mean, &prev_seg_end, &prev_log_mean);
    prev_chromEnd = S.down_cost.chromEnd;
    int n_equality_constraints = 0;
    line_i = 1;
    while (0 <= prev_seg_end) {
      line_i++;
      if (dv_read(&S.dv, prev_seg_offset + prev_seg_end, &S.up_cost)) {
        fprintf(stderr, "oracle: backtrack read failed: %s\n", would_throw ? would_throw : "io");
        status = ORACLE_ERROR_REFERENCE_WOULD_THROW;
        goto done;
      }
      SEG("%s\t%d\t%d\t", chrom, S.up_cost.chromEnd, prev_chromEnd);
      if (prev_seg_offset == 0) {
        prev_seg_offset = data_count;
        SEG("background");
      } else {
        prev_seg_offset = 0;
        SEG("peak");
      }
      SEG("\t%g\n", O_EXP(best_log_mean));
      prev_chromEnd = S.up_cost.chromEnd;
      if (prev_log_mean != INFINITY) {
        best_log_mean = prev_log_mean; /* equality constraint inactive */
      } else {
        n_equality_constraints++;
      }
      fun_findMean(&S.up_cost, best_log_mean, &prev_seg_end, &prev_log_mean);
    }
    SEG("%s\t%d\t%d\tbackground\t%g\n", chrom, first_chromStart, prev_chromEnd,
        O_EXP(best_log_mean));
    int n_peaks = (line_i - 1) / 2;
    LOSS("%.20g\t%d\t%d\t%d\t%d\t%.20g\t%.20g\t%d\t%.20g\t%.20g\n", penalty, line_i, n_peaks,
         (int)cum_weight_i, data_count, best_cost, best_cost * cum_weight_i - penalty * n_peaks,
         n_equality_constraints, total_intervals / (data_count * 2), max_intervals);
  }
  if (S.loss_file && fflush(S.loss_file)) loss_failed = 1;
  if (S.segments_file && fflush(S.segments_file)) segments_failed = 1;
  if (loss_failed) {
    status = ORACLE_ERROR_WRITING_LOSS_OUTPUT;
  } else if (segments_failed) {
    status = ORACLE_ERROR_WRITING_SEGMENTS_OUTPUT;
  }
done:
  solver_cleanup(&S);
  return status;
}
