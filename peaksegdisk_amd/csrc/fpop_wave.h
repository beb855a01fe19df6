/* fpop_wave.h -- wavefront-level operations on piecewise Poisson-loss functions.
 *
 * One gfx950 wavefront (64 lanes) owns one cost function.  A function is a list of pieces
 * in struct-of-arrays form in LDS (an HBM scratch accessor can be plugged into the same
 * templates); lane i works on piece / merged interval i (chunks of 64 for longer lists).
 *
 * These are re-designs, not translations, of the reference's sequential std::list walks
 * (/root/reference/src/funPieceListLog.cpp, "fpl"):
 *   min_less_wave  <- set_to_min_less_of   fpl:236-437
 *   min_more_wave  <- set_to_min_more_of   fpl:439-616
 *   min_env_wave   <- set_to_min_env_of + push_min_pieces + push_piece  fpl:832-860,870-1285
 * Each produces bit-identical lists to the sequential algorithm:
 *   - everything a walk computes about a piece that does not depend on the walk's state
 *     (end costs, argmin, the "search mode" decision) is evaluated by all lanes at once;
 *   - the walk's state machine (search for a minimum / follow a constant) then advances
 *     with ballots; while following a constant every remaining piece tests the crossing
 *     speculatively (root finding in parallel) and the lowest (highest) lane with an event
 *     wins, which is exactly the piece the sequential walk would have stopped at;
 *   - the min-envelope's merged intervals are independent given the two input lists, so
 *     each lane classifies one interval (up to two Newton solves) and emits 1-3 candidate
 *     pieces; candidates are compacted with a ballot/prefix scan.  push_piece's coalescing
 *     compares against the run head with a non-transitive tolerance, so the scan is only
 *     used when every "same function" decision is also a bitwise equality (then comparing
 *     with the predecessor is equivalent); otherwise lane 0 replays the interval list
 *     sequentially (rare; counted in the stats).
 *
 * The three operations return the piece count (>= 0) or -(WERR_* bits).  Each exists as an
 * inline body (*_impl) and an out-of-line wrapper (*_wave): inlining everything -- the LDS
 * and the HBM instantiations, twice -- made a ~200 KB kernel against a 64 KB I-cache, so only
 * the latency build inlines, and only the LDS instantiations (see the end of this file).
 */
#include "fpop_types.h"

/* NO include guard: everything below is compiled once per build variant, into namespace
 * psd::PSD_VARIANT.  The includer defines
 *   PSD_VARIANT       namespace of this variant (lat, thr)
 *   PSD_LDS_CAP       pieces per LDS-resident list
 *   PSD_HELPER_WAVES  defined: 4 waves per workgroup (two chains + their helper waves)
 *   PSD_MATH_VK       defined: exp/log with their constants in vector registers */
#if !defined(PSD_VARIANT) || !defined(PSD_LDS_CAP)
#error "define PSD_VARIANT and PSD_LDS_CAP before including fpop_wave.h / fpop_kernels.h"
#endif
#include "fpop_pieces.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace psd {
namespace PSD_VARIANT {

constexpr int LDS_CAP = PSD_LDS_CAP; /* pieces per LDS-resident list */

/* one piece list, struct-of-arrays (fields of funPieceListLog.h:11-34) */
struct ListStore {
  double Lin[LDS_CAP], Log[LDS_CAP], Con[LDS_CAP], mn[LDS_CAP], mx[LDS_CAP], prv[LDS_CAP];
  int di[LDS_CAP];
};
/* per-wave temporaries, one slot per input piece / merged interval */
struct ScratchStore {
  double lc[LDS_CAP], rc[LDS_CAP];   /* getCost at the piece's left / right end */
  double om[LDS_CAP], mu[LDS_CAP];   /* argmin_mean(), argmin() */
  double muc[LDS_CAP], oc2[LDS_CAP]; /* getCost(argmin()), PoissonLoss(argmin_mean()) */
  int cls[LDS_CAP];
  int iv[2 * LDS_CAP];
};
/* Helper waves (PSD_HELPER_WAVES): every chain's main wave has a second wave that runs the
 * longest dependent chain of the envelope classification concurrently: given the difference
 * piece of every interval (HOP_ROOT, posted as soon as it is formed), it derives the optimum
 * and has_two_roots itself and does the larger-root Newton solves, while the main wave
 * evaluates end costs, midpoint and the smaller-root solves.  Lane k of the helper works on
 * lane k's interval; arguments and results cross through this LDS mailbox.  HOP_BARRIER makes
 * the helper join a workgroup barrier, HOP_EXIT ends it. */
enum {
  HOP_BARRIER = 1, HOP_EXIT = 2, HOP_ROOT = 3,
  /* lists in HBM (functions that outgrew LDS: adversarial data): the helper takes every
   * second chunk of 64 pieces / merged intervals of the chain wave's operation */
  HOP_HBM_COSTS = 4,    /* first pass of min-less / min-more: the odd chunks */
  HOP_HBM_TABLE = 5,    /* merged-interval table: the entries owned by the second function */
  HOP_HBM_CLASSIFY = 6, /* envelope classification: the odd chunks, results left in HBM */
};
struct Mail {
  int seq_cmd, seq_done, op, abort;
  int h_arg[8];   /* arguments of the HBM operations */
  int h_progress; /* HOP_HBM_CLASSIFY: chunks the helper has finished (a flag) */
  int flags[64];
  double d_lin[64], d_log[64], d_con[64], b[64]; /* HOP_ROOT: difference piece, right end */
  double res_large[64];
};

/* the workgroup's LDS: lists 0,1 = up (double-buffered), 2,3 = down, 4,5 = per-wave
 * min-less / min-more result */
struct SharedBlock {
  ListStore list[6];
  ScratchStore sc[2];
  unsigned arrived[2]; /* number of the last end-of-data-point barrier the chain wave reached */
  int n[6];
  int abort_status[3];
  int abort_err[3];
  unsigned long long total_up;
  int max_up;
  int serial[2];
  int spill_slot; /* slot of the HBM spill pool taken by this problem (take_spill_slot) */
  int bt_next;    /* checkpointed store: block the decoding needs next, -1 = done */
  unsigned long long ckpt_ovf; /* checkpointed store: room taken in the overflow pool */
#ifdef PSD_HELPER_WAVES
  Mail mail[2];
#endif
#ifdef PSD_PROFILE
  long long prof[4][N_PROF];
#endif
#ifdef PSD_SPIN_STATS
  int spin_max[4];
#endif
  long long t_begin[2];             /* cycle counter at the start of each chain wave */
  unsigned long long cur_ptr[2][3]; /* where each chain's current arena run lives (ArenaCursor) */
};

PSD_LDS SharedBlock g_sm;

/* Waits between the waves of a workgroup (the flag barrier of a data point, the helper
 * mailboxes, the progress word of a shared envelope) poll an LDS word at most this many times:
 * a wave that never comes turns into an error status instead of a hang.  A poll with its pause
 * is ~100 cycles, so the bound is seconds; the slowest legitimate wait is four orders of
 * magnitude shorter (tests/test_gpu_round4.py measures it with -DPSD_SPIN_STATS). */
constexpr int WAIT_SPIN_LIMIT = 1 << 26;
#ifdef PSD_SPIN_STATS
#define PSD_SPIN_NOTE(spin)                                                           \
  do {                                                                                \
    if (lane_id() == 0 && (spin) > g_sm.spin_max[wave_id()]) g_sm.spin_max[wave_id()] = (spin); \
  } while (0)
#else
#define PSD_SPIN_NOTE(spin) \
  do {                      \
  } while (0)
#endif
#ifdef PSD_HELPER_WAVES
constexpr int MAIL_SPIN_LIMIT = WAIT_SPIN_LIMIT;
/* main wave: wait until the helper has finished the last posted command */
PSD_D bool mail_wait(int chain) {
  Mail &m = g_sm.mail[chain];
  const int want = flag_load(&m.seq_cmd);
  for (int spin = 0; spin < MAIL_SPIN_LIMIT; spin++) {
    if (flag_load(&m.seq_done) == want) {
      PSD_SPIN_NOTE(spin);
      return true;
    }
    spin_pause();
  }
  return false;
}
/* main wave: hand the next command over (arguments already written by the lanes) */
PSD_D void mail_post(int chain, int op) {
  Mail &m = g_sm.mail[chain];
  wave_sync();
  if (lane_id() == 0) {
    m.op = op;
    flag_store(&m.seq_cmd, flag_load(&m.seq_cmd) + 1);
  }
  wave_sync(); /* no lane reads seq_cmd (mail_wait) before lane 0 has advanced it */
}
#endif

#ifdef PSD_PROFILE
#define PSD_PROF_T0()                \
  long long prof_t0_ = cycle_now(); \
  long long prof_sub_ = prof_t0_;   \
  (void)prof_sub_
#define PSD_PROF_ADD(slot)                                             \
  do {                                                                 \
    long long now_ = cycle_now();                                      \
    if (lane_id() == 0) g_sm.prof[wave_id()][slot] += now_ - prof_t0_; \
    prof_t0_ = now_;                                                   \
  } while (0)
#define PSD_PROF_SUB0() prof_sub_ = cycle_now()
#define PSD_PROF_SUB(slot)                                              \
  do {                                                                  \
    long long now_ = cycle_now();                                       \
    if (lane_id() == 0) g_sm.prof[wave_id()][slot] += now_ - prof_sub_; \
    prof_sub_ = now_;                                                   \
  } while (0)
#else
#define PSD_PROF_SUB0() \
  do {                  \
  } while (0)
#define PSD_PROF_SUB(slot) \
  do {                     \
  } while (0)
#define PSD_PROF_T0() \
  do {                \
  } while (0)
#define PSD_PROF_ADD(slot) \
  do {                     \
  } while (0)
#endif
enum {
  PROF_PRE = 0, PROF_WALK = 1, PROF_TABLE = 2, PROF_CLASSIFY = 3, PROF_COMPACT = 4,
  PROF_SCALE = 5, PROF_ARENA = 6, PROF_BARRIER = 7, PROF_SERIAL = 8, PROF_TOTAL = 9,
  PROF_C_LOAD = 10, PROF_C_MID = 11, PROF_C_OPT = 12, PROF_C_SMALL = 13, PROF_C_LARGE = 14,
  PROF_C_TAIL = 15,
  PROF_IT_SPEC = 16, PROF_IT_SMALL = 17, PROF_IT_LARGE = 18, /* wave-level Newton trip counts */
  PROF_IT_ROUNDS = 19, /* walk state-machine rounds */
  PROF_S_ASSIGN = 20, PROF_S_LOAD = 21, PROF_S_NEWTON = 22 /* inside the speculation round */
};
#ifdef PSD_PROFILE
#define PSD_PROF_ITERS(slot, steps)                                 \
  do {                                                              \
    int m_ = 0;                                                     \
    while (ballot((steps) > m_)) m_++;                              \
    if (lane_id() == 0) g_sm.prof[wave_id()][slot] += m_;           \
  } while (0)
#else
#define PSD_PROF_ITERS(slot, steps) \
  do {                             \
  } while (0)
#endif
#ifdef PSD_PROFILE
#define PSD_PROF_COUNT(slot)                                       \
  do {                                                             \
    if (lane_id() == 0) g_sm.prof[wave_id()][slot] += 1;           \
  } while (0)
#else
#define PSD_PROF_COUNT(slot) \
  do {                       \
  } while (0)
#endif

/* accessor of an LDS-resident list: g_sm.list[id], elements off.. */
struct LdsList {
  static constexpr bool in_lds = true;
  int id, off;
  PSD_M double &Lin(int i) const { return g_sm.list[id].Lin[off + i]; }
  PSD_M double &Log(int i) const { return g_sm.list[id].Log[off + i]; }
  PSD_M double &Con(int i) const { return g_sm.list[id].Con[off + i]; }
  PSD_M double &mn(int i) const { return g_sm.list[id].mn[off + i]; }
  PSD_M double &mx(int i) const { return g_sm.list[id].mx[off + i]; }
  PSD_M double &prv(int i) const { return g_sm.list[id].prv[off + i]; }
  PSD_M int &di(int i) const { return g_sm.list[id].di[off + i]; }
  PSD_M LdsList shifted(int d) const {
    LdsList r;
    r.id = id;
    r.off = off + d;
    return r;
  }
  /* Arguments of out-of-line device functions arrive in VGPRs and the compiler must assume
   * they differ between lanes: every loop and branch on them becomes an exec-mask loop.  They
   * are wave-uniform by construction; readfirstlane says so. */
  PSD_M LdsList uniformed() const {
    LdsList r;
    r.id = uniform_i(id);
    r.off = uniform_i(off);
    return r;
  }
};
/* accessor of a wave's LDS scratch arrays */
struct LdsScratch {
  int w;
  PSD_M double &lc(int i) const { return g_sm.sc[w].lc[i]; }
  PSD_M double &rc(int i) const { return g_sm.sc[w].rc[i]; }
  PSD_M double &om(int i) const { return g_sm.sc[w].om[i]; }
  PSD_M double &mu(int i) const { return g_sm.sc[w].mu[i]; }
  PSD_M double &muc(int i) const { return g_sm.sc[w].muc[i]; }
  PSD_M double &oc2(int i) const { return g_sm.sc[w].oc2[i]; }
  PSD_M int &cls(int i) const { return g_sm.sc[w].cls[i]; }
  PSD_M int &iv(int i) const { return g_sm.sc[w].iv[i]; }
  PSD_M int iv_cap() const { return 2 * LDS_CAP; }
  PSD_M LdsScratch uniformed() const {
    LdsScratch r;
    r.w = uniform_i(w);
    return r;
  }
};

/* The same two accessors over HBM: the spill path for functions with more than LDS_CAP
 * pieces (adversarial data, vignettes/Worst_case.Rmd).  `cap` pieces per list. */
struct GlobalList {
  static constexpr bool in_lds = false;
  gdouble *Lin_, *Log_, *Con_, *mn_, *mx_, *prv_;
  gint *di_;
  PSD_M gdouble &Lin(int i) const { return Lin_[i]; }
  PSD_M gdouble &Log(int i) const { return Log_[i]; }
  PSD_M gdouble &Con(int i) const { return Con_[i]; }
  PSD_M gdouble &mn(int i) const { return mn_[i]; }
  PSD_M gdouble &mx(int i) const { return mx_[i]; }
  PSD_M gdouble &prv(int i) const { return prv_[i]; }
  PSD_M gint &di(int i) const { return di_[i]; }
  PSD_M GlobalList shifted(int d) const {
    GlobalList r;
    r.Lin_ = Lin_ + d;
    r.Log_ = Log_ + d;
    r.Con_ = Con_ + d;
    r.mn_ = mn_ + d;
    r.mx_ = mx_ + d;
    r.prv_ = prv_ + d;
    r.di_ = di_ + d;
    return r;
  }
  PSD_M GlobalList uniformed() const {
    GlobalList r;
    r.Lin_ = uniform_p(Lin_);
    r.Log_ = uniform_p(Log_);
    r.Con_ = uniform_p(Con_);
    r.mn_ = uniform_p(mn_);
    r.mx_ = uniform_p(mx_);
    r.prv_ = uniform_p(prv_);
    r.di_ = uniform_p(di_);
    return r;
  }
};
struct GlobalScratch {
  gdouble *lc_, *rc_, *om_, *mu_, *muc_, *oc2_;
  gint *cls_, *iv_;
  int iv_cap_;
  PSD_M gdouble &lc(int i) const { return lc_[i]; }
  PSD_M gdouble &rc(int i) const { return rc_[i]; }
  PSD_M gdouble &om(int i) const { return om_[i]; }
  PSD_M gdouble &mu(int i) const { return mu_[i]; }
  PSD_M gdouble &muc(int i) const { return muc_[i]; }
  PSD_M gdouble &oc2(int i) const { return oc2_[i]; }
  PSD_M gint &cls(int i) const { return cls_[i]; }
  PSD_M gint &iv(int i) const { return iv_[i]; }
  PSD_M int iv_cap() const { return iv_cap_; }
  /* Results of merged intervals classified by the helper wave (HOP_HBM_CLASSIFY), one slot per
   * interval (up to 2 cap of them): the six cost arrays are contiguous in pairs (lc|rc, om|mu,
   * muc|oc2, fpop_kernels.h global_scratch) and dead once the walk is over. */
  PSD_M gdouble &coop_x1(int k) const { return lc_[k]; }
  PSD_M gdouble &coop_x2(int k) const { return om_[k]; }
  PSD_M gdouble &coop_code(int k) const { return muc_[k]; }
  PSD_M GlobalScratch uniformed() const {
    GlobalScratch r;
    r.lc_ = uniform_p(lc_);
    r.rc_ = uniform_p(rc_);
    r.om_ = uniform_p(om_);
    r.mu_ = uniform_p(mu_);
    r.muc_ = uniform_p(muc_);
    r.oc2_ = uniform_p(oc2_);
    r.cls_ = uniform_p(cls_);
    r.iv_ = uniform_p(iv_);
    r.iv_cap_ = uniform_i(iv_cap_);
    return r;
  }
};

PSD_D LdsList lds_list(int id) {
  LdsList r;
  r.id = id;
  r.off = 0;
  return r;
}

template <class L>
PSD_D Coef load_coef(const L &f, int i) {
  Coef c;
  c.Linear = f.Lin(i);
  c.Log = f.Log(i);
  c.Constant = f.Con(i);
  return c;
}

template <class L>
PSD_D void store_piece(const L &f, int i, const Coef &c, double mn, double mx, int di,
                       double prv) {
  f.Lin(i) = c.Linear;
  f.Log(i) = c.Log;
  f.Con(i) = c.Constant;
  f.mn(i) = mn;
  f.mx(i) = mx;
  f.di(i) = di;
  f.prv(i) = prv;
}

enum { CLS_STORE = 0, CLS_CONST_EDGE = 1, CLS_CONST_MU = 2 };

/* how many pieces after (before) a constant's start the all-pairs speculation of min_less
 * (min_more) covers; the rest is scanned only if no crossing was found among them */
#ifndef PSD_SPEC_WINDOW
#define PSD_SPEC_WINDOW 5
#endif
constexpr int SPEC_WINDOW = PSD_SPEC_WINDOW;
/* ... and for functions of 13 to 16 pieces (whose starts times SPEC_WINDOW no longer fit the 64
 * lanes): one piece fewer per start keeps the direct lane -> (start, piece) mapping instead of a
 * loop over the starts, which cost the min-more wave of the large-penalty problems -- the
 * slowest of a grid -- 3 % of its data point (2124 -> 2101 ms on 200 k bins x 64,
 * profiles/r04/ab_adaptive_speculation_window.log).  Speculation only decides what is solved
 * ahead of the walk, never a result. */
#ifdef PSD_NO_NARROW_WINDOW /* A/B */
constexpr int SPEC_WINDOW_NARROW = SPEC_WINDOW;
#else
constexpr int SPEC_WINDOW_NARROW = 4;
#endif

/* error bits (the reference would throw / loop / read a sentinel) */
enum {
  WERR_OVERFLOW = 1,      /* output does not fit `cap` */
  WERR_REF_THROW = 2,     /* fpl:380 decreasing degenerate linear piece */
  WERR_SENTINEL = 4,      /* push_min_pieces neighbour outside the list */
  WERR_ZERO_INTERVAL = 8, /* fpl:933 zero-size merged interval */
  WERR_ARENA = 16,        /* the in-HBM store is full */
  WERR_HELPER = 32,       /* a helper wave did not answer (never expected) */
  WERR_SERIAL = 64,       /* specialised min_env only: the step needs the sequential replay */
};

/* A lane's own copy of piece `lane` and of what the first pass computed for it (functions of
 * at most 64 pieces): the walk's state machine then reads other pieces with v_readlane
 * instead of dependent LDS round trips. */
struct LanePiece {
  Coef c;
  double mn, mx;
  double lc, rc;   /* getCost at the left / right end */
  double om, mu;   /* argmin_mean(), argmin() */
  double muc, oc2; /* getCost(argmin()), PoissonLoss(argmin_mean()) */
  int cls;
};

/* Shared first half of min-less / min-more: per piece, the costs at both ends and the
 * optimum (fpl:245-246,310-311 / 469-470,483-485); kept in scratch for every piece and in
 * registers for piece `lane`. */
template <class L, class S, class M>
PSD_D void piece_costs_wave(const L &in, int n, const S &s, LanePiece &P, M &mth, int chunk0 = 0,
                            int stride = 1) {
  const int lane = lane_id();
  /* chunks chunk0, chunk0 + stride, ...: two waves share a long function (HOP_HBM_COSTS) */
  for (int base = chunk0 * WAVE; base < n; base += stride * WAVE) {
    int i = base + lane;
    if (i < n) {
      Coef c = load_coef(in, i);
      double mn = in.mn(i), mx = in.mx(i);
#ifndef PSD_NO_PAIRED_MATH
      /* exp(mn), exp(mx) and log(argmin_mean) do not depend on one another: one interleaved
       * evaluation (peakseg_detmath_core.h, psd_exp2_log) instead of three in a row; the values
       * are those of get_cost() and piece_opt() */
      const bool has_opt = c.Log != 0;
      PieceOpt o = {0.0, 0.0, 0.0, 0.0};
      if (has_opt) o.mean = argmin_mean(c);
      double e_mn, e_mx, l_om;
      mth.exp2_log(mn == -PSD_INF ? 0.0 : mn, mx == -PSD_INF ? 0.0 : mx, has_opt ? o.mean : 1.0,
                   e_mn, e_mx, l_om);
      double lc = get_cost_e(c, mn, e_mn);
      double rc = get_cost_e(c, mx, e_mx);
      if (has_opt) {
        o.log_mean = l_om;
        o.cost = mth.cost(c, o.log_mean);
        double loss_without_log_term = c.Linear * o.mean + c.Constant; /* fpl:52-61 */
        o.cost2 = loss_without_log_term + o.log_mean * c.Log;
      }
#else
      double lc = get_cost(c, mn);
      double rc = get_cost(c, mx);
      PieceOpt o = {0.0, 0.0, 0.0, 0.0};
      if (c.Log != 0) o = piece_opt(c);
#endif
      s.lc(i) = lc;
      s.rc(i) = rc;
      s.om(i) = o.mean;
      s.mu(i) = o.log_mean;
      s.muc(i) = o.cost;
      s.oc2(i) = o.cost2;
      if (base == 0) {
        P.c = c;
        P.mn = mn;
        P.mx = mx;
        P.lc = lc;
        P.rc = rc;
        P.om = o.mean;
        P.mu = o.log_mean;
        P.muc = o.cost;
        P.oc2 = o.cost2;
      }
    }
  }
  wave_sync();
}

/* First pass of min-less: per piece the end costs and optimum (piece_costs_wave) and what the
 * walk does with the piece when it reaches it in search mode. */
#ifdef PSD_HELPER_WAVES
/* The costs of a function in HBM by two waves: the helper takes the odd chunks. */
template <class L, class S>
PSD_D bool coop_piece_costs(const L &in, int n, const S &s, LanePiece &P, int chain, int p, int id) {
  Mail &m = g_sm.mail[chain];
  if (lane_id() == 0) {
    m.h_arg[0] = p;
    m.h_arg[1] = id;
    m.h_arg[2] = n;
  }
  mail_post(chain, HOP_HBM_COSTS);
  MathFull mth;
  piece_costs_wave(in, n, s, P, mth, 0, 2);
  return mail_wait(chain);
}
#endif
/* COOP: the function is list coop_id of spill slot coop_p and the chain's helper wave takes
 * half of the chunks (latency build, lists in HBM). */
template <bool COOP = false, class L, class S, class M>
PSD_D bool min_less_pre(const L &in, int n, const S &s, LanePiece &P, M &mth, int coop_chain = 0,
                        int coop_p = 0, int coop_id = 0) {
  const int lane = lane_id();
  bool ok = true;
#ifdef PSD_HELPER_WAVES
  if (COOP) {
    ok = coop_piece_costs(in, n, s, P, coop_chain, coop_p, coop_id);
  } else
#endif
  {
    piece_costs_wave(in, n, s, P, mth);
  }
  /* what the walk does with piece i when it reaches it in search mode */
  for (int base = 0; base < n; base += WAVE) {
    int i = base + lane;
    if (i < n) {
      double Log_i = in.Log(i);
      double lc = s.lc(i), rc = s.rc(i);
      bool has_next = i + 1 < n;
      double next_left_cost = has_next ? s.lc(i + 1) : PSD_INF;
      int cls;
      if (Log_i == 0) { /* fpl:256-308 */
        double right_left_diff = rc - lc;
        bool right_left_equal = right_left_diff < NEWTON_EPSILON;
        bool next_cost_more_than_left = true;
        if (has_next) {
          double next_left_diff = next_left_cost - lc;
          next_cost_more_than_left = NEWTON_EPSILON < next_left_diff;
        }
        cls = (next_cost_more_than_left && !right_left_equal) ? CLS_CONST_EDGE : CLS_STORE;
      } else { /* fpl:309-366 */
        double mu = s.mu(i), mu_cost = s.muc(i);
        bool next_ok = true;
        if (has_next) next_ok = NEWTON_EPSILON < next_left_cost - mu_cost;
        bool cost_ok = NEWTON_EPSILON < rc - mu_cost && next_ok;
        if (mu <= in.mn(i) && cost_ok) {
          cls = CLS_CONST_EDGE;
        } else if (mu < in.mx(i) && cost_ok) {
          cls = CLS_CONST_MU;
        } else {
          cls = CLS_STORE;
        }
      }
      s.cls(i) = cls;
      if (base == 0) P.cls = cls;
    }
  }
  wave_sync();
  return ok;
}

/* First pass of min-more, as min_less_pre. */
template <bool COOP = false, class L, class S, class M>
PSD_D bool min_more_pre(const L &in, int n, const S &s, LanePiece &P, M &mth, int coop_chain = 0,
                        int coop_p = 0, int coop_id = 0) {
  const int lane = lane_id();
  bool ok = true;
#ifdef PSD_HELPER_WAVES
  if (COOP) {
    ok = coop_piece_costs(in, n, s, P, coop_chain, coop_p, coop_id);
  } else
#endif
  {
    piece_costs_wave(in, n, s, P, mth);
  }
  for (int base = 0; base < n; base += WAVE) {
    int i = base + lane;
    if (i < n) {
      int cls;
      if (in.Log(i) == 0) { /* fpl:458-467 */
        cls = CLS_STORE;
      } else { /* fpl:468-548 */
        double mu = s.mu(i), mu_cost = s.muc(i);
        bool prev_ok = true;
        if (i > 0) {
          double prev_cost_right = s.rc(i - 1);
          prev_ok = NEWTON_EPSILON < prev_cost_right - mu_cost;
        }
        double this_cost_left = s.lc(i);
        if (in.mx(i) <= mu) {
          double this_cost_diff = this_cost_left - s.rc(i);
          cls = (NEWTON_EPSILON < this_cost_diff) ? CLS_CONST_EDGE : CLS_STORE;
        } else if (in.mn(i) < mu && NEWTON_EPSILON < this_cost_left - mu_cost && prev_ok) {
          cls = CLS_CONST_MU;
        } else {
          cls = CLS_STORE;
        }
      }
      s.cls(i) = cls;
      if (base == 0) P.cls = cls;
    }
  }
  wave_sync();
  return ok;
}

/* ------------------------------------------------------------------------------------- */
/* min-less: out(x) = min_{y<=x} in(y).  All output pieces get data_i = data_i_out (the
 * driver's set_prev_seg_end) and Constant += add_const (its add(0,0,penalty/cum_weight_prev),
 * PeakSegFPOPLog.cpp:290-296). */
template <bool SMALL, bool COOP = false, class L, class S, class M>
PSD_D int min_less_impl(L in_, int n_, L out_, int cap_, S s_, int data_i_out_,
                        double add_const_, M &mth, int coop_chain = 0, int coop_p = 0,
                        int coop_id = 0) {
  const L in = in_.uniformed(), out = out_.uniformed();
  const S s = s_.uniformed();
  const int n = uniform_i(n_), cap = uniform_i(cap_), data_i_out = uniform_i(data_i_out_);
  const double add_const = uniform_d(add_const_);
  const int lane = lane_id();
  /* lane i holds piece i; SMALL: the caller guarantees n <= WAVE (the other code drops out) */
  if (SMALL) PSD_ASSUME(n <= WAVE);
  const bool small = SMALL || n <= WAVE;
  LanePiece P;
  P.c.Linear = P.c.Log = P.c.Constant = 0.0;
  P.mn = P.mx = P.lc = P.rc = P.om = P.mu = P.muc = P.oc2 = 0.0;
  P.cls = CLS_STORE;
  PSD_PROF_T0();
  if (!min_less_pre<COOP>(in, n, s, P, mth, coop_chain, coop_p, coop_id)) return -WERR_HELPER;
  PSD_PROF_ADD(PROF_PRE);
  /* uniform reads of piece j: registers of lane j when the function fits one wave */
  auto cls_at = [&](int j) -> int { return small ? rdlane_i(P.cls, j) : s.cls(j); };
  auto mu_at = [&](int j) -> double { return small ? rdlane_d(P.mu, j) : s.mu(j); };
  auto muc_at = [&](int j) -> double { return small ? rdlane_d(P.muc, j) : s.muc(j); };
  auto lc_at = [&](int j) -> double { return small ? rdlane_d(P.lc, j) : s.lc(j); };
  auto mn_at = [&](int j) -> double { return small ? rdlane_d(P.mn, j) : in.mn(j); };
  auto mx_at = [&](int j) -> double { return small ? rdlane_d(P.mx, j) : in.mx(j); };

  int err = 0;
  /* ---- all-pairs speculation ------------------------------------------------------------
   * Where the constant started at piece j ends depends only on j (its level c_j is known from
   * the first pass) and on the pieces after it, not on how the walk got to j.  When all
   * (start j, later piece k) pairs fit in one wave, every pair tests its crossing NOW, in one
   * round of Newton solves, and the walk below only looks results up.  Otherwise each
   * constant scans its remaining pieces when the walk reaches it (one round per constant). */
  bool spec = false;
  unsigned long long sp_ev = 0, sp_inside = 0, sp_bad = 0;
  double sp_mu = PSD_INF;
  int my_base = 0; /* lane j: first task lane of start j */
  int win = SPEC_WINDOW; /* pieces after a start that the speculation covers */
  if (small) {
    unsigned long long m_start = ballot(lane < n && P.cls != CLS_STORE);
    int tj = -1, tk = 0;
    /* the window a start gets: SPEC_WINDOW pieces while n starts of that many fit a wave, one
     * fewer for up to 16 pieces (64 / 4 starts): still a direct lane -> task mapping, no loop
     * over the starts */
    win = (n * SPEC_WINDOW <= WAVE) ? SPEC_WINDOW : ((n * SPEC_WINDOW_NARROW <= WAVE) ? SPEC_WINDOW_NARROW : 0);
    if (win > 0) {
      /* few pieces (the usual case): task lane = start * window + offset, no loop */
      if (m_start) {
        spec = true;
        const int cj = win == SPEC_WINDOW ? lane / SPEC_WINDOW : lane / SPEC_WINDOW_NARROW;
        const int off = lane - cj * win;
        if (cj < n && ((m_start >> cj) & 1ull) && cj + 1 + off < n) {
          tj = cj;
          tk = cj + 1 + off;
        }
        my_base = lane * win;
      }
    } else {
      win = SPEC_WINDOW;
      int total = 0;
      for (unsigned long long m = m_start; m; m &= m - 1) {
        int c = n - 1 - ctz64(m);
        total += c < SPEC_WINDOW ? c : SPEC_WINDOW;
      }
      if (total > 0 && total <= WAVE) {
        spec = true;
        int base = 0;
        for (unsigned long long m = m_start; m; m &= m - 1) {
          int j = ctz64(m), cnt = n - 1 - j;
          if (cnt > SPEC_WINDOW) cnt = SPEC_WINDOW;
          if (lane == j) my_base = base;
          if (lane >= base && lane < base + cnt) {
            tj = j;
            tk = j + 1 + (lane - base);
          }
          base += cnt;
        }
      }
    }
    if (spec) {
      bool inside = false, at_right = false, bad = false;
      int sp_steps = 0;
      if (tj >= 0) {
        double level = (s.cls(tj) == CLS_CONST_MU) ? s.muc(tj) : s.lc(tj);
        Coef c = load_coef(in, tk);
        if (c.Log == 0) {
          if (c.Linear < 0) bad = true; /* fpl:378-380 */
        } else {
          PieceOpt o = {s.om(tk), s.mu(tk), s.muc(tk), s.oc2(tk)};
          if (has_two_roots(c, o, level)) {
            sp_mu = get_smaller_root(c, o, in.mn(tk), s.lc(tk), level, &sp_steps);
            inside = in.mn(tk) < sp_mu && sp_mu < in.mx(tk);
          }
          if (!inside) at_right = s.rc(tk) <= level + NEWTON_EPSILON;
        }
      }
      PSD_PROF_ITERS(PROF_IT_SPEC, sp_steps);
      sp_ev = ballot(inside || at_right);
      sp_inside = ballot(inside);
      sp_bad = ballot(bad);
    }
  }
  PSD_PROF_ADD(PROF_SERIAL); /* diagnostic builds: the speculation round */
  int n_out = 0;
  int i0 = 0;
  double prev_min_log_mean = mn_at(0);
  /* Functions of at most 64 pieces: the walk only RECORDS, in the lane of each input piece,
   * what that piece contributes (first its own kept or partial convex piece, then the
   * constant that starts at it); everything is written in one parallel pass after the
   * walk.  Longer functions write as they go. */
  bool e1 = false, e2 = false;          /* this lane's piece emits a convex / a constant piece */
  double e1_lo = 0.0, e1_hi = 0.0;      /* convex piece: own coefficients on [e1_lo, e1_hi] */
  double e2_lo = 0.0, e2_hi = 0.0, e2_level = 0.0, e2_best = 0.0;
  for (;;) {
    PSD_PROF_COUNT(PROF_IT_ROUNDS);
    /* ---- search mode: first piece j >= i0 that starts a constant ---- */
    int j = n;
    for (int base = i0 & ~(WAVE - 1); base < n; base += WAVE) {
      int i = base + lane;
      int cls_i = small ? P.cls : ((i < n) ? s.cls(i) : CLS_STORE);
      bool hit = i >= i0 && i < n && cls_i != CLS_STORE;
      unsigned long long m = ballot(hit);
      if (m) {
        j = base + ctz64(m);
        break;
      }
    }
    /* pieces i0..j-1 are kept as they are (fpl:303-307,361-364) */
    int cnt = j - i0;
    if (small) {
      if (lane >= i0 && lane < j) {
        e1 = true;
        e1_lo = (lane == i0) ? prev_min_log_mean : P.mn;
        e1_hi = P.mx;
      }
    } else {
      if (n_out + cnt + 2 > cap) return -WERR_OVERFLOW;
      for (int base = i0; base < j; base += WAVE) {
        int i = base + lane;
        if (i < j) {
          Coef c = load_coef(in, i);
          c.Constant = c.Constant + add_const;
          c.Linear = c.Linear + 0.0;
          c.Log = c.Log + 0.0;
          double lo = (i == i0) ? prev_min_log_mean : in.mn(i);
          store_piece(out, n_out + (i - i0), c, lo, in.mx(i), data_i_out, PSD_INF);
        }
      }
      n_out += cnt;
    }
    if (cnt > 0) prev_min_log_mean = mx_at(j - 1);
    if (j == n) break;
    /* ---- piece j starts a constant piece ---- */
    double prev_min_cost, prev_best_log_mean;
    if (cls_at(j) == CLS_CONST_MU) { /* fpl:337-355 */
      double mu = mu_at(j);
      if (prev_min_log_mean < mu) {
        if (small) {
          if (lane == j) {
            e1 = true;
            e1_lo = prev_min_log_mean;
            e1_hi = mu;
          }
        } else {
          if (lane == 0) {
            Coef c = load_coef(in, j);
            c.Constant = c.Constant + add_const;
            c.Linear = c.Linear + 0.0;
            c.Log = c.Log + 0.0;
            store_piece(out, n_out, c, prev_min_log_mean, mu, data_i_out, PSD_INF);
          }
          n_out++;
        }
      }
      prev_min_log_mean = mu;
      prev_best_log_mean = mu;
      prev_min_cost = muc_at(j);
    } else { /* fpl:288-292,328-336 */
      prev_min_cost = lc_at(j);
      prev_best_log_mean = mn_at(j);
    }
    /* ---- constant mode: first piece k > j where the constant ends (fpl:367-422) ---- */
    int k_ev = -1;
    bool ev_inside = false;
    double ev_mu = 0.0;
    int scan_from = j + 1; /* first piece not covered by the speculation */
    if (spec) {
      int cntj = n - 1 - j;
      if (cntj > win) cntj = win;
      scan_from = j + 1 + cntj;
      if (cntj > 0) {
        int base = rdlane_i(my_base, j);
        unsigned long long range = ((1ull << cntj) - 1ull) << base;
        unsigned long long ev = sp_ev & range;
        unsigned long long visited = ev ? (range & lanes_below(ctz64(ev))) : range;
        if (sp_bad & visited) err |= WERR_REF_THROW;
        if (ev) {
          int src = ctz64(ev);
          k_ev = j + 1 + (src - base);
          ev_inside = ((sp_inside >> src) & 1ull) != 0;
          ev_mu = rdlane_d(sp_mu, src);
        }
      }
    }
    if (k_ev < 0) { /* pieces beyond the speculation window: scan them now */
      for (int base = scan_from; base < n; base += WAVE) {
        int k = base + lane;
        bool inside = false, at_right = false, bad = false;
        double mu = PSD_INF;
        if (k < n) {
          Coef c = load_coef(in, k);
          if (c.Log == 0) {
            if (c.Linear < 0) bad = true; /* fpl:378-380 */
          } else {
            /* optimum and end costs of piece k were computed in the first pass */
            PieceOpt o = {s.om(k), s.mu(k), s.muc(k), s.oc2(k)};
            if (has_two_roots(c, o, prev_min_cost)) {
              mu = get_smaller_root(c, o, in.mn(k), s.lc(k), prev_min_cost);
              inside = in.mn(k) < mu && mu < in.mx(k);
            }
            if (!inside) at_right = s.rc(k) <= prev_min_cost + NEWTON_EPSILON;
          }
        }
        unsigned long long m_ev = ballot(inside || at_right);
        unsigned long long m_in = ballot(inside);
        unsigned long long m_bad = ballot(bad);
        unsigned long long visited = m_ev ? lanes_below(ctz64(m_ev)) : ~0ull;
        if (m_bad & visited) err |= WERR_REF_THROW;
        if (m_ev) {
          int src = ctz64(m_ev);
          k_ev = base + src;
          ev_inside = ((m_in >> src) & 1ull) != 0;
          ev_mu = rdlane_d(mu, src);
          break;
        }
      }
    }
    Coef cc;
    cc.Linear = 0.0 + 0.0;
    cc.Log = 0.0 + 0.0;
    cc.Constant = prev_min_cost + add_const;
    /* where the constant ends: the end of the function (fpl:429-436), a crossing inside piece
     * k, which is then revisited in search mode (fpl:397-408), or the right end of piece k
     * (fpl:410-420) */
    double c_hi;
    bool last_round = false;
    if (k_ev < 0) {
      c_hi = mx_at(n - 1);
      last_round = true;
    } else if (ev_inside) {
      c_hi = ev_mu;
      i0 = k_ev;
    } else {
      c_hi = mx_at(k_ev);
      i0 = k_ev + 1;
      if (i0 == n) last_round = true;
    }
    if (small) {
      if (lane == j) {
        e2 = true;
        e2_lo = prev_min_log_mean;
        e2_hi = c_hi;
        e2_level = cc.Constant;
        e2_best = prev_best_log_mean;
      }
    } else {
      if (lane == 0)
        store_piece(out, n_out, cc, prev_min_log_mean, c_hi, data_i_out, prev_best_log_mean);
      n_out++;
    }
    prev_min_log_mean = c_hi;
    if (last_round) break;
  }
  if (small) {
    /* one parallel pass: lane i writes its convex piece, then its constant piece */
    unsigned long long m1 = ballot(e1), m2 = ballot(e2);
    unsigned long long lb = lanes_below(lane);
    n_out = popc64(m1) + popc64(m2);
    if (n_out + 2 > cap) return -WERR_OVERFLOW;
    int pos = popc64(m1 & lb) + popc64(m2 & lb);
    if (e1) {
      Coef c = P.c;
      c.Constant = c.Constant + add_const;
      c.Linear = c.Linear + 0.0;
      c.Log = c.Log + 0.0;
      store_piece(out, pos, c, e1_lo, e1_hi, data_i_out, PSD_INF);
      pos++;
    }
    if (e2) {
      Coef cc;
      cc.Linear = 0.0 + 0.0;
      cc.Log = 0.0 + 0.0;
      cc.Constant = e2_level;
      store_piece(out, pos, cc, e2_lo, e2_hi, data_i_out, e2_best);
    }
  }
  wave_sync();
  PSD_PROF_ADD(PROF_WALK);
  return err ? -err : n_out;
}

/* ------------------------------------------------------------------------------------- */
/* min-more: out(x) = min_{y>=x} in(y).  The reference builds the list with emplace_front;
 * here pieces are written downwards from out[cap-1]: the result is out[cap-n .. cap). */
template <bool SMALL, bool COOP = false, class L, class S, class M>
PSD_D int min_more_impl(L in_, int n_, L out_, int cap_, S s_, int data_i_out_, M &mth,
                        int coop_chain = 0, int coop_p = 0, int coop_id = 0) {
  const L in = in_.uniformed(), out = out_.uniformed();
  const S s = s_.uniformed();
  const int n = uniform_i(n_), cap = uniform_i(cap_), data_i_out = uniform_i(data_i_out_);
  const int lane = lane_id();
  if (SMALL) PSD_ASSUME(n <= WAVE);
  const bool small = SMALL || n <= WAVE;
  LanePiece P;
  P.c.Linear = P.c.Log = P.c.Constant = 0.0;
  P.mn = P.mx = P.lc = P.rc = P.om = P.mu = P.muc = P.oc2 = 0.0;
  P.cls = CLS_STORE;
  PSD_PROF_T0();
  if (!min_more_pre<COOP>(in, n, s, P, mth, coop_chain, coop_p, coop_id)) return -WERR_HELPER;
  PSD_PROF_ADD(PROF_PRE);
  auto cls_at = [&](int j) -> int { return small ? rdlane_i(P.cls, j) : s.cls(j); };
  auto mu_at = [&](int j) -> double { return small ? rdlane_d(P.mu, j) : s.mu(j); };
  auto muc_at = [&](int j) -> double { return small ? rdlane_d(P.muc, j) : s.muc(j); };
  auto rc_at = [&](int j) -> double { return small ? rdlane_d(P.rc, j) : s.rc(j); };
  auto mn_at = [&](int j) -> double { return small ? rdlane_d(P.mn, j) : in.mn(j); };
  auto mx_at = [&](int j) -> double { return small ? rdlane_d(P.mx, j) : in.mx(j); };

  /* all-pairs speculation, mirror image of min_less_wave: pairs (start j, earlier piece k),
   * tasks of one start ordered by decreasing k */
  bool spec = false;
  unsigned long long sp_ev = 0, sp_inside = 0;
  double sp_mu = PSD_INF;
  int my_base = 0;
  int win = SPEC_WINDOW;
  PSD_PROF_SUB0();
  if (small) {
    unsigned long long m_start = ballot(lane < n && P.cls != CLS_STORE);
    int tj = -1, tk = 0;
    win = (n * SPEC_WINDOW <= WAVE) ? SPEC_WINDOW : ((n * SPEC_WINDOW_NARROW <= WAVE) ? SPEC_WINDOW_NARROW : 0);
    if (win > 0) {
      /* few pieces (the usual case): task lane = start * window + offset, no loop */
      if (m_start & ~1ull) { /* piece 0 has no earlier piece */
        spec = true;
        const int cj = win == SPEC_WINDOW ? lane / SPEC_WINDOW : lane / SPEC_WINDOW_NARROW;
        const int off = lane - cj * win;
        if (cj < n && ((m_start >> cj) & 1ull) && off < cj) {
          tj = cj;
          tk = cj - 1 - off;
        }
        my_base = lane * win;
      }
    } else {
      win = SPEC_WINDOW;
      int total = 0;
      for (unsigned long long m = m_start; m; m &= m - 1) {
        int c = ctz64(m);
        total += c < SPEC_WINDOW ? c : SPEC_WINDOW;
      }
      if (total > 0 && total <= WAVE) {
        spec = true;
        int base = 0;
        for (unsigned long long m = m_start; m; m &= m - 1) {
          int j = ctz64(m), cnt = j;
          if (cnt > SPEC_WINDOW) cnt = SPEC_WINDOW;
          if (lane == j) my_base = base;
          if (lane >= base && lane < base + cnt) {
            tj = j;
            tk = j - 1 - (lane - base);
          }
          base += cnt;
        }
      }
    }
    if (spec) {
      bool inside = false, at_left = false;
      int sp_steps = 0;
      PSD_PROF_SUB(PROF_S_ASSIGN);
      double level = 0.0, t_mx = 0.0, t_rc = 0.0, t_mn = 0.0, t_lc = 0.0;
      Coef c = {0.0, 0.0, 0.0};
      PieceOpt o = {0.0, 0.0, 0.0, 0.0};
      if (tj >= 0) {
        level = (s.cls(tj) == CLS_CONST_MU) ? s.muc(tj) : s.rc(tj);
        c = load_coef(in, tk);
        o.mean = s.om(tk);
        o.log_mean = s.mu(tk);
        o.cost = s.muc(tk);
        o.cost2 = s.oc2(tk);
        t_mx = in.mx(tk);
        t_rc = s.rc(tk);
        t_mn = in.mn(tk);
        t_lc = s.lc(tk);
      }
      PSD_PROF_SUB(PROF_S_LOAD);
      if (tj >= 0) {
        if (c.Log == 0) {
          sp_mu = mth.log_wild(psd_div(level - c.Constant, c.Linear)); /* fpl:563 */
        } else {
          if (has_two_roots(c, o, level)) {
            sp_mu = get_larger_root(c, o, t_mx, t_rc, level, &sp_steps, mth.rare_out());
          }
        }
        inside = t_mn < sp_mu && sp_mu < t_mx;
        if (!inside) at_left = t_lc <= level + NEWTON_EPSILON;
      }
      PSD_PROF_SUB(PROF_S_NEWTON);
      PSD_PROF_ITERS(PROF_IT_SPEC, sp_steps);
      sp_ev = ballot(inside || at_left);
      sp_inside = ballot(inside);
    }
  }
  PSD_PROF_ADD(PROF_SERIAL); /* diagnostic builds: the speculation round */
  int n_out = 0; /* pieces written so far; piece p lives at out[cap-1-p] */
  int i0 = n - 1;
  double prev_max_log_mean = mx_at(n - 1);
  /* deferred emission for functions of at most 64 pieces, as in min_less_wave; in ascending
   * order a piece contributes first the constant that starts at it (it extends downwards),
   * then its own kept or partial convex piece */
  bool e1 = false, e2 = false;
  double e1_lo = 0.0, e1_hi = 0.0;
  double e2_lo = 0.0, e2_hi = 0.0, e2_level = 0.0, e2_best = 0.0;
  for (;;) {
    PSD_PROF_COUNT(PROF_IT_ROUNDS);
    /* ---- search mode, walking down from i0: first piece j <= i0 starting a constant ---- */
    int j = -1;
    if (small) {
      unsigned long long m = ballot(lane <= i0 && P.cls != CLS_STORE);
      if (m) j = msb64(m);
    } else {
      for (int base = i0 | (WAVE - 1); base >= 0; base -= WAVE) { /* base = top of a chunk */
        int i = base - lane;
        bool hit = i <= i0 && i >= 0 && s.cls(i) != CLS_STORE;
        unsigned long long m = ballot(hit);
        if (m) {
          j = base - ctz64(m);
          break;
        }
      }
    }
    int cnt = i0 - j;
    if (small) {
      if (lane > j && lane <= i0) {
        e1 = true;
        e1_lo = P.mn;
        e1_hi = (lane == i0) ? prev_max_log_mean : P.mx;
      }
    } else {
      if (n_out + cnt + 2 > cap) return -WERR_OVERFLOW;
      for (int base = i0; base > j; base -= WAVE) {
        int i = base - lane;
        if (i > j) {
          Coef c = load_coef(in, i);
          double hi = (i == i0) ? prev_max_log_mean : in.mx(i);
          store_piece(out, cap - 1 - (n_out + (i0 - i)), c, in.mn(i), hi, data_i_out, PSD_INF);
        }
      }
      n_out += cnt;
    }
    if (cnt > 0) prev_max_log_mean = mn_at(j + 1);
    if (j < 0) break;
    double prev_min_cost, prev_best_log_mean;
    if (cls_at(j) == CLS_CONST_MU) { /* fpl:524-537 */
      double mu = mu_at(j);
      if (mu < prev_max_log_mean) {
        if (small) {
          if (lane == j) {
            e1 = true;
            e1_lo = mu;
            e1_hi = prev_max_log_mean;
          }
        } else {
          if (lane == 0)
            store_piece(out, cap - 1 - n_out, load_coef(in, j), mu, prev_max_log_mean,
                        data_i_out, PSD_INF);
          n_out++;
        }
      }
      prev_max_log_mean = mu;
      prev_best_log_mean = mu;
      prev_min_cost = muc_at(j);
    } else { /* fpl:500-510 */
      prev_min_cost = rc_at(j);
      prev_best_log_mean = mx_at(j);
    }
    /* ---- constant mode: highest piece k < j where the constant ends (fpl:549-602) ---- */
    int k_ev = -1;
    bool ev_inside = false;
    double ev_mu = 0.0;
    int scan_from = j - 1; /* first piece (walking down) not covered by the speculation */
    if (spec) {
      int cntj = j < win ? j : win;
      scan_from = j - 1 - cntj;
      if (cntj > 0) {
        int base = rdlane_i(my_base, j);
        unsigned long long range = ((1ull << cntj) - 1ull) << base;
        unsigned long long ev = sp_ev & range;
        if (ev) {
          int src = ctz64(ev);
          k_ev = j - 1 - (src - base);
          ev_inside = ((sp_inside >> src) & 1ull) != 0;
          ev_mu = rdlane_d(sp_mu, src);
        }
      }
    }
    if (k_ev < 0) { /* pieces beyond the speculation window: scan them now */
      for (int base = scan_from; base >= 0; base -= WAVE) {
        int k = base - lane;
        bool inside = false, at_left = false;
        double mu = PSD_INF;
        if (k >= 0) {
          Coef c = load_coef(in, k);
          if (c.Log == 0) {
            mu = d_log(psd_div(prev_min_cost - c.Constant, c.Linear)); /* fpl:563 */
          } else {
            PieceOpt o = {s.om(k), s.mu(k), s.muc(k), s.oc2(k)};
            if (has_two_roots(c, o, prev_min_cost)) {
              mu = get_larger_root(c, o, in.mx(k), s.rc(k), prev_min_cost);
            }
          }
          inside = in.mn(k) < mu && mu < in.mx(k);
          if (!inside) at_left = s.lc(k) <= prev_min_cost + NEWTON_EPSILON;
        }
        unsigned long long m_ev = ballot(inside || at_left);
        unsigned long long m_in = ballot(inside);
        if (m_ev) {
          int src = ctz64(m_ev);
          k_ev = base - src;
          ev_inside = ((m_in >> src) & 1ull) != 0;
          ev_mu = rdlane_d(mu, src);
          break;
        }
      }
    }
    Coef cc;
    cc.Linear = 0.0;
    cc.Log = 0.0;
    cc.Constant = prev_min_cost;
    /* where the constant ends (walking down): the start of the function (fpl:608-615), a
     * crossing inside piece k, then revisited (fpl:578-590), or the left end of piece k
     * (fpl:591-601) */
    double c_lo;
    bool last_round = false;
    if (k_ev < 0) {
      c_lo = mn_at(0);
      last_round = true;
    } else if (ev_inside) {
      c_lo = ev_mu;
      i0 = k_ev;
    } else {
      c_lo = mn_at(k_ev);
      i0 = k_ev - 1;
      if (i0 < 0) last_round = true;
    }
    if (small) {
      if (lane == j) {
        e2 = true;
        e2_lo = c_lo;
        e2_hi = prev_max_log_mean;
        e2_level = prev_min_cost;
        e2_best = prev_best_log_mean;
      }
    } else {
      if (lane == 0)
        store_piece(out, cap - 1 - n_out, cc, c_lo, prev_max_log_mean, data_i_out,
                    prev_best_log_mean);
      n_out++;
    }
    prev_max_log_mean = c_lo;
    if (last_round) break;
  }
  if (small) {
    /* one parallel pass; the result occupies out[cap-n_out .. cap) in ascending order */
    unsigned long long m1 = ballot(e1), m2 = ballot(e2);
    unsigned long long lb = lanes_below(lane);
    n_out = popc64(m1) + popc64(m2);
    if (n_out + 2 > cap) return -WERR_OVERFLOW;
    int pos = cap - n_out + popc64(m1 & lb) + popc64(m2 & lb);
    if (e2) {
      Coef cc;
      cc.Linear = 0.0;
      cc.Log = 0.0;
      cc.Constant = e2_level;
      store_piece(out, pos, cc, e2_lo, e2_hi, data_i_out, e2_best);
      pos++;
    }
    if (e1) store_piece(out, pos, P.c, e1_lo, e1_hi, data_i_out, PSD_INF);
  }
  wave_sync();
  PSD_PROF_ADD(PROF_WALK);
  return n_out;
}

/* Candidates emitted for one merged interval [a,b].  Every path of push_min_pieces emits
 * one of three shapes, with the source alternating between the two input pieces:
 *   n=1: [a,b]            n=2: [a,x1] [x1,b]            n=3: [a,x1] [x1,x2] [x2,b]
 * `first` is the source of the first piece (0: piece of fun1, 1: piece of fun2).  All split
 * points are strictly inside (a,b) and ordered (the reference tests that before pushing), so
 * push_piece's zero-width guard (fpl:1261-1267) can only ever drop the n=1 shape; it is
 * applied there.  Plain scalars: nothing here is indexed at run time (an indexed struct was
 * placed in scratch memory by the compiler). */
struct Cands {
  int n, first;
  double x1, x2;
};
PSD_D void cand_one(Cands &c, int src, double a, double b) {
  c.n = (b <= a) ? 0 : 1;
  c.first = src;
}
PSD_D void cand_two(Cands &c, int first, double x) {
  c.n = 2;
  c.first = first;
  c.x1 = x;
}
PSD_D void cand_three(Cands &c, int first, double x1, double x2) {
  c.n = 3;
  c.first = first;
  c.x1 = x1;
  c.x2 = x2;
}

/* push_min_pieces (fpl:870-1259) for one merged interval [a, b] =
 * [last_min_log_mean, first_max_log_mean] of it1 = c1, it2 = c2.
 * exp(a), exp(b), the optimum of the difference piece and its end costs are each needed by
 * several of the reference's helper calls (getCost / has_two_roots / get_*_root / argmin on
 * the same diff_piece); they are evaluated once here -- identical values, fewer serial
 * transcendentals. */
PSD_D void env_interval(const Coef &c1, const Coef &c2, double a, double b, bool same_at_left,
                        bool same_at_right, Cands &out) {
  if (same_funs(c1, c2)) { /* fpl:945-951 */
    cand_one(out, 0, a, b);
    return;
  }
  Coef d;
  d.Linear = c1.Linear - c2.Linear;
  d.Log = c1.Log - c2.Log;
  d.Constant = c1.Constant - c2.Constant;
  const double ea = d_exp(a), eb = d_exp(b);
  double mid_mean = (eb + ea) / 2; /* fpl:960 */
  double cost_diff_mid = get_cost(d, d_log(mid_mean));
  if (same_at_left && same_at_right) { /* fpl:963-971 */
    cand_one(out, cost_diff_mid < 0 ? 0 : 1, a, b);
    return;
  }
  if (d.Log == 0) { /* fpl:973-1019 */
    if (d.Linear == 0) {
      cand_one(out, d.Constant < 0 ? 0 : 1, a, b);
      return;
    }
    if (d.Constant == 0) {
      cand_one(out, d.Linear < 0 ? 0 : 1, a, b);
      return;
    }
    double x = d_log(psd_div(-d.Constant, d.Linear));
    if (a < x && x < b) {
      int first = (0 < d.Linear) ? 0 : 1;
      cand_two(out, first, x);
      return;
    }
    cand_one(out, cost_diff_mid < 0 ? 0 : 1, a, b);
    return;
  }
  double cost_diff_left = get_cost_e(d, a, ea);
  double cost_diff_right = get_cost_e(d, b, eb);
  const PieceOpt o = piece_opt(d);
  bool two_roots = has_two_roots(d, o, 0.0);
  double smaller_log_mean = PSD_INF, larger_log_mean = PSD_INF;
  if (two_roots) {
    smaller_log_mean = get_smaller_root(d, o, a, cost_diff_left, 0.0);
    larger_log_mean = get_larger_root(d, o, b, cost_diff_right, 0.0);
  }
  if (same_at_right) { /* fpl:1029-1093 */
    if (two_roots) {
      double x = smaller_log_mean;
      double opt = o.log_mean; /* diff_piece.argmin() */
      if (a < x && x < opt && opt < b) {
        int first = (cost_diff_left < 0) ? 0 : 1;
        cand_two(out, first, x);
        return;
      }
      bool it1_smaller_at_mean0 = 0 < d.Log;
      if (x < a) {
        cand_one(out, it1_smaller_at_mean0 ? 1 : 0, a, b);
      } else {
        cand_one(out, it1_smaller_at_mean0 ? 0 : 1, a, b);
      }
      return;
    }
    cand_one(out, cost_diff_mid < 0 ? 0 : 1, a, b);
    return;
  }
  if (same_at_left) { /* fpl:1094-1123 */
    if (two_roots) {
      double x = larger_log_mean;
      double opt = o.log_mean;
      if (a < opt && opt < x && x < b) {
        int first = (cost_diff_right < 0) ? 1 : 0;
        cand_two(out, first, x);
        return;
      }
    }
    cand_one(out, cost_diff_mid < 0 ? 0 : 1, a, b);
    return;
  }
  /* equal on neither side (fpl:1124-1258) */
  double first_log_mean = PSD_INF, second_log_mean = PSD_INF;
  double e_smaller = 0.0;
  if (two_roots) {
    bool larger_inside = a < larger_log_mean && larger_log_mean < b;
    e_smaller = d_exp(smaller_log_mean);
    bool smaller_inside = a < smaller_log_mean && 0 < e_smaller && smaller_log_mean < b;
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
  if (first_log_mean == PSD_INF) { /* no crossing inside (fpl:1238-1258) */
    double cost_diff;
    if (absd(cost_diff_mid) < NEWTON_EPSILON) {
      cost_diff = cost_diff_right;
    } else {
      cost_diff = cost_diff_mid;
    }
    cand_one(out, cost_diff < 0 ? 0 : 1, a, b);
    return;
  }
  /* one or two crossings: both cases may test the sign of the difference at the mean-space
   * midpoint of [a, first crossing] (fpl:1178-1179,1211-1212) */
  const bool two = second_log_mean != PSD_INF;
  const bool need_before = !two || (second_log_mean - first_log_mean < first_log_mean - a);
  double cost_diff_before = 0.0;
  if (need_before) {
    double e_first = (first_log_mean == smaller_log_mean) ? e_smaller : d_exp(first_log_mean);
    double before_mean = (ea + e_first) / 2;
    cost_diff_before = get_cost(d, d_log(before_mean));
  }
  if (two) {
    bool it1_larger_before;
    if (need_before) {
      it1_larger_before = cost_diff_before < 0;
    } else {
      double log_mean_between = (first_log_mean + second_log_mean) / 2;
      double cost_diff_between = get_cost(d, log_mean_between);
      it1_larger_before = !(cost_diff_between < 0);
    }
    int first = it1_larger_before ? 0 : 1;
    cand_three(out, first, first_log_mean, second_log_mean);
  } else {
    double after_mean = (b + first_log_mean) / 2; /* a log-mean, fpl:1216 */
    double cost_diff_after = get_cost(d, after_mean);
    if (cost_diff_before < 0) {
      if (cost_diff_after < 0) {
        cand_one(out, 0, a, b);
      } else {
        cand_two(out, 0, first_log_mean);
      }
    } else {
      if (cost_diff_after < 0) {
        cand_two(out, 1, first_log_mean);
      } else {
        cand_one(out, 1, a, b);
      }
    }
  }
}

/* number of pieces of f (sorted by max_log_mean) with max_log_mean < x */
template <class L>
PSD_D int rank_mx(const L &f, int n, double x) {
  if (n <= 32) { /* independent broadcast reads beat a dependent binary search */
    int r = 0;
    for (int j = 0; j < n; j++) r += f.mx(j) < x ? 1 : 0;
    return r;
  }
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (f.mx(mid) < x) {
      lo = mid + 1;
    } else {
      hi = mid;
    }
  }
  return lo;
}

/* Everything push_min_pieces needs for merged interval (i1,i2): loads the two pieces and
 * the neighbours it inspects (fpl:876-932), classifies, returns candidates. */
template <class L>
PSD_D void env_interval_at(const L &f1, int n1, const L &f2, int n2, int i1, int i2, Cands &cands,
                           double &a_out, double &b_out, Coef &c1, Coef &c2, int &err) {
  c1 = load_coef(f1, i1);
  c2 = load_coef(f2, i2);
  double mn1 = f1.mn(i1), mx1 = f1.mx(i1), mn2 = f2.mn(i2), mx2 = f2.mx(i2);
  bool same_at_left, same_at_right;
  double last_min_log_mean, first_max_log_mean;
  bool sentinel = false;
  if (mn1 < mn2) {
    if (i2 == 0) sentinel = true;
    same_at_left = !sentinel && same_funs(load_coef(f2, i2 - 1), c1);
    last_min_log_mean = mn2;
  } else {
    last_min_log_mean = mn1;
    if (mn2 < mn1) {
      if (i1 == 0) sentinel = true;
      same_at_left = !sentinel && same_funs(load_coef(f1, i1 - 1), c2);
    } else {
      if (i1 == 0 && i2 == 0) {
        same_at_left = false;
      } else {
        if (i1 == 0 || i2 == 0) sentinel = true;
        same_at_left = !sentinel && same_funs(load_coef(f1, i1 - 1), load_coef(f2, i2 - 1));
      }
    }
  }
  if (mx1 < mx2) {
    if (i1 + 1 >= n1) sentinel = true;
    same_at_right = !sentinel && same_funs(load_coef(f1, i1 + 1), c2);
    first_max_log_mean = mx1;
  } else {
    first_max_log_mean = mx2;
    if (mx2 < mx1) {
      if (i2 + 1 >= n2) sentinel = true;
      same_at_right = !sentinel && same_funs(c1, load_coef(f2, i2 + 1));
    } else {
      if (i1 + 1 == n1 && i2 + 1 == n2) {
        same_at_right = false;
      } else {
        if (i1 + 1 >= n1 || i2 + 1 >= n2) sentinel = true;
        same_at_right = !sentinel && same_funs(load_coef(f1, i1 + 1), load_coef(f2, i2 + 1));
      }
    }
  }
  cands.n = 0;
  cands.first = 0;
  cands.x1 = cands.x2 = 0.0;
  a_out = last_min_log_mean;
  b_out = first_max_log_mean;
  if (sentinel) {
    err |= WERR_SENTINEL;
    return;
  }
  if (last_min_log_mean == first_max_log_mean) { /* fpl:933-944 */
    err |= WERR_ZERO_INTERVAL;
    return;
  }
  env_interval(c1, c2, last_min_log_mean, first_max_log_mean, same_at_left, same_at_right,
               cands);
}

/* The same classification as env_interval(), written for SIMT execution: one lane per merged
 * interval, and every transcendental evaluation site is reached by all lanes that need it at
 * the same time (predicated phases) instead of each lane walking its own branch of
 * push_min_pieces -- a wave otherwise executes the union of all branches one after another.
 * The arithmetic per lane is identical to env_interval(). */
template <bool HELP, class M>
PSD_D void env_classify_lanes(bool valid, const Coef &c1, const Coef &c2, double a, double b,
                              bool same_at_left, bool same_at_right, Cands &out, int chain,
                              int &err, M &mth) {
  out.n = 0;
  out.first = 0;
  out.x1 = out.x2 = 0.0;
  Coef d;
  d.Linear = c1.Linear - c2.Linear;
  d.Log = c1.Log - c2.Log;
  d.Constant = c1.Constant - c2.Constant;
  const bool triv = same_funs(c1, c2);  /* fpl:945-951 */
  const bool act = valid && !triv;
  PSD_PROF_T0();
  const bool both = same_at_left && same_at_right;
  const bool hard = act && !both;
  const bool degen = hard && d.Log == 0;                              /* fpl:973-1019 */
  const bool degen_root = degen && d.Linear != 0 && d.Constant != 0;  /* fpl:996 */
  const bool rootp = hard && d.Log != 0;
  bool root_posted = false;
#ifdef PSD_HELPER_WAVES
  if (HELP) {
    /* The helper wave starts on the larger roots (fpl:1027) right away: it derives the
     * optimum of the difference piece and has_two_roots itself -- same code, same bits --
     * while this wave evaluates the end costs, the midpoint and the smaller roots. */
    if (ballot(rootp)) {
      Mail &m = g_sm.mail[chain];
      const int l = lane_id();
      m.flags[l] = rootp ? 1 : 0;
      m.d_lin[l] = d.Linear;
      m.d_log[l] = d.Log;
      m.d_con[l] = d.Constant;
      m.b[l] = b;
      mail_post(chain, HOP_ROOT);
      root_posted = true;
    }
  }
#endif
  /* phases A-D: exp(a), exp(b); the cost at the mean-space midpoint (fpl:960-961); the one log
   * site for the degenerate crossing and for argmin() of the difference; the optimum of the
   * difference piece, its end costs, has_two_roots (fpl:1020-1022).  Six transcendentals per
   * interval, evaluated in three interleaved pairs (the values are those of the single calls):
   * exp(a) | exp(b), log(midpoint) | log(argmin_mean), exp for the cost at each of the two. */
  double ea = 0.0, eb = 0.0, cost_diff_mid = 0.0;
  /* (psd_div: against a constant piece the difference has the function piece's own Linear,
   * which may be 1 - k ulp; peakseg_detmath.h) */
  const double larg = degen ? psd_div(-d.Constant, d.Linear) : psd_div(-d.Log, d.Linear);
  const bool need_l = degen_root || rootp;
  double lres = 0.0;
  PieceOpt o = {0.0, 0.0, 0.0, 0.0};
  double cost_diff_left = 0.0, cost_diff_right = 0.0;
  bool two_roots = false;
#ifndef PSD_NO_PAIRED_MATH
  if (ballot(act)) {
    mth.exp2((act && a != -PSD_INF) ? a : 0.0, act ? b : 0.0, ea, eb);
    if (a == -PSD_INF) ea = 0.0; /* exp(-Inf) */
    double log_mid;
    mth.log2(act ? (eb + ea) / 2 : 1.0, need_l ? larg : 1.0, log_mid, lres);
    if (!need_l) lres = 0.0;
    double e_mid, e_opt;
    mth.exp2(log_mid == -PSD_INF ? 0.0 : log_mid, (rootp && lres != -PSD_INF) ? lres : 0.0, e_mid,
           e_opt);
    if (act) cost_diff_mid = get_cost_e(d, log_mid, e_mid);
    if (!act) ea = eb = 0.0;
    if (rootp) o.cost = get_cost_e(d, lres, e_opt);
  }
  PSD_PROF_ADD(PROF_C_MID);
#else
  if (act) {
    ea = mth.exp(a);
    eb = mth.exp(b);
    cost_diff_mid = mth.cost(d, mth.log((eb + ea) / 2));
  }
  PSD_PROF_ADD(PROF_C_MID);
  if (need_l) lres = mth.log(larg);
  if (rootp) o.cost = mth.cost(d, lres);
#endif
  if (rootp) {
    cost_diff_left = get_cost_e(d, a, ea);
    cost_diff_right = get_cost_e(d, b, eb);
    o.mean = larg;
    o.log_mean = lres;
    double loss_without_log_term = d.Linear * o.mean + d.Constant;
    o.cost2 = loss_without_log_term + o.log_mean * d.Log;
    two_roots = has_two_roots(d, o, 0.0);
  }
  PSD_PROF_ADD(PROF_C_OPT);
  /* phases E, F: the two Newton solves (fpl:1023-1028) */
  double smaller_log_mean = PSD_INF, larger_log_mean = PSD_INF;
  int it_small = 0, it_large = 0;
  if (two_roots) smaller_log_mean = get_smaller_root(d, o, a, cost_diff_left, 0.0, &it_small);
  PSD_PROF_ADD(PROF_C_SMALL);
  /* Phase G needs more evaluations for intervals equal on neither side (fpl:1124-1258).  What
   * depends on the smaller root alone is evaluated here, while the helper may still be at the
   * larger roots: exp(smaller root), and the cost before the first crossing as if the smaller
   * root were it (the usual case; replaced below when the larger root comes first). */
  const bool neither = rootp && !same_at_left && !same_at_right;
  double e_smaller = 0.0, cost_before_smaller = 0.0;
#ifndef PSD_NO_EARLY_TAIL
  constexpr bool EARLY_TAIL = HELP; /* without a helper there is no wait to fill */
#else
  constexpr bool EARLY_TAIL = false;
#endif
  if (EARLY_TAIL && ballot(neither && two_roots)) {
    const bool on = neither && two_roots;
    e_smaller = mth.exp(on ? smaller_log_mean : 0.0);
    cost_before_smaller = mth.cost(d, mth.log(on ? (ea + e_smaller) / 2 : 1.0));
    if (!on) e_smaller = 0.0;
  }
#ifdef PSD_HELPER_WAVES
  if (HELP) {
    if (root_posted) {
      if (!mail_wait(chain)) err |= WERR_HELPER;
      if (two_roots) {
        /* the early exit of get_larger_root (fpl:75-79), which the helper leaves to us */
        const bool beyond = (o.cost2 < cost_diff_right && cost_diff_right < 0.0) ||
                            (o.cost2 > cost_diff_right && cost_diff_right > 0.0);
        larger_log_mean = beyond ? b + 1 : g_sm.mail[chain].res_large[lane_id()];
      }
    }
  } else
#endif
  {
    if (two_roots) larger_log_mean = get_larger_root(d, o, b, cost_diff_right, 0.0, &it_large, mth.rare_out());
  }
  (void)root_posted;
  PSD_PROF_ADD(PROF_C_LARGE);
  PSD_PROF_ITERS(PROF_IT_SMALL, it_small);
  PSD_PROF_ITERS(PROF_IT_LARGE, it_large);
  /* phase G, the part that needs both roots */
  if (!EARLY_TAIL && neither && two_roots) e_smaller = mth.exp(smaller_log_mean);
  double first_log_mean = PSD_INF, second_log_mean = PSD_INF;
  if (neither && two_roots) {
    bool larger_inside = a < larger_log_mean && larger_log_mean < b;
    bool smaller_inside = a < smaller_log_mean && 0 < e_smaller && smaller_log_mean < b;
    if (larger_inside) {
      if (smaller_inside && smaller_log_mean < larger_log_mean) {
        first_log_mean = smaller_log_mean;
        second_log_mean = larger_log_mean;
      } else {
        first_log_mean = larger_log_mean;
      }
    } else if (smaller_inside) {
      first_log_mean = smaller_log_mean;
    }
  }
  const bool crossing = neither && first_log_mean != PSD_INF;
  const bool two = crossing && second_log_mean != PSD_INF;
  const bool need_before =
      crossing && (!two || (second_log_mean - first_log_mean < first_log_mean - a));
  const bool need_other = crossing && !(two && need_before);
  /* between the crossings (two) or after the crossing (one): both are log-means */
  const double x_other =
      two ? (first_log_mean + second_log_mean) / 2 : (b + first_log_mean) / 2;
  double cost_diff_other = 0.0;
  if (need_other) cost_diff_other = mth.cost(d, x_other);
  double cost_diff_before = 0.0;
  if (EARLY_TAIL) {
    if (need_before) cost_diff_before = cost_before_smaller;
    /* the larger root is the first crossing: exp(first crossing) and the cost before it anew */
    const bool redo = need_before && first_log_mean != smaller_log_mean;
    if (ballot(redo)) {
      const double e_first = mth.exp(redo ? first_log_mean : 0.0);
      const double c_before = mth.cost(d, mth.log(redo ? (ea + e_first) / 2 : 1.0));
      if (redo) cost_diff_before = c_before;
    }
  } else {
    /* exp(first crossing) unless it is the value already computed */
    double e_first = e_smaller;
    if (need_before && first_log_mean != smaller_log_mean) e_first = mth.exp(first_log_mean);
    if (need_before) cost_diff_before = mth.cost(d, mth.log((ea + e_first) / 2));
  }

  PSD_PROF_ADD(PROF_C_TAIL);
  /* ---- decisions (no more transcendentals) ---- */
  if (!valid) return;
  const int by_mid = cost_diff_mid < 0 ? 0 : 1;
  if (triv) {
    cand_one(out, 0, a, b);
  } else if (both) { /* fpl:963-971 */
    cand_one(out, by_mid, a, b);
  } else if (degen) {
    if (d.Linear == 0) {
      cand_one(out, d.Constant < 0 ? 0 : 1, a, b);
    } else if (d.Constant == 0) {
      cand_one(out, d.Linear < 0 ? 0 : 1, a, b);
    } else if (a < lres && lres < b) {
      cand_two(out, (0 < d.Linear) ? 0 : 1, lres);
    } else {
      cand_one(out, by_mid, a, b);
    }
  } else if (same_at_right) { /* fpl:1029-1093 */
    if (two_roots) {
      double x = smaller_log_mean, opt = o.log_mean;
      if (a < x && x < opt && opt < b) {
        cand_two(out, (cost_diff_left < 0) ? 0 : 1, x);
      } else {
        bool it1_smaller_at_mean0 = 0 < d.Log;
        if (x < a) {
          cand_one(out, it1_smaller_at_mean0 ? 1 : 0, a, b);
        } else {
          cand_one(out, it1_smaller_at_mean0 ? 0 : 1, a, b);
        }
      }
    } else {
      cand_one(out, by_mid, a, b);
    }
  } else if (same_at_left) { /* fpl:1094-1123 */
    double x = larger_log_mean, opt = o.log_mean;
    if (two_roots && a < opt && opt < x && x < b) {
      cand_two(out, (cost_diff_right < 0) ? 1 : 0, x);
    } else {
      cand_one(out, by_mid, a, b);
    }
  } else if (two) { /* fpl:1171-1205 */
    bool it1_larger_before = need_before ? (cost_diff_before < 0) : !(cost_diff_other < 0);
    cand_three(out, it1_larger_before ? 0 : 1, first_log_mean, second_log_mean);
  } else if (crossing) { /* fpl:1206-1237 */
    if (cost_diff_before < 0) {
      if (cost_diff_other < 0) {
        cand_one(out, 0, a, b);
      } else {
        cand_two(out, 0, first_log_mean);
      }
    } else {
      if (cost_diff_other < 0) {
        cand_two(out, 1, first_log_mean);
      } else {
        cand_one(out, 1, a, b);
      }
    }
  } else { /* fpl:1238-1258 */
    double cost_diff;
    if (absd(cost_diff_mid) < NEWTON_EPSILON) {
      cost_diff = cost_diff_right;
    } else {
      cost_diff = cost_diff_mid;
    }
    cand_one(out, cost_diff < 0 ? 0 : 1, a, b);
  }
}

/* Loads merged interval (i1,i2): the two pieces and the interval [a,b] (fpl:876-932 without
 * the neighbour tests, see env_neighbour_flags). */
template <class L>
PSD_D void env_load_interval(const L &f1, int n1, const L &f2, int n2, int i1, int i2, Coef &c1,
                             Coef &c2, double &a, double &b, int &err) {
  c1 = load_coef(f1, i1);
  c2 = load_coef(f2, i2);
  double mn1 = f1.mn(i1), mx1 = f1.mx(i1), mn2 = f2.mn(i2), mx2 = f2.mx(i2);
  /* the piece that started earlier / ends later must have a neighbour on that side; the
   * reference would read a std::list sentinel otherwise */
  bool sentinel = (mn1 < mn2 && i2 == 0) || (mn2 < mn1 && i1 == 0) ||
                  (mn1 == mn2 && (i1 == 0) != (i2 == 0)) || (mx1 < mx2 && i1 + 1 >= n1) ||
                  (mx2 < mx1 && i2 + 1 >= n2) ||
                  (mx1 == mx2 && (i1 + 1 == n1) != (i2 + 1 == n2));
  a = mn1 < mn2 ? mn2 : mn1;
  b = mx1 < mx2 ? mx1 : mx2;
  if (sentinel) err |= WERR_SENTINEL;
  if (a == b) err |= WERR_ZERO_INTERVAL; /* fpl:933-944 */
}

/* same_at_left / same_at_right of push_min_pieces (fpl:876-932) compare the pieces next to
 * (it1, it2) in the two input lists.  Those neighbours are exactly the pair of pieces of the
 * previous / next merged interval: if it1 starts before it2 the previous interval is
 * (it1, prev2), if it2 starts first it is (prev1, it2), if both start together (prev1, prev2)
 * -- and the test made is sameFuns of that pair in each case; symmetrically on the right.
 * So same_at_left(k) = sameFuns of interval k-1, same_at_right(k) = sameFuns of interval k+1
 * (false at the two ends of the function, fpl:894-896,919-922). */
template <class L, class S>
PSD_D void env_neighbour_flags(const L &f1, const L &f2, const S &s, int k, int K, bool valid,
                               bool triv, bool &same_at_left, bool &same_at_right) {
  const int lane = lane_id();
  unsigned long long m_triv = ballot(valid && triv);
  same_at_left = lane > 0 && ((m_triv >> (lane - 1)) & 1ull) != 0;
  same_at_right = lane < WAVE - 1 && ((m_triv >> (lane + 1)) & 1ull) != 0;
  /* chunk edges (functions with more than 64 merged intervals): look the neighbour up */
  if (valid && lane == 0 && k > 0) {
    int e = s.iv(k - 1);
    same_at_left = same_funs(load_coef(f1, e >> 16), load_coef(f2, e & 0xffff));
  }
  if (valid && lane == WAVE - 1 && k + 1 < K) {
    int e = s.iv(k + 1);
    same_at_right = same_funs(load_coef(f1, e >> 16), load_coef(f2, e & 0xffff));
  }
}

/* push_piece's "same as last" test (fpl:1270-1273) */
PSD_D bool coalesces(const Coef &last, double last_prv, int last_di, const Coef &c, double prv,
                     int di) {
  return same_funs(last, c) && prv == last_prv && di == last_di;
}
PSD_D bool bit_identical(const Coef &last, double last_prv, int last_di, const Coef &c,
                         double prv, int di) {
  return psd_d2u(last.Linear) == psd_d2u(c.Linear) && psd_d2u(last.Log) == psd_d2u(c.Log) &&
         psd_d2u(last.Constant) == psd_d2u(c.Constant) && psd_d2u(last_prv) == psd_d2u(prv) &&
         last_di == di;
}

#ifdef PSD_HELPER_WAVES
/* HOP_ROOT: the larger roots of the lanes flagged by the chain wave.  Returns nonzero in a lane
 * that met a rare exp / log argument (NB only). */
template <bool NB>
PSD_D int helper_root_lanes(Mail &m, int lane) {
  StepMath<NB> mth;
  if (m.flags[lane] & 1) {
    const Coef d = {m.d_lin[lane], m.d_log[lane], m.d_con[lane]};
    /* the optimum of the difference piece exactly as env_classify_lanes derives it */
    PieceOpt o;
    o.mean = psd_div(-d.Log, d.Linear); /* (as env_classify_lanes: the same quotient) */
    o.log_mean = mth.log(o.mean);
    o.cost = mth.cost(d, o.log_mean);
    double loss_without_log_term = d.Linear * o.mean + d.Constant;
    o.cost2 = loss_without_log_term + o.log_mean * d.Log;
    const double b = m.b[lane];
    double root = PSD_INF;
    /* NaN as the right-end cost disables the early exit: the main wave applies it */
    if (has_two_roots(d, o, 0.0))
      root = get_larger_root(d, o, b, __builtin_nan(""), 0.0, nullptr, mth.rare_out());
    m.res_large[lane] = root;
  }
  return mth.rare;
}

/* the helper's share of an operation on lists in HBM (fpop_kernels.h) */
PSD_COLD_DEV void helper_hbm_op(const DeviceArgs &a, int chain, int op);
/* Body of a helper wave: serve the main wave of `chain` until HOP_EXIT. */
PSD_D void helper_loop(int chain, const DeviceArgs &a) {
  Mail &m = g_sm.mail[chain];
  const int lane = lane_id();
  int seen = 0;
  for (;;) {
    int cmd = seen;
    for (int spin = 0;; spin++) {
      cmd = flag_load(&m.seq_cmd);
      if (cmd != seen) break; /* (not a wait FOR a wave at work: the helper idles here) */
      if (spin > MAIL_SPIN_LIMIT || flag_load(&m.abort)) return;
      spin_pause();
    }
    seen = cmd;
    const int op = uniform_i(m.op);
    if (op == HOP_EXIT) return;
    if (op == HOP_BARRIER) {
      __syncthreads();
    } else if (op == HOP_ROOT) {
      /* without the rare-argument branches first; the complete functions if one was met */
      if (ballot(helper_root_lanes<true>(m, lane) != 0)) (void)helper_root_lanes<false>(m, lane);
    } else if (op >= HOP_HBM_COSTS) {
      helper_hbm_op(*a.self, chain, op);
    }
    wave_sync();
    if (lane == 0) flag_store(&m.seq_done, seen);
    }
}
#endif

/* exact sequential replay of fpl:832-860 + push_piece on lane 0 (cold path) */
template <class L, class S>
PSD_NOINLINE int min_env_serial(L f1_, int n1_, L f2_, int n2_, L out_, int cap_, S s_, int K_) {
  const L f1 = f1_.uniformed(), f2 = f2_.uniformed(), out = out_.uniformed();
  const S s = s_.uniformed();
  const int n1 = uniform_i(n1_), n2 = uniform_i(n2_), cap = uniform_i(cap_), K = uniform_i(K_);
  const int lane = lane_id();
  int count = 0;
  int err = 0;
  if (lane == 0) {
    for (int k = 0; k < K && !err; k++) {
      int e = s.iv(k);
      int i1 = e >> 16, i2 = e & 0xffff;
      Cands cd;
      Coef c1, c2;
      double ia, ib;
      env_interval_at(f1, n1, f2, n2, i1, i2, cd, ia, ib, c1, c2, err);
      for (int q = 0; q < cd.n; q++) {
        int src = cd.first ^ (q & 1);
        double lo = q == 0 ? ia : (q == 1 ? cd.x1 : cd.x2);
        double hi = q == cd.n - 1 ? ib : (q == 0 ? cd.x1 : cd.x2);
        Coef c = src ? c2 : c1;
        double prv = src ? f2.prv(i2) : f1.prv(i1);
        int di = src ? f2.di(i2) : f1.di(i1);
        if (count > 0 && coalesces(load_coef(out, count - 1), out.prv(count - 1),
                                   out.di(count - 1), c, prv, di)) {
          out.mx(count - 1) = hi;
        } else {
          if (count >= cap) {
            err |= WERR_OVERFLOW;
            break;
          }
          store_piece(out, count, c, lo, hi, di, prv);
          count++;
        }
      }
    }
  }
  wave_sync();
  err = shfl_i(err, 0);
  count = shfl_i(count, 0);
  return err ? -err : count;
}

/* min-envelope: out = pointwise min(f1, f2). */
template <bool HELP, bool SMALL, class L, class S, class M>
PSD_D int min_env_impl(L f1_, int n1_, L f2_, int n2_, L out_, int cap_, S s_, int chain_,
                       M &mth) {
  const int chain = uniform_i(chain_);
  const L f1 = f1_.uniformed(), f2 = f2_.uniformed(), out = out_.uniformed();
  const S s = s_.uniformed();
  const int n1 = uniform_i(n1_), n2 = uniform_i(n2_), cap = uniform_i(cap_);
  const int lane = lane_id();
  const int iv_cap = s.iv_cap();
  if (SMALL) PSD_ASSUME(n1 <= 32 && n2 <= 32);
  PSD_PROF_T0();
  /* ---- merged-interval table: interval k ends at the k-th distinct max_log_mean ---- */
  int K;
  if (SMALL || (n1 <= 32 && n2 <= 32)) { /* SMALL: guaranteed by the caller */
    /* both lists in one pass: lanes 0-31 rank the ends of f1 in f2, lanes 32-63 those of f2
     * in f1 (the usual case: one ballot, one pass of broadcast reads) */
    const int side = lane >> 5, idx = lane & 31;
    const int n_own = side ? n2 : n1, n_oth = side ? n1 : n2;
    const L &own = side ? f2 : f1;
    const L &oth = side ? f1 : f2;
    const bool valid = idx < n_own;
    double x = PSD_INF; /* lanes without an end never count as "less than" anything */
    int p = 0;
    bool dup = false;
    if (valid) x = own.mx(idx);
#ifndef PSD_RANK_BY_LDS
    /* Every end is in a register of its lane (f1's in lanes 0-31, f2's in lanes 32-63), so the
     * other list's ends are broadcast with v_readlane instead of read from LDS: four
     * instructions per end and no address arithmetic.  Both halves count against every
     * broadcast end; each keeps the count it needs. */
    int p_vs_f2 = 0, p_vs_f1 = 0;
    /* four ends per trip (a loop with cross-lane reads is not unrolled by the compiler); the
     * lanes read beyond the list hold +Inf */
    for (int j = 0; j < n2; j += 4) {
      p_vs_f2 += rdlane_d(x, 32 + j) < x ? 1 : 0;
      p_vs_f2 += rdlane_d(x, 33 + j) < x ? 1 : 0;
      p_vs_f2 += rdlane_d(x, 34 + j) < x ? 1 : 0;
      p_vs_f2 += rdlane_d(x, 35 + j) < x ? 1 : 0;
    }
    for (int j = 0; j < n1; j += 4) {
      p_vs_f1 += rdlane_d(x, j) < x ? 1 : 0;
      p_vs_f1 += rdlane_d(x, j + 1) < x ? 1 : 0;
      p_vs_f1 += rdlane_d(x, j + 2) < x ? 1 : 0;
      p_vs_f1 += rdlane_d(x, j + 3) < x ? 1 : 0;
    }
    p = side ? p_vs_f1 : p_vs_f2;
    /* both lists are sorted and free of repeats: an end also present in the other list is the
     * other list's end number p */
    if (valid && p < n_oth) dup = oth.mx(p) == x;
#else
    /* the other list's ends, eight independent LDS reads per round trip */
    for (int j0 = 0; j0 < n_oth; j0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        int j = j0 + u;
        v[u] = oth.mx(j < n_oth ? j : n_oth - 1);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (valid && j0 + u < n_oth) {
          p += v[u] < x ? 1 : 0;
          dup = dup || v[u] == x;
        }
      }
    }
#endif
    unsigned long long md = ballot(dup);
    const unsigned long long half = side ? (md >> 32) : (md & 0xffffffffull);
    if (valid && !(side && dup)) {
      int k = idx + p - popc64(half & lanes_below(idx));
      if (k < iv_cap) s.iv(k) = side ? ((p << 16) | idx) : ((idx << 16) | p);
    }
    K = n1 + n2 - popc64(md & 0xffffffffull);
  } else {
    int dup_before = 0; /* ends of f1 that are also ends of f2, among earlier chunks */
    for (int base = 0; base < n1; base += WAVE) {
      int i = base + lane;
      bool valid = i < n1;
      int p = 0;
      bool dup = false;
      if (valid) {
        double x = f1.mx(i);
        p = rank_mx(f2, n2, x);
        dup = p < n2 && f2.mx(p) == x;
      }
      unsigned long long md = ballot(dup);
      if (valid) {
        int k = i + p - (dup_before + popc64(md & lanes_below(lane)));
        if (k < iv_cap) s.iv(k) = (i << 16) | p;
      }
      dup_before += popc64(md);
    }
    int dup_total = dup_before;
    dup_before = 0;
    for (int base = 0; base < n2; base += WAVE) {
      int j = base + lane;
      bool valid = j < n2;
      int q = 0;
      bool dup = false;
      if (valid) {
        double x = f2.mx(j);
        q = rank_mx(f1, n1, x);
        dup = q < n1 && f1.mx(q) == x;
      }
      unsigned long long md = ballot(dup);
      if (valid && !dup) {
        int k = j + q - (dup_before + popc64(md & lanes_below(lane)));
        if (k < iv_cap) s.iv(k) = (q << 16) | j;
      }
      dup_before += popc64(md);
    }
    K = n1 + n2 - dup_total;
  }
  /* (i1 << 16) | i2 in a signed int: both indices stay below 32768 (SPILL_CAP_MAX) */
  if (K > iv_cap || n1 > SPILL_CAP_MAX || n2 > SPILL_CAP_MAX) return -WERR_OVERFLOW;
  wave_sync();
  PSD_PROF_ADD(PROF_TABLE);

  /* ---- one lane per interval; ballot/prefix-scan compaction ---- */
  int n_out = 0;
  int err = 0;
  bool need_serial = false;
  /* source piece of the last candidate emitted so far, (list << 20) | index; carried across
   * chunks */
  int last_id = -1;
  for (int base = 0; base < K; base += WAVE) {
    int k = base + lane;
    bool valid = k < K;
    Cands cd;
    cd.n = 0;
    cd.first = 0;
    cd.x1 = cd.x2 = 0.0;
    double ia = 0.0, ib = 0.0;
    Coef c1 = {0.0, 0.0, 0.0}, c2 = {0.0, 0.0, 0.0};
    double prv1 = 0.0, prv2 = 0.0;
    int di1 = 0, di2 = 0, i1 = 0, i2 = 0;
    bool sl = false, sr = false;
    PSD_PROF_T0();
    if (valid) {
      int e = s.iv(k);
      i1 = e >> 16;
      i2 = e & 0xffff;
      env_load_interval(f1, n1, f2, n2, i1, i2, c1, c2, ia, ib, err);
      prv1 = f1.prv(i1);
      di1 = f1.di(i1);
      prv2 = f2.prv(i2);
      di2 = f2.di(i2);
    }
    env_neighbour_flags(f1, f2, s, k, K, valid, valid && same_funs(c1, c2), sl, sr);
    PSD_PROF_ADD(PROF_C_LOAD);
    env_classify_lanes<HELP>(valid && err == 0, c1, c2, ia, ib, sl, sr, cd, chain, err, mth);
    PSD_PROF_ADD(PROF_CLASSIFY);
    /* first / last candidate of this lane */
    const int src0 = cd.first, src1 = cd.first ^ 1; /* the third piece has source src0 again */
    Coef fc = src0 ? c2 : c1;
    double fprv = src0 ? prv2 : prv1;
    int fdi = src0 ? di2 : di1;
    int lsrc = cd.n == 2 ? src1 : src0;
    /* piece q of this interval spans [lo_q, hi_q] */
    const double hi0 = cd.n == 1 ? ib : cd.x1;
    const double hi1 = cd.n == 2 ? ib : cd.x2;
    bool has = valid && cd.n > 0;
    unsigned long long m_has = ballot(has);
    unsigned long long m_err = ballot(err != 0);
    if (m_err) {
      int e = 0;
      for (int l = 0; l < WAVE; l++) e |= shfl_i(err, l);
      return -e;
    }
    /* predecessor = last candidate of the nearest lower lane that has one, else the carry.
     * Only its identity crosses lanes; its fields are re-read from the input list. */
    unsigned long long lb = lanes_below(lane);
    unsigned long long below = m_has & lb;
    const int my_last_id = (lsrc << 20) | (lsrc ? i2 : i1);
    int pid = shfl_i(my_last_id, below ? msb64(below) : 0); /* per-lane source */
    if (!below) pid = last_id;
    const bool have_pred = pid >= 0;
    Coef pc = {0.0, 0.0, 0.0};
    double pprv = 0.0;
    int pdi = 0;
    if (has && have_pred) {
      const L &pl = (pid >> 20) ? f2 : f1;
      const int pi = pid & 0xfffff;
      pc = load_coef(pl, pi);
      pprv = pl.prv(pi);
      pdi = pl.di(pi);
    }
    bool head0 = true; /* does the first candidate start a new output piece? */
    bool fuzzy = false;
    if (has && have_pred) {
      bool co = coalesces(pc, pprv, pdi, fc, fprv, fdi);
      bool bi = bit_identical(pc, pprv, pdi, fc, fprv, fdi);
      head0 = !co;
      fuzzy = co && !bi;
    }
    /* candidates 2 and 3 of a lane alternate it1/it2 with same_funs(it1,it2) false, so
     * they always start a new piece -- provided the run they follow is bit-identical to
     * its head, which `fuzzy` checks. */
    if (ballot(fuzzy)) {
      need_serial = true;
      break;
    }
    int heads = has ? ((head0 ? 1 : 0) + (cd.n - 1)) : 0;
    unsigned long long hb0 = ballot((heads & 1) != 0);
    unsigned long long hb1 = ballot((heads & 2) != 0);
    int heads_before = popc64(hb0 & lb) + 2 * popc64(hb1 & lb);
    int heads_total = popc64(hb0) + 2 * popc64(hb1);
    if (n_out + heads_total > cap) return -WERR_OVERFLOW;
    int slot = n_out + heads_before - (head0 ? 0 : 1); /* piece candidate 0 belongs to */
    if (has) {
      if (head0) store_piece(out, slot, fc, ia, hi0, fdi, fprv);
      if (cd.n >= 2) {
        Coef c = src1 ? c2 : c1;
        store_piece(out, slot + 1, c, cd.x1, hi1, src1 ? di2 : di1, src1 ? prv2 : prv1);
      }
      if (cd.n >= 3) store_piece(out, slot + 2, fc, cd.x2, ib, fdi, fprv);
    }
    wave_sync();
    /* a candidate that extends the previous piece only moves that piece's right end; of
     * the members of a run only the last one (in this chunk) writes, after the heads. */
    {
      unsigned long long m_head0 = ballot(has && head0);
      if (has && !head0) {
        unsigned long long above = m_has & ~lb & ~(1ull << lane);
        bool next_is_head = true;
        if (above) next_is_head = ((m_head0 >> ctz64(above)) & 1ull) != 0;
        if (cd.n >= 2 || next_is_head) out.mx(slot) = hi0;
      }
    }
    wave_sync();
    n_out += heads_total;
    if (m_has) last_id = rdlane_i(my_last_id, msb64(m_has));
    PSD_PROF_ADD(PROF_COMPACT);
    if (SMALL) break; /* K <= 64: one chunk */
  }
#ifdef PSD_FORCE_SERIAL_ENV /* tests only: every envelope takes the sequential replay */
  need_serial = true;
#endif
  if (need_serial) {
    /* the specialised version leaves the replay (and the call it takes) to the general one */
    if (SMALL) return -WERR_SERIAL;
    if (lane == 0) g_sm.serial[wave_id()]++;
    n_out = min_env_serial(f1, n1, f2, n2, out, cap, s, K);
    PSD_PROF_ADD(PROF_SERIAL);
  }
  return n_out;
}

#ifdef PSD_HELPER_WAVES
/* ---- the envelope of two functions in HBM, by the chain wave and its helper ----------------
 * Functions of hundreds of pieces (adversarial data) are processed in chunks of 64 merged
 * intervals; the chunks are independent up to the compaction, which needs the number of
 * pieces emitted so far.  The helper wave classifies the odd chunks and leaves its results
 * (shape, first source, the two crossings, error bits) in HBM; the chain wave classifies the
 * even chunks and compacts all chunks in order, reading the helper's results as they come.
 * The arithmetic per interval is that of min_env_impl: same lists, bit for bit. */

/* With the lists in HBM the LDS-resident lists are dead storage: the ends (max_log_mean) of
 * the function a wave ranks against are staged there, so that the binary search of every lane
 * (ten to fifteen dependent reads) runs at LDS instead of L2 latency.  Chain c owns the storage
 * of lists 3c..3c+2, the chain wave the first part (the ends of f2), its helper the rest (the
 * ends of f1); functions too long for it are searched in HBM as before. */
constexpr int COOP_STAGE_DOUBLES = 3 * (int)(sizeof(ListStore) / sizeof(double));
PSD_D ldouble *coop_stage(int chain) { return (ldouble *)&g_sm.list[3 * chain]; }
template <class L>
PSD_D void coop_stage_ends(const L &f, int n, ldouble *dst) {
  for (int i = lane_id(); i < n; i += WAVE) dst[i] = f.mx(i);
  wave_sync();
}
/* number of entries of the sorted array a[0..n) below x */
PSD_D int rank_staged(const ldouble *a, int n, double x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < x) {
      lo = mid + 1;
    } else {
      hi = mid;
    }
  }
  return lo;
}

/* merged-interval table, the entries owned by f1 (every end of f1); returns how many ends of
 * f1 are also ends of f2.  staged: the ends of f2 in LDS (or nullptr) */
template <class L, class S>
PSD_D int env_table_first(const L &f1, int n1, const L &f2, int n2, const S &s,
                          const ldouble *staged) {
  const int lane = lane_id();
  const int iv_cap = s.iv_cap();
  int dup_before = 0;
  for (int base = 0; base < n1; base += WAVE) {
    int i = base + lane;
    bool valid = i < n1;
    int p = 0;
    bool dup = false;
    if (valid) {
      double x = f1.mx(i);
      if (staged) {
        p = rank_staged(staged, n2, x);
        dup = p < n2 && staged[p] == x;
      } else {
        p = rank_mx(f2, n2, x);
        dup = p < n2 && f2.mx(p) == x;
      }
    }
    unsigned long long md = ballot(dup);
    if (valid) {
      int k = i + p - (dup_before + popc64(md & lanes_below(lane)));
      if (k < iv_cap) s.iv(k) = (i << 16) | p;
    }
    dup_before += popc64(md);
  }
  return dup_before;
}
/* ... and the entries owned by f2 (its ends that are not ends of f1); staged: the ends of f1 */
template <class L, class S>
PSD_D void env_table_second(const L &f1, int n1, const L &f2, int n2, const S &s,
                            const ldouble *staged) {
  const int lane = lane_id();
  const int iv_cap = s.iv_cap();
  int dup_before = 0;
  for (int base = 0; base < n2; base += WAVE) {
    int j = base + lane;
    bool valid = j < n2;
    int q = 0;
    bool dup = false;
    if (valid) {
      double x = f2.mx(j);
      if (staged) {
        q = rank_staged(staged, n1, x);
        dup = q < n1 && staged[q] == x;
      } else {
        q = rank_mx(f1, n1, x);
        dup = q < n1 && f1.mx(q) == x;
      }
    }
    unsigned long long md = ballot(dup);
    if (valid && !dup) {
      int k = j + q - (dup_before + popc64(md & lanes_below(lane)));
      if (k < iv_cap) s.iv(k) = (q << 16) | j;
    }
    dup_before += popc64(md);
  }
}

/* one chunk of merged intervals: everything up to the candidates (the first half of the chunk
 * loop of min_env_impl) */
struct EnvLane {
  Cands cd;
  double ia, ib;
  Coef c1, c2;
  double prv1, prv2;
  int di1, di2, i1, i2;
  int err;
};
template <class L, class S>
PSD_D void env_coop_load(const L &f1, int n1, const L &f2, int n2, const S &s, int k, bool valid,
                         EnvLane &e) {
  e.cd.n = 0;
  e.cd.first = 0;
  e.cd.x1 = e.cd.x2 = 0.0;
  e.ia = e.ib = 0.0;
  e.c1.Linear = e.c1.Log = e.c1.Constant = 0.0;
  e.c2 = e.c1;
  e.prv1 = e.prv2 = 0.0;
  e.di1 = e.di2 = e.i1 = e.i2 = 0;
  e.err = 0;
  if (valid) {
    int en = s.iv(k);
    e.i1 = en >> 16;
    e.i2 = en & 0xffff;
    env_load_interval(f1, n1, f2, n2, e.i1, e.i2, e.c1, e.c2, e.ia, e.ib, e.err);
    e.prv1 = f1.prv(e.i1);
    e.di1 = f1.di(e.i1);
    e.prv2 = f2.prv(e.i2);
    e.di2 = f2.di(e.i2);
  }
}
template <class L, class S>
PSD_D void env_coop_classify(const L &f1, int n1, const L &f2, int n2, const S &s, int K, int base,
                             int chain, EnvLane &e) {
  const int k = base + lane_id();
  const bool valid = k < K;
  env_coop_load(f1, n1, f2, n2, s, k, valid, e);
  bool sl = false, sr = false;
  env_neighbour_flags(f1, f2, s, k, K, valid, valid && same_funs(e.c1, e.c2), sl, sr);
  MathFull mth;
  env_classify_lanes<false>(valid && e.err == 0, e.c1, e.c2, e.ia, e.ib, sl, sr, e.cd, chain, e.err,
                            mth);
}

/* Which chunks of merged intervals the helper classifies: all but every PSD_COOP_PERIOD-th.
 * The chain wave also compacts every chunk (and re-reads the pieces of the helper's chunks for
 * that); measured shares from 3/5 to all: profiles/r03/ab_hbm_helper_share_and_stealing.log. */
#ifndef PSD_COOP_PERIOD
#define PSD_COOP_PERIOD 4
#endif
PSD_D bool coop_helper_owns(int chunk) { return chunk % PSD_COOP_PERIOD != 0; }
/* helper-owned chunks among chunks 0..chunk */
PSD_D int coop_helper_chunks_upto(int chunk) {
  return (chunk / PSD_COOP_PERIOD) * (PSD_COOP_PERIOD - 1) + chunk % PSD_COOP_PERIOD;
}
/* the helper's share: its chunks, results to HBM, progress published chunk by chunk */
template <class L, class S>
PSD_D void env_coop_helper(const L &f1, int n1, const L &f2, int n2, const S &s, int K, int chain) {
  Mail &m = g_sm.mail[chain];
  int done = 0;
  for (int base = 0, chunk = 0; base < K; base += WAVE, chunk++) {
    if (!coop_helper_owns(chunk)) continue;
    EnvLane e;
    env_coop_classify(f1, n1, f2, n2, s, K, base, chain, e);
    const int k = base + lane_id();
    if (k < K) {
      s.coop_x1(k) = e.cd.x1;
      s.coop_x2(k) = e.cd.x2;
      s.coop_code(k) = (double)(e.cd.n | (e.cd.first << 2) | (e.err << 3));
    }
    done++;
    wave_sync();
    if (lane_id() == 0) flag_store(&m.h_progress, done);
  }
}

/* the chain wave: table (with the helper), its own chunks, and the compaction of all */
template <class L, class S>
PSD_D int min_env_coop(L f1_, int n1_, L f2_, int n2_, L out_, int cap_, S s_, int chain_, int p_,
                       int id1_, int off1_, int id2_) {
  const int chain = uniform_i(chain_);
  const L f1 = f1_.uniformed(), f2 = f2_.uniformed(), out = out_.uniformed();
  const S s = s_.uniformed();
  const int n1 = uniform_i(n1_), n2 = uniform_i(n2_), cap = uniform_i(cap_);
  const int lane = lane_id();
  const int iv_cap = s.iv_cap();
  Mail &m = g_sm.mail[chain];
  PSD_PROF_T0();
  if (lane == 0) {
    m.h_arg[0] = p_;
    m.h_arg[1] = id1_;
    m.h_arg[2] = off1_;
    m.h_arg[3] = n1;
    m.h_arg[4] = id2_;
    m.h_arg[5] = n2;
  }
  mail_post(chain, HOP_HBM_TABLE);
  const ldouble *staged = nullptr;
  if (n1 + n2 <= COOP_STAGE_DOUBLES) { /* (the helper makes the same test) */
    ldouble *dst = coop_stage(chain);
    coop_stage_ends(f2, n2, dst);
    staged = dst;
  }
  const int dup_total = env_table_first(f1, n1, f2, n2, s, staged);
  if (!mail_wait(chain)) return -WERR_HELPER;
  const int K = n1 + n2 - dup_total;
  if (K > iv_cap || n1 > SPILL_CAP_MAX || n2 > SPILL_CAP_MAX) return -WERR_OVERFLOW;
  wave_sync();
  PSD_PROF_ADD(PROF_TABLE);
  if (lane == 0) {
    m.h_arg[6] = K;
    flag_store(&m.h_progress, 0);
  }
  mail_post(chain, HOP_HBM_CLASSIFY);

  int n_out = 0;
  int err = 0;
  bool need_serial = false, overflow = false, helper_lost = false;
  int last_id = -1;
  for (int base = 0, chunk = 0; base < K; base += WAVE, chunk++) {
    const int k = base + lane;
    const bool valid = k < K;
    EnvLane e;
    PSD_PROF_T0();
    if (!coop_helper_owns(chunk)) {
      env_coop_classify(f1, n1, f2, n2, s, K, base, chain, e);
    } else {
      /* the helper's chunk: wait for it, then fetch its results and the pieces they refer to */
      const int want = coop_helper_chunks_upto(chunk);
      bool there = false;
      for (int spin = 0; spin < MAIL_SPIN_LIMIT; spin++) {
        if (rdlane_i(flag_load(&m.h_progress), 0) >= want) {
          there = true;
          PSD_SPIN_NOTE(spin);
          break;
        }
        spin_pause();
      }
      if (!there) {
        helper_lost = true;
        break;
      }
      env_coop_load(f1, n1, f2, n2, s, k, valid, e);
      if (valid) {
        const int code = (int)s.coop_code(k);
        e.cd.n = code & 3;
        e.cd.first = (code >> 2) & 1;
        e.err |= code >> 3;
        e.cd.x1 = s.coop_x1(k);
        e.cd.x2 = s.coop_x2(k);
      }
    }
    err = e.err;
    PSD_PROF_ADD(PROF_CLASSIFY);
    /* ---- compaction: as in min_env_impl ---- */
    const Cands &cd = e.cd;
    const int src0 = cd.first, src1 = cd.first ^ 1;
    Coef fc = src0 ? e.c2 : e.c1;
    double fprv = src0 ? e.prv2 : e.prv1;
    int fdi = src0 ? e.di2 : e.di1;
    int lsrc = cd.n == 2 ? src1 : src0;
    const double hi0 = cd.n == 1 ? e.ib : cd.x1;
    const double hi1 = cd.n == 2 ? e.ib : cd.x2;
    bool has = valid && cd.n > 0;
    unsigned long long m_has = ballot(has);
    unsigned long long m_err = ballot(err != 0);
    if (m_err) {
      int eb = 0;
      for (int l = 0; l < WAVE; l++) eb |= shfl_i(err, l);
      err = eb;
      break;
    }
    unsigned long long lb = lanes_below(lane);
    unsigned long long below = m_has & lb;
    const int my_last_id = (lsrc << 20) | (lsrc ? e.i2 : e.i1);
    int pid = shfl_i(my_last_id, below ? msb64(below) : 0);
    if (!below) pid = last_id;
    const bool have_pred = pid >= 0;
    Coef pc = {0.0, 0.0, 0.0};
    double pprv = 0.0;
    int pdi = 0;
    if (has && have_pred) {
      const L &pl = (pid >> 20) ? f2 : f1;
      const int pi = pid & 0xfffff;
      pc = load_coef(pl, pi);
      pprv = pl.prv(pi);
      pdi = pl.di(pi);
    }
    bool head0 = true;
    bool fuzzy = false;
    if (has && have_pred) {
      bool co = coalesces(pc, pprv, pdi, fc, fprv, fdi);
      bool bi = bit_identical(pc, pprv, pdi, fc, fprv, fdi);
      head0 = !co;
      fuzzy = co && !bi;
    }
    if (ballot(fuzzy)) {
      need_serial = true;
      break;
    }
    int heads = has ? ((head0 ? 1 : 0) + (cd.n - 1)) : 0;
    unsigned long long hb0 = ballot((heads & 1) != 0);
    unsigned long long hb1 = ballot((heads & 2) != 0);
    int heads_before = popc64(hb0 & lb) + 2 * popc64(hb1 & lb);
    int heads_total = popc64(hb0) + 2 * popc64(hb1);
    if (n_out + heads_total > cap) {
      overflow = true;
      break;
    }
    int slot = n_out + heads_before - (head0 ? 0 : 1);
    if (has) {
      if (head0) store_piece(out, slot, fc, e.ia, hi0, fdi, fprv);
      if (cd.n >= 2) {
        Coef c = src1 ? e.c2 : e.c1;
        store_piece(out, slot + 1, c, cd.x1, hi1, src1 ? e.di2 : e.di1, src1 ? e.prv2 : e.prv1);
      }
      if (cd.n >= 3) store_piece(out, slot + 2, fc, cd.x2, e.ib, fdi, fprv);
    }
    wave_sync();
    {
      unsigned long long m_head0 = ballot(has && head0);
      if (has && !head0) {
        unsigned long long above = m_has & ~lb & ~(1ull << lane);
        bool next_is_head = true;
        if (above) next_is_head = ((m_head0 >> ctz64(above)) & 1ull) != 0;
        if (cd.n >= 2 || next_is_head) out.mx(slot) = hi0;
      }
    }
    wave_sync();
    n_out += heads_total;
    if (m_has) last_id = rdlane_i(my_last_id, msb64(m_has));
    PSD_PROF_ADD(PROF_COMPACT);
  }
  /* the helper finishes its chunks whatever happened here (they are bounded work) */
  if (!mail_wait(chain) || helper_lost) return -WERR_HELPER;
  if (err) return -err;
  if (overflow) return -WERR_OVERFLOW;
#ifdef PSD_FORCE_SERIAL_ENV
  need_serial = true;
#endif
  if (need_serial) {
    if (lane == 0) g_sm.serial[wave_id()]++;
    n_out = min_env_serial(f1, n1, f2, n2, out, cap, s, K);
  }
  return n_out;
}
#endif /* PSD_HELPER_WAVES */

/* The three operations, out of line (see the head of this file).  The kernel inlines the
 * LDS instantiations into its loop -- one copy each, ~40 KB of code, and no call overhead
 * (saving and restoring ~40 scalar and ~40 vector registers per call) -- and calls these
 * wrappers for the HBM spill path. */
template <class L, class S>
PSD_NOINLINE int min_less_wave(L in, int n, L out, int cap, S s, int data_i_out,
                               double add_const) {
  MathFull mth;
  return min_less_impl<false>(in, n, out, cap, s, data_i_out, add_const, mth);
}
template <class L, class S>
PSD_NOINLINE int min_more_wave(L in, int n, L out, int cap, S s, int data_i_out) {
  MathFull mth;
  return min_more_impl<false>(in, n, out, cap, s, data_i_out, mth);
}
template <bool HELP, class L, class S>
PSD_NOINLINE int min_env_wave(L f1, int n1, L f2, int n2, L out, int cap, S s, int chain) {
  MathFull mth;
  return min_env_impl<HELP, false>(f1, n1, f2, n2, out, cap, s, chain, mth);
}
/* the same, specialised for n <= 64 (min_env: n1, n2 <= 32): what the throughput build, whose
 * operations all stay out of line, calls for nearly every data point.  Their exp / log come
 * without the rare-argument branches (StepMath): -WERR_SERIAL when such an argument was met,
 * and the caller takes the general version above. */
template <class L, class S>
PSD_NOINLINE int min_less_small_wave(L in, int n, L out, int cap, S s, int data_i_out,
                                     double add_const) {
  MathFast mth;
  const int r = min_less_impl<true>(in, n, out, cap, s, data_i_out, add_const, mth);
  return ballot(mth.rare != 0) ? -WERR_SERIAL : r;
}
template <class L, class S>
PSD_NOINLINE int min_more_small_wave(L in, int n, L out, int cap, S s, int data_i_out) {
  MathFast mth;
  const int r = min_more_impl<true>(in, n, out, cap, s, data_i_out, mth);
  return ballot(mth.rare != 0) ? -WERR_SERIAL : r;
}
template <bool HELP, class L, class S>
PSD_NOINLINE int min_env_small_wave(L f1, int n1, L f2, int n2, L out, int cap, S s, int chain) {
  MathFast mth;
  const int r = min_env_impl<HELP, true>(f1, n1, f2, n2, out, cap, s, chain, mth);
  return ballot(mth.rare != 0) ? -WERR_SERIAL : r;
}

}  // namespace PSD_VARIANT
}  // namespace psd
