/* fpop_kernels.h -- the HIP kernels of the PeakSegFPOP hot path.
 *
 *   fpop_forward_kernel   one workgroup per (penalty, contig) problem: wave 0 carries the "up"
 *                         cost function, wave 1 the "down" one (waves 2 and 3, when
 *                         PSD_HELPER_WAVES is defined, are their helpers, fpop_wave.h), through
 *                         the strictly sequential recurrence of
 *                         /root/reference/src/PeakSegFPOPLog.cpp:258-397 (one __syncthreads
 *                         per data point; the two updates of a step only read the previous
 *                         step's functions).  Live piece lists sit in LDS; each step's
 *                         backtrack record {max_log_mean, data_i, prev_log_mean} per piece --
 *                         what the reference serialises to its DiskVector (drv:12-34) -- is
 *                         appended to the in-HBM arena.
 *                         After its last data point the down wave decodes the segmentation
 *                         from the arena (backtrack_wave; drv:399-442: Minimize result, then
 *                         findMean per segment).
 *   math_probe_kernel     element-wise psd_exp / psd_log / psd_div (tests: host == device bit
 *                         for bit) and the compiler's own fp64 division.
 *
 * Included by peakseg_hip.cpp (hipcc, gfx950) and by tests/emu (g++ + hip_emu.h).
 *
 * NO include guard: peakseg_hip.cpp includes this file once per build variant (PSD_VARIANT =
 * namespace name, PSD_LDS_CAP, PSD_HELPER_WAVES), see fpop_wave.h. */
#include "fpop_wave.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace psd {
namespace PSD_VARIANT {

/* lists / scratch of spill-pool slot p (a problem's slot, see take_spill_slot) */
PSD_D GlobalList global_list(const DeviceArgs &a, int p, int id) {
  const size_t cap = (size_t)a.spill_cap;
  gdouble *f = (gdouble *)(a.spill_f64 + ((size_t)p * 48 + (size_t)id * 6) * cap);
  GlobalList r;
  r.Lin_ = f;
  r.Log_ = f + cap;
  r.Con_ = f + 2 * cap;
  r.mn_ = f + 3 * cap;
  r.mx_ = f + 4 * cap;
  r.prv_ = f + 5 * cap;
  r.di_ = (gint *)(a.spill_i32 + ((size_t)p * 12 + (size_t)id) * cap);
  return r;
}
PSD_D GlobalScratch global_scratch(const DeviceArgs &a, int p, int wave) {
  const size_t cap = (size_t)a.spill_cap;
  gdouble *f = (gdouble *)(a.spill_f64 + ((size_t)p * 48 + 36 + (size_t)wave * 6) * cap);
  gint *q = (gint *)(a.spill_i32 + ((size_t)p * 12 + 6) * cap);
  GlobalScratch r;
  r.lc_ = f;
  r.rc_ = f + cap;
  r.om_ = f + 2 * cap;
  r.mu_ = f + 3 * cap;
  r.muc_ = f + 4 * cap;
  r.oc2_ = f + 5 * cap;
  r.cls_ = q + (size_t)wave * cap;
  r.iv_ = q + 2 * cap + (size_t)wave * 2 * cap;
  r.iv_cap_ = 2 * a.spill_cap;
  return r;
}

template <class L>
PSD_D void copy_list_wave(const L &src, int n, const L &dst) {
  const int lane = lane_id();
  for (int base = 0; base < n; base += WAVE) {
    int i = base + lane;
    if (i < n)
      store_piece(dst, i, load_coef(src, i), src.mn(i), src.mx(i), src.di(i), src.prv(i));
  }
}

/* where piece offset `off` of the arena lives (a.ar_block[] must hold the block already) */
struct ArenaPtr {
  gdouble *mx, *prv;
  gint *di;
};
PSD_D ArenaPtr arena_ptr(char *block_base, int block_log2, unsigned long long within) {
  ArenaPtr r;
  r.mx = (gdouble *)block_base + within;
  r.prv = (gdouble *)(block_base + (8ull << block_log2)) + within;
  r.di = (gint *)(block_base + (16ull << block_log2)) + within;
  return r;
}
PSD_D ArenaPtr arena_at(const DeviceArgs &a, unsigned long long off) {
  const int blg = a.ar_block_log2;
  /* (every lane loads the same entry: the address is wave-uniform, say so) */
  return arena_ptr(uniform_p(agent_load_ptr(&a.ar_block[off >> blg])), blg,
                   off & ((1ull << blg) - 1ull));
}

/* The three addresses of the current chunk run are needed once per data point, by the lanes
 * that store the record: they live in LDS (g_sm.cur_ptr: one broadcast read next to the reads of
 * the record itself), not in six scalar registers that the whole loop would carry -- the latency
 * build spills scalar registers as it is (cursor in registers: 2172 ms, in LDS: 2120 ms on
 * 200 k bins x 64, profiles/r04/ab_cursor_in_lds_packed_flag_newton_trim.log). */
struct ArenaCursor {
  unsigned long long base; /* first piece of the current chunk run */
  int used, room;
  int store; /* 0: the forward pass of the checkpointed store keeps no per-step records */
};
PSD_D void cursor_clear(ArenaCursor &cur, int store) {
  cur.base = 0;
  cur.used = 0;
  cur.room = 0;
  cur.store = store;
}
PSD_D void cursor_point(const DeviceArgs &a, ArenaCursor &cur, unsigned long long base, int room) {
  const ArenaPtr q = arena_at(a, base);
  wave_sync();
  if (lane_id() == 0) {
    g_sm.cur_ptr[wave_id() & 1][0] = (unsigned long long)q.mx;
    g_sm.cur_ptr[wave_id() & 1][1] = (unsigned long long)q.prv;
    g_sm.cur_ptr[wave_id() & 1][2] = (unsigned long long)q.di;
  }
  wave_sync();
  cur.base = base;
  cur.used = 0;
  cur.room = room;
}
PSD_D gdouble *cursor_mx() { return (gdouble *)g_sm.cur_ptr[wave_id() & 1][0]; }
PSD_D gdouble *cursor_prv() { return (gdouble *)g_sm.cur_ptr[wave_id() & 1][1]; }
PSD_D gint *cursor_di() { return (gint *)g_sm.cur_ptr[wave_id() & 1][2]; }

/* Reserve arena room for a function of n pieces: the next run of whole chunks for this wave
 * (cold: once per chunk of 2^ar_chunk_log2 pieces), inside ONE block.  Returns the first piece
 * index of the run; ~0 when the arena is exhausted.  A run beyond what was mapped at launch
 * waits for the host, which maps ahead of ar_used while the kernel runs; the arena is exhausted
 * when the host says that no more will come (or never answers). */
PSD_COLD_DEV unsigned long long arena_take(const DeviceArgs &a, int n) {
  if (a.ckpt_interval > 0) return ~0ull; /* checkpointed store: the wave's region is all it has */
  const int lg = a.ar_chunk_log2, blg = a.ar_block_log2;
  const unsigned long long chunks = ((unsigned long long)uniform_i(n) + (1ull << lg) - 1ull) >> lg;
  if ((chunks << lg) > (1ull << blg)) return ~0ull; /* (a function is never longer than a block) */
  /* How long a wave waits for the host to map a block: a block is 10-340 MB at 13-35 ms per GB,
   * so a fifth of a second is generous -- and it must stay far below the bound on the waits
   * BETWEEN waves (WAIT_SPIN_LIMIT, seconds): the other chain's wave sits at the data point's
   * barrier meanwhile.  Past it the problem is parked, which costs a relaunch, not a result. */
  constexpr long long LIVE_WAIT_CYCLES = 480000000ll; /* 0.2 s at 2.4 GHz */
  constexpr int LIVE_SPIN_LIMIT = 1 << 22; /* (the emulator has no clock: polls) */
  for (;;) {
    unsigned long long first = 0;
    if (lane_id() == 0) {
      first = atomicAdd(a.ar_next_chunk, chunks);
      if (a.ar_used) sys_add_u64(a.ar_used, chunks << lg);
    }
    first = psd_d2u(rdlane_d(psd_u2d(first), 0));
    const unsigned long long lo = first << lg, hi = (first + chunks) << lg;
    if ((lo >> blg) != ((hi - 1ull) >> blg)) continue; /* straddles two blocks: the next run */
    if (hi > a.ar_cap) {
      if (a.ar_live == nullptr) return ~0ull;
      bool mapped = false;
      const long long t_wait = cycle_now();
      for (int spin = 0; spin < LIVE_SPIN_LIMIT; spin++) {
        /* (the flag first: a capacity published before it is final) */
        const unsigned long long final_now = psd_d2u(uniform_d(psd_u2d(sys_load_u64(&a.ar_live[1]))));
        const unsigned long long cap_now = psd_d2u(uniform_d(psd_u2d(sys_load_u64(&a.ar_live[0]))));
        if (hi <= cap_now) {
          mapped = true;
          break;
        }
        if (final_now || cycle_now() - t_wait > LIVE_WAIT_CYCLES) break;
        spin_pause();
      }
      if (!mapped) return ~0ull;
      /* a block the host added during this launch: its address goes into the device table */
      const unsigned long long blk = lo >> blg;
      if (lane_id() == 0)
        agent_store_ptr(&a.ar_block[blk], (char *)sys_load_u64(&a.ar_live[2ull + blk]));
      wave_sync();
    }
    return lo;
  }
}
PSD_D int arena_room_for(const DeviceArgs &a, int n) {
  const int lg = a.ar_chunk_log2;
  return (int)((((unsigned)n + (1u << lg) - 1u) >> lg) << lg);
}
/* the next run of chunks for a function of n pieces -> cursor; false when the arena is full */
PSD_D bool cursor_take(const DeviceArgs &a, ArenaCursor &cur, int n) {
  unsigned long long base = psd_d2u(uniform_d(psd_u2d(arena_take(*a.self, n))));
  if (base == ~0ull) return false;
  cursor_point(a, cur, base, arena_room_for(a, n));
  return true;
}

/* Append one function's backtrack record to the arena; returns false when it is full. */
template <class L>
PSD_D bool arena_store_wave(const DeviceArgs &a, ArenaCursor &cur, const L &f, int n,
                            unsigned long long fn_index) {
  const int lane = lane_id();
  if (!cur.store) return true;
  if (n > cur.room - cur.used && !cursor_take(a, cur, n)) return false;
  const int at = cur.used;
  for (int base = 0; base < n; base += WAVE) {
    int i = base + lane;
    if (i < n) {
      cursor_mx()[at + i] = f.mx(i);
      cursor_prv()[at + i] = f.prv(i);
      cursor_di()[at + i] = f.di(i);
    }
  }
  if (lane == 0)
    ((gull *)a.fn_ref)[fn_index] =
        ((cur.base + (unsigned long long)at) << FN_COUNT_BITS) | (unsigned long long)n;
  cur.used += n;
  return true;
}

/* Last phase of a step, fused: f <- (f * W_{t-1} + (w, -z w, 0)) * (1/W_t)  (drv:316-321 /
 * 365-370: multiply, add, multiply, no contraction) and the function's backtrack record
 * {max_log_mean, data_i, prev_log_mean} appended to the arena, each piece touched once.
 * Returns false when the arena is full. */
template <class L>
PSD_D bool scale_add_store_wave(const DeviceArgs &a, ArenaCursor &cur, const L &f, int n,
                                unsigned long long fn_index, bool store, double cum_weight_prev,
                                double add_linear, double add_log, double inv_cum_weight) {
  const int lane = lane_id();
  bool ok = true;
  store = store && cur.store != 0;
  if (store && n > cur.room - cur.used && !cursor_take(a, cur, n)) {
    ok = false;
    store = false;
  }
  const int at = cur.used;
  for (int base = 0; base < n; base += WAVE) {
    int i = base + lane;
    if (i < n) {
      double li = f.Lin(i) * cum_weight_prev;
      double lo = f.Log(i) * cum_weight_prev;
      double co = f.Con(i) * cum_weight_prev;
      double mx = f.mx(i), prv = f.prv(i);
      int di = f.di(i);
      li = li + add_linear;
      lo = lo + add_log;
      co = co + 0.0;
      f.Lin(i) = li * inv_cum_weight;
      f.Log(i) = lo * inv_cum_weight;
      f.Con(i) = co * inv_cum_weight;
#ifdef PSD_DEBUG_DUMP /* diagnostic builds: every piece of every function, as bits */
      printf("D %llu %d %d %016llx %016llx %016llx %016llx %016llx %d\n", fn_index, n, i,
             (unsigned long long)psd_d2u((double)f.Lin(i)), (unsigned long long)psd_d2u((double)f.Log(i)),
             (unsigned long long)psd_d2u((double)f.Con(i)), (unsigned long long)psd_d2u(mx),
             (unsigned long long)psd_d2u(prv), di);
#endif
      if (store) {
        cursor_mx()[at + i] = mx;
        cursor_prv()[at + i] = prv;
        cursor_di()[at + i] = di;
      }
    }
  }
  if (store) {
    if (lane == 0)
      ((gull *)a.fn_ref)[fn_index] =
          ((cur.base + (unsigned long long)at) << FN_COUNT_BITS) | (unsigned long long)n;
    cur.used += n;
  }
  return ok;
}

/* Minimize (fpl:689-712): first strict minimum over pieces of the clamped optimum. */
template <class L>
PSD_D void minimize_wave(const L &f, int n, double *best_cost, double *best_log_mean,
                         int *data_i, double *prev_log_mean) {
  const int lane = lane_id();
  double bc = PSD_INF, blm = 0.0, bprv = 0.0;
  int bdi = 0;
  for (int base = 0; base < n; base += WAVE) {
    int i = base + lane;
    double cost = PSD_INF, lm = 0.0;
    if (i < n) {
      Coef c = load_coef(f, i);
      lm = argmin(c);
      if (lm < f.mn(i)) {
        lm = f.mn(i);
      } else if (f.mx(i) < lm) {
        lm = f.mx(i);
      }
      cost = get_cost(c, lm);
    }
    /* lowest lane among those holding the chunk minimum; NaN never wins a strict '<' */
    bool usable = i < n && cost < PSD_INF;
    double v = usable ? cost : PSD_INF;
    double mn = v;
    for (int sft = 1; sft < WAVE; sft <<= 1) {
      double o = shfl_d(mn, lane ^ sft);
      mn = o < mn ? o : mn;
    }
    unsigned long long m = ballot(usable && v == mn);
    if (m && mn < bc) {
      int src = ctz64(m);
      bc = mn;
      blm = rdlane_d(lm, src);
      int ii = base + src;
      bdi = f.di(ii);
      bprv = f.prv(ii);
    }
  }
  *best_cost = bc;
  *best_log_mean = blm;
  *data_i = bdi;
  *prev_log_mean = bprv;
}

template <class LS, class LD>
PSD_D void copy_list_across(const LS &src, int n, const LD &dst) {
  const int lane = lane_id();
  for (int base = 0; base < n; base += WAVE) {
    int i = base + lane;
    if (i < n)
      store_piece(dst, i, load_coef(src, i), src.mn(i), src.mx(i), src.di(i), src.prv(i));
  }
}

/* One chain's update for data point t >= 1 (chain 0: up_t, chain 1: down_t):
 *   up_t   = min_env(min_less(down_{t-1}) + penalty/W_{t-1}, up_{t-1})   drv:273-300
 *   down_t = min_env(min_more(up_{t-1}),                    down_{t-1})  drv:324-349
 *   (t == 1: up_1 = the min-less result, down_1 = down_0)
 * then multiply, add the data point, multiply (drv:316-321,365-370).
 * Returns the new piece count or -(WERR_* bits). */
template <bool HELP, class L, class S>
PSD_D int chain_step(const DeviceArgs &a, ArenaCursor &cur, unsigned long long fn_index,
                     int chain, int t, const L &other_prev, int n_other, const L &own_prev,
                     int n_own, const L &own_new, const L &mlist, const S &sc, int cap,
                     double pen_term, double cum_weight_prev, double w, int coverage,
                     double cum_weight) {
  int nm = 0;
  /* operations out of line; in LDS the versions specialised for short functions when they
   * apply */
  /* (the specialised versions answer -WERR_SERIAL when they met a rare exp / log argument:
   * their arithmetic has no branch for those, the general versions do) */
  if (chain == 0) {
    nm = -WERR_SERIAL;
    if (L::in_lds && n_other <= WAVE)
      nm = uniform_i(min_less_small_wave(other_prev, n_other, mlist, cap, sc, t - 1, pen_term));
    if (nm == -WERR_SERIAL) nm = min_less_wave(other_prev, n_other, mlist, cap, sc, t - 1, pen_term);
  } else if (t >= 2) {
    nm = -WERR_SERIAL;
    if (L::in_lds && n_other <= WAVE)
      nm = uniform_i(min_more_small_wave(other_prev, n_other, mlist, cap, sc, t - 1));
    if (nm == -WERR_SERIAL) nm = min_more_wave(other_prev, n_other, mlist, cap, sc, t - 1);
  }
  nm = uniform_i(nm); /* return values of out-of-line functions arrive in a VGPR */
  if (nm < 0) return nm;
  int n_new;
  if (t == 1) {
    if (chain == 0) {
      copy_list_wave(mlist, nm, own_new);
      n_new = nm;
    } else {
      copy_list_wave(own_prev, n_own, own_new);
      n_new = n_own;
    }
  } else {
    const L f1 = chain == 0 ? mlist : mlist.shifted(cap - nm);
    n_new = -WERR_SERIAL;
    if (L::in_lds && nm <= 32 && n_own <= 32)
      n_new = uniform_i(
          min_env_small_wave<HELP>(f1, nm, own_prev, n_own, own_new, cap, sc, chain));
    if (n_new == -WERR_SERIAL) {
      n_new = uniform_i(min_env_wave<HELP>(f1, nm, own_prev, n_own, own_new, cap, sc, chain));
    }
  }
  if (n_new < 0) return n_new;
  PSD_PROF_T0();
  wave_sync();
  bool ok = scale_add_store_wave(a, cur, own_new, n_new, fn_index, true, cum_weight_prev, w,
                                 (double)(-coverage) * w, 1 / cum_weight);
  wave_sync();
  PSD_PROF_ADD(PROF_SCALE);
  return ok ? n_new : -WERR_ARENA;
}

/* The same update for the usual case -- data point t >= 2, lists in LDS, n_other <= 16 (so
 * that the min-less / min-more result has at most 32 pieces) and n_own <= 32 -- with the
 * specialised operations inlined and not a single call: what the latency build runs for
 * nearly every data point. */
constexpr int FAST_MAX_OTHER = 16, FAST_MAX_OWN = 32;
template <bool HELP>
PSD_D int chain_step_fast(const DeviceArgs &a, ArenaCursor &cur, unsigned long long fn_index,
                          int chain, int t, const LdsList &other_prev, int n_other,
                          const LdsList &own_prev, int n_own, const LdsList &own_new,
                          const LdsList &mlist, const LdsScratch &sc, double pen_term,
                          double cum_weight_prev, double w, int coverage, double cum_weight) {
  PSD_ASSUME(n_other <= FAST_MAX_OTHER && n_own <= FAST_MAX_OWN);
  int nm;
  MathFast mth; /* exp / log without their rare-argument branches; one test at the end */
  if (chain == 0) {
    nm = min_less_impl<true>(other_prev, n_other, mlist, LDS_CAP, sc, t - 1, pen_term, mth);
  } else {
    nm = min_more_impl<true>(other_prev, n_other, mlist, LDS_CAP, sc, t - 1, mth);
  }
  /* (an error may itself be the consequence of a rare argument's unspecified value: the
   * general path decides) */
  if (nm < 0) return ballot(mth.rare != 0) ? -WERR_SERIAL : nm;
  if (nm > 32) return -WERR_OVERFLOW; /* cannot happen: at most 2 pieces per input piece */
  const LdsList f1 = chain == 0 ? mlist : mlist.shifted(LDS_CAP - nm);
  int n_new = min_env_impl<HELP, true>(f1, nm, own_prev, n_own, own_new, LDS_CAP, sc, chain, mth);
  if (ballot(mth.rare != 0)) return -WERR_SERIAL; /* the general path redoes the data point */
  if (n_new < 0) return n_new;
  PSD_PROF_T0();
  wave_sync();
  bool ok = scale_add_store_wave(a, cur, own_new, n_new, fn_index, true, cum_weight_prev, w,
                                 (double)(-coverage) * w, 1 / cum_weight);
  wave_sync();
  PSD_PROF_ADD(PROF_SCALE);
  return ok ? n_new : -WERR_ARENA;
}

/* Move one list between LDS and the problem's slot p of the HBM spill pool (cold: only when a
 * function outgrows LDS or has shrunk again). */
PSD_COLD_DEV void move_list_hbm(const DeviceArgs &a, int p, int id, int n, int to_hbm) {
  p = uniform_i(p);
  id = uniform_i(id);
  n = uniform_i(n);
  if (uniform_i(to_hbm)) {
    copy_list_across(lds_list(id), n, global_list(a, p, id));
  } else {
    copy_list_across(global_list(a, p, id), n, lds_list(id));
  }
}

/* The general LDS step as one out-of-line function (latency build: data point 1 and functions
 * longer than chain_step_fast takes). */
template <bool HELP>
PSD_COLD_DEV int chain_step_lds(const DeviceArgs &a, ArenaCursor &cur, unsigned long long fn_index,
                                int chain, int t, int id_other_prev, int n_other, int id_own_prev,
                                int n_own, int id_own_new, double pen_term, double cum_weight_prev,
                                double w, int coverage, double cum_weight) {
  chain = uniform_i(chain);
  t = uniform_i(t);
  LdsScratch lsc;
  lsc.w = chain;
  return chain_step<HELP>(a, cur, fn_index, chain, t, lds_list(uniform_i(id_other_prev)),
                          uniform_i(n_other), lds_list(uniform_i(id_own_prev)), uniform_i(n_own),
                          lds_list(uniform_i(id_own_new)), lds_list(4 + chain), lsc, LDS_CAP,
                          uniform_d(pen_term), uniform_d(cum_weight_prev), uniform_d(w),
                          uniform_i(coverage), uniform_d(cum_weight));
}

/* Data point 0 (cold, once per problem): C^down_1 = gamma_1 / w_1 (drv:266-270), stored
 * unscaled (drv:391); there is no up function yet.  Returns the piece count of the chain's
 * function or -WERR_ARENA. */
PSD_COLD_DEV int first_point(const DeviceArgs &a, ArenaCursor &cur, unsigned long long fn0,
                             int chain, int contig, int coverage, int id_own_new) {
  chain = uniform_i(chain);
  if (chain != 1) return 0;
  const LdsList own_new = lds_list(uniform_i(id_own_new));
  contig = uniform_i(contig);
  if (lane_id() == 0) {
    Coef c;
    c.Linear = 1.0;
    c.Log = (double)(-uniform_i(coverage));
    c.Constant = 0.0;
    store_piece(own_new, 0, c, a.contig_min_log_mean[contig], a.contig_max_log_mean[contig], -1,
                -5.0);
  }
  wave_sync();
  return arena_store_wave(a, cur, own_new, 1, fn0) ? 1 : -WERR_ARENA;
}

#ifndef PSD_HBM_HELPER /* A/B: -DPSD_HBM_HELPER=1 sends the spill path's larger roots to the helper waves */
#define PSD_HBM_HELPER 0
#endif
#ifdef PSD_HELPER_WAVES
constexpr bool HBM_HELP = PSD_HBM_HELPER != 0;
#else
constexpr bool HBM_HELP = false;
#endif
#if defined(PSD_HELPER_WAVES) && !defined(PSD_NO_HBM_COOP)
#define PSD_HBM_COOP 1
/* Lists in HBM, latency build: the chain wave and its helper wave share the chunks of the
 * three parallel phases of a step (fpop_wave.h, HOP_HBM_*): functions of adversarial data have
 * hundreds of pieces, i.e. more than one wave's worth of lanes of work per phase. */
PSD_COLD_DEV void helper_hbm_op(const DeviceArgs &a, int chain, int op) {
  chain = uniform_i(chain);
  op = uniform_i(op);
  Mail &m = g_sm.mail[chain];
  const int p = uniform_i(m.h_arg[0]);
  const GlobalScratch s = global_scratch(a, p, chain);
  if (op == HOP_HBM_COSTS) {
    LanePiece P;
    P.c.Linear = P.c.Log = P.c.Constant = 0.0;
    P.mn = P.mx = P.lc = P.rc = P.om = P.mu = P.muc = P.oc2 = 0.0;
    P.cls = CLS_STORE;
    MathFull mth;
    piece_costs_wave(global_list(a, p, uniform_i(m.h_arg[1])), uniform_i(m.h_arg[2]), s, P, mth, 1, 2);
  } else {
    const GlobalList f1 = global_list(a, p, uniform_i(m.h_arg[1])).shifted(uniform_i(m.h_arg[2]));
    const int n1 = uniform_i(m.h_arg[3]);
    const GlobalList f2 = global_list(a, p, uniform_i(m.h_arg[4]));
    const int n2 = uniform_i(m.h_arg[5]);
    if (op == HOP_HBM_TABLE) {
      const ldouble *staged = nullptr;
      if (n1 + n2 <= COOP_STAGE_DOUBLES) { /* the ends of f1 behind the chain wave's copy of f2's */
        ldouble *dst = coop_stage(chain) + n2;
        coop_stage_ends(f1, n1, dst);
        staged = dst;
      }
      env_table_second(f1, n1, f2, n2, s, staged);
    } else if (op == HOP_HBM_CLASSIFY) {
      env_coop_helper(f1, n1, f2, n2, s, uniform_i(m.h_arg[6]), chain);
    }
  }
}
PSD_NOINLINE int min_less_coop_wave(GlobalList in, int n, GlobalList out, int cap, GlobalScratch s,
                                    int data_i_out, double add_const, int chain, int p, int id) {
  MathFull mth;
  return min_less_impl<false, true>(in, n, out, cap, s, data_i_out, add_const, mth, chain, p, id);
}
PSD_NOINLINE int min_more_coop_wave(GlobalList in, int n, GlobalList out, int cap, GlobalScratch s,
                                    int data_i_out, int chain, int p, int id) {
  MathFull mth;
  return min_more_impl<false, true>(in, n, out, cap, s, data_i_out, mth, chain, p, id);
}
PSD_NOINLINE int min_env_coop_wave(GlobalList f1, int n1, GlobalList f2, int n2, GlobalList out,
                                   int cap, GlobalScratch s, int chain, int p, int id1, int off1,
                                   int id2) {
  return min_env_coop(f1, n1, f2, n2, out, cap, s, chain, p, id1, off1, id2);
}
PSD_COLD_DEV int chain_step_hbm(const DeviceArgs &a, ArenaCursor &cur,
                                unsigned long long fn_index, int p, int chain, int t,
                                int id_other_prev, int n_other, int id_own_prev, int n_own,
                                int id_own_new, double pen_term, double cum_weight_prev, double w,
                                int coverage, double cum_weight) {
  p = uniform_i(p);
  chain = uniform_i(chain);
  t = uniform_i(t);
  id_other_prev = uniform_i(id_other_prev);
  id_own_prev = uniform_i(id_own_prev);
  n_other = uniform_i(n_other);
  n_own = uniform_i(n_own);
  pen_term = uniform_d(pen_term);
  const int cap = a.spill_cap;
  const GlobalList other_prev = global_list(a, p, id_other_prev);
  const GlobalList own_prev = global_list(a, p, id_own_prev);
  const GlobalList own_new = global_list(a, p, uniform_i(id_own_new));
  const GlobalList mlist = global_list(a, p, 4 + chain);
  const GlobalScratch sc = global_scratch(a, p, chain);
  int nm = 0;
  if (chain == 0) {
    nm = min_less_coop_wave(other_prev, n_other, mlist, cap, sc, t - 1, pen_term, chain, p,
                            id_other_prev);
  } else if (t >= 2) {
    nm = min_more_coop_wave(other_prev, n_other, mlist, cap, sc, t - 1, chain, p, id_other_prev);
  }
  nm = uniform_i(nm);
  if (nm < 0) return nm;
  int n_new;
  if (t == 1) {
    if (chain == 0) {
      copy_list_wave(mlist, nm, own_new);
      n_new = nm;
    } else {
      copy_list_wave(own_prev, n_own, own_new);
      n_new = n_own;
    }
  } else {
    const int off1 = chain == 0 ? 0 : cap - nm;
    n_new = uniform_i(min_env_coop_wave(mlist.shifted(off1), nm, own_prev, n_own, own_new, cap, sc,
                                        chain, p, 4 + chain, off1, id_own_prev));
  }
  if (n_new < 0) return n_new;
  wave_sync();
  bool ok = scale_add_store_wave(a, cur, own_new, n_new, fn_index, true, uniform_d(cum_weight_prev),
                                 uniform_d(w), (double)(-uniform_i(coverage)) * uniform_d(w),
                                 1 / uniform_d(cum_weight));
  wave_sync();
  return ok ? n_new : -WERR_ARENA;
}
#else
/* The same step with every list in the HBM spill area (functions that outgrew LDS): a cold,
 * out-of-line function, so that its addressing does not hold registers in the kernel's loop. */
PSD_COLD_DEV int chain_step_hbm(const DeviceArgs &a, ArenaCursor &cur,
                                unsigned long long fn_index, int p, int chain, int t,
                                int id_other_prev, int n_other, int id_own_prev, int n_own,
                                int id_own_new, double pen_term, double cum_weight_prev, double w,
                                int coverage, double cum_weight) {
  p = uniform_i(p);
  chain = uniform_i(chain);
  t = uniform_i(t);
  return chain_step<HBM_HELP>(a, cur, fn_index, chain, t,
                           global_list(a, p, uniform_i(id_other_prev)), uniform_i(n_other),
                           global_list(a, p, uniform_i(id_own_prev)), uniform_i(n_own),
                           global_list(a, p, uniform_i(id_own_new)), global_list(a, p, 4 + chain),
                           global_scratch(a, p, chain), a.spill_cap, uniform_d(pen_term),
                           uniform_d(cum_weight_prev), uniform_d(w), uniform_i(coverage),
                           uniform_d(cum_weight));
}
#endif

/* list ids: 2*chain + buffer for the two cost functions (chain 0 = up, 1 = down), 4 + chain
 * for the chain's min-less / min-more temporary.  Lists live in LDS (g_sm.list[id]) while
 * every function has at most LDS_CAP pieces; when an operation overflows, the step is redone
 * with all lists in the HBM spill area, and the problem returns to LDS once both functions
 * have shrunk below LDS_CAP/2. */
/* State of the decoding (drv:399-442) between two calls of backtrack_wave. */
struct BtState {
  double best_log_mean, prev_log_mean;
  int prev_seg_end, prev_seg_offset; /* offset 0: the next function is an up one, N: a down one */
  int n_seg, n_eq, status;
};

/* Decode the optimal segmentation (drv:399-442): one wave, right after the forward pass of
 * its problem (the arena records it follows were written by this workgroup; fast problems
 * decode while slower ones are still in their forward pass).  Follows the chain of segment
 * ends while it stays at data points >= t_lo (0: to the end); the record of function
 * (up/down, t) is fn_ref[fn_up/fn_down + t - t_origin].  Returns with bt.prev_seg_end < t_lo. */
PSD_D void backtrack_wave(const DeviceArgs &a, int p, int N, BtState &bt, unsigned long long fn_up,
                          unsigned long long fn_down, int t_origin, int t_lo) {
  const int lane = lane_id();
  int *seg_start = a.seg_start + a.prob_seg_off[p];
  double *seg_mean = a.seg_mean + a.prob_seg_off[p];
  double best_log_mean = bt.best_log_mean;
  double prev_log_mean = bt.prev_log_mean;
  int prev_seg_end = bt.prev_seg_end;
  int prev_seg_offset = bt.prev_seg_offset;
  int n_seg = bt.n_seg, n_eq = bt.n_eq, status = bt.status;
  while (t_lo <= prev_seg_end && status == 0) {
    if (n_seg >= N) { /* more segments than data points: cannot happen for a valid store */
      status = PST_BACKTRACK;
      break;
    }
    unsigned long long ref = a.fn_ref[(prev_seg_offset ? fn_down : fn_up) +
                                      (unsigned long long)(prev_seg_end - t_origin)];
    unsigned long long off = ref >> FN_COUNT_BITS;
    int n = (int)(ref & ((1ull << FN_COUNT_BITS) - 1));
    if (lane == 0) {
      seg_start[n_seg] = prev_seg_end;
      seg_mean[n_seg] = d_exp(best_log_mean);
    }
    n_seg++;
    prev_seg_offset = prev_seg_offset == 0 ? N : 0;
    if (prev_log_mean != PSD_INF) {
      best_log_mean = prev_log_mean; /* equality constraint inactive */
    } else {
      n_eq++;
    }
    /* findMean (fpl:643-653) on the restored function (drv:44-54): piece k spans
     * [max_{k-1}, max_k], the first from -Inf; the first match wins. */
    bool found = false;
    const ArenaPtr rec = arena_at(a, off); /* (a record never straddles a block) */
    for (int base = 0; base < n; base += WAVE) {
      int k = base + lane;
      bool hit = false;
      int di = 0;
      double prv = 0.0;
      if (k < n) {
        double mxk = rec.mx[k];
        double mnk = k == 0 ? -PSD_INF : rec.mx[k - 1];
        di = rec.di[k];
        prv = rec.prv[k];
        hit = mnk <= best_log_mean && best_log_mean <= mxk;
      }
      unsigned long long m = ballot(hit);
      if (m) {
        int src = ctz64(m);
        prev_seg_end = rdlane_i(di, src);
        prev_log_mean = rdlane_d(prv, src);
        found = true;
        break;
      }
    }
    if (!found) {
      status = PST_BACKTRACK;
      break;
    }
  }
  if (status == 0 && prev_seg_end < 0) { /* the first segment (drv:442) */
    if (lane == 0) {
      seg_start[n_seg] = -1;
      seg_mean[n_seg] = d_exp(best_log_mean);
    }
    n_seg++;
  }
  bt.best_log_mean = best_log_mean;
  bt.prev_log_mean = prev_log_mean;
  bt.prev_seg_end = prev_seg_end;
  bt.prev_seg_offset = prev_seg_offset;
  bt.n_seg = n_seg;
  bt.n_eq = n_eq;
  bt.status = status;
}

#ifdef PSD_HELPER_WAVES
constexpr bool USE_HELPER = true;
constexpr int FORWARD_THREADS = 256; /* waves 0,1: the two chains; waves 2,3: their helpers */
/* every workgroup barrier of a main wave is mirrored by its helper */
PSD_D void block_sync(int chain) {
  if (!mail_wait(chain)) {
    if (lane_id() == 0) g_sm.mail[chain].abort = 1;
  }
  mail_post(chain, HOP_BARRIER);
  __syncthreads();
}
#else
constexpr bool USE_HELPER = false;
constexpr int FORWARD_THREADS = 128;
PSD_D void block_sync(int) { __syncthreads(); }
#endif

/* the workgroup barrier as a call (cold paths inside the kernel's loop) */
PSD_COLD_DEV void block_sync_cold(int chain) { block_sync(uniform_i(chain)); }

/* The barrier at the end of every data point, between the two chain waves only.
 * PSD_FLAG_BARRIER (latency build): each wave publishes the barrier's number in LDS and polls
 * the other's -- no s_barrier, and the helper waves, which never touch the lists, stay out of
 * it (waking them through their mailbox for every data point cost more than the data point's
 * imbalance).  Returns false if the other wave never came (never expected: the caller
 * aborts the problem).  Otherwise the workgroup barrier. */
PSD_D bool step_sync(int chain, unsigned seq) {
#ifdef PSD_FLAG_BARRIER
  constexpr int SPIN_LIMIT = WAIT_SPIN_LIMIT; /* seconds */
  wave_sync();
  if (lane_id() == 0) flag_store((int *)&g_sm.arrived[chain], (int)seq);
  for (int spin = 0; spin < SPIN_LIMIT; spin++) {
    /* lane 0's reading decides for the wave */
    if (rdlane_i(flag_load((int *)&g_sm.arrived[1 - chain]), 0) - (int)seq >= 0) {
      PSD_SPIN_NOTE(spin);
      return true;
    }
    spin_pause();
  }
  return false;
#else
  (void)seq;
  block_sync(chain);
  return true;
#endif
}

/* Take a slot of the HBM spill pool for this workgroup's problem (cold: at most once per
 * problem).  Both chain waves call it; returns the slot, or -1 when the pool is exhausted. */
PSD_COLD_DEV int take_spill_slot(const DeviceArgs &a, int chain) {
  chain = uniform_i(chain);
  if (chain == 0 && lane_id() == 0) {
    int sl = atomicAdd(a.spill_next, 1);
    g_sm.spill_slot = sl < a.spill_slots ? sl : -1;
  }
  block_sync(chain);
  return uniform_i(g_sm.spill_slot);
}


/* ---- checkpointed store ----------------------------------------
 * Checkpoint slot k of a problem holds the two live functions after data point (k+1) K: per
 * slot 6 + 12 cap doubles {cum_weight, -, overflow offsets of the two chains, interval totals
 * of the two chains (bit patterns), then per chain Lin, Log, Con, mn, mx, prv} and 8 + 2 cap
 * ints {n_up, n_down, data point, max intervals of the two chains, spill steps, sequential
 * envelope replays of the two chains, then per chain data_i}.  The full store keeps ONE such slot per problem: the park slot, written when
 * the arena runs out (park_state) and read back when the problem is resumed.  A function with more than cap pieces lives in the overflow pool (6 n doubles from
 * 6 off, n ints from off) and the slot only holds its offset. */
constexpr int CKPT_HDR_F64 = 6, CKPT_HDR_I32 = 8;
PSD_D size_t ckpt_f64_at(const DeviceArgs &a, long long slot) {
  return (size_t)slot * (CKPT_HDR_F64 + 12 * (size_t)a.ckpt_cap);
}
PSD_D size_t ckpt_i32_at(const DeviceArgs &a, long long slot) {
  return (size_t)slot * (CKPT_HDR_I32 + 2 * (size_t)a.ckpt_cap);
}
PSD_D GlobalList ckpt_list(const DeviceArgs &a, long long slot, int chain) {
  const size_t cap = (size_t)a.ckpt_cap;
  gdouble *f = (gdouble *)(a.ckpt_f64 + ckpt_f64_at(a, slot) + CKPT_HDR_F64 + (size_t)chain * 6 * cap);
  GlobalList r;
  r.Lin_ = f;
  r.Log_ = f + cap;
  r.Con_ = f + 2 * cap;
  r.mn_ = f + 3 * cap;
  r.mx_ = f + 4 * cap;
  r.prv_ = f + 5 * cap;
  r.di_ = (gint *)(a.ckpt_i32 + ckpt_i32_at(a, slot) + CKPT_HDR_I32 + (size_t)chain * cap);
  return r;
}
/* n pieces of the overflow pool from piece offset off */
PSD_D GlobalList ckpt_overflow_list(const DeviceArgs &a, unsigned long long off, int n) {
  gdouble *f = (gdouble *)(a.ckpt_ovf_f64 + (size_t)off * 6);
  const size_t m = (size_t)n;
  GlobalList r;
  r.Lin_ = f;
  r.Log_ = f + m;
  r.Con_ = f + 2 * m;
  r.mn_ = f + 3 * m;
  r.mx_ = f + 4 * m;
  r.prv_ = f + 5 * m;
  r.di_ = (gint *)(a.ckpt_ovf_i32 + (size_t)off);
  return r;
}
/* Room in the overflow pool for the functions of this checkpoint that exceed ckpt_cap (cold:
 * adversarial data only).  Both chain waves call it with the same counts; chain 0 takes the
 * room with one atomic and publishes it.  Returns the offset of the first such function (the
 * up function's if it is one), ~0 when the pool is exhausted -- in both waves alike. */
PSD_COLD_DEV unsigned long long ckpt_take_overflow(const DeviceArgs &a, int chain, int n_up,
                                                   int n_down) {
  chain = uniform_i(chain);
  n_up = uniform_i(n_up);
  n_down = uniform_i(n_down);
  const unsigned long long want = (unsigned long long)(n_up > a.ckpt_cap ? n_up : 0) +
                                  (unsigned long long)(n_down > a.ckpt_cap ? n_down : 0);
  if (chain == 0 && lane_id() == 0) {
    unsigned long long off = atomicAdd(a.ckpt_ovf_next, want);
    g_sm.ckpt_ovf = off + want <= a.ckpt_ovf_cap ? off : ~0ull;
  }
  block_sync(chain);
  const unsigned long long off = psd_d2u(uniform_d(psd_u2d(g_sm.ckpt_ovf)));
  block_sync(chain); /* the word is free again before anyone can come back here */
  return off;
}
/* this chain's function (list `id`, n pieces, in LDS or in the problem's slot of the HBM spill
 * pool) and the cumulated weight -> checkpoint k; ovf: this chain's room in the overflow pool
 * when n > ckpt_cap */
PSD_COLD_DEV void ckpt_save(const DeviceArgs &a, int p, int k, int chain, int id, int n,
                            double cum_weight, int in_hbm, int spill_slot, unsigned long long ovf) {
  p = uniform_i(p);
  k = uniform_i(k);
  chain = uniform_i(chain);
  n = uniform_i(n);
  id = uniform_i(id);
  spill_slot = uniform_i(spill_slot);
  ovf = psd_d2u(uniform_d(psd_u2d(ovf)));
  const long long slot = a.prob_ckpt_off[p] + k;
  const GlobalList dst = n > a.ckpt_cap ? ckpt_overflow_list(a, ovf, n) : ckpt_list(a, slot, chain);
  if (uniform_i(in_hbm)) {
    copy_list_across(global_list(a, spill_slot, id), n, dst);
  } else {
    copy_list_across(lds_list(id), n, dst);
  }
  if (lane_id() == 0) {
    a.ckpt_i32[ckpt_i32_at(a, slot) + (size_t)chain] = n;
    a.ckpt_f64[ckpt_f64_at(a, slot) + 2 + (size_t)chain] = psd_u2d(ovf);
    if (chain == 0) a.ckpt_f64[ckpt_f64_at(a, slot)] = uniform_d(cum_weight);
  }
}
/* piece count of a chain's function in checkpoint k */
PSD_COLD_DEV int ckpt_count(const DeviceArgs &a, int p, int k, int chain) {
  const long long slot = a.prob_ckpt_off[uniform_i(p)] + uniform_i(k);
  return uniform_i(a.ckpt_i32[ckpt_i32_at(a, slot) + (size_t)uniform_i(chain)]);
}
/* checkpoint k -> list `id` (LDS, or the problem's spill slot when to_hbm); returns the piece
 * count */
PSD_COLD_DEV int ckpt_load(const DeviceArgs &a, int p, int k, int chain, int id, int to_hbm,
                           int spill_slot) {
  p = uniform_i(p);
  k = uniform_i(k);
  chain = uniform_i(chain);
  id = uniform_i(id);
  spill_slot = uniform_i(spill_slot);
  const long long slot = a.prob_ckpt_off[p] + k;
  const int n = uniform_i(a.ckpt_i32[ckpt_i32_at(a, slot) + (size_t)chain]);
  const unsigned long long ovf =
      psd_d2u(uniform_d(a.ckpt_f64[ckpt_f64_at(a, slot) + 2 + (size_t)chain]));
  const GlobalList src = n > a.ckpt_cap ? ckpt_overflow_list(a, ovf, n) : ckpt_list(a, slot, chain);
  if (uniform_i(to_hbm)) {
    copy_list_across(src, n, global_list(a, spill_slot, id));
  } else {
    copy_list_across(src, n, lds_list(id));
  }
  return n;
}
PSD_COLD_DEV double ckpt_cum_weight(const DeviceArgs &a, int p, int k) {
  const long long slot = a.prob_ckpt_off[uniform_i(p)] + uniform_i(k);
  return a.ckpt_f64[ckpt_f64_at(a, slot)];
}
/* what a parked problem needs besides its two functions (slot 0 of the problem) */
PSD_COLD_DEV void park_counters_save(const DeviceArgs &a, int p, int chain, int t,
                                     unsigned long long total_intervals, int max_intervals,
                                     int spill_steps) {
  const long long slot = a.prob_ckpt_off[uniform_i(p)];
  chain = uniform_i(chain);
  if (lane_id() == 0) {
    a.ckpt_f64[ckpt_f64_at(a, slot) + 4 + (size_t)chain] = psd_u2d(total_intervals);
    a.ckpt_i32[ckpt_i32_at(a, slot) + 3 + (size_t)chain] = max_intervals;
    a.ckpt_i32[ckpt_i32_at(a, slot) + 6 + (size_t)chain] = g_sm.serial[chain];
    if (chain == 1) {
      a.ckpt_i32[ckpt_i32_at(a, slot) + 2] = t;
      a.ckpt_i32[ckpt_i32_at(a, slot) + 5] = spill_steps;
    }
  }
}
PSD_COLD_DEV unsigned long long park_total_intervals(const DeviceArgs &a, int p, int chain) {
  const long long slot = a.prob_ckpt_off[uniform_i(p)];
  return psd_d2u(uniform_d(a.ckpt_f64[ckpt_f64_at(a, slot) + 4 + (size_t)uniform_i(chain)]));
}
PSD_COLD_DEV int park_int(const DeviceArgs &a, int p, int which) {
  const long long slot = a.prob_ckpt_off[uniform_i(p)];
  return uniform_i(a.ckpt_i32[ckpt_i32_at(a, slot) + (size_t)uniform_i(which)]);
}

/* PSD_KERNEL_WAVES_PER_EU (throughput build): keep the kernel's own register use within the
 * budget of that many waves per SIMD, as the out-of-line operations already are */
#undef PSD_KERNEL_OCC
#if defined(PSD_KERNEL_WAVES_PER_EU) && !defined(PSD_EMU)
#define PSD_KERNEL_OCC \
  __attribute__((amdgpu_waves_per_eu(PSD_KERNEL_WAVES_PER_EU, PSD_KERNEL_WAVES_PER_EU)))
#else
#define PSD_KERNEL_OCC
#endif
/* The kernel body.  CKPT = false: the full store, one pass (everything about passes and
 * checkpoints below folds away: the loop is the one the latency numbers were tuned on);
 * CKPT = true: the checkpointed store's passes. */
template <bool CKPT>
PSD_D void forward_body(const DeviceArgs &a) {
  const int p = a.prob_order[blockIdx.x];
  const int chain = uniform_i(wave_id()) & 1; /* uniform per wave: say so (scalar branches) */
  const int lane = lane_id();
  const int contig = a.prob_contig[p];
  const int N = a.contig_n[contig];
  const double penalty = a.prob_penalty[p];
  const int *count = a.count + a.contig_off[contig];
  const int *weight = a.weight + a.contig_off[contig];
  const LdsList mlist = lds_list(4 + chain);
  LdsScratch lsc;
  lsc.w = chain;

  psd_tables_init(); /* exp/log tables -> LDS */
  if (threadIdx.x == 0) {
    if (a.started) atomicAdd_system(a.started, 1);
    g_sm.abort_status[0] = g_sm.abort_status[1] = g_sm.abort_status[2] = 0;
    g_sm.abort_err[0] = g_sm.abort_err[1] = g_sm.abort_err[2] = 0;
    for (int i = 0; i < 6; i++) g_sm.n[i] = 0;
    g_sm.serial[0] = g_sm.serial[1] = 0;
    g_sm.total_up = 0; /* (read by the down wave's result even when the pass stops early) */
    g_sm.max_up = 0;
    g_sm.arrived[0] = g_sm.arrived[1] = 0xffffffffu;
#ifdef PSD_SPIN_STATS
    g_sm.spin_max[0] = g_sm.spin_max[1] = g_sm.spin_max[2] = g_sm.spin_max[3] = 0;
#endif
#ifdef PSD_PROFILE
    for (int i = 0; i < N_PROF; i++)
      g_sm.prof[0][i] = g_sm.prof[1][i] = g_sm.prof[2][i] = g_sm.prof[3][i] = 0;
#endif
#ifdef PSD_HELPER_WAVES
    for (int c = 0; c < 2; c++) {
      g_sm.mail[c].seq_cmd = g_sm.mail[c].seq_done = 0;
      g_sm.mail[c].op = 0;
      g_sm.mail[c].abort = 0;
    }
#endif
  }
  __syncthreads();
#ifdef PSD_HELPER_WAVES
  if (wave_id() >= 2) {
    helper_loop(chain, a);
    return;
  }
#endif

  /* Passes over the data.  Full store (ckpt_interval = 0): one pass, every function recorded,
   * then the decoding.  Checkpointed store: a forward pass that records nothing but a
   * checkpoint every K data points, then, led by the decoding, one pass per block of K data
   * points that holds a segment end, from that block's checkpoint and with the records kept
   * in this wave's region of the arena. */
  const int K = CKPT ? a.ckpt_interval : 0;
  const unsigned long long fn_stride = K > 0 ? (unsigned long long)K + 1ull : (unsigned long long)N;
  const unsigned long long fn_up = (unsigned long long)a.prob_fn_off[p];
  const unsigned long long fn_down = fn_up + fn_stride;
  const unsigned long long fn_mine = chain == 1 ? fn_down : fn_up;
  ArenaCursor cur;
  cursor_clear(cur, K > 0 ? 0 : 1);
  unsigned long long total_intervals = 0;
  int max_intervals = 0;
  int spill_steps = 0;
  int status = 0;
  double cum_weight_i = 0.0, cum_weight_prev_i = -1.0;
  int cnt_reg = 0, wt_reg = 0;
  int b = 0;        /* buffer holding step t-1 */
  bool in_hbm = false; /* where the lists of step t-1 live */
  int spill_slot = -1; /* this problem's slot of the HBM spill pool, taken on first overflow */
  int t = 0;
  int t_lo = 0, t_hi = N; /* data points of this pass */
  int t_first = 0;        /* where this pass starts: t_lo, or the data point a parked problem resumes at */
  int t_origin = 0;       /* fn_ref index of data point t: t - t_origin */
  int next_ckpt = K > 0 ? K : -1;
  bool forward = true;    /* the first pass */
  int step_reached = 0;
  int parked = 0;
  BtState bt;
  bt.best_log_mean = bt.prev_log_mean = 0.0;
  bt.prev_seg_end = -1;
  bt.prev_seg_offset = 0;
  bt.n_seg = bt.n_eq = bt.status = 0;
  ProbResult r;
  r.best_cost = 0.0;
  r.best_log_mean = 0.0;
  r.prev_log_mean = 0.0;
  r.prev_seg_end = -1;
  unsigned sync_no = 0; /* parity slot of the abort flags: one per barrier */
  if (lane == 0) g_sm.t_begin[chain] = cycle_now(); /* (in LDS: read once, at the end) */
  bool resumed_abort = false;
  if (!CKPT && a.prob_resume != nullptr) {
    /* A problem parked by an earlier launch (the arena had run out): its two functions, the
     * cumulated weight and its counters come back from the park slot and the pass goes on at
     * the data point it had reached. */
    const int t_resume = uniform_i(a.prob_resume[p]);
    if (t_resume > 0) {
      device_fence();
      in_hbm = uniform_i(ckpt_count(*a.self, p, 0, 0)) > LDS_CAP ||
               uniform_i(ckpt_count(*a.self, p, 0, 1)) > LDS_CAP;
      if (in_hbm) {
        spill_slot = uniform_i(take_spill_slot(*a.self, chain));
        if (spill_slot < 0) {
          /* (the pool has no slot yet for these long functions: the park slot stays as it is
           * and the problem resumes here again once the host has enlarged the pool) */
          status = PST_SPILL_FULL;
          resumed_abort = true;
          parked = 1;
          step_reached = t_resume;
        }
      }
      if (!resumed_abort) {
        const int n_ck = uniform_i(ckpt_load(*a.self, p, 0, chain, 2 * chain, in_hbm ? 1 : 0,
                                             spill_slot));
        if (lane == 0) g_sm.n[2 * chain] = n_ck;
        cum_weight_i = uniform_d(ckpt_cum_weight(*a.self, p, 0));
        cum_weight_prev_i = cum_weight_i;
        total_intervals = park_total_intervals(*a.self, p, chain);
        max_intervals = park_int(*a.self, p, 3 + chain);
        spill_steps = park_int(*a.self, p, 5);
        {
          const int n_serial = park_int(*a.self, p, 6 + chain);
          if (lane == 0) g_sm.serial[chain] = n_serial;
        }
        t_first = t_resume; /* (t_lo stays 0: the decoding goes back to the first data point) */
        block_sync(chain); /* both functions are in place before either chain reads the other's */
      }
    }
  }
  for (;;) { /* passes */
  if (resumed_abort) break;
  for (t = t_first; t < t_hi; t++) {
    PSD_PROF_T0();
    if (!CKPT) cur.store = 1; /* known to the compiler even after a cold call returned the cursor */
    if ((t & 63) == 0 || t == t_first) { /* coalesced read of the next 64 data points */
      int tt = (t & ~63) + lane;
      cnt_reg = tt < N ? count[tt] : 0;
      wt_reg = tt < N ? weight[tt] : 0;
    }
    const int coverage = rdlane_i(cnt_reg, t & 63);
    const double w = (double)rdlane_i(wt_reg, t & 63);
    const double cum_weight_new = cum_weight_i + w;
    const int nb = b ^ 1;
    const int id_own_prev = 2 * chain + b, id_own_new = 2 * chain + nb;
    const int id_other_prev = 2 * (1 - chain) + b;
    const int n_own = uniform_i(g_sm.n[id_own_prev]);
    const int n_other = uniform_i(g_sm.n[id_other_prev]);
    const unsigned long long fn_index = fn_mine + (unsigned long long)(t - t_origin);
    /* come back from HBM when both functions fit comfortably again */
    if (in_hbm && n_own <= LDS_CAP / 2 && n_other <= LDS_CAP / 2) {
      move_list_hbm(*a.self, spill_slot, id_own_prev, n_own, 0);
      in_hbm = false;
      block_sync(chain);
    }
    int n_new = 0;
    for (;;) { /* at most two passes: LDS, then HBM after an overflow */
      if (t == 0) {
        ArenaCursor cur0 = cur; /* only this copy has its address taken */
        n_new = uniform_i(first_point(*a.self, cur0, fn_index, chain, contig, coverage, id_own_new));
        cur = cur0;
      } else if (!in_hbm) {
#ifdef PSD_CALL_LDS_OPS /* throughput build: operations out of line (register budget) */
        n_new = chain_step<USE_HELPER>(a, cur, fn_index, chain, t,
                           lds_list(id_other_prev), n_other, lds_list(id_own_prev),
                           n_own, lds_list(id_own_new), mlist, lsc, LDS_CAP,
                           penalty / cum_weight_prev_i, cum_weight_prev_i, w, coverage,
                           cum_weight_new);
#else
        n_new = -WERR_SERIAL;
        if (t >= 2 && n_other <= FAST_MAX_OTHER && n_own <= FAST_MAX_OWN) {
          n_new = chain_step_fast<USE_HELPER>(
              a, cur, fn_index, chain, t, lds_list(id_other_prev), n_other,
              lds_list(id_own_prev), n_own, lds_list(id_own_new), mlist, lsc,
              penalty / cum_weight_prev_i, cum_weight_prev_i, w, coverage, cum_weight_new);
        }
        if (n_new == -WERR_SERIAL) { /* not the usual case, or it needs the sequential replay */
          ArenaCursor cur_gen = cur; /* only this copy has its address taken */
          n_new = uniform_i(chain_step_lds<USE_HELPER>(
              *a.self, cur_gen, fn_index, chain, t, id_other_prev, n_other,
              id_own_prev, n_own, id_own_new, penalty / cum_weight_prev_i, cum_weight_prev_i, w,
              coverage, cum_weight_new));
          cur = cur_gen;
        }
#endif
      } else {
        ArenaCursor cur_hbm = cur; /* only this copy has its address taken */
        n_new = uniform_i(chain_step_hbm(*a.self, cur_hbm, fn_index, spill_slot, chain, t,
                               id_other_prev, n_other, id_own_prev, n_own, id_own_new,
                               penalty / cum_weight_prev_i, cum_weight_prev_i, w, coverage,
                               cum_weight_new));
        cur = cur_hbm;
        /* With the lists in HBM the two waves first meet at the workgroup barrier, so that the
         * flag barrier below finds the other wave there already: a wave that polled LDS while
         * the other one worked through lists in HBM with FLAT instructions (issued to the LDS
         * pipeline too) cost half as much again per data point (config 5, 1e5 data points:
         * 7.6 s -> 11.4 s; profiles/r02/ab_step_barrier.log).  The lists are reached with
         * global_* instructions now and polling is harmless; the barrier stays as a guard.
         * in_hbm is the same in both waves. */
        block_sync_cold(chain);
      }
      /* ---- end of pass: report, store the backtrack record, meet the other wave ---- */
      PSD_PROF_T0();
      const unsigned slot = sync_no % 3u;
      if (lane == 0) g_sm.n[id_own_new] = n_new < 0 ? 0 : n_new;
      if (n_new < 0) {
        if (lane == 0) {
          g_sm.abort_err[slot] = -n_new;
          g_sm.abort_status[slot] = ((-n_new) & WERR_ARENA)      ? PST_ARENA_FULL
                                    : ((-n_new) & WERR_OVERFLOW) ? PST_LDS_OVERFLOW
                                                                 : PST_REF_THROW;
        }
      }
      PSD_PROF_ADD(PROF_ARENA);
      if (!step_sync(chain, sync_no)) {
        status = PST_REF_THROW;
        if (lane == 0) g_sm.abort_err[slot] = WERR_HELPER;
        break;
      }
      PSD_PROF_ADD(PROF_BARRIER);
      status = uniform_i(g_sm.abort_status[slot]);
      /* three rotating slots: the one cleared here is first written two barriers later */
      if (lane == 0) g_sm.abort_status[(sync_no + 2u) % 3u] = 0;
      sync_no++;
#ifdef PSD_PARK_ON_LDS_OVERFLOW
      /* the packed build: a function that outgrows its short lists is for the throughput build
       * (the problem is parked below and resumed there), not for the HBM path */
      if (status == PST_LDS_OVERFLOW && !CKPT && a.prob_resume != nullptr && t > 0) break;
#endif
      if (status == PST_LDS_OVERFLOW && !in_hbm && a.spill_cap > LDS_CAP && t > 0) {
        /* redo this data point with the lists in HBM: every wave moves its own t-1 list */
        if (spill_slot < 0) {
          spill_slot = uniform_i(take_spill_slot(*a.self, chain)); /* a call's result: say it is uniform */
          if (spill_slot < 0) {
            status = PST_SPILL_FULL;
            break;
          }
        }
        move_list_hbm(*a.self, spill_slot, id_own_prev, n_own, 1);
        in_hbm = true;
        status = 0;
        block_sync(chain);
        continue;
      }
      break;
    }
    if (status != 0) {
      /* (no spill slot: this data point's inputs are still in LDS -- the pool was needed for
       * its result -- so the problem parks like one that ran out of arena and goes on at this
       * data point when the host has enlarged the pool, instead of starting over) */
#ifdef PSD_PARK_ON_LDS_OVERFLOW
      const bool park_now = status == PST_ARENA_FULL || status == PST_SPILL_FULL ||
                            status == PST_LDS_OVERFLOW;
#else
      const bool park_now = status == PST_ARENA_FULL || status == PST_SPILL_FULL;
#endif
      if (!CKPT && park_now && a.prob_resume != nullptr && t > 0) {
        /* Out of arena (or of spill slots): park.  The functions of data point t-1 (this step's inputs, untouched)
         * go to the park slot; the host adds arena blocks and resumes the problem at t.  Both
         * waves see the status, so both come here.  (Functions too long for a slot need the
         * overflow pool; without room there the problem is simply rerun from the start.) */
        const int n_up = uniform_i(g_sm.n[b]), n_down = uniform_i(g_sm.n[2 + b]);
        unsigned long long ovf = 0;
        bool can_park = true;
        if (n_up > a.ckpt_cap || n_down > a.ckpt_cap) {
          ovf = psd_d2u(uniform_d(psd_u2d(ckpt_take_overflow(*a.self, chain, n_up, n_down))));
          if (ovf == ~0ull) can_park = false;
          if (can_park && chain == 1 && n_up > a.ckpt_cap) ovf += (unsigned long long)n_up;
        }
        if (can_park) {
          ckpt_save(*a.self, p, 0, chain, 2 * chain + b, chain == 0 ? n_up : n_down, cum_weight_i,
                    in_hbm ? 1 : 0, spill_slot, ovf);
          park_counters_save(*a.self, p, chain, t, total_intervals, max_intervals, spill_steps);
          parked = 1;
        }
      }
      break;
    }
    if (!CKPT || forward) {
      total_intervals += (unsigned long long)n_new;
      if (max_intervals < n_new) max_intervals = n_new;
      if (in_hbm) spill_steps++;
    }
    cum_weight_i = cum_weight_new;
    cum_weight_prev_i = cum_weight_i;
    b = nb;
    if (CKPT && t == next_ckpt) { /* forward pass: keep the two live functions */
      /* both counts are visible to both waves since the barrier of this data point */
      const int n_up = uniform_i(g_sm.n[b]), n_down = uniform_i(g_sm.n[2 + b]);
      unsigned long long ovf = 0;
      if (n_up > a.ckpt_cap || n_down > a.ckpt_cap) { /* beyond a slot: the overflow pool */
        ovf = psd_d2u(uniform_d(psd_u2d(ckpt_take_overflow(*a.self, chain, n_up, n_down))));
        if (ovf == ~0ull) {
          status = PST_CKPT_FULL; /* in both waves */
          break;
        }
        if (chain == 1 && n_up > a.ckpt_cap) ovf += (unsigned long long)n_up;
      }
      ckpt_save(*a.self, p, t / K - 1, chain, 2 * chain + b, n_new, cum_weight_i, in_hbm ? 1 : 0,
                spill_slot, ovf);
      next_ckpt += K;
    }
  }
  if (!CKPT || forward) step_reached = t;
  if (status != 0) break;
  /* ---- after a pass ---- */
  if (forward) {
    /* Minimize the final down function (drv:404-406) */
    if (chain == 0 && lane == 0) {
      g_sm.total_up = total_intervals;
      g_sm.max_up = max_intervals;
    }
  }
  device_fence(); /* the stored functions of both chains are read back by backtrack_wave */
  block_sync(chain);
  if (chain == 1) {
    if (forward) {
      const int id = 2 + b;
      if (in_hbm) {
        minimize_wave(global_list(a, spill_slot, id), g_sm.n[id], &r.best_cost, &r.best_log_mean,
                      &r.prev_seg_end, &r.prev_log_mean);
      } else {
        minimize_wave(lds_list(id), g_sm.n[id], &r.best_cost, &r.best_log_mean, &r.prev_seg_end,
                      &r.prev_log_mean);
      }
      bt.best_log_mean = r.best_log_mean;
      bt.prev_log_mean = r.prev_log_mean;
      bt.prev_seg_end = r.prev_seg_end;
    }
    /* decode as far as the records at hand reach: everything (full store), the segment ends
     * inside the block just recomputed, or -- after the forward pass of the checkpointed store
     * -- nothing but the one-segment model's only row */
    backtrack_wave(a, p, N, bt, fn_up, fn_down, t_origin, (K > 0 && forward) ? N : t_lo);
  }
  if (!CKPT || K == 0) break;
  /* checkpointed store: which block holds the next segment end?  (-1: decoding complete) */
  if (chain == 1 && lane == 0) {
    int c = -1;
    if (bt.status == 0 && bt.prev_seg_end >= 0)
      c = bt.prev_seg_end == 0 ? 0 : (bt.prev_seg_end - 1) / K;
    g_sm.bt_next = c;
  }
  block_sync(chain);
  const int c = uniform_i(g_sm.bt_next);
  if (c < 0) break;
  /* next pass: data points c K + 1 .. (c + 1) K (block 0 also redoes data point 0), from the
   * checkpoint after data point c K, records into this wave's region of the arena */
  forward = false;
  next_ckpt = -1;
  t_origin = c * K;
  t_lo = c == 0 ? 0 : c * K + 1;
  t_hi = (c + 1) * K + 1 < N ? (c + 1) * K + 1 : N;
  t_first = t_lo;
  {
    /* this wave's region: regions are packed whole into the arena's blocks */
    const unsigned long long per_block = (1ull << a.ar_block_log2) / a.ckpt_region;
    const unsigned long long region = (unsigned long long)p * 2ull + (unsigned long long)chain;
    cursor_point(a, cur, ((region / per_block) << a.ar_block_log2) + (region % per_block) * a.ckpt_region,
                 (int)a.ckpt_region);
    cur.store = 1;
  }
  in_hbm = false;
  b = 0;
  if (c == 0) {
    cum_weight_i = 0.0;
    cum_weight_prev_i = -1.0;
    if (lane == 0) g_sm.n[2 * chain] = g_sm.n[2 * chain + 1] = 0;
  } else {
    device_fence();
    /* functions that do not fit LDS: the block starts with the lists in the HBM spill area
     * (the same decision in both waves: both counts are in the checkpoint) */
    in_hbm = uniform_i(ckpt_count(*a.self, p, c - 1, 0)) > LDS_CAP ||
             uniform_i(ckpt_count(*a.self, p, c - 1, 1)) > LDS_CAP;
    if (in_hbm && spill_slot < 0) {
      spill_slot = uniform_i(take_spill_slot(*a.self, chain));
      if (spill_slot < 0) {
        status = PST_SPILL_FULL;
        break;
      }
    }
    const int n_ck = uniform_i(ckpt_load(*a.self, p, c - 1, chain, 2 * chain, in_hbm ? 1 : 0,
                                         spill_slot));
    if (lane == 0) g_sm.n[2 * chain] = n_ck;
    cum_weight_i = uniform_d(ckpt_cum_weight(*a.self, p, c - 1));
    cum_weight_prev_i = cum_weight_i;
  }
  block_sync(chain); /* both functions are in LDS before either chain reads the other's */
  } /* passes */
#ifdef PSD_HELPER_WAVES
  if (mail_wait(chain)) mail_post(chain, HOP_EXIT);
  else if (lane == 0) g_sm.mail[chain].abort = 1;
#endif
#ifdef PSD_PROFILE
  if (lane == 0 && a.prof) {
    long long *dst = a.prof + ((long long)p * 2 + chain) * N_PROF;
    for (int i = 0; i < N_PROF; i++) dst[i] = g_sm.prof[chain][i];
    dst[PROF_TOTAL] = cycle_now() - g_sm.t_begin[chain];
  }
#endif
  if (chain == 1) {
    r.status = status != 0 ? status : bt.status;
    r.wave_err = g_sm.abort_err[0] | g_sm.abort_err[1] | g_sm.abort_err[2];
    r.max_intervals = max_intervals > g_sm.max_up ? max_intervals : g_sm.max_up;
    r.total_intervals = total_intervals + g_sm.total_up;
    r.n_segments = bt.n_seg;
    r.n_equality = bt.n_eq;
    r.n_serial_env = g_sm.serial[0] + g_sm.serial[1];
    r.step_reached = step_reached;
    r.spill_steps = spill_steps;
    r.parked = parked;
    r.max_spin = 0;
#ifdef PSD_SPIN_STATS
    for (int w = 0; w < 4; w++) r.max_spin = g_sm.spin_max[w] > r.max_spin ? g_sm.spin_max[w] : r.max_spin;
#endif
    r.cycles = cycle_now() - g_sm.t_begin[chain];
    if (lane == 0) a.result[p] = r;
  }
}

__global__ __launch_bounds__(FORWARD_THREADS) PSD_KERNEL_OCC void fpop_forward_kernel(
    DeviceArgs a) {
  forward_body<false>(a);
}
/* the same for problem sets that use the checkpointed store (a.ckpt_interval > 0) */
__global__ __launch_bounds__(FORWARD_THREADS) PSD_KERNEL_OCC void fpop_forward_ckpt_kernel(
    DeviceArgs a) {
  forward_body<true>(a);
}

__global__ void math_probe_kernel(int op, int n, const double *x, double *y) {
  psd_tables_init();
  int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < n) {
    if (op == 2) /* (x holds n numerators, then n denominators) */
      y[i] = psd_div(x[i], x[n + i]);
    else if (op == 3) /* the compiler's division sequence as it is (what psd_div repairs) */
      y[i] = x[i] / x[n + i];
    else
      y[i] = op == 0 ? d_exp(x[i]) : d_log(x[i]);
  }
}

}  // namespace PSD_VARIANT
}  // namespace psd
