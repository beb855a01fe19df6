"""Device-resident (penalty x contig) problem sets: the additive grid entry of the C ABI
(include/peaksegdisk_hip.h, peakseg_hip_problem_set_*).  One problem = one dynamic program
of /root/reference/src/PeakSegFPOPLog.cpp:258-442 for a (contig, penalty) pair."""
import ctypes

import numpy as np

from . import _native


class ProblemSet:
    """contigs: list of (count, weight) int32 arrays; problems: list of (contig_index, penalty)."""

    def __init__(self, contigs, problems, device=0, arena_pieces=0, lib=None):
        self._lib = lib or _native.lib
        self._h = ctypes.c_void_p()
        self.contigs = [(np.ascontiguousarray(c, dtype=np.int32),
                         np.ascontiguousarray(w, dtype=np.int32)) for c, w in contigs]
        self.problems = [(int(c), float(p)) for c, p in problems]
        nc = len(self.contigs)
        n_bins = (ctypes.c_int * nc)(*[len(c) for c, _ in self.contigs])
        cptr = (ctypes.c_void_p * nc)(*[c.ctypes.data for c, _ in self.contigs])
        wptr = (ctypes.c_void_p * nc)(*[w.ctypes.data for _, w in self.contigs])
        npb = len(self.problems)
        pc = (ctypes.c_int * npb)(*[c for c, _ in self.problems])
        pp = (ctypes.c_double * npb)(*[p for _, p in self.problems])
        st = self._lib.peakseg_hip_problem_set_create(
            device, nc, n_bins, cptr, wptr, npb, pc, pp, ctypes.c_ulonglong(arena_pieces),
            ctypes.byref(self._h))
        if st != 0:
            raise RuntimeError("peakseg_hip_problem_set_create: status %d: %s" % (
                st, self._lib.peakseg_hip_last_error().decode()))
        self.bins_per_solve = sum(len(self.contigs[c][0]) for c, _ in self.problems)

    def solve(self):
        """Forward DP + backtrack for every problem; returns (forward_ms, backtrack_ms)."""
        f = ctypes.c_float()
        b = ctypes.c_float()
        st = self._lib.peakseg_hip_problem_set_solve(self._h, ctypes.byref(f), ctypes.byref(b))
        if st != 0:
            raise RuntimeError("peakseg_hip_problem_set_solve: status %d: %s" % (
                st, self._lib.peakseg_hip_last_error().decode()))
        return f.value, b.value

    def result(self, p):
        r = _native.PsdResult()
        if self._lib.peakseg_hip_problem_set_result(self._h, p, ctypes.byref(r)) != 0:
            raise RuntimeError("no result for problem %d" % p)
        return r

    def segments(self, p):
        """(seg_start_index, seg_mean) in the reference's output order (last segment first)."""
        r = self.result(p)
        start = np.empty(max(r.n_segments, 1), dtype=np.int32)
        mean = np.empty(max(r.n_segments, 1), dtype=np.float64)
        n = self._lib.peakseg_hip_problem_set_segments(
            self._h, p, r.n_segments, start.ctypes.data, mean.ctypes.data)
        if n < 0:
            raise RuntimeError("segments(%d): %s" % (p, self._lib.peakseg_hip_last_error().decode()))
        return start[:n], mean[:n]

    def export_db(self, p, chrom_end, path):
        ce = np.ascontiguousarray(chrom_end, dtype=np.int32)
        if self._lib.peakseg_hip_problem_set_export_db(self._h, p, ce.ctypes.data,
                                                       path.encode()) != 0:
            raise RuntimeError("export_db failed")

    @property
    def kernel_build(self):
        """'lat' or 'thr': which build of the forward kernel the last solve() used."""
        return self._lib.peakseg_hip_problem_set_kernel_build(self._h).decode()

    @property
    def hbm_bytes(self):
        return int(self._lib.peakseg_hip_problem_set_bytes(self._h))

    @property
    def checkpoint_interval(self):
        """0: every cost function is kept in HBM; K > 0: checkpointed store (recompute)."""
        return int(self._lib.peakseg_hip_problem_set_checkpoint_interval(self._h))

    @property
    def arena_bytes_used(self):
        """bytes of the arena the last solve() handed out"""
        return int(self._lib.peakseg_hip_problem_set_arena_bytes_used(self._h))

    @property
    def solve_stats(self):
        """(kernel launches, data points worked through) of the last solve(): one launch and the
        sum of the problems' lengths unless a store had to grow."""
        n = ctypes.c_int()
        steps = ctypes.c_ulonglong()
        self._lib.peakseg_hip_problem_set_solve_stats(self._h, ctypes.byref(n), ctypes.byref(steps))
        return n.value, steps.value

    @property
    def arena_stats(self):
        """(pieces per arena block, blocks mapped now, blocks the last solve() mapped while its
        kernels were running)"""
        bp = ctypes.c_ulonglong()
        nb = ctypes.c_int()
        live = ctypes.c_int()
        self._lib.peakseg_hip_problem_set_arena_stats(self._h, ctypes.byref(bp), ctypes.byref(nb),
                                                      ctypes.byref(live))
        return bp.value, nb.value, live.value

    @property
    def park_stats(self):
        """(problems parked by the last solve()'s launches, pieces of the overflow pool those
        parks took)"""
        n = ctypes.c_int()
        pool = ctypes.c_ulonglong()
        self._lib.peakseg_hip_problem_set_park_stats(self._h, ctypes.byref(n), ctypes.byref(pool))
        return n.value, pool.value

    def pack_tables(self):
        """Pack every problem's segment table at its exact size in HBM.  Returns (rows int64[k],
        device address of the packed seg_start int32 array, of the packed seg_mean float64
        array, total rows); the addresses stay valid until the next solve() / close()."""
        rows = np.zeros(len(self.problems), dtype=np.int64)
        ps = ctypes.c_void_p()
        pm = ctypes.c_void_p()
        total = self._lib.peakseg_hip_problem_set_pack_tables(
            self._h, rows.ctypes.data, ctypes.byref(ps), ctypes.byref(pm))
        if total < 0:
            raise RuntimeError("pack_tables: %s" % self._lib.peakseg_hip_last_error().decode())
        return rows, ps.value or 0, pm.value or 0, int(total)

    def packed_download(self, total):
        """the packed tables of pack_tables() as host arrays"""
        start = np.empty(total, dtype=np.int32)
        mean = np.empty(total, dtype=np.float64)
        if self._lib.peakseg_hip_problem_set_packed_download(self._h, start.ctypes.data,
                                                             mean.ctypes.data) != 0:
            raise RuntimeError("packed_download: %s" % self._lib.peakseg_hip_last_error().decode())
        return start, mean

    def cycles_per_step(self, p):
        """shader cycles per data point of problem p in the last solve() (0.0 when unknown,
        e.g. under the SIMT emulator, or when the problem was resumed after a park)"""
        cyc = int(self._lib.peakseg_hip_problem_set_cycles(self._h, p))
        if cyc <= 0 or self.solve_stats[0] != 1:
            return 0.0
        return cyc / float(len(self.contigs[self.problems[p][0]][0]))

    def set_penalty(self, p, penalty):
        """Change one problem's penalty in place; the next solve() reuses contig and arena."""
        if self._lib.peakseg_hip_problem_set_set_penalty(self._h, p, float(penalty)) != 0:
            raise RuntimeError("set_penalty(%d) failed" % p)
        self.problems[p] = (self.problems[p][0], float(penalty))

    def close(self):
        if self._h:
            self._lib.peakseg_hip_problem_set_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
