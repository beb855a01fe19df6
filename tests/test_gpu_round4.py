"""Round-4 parity and robustness tests on an MI355X (through the C ABI).

  * the two search fixtures the reference's tests hold and rounds 1-3 skipped:
    `sequentialSearch_dir(data.dir, most.peaks-1L)` on Mono27ac
    (tests/testthat/test-TRAVIS-sequentialSearch.R:31-35: the small-penalty end of the search,
    where the next penalty approaches 0 and the "not a new model" exits fire) and
    "peaks.int=5 but max=2 peaks for N=6 data" (tests/testthat/test-CRAN-sequentialSearch.R:22-26);
  * parked problems whose functions are too long for a park slot (the overflow pool), over
    several arena exhaustions in one solve;
  * the spin bounds that guard the waves' hand-shakes against a hang, measured against the
    slowest legitimate step.
"""
import ctypes
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from test_gpu_round3 import _oracle_search

GPU = pytest.mark.gpu

SIX_POINTS = [(0, 1, 3), (1, 2, 9), (2, 3, 18), (3, 4, 15), (4, 5, 20), (5, 6, 2)]


@pytest.fixture(scope="module")
def psd():
    import __graft_entry__ as entry
    entry.build_hip()
    entry.build_oracle()
    import peaksegdisk_amd
    from peaksegdisk_amd import _native
    assert _native.lib.peakseg_hip_device_count() >= 1, "no HIP device: GPU tests need an MI355X"
    return peaksegdisk_amd


def _search_dirs(tmp_path, text=None):
    gdir = tmp_path / "gpu" / "chr11-60000-580000"
    odir = tmp_path / "oracle" / "chr11-60000-580000"
    for d in (gdir, odir):
        d.mkdir(parents=True)
        if text is None:
            shutil.copy(os.path.join(GOLDEN, "Mono27ac.bedGraph"), str(d / "coverage.bedGraph"))
        else:
            (d / "coverage.bedGraph").write_text(text)
    return gdir, odir


def check_search_against_oracle_loop(psd, gdir, odir, target):
    """sequentialSearch_dir on gdir against the reference's loop driven by the oracle on odir:
    penalty strings (the file names), peaks, iteration/under/over columns, the chosen model and
    every model's two files."""
    fit = psd.sequentialSearch_dir(str(gdir), target)
    trace = _oracle_search(str(odir), target)
    got_pen = [psd.paste(float(p)) for p in fit.others["penalty"]]
    assert got_pen == [m["penalty"] for m in trace]
    assert list(fit.others["peaks"]) == [m["peaks"] for m in trace]
    assert list(fit.others["iteration"]) == [1, 1] + list(range(2, len(trace)))
    for m in trace:
        pre_g = "%s_penalty=%s" % (str(gdir / "coverage.bedGraph"), m["penalty"])
        pre_o = "%s_penalty=%s" % (str(odir / "coverage.bedGraph"), m["penalty"])
        for suffix in ("_segments.bed", "_loss.tsv"):
            assert open(pre_g + suffix, "rb").read() == open(pre_o + suffix, "rb").read(), m
    return fit, trace


@GPU
def test_search_for_most_peaks_minus_one(psd, tmp_path, text=None, most_peaks=3199):
    """test-TRAVIS-sequentialSearch.R:31-35: `most.peaks <- fit$others[penalty==0, peaks]`,
    then `sequentialSearch_dir(data.dir, most.peaks-1L)` "returns something".  On Mono27ac the
    penalty-0 model has 3199 peaks, so the target is 3198: the search works its
    way down to penalties near 0.  Here it must ask for exactly the penalties the reference's
    loop asks for when the oracle computes the models, end by the same rule, and return a model
    with at most 3198 peaks."""
    gdir, odir = _search_dirs(tmp_path, text)
    if most_peaks is None:  # other data than Mono27ac: what the penalty-0 model has
        from test_gpu_round3 import _oracle_model
        probe = tmp_path / "probe"
        probe.mkdir()
        shutil.copy(str(odir / "coverage.bedGraph"), str(probe / "coverage.bedGraph"))
        most_peaks = _oracle_model(str(probe / "coverage.bedGraph"), "0")["peaks"]
    fit, trace = check_search_against_oracle_loop(psd, gdir, odir, most_peaks - 1)
    assert trace[0]["penalty"] == "0" and trace[0]["peaks"] == most_peaks
    assert trace[1]["penalty"] == "Inf" and trace[1]["peaks"] == 0
    assert len(trace) > (4 if text is None else 2), "the search ended at once"
    chosen_peaks = int(fit.loss["peaks"].iloc[0])
    assert chosen_peaks <= most_peaks - 1
    # the candidate rule of R/sequentialSearch_dir.R:68-86: the target itself if a model has it,
    # else the simpler model of the final bracket
    peaks = [m["peaks"] for m in trace]
    if most_peaks - 1 in peaks:
        assert chosen_peaks == most_peaks - 1
    else:
        assert chosen_peaks == max(p for p in peaks if p < most_peaks - 1)
    pens = [float(m["penalty"]) for m in trace[2:]]
    assert min(pens) < 1.0 or text is not None, pens  # the small-penalty end was reached


@GPU
def test_search_with_too_many_peaks_on_six_points(psd, tmp_path):
    """test-CRAN-sequentialSearch.R:8-26: six data points, `sequentialSearch_dir(data.dir, 5L)`
    is an error, "peaks.int=5 but max=2 peaks for N=6 data" -- after the two models of the first
    iteration have been computed (penalty 0 is a dynamic program on the GPU), which leave their
    files.  Through the C ABI and through the Python mirror; a feasible target on the same
    data follows the oracle-driven loop."""
    from peaksegdisk_amd import _native
    text = "".join("chr1\t%d\t%d\t%d\n" % r for r in SIX_POINTS)
    gdir, odir = _search_dirs(tmp_path, text)
    rows = (_native.PsdSearchRow * 16)()
    n = ctypes.c_int()
    chosen = ctypes.c_int()
    st = _native.lib.PeakSegFPOP_sequential_search(os.fsencode(str(gdir)), 5, 0, 16, rows,
                                                   ctypes.byref(n), ctypes.byref(chosen))
    assert st == _native.ERROR_SEARCH_TOO_MANY_PEAKS
    assert _native.last_error() == "peaks.int=5 but max=2 peaks for N=6 data"
    assert n.value == 2 and chosen.value == -1
    assert [rows[k].penalty_str.decode() for k in range(2)] == ["0", "Inf"]
    assert rows[0].peaks == 2 and rows[0].bases == 6 and rows[1].peaks == 0  # 2 peaks
    for pen in ("0", "Inf"):
        assert os.path.exists("%s_penalty=%s_loss.tsv" % (str(gdir / "coverage.bedGraph"), pen))
    with pytest.raises(ValueError, match="peaks.int=5 but max=2 peaks for N=6 data"):
        psd.sequentialSearch_dir(str(gdir), 5)
    # max=2 itself and the one-peak model are found, by the reference's sequence
    for target in (2, 1):
        fit, trace = check_search_against_oracle_loop(psd, gdir, odir, target)
        assert int(fit.loss["peaks"].iloc[0]) <= target


@GPU
def test_parked_long_functions_survive_several_exhaustions(psd, oracle_det, tmp_path, monkeypatch,
                                                           n_bins=6000):
    """a parked problem whose functions exceed a park slot keeps them
    in the overflow pool from the launch that parked it until the workgroup that resumes it
    has read them back; the pool used to be emptied before EVERY launch, so a problem that
    parked again early in a relaunch could overwrite what a still-queued workgroup had not
    loaded yet.  Adversarial counts (functions of hundreds of pieces), an arena far too small,
    no growth under the kernel: several exhaustions in one solve, every park through the
    overflow pool, and the stores equal the oracle's."""
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.increasing_coverage(n_bins)
    pens = ["100", "300", "1000", "30"]
    monkeypatch.setenv("PEAKSEG_HIP_NO_CHECKPOINT", "1")
    monkeypatch.setenv("PEAKSEG_HIP_PIECES_PER_FUNCTION", "1")
    monkeypatch.setenv("PEAKSEG_HIP_NO_LIVE_GROWTH", "1")
    monkeypatch.setenv("PEAKSEG_HIP_ARENA_BLOCK_LOG2", "14" if n_bins >= 4000 else "12")
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens])
    pset.solve()
    launches, steps = pset.solve_stats
    parks, pool_pieces = pset.park_stats
    assert launches >= 3, "fewer than two exhaustions: %r" % (pset.solve_stats,)
    assert steps == n_bins * len(pens), "a problem was restarted instead of resumed"
    assert parks >= 2 and pool_pieces > 128 * 2, (parks, pool_pieces)
    bg = str(tmp_path / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    for i, pen in enumerate(pens):
        r = pset.result(i)
        assert r.status == 0 and (r.max_intervals > 128 or pen == "30")
        db_o = str(tmp_path / ("o%d.db" % i))
        assert oracle_det.solve(bg, pen, db_o) == 0
        db_g = str(tmp_path / ("g%d.db" % i))
        pset.export_db(i, ce, db_g)
        assert open(db_g, "rb").read() == open(db_o, "rb").read(), pen
    pset.close()


@GPU
def test_spin_bounds_leave_three_orders_of_magnitude(psd, tmp_path, n_bins=30000):
    """The waves of a workgroup meet through LDS flags (step_sync, the helper mailboxes) with a
    bound on the number of polls, so that a lost wave becomes an error status instead of a hang
    (fpop_kernels.h SPIN_LIMIT, fpop_wave.h MAIL_SPIN_LIMIT: 2^26 polls).  The bound must never
    be reached by a wave that is merely slow.  A build with -DPSD_SPIN_STATS records the largest
    poll count of every wait; on the slowest legitimate steps there are -- lists in HBM,
    functions of several hundred pieces, chunk 0's Newton solves running to their 100-step cap
    (adversarial counts, penalty 100) -- it must stay below a thousandth of the bound."""
    import __graft_entry__ as entry
    from peaksegdisk_amd import ProblemSet, _native, synthetic
    csrc = os.path.join(ROOT, "peaksegdisk_amd", "csrc")
    lib_path = str(tmp_path / "libpeaksegdisk_hip_spin.so")
    subprocess.run([entry.HIPCC] + entry.HIP_FLAGS + [ "-DPSD_SPIN_STATS",
                    "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                    os.path.join(csrc, "peakseg_hip.cpp"), "-o", lib_path], check=True)
    lib = _native.declare(ctypes.CDLL(lib_path))
    lib.peakseg_hip_spin_limit.restype = ctypes.c_longlong
    limit = lib.peakseg_hip_spin_limit()
    assert limit == 1 << 26
    cs, ce, cnt = synthetic.increasing_coverage(n_bins)
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, 100.0)], lib=lib)
    pset.solve()
    r = pset.result(0)
    assert r.status == 0 and r.spill_steps > n_bins - 2000 and r.max_intervals > 300
    worst = lib.peakseg_hip_problem_set_max_spin(pset._h, 0)
    pset.close()
    # ... and the LDS-resident latency path on ordinary data
    cs, ce, cnt = synthetic.poisson_coverage(50000, seed=3)
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, 0.5), (0, 2000.0)], lib=lib)
    pset.solve()
    worst_lds = max(lib.peakseg_hip_problem_set_max_spin(pset._h, p) for p in range(2))
    pset.close()
    print("largest poll count of a wait: %d with the lists in HBM, %d in LDS; bound %d"
          % (worst, worst_lds, limit))
    assert 0 < worst_lds and 0 < worst
    assert max(worst, worst_lds) * 1000 < limit


@GPU
def test_packed_build_parity_planner_and_hand_over(psd, oracle_det, tmp_path, monkeypatch,
                                                   n_contigs=400, n_bins=1200, adv_bins=2500,
                                                   many=(2100, 300)):
    """Round 4: a third build of the forward kernel for sets of many problems of similar length
    (40-piece LDS lists, three waves per SIMD, six workgroups per CU; +18 % on 6144 equal
    problems, profiles/r04/ab_thr_occupancy_*.log).  (a) The planner picks it for such a set and
    its results equal the throughput build's bit for bit, and the oracle's stores; (b) a problem
    whose functions outgrow 40 pieces is parked by the packed build and resumed on the
    throughput build (64-piece lists, and from there the HBM path): two launches, every data point
    computed once, stores equal to the oracle's; (c) after a set that handed over many of its
    problems, the planner leaves the packed build alone."""
    from peaksegdisk_amd import ProblemSet, synthetic
    pens = ["0.2", "3", "40", "600", "9000", "150000", "2500000"]
    contigs, data = [], []
    for c in range(n_contigs):
        cs, ce, cnt = synthetic.poisson_coverage(n_bins + c % 50, seed=300 + c)
        contigs.append((cnt, (ce - cs).astype(np.int32)))
        data.append((cs, ce, cnt))
    # (2800 problems: well beyond the 1024 the throughput build holds at once, so that what a CU
    # holds, not one problem's speed, decides when the set ends)
    problems = [(c, float(p)) for c in range(n_contigs) for p in pens]
    tables = {}
    for knob, want in (("", "pk"), ("PEAKSEG_HIP_NO_PACKED", "thr")):
        if knob:
            monkeypatch.setenv(knob, "1")
        pset = ProblemSet(contigs, problems)
        pset.solve()
        assert pset.kernel_build == want, pset.kernel_build
        tables[want] = [(pset.result(i).max_intervals, pset.result(i).total_intervals,
                         pset.result(i).best_cost) + pset.segments(i) for i in range(len(problems))]
        if want == "pk":
            # (every data point once, also if one of these functions outgrew 40 pieces and its
            # problem went on on the wider build)
            launches, steps = pset.solve_stats
            assert launches <= 2 and steps == sum(len(contigs[c][0]) for c, _ in problems)
            for c in (0, n_contigs // 2, n_contigs - 1):  # whole stores against the oracle
                cs, ce, cnt = data[c]
                bg = str(tmp_path / ("c%d.bedGraph" % c))
                synthetic.write_bedgraph(bg, cs, ce, cnt)
                for k, pen in enumerate(pens):
                    db_o = str(tmp_path / "o.db")
                    assert oracle_det.solve(bg, pen, db_o) == 0
                    db_g = str(tmp_path / "g.db")
                    pset.export_db(c * len(pens) + k, ce, db_g)
                    assert open(db_g, "rb").read() == open(db_o, "rb").read(), (c, pen)
        pset.close()
        if knob:
            monkeypatch.delenv(knob)
    for a, b in zip(tables["pk"], tables["thr"]):
        assert a[:3] == b[:3] and np.array_equal(a[3], b[3])
        assert np.array_equal(a[4].view(np.uint64), b[4].view(np.uint64))
    # (b) functions that outgrow the packed build's lists
    cs, ce, cnt = synthetic.increasing_coverage(adv_bins)
    monkeypatch.setenv("PEAKSEG_HIP_VARIANT", "pk")
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, 100.0), (0, 30.0)])
    pset.solve()
    assert pset.kernel_build == "pk"
    launches, steps = pset.solve_stats
    assert launches == 2 and steps == 2 * adv_bins, pset.solve_stats
    bg = str(tmp_path / "adv.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    for i, pen in enumerate(("100", "30")):
        r = pset.result(i)
        assert r.status == 0 and r.max_intervals > 40
        db_o = str(tmp_path / "oa.db")
        assert oracle_det.solve(bg, pen, db_o) == 0
        db_g = str(tmp_path / "ga.db")
        pset.export_db(i, ce, db_g)
        assert open(db_g, "rb").read() == open(db_o, "rb").read(), pen
    # (the two handed-over problems get a CU each: the latency build's 128-piece lists, and
    # beyond those the HBM path)
    assert pset.result(0).max_intervals > 128 and pset.result(0).spill_steps > 0
    pset.close()
    monkeypatch.delenv("PEAKSEG_HIP_VARIANT")
    # (c) the planner cannot know how long functions get: a set it gave to the packed build and
    # that handed over more than one problem in twenty turns the packed build off for the sets
    # this process plans afterwards
    n_many, bins_many = many
    cs, ce, cnt = synthetic.increasing_coverage(bins_many)
    long_set = ([(cnt, (ce - cs).astype(np.int32))], [(0, 100.0)] * n_many)
    pset = ProblemSet(*long_set)
    pset.solve()
    # (every problem is handed over here; those the park pool had no room for start over on the
    # wider build, after the pool has grown)
    assert pset.kernel_build == "pk"
    assert pset.solve_stats[0] >= 2 and pset.solve_stats[1] >= n_many * bins_many
    first = (pset.result(0).best_cost, pset.result(0).max_intervals) + pset.segments(0)
    assert first[1] > 40
    pset.close()
    pset = ProblemSet(*long_set)
    pset.solve()
    assert pset.kernel_build == "thr"
    again = (pset.result(n_many - 1).best_cost, pset.result(n_many - 1).max_intervals) \
        + pset.segments(n_many - 1)
    assert first[:2] == again[:2] and np.array_equal(first[2], again[2])
    assert np.array_equal(first[3].view(np.uint64), again[3].view(np.uint64))
    pset.close()


@GPU
def test_spill_pool_exhaustion_parks_instead_of_starting_over(psd, oracle_det, tmp_path, monkeypatch,
                                                              n_bins=2500, n_problems=6):
    """Functions that outgrow the LDS lists move to a slot of the HBM spill pool; when more
    problems need one at the same time than the pool has, rounds 1-3 started those problems
    over after enlarging the pool.  Now they park at the data point that needed the slot (its
    inputs are still in LDS) and go on from there: a pool of one slot and six problems that all
    need it at the same data point take three launches (1, 4, 16 slots) and compute every data
    point once; stores equal to the oracle's."""
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.increasing_coverage(n_bins)
    monkeypatch.setenv("PEAKSEG_HIP_SPILL_SLOTS", "1")
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, 100.0)] * n_problems)
    pset.solve()
    launches, steps = pset.solve_stats
    assert launches == 3 and steps == n_problems * n_bins, pset.solve_stats
    bg = str(tmp_path / "adv.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    db_o = str(tmp_path / "o.db")
    assert oracle_det.solve(bg, "100", db_o) == 0
    want = open(db_o, "rb").read()
    for i in range(n_problems):
        r = pset.result(i)
        assert r.status == 0 and r.spill_steps > 0
        db_g = str(tmp_path / "g.db")
        pset.export_db(i, ce, db_g)
        assert open(db_g, "rb").read() == want, i
    pset.close()


KNOB_COMBINATIONS = [
    # packed build, everything that can run short runs short: hand-over to the wider builds,
    # parks for want of arena (no growth under the kernel, tiny blocks) and of spill slots
    {"PEAKSEG_HIP_VARIANT": "pk", "PEAKSEG_HIP_SPILL_SLOTS": "1", "PEAKSEG_HIP_PIECES_PER_FUNCTION": "1",
     "PEAKSEG_HIP_NO_LIVE_GROWTH": "1", "PEAKSEG_HIP_ARENA_BLOCK_LOG2": "12"},
    # throughput build, growth under the kernel with tiny blocks, one spill slot, a small pool
    # for the parked long functions
    {"PEAKSEG_HIP_VARIANT": "thr", "PEAKSEG_HIP_SPILL_SLOTS": "1", "PEAKSEG_HIP_PIECES_PER_FUNCTION": "1",
     "PEAKSEG_HIP_ARENA_BLOCK_LOG2": "12", "PEAKSEG_HIP_CKPT_OVERFLOW": "256"},
    # latency build on plain allocations, no growth under the kernel
    {"PEAKSEG_HIP_VARIANT": "lat", "PEAKSEG_HIP_NO_VMM": "1", "PEAKSEG_HIP_SPILL_SLOTS": "1",
     "PEAKSEG_HIP_PIECES_PER_FUNCTION": "1", "PEAKSEG_HIP_NO_LIVE_GROWTH": "1",
     "PEAKSEG_HIP_ARENA_BLOCK_LOG2": "13", "PEAKSEG_HIP_CKPT_OVERFLOW": "256"},
    # problems that cannot park (rounds 1-2): every exhaustion starts them over
    {"PEAKSEG_HIP_VARIANT": "thr", "PEAKSEG_HIP_NO_PARK": "1", "PEAKSEG_HIP_SPILL_SLOTS": "1",
     "PEAKSEG_HIP_PIECES_PER_FUNCTION": "1", "PEAKSEG_HIP_NO_LIVE_GROWTH": "1",
     "PEAKSEG_HIP_ARENA_BLOCK_LOG2": "12"},
    # the planner's own choice, growth under the kernel, defaults otherwise
    {"PEAKSEG_HIP_SPILL_SLOTS": "2", "PEAKSEG_HIP_ARENA_BLOCK_LOG2": "12"},
    # the checkpointed store (no parking there): one spill slot, a small pool for the
    # checkpoints of long functions, on both builds it can run on
    {"PEAKSEG_HIP_CHECKPOINT": "64", "PEAKSEG_HIP_VARIANT": "thr", "PEAKSEG_HIP_SPILL_SLOTS": "1",
     "PEAKSEG_HIP_CKPT_OVERFLOW": "256", "PEAKSEG_HIP_ARENA_BLOCK_LOG2": "12"},
    {"PEAKSEG_HIP_CHECKPOINT": "37", "PEAKSEG_HIP_VARIANT": "lat", "PEAKSEG_HIP_SPILL_SLOTS": "1",
     "PEAKSEG_HIP_CKPT_OVERFLOW": "256"},
]


@GPU
def test_knob_combinations_on_a_mixed_set(psd, oracle_det, tmp_path, monkeypatch, n_poisson=3000,
                                          n_increasing=1200, n_shapes=3):
    """The ways a solve can run short -- arena (with and without growth under the kernel), spill
    slots, the pool for parked long functions, the packed build's lists -- were each tested on
    data made for them.  Here they meet: one set of Poisson coverage, increasing counts (functions
    of hundreds of pieces) and three of the odd data shapes, several penalties each, solved under
    combinations of the knobs that make everything scarce at once.  Every problem of every
    combination: status 0 and the whole store equal to the oracle's database (the checkpointed
    store keeps no database: segment table and summary against the oracle's files)."""
    from peaksegdisk_amd import ProblemSet, synthetic
    from test_gpu_parity import varied_shape_cases
    from test_gpu_round3 import check_tables_vs_oracle_files
    data = []
    cs, ce, cnt = synthetic.poisson_coverage(n_poisson, seed=77)
    data.append((cnt, ce - cs, cs, ce, ["0.5", "40", "3000", "200000"]))
    cs, ce, cnt = synthetic.increasing_coverage(n_increasing)
    data.append((cnt, ce - cs, cs, ce, ["100", "30", "1000"]))
    for cnt, w, cs, ce, pens in varied_shape_cases(n_shapes, 4242):
        data.append((cnt, w, cs, ce, pens))
    contigs = [(np.asarray(c).astype(np.int32), np.asarray(w).astype(np.int32)) for c, w, _, _, _ in data]
    problems, want, files = [], [], []
    for k, (cnt, w, cs, ce, pens) in enumerate(data):
        bg = str(tmp_path / ("k%d.bedGraph" % k))
        synthetic.write_bedgraph(bg, cs, ce, cnt)
        for pen in pens:
            db_o = str(tmp_path / "o.db")
            assert oracle_det.solve(bg, pen, db_o) == 0
            want.append(open(db_o, "rb").read())
            problems.append((k, float(pen)))
            files.append((bg, pen))
    for knobs in KNOB_COMBINATIONS:
        for name, value in knobs.items():
            monkeypatch.setenv(name, value)
        pset = ProblemSet(contigs, problems)
        pset.solve()
        for i, (k, _) in enumerate(problems):
            r = pset.result(i)
            assert r.status == 0, (knobs, i, r.kernel_status)
            if pset.checkpoint_interval > 0:
                start, mean = pset.segments(i)
                check_tables_vs_oracle_files(
                    files[i][0], files[i][1], np.asarray(data[k][2]), np.asarray(data[k][3]), start,
                    mean, [r.n_segments, r.n_equality_constraints, r.max_intervals,
                           r.total_intervals, r.best_cost])
                continue
            db_g = str(tmp_path / "g.db")
            pset.export_db(i, np.asarray(data[k][3]).astype(np.int32), db_g)
            assert open(db_g, "rb").read() == want[i], (knobs, i)
        print(knobs, pset.kernel_build, pset.solve_stats, pset.park_stats)
        pset.close()
        for name in knobs:
            monkeypatch.delenv(name)


def _near_tie_problem():
    import json
    g = json.load(open(os.path.join(GOLDEN, "division_near_tie.json")))
    return (np.array(g["chromStart"], dtype=np.int32), np.array(g["chromEnd"], dtype=np.int32),
            np.array(g["count"], dtype=np.int32), g["penalties"], g["db_sha256"])


@GPU
def test_division_near_a_midpoint(psd, oracle_det, tmp_path, monkeypatch):
    """The MI355X's fp64 division sequence is one unit off when the exact quotient lies within
    ~2^-50 of a unit of a midpoint between two doubles, which divisors of 1 - k ulp produce --
    and the Linear coefficient of a piece that has seen every weight is such a number.  The
    64-bin problem of tests/golden/division_near_tie.json has that quotient at data point 47
    (0x1.6666666666663p-1 / 0x1.ffffffffffffbp-1): before psd_div (include/peakseg_detmath.h) its
    stores differed from the oracle's in a prev_log_mean and a breakpoint, on every build."""
    from peaksegdisk_amd import ProblemSet, synthetic
    import hashlib
    cs, ce, cnt, pens, sha = _near_tie_problem()
    bg = str(tmp_path / "c.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    want = []
    for pen in pens:
        db_o = str(tmp_path / "o.db")
        assert oracle_det.solve(bg, pen, db_o) == 0
        want.append(open(db_o, "rb").read())
        assert hashlib.sha256(want[-1]).hexdigest() == sha["det:" + pen]
    for build in ("lat", "thr", "pk"):
        monkeypatch.setenv("PEAKSEG_HIP_VARIANT", build)
        pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens])
        pset.solve()
        for i, pen in enumerate(pens):
            assert pset.result(i).status == 0
            db_g = str(tmp_path / "g.db")
            pset.export_db(i, ce, db_g)
            assert open(db_g, "rb").read() == want[i], (build, pen)
        pset.close()
    monkeypatch.delenv("PEAKSEG_HIP_VARIANT")


@GPU
def test_psd_div_is_ieee_division(psd):
    """psd_div on the device == the host's division, bit for bit, on random operands and on the
    structured ones that come closest to midpoints (divisors a few units either side of powers
    of two under numerators a few units off small fractions: tools/div_probe.py, where the
    compiler's own sequence is 29 of 8 million off)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import div_probe
    from peaksegdisk_amd import _native
    raw_off = 0
    for name, a, b in div_probe.cases(n=2000000, seed=17):
        want = a / b
        got = div_probe.device_div(2, a, b)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), name
        raw = div_probe.device_div(3, a, b)
        raw_off += int(np.count_nonzero(raw.view(np.uint64) != want.view(np.uint64)))
    print("quotients the compiler's own sequence has one unit off: %d" % raw_off)
    assert raw_off > 0, "the hardware's division agrees with IEEE on the known near-midpoint quotient?"

