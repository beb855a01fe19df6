"""Round-2 parity and robustness tests on an MI355X (through the C ABI):

  * endpoint parity against the reference's own arithmetic (glibc exp/log, oracle libm build)
    at BASELINE.json sizes: 1e6 bins x 64 penalties on 16 penalties of the grid, and one
    1e7-bin contig x 64 penalties on 2 penalties; per penalty, how many stored functions differ
    in piece count between the deterministic and the glibc arithmetic;
  * the resident penalty search: exact penalty sequence, in order;
  * PeakSegFPOP_dir's cache protocol in the batch entry;
  * the rare paths on the device: Newton step-cap fallback, spill-pool and arena regrowth,
    several processes on one GPU under PEAKSEG_HIP_MAX_BYTES, batches of more problems than
    file descriptors, every contig of a throughput-build set against the oracle.
"""
import ctypes
import json
import os
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import GOLDEN, ORACLE_DIR, ROOT, read_loss, read_segments

GPU = pytest.mark.gpu
REL_TOL = 1e-6
CLI_LIBM = os.path.join(ORACLE_DIR, "_build", "oracle_cli_libm")
CLI_DET = os.path.join(ORACLE_DIR, "_build", "oracle_cli_det")


@pytest.fixture(scope="module")
def psd():
    import __graft_entry__ as entry
    entry.build_hip()
    entry.build_oracle()
    import peaksegdisk_amd
    from peaksegdisk_amd import _native
    assert _native.lib.peakseg_hip_device_count() >= 1, "no HIP device: GPU tests need an MI355X"
    return peaksegdisk_amd


def write_bedgraph_chunked(path, cs, ce, cnt, chrom="chrSynth", chunk=500000):
    with open(path, "w") as f:
        for o in range(0, len(cnt), chunk):
            f.write("".join("%s\t%d\t%d\t%d\n" % (chrom, s, e, c) for s, e, c in zip(
                cs[o:o + chunk].tolist(), ce[o:o + chunk].tolist(), cnt[o:o + chunk].tolist())))


def db_piece_counts(path, n_bins):
    """Piece count of every stored function of a DiskVector-format file
    (PeakSegFPOPLog.cpp:12-34,76-141): 2N 16-byte positions, then {int size, int n, ...}."""
    with open(path, "rb") as f:
        table = np.frombuffer(f.read(32 * n_bins), dtype=np.int64)[0::2]
        data = np.memmap(path, dtype=np.uint8, mode="r")
        pos = table[table > 0]
        idx = pos[:, None] + np.arange(4, 8)[None, :]
        n = np.ascontiguousarray(np.asarray(data[idx])).view(np.int32)[:, 0]
    out = np.zeros(2 * n_bins, dtype=np.int64)
    out[table > 0] = n
    return out


def run_cli(cli, bg, pen, db):
    st = subprocess.run([cli, bg, pen, db], stdout=subprocess.DEVNULL).returncode
    assert st == 0, (cli, pen, st)


def check_against_libm_files(pset, i, pen, bg, cs, ce, n_bins):
    """segment rows (coordinates, 6-digit means) and the integer loss fields identical to the
    glibc-arithmetic oracle's files; total.loss within 1e-6 relative.  (Columns through pandas'
    C reader and formatter: the small penalties of a 1e6-bin contig have 5e5 rows each.)"""
    from conftest import format_g, read_segment_columns
    f_start, f_end, f_status, f_mean = read_segment_columns("%s_penalty=%s_segments.bed" % (bg, pen))
    start, mean = pset.segments(i)
    assert len(f_start) == len(start), pen
    got_start = np.where(start < 0, int(cs[0]), ce[np.maximum(start, 0)])
    assert np.array_equal(f_start, got_start), pen
    assert np.array_equal(f_end[1:], got_start[:-1]), pen
    assert list(f_status[:3]) == ["background", "peak", "background"][:len(f_status)]
    assert list(f_mean) == format_g(mean), pen
    r = pset.result(i)
    loss = read_loss("%s_penalty=%s_loss.tsv" % (bg, pen)).split("\t")
    assert [int(v) for v in loss[1:5]] == [r.n_segments, r.n_peaks, int(ce[-1] - cs[0]), n_bins]
    assert int(loss[7]) == r.n_equality_constraints, pen
    cw = float(loss[3])
    total = r.best_cost * cw - float(pen) * r.n_peaks
    assert total == pytest.approx(float(loss[6]), rel=REL_TOL, abs=1e-6), pen
    return float(loss[8]), float(loss[9])


def _oracle_files(background_oracles, spec, tmp_path, runs):
    """The oracle files a full-size test compares with: from the session's background runs
    (conftest.BackgroundOracles: started with the session, done by the time these tests run,
    last), or started now when the test runs on its own."""
    from peaksegdisk_amd import synthetic
    name = spec["name"]
    if name not in background_oracles.jobs:
        pens = synthetic.penalty_grid(64)
        background_oracles.root = str(tmp_path)
        background_oracles.start(name, spec["n_bins"], spec["seed"],
                                 [(cli, pens[i], sub) for cli, i, sub in runs])
    return background_oracles.wait(name)


@GPU
def test_endpoints_vs_glibc_arithmetic_1e6_x64(psd, background_oracles, tmp_path):
    """BASELINE.json configs[1] at full size against the arithmetic the reference itself
    uses: ALL 64 penalties solved by oracle_cli_libm (round 2 sampled 17); every 8th penalty
    also by the deterministic build -- whose files the GPU's tables must equal exactly -- to
    report how many stored functions differ in piece count between the two arithmetics.  On the
    same solve: the size-independent properties (test_gpu_parity.check_grid_properties:
    well-formed tables, up/down constraint, the loss recomputed from the segmentation, peaks
    monotone in the penalty) and determinism of a second solve, which also has to find the
    arena the first one grew (round 4: one solve of this grid per test run instead of two; the
    72 oracle processes run in the background of the whole session, conftest.py)."""
    from conftest import LONG_1E6
    from peaksegdisk_amd import ProblemSet, synthetic
    n_bins = LONG_1E6["n_bins"]
    pens = synthetic.penalty_grid(64)
    pick = list(range(64))
    div_pick = LONG_1E6["det_picks"]
    t0 = time.time()
    cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=LONG_1E6["seed"])
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens])
    f_ms, _ = pset.solve()
    dirs, dbs, _ = _oracle_files(background_oracles, LONG_1E6, tmp_path,
                                 [(CLI_LIBM, i, "libm") for i in pick] +
                                 [(CLI_DET, i, "det") for i in div_pick])
    bg, dbg = dirs["libm"], dirs["det"]
    report = {"bins": n_bins, "kernel_ms": f_ms, "oracle_wall_s": time.time() - t0, "penalties": {}}
    for i in pick:
        assert pset.result(i).status == 0
        mean_int, max_int = check_against_libm_files(pset, i, pens[i], bg, cs, ce, n_bins)
        r = pset.result(i)
        report["penalties"][pens[i]] = {
            "segments": r.n_segments, "mean_intervals_gpu": r.total_intervals / (2.0 * n_bins),
            "mean_intervals_glibc": mean_int, "max_intervals_gpu": r.max_intervals,
            "max_intervals_glibc": max_int}
    for i in div_pick:
        a = db_piece_counts(dbs[("det", pens[i])], n_bins)
        b = db_piece_counts(dbs[("libm", pens[i])], n_bins)
        r = pset.result(i)
        assert int(a.sum()) == r.total_intervals  # the GPU store has the det oracle's counts
        report["penalties"][pens[i]]["functions_with_different_piece_count_det_vs_glibc"] = \
            int((a != b).sum())
        report["penalties"][pens[i]]["functions"] = int(2 * n_bins - 1)
        # ... and the deterministic oracle's files exactly
        from conftest import format_g, read_segment_columns
        d_start, _, _, d_mean = read_segment_columns("%s_penalty=%s_segments.bed" % (dbg, pens[i]))
        start, mean = pset.segments(i)
        assert np.array_equal(d_start, np.where(start < 0, int(cs[0]), ce[np.maximum(start, 0)]))
        assert list(d_mean) == format_g(mean)
        loss = read_loss("%s_penalty=%s_loss.tsv" % (dbg, pens[i])).split("\t")
        assert loss[5] == "%.20g" % r.best_cost and float(loss[9]) == r.max_intervals
        assert int(loss[7]) == r.n_equality_constraints
        assert float(loss[8]) == r.total_intervals / (2.0 * n_bins)
    # size-independent properties of all 64 problems, then determinism of a second solve
    from test_gpu_parity import check_grid_properties
    first = check_grid_properties(pset, pens, cs, ce, cnt, n_bins)
    peaks = [pset.result(i).n_peaks for i in range(64)]
    assert peaks[0] > 1000 * peaks[-1] > 0
    report["arena_blocks_mapped_under_the_kernel"] = pset.arena_stats[2]
    pset.solve()
    assert pset.solve_stats[0] == 1 and pset.arena_stats[2] == 0
    for i in (0, 31, 63):
        start, mean = pset.segments(i)
        assert np.array_equal(start, first[i][0])
        assert np.array_equal(mean.view(np.uint64), first[i][1].view(np.uint64))
        assert pset.result(i).best_cost == first[i][2]
    pset.close()
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_vs_glibc_1e6x64.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))


@GPU
def test_endpoints_vs_glibc_arithmetic_1e7_x64(psd, background_oracles, tmp_path):
    """The north_star size: one 1e7-bin contig x 64 penalties on one GPU, four penalties
    checked against oracle_cli_libm (which needs two to three minutes each: they run in the
    background of the session, conftest.py)."""
    from conftest import LONG_1E7
    from peaksegdisk_amd import ProblemSet, synthetic
    n_bins = LONG_1E7["n_bins"]
    pens = synthetic.penalty_grid(64)
    pick = LONG_1E7["picks"]
    cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=LONG_1E7["seed"])
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens])
    f_ms, _ = pset.solve()
    for i in range(64):
        assert pset.result(i).status == 0, (i, pset.result(i).kernel_status)
    dirs, _, _ = _oracle_files(background_oracles, LONG_1E7, tmp_path,
                               [(CLI_LIBM, i, "libm") for i in pick])
    bg = dirs["libm"]
    report = {"bins": n_bins, "penalties_solved": 64, "kernel_ms": f_ms,
              "bins_per_s": n_bins * 64 / (f_ms / 1e3), "hbm_bytes": pset.hbm_bytes,
              "arena_bytes_used": pset.arena_bytes_used, "checked": {}}
    for i in pick:
        mean_int, max_int = check_against_libm_files(pset, i, pens[i], bg, cs, ce, n_bins)
        r = pset.result(i)
        report["checked"][pens[i]] = {"segments": r.n_segments, "max_intervals_glibc": max_int,
                                      "max_intervals_gpu": r.max_intervals}
    peaks = [pset.result(i).n_peaks for i in range(64)]
    assert all(a >= b for a, b in zip(peaks, peaks[1:]))
    pset.close()
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_vs_glibc_1e7x64.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))


def _oracle_search(oracle, problem_dir, peaks_int):
    """sequentialSearch_dir (R/sequentialSearch_dir.R:31-99) driven by the CPU oracle, one
    solve per model: the sequence the resident driver must reproduce."""
    from peaksegdisk_amd.api import paste
    bg = os.path.join(problem_dir, "coverage.bedGraph")

    def model(pen_str):
        assert oracle.solve(bg, pen_str) == 0
        os.unlink("%s_penalty=%s.db" % (bg, pen_str)) if os.path.exists(
            "%s_penalty=%s.db" % (bg, pen_str)) else None
        c = read_loss("%s_penalty=%s_loss.tsv" % (bg, pen_str)).split("\t")
        return {"penalty": pen_str, "peaks": int(c[2]), "total.loss": float(c[6])}
    trace = [model("0"), model("Inf")]
    over, under = trace[0], trace[1]
    while True:
        if peaks_int in (under["peaks"], over["peaks"]):
            break
        pen = (over["total.loss"] - under["total.loss"]) / (under["peaks"] - over["peaks"])
        if pen < 0:
            break
        m = model(paste(pen))
        trace.append(m)
        if m["peaks"] in (under["peaks"], over["peaks"]):
            break
        if m["peaks"] < peaks_int:
            under = m
        else:
            over = m
    return trace


@GPU
def test_resident_search_exact_sequence(psd, oracle_det, known_answers, tmp_path):
    """sequentialSearch_dir(Mono27ac, 19L): the models IN ORDER.  Peaks equal the reference's
    trace (known_answers.json, survey_8c; test-TRAVIS-sequentialSearch.R:25-29 asserts the
    19); penalty strings equal, character for character, those of the same loop driven by the
    oracle built on the same arithmetic, and agree with the reference's to 1e-9 relative."""
    gdir = tmp_path / "gpu" / "chr11-60000-580000"
    odir = tmp_path / "oracle" / "chr11-60000-580000"
    for d in (gdir, odir):
        d.mkdir(parents=True)
        shutil.copy(os.path.join(GOLDEN, "Mono27ac.bedGraph"), str(d / "coverage.bedGraph"))
    fit = psd.sequentialSearch_dir(str(gdir), 19)
    want = known_answers["mono27ac"]["sequential_search_19"]
    assert list(fit.others["peaks"]) == want["peaks"]
    assert list(fit.others["iteration"]) == [1, 1] + list(range(2, len(want["peaks"])))
    assert int(fit.loss["peaks"].iloc[0]) == 19
    trace = _oracle_search(oracle_det, str(odir), 19)
    got_pen = [psd.paste(float(p)) for p in fit.others["penalty"]]
    assert got_pen == [m["penalty"] for m in trace]
    assert [float(p) for p in got_pen] == pytest.approx([float(p) for p in want["penalties"]],
                                                        rel=1e-9)
    # every model left the reference's three files, identical to the oracle's two + a timing row
    for m in trace:
        pre_g = "%s_penalty=%s" % (str(gdir / "coverage.bedGraph"), m["penalty"])
        pre_o = "%s_penalty=%s" % (str(odir / "coverage.bedGraph"), m["penalty"])
        for suffix in ("_segments.bed", "_loss.tsv"):
            assert open(pre_g + suffix, "rb").read() == open(pre_o + suffix, "rb").read()
        t = open(pre_g + "_timing.tsv").read().split("\t")
        assert len(t) == 3 and float(t[0]) == float(m["penalty"]) and float(t[2]) > 0
    # under / over columns as R fills them (NA in iteration 1)
    assert np.isnan(fit.others["under"].iloc[0]) and np.isnan(fit.others["over"].iloc[1])
    assert list(fit.others["under"].iloc[2:5]) == [0, 0, 17]
    # a second search is served from the cache: same answer, no dynamic program
    from peaksegdisk_amd import _native
    rows = (_native.PsdSearchRow * 64)()
    n = ctypes.c_int()
    chosen = ctypes.c_int()
    assert _native.lib.PeakSegFPOP_sequential_search(os.fsencode(str(gdir)), 19, 0, 64, rows,
                                                     ctypes.byref(n), ctypes.byref(chosen)) == 0
    assert n.value == len(trace) and all(rows[k].cached for k in range(n.value))
    assert rows[chosen.value].peaks == 19
    # target above the maximum: the reference's message
    with pytest.raises(ValueError, match="peaks.int=300000 but max=259999 peaks for N=520000"):
        psd.sequentialSearch_dir(str(gdir), 300000)


@GPU
def test_search_batch_equals_single_searches(psd, tmp_path, n_bins=4000, with_mono=True):
    """sequentialSearch_dir_batch: several directories searched together -- the models of one
    iteration in one launch -- give, directory by directory, what sequentialSearch_dir gives on
    a copy of that directory alone: the same penalties in the same order (strings), the same
    rows, byte-identical files.  Targets differ per directory, so the searches end after
    different numbers of iterations (one directory is Mono27ac with its known 19-peak trace)."""
    from peaksegdisk_amd import synthetic
    specs = [("mono", None, 19), ("s1", 81, 7), ("s2", 82, 2), ("s3", 83, 0), ("s4", 84, 11)]
    if not with_mono:  # the emulator rehearsal: small contigs only
        specs = specs[1:4]
    together, alone, targets = [], [], []
    for name, seed, target in specs:
        for root, acc in ((tmp_path / "together", together), (tmp_path / "alone", alone)):
            d = root / name
            d.mkdir(parents=True)
            if seed is None:
                shutil.copy(os.path.join(GOLDEN, "Mono27ac.bedGraph"), str(d / "coverage.bedGraph"))
            else:
                cs, ce, cnt = synthetic.poisson_coverage(n_bins + 500 * (seed - 80), seed=seed)
                synthetic.write_bedgraph(str(d / "coverage.bedGraph"), cs, ce, cnt)
            acc.append(str(d))
        targets.append(target)
    fits = psd.sequentialSearch_dir_batch(together, targets)
    assert len(fits) == len(specs)
    n_models = []
    for d_t, d_a, target, fit in zip(together, alone, targets, fits):
        ref = psd.sequentialSearch_dir(d_a, target)
        cols = ["penalty", "segments", "peaks", "bases", "total.loss", "iteration", "under", "over"]
        a, b = fit.others[cols], ref.others[cols]
        assert a.shape == b.shape
        assert [psd.paste(float(x)) for x in a["penalty"]] == [psd.paste(float(x)) for x in b["penalty"]]
        assert a.fillna(-1).values.tolist() == b.fillna(-1).values.tolist()
        assert int(fit.loss["peaks"].iloc[0]) == int(ref.loss["peaks"].iloc[0])
        assert psd.paste(float(fit.loss["penalty"].iloc[0])) == psd.paste(float(ref.loss["penalty"].iloc[0]))
        for x in a["penalty"]:
            pen = psd.paste(float(x))
            for suffix in ("_segments.bed", "_loss.tsv"):
                f_t = "%s_penalty=%s%s" % (os.path.join(d_t, "coverage.bedGraph"), pen, suffix)
                f_a = "%s_penalty=%s%s" % (os.path.join(d_a, "coverage.bedGraph"), pen, suffix)
                assert open(f_t, "rb").read() == open(f_a, "rb").read()
        n_models.append(len(a))
    if with_mono:
        assert int(fits[0].loss["peaks"].iloc[0]) == 19 and n_models[0] == 11
    assert len(set(n_models)) > 1  # the searches did not all take the same number of rounds
    # one target for all, everything served from the cache now
    again = psd.sequentialSearch_dir_batch(together[-3:-1], 2)
    assert [int(f.loss["peaks"].iloc[0]) <= 2 for f in again] == [True, True]
    # a directory listed twice is refused (two searches would write the same files at once)
    with pytest.raises(psd.PeakSegError) as e:
        psd.sequentialSearch_dir_batch([together[-2], together[-2]], 2)
    assert e.value.status == 15
    # a directory without data fails alone, with the reference's status
    bad = tmp_path / "together" / "missing"
    bad.mkdir()
    with pytest.raises(psd.PeakSegError) as e:
        psd.sequentialSearch_dir_batch([together[-2], str(bad)], 2)
    assert e.value.status == 3


@GPU
def test_dir_batch_cache_and_timing(psd, oracle_det, tmp_path, n_bins=3000):
    """PeakSegFPOP_dir_batch: first call solves everything in one launch and writes
    _timing.tsv; the second is all cache hits and leaves the files untouched; a damaged file
    is recomputed (test-CRAN-PeakSegFPOP_dir.R:38-57)."""
    from peaksegdisk_amd import synthetic
    dirs, pens = [], []
    for c in range(3):
        d = tmp_path / ("prob%d" % c)
        d.mkdir()
        cs, ce, cnt = synthetic.poisson_coverage(n_bins + n_bins // 6 * c, seed=70 + c)
        synthetic.write_bedgraph(str(d / "coverage.bedGraph"), cs, ce, cnt)
        for pen in ("0", "12.5", "Inf", "4000"):
            dirs.append(str(d))
            pens.append(pen)
    fits = psd.PeakSegFPOP_dir_batch(dirs, pens)
    assert not any(f.cached for f in fits)
    for d, pen, f in zip(dirs, pens, fits):
        bg = os.path.join(d, "coverage.bedGraph")
        pre = "%s_penalty=%s" % (bg, pen)
        obg = str(tmp_path / "o.bedGraph")
        shutil.copy(bg, obg)
        assert oracle_det.solve(obg, pen) == 0
        assert open(pre + "_loss.tsv", "rb").read() == \
            open("%s_penalty=%s_loss.tsv" % (obg, pen), "rb").read()
        assert open(pre + "_segments.bed", "rb").read() == \
            open("%s_penalty=%s_segments.bed" % (obg, pen), "rb").read()
        odb = "%s_penalty=%s.db" % (obg, pen)
        mb = os.path.getsize(odb) / 1024 / 1024 if os.path.exists(odb) else 0
        assert float(f.loss["megabytes"].iloc[0]) == pytest.approx(mb, rel=1e-12)
        assert not os.path.exists(pre + ".db")
        if os.path.exists(odb):
            os.unlink(odb)
    stamp = {p: os.stat(p).st_mtime_ns for d in set(dirs) for p in
             [os.path.join(d, x) for x in os.listdir(d)]}
    again = psd.PeakSegFPOP_dir_batch(dirs, pens)
    assert all(f.cached for f in again)
    assert stamp == {p: os.stat(p).st_mtime_ns for p in stamp}
    open("%s_penalty=%s_loss.tsv" % (os.path.join(dirs[1], "coverage.bedGraph"), pens[1]),
         "w").close()
    third = psd.PeakSegFPOP_dir_batch(dirs, pens)
    assert [f.cached for f in third] == [i != 1 for i in range(len(dirs))]
    assert third[1].loss["peaks"].iloc[0] == fits[1].loss["peaks"].iloc[0]


def _build_variant(tmp_path, name, flags):
    import __graft_entry__ as entry
    from peaksegdisk_amd import _native
    csrc = os.path.join(ROOT, "peaksegdisk_amd", "csrc")
    lib_path = str(tmp_path / ("libpeaksegdisk_hip_%s.so" % name))
    subprocess.run([entry.HIPCC] + entry.HIP_FLAGS + flags + [
                    "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                    os.path.join(csrc, "peakseg_hip.cpp"), "-o", lib_path], check=True)
    return _native.declare(ctypes.CDLL(lib_path))


@GPU
def test_newton_step_cap_fallback_on_device(psd, tmp_path):
    """The root finders' 100-step fallback (fpl:109-120,170-183; larger_root_full /
    smaller_root_full in fpop_pieces.h) never triggers on real data.  Library and oracle built
    with the cap lowered to 5 steps, so that most solves take it, compared on the device: both
    kernel builds, whole store byte for byte."""
    from conftest import Oracle
    from peaksegdisk_amd import ProblemSet, synthetic
    lib5 = _build_variant(tmp_path, "steps5", ["-DPSD_NEWTON_STEPS=5"])
    oracle5 = Oracle("det_steps5")
    oracle100 = Oracle("det")
    cs, ce, cnt = synthetic.poisson_coverage(6000, seed=21)
    pens = ["0.5", "40", "3000", "70000"]
    bg = str(tmp_path / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    want, differs = [], False
    for i, pen in enumerate(pens):
        db5 = str(tmp_path / ("o5_%d.db" % i))
        assert oracle5.solve(bg, pen, db5) == 0
        want.append(open(db5, "rb").read())
        db100 = str(tmp_path / ("o100_%d.db" % i))
        assert oracle100.solve(bg, pen, db100) == 0
        differs = differs or open(db100, "rb").read() != want[-1]
    assert differs, "cap of 5 steps did not change anything: fallback not exercised"
    for build in ("lat", "thr", "pk"):
        os.environ["PEAKSEG_HIP_VARIANT"] = build
        try:
            pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens],
                              lib=lib5)
            pset.solve()
            assert pset.kernel_build == build
            for i, pen in enumerate(pens):
                assert pset.result(i).status == 0
                dbg = str(tmp_path / "g5.db")
                pset.export_db(i, ce, dbg)
                assert open(dbg, "rb").read() == want[i], (build, pen)
            pset.close()
        finally:
            del os.environ["PEAKSEG_HIP_VARIANT"]


@GPU
def test_spill_pool_and_arena_regrowth(psd, oracle_det, tmp_path, monkeypatch, n_bins=2500):
    """More problems spill at once than the pool has slots, and the arena estimate is far too
    small: the set is rerun with a larger pool / arena and still equals the oracle."""
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.increasing_coverage(n_bins)
    pens = ["100", "300", "1000", "30"]
    monkeypatch.setenv("PEAKSEG_HIP_SPILL_SLOTS", "1")
    monkeypatch.setenv("PEAKSEG_HIP_PIECES_PER_FUNCTION", "1")
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens])
    pset.solve()
    bg = str(tmp_path / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    n_spilled = 0
    for i, pen in enumerate(pens):
        r = pset.result(i)
        assert r.status == 0
        n_spilled += 1 if r.spill_steps > 0 else 0
        db_o = str(tmp_path / ("o%d.db" % i))
        assert oracle_det.solve(bg, pen, db_o) == 0
        db_g = str(tmp_path / ("g%d.db" % i))
        pset.export_db(i, ce, db_g)
        assert open(db_g, "rb").read() == open(db_o, "rb").read(), pen
    assert n_spilled >= 2, "the one-slot pool was never exhausted"
    assert pset.arena_bytes_used > 2 * n_bins * 4 * 20  # far beyond one piece per function
    pset.close()
    # an explicit cap that cannot hold the tables is refused with the memory status
    monkeypatch.setenv("PEAKSEG_HIP_MAX_BYTES", "4K")
    with pytest.raises(RuntimeError, match="status 14"):
        ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, 100.0)])


_WORKER = r"""
import os, sys, shutil
sys.path.insert(0, %(root)r)
import peaksegdisk_amd as psd
d = sys.argv[1]
fit = psd.PeakSegFPOP_dir(d, "1952.6")
print(int(fit.loss["peaks"].iloc[0]), int(fit.loss["segments"].iloc[0]))
"""


@GPU
def test_several_processes_share_one_gpu(psd, tmp_path):
    """R's future workers are separate processes on one device (SURVEY.md section 8b,
    "Threading"): four processes solve Mono27ac concurrently, each capped by
    PEAKSEG_HIP_MAX_BYTES, and all get the known answer."""
    procs = []
    env = dict(os.environ, PEAKSEG_HIP_MAX_BYTES="512M")
    for k in range(4):
        d = tmp_path / ("w%d" % k)
        d.mkdir()
        shutil.copy(os.path.join(GOLDEN, "Mono27ac.bedGraph"), str(d / "coverage.bedGraph"))
        procs.append(subprocess.Popen([sys.executable, "-c", _WORKER % {"root": ROOT}, str(d)],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                                      text=True))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-2000:]
        assert out.split() == ["17", "35"]


@GPU
def test_batch_larger_than_the_descriptor_limit(psd, oracle_det, tmp_path):
    """PeakSegFPOP_disk_batch with 700 problems (the usual soft limit is 1024 descriptors and
    the batch used to hold two per problem): every status 0, spot checks against the oracle."""
    from peaksegdisk_amd import _native, synthetic
    n = 700
    files, pens = [], []
    for c in range(n // 7):
        cs, ce, cnt = synthetic.poisson_coverage(300 + c, seed=300 + c)
        bg = str(tmp_path / ("c%d.bedGraph" % c))
        synthetic.write_bedgraph(bg, cs, ce, cnt)
        for pen in ("0", "0.5", "7", "60", "500", "4000", "Inf"):
            files.append(bg)
            pens.append(pen)
    a = (ctypes.c_char_p * n)(*[os.fsencode(f) for f in files])
    b = (ctypes.c_char_p * n)(*[p.encode() for p in pens])
    dbs = ["%s_penalty=%s.db" % (f, p) for f, p in zip(files, pens)]
    c = (ctypes.c_char_p * n)(*[os.fsencode(d) for d in dbs])
    status = (ctypes.c_int * n)()
    assert _native.lib.PeakSegFPOP_disk_batch(n, a, b, c, status) == 0
    assert not any(status)
    for k in range(0, n, 53):
        obg = str(tmp_path / "o.bedGraph")
        shutil.copy(files[k], obg)
        assert oracle_det.solve(obg, pens[k]) == 0
        for suffix in ("_segments.bed", "_loss.tsv"):
            assert open("%s_penalty=%s%s" % (files[k], pens[k], suffix), "rb").read() == \
                open("%s_penalty=%s%s" % (obg, pens[k], suffix), "rb").read()


@GPU
def test_throughput_set_every_contig_vs_oracle(psd, tmp_path):
    """A set that picks the throughput build by itself (40 contigs x 8 penalties = 320
    problems): EVERY problem against the oracle's files (round 1 compared one contig)."""
    from peaksegdisk_amd import ProblemSet, synthetic
    n_contigs, n_bins = 40, 2500
    pens = ["0.2", "3", "45", "600", "2500", "8000", "30000", "90000"]
    contigs, data = [], []
    for k in range(n_contigs):
        cs, ce, cnt = synthetic.poisson_coverage(n_bins + 37 * k, seed=500 + k)
        contigs.append((cnt, (ce - cs).astype(np.int32)))
        data.append((cs, ce, cnt))
    problems = [(k, float(p)) for k in range(n_contigs) for p in pens]
    pset = ProblemSet(contigs, problems)
    pset.solve()
    assert "thr" in pset.kernel_build  # "lat+thr" when the planner mixes

    def oracle_job(k):
        cs, ce, cnt = data[k]
        bg = str(tmp_path / ("c%d.bedGraph" % k))
        synthetic.write_bedgraph(bg, cs, ce, cnt)
        for pen in pens:
            run_cli(CLI_DET, bg, pen, bg + ".db")
        return bg
    with ThreadPoolExecutor(max_workers=len(os.sched_getaffinity(0))) as pool:
        bgs = list(pool.map(oracle_job, range(n_contigs)))
    for k in range(n_contigs):
        cs, ce, cnt = data[k]
        for j, pen in enumerate(pens):
            i = k * len(pens) + j
            r = pset.result(i)
            assert r.status == 0
            segs = read_segments("%s_penalty=%s_segments.bed" % (bgs[k], pen))
            start, mean = pset.segments(i)
            assert [s[1] for s in segs] == [int(cs[0]) if q < 0 else int(ce[q]) for q in start]
            assert [s[4] for s in segs] == ["%g" % v for v in mean]
            loss = read_loss("%s_penalty=%s_loss.tsv" % (bgs[k], pen)).split("\t")
            assert loss[5] == "%.20g" % r.best_cost and float(loss[9]) == r.max_intervals
            assert float(loss[8]) == r.total_intervals / (2.0 * len(cnt))
    pset.close()


@GPU
def test_checkpointed_store_equals_full_store(psd, oracle_det, tmp_path, monkeypatch, n_bins=20000,
                                              intervals=(16, 100, 2048, 50000)):
    """with PEAKSEG_HIP_CHECKPOINT=K the forward pass keeps a checkpoint
    every K data points instead of every cost function, and the decoding recomputes the blocks
    it walks through.  Segment tables, losses and interval statistics must equal the full
    store's bit for bit (and the oracle's files), for block sizes below, around and above the
    segment lengths, in both kernel builds; the HBM held shrinks accordingly."""
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=11)
    w = (ce - cs).astype(np.int32)
    pens = ["0", "0.3", "25", "800", "20000", "1000000"]
    problems = [(0, float(p)) for p in pens]
    monkeypatch.setenv("PEAKSEG_HIP_NO_CHECKPOINT", "1")
    full = ProblemSet([(cnt, w)], problems)
    full.solve()
    assert full.checkpoint_interval == 0
    want = []
    for i in range(len(pens)):
        r = full.result(i)
        assert r.status == 0
        s0, m0 = full.segments(i)
        want.append((r.n_segments, r.n_equality_constraints, r.max_intervals, r.total_intervals,
                     r.best_cost, s0.copy(), m0.copy()))
    full_bytes = full.hbm_bytes
    full.close()
    monkeypatch.delenv("PEAKSEG_HIP_NO_CHECKPOINT")
    bg = str(tmp_path / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    for K in intervals:
        for build in ("lat", "thr"):
            monkeypatch.setenv("PEAKSEG_HIP_CHECKPOINT", str(K))
            monkeypatch.setenv("PEAKSEG_HIP_VARIANT", build)
            ck = ProblemSet([(cnt, w)], problems)
            ck.solve()
            assert ck.checkpoint_interval == K and ck.kernel_build == build
            # the regions' default size must hold every block: a relaunch here means solving a
            # problem twice (round 3: regions of 14 pieces per function doubled config 4's time)
            assert ck.solve_stats[0] == 1, (K, build, ck.solve_stats)
            for i in range(len(pens)):
                r = ck.result(i)
                assert r.status == 0, (K, build, pens[i], r.kernel_status)
                s1, m1 = ck.segments(i)
                got = (r.n_segments, r.n_equality_constraints, r.max_intervals, r.total_intervals,
                       r.best_cost)
                assert got == want[i][:5], (K, build, pens[i])
                assert np.array_equal(s1, want[i][5]), (K, build, pens[i])
                assert np.array_equal(m1.view(np.uint64), want[i][6].view(np.uint64))
            if K == 2048:
                assert ck.hbm_bytes < full_bytes
            ck.close()
    monkeypatch.delenv("PEAKSEG_HIP_CHECKPOINT")
    monkeypatch.delenv("PEAKSEG_HIP_VARIANT")
    # against the oracle's files as well (the full store was; this closes the loop)
    for i in (1, 3):
        assert oracle_det.solve(bg, pens[i]) == 0
        segs = read_segments("%s_penalty=%s_segments.bed" % (bg, pens[i]))
        assert [s[1] for s in segs] == [int(cs[0]) if q < 0 else int(ce[q]) for q in want[i][5]]


@GPU
def test_checkpointed_store_region_regrowth_and_limits(psd, tmp_path, monkeypatch, n_bins=6000,
                                                       auto_bins=200000, adv_bins=3000,
                                                       adv_builds=(("lat", ""), ("thr", ""),
                                                                   ("lat", "1024"))):
    """A block whose records outgrow the wave's region makes the host enlarge the regions and
    rerun (a clean error when the cap forbids it); functions that have outgrown LDS are
    checkpointed too (overflow pool) and recomputed from there; a memory cap that the full
    store would exceed selects the checkpointed store by itself."""
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=12)
    w = (ce - cs).astype(np.int32)
    monkeypatch.setenv("PEAKSEG_HIP_NO_CHECKPOINT", "1")
    full = ProblemSet([(cnt, w)], [(0, 500.0)])
    full.solve()
    want = full.segments(0)
    full.close()
    monkeypatch.delenv("PEAKSEG_HIP_NO_CHECKPOINT")
    monkeypatch.setenv("PEAKSEG_HIP_CHECKPOINT", "256")
    monkeypatch.setenv("PEAKSEG_HIP_PIECES_PER_FUNCTION", "1")  # regions far too small
    ck = ProblemSet([(cnt, w)], [(0, 500.0)])
    ck.solve()
    got = ck.segments(0)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint64),
                                                              want[1].view(np.uint64))
    ck.close()
    monkeypatch.delenv("PEAKSEG_HIP_PIECES_PER_FUNCTION")
    # adversarial data: functions of hundreds of pieces at the checkpoints (lists in the HBM
    # spill area): they are checkpointed into the overflow pool, the blocks that hold segment
    # ends start from there in HBM mode, and the result equals the full store's.  Both builds;
    # an overflow pool that is too small at first (one rerun with four times the pool).
    c2s, c2e, c2 = synthetic.increasing_coverage(adv_bins)
    w2 = (c2e - c2s).astype(np.int32)
    monkeypatch.delenv("PEAKSEG_HIP_CHECKPOINT")
    monkeypatch.setenv("PEAKSEG_HIP_NO_CHECKPOINT", "1")
    full = ProblemSet([(c2, w2)], [(0, 100.0), (0, 2.0)])
    full.solve()
    want2 = [(full.result(i).max_intervals, full.result(i).total_intervals, full.result(i).best_cost)
             + full.segments(i) for i in range(2)]
    assert want2[0][0] > 128 and len(want2[0][3]) > 3
    full.close()
    monkeypatch.delenv("PEAKSEG_HIP_NO_CHECKPOINT")
    monkeypatch.setenv("PEAKSEG_HIP_CHECKPOINT", "256")
    for build, pool in adv_builds:
        monkeypatch.setenv("PEAKSEG_HIP_VARIANT", build)
        if pool:
            monkeypatch.setenv("PEAKSEG_HIP_CKPT_OVERFLOW", pool)
        adv = ProblemSet([(c2, w2)], [(0, 100.0), (0, 2.0)])
        adv.solve()
        assert adv.checkpoint_interval == 256 and adv.kernel_build == build
        for i in range(2):
            r = adv.result(i)
            assert r.status == 0, (build, i, r.kernel_status)
            assert (r.max_intervals, r.total_intervals, r.best_cost) == want2[i][:3]
            s1, m1 = adv.segments(i)
            assert np.array_equal(s1, want2[i][3])
            assert np.array_equal(m1.view(np.uint64), want2[i][4].view(np.uint64))
        adv.close()
    monkeypatch.delenv("PEAKSEG_HIP_VARIANT")
    monkeypatch.delenv("PEAKSEG_HIP_CKPT_OVERFLOW", raising=False)
    # Regrowth that the memory cap cannot hold must end with a clean error, never with an arena
    # smaller than the regions the kernel indexes: the same adversarial data, a
    # cap just above what the set holds after its first, far too small, allocation.
    monkeypatch.setenv("PEAKSEG_HIP_PIECES_PER_FUNCTION", "1")
    # (blocks of 2048 data points: the regions the decoding needs, 2049 functions of some 250
    # pieces per chain, are larger than the smallest arena the library maps)
    monkeypatch.setenv("PEAKSEG_HIP_CHECKPOINT", "2048" if adv_bins > 2500 else "256")
    # (arena blocks of 4096 pieces instead of the usual 2^19 and more, so that the regrown
    # regions need memory the first allocation did not already hold)
    monkeypatch.setenv("PEAKSEG_HIP_ARENA_BLOCK_LOG2", "12")
    probe = ProblemSet([(c2, w2)], [(0, 100.0)])
    held = probe.hbm_bytes
    probe.close()
    monkeypatch.setenv("PEAKSEG_HIP_MAX_BYTES", str(held + 4096))
    capped = ProblemSet([(c2, w2)], [(0, 100.0)])
    with pytest.raises(RuntimeError, match="do not fit|cannot grow"):
        capped.solve()
    capped.close()
    monkeypatch.delenv("PEAKSEG_HIP_MAX_BYTES")
    monkeypatch.delenv("PEAKSEG_HIP_PIECES_PER_FUNCTION")
    monkeypatch.delenv("PEAKSEG_HIP_CHECKPOINT")
    monkeypatch.delenv("PEAKSEG_HIP_ARENA_BLOCK_LOG2")
    if not auto_bins:
        return
    # automatic choice under a cap: 200 k bins x 4 penalties need ~280 MB in full (estimate:
    # 348 B per bin and penalty), about 40 MB checkpointed
    cs, ce, cnt = synthetic.poisson_coverage(auto_bins, seed=13)
    w = (ce - cs).astype(np.int32)
    problems = [(0, p) for p in (0.5, 30.0, 900.0, 40000.0)]
    full = ProblemSet([(cnt, w)], problems)
    assert full.checkpoint_interval == 0
    full.solve()
    want = [full.segments(i) for i in range(4)]
    full_bytes = full.hbm_bytes
    full.close()
    monkeypatch.setenv("PEAKSEG_HIP_MAX_BYTES", "100M")
    auto = ProblemSet([(cnt, w)], problems)
    assert auto.checkpoint_interval > 0 and auto.hbm_bytes < 100 * 2 ** 20 < full_bytes
    auto.solve()
    for i in range(4):
        got = auto.segments(i)
        assert np.array_equal(got[0], want[i][0])
        assert np.array_equal(got[1].view(np.uint64), want[i][1].view(np.uint64))
    auto.close()


@GPU
def test_mixed_launch_of_unequal_contigs(psd, tmp_path, monkeypatch, n_long=6000, n_short=300,
                                         n_short_contigs=70):
    """A set that oversubscribes the chip with contigs of very different lengths: the longest
    problems run on the latency build (a CU each), the rest packed on the throughput build, in
    one solve ("lat+thr").  Every problem against the oracle's files; the same set forced onto
    one build gives the same bits."""
    from peaksegdisk_amd import ProblemSet, synthetic
    monkeypatch.delenv("PEAKSEG_HIP_VARIANT", raising=False)
    lens = [n_long, n_long + 500] + [n_short + 3 * k for k in range(n_short_contigs)]
    pens = ["0.4", "12", "300", "9000"]
    contigs, data = [], []
    for k, n in enumerate(lens):
        cs, ce, cnt = synthetic.poisson_coverage(n, seed=900 + k)
        contigs.append((cnt, (ce - cs).astype(np.int32)))
        data.append((cs, ce, cnt))
    problems = [(k, float(p)) for k in range(len(lens)) for p in pens]
    assert len(problems) > 256
    pset = ProblemSet(contigs, problems)
    pset.solve()
    assert pset.kernel_build == "lat+thr"

    def oracle_job(k):
        cs, ce, cnt = data[k]
        bg = str(tmp_path / ("c%d.bedGraph" % k))
        synthetic.write_bedgraph(bg, cs, ce, cnt)
        for pen in pens:
            run_cli(CLI_DET, bg, pen, bg + ".db")
        return bg
    with ThreadPoolExecutor(max_workers=len(os.sched_getaffinity(0))) as pool:
        bgs = list(pool.map(oracle_job, range(len(lens))))
    monkeypatch.setenv("PEAKSEG_HIP_VARIANT", "thr")
    one = ProblemSet(contigs, problems)
    one.solve()
    assert one.kernel_build == "thr"
    for k in range(len(lens)):
        cs, ce, cnt = data[k]
        for j, pen in enumerate(pens):
            i = k * len(pens) + j
            r = pset.result(i)
            assert r.status == 0
            segs = read_segments("%s_penalty=%s_segments.bed" % (bgs[k], pen))
            start, mean = pset.segments(i)
            assert [s[1] for s in segs] == [int(cs[0]) if q < 0 else int(ce[q]) for q in start]
            assert [s[4] for s in segs] == ["%g" % v for v in mean]
            loss = read_loss("%s_penalty=%s_loss.tsv" % (bgs[k], pen)).split("\t")
            assert loss[5] == "%.20g" % r.best_cost and float(loss[9]) == r.max_intervals
            s1, m1 = one.segments(i)
            assert np.array_equal(start, s1) and np.array_equal(mean.view(np.uint64),
                                                                m1.view(np.uint64))
    pset.close()
    one.close()
