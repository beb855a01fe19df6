"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on an
MI355X.  Integer fields, segment rows and whole output files must be bit/byte identical to
oracle/_build/liboracle_det.so (same deterministic exp/log); against the libm build --
the arithmetic the reference itself uses -- endpoints/states must be identical and the
Poisson loss within 1e-6 relative (BASELINE.json north_star)."""
import ctypes
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN, read_loss, read_segments

GPU = pytest.mark.gpu
REL_TOL = 1e-6


@pytest.fixture(scope="module")
def psd():
    import __graft_entry__ as entry
    entry.build_hip()
    import peaksegdisk_amd
    from peaksegdisk_amd import _native
    assert _native.lib.peakseg_hip_device_count() >= 1, "no HIP device: GPU tests need an MI355X"
    return peaksegdisk_amd


def _solve_hip(psd, bg, pen, db=None):
    from peaksegdisk_amd import _native
    if db is None:
        db = "%s_penalty=%s.db" % (bg, pen)
    return _native.lib.PeakSegFPOP_disk(os.fsencode(bg), pen.encode(), os.fsencode(db))


def _files(bg, pen):
    pre = "%s_penalty=%s" % (bg, pen)
    return open(pre + "_segments.bed", "rb").read(), open(pre + "_loss.tsv", "rb").read()


@GPU
def test_device_math_is_bit_identical_to_host(psd, oracle_det):
    from peaksegdisk_amd import _native
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-30, 30, 200000), rng.uniform(-745, 709, 50000),
                         rng.uniform(-1e-5, 1e-5, 10000),
                         np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 709.78, -745.1, 1e-320])])
    pos = np.concatenate([rng.uniform(0, 100, 200000), np.exp(rng.uniform(-700, 700, 50000)),
                          1 + rng.uniform(-1e-3, 1e-3, 10000),
                          np.array([0.0, 1.0, np.inf, -1.0, np.nan, 5e-324, 2.2e-308])])
    for op, x in ((0, xs), (1, pos)):
        x = np.ascontiguousarray(x)
        y = np.empty_like(x)
        assert _native.lib.peakseg_hip_math_probe(op, x.size, x.ctypes.data, y.ctypes.data) == 0
        want = oracle_det.math("exp" if op == 0 else "log", x)
        assert np.array_equal(y.view(np.uint64), want.view(np.uint64))


@GPU
def test_known_answers_byte_identical(psd, oracle_det, known_answers, tmp_path):
    for i, case in enumerate(known_answers["solve_cases"]):
        g = tmp_path / ("g%d" % i)
        o = tmp_path / ("o%d" % i)
        g.mkdir()
        o.mkdir()
        gb, ob = str(g / "coverage.bedGraph"), str(o / "coverage.bedGraph")
        open(gb, "w").write(case["bedGraph"])
        open(ob, "w").write(case["bedGraph"])
        assert _solve_hip(psd, gb, case["penalty"]) == case["status"], case["name"]
        assert oracle_det.solve(ob, case["penalty"]) == case["status"]
        assert _files(gb, case["penalty"]) == _files(ob, case["penalty"]), case["name"]
        if "segments" in case:
            segs = read_segments("%s_penalty=%s_segments.bed" % (gb, case["penalty"]))
            assert segs == [list(r) for r in case["segments"]]
        if "db_bytes" in case:
            assert os.path.getsize(gb + "_penalty=%s.db" % case["penalty"]) == case["db_bytes"]


@GPU
def test_error_cases(psd, known_answers, tmp_path):
    from peaksegdisk_amd import _native
    for i, case in enumerate(known_answers["error_cases"]):
        d = tmp_path / ("e%d" % i)
        d.mkdir()
        bg = str(d / "missing") if case["bedGraph"] is None else str(d / "coverage.bedGraph")
        if case["bedGraph"] is not None:
            open(bg, "w").write(case["bedGraph"])
        pre = "%s_penalty=%s" % (bg, case["penalty"])
        db = pre + ".db"
        if case.get("block") == "segments":
            os.mkdir(pre + "_segments.bed")
        elif case.get("block") == "loss":
            os.mkdir(pre + "_loss.tsv")
        elif case.get("block") == "db":
            os.mkdir(db)
        st = _solve_hip(psd, bg, case["penalty"], db)
        assert st == case["status"], case["name"]
        assert case["message"] in _native.status_message(st, bg, case["penalty"], db)


MONO_PENALTIES = ["1952.6", "0", "10000", "Inf", "157.994737329317", "1952.66876946418",
                  "313.236999786254", "605.365673153523", "1094.30971435653", "1548.04195002439",
                  "1752.96395944988", "1694.47781380514", "1715.84956360692"]


@GPU
def test_mono27ac_files_and_store_identical(psd, oracle_det, oracle_libm, known_answers, tmp_path):
    from peaksegdisk_amd import _native
    dirs = {}
    for name in ("gpu", "det", "libm"):
        d = tmp_path / name
        d.mkdir()
        shutil.copy(os.path.join(GOLDEN, "Mono27ac.bedGraph"), str(d / "coverage.bedGraph"))
        dirs[name] = str(d / "coverage.bedGraph")
    n = len(MONO_PENALTIES)
    arr = lambda xs: (ctypes.c_char_p * n)(*[x.encode() for x in xs])
    status = (ctypes.c_int * n)()
    st = _native.lib.PeakSegFPOP_disk_batch(
        n, arr([dirs["gpu"]] * n), arr(MONO_PENALTIES),
        arr(["%s_penalty=%s.db" % (dirs["gpu"], p) for p in MONO_PENALTIES]), status)
    assert st == 0 and list(status) == [0] * n, _native.last_error()
    want_peaks = dict(zip(known_answers["mono27ac"]["sequential_search_19"]["penalties"],
                          known_answers["mono27ac"]["sequential_search_19"]["peaks"]))
    for pen in MONO_PENALTIES:
        assert oracle_det.solve(dirs["det"], pen) == 0
        assert oracle_libm.solve(dirs["libm"], pen) == 0
        g_seg, g_loss = _files(dirs["gpu"], pen)
        assert (g_seg, g_loss) == _files(dirs["det"], pen), pen
        l_seg, l_loss = _files(dirs["libm"], pen)
        assert g_seg == l_seg, pen  # endpoints, states, 6-digit means vs the libm arithmetic
        gl, ll = g_loss.decode().split("\t"), l_loss.decode().split("\t")
        assert gl[1:5] == ll[1:5] and gl[7] == ll[7]
        assert float(gl[6]) == pytest.approx(float(ll[6]), rel=REL_TOL)
        if pen in want_peaks:
            assert int(gl[2]) == want_peaks[pen]
        if pen != "Inf":  # sparse db file has the size of the reference's store
            assert os.path.getsize("%s_penalty=%s.db" % (dirs["gpu"], pen)) == \
                os.path.getsize("%s_penalty=%s.db" % (dirs["det"], pen))


@GPU
def test_arena_equals_oracle_db(psd, oracle_det, tmp_path):
    """Every stored function (max_log_mean, data_i, prev_log_mean per piece, for both chains
    and every data point) equals the oracle's DiskVector file byte for byte."""
    from peaksegdisk_amd import ProblemSet
    cov = np.loadtxt(os.path.join(GOLDEN, "Mono27ac.bedGraph"), usecols=(1, 2, 3), dtype=np.int64)
    cs, ce, cnt = cov[:, 0], cov[:, 1], cov[:, 2]
    pens = ["1952.6", "0.5", "31.7"]
    pset = ProblemSet([(cnt.astype(np.int32), (ce - cs).astype(np.int32))],
                      [(0, float(p)) for p in pens])
    pset.solve()
    bg = str(tmp_path / "coverage.bedGraph")
    shutil.copy(os.path.join(GOLDEN, "Mono27ac.bedGraph"), bg)
    for i, pen in enumerate(pens):
        db_o = str(tmp_path / ("oracle_%d.db" % i))
        assert oracle_det.solve(bg, pen, db_o) == 0
        db_g = str(tmp_path / ("gpu_%d.db" % i))
        pset.export_db(i, ce, db_g)
        assert open(db_g, "rb").read() == open(db_o, "rb").read(), pen
    pset.close()


@GPU
@pytest.mark.parametrize("n_bins,seed", [(20000, 11), (3000, 12)])
def test_synthetic_grid_vs_oracle(psd, oracle_det, oracle_libm, tmp_path, n_bins, seed):
    from peaksegdisk_amd import _native, synthetic
    cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=seed)
    pens = synthetic.penalty_grid(16)
    paths = {}
    for name in ("gpu", "det", "libm"):
        d = tmp_path / name
        d.mkdir()
        paths[name] = str(d / "coverage.bedGraph")
        synthetic.write_bedgraph(paths[name], cs, ce, cnt)
    n = len(pens)
    arr = lambda xs: (ctypes.c_char_p * n)(*[x.encode() for x in xs])
    status = (ctypes.c_int * n)()
    st = _native.lib.PeakSegFPOP_disk_batch(
        n, arr([paths["gpu"]] * n), arr(pens),
        arr(["%s_penalty=%s.db" % (paths["gpu"], p) for p in pens]), status)
    assert st == 0 and list(status) == [0] * n, _native.last_error()
    for pen in pens:
        assert oracle_det.solve(paths["det"], pen) == 0
        assert _files(paths["gpu"], pen) == _files(paths["det"], pen), pen
    for pen in pens[::5]:
        assert oracle_libm.solve(paths["libm"], pen) == 0
        g_seg, g_loss = _files(paths["gpu"], pen)
        l_seg, l_loss = _files(paths["libm"], pen)
        assert g_seg == l_seg
        assert float(g_loss.split(b"\t")[6]) == pytest.approx(float(l_loss.split(b"\t")[6]),
                                                              rel=REL_TOL)


@GPU
def test_python_entry_points(psd, tmp_path, with_search=True):
    """PeakSegFPOP_vec / _df / _dir / sequentialSearch_dir through the HIP library, with the
    reference tests' expectations (test-CRAN-PeakSegFPOP_vec.R, test-TRAVIS-sequentialSearch.R)."""
    fit_inf = psd.PeakSegFPOP_vec(np.array([1, 3, 0, 4, 2], dtype=np.int32), float("inf"))
    assert len(fit_inf.segments) == 1
    fit0 = psd.PeakSegFPOP_vec(np.array([1, 3, 0, 4, 2], dtype=np.int32), 0)
    assert len(fit0.segments) == 5
    d = tmp_path / "chr11-60000-580000"
    d.mkdir()
    shutil.copy(os.path.join(GOLDEN, "Mono27ac.bedGraph"), str(d / "coverage.bedGraph"))
    if not with_search:  # the emulator rehearsal runs the search in test_gpu_round2's test
        fit = psd.PeakSegFPOP_dir(str(d), "1952.6")
        assert int(fit.loss["peaks"].iloc[0]) == 17
        return
    fit = psd.sequentialSearch_dir(str(d), 19)
    assert int(fit.loss["peaks"].iloc[0]) == 19
    # the models in the order the reference visits them (test_gpu_round2 checks the penalties)
    assert list(fit.others["peaks"]) == [3199, 0, 224, 17, 74, 35, 25, 21, 18, 20, 19]


@GPU
@pytest.mark.parametrize("n_bins", [4000])
def test_adversarial_increasing_counts_spill(psd, oracle_det, tmp_path, n_bins):
    """vignettes/Worst_case.Rmd:26-28: count = 1..N makes the piece lists grow far beyond the
    128 pieces that fit in LDS, so the HBM spill path (and the way back to LDS) is exercised;
    results must still equal the oracle's byte for byte, including every stored function."""
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.increasing_coverage(n_bins)
    pens = ["100", "0", "10000"]
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens])
    pset.solve()
    bg = str(tmp_path / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    spilled = 0
    for i, pen in enumerate(pens):
        r = pset.result(i)
        assert r.status == 0
        spilled += r.spill_steps
        db_o = str(tmp_path / ("oracle_%d.db" % i))
        assert oracle_det.solve(bg, pen, db_o) == 0
        loss = read_loss("%s_penalty=%s_loss.tsv" % (bg, pen)).split("\t")
        assert int(loss[1]) == r.n_segments
        assert float(loss[9]) == r.max_intervals
        assert float(loss[8]) == r.total_intervals / (2.0 * n_bins)
        db_g = str(tmp_path / ("gpu_%d.db" % i))
        pset.export_db(i, ce, db_g)
        assert open(db_g, "rb").read() == open(db_o, "rb").read(), pen
        start, mean = pset.segments(i)
        segs = read_segments("%s_penalty=%s_segments.bed" % (bg, pen))
        assert len(segs) == len(start)
        assert [s[1] for s in segs] == [0 if k < 0 else int(ce[k]) for k in start]
    assert spilled > 0, "test did not reach the spill path"
    assert pset.result(0).max_intervals > 128
    pset.close()


@GPU
@pytest.mark.parametrize("n_bins,n_contigs", [(3000, 72)])
def test_throughput_build_identical(psd, oracle_det, tmp_path, monkeypatch, n_bins, n_contigs):
    """The library carries three builds of the forward kernel (peakseg_hip.cpp; the third, "pk", has
    its own test in test_gpu_round4.py): "lat" (helper
    waves, 128 pieces per LDS list) for sets that fit the chip at 2 workgroups per CU, "thr"
    (no helper waves, 64 pieces per LDS list, 4 workgroups per CU) beyond that.  Same results:
    a set large enough to pick "thr" by itself is compared with the same set forced onto
    "lat", and one contig's stored functions with the oracle's db byte for byte."""
    from peaksegdisk_amd import ProblemSet, synthetic
    contigs, ends = [], []
    for k in range(n_contigs):
        cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=50 + k)
        contigs.append((cnt, (ce - cs).astype(np.int32)))
        ends.append((cs, ce, cnt))
    pens = ["0.2", "3", "45", "600", "2500", "8000", "30000", "90000"]
    problems = [(k, float(p)) for k in range(n_contigs) for p in pens]
    monkeypatch.delenv("PEAKSEG_HIP_VARIANT", raising=False)
    auto = ProblemSet(contigs, problems)
    auto.solve()
    want = "thr" if len(problems) > 256 else "lat"
    assert auto.kernel_build == want
    monkeypatch.setenv("PEAKSEG_HIP_VARIANT", "lat" if want == "thr" else "thr")
    other = ProblemSet(contigs, problems)
    other.solve()
    assert other.kernel_build != auto.kernel_build
    monkeypatch.delenv("PEAKSEG_HIP_VARIANT")
    for p in range(len(problems)):
        ra, rb = auto.result(p), other.result(p)
        assert ra.status == 0 and rb.status == 0
        assert (ra.n_segments, ra.max_intervals, ra.total_intervals, ra.best_cost) == \
               (rb.n_segments, rb.max_intervals, rb.total_intervals, rb.best_cost)
        sa, ma = auto.segments(p)
        sb, mb = other.segments(p)
        assert np.array_equal(sa, sb) and np.array_equal(ma.view(np.uint64), mb.view(np.uint64))
    # contig 0 against the oracle, stored functions included, from both builds
    cs, ce, cnt = ends[0]
    bg = str(tmp_path / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    for i, pen in enumerate(pens):
        db_o = str(tmp_path / ("oracle_%d.db" % i))
        assert oracle_det.solve(bg, pen, db_o) == 0
        for name, ps in (("auto", auto), ("other", other)):
            db_g = str(tmp_path / ("%s_%d.db" % (name, i)))
            ps.export_db(i, ce, db_g)
            assert open(db_g, "rb").read() == open(db_o, "rb").read(), (name, pen)
    auto.close()
    other.close()


def _shaped_counts(kind, n, rng):
    if kind == 0:    # heavy-tailed counts
        cnt = np.floor(rng.pareto(1.2, n) * 3)
    elif kind == 1:  # long zero runs with bursts
        cnt = np.where(rng.random(n) < 0.9, 0, rng.integers(1, 2000, n))
    elif kind == 2:  # huge counts
        cnt = rng.integers(0, 2000000, n)
    elif kind == 3:  # slowly varying
        cnt = (50 + 40 * np.sin(np.arange(n) / 37.0) + rng.normal(0, 3, n)).clip(0)
    elif kind == 4:  # steps
        cnt = np.repeat(rng.integers(0, 100, n // 50 + 1), 50)[:n]
    else:            # 0/1 data
        cnt = rng.random(n) < 0.3
    return np.asarray(cnt).astype(np.int32)


def varied_shape_cases(n_cases, seed):
    """(count, width, chromStart, chromEnd, penalty strings) of the data shapes below; also
    solved by both oracle builds in tests/test_oracle_census.py"""
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        n = int(rng.integers(500, 4000))
        cnt = _shaped_counts(case % 6, n, rng)
        w = rng.integers(1, 500, n).astype(np.int64)
        ce = np.cumsum(w)
        cs = ce - w
        pens = ["0", "%.15g" % float(10 ** rng.uniform(-2, 6)),
                "%.15g" % float(10 ** rng.uniform(0, 4))]
        yield cnt, w, cs, ce, pens


VARIED_SHAPES_RUN = (18, 2024)


@GPU
@pytest.mark.parametrize("n_cases,seed", [VARIED_SHAPES_RUN])
def test_varied_data_shapes(psd, oracle_det, tmp_path, monkeypatch, n_cases, seed):
    """Coverage that does not look like the Poisson generator: heavy tails, long zero runs,
    counts in the millions, smooth ramps, steps, 0/1 data, random bin widths and penalties
    (0 included).  All three kernel builds; every stored function against the oracle's db."""
    from peaksegdisk_amd import ProblemSet, synthetic
    for case, (cnt, w, cs, ce, pens) in enumerate(varied_shape_cases(n_cases, seed)):
        bg = str(tmp_path / ("c%d.bedGraph" % case))
        synthetic.write_bedgraph(bg, cs, ce, cnt)
        want = []
        for i, pen in enumerate(pens):
            db_o = str(tmp_path / ("o_%d_%d.db" % (case, i)))
            assert oracle_det.solve(bg, pen, db_o) == 0
            want.append(open(db_o, "rb").read())
            os.unlink(db_o)
        for build in ("lat", "thr", "pk"):
            monkeypatch.setenv("PEAKSEG_HIP_VARIANT", build)
            pset = ProblemSet([(cnt, w.astype(np.int32))], [(0, float(p)) for p in pens])
            pset.solve()
            assert pset.kernel_build == build
            for i, pen in enumerate(pens):
                assert pset.result(i).status == 0, (case, pen, build)
                db_g = str(tmp_path / "g.db")
                pset.export_db(i, ce.astype(np.int32), db_g)
                assert open(db_g, "rb").read() == want[i], (case, pen, build)
            pset.close()
        monkeypatch.delenv("PEAKSEG_HIP_VARIANT")


@GPU
def test_sequential_envelope_replay(psd, oracle_det, tmp_path):
    """The sequential replay of min_env (min_env_serial) and the way the latency build's
    specialised step hands a data point over to the general step: no data set has needed them
    so far, so a build with the replay forced for every envelope (-DPSD_FORCE_SERIAL_ENV) is
    compared with the oracle on the device."""
    import subprocess
    import __graft_entry__ as entry
    from conftest import ROOT
    from peaksegdisk_amd import ProblemSet, _native, synthetic
    csrc = os.path.join(ROOT, "peaksegdisk_amd", "csrc")
    cs, ce, cnt = synthetic.poisson_coverage(5000, seed=31)
    pens = ["0.7", "60", "5000"]
    bg = str(tmp_path / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    # the same for the other hand-over that nothing realistic triggers: a step whose exp / log
    # met a rare argument (the specialised step evaluates them without the branch for those)
    # is redone by the general step; forced for every step with -DPSD_FORCE_RARE
    procs = {}
    for variant, flag in (("serial", "-DPSD_FORCE_SERIAL_ENV"), ("rare", "-DPSD_FORCE_RARE")):
        lib_path = str(tmp_path / ("libpeaksegdisk_hip_%s.so" % variant))
        procs[variant] = (lib_path, subprocess.Popen(
            [entry.HIPCC] + entry.HIP_FLAGS + [ flag,
             "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
             os.path.join(csrc, "peakseg_hip.cpp"), "-o", lib_path]))
    for variant, (lib_path, proc) in procs.items():
        assert proc.wait() == 0
        lib = _native.declare(ctypes.CDLL(lib_path))
        pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens],
                          lib=lib)
        pset.solve()
        for i, pen in enumerate(pens):
            r = pset.result(i)
            assert r.status == 0 and (r.n_serial_env > 0 or variant == "rare")
            want = str(tmp_path / ("o_%d.db" % i))
            if not os.path.exists(want):
                assert oracle_det.solve(bg, pen, want) == 0
            got = str(tmp_path / ("g_%s_%d.db" % (variant, i)))
            pset.export_db(i, ce, got)
            assert open(got, "rb").read() == open(want, "rb").read(), (variant, pen)
        pset.close()


def check_grid_properties(pset, pens, cs, ce, cnt, n_bins):
    """Size-independent properties of a solved (one contig x penalties) grid -- what the
    full-size tests check where the oracle cannot follow:
      * every problem solves; segment tables are well formed (odd row count, strictly
        decreasing starts, last row = first_chromStart);
      * up/down constraint: means rise into every peak and fall out of it;
      * the reported total loss equals the Poisson loss recomputed from the segmentation
        (sum over segments of w*m - z*log m) to 1e-6 relative -- a checksum of the whole
        forward pass and backtrack;
      * the number of peaks does not increase with the penalty.
    Returns per problem (seg_start, seg_mean, best_cost) for a determinism check."""
    w = (ce - cs).astype(np.float64)
    cw = np.concatenate([[0.0], np.cumsum(w)])
    cz = np.concatenate([[0.0], np.cumsum(w * cnt)])
    peaks, first = [], []
    for i, pen in enumerate(pens):
        r = pset.result(i)
        assert r.status == 0, (pen, r.kernel_status)
        start, mean = pset.segments(i)
        first.append((start.copy(), mean.copy(), r.best_cost))
        assert len(start) == r.n_segments and len(start) % 2 == 1
        assert start[-1] == -1 and (np.diff(start[:-1]) < 0).all()
        # genomic order: segment g covers data (lo[g], hi[g]]
        lo = start[::-1] + 1
        hi = np.concatenate([lo[1:], [n_bins]])
        m = mean[::-1]
        up = m[1::2] >= m[0:-1:2]      # background -> peak
        down = m[1::2] >= m[2::2]      # peak -> background
        assert up.all() and down.all(), pen
        seg_w = cw[hi] - cw[lo]
        seg_z = cz[hi] - cz[lo]
        with np.errstate(divide="ignore", invalid="ignore"):
            loss = np.where(seg_z > 0, seg_w * m - seg_z * np.log(m), seg_w * m).sum()
        total = r.best_cost * cw[-1] - float(pen) * r.n_peaks
        assert total == pytest.approx(loss, rel=1e-6, abs=1e-6), pen
        peaks.append(r.n_peaks)
    assert all(a >= b for a, b in zip(peaks, peaks[1:])), peaks
    return first


# PSD_FUZZ_EXTRA="4000:11,4000:12": more (cases, seed) runs of the fuzz test, for soak runs on a
# GPU box (profiles/r02/fuzz_soak.log); the suite itself runs the first pair only
_FUZZ_RUNS = [(300, 5)] + [tuple(int(v) for v in item.split(":"))
                           for item in os.environ.get("PSD_FUZZ_EXTRA", "").split(",") if item]


def fuzz_cases(n_cases, seed):
    """(count, width, chromStart, chromEnd, penalty string) of the tiny problems below; also
    solved by both oracle builds in tests/test_oracle_census.py"""
    rng = np.random.default_rng(seed)
    for c in range(n_cases):
        n = int(rng.integers(2, 41))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            cnt = rng.integers(0, 3, n)
        elif kind == 1:
            cnt = rng.integers(0, 50, n)
        elif kind == 2:
            cnt = np.sort(rng.integers(0, 30, n))
        elif kind == 3:
            cnt = np.sort(rng.integers(0, 30, n))[::-1]
        else:
            cnt = np.repeat(rng.integers(0, 8, (n + 3) // 4), 4)[:n]
        if (cnt == cnt[0]).all():
            cnt[-1] += 1  # constant data takes the closed-form branch on the host, not the DP
        wid = rng.integers(1, 30, n)
        end = np.cumsum(wid)
        start = end - wid
        pen = "0" if rng.random() < 0.15 else "%.15g" % (10.0 ** rng.uniform(-2, 6))
        yield cnt, wid, start, end, pen


@GPU
@pytest.mark.parametrize("n_cases,seed", _FUZZ_RUNS)
def test_fuzz_tiny_problems(psd, oracle_det, tmp_path, n_cases, seed):
    """Many tiny random problems in one problem set (ragged lengths 1..40, zeros, repeated
    counts, increasing/decreasing runs, wide count ranges, penalties from 0 to 1e6): the rare
    branches of the piece algebra (degenerate linear pieces, equal-at-left/right, crossings at
    interval ends) against the oracle, problem by problem."""
    from peaksegdisk_amd import ProblemSet
    contigs, problems, texts, pens = [], [], [], []
    for c, (cnt, wid, start, end, pen) in enumerate(fuzz_cases(n_cases, seed)):
        contigs.append((cnt.astype(np.int32), wid.astype(np.int32)))
        problems.append((c, float(pen)))
        pens.append(pen)
        texts.append(("".join("chrF\t%d\t%d\t%d\n" % t for t in zip(start, end, cnt)), end))
    pset = ProblemSet(contigs, problems)
    pset.solve()
    for c in range(n_cases):
        d = tmp_path / ("f%d" % c)
        d.mkdir()
        bg = str(d / "coverage.bedGraph")
        open(bg, "w").write(texts[c][0])
        assert oracle_det.solve(bg, pens[c]) == 0
        r = pset.result(c)
        assert r.status == 0, (c, r.kernel_status)
        seg_start, seg_mean = pset.segments(c)
        segs = read_segments("%s_penalty=%s_segments.bed" % (bg, pens[c]))
        end = texts[c][1]
        assert [s[1] for s in segs] == [0 if k < 0 else int(end[k]) for k in seg_start], c
        assert [s[4] for s in segs] == ["%g" % v for v in seg_mean], c
        loss = read_loss("%s_penalty=%s_loss.tsv" % (bg, pens[c])).split("\t")
        assert loss[5] == "%.20g" % r.best_cost, c
        assert int(loss[7]) == r.n_equality_constraints
        assert float(loss[9]) == r.max_intervals
        assert float(loss[8]) == r.total_intervals / (2.0 * len(end))
    pset.close()
