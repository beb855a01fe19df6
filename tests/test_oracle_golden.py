"""Pin the CPU oracle (oracle/) against the reference's own known answers.

liboracle_libm.so (glibc exp/log, as the reference) must reproduce every digit recorded in
tests/golden/known_answers.json; liboracle_det.so (deterministic exp/log shared with the
HIP kernels) must agree on every integer field and segment row, and to 1e-6 relative on
the floating-point loss fields (BASELINE.json north_star tolerance).
"""
import hashlib
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN, read_loss, read_segments

REL_TOL = 1e-6  # north_star: total Poisson loss within 1e-6 relative


def _write(tmp_path, text, name="coverage.bedGraph"):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def _outs(bg, pen):
    pre = "%s_penalty=%s" % (bg, pen)
    return pre + "_segments.bed", pre + "_loss.tsv"


def _check_case(oracle, case, tmp_path, exact_floats):
    bg = _write(tmp_path, case["bedGraph"])
    status = oracle.solve(bg, case["penalty"])
    assert status == case["status"]
    seg_path, loss_path = _outs(bg, case["penalty"])
    segs = read_segments(seg_path)
    loss = read_loss(loss_path).split("\t")
    if "segments" in case:
        assert segs == [[c, s, e, st, m] for c, s, e, st, m in case["segments"]]
    if "n_segments" in case:
        assert len(segs) == case["n_segments"]
        assert int(loss[1]) == case["n_segments"]
    if "peaks" in case:
        assert int(loss[2]) == case["peaks"]
        assert sum(r[3] == "peak" for r in segs) == case["peaks"]
    if "loss_row" in case:
        want = case["loss_row"].split("\t")
        if exact_floats:
            assert loss == want
        else:
            for i, (a, b) in enumerate(zip(loss, want)):
                if i in (5, 6, 8):
                    assert float(a) == pytest.approx(float(b), rel=REL_TOL)
                else:
                    assert a == b
    if "db_bytes" in case:
        assert os.path.getsize(bg + "_penalty=%s.db" % case["penalty"]) == case["db_bytes"]


def test_solve_cases_libm_exact(oracle_libm, known_answers, tmp_path):
    for i, case in enumerate(known_answers["solve_cases"]):
        d = tmp_path / ("c%d" % i)
        d.mkdir()
        _check_case(oracle_libm, case, d, exact_floats=True)


def test_solve_cases_detmath(oracle_det, known_answers, tmp_path):
    for i, case in enumerate(known_answers["solve_cases"]):
        d = tmp_path / ("c%d" % i)
        d.mkdir()
        _check_case(oracle_det, case, d, exact_floats=False)


@pytest.mark.parametrize("kind", ["libm", "det"])
def test_error_cases(kind, oracle_libm, oracle_det, known_answers, tmp_path):
    oracle = oracle_libm if kind == "libm" else oracle_det
    for i, case in enumerate(known_answers["error_cases"]):
        d = tmp_path / ("e%d" % i)
        d.mkdir()
        if case["bedGraph"] is None:
            bg = str(d / "does_not_exist")
        else:
            bg = _write(d, case["bedGraph"])
        pre = "%s_penalty=%s" % (bg, case["penalty"])
        db = pre + ".db"
        block = case.get("block")
        if block == "segments":
            os.mkdir(pre + "_segments.bed")
        elif block == "loss":
            os.mkdir(pre + "_loss.tsv")
        elif block == "db":
            os.mkdir(db)
        assert oracle.solve(bg, case["penalty"], db) == case["status"], case["name"]


def test_validation_order(oracle_libm, tmp_path):
    """penalty is checked before the input is opened; the input is fully validated before
    any output file is created (ref: PeakSegFPOPLog.cpp:145-163,173-209,222-223)."""
    missing = str(tmp_path / "nope")
    assert oracle_libm.solve(missing, "foobar") == 10
    assert oracle_libm.solve(missing, "-1") == 2
    bg = _write(tmp_path, "chr1 0 1 5\nchr1 2 3 3")
    assert oracle_libm.solve(bg, "1") == 6
    assert not os.path.exists(bg + "_penalty=1_loss.tsv")
    assert not os.path.exists(bg + "_penalty=1_segments.bed")


def test_mono27ac_fixture_hash(known_answers):
    with open(os.path.join(GOLDEN, "Mono27ac.bedGraph"), "rb") as f:
        data = f.read()
    assert hashlib.sha256(data).hexdigest() == known_answers["mono27ac"]["sha256"]
    assert data.count(b"\n") == known_answers["mono27ac"]["rows"]


@pytest.mark.parametrize("pen", ["1952.6", "0", "10000", "Inf"])
def test_mono27ac(pen, oracle_libm, oracle_det, known_answers, tmp_path):
    want = known_answers["mono27ac"]["penalties"][pen]
    results = {}
    for oracle in (oracle_libm, oracle_det):
        d = tmp_path / oracle.kind
        d.mkdir()
        bg = str(d / "coverage.bedGraph")
        shutil.copy(os.path.join(GOLDEN, "Mono27ac.bedGraph"), bg)
        assert oracle.solve(bg, pen) == 0
        seg_path, loss_path = _outs(bg, pen)
        segs = open(seg_path).read()
        loss = read_loss(loss_path).split("\t")
        results[oracle.kind] = (segs, loss)
        assert int(loss[1]) == want["segments"]
        assert int(loss[2]) == want["peaks"]
        assert int(loss[3]) == 520000 and int(loss[4]) == 6921
        if "equality_constraints" in want:
            assert int(loss[7]) == want["equality_constraints"]
        if "max_intervals" in want:
            assert float(loss[9]) == want["max_intervals"]
        if "first_segment_row" in want:
            assert segs.split("\n")[0] == want["first_segment_row"]
        if oracle.kind == "libm":  # every recorded digit
            assert loss[6] == want["total_loss"]
            if "mean_pen_cost" in want:
                assert loss[5] == want["mean_pen_cost"]
            if "mean_intervals" in want:
                assert loss[8] == want["mean_intervals"]
        else:
            assert float(loss[6]) == pytest.approx(float(want["total_loss"]), rel=REL_TOL)
    # endpoints, states and 6-digit means identical between the two math builds
    assert results["libm"][0] == results["det"][0]


def test_division_near_tie_fixture(oracle_libm, oracle_det, tmp_path):
    """tests/golden/division_near_tie.json: 64 bins whose data point 47 divides
    0x1.6666666666663p-1 by 0x1.ffffffffffffbp-1 (the quotient 6e-17 of a unit from a midpoint,
    which the MI355X's division sequence gets one unit off, include/peakseg_detmath.h): the
    stores both oracle builds write are pinned by their hashes, and the host's division gives
    the IEEE quotient."""
    import json
    from peaksegdisk_amd import synthetic
    g = json.load(open(os.path.join(GOLDEN, "division_near_tie.json")))
    assert (float.fromhex("0x1.6666666666663p-1") / float.fromhex("0x1.ffffffffffffbp-1")).hex() \
        == "0x1.6666666666667p-1"
    bg = str(tmp_path / "c.bedGraph")
    synthetic.write_bedgraph(bg, np.array(g["chromStart"]), np.array(g["chromEnd"]), np.array(g["count"]))
    for name, oracle in (("det", oracle_det), ("libm", oracle_libm)):
        for pen in g["penalties"]:
            db = str(tmp_path / "o.db")
            assert oracle.solve(bg, pen, db) == 0
            assert hashlib.sha256(open(db, "rb").read()).hexdigest() == g["db_sha256"]["%s:%s" % (name, pen)]


def test_division_repair_restores_the_ieee_quotient(oracle_det):
    """psd_div_repair (include/peakseg_detmath.h), what the device applies to the quotient of its
    own division sequence, run on the host: handed the IEEE quotient or either of its
    neighbours -- for random operands of both signs and many magnitudes, and for the operands
    that come closest to midpoints -- it returns the IEEE quotient; zero, infinite and NaN
    quotients pass through."""
    import ctypes
    rng = np.random.default_rng(11)
    n = 400000
    a = rng.uniform(0.5, 2, n) * 2.0 ** rng.integers(-300, 300, n) * rng.choice([-1.0, 1.0], n)
    b = rng.uniform(0.5, 2, n) * 2.0 ** rng.integers(-300, 300, n) * rng.choice([-1.0, 1.0], n)

    def pert(v, k):
        return (np.ascontiguousarray(v).view(np.int64) + k).view(np.float64)
    frac = pert(rng.integers(1, 200, n) / rng.integers(1, 200, n), rng.integers(-8, 9, n))
    near = pert(np.ones(n), -rng.integers(0, 64, n))
    a = np.concatenate([a, frac, [float.fromhex("0x1.6666666666663p-1")]])
    b = np.concatenate([b, near, [float.fromhex("0x1.ffffffffffffbp-1")]])
    want = a / b

    def repair(q):
        x = np.ascontiguousarray(np.concatenate([q, a, b]))
        y = np.empty(a.size)
        oracle_det.lib.oracle_div_repair_vec(ctypes.c_int(a.size), x.ctypes.data_as(ctypes.c_void_p),
                                             y.ctypes.data_as(ctypes.c_void_p))
        return y
    for q in (want, np.nextafter(want, np.inf), np.nextafter(want, -np.inf)):
        assert np.array_equal(repair(q).view(np.uint64), want.view(np.uint64))
    # special quotients stay as they are
    a = np.array([0.0, -0.0, 1.0, -1.0, np.nan, np.inf])
    b = np.array([3.0, 3.0, 0.0, 0.0, 2.0, 2.0])
    with np.errstate(divide="ignore", invalid="ignore"):
        want = a / b
    got = repair(want)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))

