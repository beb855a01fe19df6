#!/usr/bin/env python3
"""Census: the deterministic-arithmetic oracle against the glibc-arithmetic oracle (the
reference's arithmetic) on the data sets of test_varied_data_shapes and test_fuzz_tiny_problems.
No GPU.  tests/test_oracle_census.py asserts on run()'s result; run as a script it prints the
JSON report (a copy is kept under profiles/r04/).

usage: python tools/oracle_census.py [out.json]
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _piece_counts(db, n_bins):
    """piece count of every stored function of a DiskVector-format file"""
    with open(db, "rb") as f:
        raw = f.read()
    table = np.frombuffer(raw[:32 * n_bins], dtype=np.int64)[0::2]
    out = np.zeros(2 * n_bins, dtype=np.int64)
    for k, pos in enumerate(table.tolist()):
        if pos > 0:
            out[k] = int(np.frombuffer(raw[pos + 4:pos + 8], dtype=np.int32)[0])
    return out


def _compare(oracle_det, oracle_libm, work, name, text, pen, n_bins, acc):
    files = {}
    for kind, oracle in (("det", oracle_det), ("libm", oracle_libm)):
        d = os.path.join(work, "%s_%s" % (name, kind))
        os.makedirs(d)
        bg = os.path.join(d, "coverage.bedGraph")
        with open(bg, "w") as f:
            f.write(text)
        db = bg + ".db"
        st = oracle.solve(bg, pen, db)
        assert st == 0, (name, pen, kind, st)
        files[kind] = (open("%s_penalty=%s_segments.bed" % (bg, pen), "rb").read(),
                       open("%s_penalty=%s_loss.tsv" % (bg, pen)).read().rstrip("\n").split("\t"),
                       _piece_counts(db, n_bins) if os.path.exists(db) else None)
        shutil.rmtree(d)
    (seg_d, loss_d, pc_d), (seg_l, loss_l, pc_l) = files["det"], files["libm"]
    acc["problems"] += 1
    acc["segment_files_identical"] += int(seg_d == seg_l)
    if seg_d != seg_l:
        acc["differing"].append({"case": name, "penalty": pen})
    # segments, peaks, bases, bedGraph.lines, equality.constraints, max.intervals
    ints_same = all(loss_d[k] == loss_l[k] for k in (1, 2, 3, 4, 7)) and \
        float(loss_d[9]) == float(loss_l[9])
    acc["integer_loss_fields_identical"] += int(ints_same)
    for col in (5, 6):
        a, b = float(loss_d[col]), float(loss_l[col])
        rel = abs(a - b) / max(abs(a), abs(b), 1e-300) if a != b else 0.0
        acc["max_rel_diff_total_loss"] = max(acc["max_rel_diff_total_loss"], rel)
    acc["mean_intervals_differ"] += int(loss_d[8] != loss_l[8])
    if pc_d is not None and pc_l is not None:
        acc["functions"] += int((pc_d > 0).sum())
        acc["functions_with_different_piece_count"] += int((pc_d != pc_l).sum())


def _acc():
    return {"problems": 0, "segment_files_identical": 0, "integer_loss_fields_identical": 0,
            "max_rel_diff_total_loss": 0.0, "mean_intervals_differ": 0, "functions": 0,
            "functions_with_different_piece_count": 0, "differing": []}


def run(oracle_det, oracle_libm, work):
    import test_gpu_parity as gp
    report = {"varied_shapes": _acc(), "fuzz": _acc()}
    n_cases, seed = gp.VARIED_SHAPES_RUN
    for case, (cnt, w, cs, ce, pens) in enumerate(gp.varied_shape_cases(n_cases, seed)):
        text = "".join("chrSynth\t%d\t%d\t%d\n" % t for t in zip(cs.tolist(), ce.tolist(),
                                                                   cnt.tolist()))
        for i, pen in enumerate(pens):
            _compare(oracle_det, oracle_libm, work, "shape%d_%d" % (case, i), text, pen, len(cnt),
                     report["varied_shapes"])
    n_cases, seed = gp._FUZZ_RUNS[0]
    for c, (cnt, wid, start, end, pen) in enumerate(gp.fuzz_cases(n_cases, seed)):
        text = "".join("chrF\t%d\t%d\t%d\n" % t for t in zip(start.tolist(), end.tolist(),
                                                               cnt.tolist()))
        _compare(oracle_det, oracle_libm, work, "fuzz%d" % c, text, pen, len(cnt), report["fuzz"])
    report["what"] = ("oracle_det (deterministic exp/log, what the HIP kernels compute with) vs "
                      "oracle_libm (glibc exp/log, the reference's arithmetic): %d data shapes x 3 "
                      "penalties of test_varied_data_shapes, %d problems of test_fuzz_tiny_problems"
                      % (gp.VARIED_SHAPES_RUN[0], gp._FUZZ_RUNS[0][0]))
    return report


if __name__ == "__main__":
    from conftest import Oracle, _build_oracle
    _build_oracle()
    work = tempfile.mkdtemp(prefix="psd_census_")
    try:
        rep = run(Oracle("det"), Oracle("libm"), work)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    text = json.dumps(rep, indent=1)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(text + "\n")
