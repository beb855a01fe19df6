"""How far can the deterministic arithmetic drift from the reference's on data UNLIKE the
synthetic generator?  ("what's weak" 1.)

The HIP kernels are checked bit for bit against the oracle built on the deterministic exp/log
(`oracle_det`); the reference computes with glibc's.  At BASELINE sizes the two oracle builds
are compared on Poisson-shaped data by the `-m gpu` tests (profiles/r03/parity_vs_glibc_*.json).
This CPU test closes the rest of the gap: every data set of `test_varied_data_shapes` (18 shapes:
heavy tails, zero runs, counts in the millions, ramps, steps, 0/1 data) and every problem of
`test_fuzz_tiny_problems` (300 tiny problems) is solved by BOTH oracle builds.  Required: the
segment files are byte-identical (coordinates, states, 6-digit means) and the integer fields of
the loss row agree; `total.loss` and `mean.pen.cost` agree to 1e-6 relative (north_star's
tolerance).  Reported (tools/oracle_census.py prints the table; profiles/r04/ keeps a copy): how
many stored functions differ in piece count between the two arithmetics.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_det_vs_libm_census_on_shapes_and_fuzz(oracle_det, oracle_libm, tmp_path):
    import oracle_census
    report = oracle_census.run(oracle_det, oracle_libm, str(tmp_path))
    assert report["varied_shapes"]["problems"] == 18 * 3
    assert report["fuzz"]["problems"] == 300
    for name in ("varied_shapes", "fuzz"):
        r = report[name]
        assert r["segment_files_identical"] == r["problems"], r
        assert r["integer_loss_fields_identical"] == r["problems"], r
        assert r["max_rel_diff_total_loss"] <= 1e-6, r
    # the margin, as a number: a handful of stored functions at most change their piece count
    vs = report["varied_shapes"]
    assert vs["functions_with_different_piece_count"] <= vs["functions"] * 1e-3, vs
