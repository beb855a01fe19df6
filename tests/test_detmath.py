"""include/peakseg_detmath.h: accuracy of the deterministic exp/log (< 1 ulp vs mpmath),
IEEE special cases equal to libm's, and how often it differs from glibc at all."""
import math

import numpy as np
import pytest

mp = pytest.importorskip("mpmath")


def _max_ulp(fn_mp, xs, ys):
    worst = 0.0
    for x, y in zip(xs, ys):
        exact = fn_mp(mp.mpf(float(x)))
        u = math.ulp(float(exact))
        worst = max(worst, float(abs(mp.mpf(float(y)) - exact) / u))
    return worst


def test_exp_accuracy(oracle_det):
    mp.mp.prec = 200
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.uniform(-10, 10, 4000), rng.uniform(-708, 709, 2000),
                         rng.uniform(-1e-3, 1e-3, 1000), rng.uniform(-745, -708, 500)])
    ys = oracle_det.math("exp", xs)
    assert _max_ulp(mp.exp, xs, ys) < 1.0


def test_log_accuracy(oracle_det):
    mp.mp.prec = 200
    rng = np.random.default_rng(8)
    xs = np.concatenate([rng.uniform(0, 100, 4000), np.exp(rng.uniform(-700, 700, 2000)),
                         1 + rng.uniform(-1e-2, 1e-2, 2000), rng.uniform(0.5, 2, 2000),
                         np.array([5e-324, 1e-310, 2.2250738585072014e-308])])
    ys = oracle_det.math("log", xs)
    assert _max_ulp(mp.log, xs, ys) < 1.0


def test_special_values_match_libm(oracle_det, oracle_libm):
    sp = np.array([0.0, -0.0, 1.0, np.inf, -np.inf, np.nan, -1.0, 710.0, 709.78, -745.2,
                   -745.0, -800.0, 1e-320, 1e308, 2.0, 0.5])
    for fn in ("exp", "log"):
        a = oracle_det.math(fn, sp)
        b = oracle_libm.math(fn, sp)
        assert np.array_equal(np.isnan(a), np.isnan(b)), fn
        m = ~np.isnan(a)
        fin = m & np.isfinite(b)
        assert np.array_equal(a[m & ~fin], b[m & ~fin]), fn
        assert np.allclose(a[fin], b[fin], rtol=4e-16, atol=0), fn
    assert oracle_det.math("log", np.array([1.0]))[0] == 0.0
    assert oracle_det.math("exp", np.array([0.0]))[0] == 1.0


def test_mostly_equal_to_glibc(oracle_det, oracle_libm):
    rng = np.random.default_rng(9)
    x = rng.uniform(-20, 20, 20000)
    d = oracle_det.math("exp", x) != oracle_libm.math("exp", x)
    assert d.mean() < 0.15
    x = rng.uniform(1e-3, 50, 20000)
    d = oracle_det.math("log", x) != oracle_libm.math("log", x)
    assert d.mean() < 0.15
