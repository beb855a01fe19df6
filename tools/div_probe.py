#!/usr/bin/env python3
"""The device's fp64 division against IEEE division (the host's), operand by operand: the
compiler's own sequence (peakseg_hip_math_probe op 3) and the library's psd_div (op 2,
include/peakseg_detmath.h), on random operands and on the structured ones that come within
~2^-50 of a unit of a midpoint between two doubles: divisors a few units below a power of two
under numerators a few units off small fractions.

usage (GPU box): python tools/div_probe.py > profiles/rNN/div_probe.log
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peaksegdisk_amd import _native  # noqa: E402


def device_div(op, a, b):
    x = np.ascontiguousarray(np.concatenate([a, b]))
    y = np.empty(a.size)
    assert _native.lib.peakseg_hip_math_probe(op, a.size, x.ctypes.data, y.ctypes.data) == 0
    return y


def pert(v, k):
    """v moved by k units in the last place"""
    return (np.ascontiguousarray(v).view(np.int64) + k).view(np.float64)


def cases(n=8000000, seed=5):
    rng = np.random.default_rng(seed)
    frac = pert(rng.integers(1, 200, n) / rng.integers(1, 200, n), rng.integers(-8, 9, n))
    wide = pert(rng.integers(1, 200, n) / rng.integers(1, 200, n), rng.integers(-60, 61, n))
    pow2 = 2.0 ** rng.integers(-3, 4, n)
    yield "the quotient that was found", np.array([float.fromhex("0x1.6666666666663p-1")]), \
        np.array([float.fromhex("0x1.ffffffffffffbp-1")])
    yield "random (0.5,2) / (0.5,2)", rng.uniform(0.5, 2, n), rng.uniform(0.5, 2, n)
    yield "fractions +-8 ulp / (1 - k ulp), k < 64", frac, pert(np.ones(n), -rng.integers(0, 64, n))
    yield "fractions +-8 ulp / (1 + k ulp), k < 64", frac, pert(np.ones(n), rng.integers(1, 64, n))
    yield "fractions +-60 ulp / (2^n - k ulp), k < 400", wide, pert(pow2, -rng.integers(1, 400, n))
    yield "fractions +-60 ulp / (2^n + k ulp), k < 400", wide, pert(pow2, rng.integers(1, 400, n))


def main():
    for name, a, b in cases():
        want = a / b
        line = "%-46s n %8d" % (name, a.size)
        for op, label in ((3, "compiler's sequence"), (2, "psd_div")):
            got = device_div(op, a, b)
            bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
            line += " | %s: %d off" % (label, bad.size)
            if op == 3:
                shown = bad[:3]
        print(line)
        for i in shown:
            print("      %s / %s: device %s, IEEE %s" % (float(a[i]).hex(), float(b[i]).hex(),
                                                      float(device_div(3, a[i:i + 1], b[i:i + 1])[0]).hex(),
                                                      float(want[i]).hex()))


if __name__ == "__main__":
    main()
