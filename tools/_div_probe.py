import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peaksegdisk_amd import _native
lib = _native.declare(ctypes.CDLL(os.environ.get("REPRO_LIB", "tools/_build/lib_d_div.so")))
def dev_div(a, b):
    x = np.ascontiguousarray(np.concatenate([a, b])); y = np.empty(a.size)
    assert lib.peakseg_hip_math_probe(2, a.size, x.ctypes.data, y.ctypes.data) == 0
    return y
def sweep(name, a, b):
    y = dev_div(a, b); w = a / b
    d = np.nonzero(y.view(np.uint64) != w.view(np.uint64))[0]
    d = d[~(np.isnan(y[d]) & np.isnan(w[d]))]
    print(name, "n", a.size, "mismatches", d.size)
    for i in d[:6]:
        print("   ", float(a[i]).hex(), "/", float(b[i]).hex(), "device", float(y[i]).hex(), "host", float(w[i]).hex())
import struct
def dd(h): return struct.unpack('<d', struct.pack('<Q', int(h, 16)))[0]
sweep("the pair", np.array([-dd('bfe6666666666663')]), np.array([dd('3feffffffffffffb')]))
rng = np.random.default_rng(5)
sweep("random (0.5,2)/(0.5,2)", rng.uniform(0.5, 2, 8000000), rng.uniform(0.5, 2, 8000000))
# small rationals perturbed by a few ulps: p/q * (1 +- k ulp) over r/s * (1 +- m ulp)
def pert(v, k): return (v.view(np.int64) + k).view(np.float64)
p = rng.integers(1, 200, 8000000).astype(float); q = rng.integers(1, 200, 8000000).astype(float)
a = pert(p / q, rng.integers(-8, 9, 8000000)); b = pert(np.ones(8000000) * rng.integers(1, 50, 8000000) / rng.integers(1, 50, 8000000), rng.integers(-8, 9, 8000000))
sweep("perturbed rationals", a, b)
b1 = pert(np.ones(8000000), -rng.integers(0, 64, 8000000))
sweep("x / (1 - k ulp)", a, b1)

b2 = pert(np.ones(8000000), rng.integers(1, 64, 8000000))
sweep("x / (1 + k ulp)", a, b2)
b3 = pert(np.ones(8000000) * 2.0 ** rng.integers(-3, 4, 8000000), -rng.integers(1, 400, 8000000))
sweep("x / (2^n - k ulp), k < 400", pert(rng.integers(1, 200, 8000000) / rng.integers(1, 200, 8000000), rng.integers(-60, 61, 8000000)), b3)
b4 = pert(np.ones(8000000) * 2.0 ** rng.integers(-3, 4, 8000000), rng.integers(1, 400, 8000000))
sweep("x / (2^n + k ulp), k < 400", pert(rng.integers(1, 200, 8000000) / rng.integers(1, 200, 8000000), rng.integers(-60, 61, 8000000)), b4)
