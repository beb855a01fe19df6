#!/usr/bin/env python3
"""Diagnostic A/B: time the forward kernel of several builds of the HIP library on the same
problem set.  usage: python tools/ab_libs.py [--bins N] [--contigs C] lib1.so lib2.so ...
(C contigs of N bins x 64 penalties = 64*C problems in one launch; C=1 is the latency-bound
bench workload, C>=8 fills the chip)."""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peaksegdisk_amd import _native, synthetic  # noqa: E402
from peaksegdisk_amd.grid import ProblemSet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bins", type=int, default=100000)
ap.add_argument("--contigs", type=int, default=1)
ap.add_argument("--unequal", action="store_true",
                help="contig lengths log-uniform in [bins/100, bins] (the config-4 shape)")
ap.add_argument("--increasing", action="store_true",
                help="config 5: one contig of counts 1..bins at penalty 100 (lists in HBM)")
ap.add_argument("libs", nargs="+")
args = ap.parse_args()
contigs = []
lens = [args.bins] * args.contigs
if args.unequal:
    lens = np.exp(np.random.default_rng(4).uniform(np.log(args.bins / 100), np.log(args.bins),
                                                   args.contigs)).astype(int)
for k in range(args.contigs):
    cs, ce, cnt = synthetic.poisson_coverage(int(lens[k]), seed=1 + k)
    contigs.append((cnt, (ce - cs).astype(np.int32)))
pens = synthetic.penalty_grid(64)
problems = [(k, float(p)) for k in range(args.contigs) for p in pens]
if args.increasing:
    cs, ce, cnt = synthetic.increasing_coverage(args.bins)
    contigs = [(cnt, (ce - cs).astype(np.int32))]
    lens = [args.bins]
    pens = ["100"]
    problems = [(0, 100.0)]


def bind(path):
    """Older builds lack the newest entry points: bind what ProblemSet needs."""
    lib = ctypes.CDLL(os.path.abspath(path))
    try:
        return _native.declare(lib)
    except AttributeError:
        c = ctypes
        lib.peakseg_hip_problem_set_create.argtypes = [
            c.c_int, c.c_int, c.POINTER(c.c_int), c.POINTER(c.c_void_p), c.POINTER(c.c_void_p),
            c.c_int, c.POINTER(c.c_int), c.POINTER(c.c_double), c.c_ulonglong,
            c.POINTER(c.c_void_p)]
        lib.peakseg_hip_problem_set_solve.argtypes = [
            c.c_void_p, c.POINTER(c.c_float), c.POINTER(c.c_float)]
        lib.peakseg_hip_problem_set_destroy.argtypes = [c.c_void_p]
        lib.peakseg_hip_problem_set_destroy.restype = None
        lib.peakseg_hip_last_error.restype = c.c_char_p
        return lib


for name in args.libs:
    lib = bind(name)
    ps = ProblemSet(contigs, problems, lib=lib)
    ps.solve()
    f = [ps.solve()[0] for _ in range(1 if (args.unequal or args.increasing) else 2)]
    print(name, getattr(ps, "kernel_build", ""), "forward ms", min(f), "=> bins/s",
          int(np.sum(lens)) * len(pens) / (min(f) / 1e3), flush=True)
    ps.close()
