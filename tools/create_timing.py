#!/usr/bin/env python3
"""Diagnostic: where the creation of the bench's problem set (1e6 bins x 64 penalties) spends its
time (PEAKSEG_HIP_TIMING=1 prints the library's phases), three times in one process."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PEAKSEG_HIP_TIMING"] = "1"
from peaksegdisk_amd import ProblemSet, synthetic  # noqa: E402

cs, ce, cnt = synthetic.poisson_coverage(1000000, seed=1)
w = (ce - cs).astype(np.int32)
pens = [float(p) for p in synthetic.penalty_grid(64)]
for k in range(3):
    t0 = time.time()
    ps = ProblemSet([(cnt, w)], [(0, p) for p in pens])
    t1 = time.time()
    f_ms, _ = ps.solve()
    t2 = time.time()
    print("create %.3f s, solve %.3f s (kernel %.1f ms), resident %.2f GB, arena stats %r" % (
        t1 - t0, t2 - t1, f_ms, ps.hbm_bytes / 1e9, ps.arena_stats), flush=True)
    ps.close()
