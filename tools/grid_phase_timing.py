#!/usr/bin/env python3
"""Diagnostic: where the wall time of `bench.py --mode grid` goes outside the kernel
(problem-set creation = upload + arena, solve, download of the segment tables, close)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from peaksegdisk_amd import synthetic  # noqa: E402
from peaksegdisk_amd.grid import ProblemSet  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
lens = bench.grid_contig_lengths(24, scale)
contigs = []
for k, n in enumerate(lens):
    cs, ce, cnt = synthetic.poisson_coverage(int(n), seed=100 + k)
    contigs.append((cnt, (ce - cs).astype(np.int32)))
pens = [float(p) for p in synthetic.penalty_grid(64)]
problems = [(c, p) for c in range(len(contigs)) for p in pens]
for rep in range(2):
    t0 = time.time()
    pset = ProblemSet(contigs, problems)
    t1 = time.time()
    f_ms, _ = pset.solve()
    t2 = time.time()
    rows = 0
    for k in range(len(problems)):
        pset.result(k)
        start, mean = pset.segments(k)
        rows += len(start)
    t3 = time.time()
    hbm, build = pset.hbm_bytes, pset.kernel_build
    pset.close()
    t4 = time.time()
    print("rep %d: create %.3f s, solve %.3f s (kernel %.3f s), results+segments %.3f s (%d rows), "
          "close %.3f s; resident %.1f GB, build %s" % (rep, t1 - t0, t2 - t1, f_ms / 1e3, t3 - t2,
                                                       rows, t4 - t3, hbm / 1e9, build),
          flush=True)
