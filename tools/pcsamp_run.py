#!/usr/bin/env python3
"""Diagnostic workload for rocprofv3 PC sampling: the 64-penalty grid on a short contig, a few
solves.  usage: python tools/pcsamp_run.py [bins] [solves]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peaksegdisk_amd import ProblemSet, synthetic  # noqa: E402

bins = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
solves = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cs, ce, cnt = synthetic.poisson_coverage(bins, seed=1)
pens = synthetic.penalty_grid(64)
ps = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens])
for _ in range(solves):
    print("forward ms", ps.solve()[0], flush=True)
ps.close()
