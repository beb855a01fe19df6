#!/usr/bin/env python3
"""Wall time of the file-level batch entry (PeakSegFPOP_disk_batch: parse the bedGraph once,
upload, solve the 64-penalty grid in one launch, write 64 x (segments.bed, loss.tsv)) next to
the kernel time of the same grid -- what the boundary adds around the hot path.

usage: python tools/file_api_timing.py [bins]"""
import ctypes
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peaksegdisk_amd import ProblemSet, _native, synthetic  # noqa: E402

bins = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
cs, ce, cnt = synthetic.poisson_coverage(bins, seed=1)
pens = synthetic.penalty_grid(64)
work = tempfile.mkdtemp(prefix="psd_fileapi_")
try:
    bg = os.path.join(work, "coverage.bedGraph")
    t0 = time.time()
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    t_write_input = time.time() - t0
    n = len(pens)
    files = (ctypes.c_char_p * n)(*[os.fsencode(bg)] * n)
    pstr = (ctypes.c_char_p * n)(*[p.encode() for p in pens])
    dbs = (ctypes.c_char_p * n)(*[os.fsencode("%s_penalty=%s.db" % (bg, p)) for p in pens])
    status = (ctypes.c_int * n)()
    t0 = time.time()
    rc = _native.lib.PeakSegFPOP_disk_batch(n, files, pstr, dbs, status)
    t_batch = time.time() - t0
    out_bytes = sum(os.path.getsize(os.path.join(work, f)) for f in os.listdir(work)
                    if f.endswith("_segments.bed") or f.endswith("_loss.tsv"))
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens])
    pset.solve()
    k_ms, _ = pset.solve()
    pset.close()
    print(json.dumps({"bins": bins, "penalties": n, "rc": rc,
                      "batch_wall_s": t_batch, "kernel_s": k_ms / 1e3,
                      "boundary_overhead_s": t_batch - k_ms / 1e3,
                      "output_bytes": out_bytes, "bedGraph_bytes": os.path.getsize(bg),
                      "bins_per_s_through_files": bins * n / t_batch}))
finally:
    shutil.rmtree(work, ignore_errors=True)
