#!/usr/bin/env python3
"""Diagnostic: build the HIP library with -DPSD_PROFILE (in-kernel cycle stamps around the
phases of the forward kernel) into gpurun_out/, run a small grid and print, per wave, the share
of cycles spent in each phase.  Shares only -- a stamped build is slower than the real one.

usage (on the GPU box): python tools/phase_profile.py [bins] [penalties]
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (slot 8 is PROF_SERIAL: in the walks it times the all-pairs SPECULATION ROUND -- the Newton
# solves of every (start, later piece) pair -- and in the envelope the sequential replay, which
# has run 0 times on every data set so far; rounds 1-3 labelled it "env serial")
NAMES = ["pre(costs+classes)", "walk(state machine+roots)", "env table", "env classify",
         "env compact", "scale_add", "arena", "barrier wait", "speculation round", "TOTAL",
         "c.load", "c.mid", "c.opt", "c.small", "c.large", "c.tail",
         "it.spec", "it.small", "it.large", "-", "s.assign", "s.load", "s.newton", "-"]
NP = 24


def main():
    bins = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    npen = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    lib_path = os.path.join(out, "libpeaksegdisk_hip_prof.so")
    csrc = os.path.join(ROOT, "peaksegdisk_amd", "csrc")
    import __graft_entry__ as entry
    subprocess.run([entry.HIPCC] + entry.HIP_FLAGS + ["-DPSD_PROFILE"]
                   + os.environ.get("PSD_PROFILE_FLAGS", "").split() + [
                    "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                    os.path.join(csrc, "peakseg_hip.cpp"), "-o", lib_path], check=True)
    from peaksegdisk_amd import _native, synthetic
    from peaksegdisk_amd.grid import ProblemSet
    lib = _native.declare(ctypes.CDLL(lib_path))
    lib.peakseg_hip_problem_set_profile.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    if os.environ.get("PSD_PROFILE_DATA") == "increasing": # config 5: long functions
        cs, ce, cnt = synthetic.increasing_coverage(bins)
        pens = [100.0] * npen
    else:
        cs, ce, cnt = synthetic.poisson_coverage(bins, seed=1)
        pens = synthetic.penalty_grid(npen)
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, float(p)) for p in pens], lib=lib)
    f_ms, b_ms = pset.solve()
    print("bins=%d penalties=%d forward=%.1f ms backtrack=%.1f ms (stamped build)" % (
        bins, npen, f_ms, b_ms))
    for p in range(npen):
        buf = np.zeros(2 * NP, dtype=np.int64)
        if lib.peakseg_hip_problem_set_profile(pset._h, p, buf.ctypes.data) < 0:
            raise SystemExit("profile not available")
        r = pset.result(p)
        for w in range(2):
            v = buf[w * NP:(w + 1) * NP]
            tot = float(v[9])
            shares = " ".join("%s=%.1f%%" % (NAMES[i].split("(")[0], 100.0 * v[i] / tot)
                              for i in list(range(9)) + list(range(10, 16)) + [20, 21, 22])
            shares += " | wave-level Newton trips per step: spec=%.1f small=%.1f large=%.1f walk rounds=%.1f" % (
                v[16] / bins, v[17] / bins, v[18] / bins, v[19] / bins)
            print("pen=%-18s wave%d cyc/step=%7.0f mean_int=%.2f | %s" % (
                pens[p], w, tot / bins, r.total_intervals / (2.0 * bins), shares))
    pset.close()


if __name__ == "__main__":
    main()
