#!/usr/bin/env python3
"""Soak run (GPU box): random mixed problem sets under random combinations of the knobs that make
resources scarce -- arena (blocks, estimate, growth under the kernel on/off, plain allocations),
spill slots, the pool for parked long functions, the kernel build, parking on/off, the
checkpointed store -- every problem against the deterministic oracle (whole store, or segment
table + summary under the checkpointed store).  tests/test_gpu_round4.py::
test_knob_combinations_on_a_mixed_set is the fixed version of this in the suite.

usage: python tools/knob_soak.py [iterations] [seed] [first]     (exit code 1 on the first mismatch;
PSD_SOAK_EMU=1: on the SIMT emulator of tests/emu instead of the GPU)
"""
import os
import sys
import tempfile
import shutil

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

POOLS = {
    "PEAKSEG_HIP_VARIANT": [None, None, "lat", "thr", "pk"],
    "PEAKSEG_HIP_SPILL_SLOTS": [None, "1", "2", "3"],
    "PEAKSEG_HIP_PIECES_PER_FUNCTION": [None, "1", "2"],
    "PEAKSEG_HIP_NO_LIVE_GROWTH": [None, "1"],
    "PEAKSEG_HIP_ARENA_BLOCK_LOG2": ["11", "12", "13", "15"],
    "PEAKSEG_HIP_CKPT_OVERFLOW": [None, "128", "512"],
    "PEAKSEG_HIP_NO_VMM": [None, None, "1"],
    "PEAKSEG_HIP_NO_PARK": [None, None, None, "1"],
    "PEAKSEG_HIP_CHECKPOINT": [None, None, None, None, None, None, "23", "64", "300"],
    "PEAKSEG_HIP_SPILL_CAP": [None, None, "2048"],
}


def main():
    iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # (replay: skip the iterations before)
    from conftest import Oracle
    from peaksegdisk_amd import ProblemSet, synthetic
    if os.environ.get("PSD_SOAK_EMU"):  # rehearsal on the SIMT emulator (tests/emu), no GPU
        import ctypes
        from peaksegdisk_amd import _native
        _native.lib = _native.declare(ctypes.CDLL(os.path.join(
            ROOT, "tests", "emu", "_build", "libpeaksegdisk_emu.so")))
    from test_gpu_parity import varied_shape_cases
    from test_gpu_round3 import check_tables_vs_oracle_files
    oracle = Oracle("det")
    rng = np.random.default_rng(seed)
    launches_seen, builds_seen = {}, {}
    for it in range(iterations):
        work = tempfile.mkdtemp(prefix="psd_soak_")
        data = []
        for _ in range(int(rng.integers(1, 3))):
            cs, ce, cnt = synthetic.poisson_coverage(int(rng.integers(300, 5000)), seed=int(rng.integers(1 << 30)))
            data.append((cnt, ce - cs, cs, ce, ["%.15g" % float(10 ** rng.uniform(-1, 6)) for _ in range(3)]))
        for _ in range(int(rng.integers(0, 3))):
            cs, ce, cnt = synthetic.increasing_coverage(int(rng.integers(200, 1600)))
            data.append((cnt, ce - cs, cs, ce, ["%.15g" % float(10 ** rng.uniform(0.5, 3.5)) for _ in range(2)]))
        for cnt, w, cs, ce, pens in varied_shape_cases(int(rng.integers(1, 4)), int(rng.integers(1 << 30))):
            data.append((cnt, w, cs, ce, pens))
        contigs = [(np.asarray(c).astype(np.int32), np.asarray(w).astype(np.int32)) for c, w, _, _, _ in data]
        problems, want, files = [], [], []
        for k, (cnt, w, cs, ce, pens) in enumerate(data if it >= first else []):
            bg = os.path.join(work, "k%d.bedGraph" % k)
            synthetic.write_bedgraph(bg, cs, ce, cnt)
            for pen in pens:
                db_o = os.path.join(work, "o.db")
                assert oracle.solve(bg, pen, db_o) == 0
                want.append(open(db_o, "rb").read())
                problems.append((k, float(pen)))
                files.append((bg, pen))
        knobs = {}
        for name, pool in POOLS.items():
            v = pool[int(rng.integers(len(pool)))]
            if v is not None:
                knobs[name] = v
        if "PEAKSEG_HIP_CHECKPOINT" in knobs and knobs.get("PEAKSEG_HIP_VARIANT") == "pk":
            del knobs["PEAKSEG_HIP_VARIANT"]  # (the packed build needs park slots: not with this store)
        if it < first:
            shutil.rmtree(work, ignore_errors=True)
            continue
        for name, value in knobs.items():
            os.environ[name] = value
        try:
            pset = ProblemSet(contigs, problems)
            pset.solve()
            for i, (k, _) in enumerate(problems):
                r = pset.result(i)
                if r.status != 0:
                    raise AssertionError("status %d / %d of problem %d" % (r.status, r.kernel_status, i))
                if pset.checkpoint_interval > 0:
                    start, mean = pset.segments(i)
                    check_tables_vs_oracle_files(
                        files[i][0], files[i][1], np.asarray(data[k][2]), np.asarray(data[k][3]), start,
                        mean, [r.n_segments, r.n_equality_constraints, r.max_intervals,
                               r.total_intervals, r.best_cost])
                    continue
                db_g = os.path.join(work, "g.db")
                pset.export_db(i, np.asarray(data[k][3]).astype(np.int32), db_g)
                if open(db_g, "rb").read() != want[i]:
                    raise AssertionError("store of problem %d differs" % i)
            key = pset.kernel_build + (" ckpt" if pset.checkpoint_interval > 0 else "")
            builds_seen[key] = builds_seen.get(key, 0) + 1
            launches_seen[pset.solve_stats[0]] = launches_seen.get(pset.solve_stats[0], 0) + 1
            pset.close()
        except Exception as e:  # noqa: BLE001
            print("MISMATCH in iteration %d (seed %d): %s\n  knobs %s\n  set: %s" % (
                it, seed, e, knobs, [(len(d[0]), d[4]) for d in data]), flush=True)
            sys.exit(1)
        finally:
            for name in knobs:
                del os.environ[name]
            shutil.rmtree(work, ignore_errors=True)
        if it % 10 == 9:
            print("%d iterations, %s, launches per solve %s" % (it + 1, builds_seen, dict(sorted(launches_seen.items()))),
                  flush=True)
    print("soak ok: %d iterations (seed %d), builds %s, launches per solve %s" % (
        iterations, seed, builds_seen, dict(sorted(launches_seen.items()))))


if __name__ == "__main__":
    main()
