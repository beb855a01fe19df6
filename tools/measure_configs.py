#!/usr/bin/env python3
"""Measure the other BASELINE.json configurations on one MI355X (they are parity-test cases,
not bench lines; these numbers go into DESIGN.md / profiles/):

  c4   24 contigs x 64 penalties (configs[3]) scaled to one GPU: contig lengths log-uniform in
       [1e4, 1e6] instead of [1e5, 1e7] so that the store fits one device; 1536 problems in one
       launch -- the regime where the chip is actually filled
  c5   adversarial increasing counts (configs[4], vignettes/Worst_case.Rmd) at penalty 100:
       lists far beyond LDS -> HBM spill path
  c3   sequentialSearch_dir (configs[2]) on one synthetic contig through the resident native
       driver, optionally checked against a trace of the same search driven by the CPU oracle
  (configs[3] itself is `python bench.py --mode grid`)

usage: python tools/measure_configs.py c4|c5 [size] | c3 bins target_peaks [seed [trace]]
       | c3b [contigs [longest_bins]]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def c4(scale_hi=1e6):
    from peaksegdisk_amd import synthetic
    from peaksegdisk_amd.parallel import solve_grid
    rng = np.random.default_rng(4)
    lens = np.exp(rng.uniform(np.log(scale_hi / 100), np.log(scale_hi), 24)).astype(int)
    contigs = []
    for k, n in enumerate(lens):
        cs, ce, cnt = synthetic.poisson_coverage(int(n), seed=100 + k)
        contigs.append((cnt, (ce - cs).astype(np.int32)))
    pens = [float(p) for p in synthetic.penalty_grid(64)]
    t0 = time.time()
    out = solve_grid(contigs, pens, None, 0)
    wall = time.time() - t0
    bins = int(lens.sum()) * 64
    peaks = sum(int((v["summary"][0] - 1) // 2) for v in out.values())
    return {"config": "c4-scaled", "contigs": 24, "penalties": 64, "problems": len(out),
            "contig_bins_min": int(lens.min()), "contig_bins_max": int(lens.max()),
            "bin_penalty_pairs": bins, "wall_s_incl_upload_and_download": wall,
            "bins_per_s": bins / wall, "total_peaks": peaks}


def c5(n=100000):
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.increasing_coverage(n)
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, 100.0)])
    f_ms, b_ms = pset.solve()
    r = pset.result(0)
    out = {"config": "c5", "bins": n, "penalty": 100, "forward_ms": f_ms, "backtrack_ms": b_ms,
           "status": r.status, "segments": r.n_segments, "max_intervals": r.max_intervals,
           "mean_intervals": r.total_intervals / (2.0 * n), "spill_steps": r.spill_steps,
           "bins_per_s": n / ((f_ms + b_ms) / 1e3), "hbm_bytes": pset.hbm_bytes}
    pset.close()
    return out


def c3(n=2000000, peaks=300, seed=3, trace=None):
    """sequentialSearch_dir through the resident native driver.  trace: a log of the same search
    driven by the CPU oracle (tools: one dict per model, as profiles/r02/
    config3_cpu_trace_1e7_det_oracle.log -- which is the trace of `c3 10000000 3148 1`: SEED 1,
    the bench's contig; with another seed the search never meets 3148 peaks and runs on); the
    models visited must then equal its first ones, penalty strings included."""
    import ast
    import peaksegdisk_amd as psd
    from peaksegdisk_amd import synthetic
    cs, ce, cnt = synthetic.poisson_coverage(n, seed=seed)
    d = tempfile.mkdtemp(prefix="psd_c3_")
    pdir = os.path.join(d, "chrSynth-0-%d" % int(ce[-1]))
    os.makedirs(pdir)
    bg = os.path.join(pdir, "coverage.bedGraph")
    with open(bg, "w") as f:
        for o in range(0, n, 500000):
            f.write("".join("chrSynth\t%d\t%d\t%d\n" % t for t in zip(
                cs[o:o + 500000].tolist(), ce[o:o + 500000].tolist(),
                cnt[o:o + 500000].tolist())))
    os.environ["PEAKSEG_HIP_TIMING"] = "1"
    t0 = time.time()
    fit = psd.sequentialSearch_dir(pdir, peaks)
    wall = time.time() - t0
    others = fit.others
    out = {"config": "c3", "bins": n, "target_peaks": peaks,
           "found_peaks": int(fit.loss["peaks"].iloc[0]), "models": int(len(others)),
           "penalties": [psd.paste(float(x)) for x in others["penalty"]],
           "peaks_by_model": [int(x) for x in others["peaks"]],
           "seconds_by_model": [float(x) for x in others["seconds"]], "wall_s": wall,
           "dp_bins_per_s": n * (len(others) - 1) / wall}
    if trace:
        want = [ast.literal_eval(ln) for ln in open(trace) if ln.startswith("{")]
        k = len(out["penalties"])
        out["matches_cpu_oracle_trace"] = (
            out["penalties"] == [m["penalty"] for m in want[:k]] and
            out["peaks_by_model"] == [m["peaks"] for m in want[:k]])
        out["cpu_oracle_seconds_by_model"] = [round(m["seconds"], 1) for m in want[:k]]
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    return out


def c3b(n_contigs=24, scale_hi=1e6):
    """sequentialSearch_dir on n_contigs problem directories together (the batched search):
    contig lengths log-uniform in [scale_hi/100, scale_hi], target = about half of each contig's
    true peaks.  Compared with the longest contig searched alone (a copy of its directory)."""
    import shutil
    import peaksegdisk_amd as psd
    from peaksegdisk_amd import synthetic
    rng = np.random.default_rng(4)
    lens = np.exp(rng.uniform(np.log(scale_hi / 100), np.log(scale_hi), n_contigs)).astype(int)
    root = tempfile.mkdtemp(prefix="psd_c3b_")
    dirs, targets = [], []
    for k, n in enumerate(lens):
        cs, ce, cnt = synthetic.poisson_coverage(int(n), seed=200 + k)
        d = os.path.join(root, "chrSynth%d-0-%d" % (k, int(ce[-1])))
        os.makedirs(d)
        synthetic.write_bedgraph(os.path.join(d, "coverage.bedGraph"), cs, ce, cnt)
        dirs.append(d)
        targets.append(max(1, int(n) // 4400))
    longest = int(np.argmax(lens))
    alone = os.path.join(root, "alone")
    os.makedirs(alone)
    shutil.copy(os.path.join(dirs[longest], "coverage.bedGraph"), alone)
    t0 = time.time()
    fits = psd.sequentialSearch_dir_batch(dirs, targets)
    wall = time.time() - t0
    t0 = time.time()
    one = psd.sequentialSearch_dir(alone, targets[longest])
    wall_one = time.time() - t0
    models = [int(len(f.others)) for f in fits]
    dp_bins = int(sum(int(n) * (m - 1) for n, m in zip(lens, models)))
    same = [psd.paste(float(x)) for x in fits[longest].others["penalty"]] == \
        [psd.paste(float(x)) for x in one.others["penalty"]]
    out = {"config": "c3-batched", "contigs": n_contigs, "contig_bins_min": int(lens.min()),
           "contig_bins_max": int(lens.max()), "models_per_contig": models,
           "found_peaks": [int(f.loss["peaks"].iloc[0]) for f in fits], "target_peaks": targets,
           "dynamic_program_bins": dp_bins, "wall_s_all_contigs_together": wall,
           "dp_bins_per_s_together": dp_bins / wall,
           "wall_s_longest_contig_alone": wall_one, "models_longest_contig": models[longest],
           "longest_contig_same_penalties_alone_and_together": same}
    shutil.rmtree(root, ignore_errors=True)
    return out


if __name__ == "__main__":
    which = sys.argv[1]
    if which == "c3b":  # c3b [contigs [longest]]
        a = sys.argv[2:]
        res = c3b(int(a[0]) if a else 24, float(a[1]) if len(a) > 1 else 1e6)
        print(json.dumps(res))
        sys.exit(0)
    if which == "c3":  # c3 bins target_peaks seed [trace]
        a = sys.argv[2:]
        res = c3(int(float(a[0])), int(a[1]), int(a[2]) if len(a) > 2 else 3,
                 a[3] if len(a) > 3 else None)
    else:
        arg = [float(a) for a in sys.argv[2:]]
        fn = {"c4": c4, "c5": c5}[which]
        res = fn(*[int(a) if which != "c4" else a for a in arg])
    print(json.dumps(res))
