#!/usr/bin/env python3
"""bench.py -- coverage bins/sec of the PeakSegFPOP hot path on a 64-penalty grid.

A "step" is one pass of the hot path (forward functional-pruning DP + backtrack) over one
batch: a seeded synthetic Poisson-coverage contig of --bins data points x --penalties
log-spaced penalties, already resident in HBM when the timed region starts.  At N GPUs every
rank solves its own contig x penalty grid (independent problems, weak scaling) and the
segment tables are gathered to rank 0 over RCCL inside the timed region.

Prints ONE JSON line (rank 0): metric/value/unit per BASELINE.json, plus
  roofline     -- the kernel (forward pass with the decoding fused in): algorithmic HBM bytes
                  (24 B per (bin,penalty) + 20 B per stored piece; decoding: per
                  segment 12 B written, 8 B + 28 B per piece of the function read) / HIP-event
                  kernel time, against 8 TB/s
  cpu_baseline -- the CPU oracle (libm build, disk-backed store like the reference) timed on
                  a bounded sample of the same workload on this host.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(chrom_start, chrom_end, count, penalties, budget_bins):
    """Time oracle_cli_libm (one process per (contig, penalty), db on local scratch) on a
    bounded sample: the first `budget_bins` data points at a spread of the grid's penalties."""
    cli = os.path.join(ROOT, "oracle", "_build", "oracle_cli_libm")
    if not os.path.exists(cli):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    from peaksegdisk_amd import synthetic
    n = min(budget_bins, len(count))
    pick = sorted(set(np.linspace(0, len(penalties) - 1, 8).round().astype(int).tolist()))
    work = tempfile.mkdtemp(prefix="psd_cpu_")
    try:
        bg = os.path.join(work, "coverage.bedGraph")
        synthetic.write_bedgraph(bg, chrom_start[:n], chrom_end[:n], count[:n])
        t0 = time.time()
        for i in pick:
            st = subprocess.run([cli, bg, penalties[i], os.path.join(work, "db")]).returncode
            if st != 0:
                raise RuntimeError("oracle_cli_libm status %d" % st)
        wall = time.time() - t0
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return {
        "value": n * len(pick) / wall, "unit": "bins/s", "cores": 1, "kind": "port",
        "sample": "first %d bins of the contig x %d of the %d penalties, oracle_cli_libm "
                  "(C restatement, glibc exp/log, disk-backed store), one process per problem, "
                  "%.1f s" % (n, len(pick), len(penalties), wall),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bins", type=int, default=1000000)
    ap.add_argument("--penalties", type=int, default=64)
    ap.add_argument("--cpu-bins", type=int, default=200000)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # RCCL ("nccl") on GPUs; tests/test_bench_multirank.py rehearses the N>1 plumbing on CPU
    # with PSD_BENCH_BACKEND=gloo
    backend = os.environ.get("PSD_BENCH_BACKEND", "nccl")
    on_gpu = backend == "nccl"
    if world > 1:
        import torch.distributed as dist
        if on_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    device = local_rank if (world > 1 and on_gpu) else 0

    import __graft_entry__ as entry
    if rank == 0 or not os.path.exists(entry.LIB):
        entry.build_hip()
    if dist is not None:
        dist.barrier()
    from peaksegdisk_amd import ProblemSet, synthetic
    from peaksegdisk_amd.parallel import gather_segment_tables

    # every rank: its own contig (seed 1 + rank), the same penalty grid
    cs, ce, cnt = synthetic.poisson_coverage(args.bins, seed=1 + rank)
    weight = (ce - cs).astype(np.int32)
    pen_str = synthetic.penalty_grid(args.penalties)
    problems = [(0, float(p)) for p in pen_str]
    pset = ProblemSet([(cnt, weight)], problems, device=device)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def one_step():
        f_ms, _ = pset.solve()
        tables = [pset.segments(p) for p in range(len(problems))]
        gathered = gather_segment_tables(tables, dist, device)
        return f_ms, gathered

    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.time()
    fwd = []
    for _ in range(args.steps):
        f_ms, gathered = one_step()
        fwd.append(f_ms)
    sync()
    elapsed = time.time() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # parity gate on what was just computed: structural checks that hold for every solution
    results = [pset.result(p) for p in range(len(problems))]
    for p, r in enumerate(results):
        assert r.status == 0, "problem %d failed with kernel status %d" % (p, r.kernel_status)
        start, mean = pset.segments(p)
        assert len(start) == r.n_segments and len(start) % 2 == 1
        assert start[-1] == -1 and (np.diff(start[:-1]) < 0).all()

    if rank == 0:
        # HBM traffic of one forward launch from the PMC counters (collected separately with
        # rocprofv3 --pmc, see profiles/r01/pmc_traffic.json); only valid for that workload
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            if pmc["workload"]["bins"] == args.bins and \
                    pmc["workload"]["penalties"] == args.penalties:
                traffic = pmc["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        units = args.bins * args.penalties * world * args.steps
        total_pieces = sum(r.total_intervals for r in results)
        alg_bytes = 24.0 * args.bins * args.penalties + 20.0 * total_pieces
        for r in results:  # decoding: one function looked up per segment
            alg_bytes += r.n_segments * (20.0 + 28.0 * r.total_intervals / (2.0 * args.bins))
        fwd_s = float(np.mean(fwd)) / 1e3
        out = {
            "metric": "coverage bins/sec across 64-penalty grid",
            "value": units / elapsed,
            "unit": "bins/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%d-bin synthetic Poisson coverage x %d-penalty grid per GPU "
                            "(BASELINE.json configs[1])" % (args.bins, args.penalties),
                "bins": args.bins, "penalties": args.penalties,
                "penalty_grid": "10^seq(-1,5), 15 significant digits", "seed": "1+rank",
                "sharding": "independent (contig x penalty) problems per rank; RCCL gather of "
                            "segment tables to rank 0",
            },
            "roofline": {
                "bound": "hbm", "kernel": "fpop_forward_kernel",
                "achieved": alg_bytes / fwd_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / fwd_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": float(np.mean(fwd)),  # forward pass + decoding, one launch
                "dp_steps_per_s_per_problem": args.bins / fwd_s,
                "mean_intervals": total_pieces / (2.0 * args.bins * args.penalties),
            },
            "hbm_bytes_resident": pset.hbm_bytes,
            "kernel_build": pset.kernel_build,
            "serial_env_replays": int(sum(r.n_serial_env for r in results)),
        }
        if not args.no_cpu and world == 1:  # the CPU baseline is timed at N=1 only
            out["cpu_baseline"] = cpu_baseline(cs, ce, cnt, pen_str, args.cpu_bins)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    pset.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
