#!/usr/bin/env python3
"""Turn what tools/collect_profiles.sh left in gpurun_out/prof/ into the files kept under
profiles/<round>/: CSV summaries of the rocpd databases, the bench logs and pmc_traffic.json
(HBM traffic of one full-size launch; SQ counters per problem and data point).

usage: python tools/update_profiles.py r01"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")


def counters(path, kernel_part):
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel_part in row["Kernel_Name"]:
                out[row["Counter_Name"]] = float(row["Counter_Value_Sum"])
    return out


def main():
    dst = os.path.join(ROOT, "profiles", sys.argv[1])
    os.makedirs(os.path.join(dst, "pmc"), exist_ok=True)
    summ = [sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py")]
    jobs = [("kt/bench100k_results.db", "rocprofv3_bench100k"),
            ("pmc_FETCH_SIZE/pmc_results.db", "pmc/fetch_size_1e6x64"),
            ("pmc_WRITE_SIZE/pmc_results.db", "pmc/write_size_1e6x64"),
            ("pmc_sq1/pmc_results.db", "pmc/sq_pass1_20kx64"),
            ("pmc_sq2/pmc_results.db", "pmc/sq_pass2_20kx64")]
    for db, prefix in jobs:
        subprocess.run(summ + [os.path.join(SRC, db), os.path.join(dst, prefix)], check=True)
    for name in ("bench_1e6x64_full.json.log", "bench100k_under_rocprofv3.json.log"):
        shutil.copy(os.path.join(SRC, name), os.path.join(dst, name))
    fetch = counters(os.path.join(dst, "pmc", "fetch_size_1e6x64_counters.csv"), "fpop_forward")
    write = counters(os.path.join(dst, "pmc", "write_size_1e6x64_counters.csv"), "fpop_forward")
    sq = counters(os.path.join(dst, "pmc", "sq_pass1_20kx64_counters.csv"), "fpop_forward")
    sq.update(counters(os.path.join(dst, "pmc", "sq_pass2_20kx64_counters.csv"), "fpop_forward"))
    steps = 64 * 20000.0  # problems x data points of the SQ runs
    waves = sq.pop("SQ_WAVES")
    per_step = {k: v / steps for k, v in sq.items() if k != "SQ_WAVE_CYCLES"}
    per_step["SQ_WAVE_CYCLES_per_wave"] = sq["SQ_WAVE_CYCLES"] / waves / 20000.0
    per_step["_comment"] = (
        "sums over the launch / (64 problems x 20000 data points): both chain waves and both "
        "helper waves of a problem together; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_* count "
        "quad-cycles (pmc/sq_pass*_counters.csv)")
    out = {
        "_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, "
                    "--kernel-trace only; tools/collect_profiles.sh) of `python3 bench.py --steps 1 "
                    "--warmup 0 --no-cpu` on one MI355X; the forward kernel (with the decoding "
                    "fused in), one launch; values in KiB as reported. Reads are 4-8 B/lane, for "
                    "which the gfx950 FETCH_SIZE undercount (exactly 1/2 for 16 B/lane streams) "
                    "is uncalibrated: reported as is.",
        "workload": {"bins": 1000000, "penalties": 64, "seed": 1},
        "kernel": "psd::lat::fpop_forward_kernel",
        "fetch_size_kib": fetch["FETCH_SIZE"],
        "write_size_kib": write["WRITE_SIZE"],
        "traffic_bytes_per_launch": int((fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024),
        "sq_counters_20k_bins": per_step,
    }
    with open(os.path.join(dst, "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    phase = os.path.join(SRC, "phase_shares.log")
    if os.path.exists(phase):
        shutil.copy(phase, os.path.join(dst, "phase_shares_final_kernel.log"))
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "chain_floor.py"),
                        os.path.join(dst, "phase_shares_final_kernel.log"),
                        os.path.join(dst, "pmc_traffic.json"),
                        os.path.join(dst, "chain_floor.json")], check=True,
                       stdout=subprocess.DEVNULL)


if __name__ == "__main__":
    main()
