#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output (the default output format of this ROCm) into the
small CSV summaries kept under profiles/: per-kernel statistics (the `--stats` table) and,
for --pmc runs, the per-kernel counter sums.

usage: python tools/rocpd_summary.py <results.db> <out_prefix>"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    prefix = sys.argv[2]
    cur = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in cur:
            w.writerow(r)
    cur = db.execute("select kernel_name, counter_name, count(*), sum(value) from "
                     "counters_collection group by kernel_name, counter_name")
    rows = cur.fetchall()
    if rows:
        with open(prefix + "_counters.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Counter_Value_Sum"])
            for r in rows:
                w.writerow(r)


if __name__ == "__main__":
    main()
