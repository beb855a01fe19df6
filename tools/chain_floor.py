#!/usr/bin/env python3
"""What bounds a data point of the forward kernel from below, as numbers a reader can compare
with the measured cycles per data point.

The HBM roofline says nothing about this kernel: a data point is one
in-order wave working through a chain of DEPENDENT operations, and the same wave also has to
ISSUE every instruction of the step.  Two floors, both built from this repository's own
measurements:

  chain floor  the dependent chain of the slower chain wave: per phase, the operations that
               cannot overlap -- LDS round trips, exp/log evaluations, divisions, Newton trips
               (wave-level trip counts are fixed by bit-exactness: the maximum over the lanes
               of the reference's own iterations), exec-mask regions, cross-lane reductions --
               each priced at the latency tools/*_probe.cpp measured on one wave alone
               (profiles/r03/probe_results.log).  The table below is read off the source and
               checked against the ISA of the built kernel (counts of ds_read / v_rcp_f64 /
               s_cbranch in the hot blocks, profiles/r04/isa_hot_loop_counts.txt).
  issue floor  instructions one chain wave issues per data point (SQ_INSTS_* of the rocprofv3
               PMC pass, minus the helper waves' share) x the issue cost of each class.

A wave alone on its SIMD cannot overlap its own stalls with its own issue, so the measured
time lies between max(chain, issue) and chain + issue.

usage: python tools/chain_floor.py <phase_profile.log> <pmc_traffic.json> <out.json>
"""
import json
import re
import sys

# measured on one wave alone on its SIMD (profiles/r03/probe_results.log, tools/*_probe.cpp)
COST = {
    "lds_round_trip": 110.0,     # dependent LDS read: address, ds_read, s_waitcnt
    "exp": 156.0, "log": 216.0,  # table exp / log (each contains one LDS round trip)
    "div": 84.0,                 # f64 division
    "fma": 4.8,                  # f64 FMA, dependent or not
    "exec_region": 22.0,         # entering an exec-masked region
    "smaller_root_trip": 368.0,  # one Newton trip of get_smaller_root (an exp, a division)
    "larger_root_trip": 477.0,   # one Newton trip of get_larger_root (a log, two divisions)
    "ballot": 8.0, "readlane_pair": 8.0, "bpermute": 110.0, "lds_store_issue": 8.0,
    "uniform_branch": 16.0, "poll_wakeup": 100.0,
}

# Dependent operations per phase of one data point on a chain wave (the LDS-resident step of
# chain_step_fast; file:line = peaksegdisk_amd/csrc).  "par" entries overlap with the longest
# entry of the same phase and are listed for the record only.
PHASES = [
    ("loop top", "fpop_kernels.h forward_body", [
        ("lds_round_trip", 1, "n[] of the two input functions"),
        ("div", 1, "penalty / cum_weight_prev (par: under the round trip)", "par")]),
    ("first pass (piece costs, classes)", "fpop_wave.h piece_costs_wave, min_*_pre", [
        ("lds_round_trip", 1, "coefficients and ends of piece i"),
        ("div", 1, "argmin_mean = -Log/Linear"),
        ("log", 1, "exp(mn) | exp(mx) | log(argmin_mean) interleaved: the log is the longest"),
        ("exp", 1, "cost at the optimum"),
        ("fma", 6, "cost assembly"),
        ("lds_round_trip", 1, "neighbour's left cost for the class of piece i"),
        ("fma", 6, "class decision"),
        ("exec_region", 3, "")]),
    ("speculation round", "fpop_wave.h min_less_impl / min_more_impl", [
        ("fma", 10, "task lane -> (start, piece): integer division by the window"),
        ("lds_round_trip", 1, "level, piece, optimum, end costs of the pair"),
        ("fma", 4, "has_two_roots"),
        ("NEWTON_SPEC", 1, "wave-level trips x cycles per trip"),
        ("ballot", 3, "event / inside / bad masks"),
        ("exec_region", 3, "")]),
    ("walk (state machine)", "fpop_wave.h min_*_impl", [
        ("WALK_ROUNDS", 1, "rounds x (search ballot, 8 readlane pairs, event look-up, 3 uniform "
                           "branches, 12 predicated updates)"),
        ("ballot", 2, "emission masks"),
        ("lds_store_issue", 14, "one parallel pass: two pieces of 7 fields per lane"),
        ("exec_region", 2, "")]),
    ("interval table", "fpop_wave.h min_env_impl", [
        ("lds_round_trip", 1, "own end x"),
        ("TABLE_ENDS", 1, "(n1 + n2) broadcast ends x (readlane pair, compare, add)"),
        ("lds_round_trip", 1, "duplicate test against the other list's end"),
        ("ballot", 2, ""), ("lds_store_issue", 1, "")]),
    ("interval loads", "fpop_wave.h env_load_interval, env_neighbour_flags", [
        ("lds_round_trip", 1, "iv(k)"),
        ("lds_round_trip", 1, "the two pieces of the interval"),
        ("fma", 6, "same_funs, interval ends"), ("ballot", 1, "")]),
    ("classification", "fpop_wave.h env_classify_lanes + helper_root_lanes", [
        ("lds_store_issue", 6, "difference piece to the mailbox"),
        ("CLASSIFY", 1, "max(chain wave: exp|exp, log|log, exp|exp, costs, smaller-root trips, "
                        "early tail exp-log-exp; helper: wake-up, mailbox read, log, exp, "
                        "larger-root trips, final log) -- see 'classification' in the output"),
        ("lds_round_trip", 1, "the helper's roots"),
        ("exp", 1, "cost on the other side of the crossing"),
        ("REDO", 1, "exp-log-exp again when the larger root is the first crossing (share of steps)"),
        ("fma", 20, "decisions (selects)"),
        ("exec_region", 6, "")]),
    ("compaction", "fpop_wave.h min_env_impl", [
        ("ballot", 3, ""), ("bpermute", 1, "predecessor's identity"),
        ("lds_round_trip", 1, "predecessor's fields"),
        ("fma", 6, "coalescing tests"), ("ballot", 3, "heads"),
        ("lds_store_issue", 21, "up to three pieces of 7 fields"),
        ("ballot", 1, "run ends"), ("exec_region", 4, "")]),
    ("rescale + arena record", "fpop_kernels.h scale_add_store_wave", [
        ("div", 1, "1 / cum_weight"),
        ("lds_round_trip", 1, "the function and the cursor's addresses"),
        ("fma", 3, "multiply, add, multiply"),
        ("lds_store_issue", 6, "3 to LDS, 3 to HBM (not waited for)")]),
    ("end of the data point", "fpop_kernels.h step_sync", [
        ("lds_round_trip", 1, "the other wave's arrival flag (it is there already in the floor)"),
        ("lds_round_trip", 1, "abort status")]),
]


def parse_trips(path):
    """wave-level Newton trips and walk rounds per data point of the slowest problems (largest
    cycles per data point) of a tools/phase_profile.py log"""
    rows = []
    for line in open(path):
        m = re.match(r"pen=(\S+)\s+wave(\d) cyc/step=\s*(\d+) mean_int=([\d.]+).*spec=([\d.]+) "
                     r"small=([\d.]+) large=([\d.]+) walk rounds=([\d.]+)", line)
        if m:
            rows.append({"pen": m.group(1), "wave": int(m.group(2)), "cyc": float(m.group(3)),
                         "mean_int": float(m.group(4)), "spec": float(m.group(5)),
                         "small": float(m.group(6)), "rounds": float(m.group(8))})
    worst = max(r["cyc"] for r in rows)
    slow = [r for r in rows if r["cyc"] >= 0.97 * worst]
    out = {}
    for w in (0, 1):
        sel = [r for r in slow if r["wave"] == w] or [r for r in rows if r["wave"] == w]
        out[w] = {k: sum(r[k] for r in sel) / len(sel) for k in ("spec", "small", "rounds", "mean_int")}
    return out, sorted({r["pen"] for r in slow})


def main():
    trips, pens = parse_trips(sys.argv[1])
    pmc = json.load(open(sys.argv[2]))
    sq = pmc["sq_counters_20k_bins"]
    C = COST
    result = {"costs_cycles": C, "slowest_penalties": pens, "waves": {}}
    for w in (0, 1):
        t = trips[w]
        spec_trip = C["smaller_root_trip"] if w == 0 else C["larger_root_trip"]
        n_ends = 2.0 * t["mean_int"] + 2.0  # ends of the walk's result + of the own function
        # classification: the chain wave's chain against its helper's
        chain = (2 * C["exp"] - C["exp"]) + C["log"] + C["exp"] + 6 * C["fma"] + \
            t["small"] * C["smaller_root_trip"] + (C["exp"] + C["log"] + C["exp"] + 6 * C["fma"])
        # (the helper's larger-root solves take about as many trips as the min-more wave's
        # speculation round, whose trips the stamped build does count)
        large_trips = trips[1]["spec"]
        helper = C["poll_wakeup"] + C["lds_round_trip"] + C["log"] + C["exp"] + 6 * C["fma"] + \
            large_trips * C["larger_root_trip"] + C["log"] + C["lds_store_issue"]
        special = {
            "NEWTON_SPEC": t["spec"] * spec_trip,
            "WALK_ROUNDS": t["rounds"] * (C["ballot"] + 8 * C["readlane_pair"] + 60.0 +
                                          3 * C["uniform_branch"] + 12 * C["fma"]),
            "TABLE_ENDS": n_ends * (C["readlane_pair"] + 2 * C["fma"]),
            "CLASSIFY": max(chain, helper),
            "REDO": 0.5 * (C["exp"] + C["log"] + C["exp"]),
        }
        phases = []
        total = 0.0
        for name, where, ops in PHASES:
            cyc = 0.0
            for op in ops:
                if len(op) > 3 and op[3] == "par":
                    continue
                cyc += special[op[0]] if op[0] in special else C[op[0]] * op[1]
            phases.append({"phase": name, "source": where, "cycles": round(cyc, 1)})
            total += cyc
        result["waves"][str(w)] = {
            "chain": "min-less / up" if w == 0 else "min-more / down",
            "wave_level_trips": {"speculation": t["spec"], "smaller_roots": t["small"],
                                 "larger_roots_on_helper": large_trips, "walk_rounds": t["rounds"]},
            "classification": {"chain_wave_before_the_wait": round(chain, 1),
                               "helper_wave": round(helper, 1)},
            "phases": phases, "chain_floor_cycles": round(total, 1)}
    # issue floor: the chain waves' share of the instructions of a data point
    helper_valu = 2.0 * (trips[1]["spec"] * 75.0 + 110.0)  # 75 VALU per larger-root trip (ISA)
    valu = (sq["SQ_INSTS_VALU"] - helper_valu) / 2.0
    salu = (sq["SQ_INSTS_SALU"] - 2.0 * 80.0) / 2.0
    lds = (sq["SQ_INSTS_LDS"] - 2.0 * 30.0) / 2.0
    per_valu = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_INSTS_VALU"]
    per_salu = 4.0 * sq["SQ_ACTIVE_INST_SCA"] / sq["SQ_INSTS_SALU"]
    per_lds = 4.0 * sq["SQ_ACTIVE_INST_LDS"] / sq["SQ_INSTS_LDS"]
    issue = valu * per_valu + salu * per_salu + lds * per_lds
    result["issue_floor"] = {
        "per_chain_wave_instructions": {"valu": round(valu, 1), "salu": round(salu, 1), "lds": round(lds, 1)},
        "busy_cycles_per_instruction": {"valu": round(per_valu, 2), "salu": round(per_salu, 2),
                                        "lds": round(per_lds, 2)},
        "cycles": round(issue, 1),
        "note": "mean over the 64 problems of the PMC run (20 k bins), not the slowest problem"}
    result["chain_floor_cycles_per_step"] = max(result["waves"]["0"]["chain_floor_cycles"],
                                                result["waves"]["1"]["chain_floor_cycles"])
    result["issue_floor_cycles_per_step"] = round(issue, 1)
    with open(sys.argv[3], "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
