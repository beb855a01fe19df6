#!/usr/bin/env python3
"""Instruction counts of the built forward kernel (latency build), read off its ISA: what
tools/chain_floor.py's table of dependent operations is checked against, and what the register
allocator had to do to fit the kernel (spills into vector lanes / accumulation registers).

Compiles peakseg_hip.cpp for the device only (-S), takes psd::lat::fpop_forward_kernel and prints
  * resource usage (registers, scratch, LDS),
  * instructions by class over the whole kernel (hot and cold blocks together),
  * the Newton loops: innermost loops that contain two / one f64 divisions and a table look-up
    (get_larger_root: a log and two divisions per trip; get_smaller_root: an exp and one), with
    their instruction counts per trip.

usage: python tools/isa_counts.py [out.txt]   (no GPU needed)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

KERNEL = "_ZN3psd3lat19fpop_forward_kernelENS_10DeviceArgsE"


def main():
    csrc = os.path.join(ROOT, "peaksegdisk_amd", "csrc")
    work = tempfile.mkdtemp(prefix="psd_isa_")
    asm = os.path.join(work, "dev.s")
    flags = [f for f in entry.HIP_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run([entry.HIPCC] + flags + ["--cuda-device-only", "-S",
                    "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                    os.path.join(csrc, "peakseg_hip.cpp"), "-o", asm], check=True,
                   stderr=subprocess.DEVNULL)
    text = open(asm).read()
    start = text.index("\n" + KERNEL + ":")
    end = text.index(".end_amdhsa_kernel", start)
    body = text[start:end]
    lines = [ln.split(";")[0].strip() for ln in body.splitlines()]  # (comments follow a ';')
    insts = [ln for ln in lines if ln and not ln.startswith((".", ";")) and not ln.endswith(":")]
    out = []
    out.append("psd::lat::fpop_forward_kernel (%s), %s" % (
        " ".join(f for f in entry.HIP_FLAGS if f.startswith("-O") or f.startswith("-f")), "gfx950"))
    for key in ("next_free_vgpr", "next_free_sgpr", "accum_offset", "private_segment_fixed_size",
                "group_segment_fixed_size"):
        m = re.search(r"\.amdhsa_%s (\d+)" % key, body)
        if m:
            out.append("  %-28s %s" % (key, m.group(1)))
    for key in ("sgpr_spill_count", "vgpr_spill_count"):
        m = re.search(r"\.name:\s+%s.*?\.%s:\s+(\d+)" % (KERNEL, key), text, re.S)
        if m:
            out.append("  %-28s %s" % (key, m.group(1)))
    classes = [
        ("all instructions", lambda i: True),
        ("VALU (v_*)", lambda i: i.startswith("v_")),
        ("  f64 arithmetic", lambda i: re.match(r"v_(fma|fmac|mul|add|div_\w+|rcp|min|max)_f64", i)),
        ("  divisions (v_rcp_f64)", lambda i: i.startswith("v_rcp_f64")),
        ("  v_readlane / v_writelane", lambda i: i.startswith(("v_readlane", "v_writelane"))),
        ("  v_accvgpr_read / write (VGPR spills to AGPRs)", lambda i: i.startswith("v_accvgpr")),
        ("SALU (s_*)", lambda i: i.startswith("s_")),
        ("  branches", lambda i: i.startswith(("s_cbranch", "s_branch"))),
        ("  s_waitcnt", lambda i: i.startswith("s_waitcnt")),
        ("  calls (s_swappc)", lambda i: i.startswith("s_swappc")),
        ("LDS reads (ds_read*)", lambda i: i.startswith("ds_read")),
        ("LDS writes (ds_write*)", lambda i: i.startswith("ds_write")),
        ("LDS other (ds_bpermute, ds_add, ...)", lambda i: i.startswith("ds_") and not i.startswith(("ds_read", "ds_write"))),
        ("global loads", lambda i: i.startswith("global_load")),
        ("global stores", lambda i: i.startswith("global_store")),
        ("scratch (register spills to memory)", lambda i: i.startswith("scratch_")),
    ]
    out.append("instructions in the kernel's body (hot and cold blocks):")
    for name, pred in classes:
        out.append("  %-50s %6d" % (name, sum(1 for i in insts if pred(i))))
    # innermost loops: a label, then a backward branch to it with no other label in between
    out.append("Newton loops (innermost loops with f64 divisions and a table look-up):")
    label_at = {}
    seq = []
    for ln in lines:
        if ln.endswith(":") and ln.startswith(".LBB"):
            label_at[ln[:-1]] = len(seq)
        elif ln and not ln.startswith((".", ";")):
            seq.append(ln)
    labels_sorted = sorted(label_at.items(), key=lambda kv: kv[1])
    for name, at in labels_sorted:
        nxt = min([a for _, a in labels_sorted if a > at] + [len(seq)])
        block = seq[at:nxt]
        back = [k for k, i in enumerate(block) if re.match(r"s_cbranch_\w+\s+%s\b" % re.escape(name), i)]
        if not back:
            continue
        block = block[:back[0] + 1]
        n_div = sum(1 for i in block if i.startswith("v_rcp_f64"))
        n_lds = sum(1 for i in block if i.startswith("ds_read"))
        if n_div == 0 or n_lds == 0:
            continue
        kind = "larger root (log, two divisions)" if n_div == 2 else "smaller root (exp, one division)"
        out.append("  %-12s %-36s %3d instructions per trip: %d VALU, %d SALU, %d LDS" % (
            name, kind, len(block), sum(1 for i in block if i.startswith("v_")),
            sum(1 for i in block if i.startswith("s_")), n_lds))
    text_out = "\n".join(out) + "\n"
    sys.stdout.write(text_out)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(text_out)


if __name__ == "__main__":
    main()
