"""CPU-side checks of the real HIP library (no compute calls): it loads, exports every symbol
that include/peaksegdisk_hip.h declares, keeps the reference's validation order and status
texts on the paths that never reach the GPU, and fails loudly -- no CPU fallback -- when a
dynamic program is requested without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as entry
    entry.build_hip()
    from peaksegdisk_amd import _native
    return _native


def test_every_declared_symbol_is_exported(native):
    header = open(os.path.join(ROOT, "include", "peaksegdisk_hip.h")).read()
    declared = set(re.findall(r"\b(PeakSegFPOP_\w+|peakseg_hip_\w+)\s*\(", header))
    declared.discard("PeakSegFPOP_interface")
    assert declared == set(native.EXPORTED_SYMBOLS)
    raw = ctypes.CDLL(native.LIB_PATH)
    for name in sorted(declared):
        assert getattr(raw, name) is not None, name


def test_library_is_the_in_tree_hip_build(native):
    assert native.LIB_PATH.startswith(os.path.join(ROOT, "peaksegdisk_amd", "lib"))
    blob = open(native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob            # carries a gfx950 code object
    assert b"fpop_forward_kernel" in blob


def _has_gpu(native):
    return native.lib.peakseg_hip_device_count() > 0


def test_status_messages_match_reference_glue(native):
    """texts of /root/reference/src/interface.cpp:16-55"""
    m = native.status_message
    assert m(0, "f", "1", "d") == ""
    assert m(1, "f", "NAN", "d") == "penalty=NAN but must be finite"
    assert m(2, "f", "-0.1", "d") == "penalty=-0.1 must be non-negative"
    assert m(3, "f", "1", "d") == "unable to open input file for reading f"
    assert m(4, "f", "1", "d") == "each line of input data file f should have exactly four columns"
    assert m(5, "f", "1", "d") == "fourth column of input data file f should be integer"
    assert m(6, "f", "1", "d") == "there should be no gaps (columns 2-3) in input data file f"
    assert m(7, "f", "1", "d") == "unable to write to cost function database file d"
    assert m(8, "f", "1", "d") == "unable to write to loss output file f_penalty=1_loss.tsv"
    assert m(11, "f", "1", "d") == "unable to write to segments output file f_penalty=1_segments.bed"
    assert m(9, "f", "1", "d") == "input file f contains no data"
    assert m(10, "f", "foobar", "d") == \
        "penalty string 'foobar' is not numeric; it should be convertible to double"
    assert m(99, "f", "1", "d") == "error code 99"


def test_host_only_paths_need_no_gpu(native, known_answers, tmp_path):
    """penalty / input validation and the trivial one-segment model never touch the device."""
    lib = native.lib
    for i, case in enumerate(known_answers["error_cases"]):
        if case.get("block") in ("segments", "loss", "db"):
            continue  # these reach the DP branch
        d = tmp_path / ("e%d" % i)
        d.mkdir()
        bg = str(d / "missing") if case["bedGraph"] is None else str(d / "coverage.bedGraph")
        if case["bedGraph"] is not None:
            open(bg, "w").write(case["bedGraph"])
        st = lib.PeakSegFPOP_disk(bg.encode(), case["penalty"].encode(), (bg + ".db").encode())
        assert st == case["status"], case["name"]
    # trivial models: penalty Inf, constant data
    bg = str(tmp_path / "coverage.bedGraph")
    open(bg, "w").write("chr1\t0\t10\t2\nchr1\t10\t20\t10\n")
    assert lib.PeakSegFPOP_disk(bg.encode(), b"Inf", (bg + ".db").encode()) == 0
    assert open(bg + "_penalty=Inf_segments.bed").read() == "chr1\t0\t20\tbackground\t6\n"
    loss = open(bg + "_penalty=Inf_loss.tsv").read().split("\t")
    assert loss[:5] == ["Inf", "1", "0", "20", "2"]
    open(bg, "w").write("chr6\t1\t2\t5\nchr6\t2\t3\t5\nchr6\t3\t4\t5\n")
    assert lib.PeakSegFPOP_disk(bg.encode(), b"0", (bg + ".db").encode()) == 0
    assert open(bg + "_penalty=0_segments.bed").read() == "chr6\t1\t4\tbackground\t5\n"


def test_dp_without_gpu_fails_loudly(native, tmp_path):
    if _has_gpu(native):
        pytest.skip("a GPU is visible: the no-device path cannot be exercised here")
    bg = str(tmp_path / "coverage.bedGraph")
    open(bg, "w").write("chr1\t0\t10\t2\nchr1\t10\t20\t10\nchr1\t20\t30\t14\n")
    st = native.lib.PeakSegFPOP_disk(bg.encode(), b"10.5", (bg + ".db").encode())
    assert st == native.ERROR_NO_HIP_DEVICE
    assert "no CPU fallback" in native.last_error()
    import peaksegdisk_amd as psd
    with pytest.raises(psd.PeakSegError) as ei:
        psd.PeakSegFPOP_file(bg, "10.5")
    assert ei.value.status == native.ERROR_NO_HIP_DEVICE
    with pytest.raises(RuntimeError):
        psd.ProblemSet([(np.array([1, 2], np.int32), np.array([1, 1], np.int32))], [(0, 1.0)])
    x = np.zeros(4)
    assert native.lib.peakseg_hip_math_probe(0, 4, x.ctypes.data, x.ctypes.data) == \
        native.ERROR_NO_HIP_DEVICE


def test_product_never_touches_the_oracle():
    """Nothing under peaksegdisk_amd/ may include, link or load anything under oracle/ or the
    emulator (tests/emu)."""
    pkg = os.path.join(ROOT, "peaksegdisk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "oracle/" not in text.replace(
                    "the CPU\n * oracle in oracle/", ""), os.path.join(dirpath, f)
                assert "libpeaksegdisk_emu" not in text, f
    blob = open(os.path.join(pkg, "lib", "libpeaksegdisk_hip.so"), "rb").read()
    assert b"oracle_PeakSegFPOP_disk" not in blob


def test_r_paste_formatting():
    """paste() of the doubles the reference's R layer turns into penalty strings."""
    from peaksegdisk_amd import paste
    assert paste(0) == "0" and paste(0.0) == "0"
    assert paste(float("inf")) == "Inf"
    assert paste(1952.6) == "1952.6"
    assert paste(10.5) == "10.5"
    assert paste(1e5) == "1e+05"
    assert paste(123456.0) == "123456"
    assert paste(0.1) == "0.1"
    assert paste(0.0001) == "1e-04"
    assert paste(157.99473732931699033) == "157.994737329317"
    assert paste(1952.6687694641800423) == "1952.66876946418"
    assert paste(866939314852865280.0) == "866939314852865280"
    assert paste("10.50") == "10.50"


def test_header_is_plain_c(tmp_path):
    """include/peaksegdisk_hip.h must be consumable from C (the R glue and cgo-style bindings):
    no C++ or torch types in the signatures."""
    import subprocess
    src = tmp_path / "use.c"
    src.write_text('#include "peaksegdisk_hip.h"\n'
                   'int use(char *a, char *b, char *c) { psd_result r; (void)r;\n'
                   '  return PeakSegFPOP_disk(a, b, c); }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only",
                    "-I" + os.path.join(ROOT, "include"), str(src)], check=True)


def test_detmath_header_is_plain_c(tmp_path):
    import subprocess
    src = tmp_path / "m.c"
    src.write_text('#include "peakseg_detmath.h"\n'
                   'double f(double x) { return psd_exp(x) + psd_log(x); }\n')
    subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-fsyntax-only", "-ffp-contract=off",
                    "-I" + os.path.join(ROOT, "include"), str(src)], check=True)


def test_fast_parser_equals_sscanf(native, tmp_path):
    """The byte-scanning bedGraph parser must behave exactly like the reference's
    sscanf("%s %d %d %d%s") on every line: fuzz well-formed and malformed files and compare
    status, line count and a hash of everything parsed with the sscanf-only path."""
    import random
    rng = random.Random(7)
    lib = native.lib

    def probe(path, fast):
        n = ctypes.c_int()
        h = ctypes.c_ulonglong()
        st = lib.peakseg_hip_parse_probe(path.encode(), fast, ctypes.byref(n), ctypes.byref(h))
        return st, n.value, h.value

    def good_line(start, end, cnt, sep="\t"):
        return "chr%d%s%d%s%d%s%d" % (rng.randrange(30), sep, start, sep, end, sep, cnt)

    mutations = [
        lambda s: s + " extra", lambda s: s + ".5", lambda s: s.replace("\t", "  \t "),
        lambda s: "  " + s, lambda s: s + " \r", lambda s: s.rsplit("\t", 1)[0],
        lambda s: "", lambda s: "   ", lambda s: s.replace("\t", ",", 1),
        lambda s: s + "\t", lambda s: s.replace("\t", "\t+", 1),
        lambda s: s.replace("\t", "\t-", 1), lambda s: s[:-1] + "x",
        lambda s: "c" * 150 + s[4:], lambda s: s.rsplit("\t", 1)[0] + "\t12345678901",
        lambda s: s.rsplit("\t", 1)[0] + "\t0x10", lambda s: s.replace("\t", "\v", 1),
    ]
    statuses = set()
    for case in range(400):
        lines = []
        pos = rng.randrange(1000)
        for _ in range(rng.randrange(1, 12)):
            w = rng.randrange(1, 50)
            lines.append(good_line(pos, pos + w, rng.randrange(0, 40),
                                   sep=rng.choice(["\t", " ", "  ", "\t "])))
            pos += w
        if case % 3:
            i = rng.randrange(len(lines))
            lines[i] = rng.choice(mutations)(lines[i])
        if rng.random() < 0.1:
            lines[rng.randrange(len(lines))] = good_line(5, 3, 1)  # gap / reversed
        text = "\n".join(lines) + ("\n" if rng.random() < 0.7 else "")
        path = str(tmp_path / ("p%d.bedGraph" % case))
        open(path, "w").write(text)
        a, b = probe(path, 1), probe(path, 0)
        assert a == b, (case, text, a, b)
        statuses.add(a[0])
    assert {0, 4, 5, 6} <= statuses
    open(str(tmp_path / "empty"), "w").write("")
    assert probe(str(tmp_path / "empty"), 1) == probe(str(tmp_path / "empty"), 0)
    assert probe(str(tmp_path / "empty"), 1)[0] == 9


def test_kernel_resource_budgets(tmp_path):
    """The builds of the forward kernel are defined by their occupancy (DESIGN.md sections 3, 8):
    the throughput build must fit 2 waves per SIMD and 4 workgroups per CU (<= 40 KB LDS), the
    packed build 3 waves per SIMD and 6 workgroups per CU (<= 26.6 KB) -- a register or LDS
    regression silently costs a third to a half of their throughput -- and the latency build
    must fit one workgroup of 4 waves.  Checked on the compiler's resource remarks (no GPU
    needed)."""
    import re
    import subprocess
    import __graft_entry__ as entry
    csrc = os.path.join(ROOT, "peaksegdisk_amd", "csrc")
    out = subprocess.run(
        [entry.HIPCC] + entry.HIP_FLAGS + [ "-Rpass-analysis=kernel-resource-usage",
         "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
         os.path.join(csrc, "peakseg_hip.cpp"), "-o", str(tmp_path / "lib.so")],
        capture_output=True, text=True, check=True).stderr
    info = {}
    name = None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            info[name] = {}
        for key in ("Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and name:
                info[name][key] = int(m.group(1))
    thr = next(v for k, v in info.items() if "3thr19fpop_forward_kernel" in k)
    lat = next(v for k, v in info.items() if "3lat19fpop_forward_kernel" in k)
    pk = next(v for k, v in info.items() if "2pk19fpop_forward_kernel" in k)
    assert pk["Occupancy [waves/SIMD]"] >= 3, pk
    assert pk["LDS Size [bytes/block]"] <= 160 * 1024 // 6, pk
    assert thr["Occupancy [waves/SIMD]"] >= 2, thr
    assert thr["LDS Size [bytes/block]"] <= 40 * 1024, thr
    assert lat["Occupancy [waves/SIMD]"] >= 1 and lat["LDS Size [bytes/block]"] <= 160 * 1024, lat
