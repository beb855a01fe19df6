"""Host-side layers that need no GPU: R's paste() port, the PeakSegFPOP_dir cache predicate in
the batch entry, the penalty search's failure mode without a device, the R glue file, and
bench.py started exactly as the driver starts it (`python bench.py --gpus 2`, no torchrun
environment: two gloo ranks, kernels under the SIMT emulator)."""
import ctypes
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as entry
    entry.build_hip()
    from peaksegdisk_amd import _native
    return _native


def test_paste_double_matches_r_mirror(native):
    """peakseg_hip_paste_double (penalties of the search, _timing.tsv) against the Python
    mirror of R's paste(), including the 11 penalties of the reference's Mono27ac search
    (tests/golden/known_answers.json, survey_8c)."""
    from peaksegdisk_amd import api
    rng = np.random.default_rng(7)
    vals = [0.0, 1.0, 0.1, 1e5, 1e15, 1e16, 1e-5, 1.5e-10, float("inf"), 866939314852865280.0,
            100000.0, 1234567.0, 2 / 3, 1e22, 12345678901234.5, 0.000123456789012345678]
    vals += list(rng.uniform(1, 10, 4000) * 10.0 ** rng.uniform(-20, 25, 4000))
    vals += [round(float(v), int(d)) for v, d in zip(rng.uniform(0, 1e6, 2000),
                                                   rng.integers(0, 7, 2000))]
    buf = ctypes.create_string_buffer(64)
    for v in vals:
        native.lib.peakseg_hip_paste_double(v, buf, 64)
        assert buf.value.decode() == api.paste(v), v
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        trace = json.load(f)["mono27ac"]["sequential_search_19"]["penalties"]
    for s in trace:
        native.lib.peakseg_hip_paste_double(float(s), buf, 64)
        assert buf.value.decode() == s


def _fake_cached_dir(d, pen="10.5", bases=40, consistent=True):
    """A problem directory whose result files pass (or fail) the reference's cache test
    (R/PeakSegFPOP_dir.R:70-93) without any solver having run."""
    os.makedirs(d, exist_ok=True)
    bg = os.path.join(d, "coverage.bedGraph")
    with open(bg, "w") as f:
        f.write("chr1\t0\t10\t2\nchr1\t10\t20\t10\nchr1\t20\t30\t14\nchr1\t30\t40\t13\n")
    pre = "%s_penalty=%s" % (bg, pen)
    with open(pre + "_segments.bed", "w") as f:
        f.write("chr1\t30\t40\tbackground\t12.3333\nchr1\t10\t30\tpeak\t12.3333\n"
                "chr1\t%d\t10\tbackground\t2\n" % (0 if consistent else 5))
    with open(pre + "_loss.tsv", "w") as f:
        f.write("10.5\t3\t1\t%d\t4\t-13.5729\t-553.416\t1\t1.25\t2\n" % bases)
    with open(pre + "_timing.tsv", "w") as f:
        f.write("10.5\t0.000392913818359375\t0.01\n")
    return bg, pre


def _dir_batch(native, dirs, pens):
    n = len(dirs)
    a = (ctypes.c_char_p * n)(*[os.fsencode(d) for d in dirs])
    b = (ctypes.c_char_p * n)(*[p.encode() for p in pens])
    status = (ctypes.c_int * n)()
    cached = (ctypes.c_int * n)()
    rc = native.lib.PeakSegFPOP_dir_batch(n, a, b, status, cached)
    return rc, list(status), list(cached)


def test_dir_batch_cache_protocol_without_gpu(native, tmp_path):
    """Consistent result files are reused (no solver, hence no GPU needed); anything the
    reference's predicate rejects is recomputed -- which, without a device, is status 12."""
    if native.lib.peakseg_hip_device_count() > 0:
        pytest.skip("a GPU is visible: the recompute branch would succeed")
    good = str(tmp_path / "good")
    _fake_cached_dir(good)
    rc, status, cached = _dir_batch(native, [good], ["10.5"])
    assert (rc, status, cached) == (0, [0], [1])
    before = open(os.path.join(good, "coverage.bedGraph_penalty=10.5_loss.tsv")).read()
    # inconsistent: segments do not start where the coverage starts
    bad = str(tmp_path / "bad")
    _fake_cached_dir(bad, consistent=False)
    # inconsistent: bases of the loss file do not match the segments
    bad2 = str(tmp_path / "bad2")
    _fake_cached_dir(bad2, bases=41)
    # empty timing file (test-CRAN-PeakSegFPOP_dir.R:38-57)
    bad3 = str(tmp_path / "bad3")
    _, pre3 = _fake_cached_dir(bad3)
    open(pre3 + "_timing.tsv", "w").close()
    # a different penalty has no files at all
    rc, status, cached = _dir_batch(native, [good, bad, bad2, bad3, good],
                                    ["10.5", "10.5", "10.5", "10.5", "11"])
    assert cached == [1, 0, 0, 0, 0]
    assert status == [0, 12, 12, 12, 12] and rc == 12
    assert open(os.path.join(good, "coverage.bedGraph_penalty=10.5_loss.tsv")).read() == before
    # validation errors keep the reference's codes in the batch as well
    rc, status, cached = _dir_batch(native, [good, good, str(tmp_path / "nowhere")],
                                    ["foobar", "-1", "3"])
    assert status == [10, 2, 3]


def test_sequential_search_without_gpu(native, tmp_path):
    """The search's first model is penalty "0": a dynamic program, so without a device the
    call fails with status 12 before anything is returned (no CPU fallback); bad arguments
    are refused."""
    if native.lib.peakseg_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    d = tmp_path / "prob"
    d.mkdir()
    shutil.copy(os.path.join(GOLDEN, "Mono27ac.bedGraph"), str(d / "coverage.bedGraph"))
    rows = (native.PsdSearchRow * 8)()
    n = ctypes.c_int(-1)
    chosen = ctypes.c_int(0)
    st = native.lib.PeakSegFPOP_sequential_search(os.fsencode(str(d)), 19, 0, 8, rows,
                                                  ctypes.byref(n), ctypes.byref(chosen))
    assert st == native.ERROR_NO_HIP_DEVICE and n.value == 0 and chosen.value == -1
    st = native.lib.PeakSegFPOP_sequential_search(os.fsencode(str(d)), -1, 0, 8, rows,
                                                  ctypes.byref(n), ctypes.byref(chosen))
    assert st == native.ERROR_SEARCH_ARGUMENTS
    from peaksegdisk_amd import api
    with pytest.raises(api.PeakSegError) as e:
        api.sequentialSearch_dir(str(d), 19)
    assert e.value.status == 12
    # the batched search: every directory fails alone with the same status
    d2 = tmp_path / "prob2"
    d2.mkdir()
    shutil.copy(os.path.join(GOLDEN, "Mono27ac.bedGraph"), str(d2 / "coverage.bedGraph"))
    dirs = (ctypes.c_char_p * 2)(os.fsencode(str(d)), os.fsencode(str(d2)))
    peaks = (ctypes.c_int * 2)(19, 3)
    rows2 = (native.PsdSearchRow * 16)()
    n2, chosen2, st2 = (ctypes.c_int * 2)(), (ctypes.c_int * 2)(), (ctypes.c_int * 2)()
    rc = native.lib.PeakSegFPOP_sequential_search_batch(2, dirs, peaks, 0, 8, rows2, n2, chosen2,
                                                        st2)
    assert rc == native.ERROR_NO_HIP_DEVICE and list(st2) == [12, 12] and list(chosen2) == [-1, -1]
    # the same directory twice is refused for the second one
    dirs = (ctypes.c_char_p * 2)(os.fsencode(str(d)), os.fsencode(str(d)))
    native.lib.PeakSegFPOP_sequential_search_batch(2, dirs, peaks, 0, 8, rows2, n2, chosen2, st2)
    assert list(st2) == [12, native.ERROR_SEARCH_ARGUMENTS]
    with pytest.raises(api.PeakSegError) as e:
        api.sequentialSearch_dir_batch([str(d)], 19)
    assert e.value.status == 12
    assert api.sequentialSearch_dir_batch([], 3) == []
    with pytest.raises(ValueError):
        api.sequentialSearch_dir_batch([str(d)], [1, 2])


def test_r_glue_file(tmp_path):
    """r/src/interface.cpp registers the reference's routine with the reference's signature
    (src/interface.cpp:8-11,58-73).  With R installed it is compiled against R's headers; in
    this image it is syntax-checked against declaration-only shims of the R names it uses."""
    src = os.path.join(ROOT, "r", "src", "interface.cpp")
    text = open(src).read()
    assert '{"PeakSegFPOP_interface", (DL_FUNC)&PeakSegFPOP_interface, 3, PeakSegFPOP_types}' in text
    assert "R_useDynamicSymbols(info, FALSE)" in text and "R_init_PeakSegDisk" in text
    assert "{STRSXP, STRSXP, STRSXP}" in text
    r_home = shutil.which("R")
    if r_home:
        inc = subprocess.run(["R", "CMD", "config", "--cppflags"], capture_output=True,
                             text=True).stdout.split()
        cmd = ["g++", "-std=gnu++17", "-c", "-fPIC", "-I" + os.path.join(ROOT, "include")] + \
            inc + [src, "-o", str(tmp_path / "interface.o")]
    else:
        cmd = ["g++", "-std=gnu++17", "-fsyntax-only", "-Wall", "-Werror",
               "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "tests", "r_shim"), src]
    subprocess.run(cmd, check=True)
    makevars = open(os.path.join(ROOT, "r", "src", "Makevars")).read()
    assert "-lpeaksegdisk_hip" in makevars


def _emu_lib():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
    return os.path.join(ROOT, "tests", "emu", "_build", "libpeaksegdisk_emu.so")


def _run_bench(extra, env_extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env,
                          capture_output=True, text=True, timeout=600)


def test_bench_launches_its_ranks_as_the_driver_does():
    """`python bench.py --gpus 2 --steps K --warmup W` with no torchrun environment must run
    TWO ranks and report n_gpus == 2 (round 1 ignored --gpus)."""
    env = {"PSD_BENCH_BACKEND": "gloo", "PSD_BENCH_TEST_LIB": _emu_lib()}
    p = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--bins", "150",
                    "--penalties", "3", "--no-cpu"], env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert abs(d["value"] - 150 * 3 * 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    # configs[3] as a bench mode: fixed work dealt over the ranks
    p = _run_bench(["--gpus", "2", "--mode", "grid", "--steps", "1", "--warmup", "0",
                    "--grid-contigs", "5", "--grid-scale", "0.002", "--penalties", "3"], env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["contigs"] == 5 and d["value"] > 0


def test_bench_refuses_a_world_that_is_not_gpus():
    env = {"PSD_BENCH_BACKEND": "gloo", "PSD_BENCH_TEST_LIB": _emu_lib(), "WORLD_SIZE": "1",
           "RANK": "0", "LOCAL_RANK": "0"}
    p = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--bins", "100",
                    "--penalties", "2", "--no-cpu"], env)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_bench_single_rank_line_on_emulator():
    """N = 1 through the emulator: the JSON line carries roofline, cpu_baseline with both
    legs (one core, all cores) and the upload-inclusive rate."""
    env = {"PSD_BENCH_BACKEND": "gloo", "PSD_BENCH_TEST_LIB": _emu_lib()}
    p = _run_bench(["--steps", "1", "--warmup", "0", "--bins", "400", "--penalties", "3",
                    "--cpu-bins", "200", "--cpu-bins-all", "400"], env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert cb["all_cores"]["cores"] >= 1 and cb["all_cores"]["value"] > 0 and cb["cpu_model"]
    assert "all 3 penalties" in cb["sample"]
    assert d["value_incl_upload"] > 0 and d["value_incl_upload"] <= d["value"] * 1.0001


def test_write_failures_on_the_host_only_branch(tmp_path):
    """tests/testthat/test-TRAVIS-out-of-disk-space.R:37-94 needs a full filesystem (a tmpfs
    mounted with sudo); a child process under RLIMIT_FSIZE gives the same write errors.  Here
    the closed-form branch (penalty "Inf", drv:224-243: no device work): the loss file cannot
    be written -> code 8, only the segments file cannot -> code 11, with the reference's
    messages (interface.cpp:38-46).  The DP branch is covered on the GPU
    (test_gpu_round3.test_write_failures_injected_on_the_dp_branch)."""
    from test_gpu_round3 import output_sizes, run_with_file_size_limit
    chrom = "chr" + "x" * 90  # a segments row longer than the loss row
    rows = "".join("%s\t%d\t%d\t%d\n" % (chrom, 10 * i, 10 * i + 10, c)
                   for i, c in enumerate([2, 10, 14, 13]))
    ref = str(tmp_path / "unlimited.bedGraph")
    open(ref, "w").write(rows)
    assert run_with_file_size_limit(ref, "Inf", -1) == (0, "")
    loss, seg, db = output_sizes(ref, "Inf")
    assert 0 < loss < seg and db == 0
    for name, limit, code, text in [
            ("loss", loss - 1, 8, "unable to write to loss output file"),
            ("segments", (loss + seg) // 2, 11, "unable to write to segments output file")]:
        bg = str(tmp_path / (name + ".bedGraph"))
        open(bg, "w").write(rows)
        st, msg = run_with_file_size_limit(bg, "Inf", limit)
        assert st == code, (name, st, msg)
        assert text in msg and bg + "_penalty=Inf" in msg, msg
