"""Shared pytest plumbing.

`-m "not gpu"` tests: the oracle against the golden vectors, host logic, C-ABI symbols.
`-m gpu` tests: the HIP path through the C-ABI against the oracle (needs an MI355X).
"""
import ctypes
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The three GPU tests that take a minute or more run LAST, longest last: if a slow box runs the
# driver's `pytest -x` into its time limit, the kill loses the fewest tests.
_LONG_GPU_TESTS = ["test_endpoints_vs_glibc_arithmetic_1e6_x64", "test_sequential_search_on_a_long_contig",
                   "test_endpoints_vs_glibc_arithmetic_1e7_x64"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = item.name.split("[")[0]
        return _LONG_GPU_TESTS.index(name) + 1 if name in _LONG_GPU_TESTS else 0
    items.sort(key=rank)  # stable: everything else keeps its order


@pytest.fixture(autouse=True)
def _flush_c_stdio():
    """The oracle prints the reference's messages with printf: flush them into the test's own
    captured output instead of leaving them for process exit (after pytest's summary)."""
    yield
    ctypes.CDLL(None).fflush(None)


def _build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


class Oracle:
    """ctypes view of oracle/_build/liboracle_<kind>.so (test infrastructure)."""

    def __init__(self, kind):
        path = os.path.join(ORACLE_DIR, "_build", "liboracle_%s.so" % kind)
        self.lib = ctypes.CDLL(path)
        self.lib.oracle_PeakSegFPOP_disk.argtypes = [ctypes.c_char_p] * 3
        self.lib.oracle_PeakSegFPOP_disk.restype = ctypes.c_int
        self.lib.oracle_math_kind.restype = ctypes.c_char_p
        self.kind = kind

    def solve(self, bedgraph, penalty, db=None):
        if db is None:
            db = "%s_penalty=%s.db" % (bedgraph, penalty)
        return self.lib.oracle_PeakSegFPOP_disk(
            os.fsencode(bedgraph), penalty.encode(), os.fsencode(db))

    def math(self, fn, x):
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        getattr(self.lib, "oracle_%s_vec" % fn)(
            ctypes.c_int(x.size), x.ctypes.data_as(ctypes.c_void_p),
            y.ctypes.data_as(ctypes.c_void_p))
        return y


@pytest.fixture(scope="session")
def oracle_libm():
    _build_oracle()
    return Oracle("libm")


@pytest.fixture(scope="session")
def oracle_det():
    _build_oracle()
    return Oracle("det")


@pytest.fixture(scope="session")
def known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)


def read_segments(path):
    rows = []
    with open(path) as f:
        for line in f:
            c = line.rstrip("\n").split("\t")
            rows.append([c[0], int(c[1]), int(c[2]), c[3], c[4]])
    return rows


def read_loss(path):
    with open(path) as f:
        return f.read().rstrip("\n")


def read_segment_columns(path):
    """A _segments.bed file as columns (chromStart, chromEnd int64 arrays; status, mean text
    arrays), parsed by pandas' C reader: the files of the small penalties of a 1e6-bin contig
    have half a million rows each."""
    import pandas as pd
    df = pd.read_csv(path, sep="\t", header=None, names=["chrom", "start", "end", "status", "mean"],
                     dtype={"chrom": str, "start": "int64", "end": "int64", "status": str,
                            "mean": str}, na_filter=False)
    return (df["start"].to_numpy(), df["end"].to_numpy(), df["status"].to_numpy(),
            df["mean"].to_numpy())


def format_g(values):
    """the reference prints segment means with the stream's default precision: "%g" """
    return ["%g" % v for v in values.tolist()]


class BackgroundOracles:
    """Oracle runs of the full-size GPU tests, started when the session starts and collected by
    the tests that need them -- which run last (pytest_collection_modifyitems), so the host
    cores work through the oracle's minutes while the GPU runs the other tests.  Niced: the
    other tests' own oracle runs come first.  (Test infrastructure; round 3's GPU test step
    spent 290 of its 615 s waiting for oracle processes.)"""

    def __init__(self, root):
        import threading
        self.root = root
        self.jobs = {}      # name -> {"thread": Thread, "procs": {(cli, pen): Popen}, ...}
        # a few cores stay free for the tests in the foreground (their own oracle runs, the
        # host side of the GPU runs): 76 processes at once made the small tests 35 s slower
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 2
        self.slots = threading.Semaphore(max(2, cores - 6))

    def start(self, name, n_bins, seed, runs):
        """runs: list of (cli path, penalty string, subdir): the bedGraph is written once per
        subdir (hard links of one file), each run is one process"""
        import threading
        job = {"procs": {}, "error": None, "dirs": {}}
        self.jobs[name] = job

        def work():
            try:
                import numpy as np
                sys.path.insert(0, ROOT)
                from peaksegdisk_amd import synthetic
                cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=seed)
                base = os.path.join(self.root, name)
                os.makedirs(base)
                first = None
                for _, _, sub in runs:
                    if sub in job["dirs"]:
                        continue
                    d = os.path.join(base, sub)
                    os.makedirs(d)
                    bg = os.path.join(d, "coverage.bedGraph")
                    if first is None:
                        with open(bg, "w") as f:
                            for o in range(0, n_bins, 500000):
                                f.write("".join("chrSynth\t%d\t%d\t%d\n" % t for t in zip(
                                    cs[o:o + 500000].tolist(), ce[o:o + 500000].tolist(),
                                    cnt[o:o + 500000].tolist())))
                        first = bg
                    else:
                        os.link(first, bg)
                    job["dirs"][sub] = bg
                job["data"] = (cs, ce, cnt)
                def reap(proc):
                    proc.wait()
                    self.slots.release()
                for k, (cli, pen, sub) in enumerate(runs):
                    bg = job["dirs"][sub]
                    job.setdefault("dbs", {})[(sub, pen)] = os.path.join(base, "%s_%d.db" % (sub, k))
                    self.slots.acquire()
                    if job.get("closed"):
                        self.slots.release()
                        return
                    proc = subprocess.Popen(
                        ["nice", "-n", "15", cli, bg, pen, os.path.join(base, "%s_%d.db" % (sub, k))],
                        stdout=subprocess.DEVNULL)
                    job["procs"][(sub, pen)] = proc
                    threading.Thread(target=reap, args=(proc,), daemon=True).start()
            except Exception as e:  # reported by wait()
                job["error"] = e
        job["thread"] = threading.Thread(target=work)
        job["thread"].start()

    def wait(self, name):
        """-> (bedGraph path per subdir, db path per (subdir, penalty), (cs, ce, cnt)) once
        every process of the job has ended with status 0"""
        job = self.jobs[name]
        job["thread"].join()
        if job["error"] is not None:
            raise job["error"]
        for key, proc in job["procs"].items():
            assert proc.wait() == 0, key
        return job["dirs"], job["dbs"], job["data"]

    def close(self):
        for job in self.jobs.values():
            job["closed"] = True
        for job in self.jobs.values():
            for proc in list(job["procs"].values()):
                if proc.poll() is None:
                    proc.kill()
            job["thread"].join()
            for proc in list(job["procs"].values()):
                if proc.poll() is None:
                    proc.kill()


LONG_1E6 = {"name": "grid_1e6", "n_bins": 1000000, "seed": 1, "det_picks": [4, 12, 20, 28, 36, 44, 52, 63]}
LONG_1E7 = {"name": "grid_1e7", "n_bins": 10000000, "seed": 1, "picks": [9, 27, 44, 60]}


@pytest.fixture(scope="session")
def background_oracles(request, tmp_path_factory):
    """Started by the first GPU test module that runs: the oracle processes of the two full-size
    grid tests."""
    names = {item.name.split("[")[0] for item in request.session.items}
    bo = BackgroundOracles(str(tmp_path_factory.mktemp("bg_oracles")))
    if os.environ.get("PSD_NO_BACKGROUND_ORACLES"):
        yield bo
        return
    _build_oracle()
    sys.path.insert(0, ROOT)
    from peaksegdisk_amd import synthetic
    pens = synthetic.penalty_grid(64)
    cli_libm = os.path.join(ORACLE_DIR, "_build", "oracle_cli_libm")
    cli_det = os.path.join(ORACLE_DIR, "_build", "oracle_cli_det")
    if "test_endpoints_vs_glibc_arithmetic_1e7_x64" in names:
        # first: these take two to three minutes each
        bo.start(LONG_1E7["name"], LONG_1E7["n_bins"], LONG_1E7["seed"],
                 [(cli_libm, pens[i], "libm") for i in LONG_1E7["picks"]])
    if "test_endpoints_vs_glibc_arithmetic_1e6_x64" in names:
        bo.start(LONG_1E6["name"], LONG_1E6["n_bins"], LONG_1E6["seed"],
                 [(cli_libm, pens[i], "libm") for i in range(64)] +
                 [(cli_det, pens[i], "det") for i in LONG_1E6["det_picks"]])
    yield bo
    bo.close()


@pytest.fixture(scope="session", autouse=True)
def _start_background_oracles(request):
    """`-m gpu` runs: the full-size tests' oracle processes start with the session."""
    if request.config.getoption("markexpr", "") == "gpu":
        request.getfixturevalue("background_oracles")
    yield
