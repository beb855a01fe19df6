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
