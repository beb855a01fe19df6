"""bench.py's N>1 path (per-rank problem sets, barrier + max-over-ranks timing, gather of the
segment tables, rank-0 JSON line) rehearsed on CPU: two gloo ranks, kernels under the SIMT
emulator, a tiny workload."""
import io
import json
import os
import socket
import sys

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_q, extra_args=()):
    import contextlib
    import ctypes
    sys.path.insert(0, ROOT)
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                       "PSD_BENCH_BACKEND": "gloo"})
    from peaksegdisk_amd import _native
    _native.lib = _native.declare(ctypes.CDLL(
        os.path.join(ROOT, "tests", "emu", "_build", "libpeaksegdisk_emu.so")))
    import bench
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--bins", "400",
                "--penalties", "4", "--no-cpu"] + list(extra_args)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    out_q.put((rank, buf.getvalue()))


def test_bench_two_ranks_gloo():
    import subprocess
    import torch.multiprocessing as mp
    import __graft_entry__ as entry
    entry.build_hip()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    outs = dict(q.get(timeout=10) for _ in procs)
    assert outs[1].strip() == ""            # only rank 0 prints
    lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "bins/s" and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 0 and abs(d["value"] - 400 * 4 * 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) \
        < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0
    assert d["vs_baseline"] is None and d["dtype"] == "f64"


def _run_ranks(world, extra_args=()):
    import subprocess
    import torch.multiprocessing as mp
    import __graft_entry__ as entry
    entry.build_hip()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, tuple(extra_args)))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    outs = dict(q.get(timeout=10) for _ in procs)
    for r in range(1, world):
        assert outs[r].strip() == ""        # only rank 0 prints
    lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_eight_ranks_weak_gloo():
    """The N = 8 line of the driver's scaling run, rehearsed: 8 gloo ranks, each its own contig x
    4 penalties; the line's arithmetic and rank 0 holding all 8 x 4 tables at the end."""
    d = _run_ranks(8)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["steps"] == 2
    assert abs(d["value"] - 400 * 4 * 8 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    assert d["tables_on_rank0"] == 8 * 4
    assert d["config"]["bins"] == 400 and d["config"]["penalties"] == 4


def test_bench_eight_ranks_grid_gloo():
    """BASELINE.json configs[3] on 8 ranks (tiny contigs): problems dealt to the ranks, one
    gather, rank 0 ends with every (contig, penalty) table; strong scaling arithmetic."""
    import bench
    d = _run_ranks(8, ["--mode", "grid", "--grid-contigs", "12", "--grid-scale", "0.0006",
                       "--penalties", "3"])
    lengths = bench.grid_contig_lengths(12, 0.0006)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["config"]["total_bins"] == sum(lengths) and d["config"]["contigs"] == 12
    assert abs(d["value"] - sum(lengths) * 3 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    assert d["tables_on_rank0"] == 12 * 3
