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


def _worker(rank, world, port, out_q):
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
                "--penalties", "4", "--no-cpu"]
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
