"""The N>1 path on CPU: world_size-2 gloo processes exercise problem sharding and the
variable-length gather of segment tables (peaksegdisk_amd/parallel.py) that bench.py and the
multi-GPU driver use over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _tables_for_rank(rank):
    rng = np.random.default_rng(100 + rank)
    tables = []
    for k in range(3 + rank):           # ragged: different number of problems per rank
        n = int(rng.integers(1, 40)) if k != 1 else 1
        start = np.sort(rng.integers(0, 1000, size=n)).astype(np.int32)[::-1].copy()
        start[-1] = -1
        tables.append((start, rng.random(n)))
    return tables


def _worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from peaksegdisk_amd.parallel import gather_segment_tables
    got = gather_segment_tables(_tables_for_rank(rank), dist, None)
    if rank == 0:
        ok = len(got) == world
        for r in range(world):
            want = _tables_for_rank(r)
            ok = ok and len(got[r]) == len(want)
            for (gs, gm), (ws, wm) in zip(got[r], want):
                ok = ok and np.array_equal(gs, ws) and np.array_equal(gm, wm)
        out_q.put(bool(ok))
    else:
        out_q.put(got is None)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_segment_tables_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == [True, True]


def test_shard_problems_balanced_and_deterministic():
    from peaksegdisk_amd.parallel import shard_problems
    rng = np.random.default_rng(0)
    n_bins = np.exp(rng.uniform(np.log(1e5), np.log(1e7), 24))
    costs = [float(n) * (1.0 + 3.0 * p / 63.0) for n in n_bins for p in range(64)]  # 24 x 64
    shards = shard_problems(costs, 8)
    assert sorted(i for s in shards for i in s) == list(range(len(costs)))
    loads = [sum(costs[i] for i in s) for s in shards]
    assert max(loads) / min(loads) < 1.01
    assert shards == shard_problems(costs, 8)
    assert shard_problems([5.0, 1.0], 4)[0] == [0]


def test_pack_unpack_roundtrip_with_empty():
    from peaksegdisk_amd.parallel import pack_tables, unpack_tables
    rows, start, mean = pack_tables([])
    assert len(rows) == 0 and unpack_tables(rows, start, mean) == []
    t = _tables_for_rank(1)
    back = unpack_tables(*pack_tables(t))
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(t, back))


def _grid_worker(rank, world, port, out_q):
    """2 gloo ranks solve a 3-contig x 4-penalty grid with the emulated kernels; rank 0 checks
    every problem against a single-process solve of the same grid."""
    import ctypes
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from peaksegdisk_amd import _native, synthetic
    from peaksegdisk_amd.parallel import solve_grid
    emu = _native.declare(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "_build",
                                                   "libpeaksegdisk_emu.so")))
    contigs = []
    for k, n in enumerate((300, 500, 200)):
        cs, ce, cnt = synthetic.poisson_coverage(n, seed=50 + k)
        contigs.append((cnt, (ce - cs).astype(np.int32)))
    pens = [0.3, 7.0, 150.0, 4000.0]
    got = solve_grid(contigs, pens, dist, None, lib=emu)
    ok = True
    if rank == 0:
        ref = solve_grid(contigs, pens, None, 0, lib=emu)
        ok = set(got) == set(ref) and len(got) == 12
        for key in ref:
            ok = ok and np.array_equal(got[key]["seg_start"], ref[key]["seg_start"])
            ok = ok and np.array_equal(got[key]["seg_mean"], ref[key]["seg_mean"])
            ok = ok and np.array_equal(got[key]["summary"], ref[key]["summary"])
            ok = ok and got[key]["summary"][0] == len(got[key]["seg_start"])
    else:
        ok = got is None
    out_q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_solve_grid_sharded_gloo_world2():
    import subprocess
    import torch.multiprocessing as mp
    import __graft_entry__ as entry
    entry.build_hip()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")], check=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    results = [q.get(timeout=10) for _ in procs]
    assert results == [True, True]
