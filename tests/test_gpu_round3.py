"""Round-3 parity tests on an MI355X (through the C ABI): the BASELINE.json configurations
that round 2 had only measured, now compared with the oracle under the driver's `-m gpu` run.

  * configs[3]: `parallel.solve_grid` (the dealing + packing + gather layer) on 24 contigs x 64
    penalties, EVERY problem against the oracle's files; pack/unpack through device tensors;
  * configs[4]: the Worst_case vignette's adversarial counts at 1e5 data points (every step
    through the HBM spill path) and the vignette's own grid (penalties 0, 1e2, 1e4, 1e6 x
    N = 10, 100, 1000 on increasing counts and on Mono27ac prefixes);
  * configs[2]: `sequentialSearch_dir` on a 5e5-bin contig, penalty string for penalty string
    against the same loop driven by the oracle (which runs while the GPU searches);
  * write failures of the three output files injected with RLIMIT_FSIZE
    (reference: tests/testthat/test-TRAVIS-out-of-disk-space.R:37-94 mounts a full tmpfs).
"""
import os
import shutil
import subprocess
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import GOLDEN, ORACLE_DIR, ROOT, read_loss, read_segments

GPU = pytest.mark.gpu
CLI_DET = os.path.join(ORACLE_DIR, "_build", "oracle_cli_det")


@pytest.fixture(scope="module")
def psd():
    import __graft_entry__ as entry
    entry.build_hip()
    entry.build_oracle()
    import peaksegdisk_amd
    from peaksegdisk_amd import _native
    assert _native.lib.peakseg_hip_device_count() >= 1, "no HIP device: GPU tests need an MI355X"
    return peaksegdisk_amd


def run_cli(cli, bg, pen, db):
    st = subprocess.run([cli, bg, pen, db], stdout=subprocess.DEVNULL).returncode
    assert st == 0, (cli, pen, st)
    if os.path.exists(db):
        os.unlink(db)


def check_tables_vs_oracle_files(bg, pen, cs, ce, start, mean, summary):
    """One problem's segment table and summary = [n_segments, n_equality, max_intervals,
    total_intervals, best_cost] against the files the deterministic oracle wrote."""
    segs = read_segments("%s_penalty=%s_segments.bed" % (bg, pen))
    assert len(segs) == len(start) == int(summary[0]), pen
    got_start = np.where(start < 0, int(cs[0]), ce[np.maximum(start, 0)])
    assert np.array_equal(np.array([s[1] for s in segs]), got_start), pen
    assert [s[4] for s in segs] == ["%g" % v for v in mean], pen
    loss = read_loss("%s_penalty=%s_loss.tsv" % (bg, pen)).split("\t")
    n = len(ce)
    assert int(loss[1]) == int(summary[0]) and int(loss[7]) == int(summary[1]), pen
    assert float(loss[9]) == summary[2] and float(loss[8]) == summary[3] / (2.0 * n), pen
    assert loss[5] == "%.20g" % summary[4], pen


@GPU
def test_solve_grid_on_device(psd, tmp_path, n_contigs=24, scale=0.01, n_pen=64, oracle_every=2):
    """BASELINE.json configs[3] on one rank: parallel.solve_grid deals 24 contigs (lengths
    log-uniform, bench.grid_contig_lengths) x 64 penalties = 1536 problems, solves them in
    one device problem set and returns them through the gather layer.  On every contig, every
    second penalty of the grid (768 problems; `oracle_every=1`: all of them, as rounds 2-3 did
    for 50 s of this test on host cores) has its seg_start, "%g" means and summary compared
    with the deterministic oracle's files (a process pool writes those while the GPU works)."""
    import bench
    from peaksegdisk_amd import synthetic
    from peaksegdisk_amd.parallel import solve_grid
    lengths = bench.grid_contig_lengths(n_contigs, scale)
    pen_str = synthetic.penalty_grid(n_pen)
    data, contigs = [], []
    for c, n in enumerate(lengths):
        cs, ce, cnt = synthetic.poisson_coverage(n, seed=101 + c)
        data.append((cs, ce, cnt))
        contigs.append((cnt, (ce - cs).astype(np.int32)))

    def oracle_job(job):
        c, lo, hi = job
        cs, ce, cnt = data[c]
        bg = str(tmp_path / ("c%d_%d" % (c, lo)) / "coverage.bedGraph")
        os.makedirs(os.path.dirname(bg))
        synthetic.write_bedgraph(bg, cs, ce, cnt)
        for pen in pen_str[lo:hi:oracle_every]:
            run_cli(CLI_DET, bg, pen, bg + ".db")
        return c, lo, hi, bg
    # longest contigs first, each contig's penalties in 4 slices: an even pool
    step = max(1, n_pen // 4)
    jobs = [(c, lo, min(n_pen, lo + step)) for c in np.argsort(lengths)[::-1].tolist()
            for lo in range(0, n_pen, step)]
    workers = max(1, bench.host_cores())
    stats = {}
    with ThreadPoolExecutor(max_workers=workers) as pool:
        futs = [pool.submit(oracle_job, j) for j in jobs]
        out = solve_grid(contigs, [float(p) for p in pen_str], None, 0, stats=stats)
        done = [f.result() for f in futs]
    assert len(out) == n_contigs * n_pen
    for c, lo, hi, bg in done:
        cs, ce, cnt = data[c]
        for p in range(lo, hi, oracle_every):
            res = out[(c, p)]
            check_tables_vs_oracle_files(bg, pen_str[p], cs, ce, res["seg_start"],
                                         res["seg_mean"], res["summary"])
    assert stats["problems"] == n_contigs * n_pen


_RCCL_CHILD = r"""
import os, sys
import numpy as np
import torch                      # first: one HIP runtime in the process (as bench.py does)
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from peaksegdisk_amd.parallel import gather_segment_tables, pack_tables, unpack_tables
rng = np.random.default_rng(3)
tables = []
for n in [0, 1, 5, 70000, 3, 0, 1234]:
    tables.append((rng.integers(-2, 2 ** 31 - 1, n).astype(np.int32),
                   rng.standard_normal(n) * 10.0 ** rng.integers(-300, 300, n)))
rows, start, mean = pack_tables(tables)
dev = torch.device("cuda", 0)
back = [torch.from_numpy(a).to(dev).clone().cpu().numpy() for a in (rows, start, mean)]
got = unpack_tables(*back)
assert len(got) == len(tables)
for (s0, m0), (s1, m1) in zip(tables, got):
    assert np.array_equal(s0, s1) and np.array_equal(m0.view(np.uint64), m1.view(np.uint64))
# one rank of RCCL on this GPU: the collective part of the gather
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1,
                        device_id=dev)
out = gather_segment_tables(tables, dist, 0, always_collective=True)
assert len(out) == 1 and len(out[0]) == len(tables)
for (s0, m0), (s1, m1) in zip(tables, out[0]):
    assert np.array_equal(s0, s1) and np.array_equal(m0.view(np.uint64), m1.view(np.uint64))
# round 4: a solved problem set's tables, packed IN HBM by the library and handed to the
# collective as device tensors that alias the packed arrays (no per-problem copy, no numpy on
# the sending side): what every rank > 0 sends to rank 0
from peaksegdisk_amd import ProblemSet, synthetic
from peaksegdisk_amd.parallel import gather_problem_set, packed_payload
cs, ce, cnt = synthetic.poisson_coverage(30000, seed=9)
pens = [0.0, 0.4, 30.0, 2500.0, 1e6]
pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, p) for p in pens])
pset.solve()
payload = packed_payload(pset, dev)
assert all(t.is_cuda for t in payload)
rows, ptr_s, ptr_m, total = pset.pack_tables()
assert payload[1].data_ptr() == ptr_s and payload[2].data_ptr() == ptr_m  # aliases, not copies
got = gather_problem_set(pset, dist, 0, always_collective=True, extra=np.arange(3.0))
assert len(got) == 1 and list(got[0][1]) == [0.0, 1.0, 2.0]
for p, (gs, gm) in enumerate(got[0][0]):
    s0, m0 = pset.segments(p)
    assert len(s0) == pset.result(p).n_segments
    assert np.array_equal(gs, s0) and np.array_equal(gm.view(np.uint64), m0.view(np.uint64))
pset.close()
dist.barrier()
dist.destroy_process_group()
print("rccl-one-rank ok")
"""


@GPU
def test_pack_tables_through_device_tensors(psd):
    """The gather's payload (parallel.pack_tables) survives the trip numpy -> HBM -> numpy
    bit for bit, empty and ragged tables included, and the collective part of
    gather_segment_tables runs on RCCL with one rank on this GPU; and a solved problem set's
    tables go HBM -> RCCL: packed on the device by the library, aliased as cuda tensors, sent
    from there (gather_problem_set).  In a child process: torch ships its own HIP runtime,
    which must be the first one loaded (bench.py imports torch before the library for the
    same reason)."""
    import bench
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", _RCCL_CHILD % {"root": ROOT,
                                                             "port": bench.free_port()}],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "rccl-one-rank ok" in p.stdout, p.stdout + p.stderr


@GPU
def test_adversarial_counts_1e5(psd, oracle_det, tmp_path, n_bins=100000, want_max=741,
                                spill_from=2000):
    """BASELINE.json configs[4] at a tenth of its length: count = 1..N
    (vignettes/Worst_case.Rmd:26-28), penalty 100: every data point beyond the first few
    hundred runs with the lists in the HBM spill area (functions of up to 741 pieces).  Whole
    output files equal the oracle's; the oracle solves while the GPU does."""
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.increasing_coverage(n_bins)
    bg = str(tmp_path / "o" / "coverage.bedGraph")
    os.makedirs(os.path.dirname(bg))
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    th = threading.Thread(target=run_cli, args=(CLI_DET, bg, "100", bg + ".db"))
    th.start()
    gbg = str(tmp_path / "g" / "coverage.bedGraph")
    os.makedirs(os.path.dirname(gbg))
    shutil.copy(bg, gbg)
    pset = ProblemSet([(cnt, (ce - cs).astype(np.int32))], [(0, 100.0)])
    pset.solve()
    r = pset.result(0)
    start, mean = pset.segments(0)
    pset.close()
    assert r.status == 0
    if want_max:
        assert r.max_intervals == want_max
    assert r.spill_steps >= n_bins - spill_from and r.max_intervals > 128
    # and through the file-level boundary: byte-identical files
    from peaksegdisk_amd import _native
    st = _native.lib.PeakSegFPOP_disk(gbg.encode(), b"100", (gbg + ".db").encode())
    assert st == 0
    th.join()
    check_tables_vs_oracle_files(bg, "100", cs, ce, start, mean,
                                 [r.n_segments, r.n_equality_constraints, r.max_intervals,
                                  r.total_intervals, r.best_cost])
    for suffix in ("_segments.bed", "_loss.tsv"):
        assert open(gbg + "_penalty=100" + suffix, "rb").read() == \
            open(bg + "_penalty=100" + suffix, "rb").read()


@GPU
def test_worst_case_vignette_grid(psd, oracle_det, tmp_path, sizes=(10, 100, 1000)):
    """vignettes/Worst_case.Rmd:22-41: penalties {0, 1e2, 1e4, 1e6} x N {10, 100, 1000} on
    increasing counts AND on prefixes of Mono27ac; the vignette states that the large penalty
    on increasing data reaches the (N+1)/2-interval worst case (:48-64).  One batch call;
    every file equals the oracle's, and the worst case is what it says."""
    from peaksegdisk_amd import _native, synthetic
    mono = [l.split("\t") for l in open(os.path.join(GOLDEN, "Mono27ac.bedGraph")).read()
            .splitlines()]
    pens = ["0", "100", "10000", "1000000"]
    files, gfiles = [], []
    for n in sizes:
        cs, ce, cnt = synthetic.increasing_coverage(n)
        for kind in ("increasing", "mono"):
            for root, keep in ((tmp_path / "o", files), (tmp_path / "g", gfiles)):
                d = root / ("%s_%d" % (kind, n))
                d.mkdir(parents=True)
                bg = str(d / "coverage.bedGraph")
                if kind == "increasing":
                    synthetic.write_bedgraph(bg, cs, ce, cnt)
                else:
                    with open(bg, "w") as f:
                        f.write("".join("\t".join(r) + "\n" for r in mono[:n]))
                keep.append((kind, n, bg))
    import ctypes
    bgs = [bg for _, _, bg in gfiles for _ in pens]
    ps = [p for _ in gfiles for p in pens]
    dbs = [bg + "_penalty=%s.db" % p for bg, p in zip(bgs, ps)]
    n = len(bgs)
    arr = lambda xs: (ctypes.c_char_p * n)(*[x.encode() for x in xs])
    status = (ctypes.c_int * n)()
    assert _native.lib.PeakSegFPOP_disk_batch(n, arr(bgs), arr(ps), arr(dbs), status) == 0
    assert list(status) == [0] * n
    for (kind, size, obg), (_, _, gbg) in zip(files, gfiles):
        for pen in pens:
            assert oracle_det.solve(obg, pen) == 0
            for suffix in ("_segments.bed", "_loss.tsv"):
                a = open("%s_penalty=%s%s" % (gbg, pen, suffix), "rb").read()
                b = open("%s_penalty=%s%s" % (obg, pen, suffix), "rb").read()
                assert a == b, (kind, size, pen, suffix)
            # the database left behind (DP branch only) has the size of the reference's
            gdb, odb = "%s_penalty=%s.db" % (gbg, pen), "%s_penalty=%s.db" % (obg, pen)
            assert os.path.exists(gdb) == os.path.exists(odb), (kind, size, pen)
            if os.path.exists(odb):
                assert os.path.getsize(gdb) == os.path.getsize(odb), (kind, size, pen)
        if kind == "increasing":
            # Worst_case.Rmd:48-64: with a penalty, the mean number of intervals grows linearly
            # with N on this data (it stays near 3 at penalty 0), bounded by (N+1)/2
            loss = [read_loss("%s_penalty=%s_loss.tsv" % (gbg, p)).split("\t") for p in pens]
            mean_int = [float(c[8]) for c in loss]
            assert mean_int[0] < 3 and max(mean_int) <= (size + 1) / 2.0
            if size >= 100:
                assert max(mean_int[1:]) > size / 6.0
            if size >= 1000:
                assert max(float(c[9]) for c in loss) > 128  # beyond LDS: the spill path


def _oracle_model(bg, pen_str, cli=CLI_DET):
    run_cli(cli, bg, pen_str, "%s_penalty=%s.db" % (bg, pen_str))
    c = read_loss("%s_penalty=%s_loss.tsv" % (bg, pen_str)).split("\t")
    return {"penalty": pen_str, "peaks": int(c[2]), "total.loss": float(c[6])}


def _oracle_search(problem_dir, peaks_int, cli=CLI_DET, known=None):
    """sequentialSearch_dir (R/sequentialSearch_dir.R:31-99) driven by the oracle executable.
    known: the oracle's models by penalty string, computed beforehand (all at once, in
    parallel); the loop then only looks them up, and a penalty it asks for that is not among
    them fails the test."""
    from peaksegdisk_amd.api import paste
    bg = os.path.join(problem_dir, "coverage.bedGraph")

    def model(pen_str):
        if known is not None:
            assert pen_str in known, "the reference's loop asks for penalty %s" % pen_str
            return known[pen_str]
        return _oracle_model(bg, pen_str, cli)
    with ThreadPoolExecutor(max_workers=2) as pool:  # iteration 1: two independent models
        trace = list(pool.map(model, ["0", "Inf"]))
    over, under = trace[0], trace[1]
    while True:
        if peaks_int in (under["peaks"], over["peaks"]):
            break
        pen = (over["total.loss"] - under["total.loss"]) / (under["peaks"] - over["peaks"])
        if pen < 0:
            break
        m = model(paste(pen))
        trace.append(m)
        if m["peaks"] in (under["peaks"], over["peaks"]):
            break
        if m["peaks"] < peaks_int:
            under = m
        else:
            over = m
    return trace


@GPU
def test_sequential_search_on_a_long_contig(psd, tmp_path, n_bins=500000, peaks_int=354):
    """BASELINE.json configs[2] at a twentieth of its length: sequentialSearch_dir on a 5e5-bin
    synthetic contig.  The resident driver must ask for the models the reference's loop asks
    for: same penalty strings in the same order, same peaks, and the chosen model's files
    byte-identical to the oracle's.  The oracle solves every penalty the search visited (each
    one as soon as the search has left its files) and the reference's loop is then replayed on
    the oracle's results: it must ask for exactly those penalties, in that order.
    (Target 354: on this contig the reference's loop visits penalties 0, Inf, 127.64..., 3059.84...,
    167.49..., 207.54..., 250.14..., 307.86..., 272.41... -- 209836, 0, 7085, 203, 2708, 1080, 470,
    258, 354 peaks -- nine models, eight dynamic programs one after the other.  Rounds 2-3
    searched a 1e6-bin contig for 500 peaks, fifteen models and 141 s of the driver's GPU test
    step; round 4 first cut the path to nine models, then the contig to 5e5 bins: the long
    searches of this suite are Mono27ac's 3198 peaks, tests/test_gpu_round4.py, and the search
    for 613 peaks on 1e6 bins that `PSD_LONG_SEARCH=1` brings back.)"""
    if os.environ.get("PSD_LONG_SEARCH") == "1":
        n_bins, peaks_int = 1000000, 613
    from peaksegdisk_amd import synthetic
    cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=1)
    gdir = tmp_path / "gpu" / "chrSynth-0-1"
    odir = tmp_path / "oracle" / "chrSynth-0-1"
    for d in (gdir, odir):
        d.mkdir(parents=True)
    bg = str(odir / "coverage.bedGraph")
    with open(bg, "w") as f:
        for o in range(0, n_bins, 500000):
            f.write("".join("chrSynth\t%d\t%d\t%d\n" % t for t in zip(
                cs[o:o + 500000].tolist(), ce[o:o + 500000].tolist(), cnt[o:o + 500000].tolist())))
    os.link(bg, str(gdir / "coverage.bedGraph"))
    t0 = time.time()
    # The oracle solves each model WHILE the search goes on (round 4: the driver's GPU test step
    # had 290 s of oracle work on host cores behind the GPU's): the search leaves a model's
    # _loss.tsv the moment that model is done, a watcher starts the oracle on that penalty at
    # once, and when the search returns only the last model's oracle run is still outstanding.
    gbg = str(gdir / "coverage.bedGraph")
    futures, stop = {}, threading.Event()
    pool = ThreadPoolExecutor(max_workers=max(2, len(os.sched_getaffinity(0)) - 1))

    def watch():
        prefix, suffix = os.path.basename(gbg) + "_penalty=", "_loss.tsv"
        while True:
            done = stop.is_set()  # (one more look after the search has returned)
            for name in os.listdir(str(gdir)):
                if name.startswith(prefix) and name.endswith(suffix):
                    pen = name[len(prefix):-len(suffix)]
                    if pen not in futures and os.path.getsize(os.path.join(str(gdir), name)) > 0:
                        futures[pen] = pool.submit(_oracle_model, bg, pen)
            if done:
                return
            time.sleep(0.05)
    watcher = threading.Thread(target=watch)
    watcher.start()
    try:
        fit = psd.sequentialSearch_dir(str(gdir), peaks_int)
    finally:
        stop.set()
        watcher.join()
    gpu_s = time.time() - t0
    got_pen = [psd.paste(float(p)) for p in fit.others["penalty"]]
    assert sorted(futures) == sorted(got_pen)
    known = {pen: futures[pen].result() for pen in got_pen}
    pool.shutdown()
    trace = _oracle_search(str(odir), peaks_int, known=known)
    assert got_pen == [m["penalty"] for m in trace]
    assert list(fit.others["peaks"]) == [m["peaks"] for m in trace]
    chosen = psd.paste(float(fit.loss["penalty"].iloc[0]))
    for suffix in ("_segments.bed", "_loss.tsv"):
        assert open("%s_penalty=%s%s" % (str(gdir / "coverage.bedGraph"), chosen, suffix),
                    "rb").read() == open("%s_penalty=%s%s" % (bg, chosen, suffix), "rb").read()
    print("search of %d bins: %d models, %.1f s on the GPU path, %.1f s with the oracle's models"
          % (n_bins, len(trace), gpu_s, time.time() - t0))


@GPU
def test_checkpointed_store_large_penalties_single_launch(psd, monkeypatch, n_bins=1000000):
    """The checkpointed store cannot park a problem: a block whose records outgrow its region
    during the decoding's recomputation costs a second solve from the first data point.  Large
    penalties on a long contig are where functions are longest (8-9 pieces on average, 25 at
    most): with the default region size they must go through in ONE launch and equal the full
    store bit for bit.  (Round 3 lowered the arena estimate to 7 pieces per function, the
    regions followed to 14, and config 4 at 0.4 of its size took 149 s instead of 77 s.)"""
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=5)
    w = (ce - cs).astype(np.int32)
    # (at this length and these two penalties regions of 14 per function are outgrown)
    problems = [(0, float(p)) for p in ("13894.9549437314", "100000")]
    monkeypatch.delenv("PEAKSEG_HIP_PIECES_PER_FUNCTION", raising=False)
    monkeypatch.setenv("PEAKSEG_HIP_NO_CHECKPOINT", "1")
    full = ProblemSet([(cnt, w)], problems)
    full.solve()
    want = [(full.result(i).n_segments, full.result(i).total_intervals, full.result(i).best_cost)
            + tuple(a.copy() for a in full.segments(i)) for i in range(len(problems))]
    full.close()
    monkeypatch.delenv("PEAKSEG_HIP_NO_CHECKPOINT")
    monkeypatch.setenv("PEAKSEG_HIP_CHECKPOINT", "2048")
    ck = ProblemSet([(cnt, w)], problems)
    ck.solve()
    assert ck.checkpoint_interval == 2048
    assert ck.solve_stats[0] == 1, "a region was outgrown: %r" % (ck.solve_stats,)
    for i in range(len(problems)):
        r = ck.result(i)
        assert r.status == 0
        s1, m1 = ck.segments(i)
        assert (r.n_segments, r.total_intervals, r.best_cost) == want[i][:3]
        assert np.array_equal(s1, want[i][3])
        assert np.array_equal(m1.view(np.uint64), want[i][4].view(np.uint64))
    ck.close()


_FSIZE_CHILD = r"""
import ctypes, os, resource, signal, sys
sys.path.insert(0, %(root)r)
from peaksegdisk_amd import _native
if os.environ.get("PSD_TEST_NATIVE_LIB"):  # CPU rehearsal: kernels under the SIMT emulator
    _native.lib = _native.declare(ctypes.CDLL(os.environ["PSD_TEST_NATIVE_LIB"]))
lib = _native.lib
bg, pen, limit, warm = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
if warm != "-":  # touch the device (and whatever files the runtime creates) before the limit
    assert lib.PeakSegFPOP_disk(warm.encode(), pen.encode(), (warm + ".db").encode()) == 0
signal.signal(signal.SIGXFSZ, signal.SIG_IGN)  # write() then fails with EFBIG, as on a full disk
if limit >= 0:
    resource.setrlimit(resource.RLIMIT_FSIZE, (limit, limit))
st = lib.PeakSegFPOP_disk(bg.encode(), pen.encode(), (bg + ".db").encode())
buf = ctypes.create_string_buffer(2000)
lib.PeakSegFPOP_status_message(st, bg.encode(), pen.encode(), (bg + ".db").encode(), buf, 2000)
sys.stdout.write("%%d\n%%s\n" %% (st, buf.value.decode()))
"""


def run_with_file_size_limit(bg, pen, limit, warm="-"):
    """PeakSegFPOP_disk(bg, pen, bg.db) in a child process whose RLIMIT_FSIZE is `limit` bytes
    (-1: unlimited); returns (status, message of PeakSegFPOP_status_message)."""
    p = subprocess.run([sys.executable, "-c", _FSIZE_CHILD % {"root": ROOT}, bg, pen, str(limit),
                        warm], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    st, msg = p.stdout.split("\n")[:2]
    return int(st), msg


def output_sizes(bg, pen):
    pre = "%s_penalty=%s" % (bg, pen)
    db = bg + ".db"
    return (os.path.getsize(pre + "_loss.tsv"), os.path.getsize(pre + "_segments.bed"),
            os.path.getsize(db) if os.path.exists(db) else 0)


@GPU
def test_write_failures_injected_on_the_dp_branch(psd, tmp_path, n_bins=3000):
    """test-TRAVIS-out-of-disk-space.R:37-94 fills a tiny tmpfs (needs sudo); here a child
    process lowers RLIMIT_FSIZE before the call.  As in the reference's own test, a dynamic
    program under a size limit fails on its cost-function database (code 7, "unable to write to
    cost function database file", :37-41) -- the database of a problem is always larger than
    its two text files, and the reference writes it first -- whatever the limit; the loss /
    segments write errors (8, 11) are reached on the branch without a database (penalty "Inf",
    :84-91): tests/test_host_layers_cpu.py::test_write_failures_on_the_host_only_branch, and
    with unwritable paths in tests/golden/known_answers.json."""
    from peaksegdisk_amd import synthetic
    cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=77)
    warm = str(tmp_path / "warm.bedGraph")
    synthetic.write_bedgraph(warm, cs[:200], ce[:200], cnt[:200])
    ref = str(tmp_path / "unlimited.bedGraph")
    synthetic.write_bedgraph(ref, cs, ce, cnt)
    for pen in ("0", "5000"):
        assert run_with_file_size_limit(ref, pen, -1, warm) == (0, "")
        loss, seg, db = output_sizes(ref, pen)
        assert 0 < loss and 0 < seg and max(loss, seg) < db
        for name, limit in (("tiny", 8), ("between", (max(loss, seg) + db) // 2), ("just", db - 1)):
            bg = str(tmp_path / ("%s_%s.bedGraph" % (name, pen)))
            synthetic.write_bedgraph(bg, cs, ce, cnt)
            st, msg = run_with_file_size_limit(bg, pen, limit, warm)
            assert st == 7, (name, pen, st, msg)
            assert msg == "unable to write to cost function database file " + bg + ".db"
        # a limit of exactly the database's size is enough
        bg = str(tmp_path / ("fits_%s.bedGraph" % pen))
        synthetic.write_bedgraph(bg, cs, ce, cnt)
        assert run_with_file_size_limit(bg, pen, db, warm) == (0, "")


@GPU
def test_batch_with_duplicate_problems(psd, oracle_det, tmp_path, n_bins=2000):
    """The same (bedGraph, penalty) pair listed twice in a batch names one pair of output
    files: it is solved once (two writer threads on the same paths would race) and both
    entries report the same status; a second database name is left as the first is.  The
    same file under ANOTHER name (a symlink) names other output files: they are written too
    (round 3 skipped them and still reported success), from the one dynamic program the pair
    needs.  The directory batch dedupes the same way (one _timing.tsv, both flagged not
    cached)."""
    import ctypes
    from peaksegdisk_amd import _native, synthetic
    cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=31)
    d = tmp_path / "prob"
    d.mkdir()
    bg = str(d / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    obg = str(tmp_path / "oracle.bedGraph")
    shutil.copy(bg, obg)
    link = str(tmp_path / "alias.bedGraph")  # another name of the same file
    os.symlink(bg, link)
    bgs = [bg, bg, link, bg]
    pens = ["25", "25", "25", "700"]
    dbs = [str(tmp_path / ("db%d" % k)) for k in range(4)]
    arr = lambda xs: (ctypes.c_char_p * len(xs))(*[x.encode() for x in xs])
    status = (ctypes.c_int * 4)(9, 9, 9, 9)
    assert _native.lib.PeakSegFPOP_disk_batch(4, arr(bgs), arr(pens), arr(dbs), status) == 0
    assert list(status) == [0, 0, 0, 0]
    for pen in ("25", "700"):
        assert oracle_det.solve(obg, pen) == 0
        for suffix in ("_segments.bed", "_loss.tsv"):
            assert open("%s_penalty=%s%s" % (bg, pen, suffix), "rb").read() == \
                open("%s_penalty=%s%s" % (obg, pen, suffix), "rb").read()
    want = os.path.getsize("%s_penalty=25.db" % obg)
    assert [os.path.getsize(p) for p in dbs[:3]] == [want] * 3
    # the alias names its own output files: the reference would have written them
    for suffix in ("_segments.bed", "_loss.tsv"):
        assert open(link + "_penalty=25" + suffix, "rb").read() == \
            open("%s_penalty=25%s" % (obg, suffix), "rb").read()
    # directory batch: the same pair twice
    dirs = arr([str(d), str(d)])
    st2 = (ctypes.c_int * 2)(9, 9)
    cached = (ctypes.c_int * 2)(9, 9)
    assert _native.lib.PeakSegFPOP_dir_batch(2, dirs, arr(["3000", "3000"]), st2, cached) == 0
    assert list(st2) == [0, 0] and list(cached) == [0, 0]
    assert oracle_det.solve(obg, "3000") == 0
    assert open(bg + "_penalty=3000_loss.tsv", "rb").read() == \
        open(obg + "_penalty=3000_loss.tsv", "rb").read()
    assert len(open(bg + "_penalty=3000_timing.tsv").read().split("\t")) == 3


@GPU
def test_arena_regrowth_resumes_instead_of_repeating(psd, oracle_det, tmp_path, monkeypatch,
                                                     n_bins=20000):
    """An arena estimate that is far too small (one piece per stored function): the kernel parks
    the problems that run out of room, the host adds a segment to the arena and relaunches the
    parked problems from the data point they had reached.  Nothing is computed twice -- the
    launches together work through exactly the problems' data points -- and the store, now
    spread over several segments, still equals the oracle's database byte for byte.  With
    PEAKSEG_HIP_NO_PARK=1 (rounds 1-2: rerun from the start) the result is the same and the
    work is not."""
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=21)
    w = (ce - cs).astype(np.int32)
    pens = ["0.3", "7", "60", "500", "4000", "30000", "200000", "1000000"]
    problems = [(0, float(p)) for p in pens]
    monkeypatch.setenv("PEAKSEG_HIP_NO_CHECKPOINT", "1")
    monkeypatch.setenv("PEAKSEG_HIP_PIECES_PER_FUNCTION", "1")
    # round 4: the arena normally grows WHILE the kernel runs (next test); this test is about
    # what happens when that cannot keep up or is switched off.  Small blocks: the usual 2^19
    # pieces would hold this whole problem set.
    monkeypatch.setenv("PEAKSEG_HIP_NO_LIVE_GROWTH", "1")
    monkeypatch.setenv("PEAKSEG_HIP_ARENA_BLOCK_LOG2", "14" if n_bins >= 10000 else "12")
    pset = ProblemSet([(cnt, w)], problems)
    pset.solve()
    launches, steps = pset.solve_stats
    assert launches >= 2, "the arena was never exhausted"
    assert steps == n_bins * len(pens), (launches, steps)
    assert pset.arena_stats[2] == 0
    bg = str(tmp_path / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    want = []
    for i, pen in enumerate(pens):
        r = pset.result(i)
        assert r.status == 0
        want.append((r.n_segments, r.max_intervals, r.total_intervals, r.best_cost)
                    + pset.segments(i))
        if i in (1, 4, 7):
            db_o = str(tmp_path / ("o%d.db" % i))
            assert oracle_det.solve(bg, pen, db_o) == 0
            db_g = str(tmp_path / ("g%d.db" % i))
            pset.export_db(i, ce, db_g)
            assert open(db_g, "rb").read() == open(db_o, "rb").read(), pen
    # a second solve of the same set reuses the grown arena: one launch
    pset.solve()
    assert pset.solve_stats == (1, n_bins * len(pens))
    pset.close()
    monkeypatch.setenv("PEAKSEG_HIP_NO_PARK", "1")
    rerun = ProblemSet([(cnt, w)], problems)
    rerun.solve()
    launches2, steps2 = rerun.solve_stats
    assert launches2 >= 2 and steps2 > n_bins * len(pens)
    for i in range(len(pens)):
        r = rerun.result(i)
        got = (r.n_segments, r.max_intervals, r.total_intervals, r.best_cost) + rerun.segments(i)
        assert got[:4] == want[i][:4]
        assert np.array_equal(got[4], want[i][4])
        assert np.array_equal(got[5].view(np.uint64), want[i][5].view(np.uint64))
    rerun.close()


@GPU
def test_arena_grows_while_the_kernel_runs(psd, oracle_det, tmp_path, monkeypatch, n_bins=200000,
                                           block_log2=""):
    """Round 4: every arena block is an address range of its own, so the host can map blocks
    while the kernel runs (tools/vmm_block_probe.cpp; behind ONE range the mapping calls wait
    for the kernel).  A set is created with only its first blocks mapped; a second host thread
    maps ahead of what the waves have taken.  With an estimate that is far too small (one piece
    per function) the solve still needs ONE launch, parks nothing, computes every data point
    once, has mapped blocks under the kernel, and the store -- spread over blocks that did not
    exist at launch -- equals the oracle's database byte for byte.  The same with plain
    allocations instead of the virtual-memory calls (PEAKSEG_HIP_NO_VMM=1)."""
    from peaksegdisk_amd import ProblemSet, synthetic
    cs, ce, cnt = synthetic.poisson_coverage(n_bins, seed=23)
    w = (ce - cs).astype(np.int32)
    pens = ["0.3", "7", "60", "500", "4000", "30000", "200000", "1000000"]
    problems = [(0, float(p)) for p in pens]
    monkeypatch.setenv("PEAKSEG_HIP_NO_CHECKPOINT", "1")
    monkeypatch.setenv("PEAKSEG_HIP_PIECES_PER_FUNCTION", "1")
    if block_log2:
        monkeypatch.setenv("PEAKSEG_HIP_ARENA_BLOCK_LOG2", block_log2)
    bg = str(tmp_path / "coverage.bedGraph")
    synthetic.write_bedgraph(bg, cs, ce, cnt)
    want_db = {}
    for no_vmm in (False, True):
        if no_vmm:
            monkeypatch.setenv("PEAKSEG_HIP_NO_VMM", "1")
        pset = ProblemSet([(cnt, w)], problems)
        held_before = pset.hbm_bytes
        block_pieces, blocks_before, _ = pset.arena_stats
        pset.solve()
        assert pset.solve_stats == (1, n_bins * len(pens)), pset.solve_stats
        _, blocks_after, live = pset.arena_stats
        assert live > 0 and blocks_after == blocks_before + live, (blocks_before, blocks_after, live)
        assert pset.arena_bytes_used > blocks_before * block_pieces * 20
        assert pset.hbm_bytes == held_before + live * block_pieces * 20
        for i in (1, 4, 7):
            assert pset.result(i).status == 0
            if i not in want_db:
                db_o = str(tmp_path / ("o%d.db" % i))
                assert oracle_det.solve(bg, pens[i], db_o) == 0
                want_db[i] = open(db_o, "rb").read()
            db_g = str(tmp_path / ("g%d_%d.db" % (i, no_vmm)))
            pset.export_db(i, ce, db_g)
            assert open(db_g, "rb").read() == want_db[i], (pens[i], no_vmm)
        # a second solve finds the arena it needs: nothing is mapped, one launch
        pset.solve()
        assert pset.solve_stats == (1, n_bins * len(pens)) and pset.arena_stats[2] == 0
        pset.close()
