"""Multi-GPU layer: (contig x penalty) problems are independent, so
they are dealt to ranks with no data-path collective; the only exchange is one
variable-length gather of the segment tables to rank 0 (RCCL over xGMI on GPUs; gloo in the
CPU tests).  One process per GPU, torch.distributed for the plumbing."""
import numpy as np


def shard_problems(costs, world_size):
    """Longest-processing-time-first dealing of problems to ranks.

    costs[i] is the predicted cost of problem i (bins x a penalty factor).  Returns a list of
    index lists, one per rank; deterministic (ties broken by problem index)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += costs[i]
    for s in shards:
        s.sort()
    return shards


def pack_tables(tables):
    """tables: list of (seg_start int32[n], seg_mean float64[n]).  Returns (rows int64[k],
    start int32[sum n], mean float64[sum n])."""
    rows = np.array([len(s) for s, _ in tables], dtype=np.int64)
    if len(tables):
        start = np.concatenate([np.asarray(s, dtype=np.int32) for s, _ in tables])
        mean = np.concatenate([np.asarray(m, dtype=np.float64) for _, m in tables])
    else:
        start = np.zeros(0, dtype=np.int32)
        mean = np.zeros(0, dtype=np.float64)
    return rows, start, mean


def unpack_tables(rows, start, mean):
    out = []
    o = 0
    for n in rows.tolist():
        out.append((start[o:o + n].copy(), mean[o:o + n].copy()))
        o += n
    return out


def gather_segment_tables(tables, dist=None, device=None, always_collective=False):
    """Gather every rank's segment tables on rank 0.

    dist is torch.distributed (already initialised) or None for a single process.  Returns on
    rank 0 a list over ranks of lists of (seg_start, seg_mean); on other ranks None.
    Protocol: one all_gather of the per-rank totals (rows, segments),
    then every other rank sends its three packed arrays to rank 0 at their exact sizes --
    grouped point-to-point sends/receives (ncclSend/ncclRecv inside one group on RCCL; xGMI is
    point-to-point, so this is the natural shape and no rank is padded to the largest table:
    penalty ~ 0 on a 1e7-bin contig is 5.5 M rows, a large penalty a few thousand).
    always_collective: run the all_gather even for a single rank (tests: RCCL on one GPU)."""
    if dist is None or (dist.get_world_size() == 1 and not always_collective):
        return [tables]
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    dev = torch.device("cuda", device) if dist.get_backend() == "nccl" else torch.device("cpu")
    rows, start, mean = pack_tables(tables)
    meta = torch.tensor([len(rows), len(start)], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().tolist() for m in metas]
    dtypes = (torch.int64, torch.int32, torch.float64)
    ops, bufs = [], {}
    if rank == 0:
        for r in range(1, world):
            n_rows, n_seg = metas[r]
            bufs[r] = [torch.empty(n, dtype=dt, device=dev)
                       for n, dt in zip((n_rows, n_seg, n_seg), dtypes)]
            ops += [dist.P2POp(dist.irecv, t, r) for t in bufs[r] if t.numel() > 0]
    else:
        payload = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (rows, start, mean)]
        ops = [dist.P2POp(dist.isend, t, 0) for t in payload if t.numel() > 0]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != 0:
        return None
    out = [tables]
    for r in range(1, world):
        out.append(unpack_tables(*[t.cpu().numpy() for t in bufs[r]]))
    return out


def predicted_cost(n_bins, penalty_rank, n_penalties):
    """Relative cost of one problem for load balancing: proportional to the contig length,
    rising about 2x from the smallest to the largest penalty of a log-spaced grid (measured:
    28 k -> 47 k cycles per data point, profiles/r01/phase_shares_stamped_build.log)."""
    return float(n_bins) * (1.0 + penalty_rank / max(1.0, n_penalties - 1.0))


def solve_grid(contigs, penalties, dist=None, device=0, lib=None, stats=None):
    """Solve every (contig, penalty) problem of a grid across the ranks of `dist`
    (BASELINE.json configs[3]: 24 contigs x 64 penalties over 8 GPUs).

    contigs: list of (count, weight) int32 arrays, identical on every rank; penalties: list of
    floats.  Problems are dealt to ranks longest-first; each rank uploads only the contigs it
    needs, solves its shard in one problem set, and rank 0 receives everything through one
    gather.  Returns on rank 0 a dict (contig_index, penalty_index) -> dict(seg_start,
    seg_mean, summary) where summary = [n_segments, n_equality, max_intervals,
    total_intervals, best_cost]; None on other ranks.  stats (a dict, optional) receives this
    rank's forward_ms (HIP events), kernel_build, hbm_bytes and the wall seconds of its phases
    (create_s = upload + arena, solve_s, tables_s = download of the tables, close_s)."""
    from .grid import ProblemSet
    world = 1 if dist is None else dist.get_world_size()
    rank = 0 if dist is None else dist.get_rank()
    order = sorted(range(len(penalties)), key=lambda i: penalties[i])
    prank = {i: r for r, i in enumerate(order)}
    problems = [(c, p) for c in range(len(contigs)) for p in range(len(penalties))]
    costs = [predicted_cost(len(contigs[c][0]), prank[p], len(penalties)) for c, p in problems]
    mine = shard_problems(costs, world)[rank]
    used = sorted({problems[i][0] for i in mine})
    local_of = {c: k for k, c in enumerate(used)}
    tables = []
    if mine:
        import time
        t0 = time.time()
        pset = ProblemSet([contigs[c] for c in used],
                          [(local_of[problems[i][0]], penalties[problems[i][1]]) for i in mine],
                          device=device or 0, lib=lib)
        t1 = time.time()
        f_ms, _ = pset.solve()
        t2 = time.time()
        if stats is not None:
            stats.update(forward_ms=f_ms, kernel_build=pset.kernel_build,
                         hbm_bytes=pset.hbm_bytes, problems=len(mine),
                         checkpoint_interval=pset.checkpoint_interval,
                         launches=pset.solve_stats[0])
        for k in range(len(mine)):
            r = pset.result(k)
            start, mean = pset.segments(k)
            # one extra row carries the summary so that a single gather moves everything
            summary = np.array([r.n_segments, r.n_equality_constraints, r.max_intervals,
                                r.total_intervals, r.best_cost], dtype=np.float64)
            tables.append((np.concatenate([start, np.full(5, -2, np.int32)]),
                           np.concatenate([mean, summary])))
        t3 = time.time()
        pset.close()
        if stats is not None:  # where this rank's wall time went, seconds
            stats.update(create_s=t1 - t0, solve_s=t2 - t1, tables_s=t3 - t2,
                         close_s=time.time() - t3)
    gathered = gather_segment_tables(tables, dist, device)
    if rank != 0:
        return None
    shards = shard_problems(costs, world)
    out = {}
    for r in range(world):
        for k, i in enumerate(shards[r]):
            start, mean = gathered[r][k]
            out[problems[i]] = {"seg_start": start[:-5], "seg_mean": mean[:-5],
                                "summary": mean[-5:]}
    return out
