"""Multi-GPU layer: (contig x penalty) problems are independent, so
they are dealt to ranks with no data-path collective; the only exchange is one
variable-length gather of the segment tables to rank 0 (RCCL over xGMI on GPUs; gloo in the
CPU tests).  One process per GPU, torch.distributed for the plumbing.

Round 4: a rank's payload goes HBM -> RCCL.  The library packs the segment tables at their
exact sizes in HBM (one kernel over all problems of the set) and the packed arrays are handed
to the grouped send as device tensors that alias them -- no hipMemcpy per problem, no numpy
round trip on the sending side.  Rank 0 downloads each rank's packed arrays once."""
import ctypes

import numpy as np

# cycles per data point of one problem on the latency build, by the problem's mean number of
# intervals -- what a rank's solve measured last (ProblemSet.cycles_per_step) replaces this
# table; it only has to order problems sensibly before anything has been measured
# (profiles/r03/phase_shares_head_round3_final_kernel.log: 19 k at 2.3 intervals, 28 k at 8-9)
DEFAULT_CYCLES_PER_STEP = (19000.0, 28000.0)


def shard_problems(costs, world_size):
    """Longest-processing-time-first dealing of problems to ranks.

    costs[i] is the predicted cost of problem i.  Returns a list of index lists, one per rank;
    deterministic (ties broken by problem index)."""
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


class _DevicePointer:
    """what torch.as_tensor needs to alias device memory it did not allocate"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False),
                                         "version": 2}


def device_array(ptr, n, dtype, torch_device):
    """A torch tensor over n elements of `dtype` at device address ptr, without a copy.  On the
    CPU rehearsal (gloo + the SIMT emulator) "device" memory is host memory."""
    import torch
    dtype = np.dtype(dtype)
    if n == 0:
        return torch.empty(0, dtype=getattr(torch, dtype.name), device=torch_device)
    if torch_device.type == "cuda":
        return torch.as_tensor(_DevicePointer(ptr, n, dtype.str), device=torch_device)
    buf = (ctypes.c_char * (n * dtype.itemsize)).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=dtype))


def packed_payload(pset, torch_device):
    """The solved set's segment tables packed at their exact sizes as tensors on torch_device:
    [rows int64[k], start int32[sum], mean float64[sum]].  The library packs them IN HBM (one
    kernel, peakseg_hip_problem_set_pack_tables); on a cuda device the packed arrays are
    aliased, not copied -- they go to RCCL from where they are -- otherwise (gloo rehearsal,
    single process) they are downloaded once."""
    import torch
    rows, ptr_s, ptr_m, total = pset.pack_tables()
    rows_t = torch.from_numpy(rows).to(torch_device)
    if torch_device.type == "cuda":
        return [rows_t, device_array(ptr_s, total, np.int32, torch_device),
                device_array(ptr_m, total, np.float64, torch_device)]
    start, mean = pset.packed_download(total)
    return [rows_t, torch.from_numpy(start), torch.from_numpy(mean)]


def gather_packed(payload, dist, torch_device):
    """payload: this rank's tensors on torch_device (any number of 1-d arrays; the same number
    and dtypes on every rank).  Returns on rank 0 a list over ranks of lists of numpy arrays
    (rank 0's own payload included), None elsewhere.  Protocol: one
    all_gather of the lengths, then every other rank sends its arrays to rank 0 at their exact
    sizes -- grouped point-to-point sends/receives (ncclSend/ncclRecv inside one group on RCCL;
    xGMI is point-to-point, so this is the natural shape and no rank is padded to the largest
    table: penalty ~ 0 on a 1e7-bin contig is 5.5 M rows, a large penalty a few thousand)."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    meta = torch.tensor([int(t.numel()) for t in payload], dtype=torch.int64, device=torch_device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().tolist() for m in metas]
    ops, bufs = [], {}
    if rank == 0:
        for r in range(1, world):
            bufs[r] = [torch.empty(n, dtype=t.dtype, device=torch_device)
                       for n, t in zip(metas[r], payload)]
            ops += [dist.P2POp(dist.irecv, t, r) for t in bufs[r] if t.numel() > 0]
    else:
        ops = [dist.P2POp(dist.isend, t.contiguous(), 0) for t in payload if t.numel() > 0]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != 0:
        return None
    out = [[t.cpu().numpy() for t in payload]]
    for r in range(1, world):
        out.append([t.cpu().numpy() for t in bufs[r]])
    return out


def _torch_device(dist, device):
    import torch
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", device or 0)
    return torch.device("cpu")


def gather_segment_tables(tables, dist=None, device=None, always_collective=False):
    """Gather every rank's segment tables (host arrays) on rank 0.

    dist is torch.distributed (already initialised) or None for a single process.  Returns on
    rank 0 a list over ranks of lists of (seg_start, seg_mean); on other ranks None.
    always_collective: run the collective even for a single rank (tests: RCCL on one GPU)."""
    if dist is None or (dist.get_world_size() == 1 and not always_collective):
        return [tables]
    import torch
    dev = _torch_device(dist, device)
    payload = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in pack_tables(tables)]
    got = gather_packed(payload, dist, dev)
    if got is None:
        return None
    return [tables] + [unpack_tables(*arrays) for arrays in got[1:]]


def gather_problem_set(pset, dist=None, device=None, always_collective=False, extra=None):
    """Gather the segment tables of every rank's solved ProblemSet on rank 0: packed in HBM
    (packed_payload) and, on RCCL, sent from there.  pset None: a rank without problems (it
    still takes part in the exchange).  extra: an optional float64 array, any length per
    rank, that travels with the tables (solve_grid's summaries).  Returns on rank 0 a list over
    ranks of (tables, extra) with tables a list of (seg_start, seg_mean); None on other ranks."""
    import torch
    collective = dist is not None and (dist.get_world_size() > 1 or always_collective)
    dev = _torch_device(dist, device) if collective else torch.device("cpu")
    if pset is None:
        payload = [torch.zeros(0, dtype=dt, device=dev)
                   for dt in (torch.int64, torch.int32, torch.float64)]
    else:
        payload = packed_payload(pset, dev)
    extra = np.zeros(0) if extra is None else extra
    payload.append(torch.from_numpy(np.ascontiguousarray(extra, dtype=np.float64)).to(dev))
    if not collective:
        arrays = [t.numpy() for t in payload]
        return [(unpack_tables(*arrays[:3]), arrays[3])]
    got = gather_packed(payload, dist, dev)
    if got is None:
        return None
    return [(unpack_tables(*arrays[:3]), arrays[3]) for arrays in got]


def predicted_cost(n_bins, penalty_rank, n_penalties, cycles_per_step=None):
    """Cost of one problem for the dealing, in shader cycles: data points x cycles per data
    point.  cycles_per_step: what the library measured (a sequence indexed by penalty rank, from
    ProblemSet.cycles_per_step of an earlier solve of the same grid, all ranks agreeing on it);
    before any measurement a ramp between the two ends of DEFAULT_CYCLES_PER_STEP."""
    if cycles_per_step is not None and len(cycles_per_step) == n_penalties:
        return float(n_bins) * float(cycles_per_step[penalty_rank])
    lo, hi = DEFAULT_CYCLES_PER_STEP
    return float(n_bins) * (lo + (hi - lo) * penalty_rank / max(1.0, n_penalties - 1.0))


def solve_grid(contigs, penalties, dist=None, device=0, lib=None, stats=None,
               cycles_per_step=None):
    """Solve every (contig, penalty) problem of a grid across the ranks of `dist`
    (BASELINE.json configs[3]: 24 contigs x 64 penalties over 8 GPUs).

    contigs: list of (count, weight) int32 arrays, identical on every rank; penalties: list of
    floats.  Problems are dealt to ranks longest-first (predicted_cost; cycles_per_step: the
    measured cycles per data point by penalty rank, identical on every rank, e.g. what an
    earlier call returned in stats["cycles_per_step"]); each rank uploads only the contigs it
    needs, solves its shard in one problem set, and rank 0 receives everything through one
    device-to-device gather.  Returns on rank 0 a dict (contig_index, penalty_index) ->
    dict(seg_start, seg_mean, summary) where summary = [n_segments, n_equality,
    max_intervals, total_intervals, best_cost]; None on other ranks.  stats (a dict, optional)
    receives this rank's forward_ms (HIP events), kernel_build, hbm_bytes, the wall seconds of
    its phases (create_s = upload + first arena blocks, solve_s, tables_s = packing + gather,
    close_s) and cycles_per_step (by penalty rank, averaged over the ranks that measured it)."""
    import time
    from .grid import ProblemSet
    world = 1 if dist is None else dist.get_world_size()
    rank = 0 if dist is None else dist.get_rank()
    order = sorted(range(len(penalties)), key=lambda i: penalties[i])
    prank = {i: r for r, i in enumerate(order)}
    problems = [(c, p) for c in range(len(contigs)) for p in range(len(penalties))]
    costs = [predicted_cost(len(contigs[c][0]), prank[p], len(penalties), cycles_per_step)
             for c, p in problems]
    shards = shard_problems(costs, world)
    mine = shards[rank]
    used = sorted({problems[i][0] for i in mine})
    local_of = {c: k for k, c in enumerate(used)}
    pset = None
    summaries = np.zeros(0, dtype=np.float64)
    measured = np.zeros(2 * len(penalties), dtype=np.float64)  # per penalty rank: sum, count
    t0 = time.time()
    if mine:
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
                         launches=pset.solve_stats[0], create_s=t1 - t0, solve_s=t2 - t1)
        res = [pset.result(k) for k in range(len(mine))]
        summaries = np.array([[r.n_segments, r.n_equality_constraints, r.max_intervals,
                               r.total_intervals, r.best_cost] for r in res],
                             dtype=np.float64).reshape(-1)
        for k, i in enumerate(mine):
            cps = pset.cycles_per_step(k)
            if cps > 0:
                measured[2 * prank[problems[i][1]]] += cps
                measured[2 * prank[problems[i][1]] + 1] += 1.0
    t3 = time.time()
    gathered = gather_problem_set(pset, dist, device, extra=np.concatenate([summaries, measured]))
    t4 = time.time()
    if pset is not None:
        pset.close()
    if stats is not None:
        stats.update(tables_s=t4 - t3, close_s=time.time() - t4)
    if dist is not None and world > 1:
        # every rank learns what the ranks measured: the next call's dealing uses it
        import torch
        dev = _torch_device(dist, device)
        m = torch.from_numpy(measured).to(dev)
        dist.all_reduce(m)
        measured = m.cpu().numpy()
    if stats is not None:
        cnt = measured[1::2]
        if (cnt > 0).all():
            stats["cycles_per_step"] = (measured[0::2] / cnt).tolist()
    if rank != 0:
        return None
    out = {}
    for r in range(world):
        tables, extra = gathered[r]
        summ = extra[:5 * len(shards[r])].reshape(-1, 5)
        for k, i in enumerate(shards[r]):
            start, mean = tables[k]
            out[problems[i]] = {"seg_start": start, "seg_mean": mean, "summary": summ[k]}
    return out
