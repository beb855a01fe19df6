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


def gather_segment_tables(tables, dist=None, device=None):
    """Gather every rank's segment tables on rank 0.

    dist is torch.distributed (already initialised) or None for a single process.  Returns on
    rank 0 a list over ranks of lists of (seg_start, seg_mean); on other ranks None.
    Protocol: all_gather of the per-rank row totals, then one padded gather per array."""
    if dist is None or dist.get_world_size() == 1:
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
    max_rows = max(m[0] for m in metas)
    max_seg = max(m[1] for m in metas)

    def padded(a, n, dtype):
        t = torch.zeros(max(n, 1), dtype=dtype, device=dev)
        if len(a):
            t[:len(a)] = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        return t

    payload = [padded(rows, max_rows, torch.int64), padded(start, max_seg, torch.int32),
               padded(mean, max_seg, torch.float64)]
    gathered = []
    for t in payload:
        bufs = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, bufs, dst=0)
        gathered.append(bufs)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        n_rows, n_seg = metas[r]
        out.append(unpack_tables(gathered[0][r][:n_rows].cpu().numpy(),
                                 gathered[1][r][:n_seg].cpu().numpy(),
                                 gathered[2][r][:n_seg].cpu().numpy()))
    return out
