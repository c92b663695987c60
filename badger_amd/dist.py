"""One process per GPU.  Reads partition independently, so the data path needs no
collective: each rank takes a contiguous range of reads, the whitelist is replicated, and
results are concatenated in rank order.  Graph rows shard the same way (every rank holds the whole
sorted rank array and emits its share of the edges: those whose smaller rank lies in its block of rows, or - the
deletion-variant join - those reported from its share of the 14-mer groups; bdg_graph_edges_part_dev cuts the shares).  torch.distributed (RCCL on the GPU box, gloo on CPU)
is used only for the barrier, the max-over-ranks clock and the final gather of records."""
import os
import time

import numpy as np
import torch
import torch.distributed as dist


def partition(n, world, rank):
    """Contiguous read-index range [lo, hi) of `rank` (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def barrier(device=None):
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def all_max(value, device=None):
    """max over ranks of a python float"""
    if not dist.is_initialized():
        return float(value)
    on_gpu = device is not None and device.type == "cuda" and dist.get_backend() == "nccl"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def all_sum(value, device=None):
    """sum over ranks of a python number"""
    if not dist.is_initialized():
        return float(value)
    on_gpu = device is not None and device.type == "cuda" and dist.get_backend() == "nccl"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0])


def timed(step, steps, device=None, settle=None):
    """K calls of step() bracketed by barrier + synchronize; returns max-over-ranks seconds.  settle: called before the
    clock starts and again behind the last step, inside the timed region - for work a step may leave waiting (the
    library's batch pipelining queues a batch's whitelist match only when the next batch begins)."""
    if settle is not None:
        settle()
    barrier(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if settle is not None:
        settle()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    barrier(device)
    return all_max(elapsed, device)


def extract_sharded(extract_fn, bases, off, umi_len=12):
    """Run extract_fn(bases, off_slice, umi_len) -> records on this rank's partition of the reads and
    return, on rank 0, all records in read order (None elsewhere)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    n = len(off) - 1
    lo, hi = partition(n, world, rank)
    mine = extract_fn(bases, off[lo:hi + 1], umi_len)
    if world == 1:
        return mine
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine.tobytes(), gathered, dst=0)
    if rank != 0:
        return None
    return np.concatenate([np.frombuffer(b, dtype=mine.dtype) for b in gathered])


def graph_row_blocks(n, world, balance="rows"):
    """Row blocks [lo, hi) of the sorted rank array, one per rank.  balance="rows": equal rows
    (neighbourhood probes: the work per row is constant); "pairs": equal numbers of (i, j > i) pairs, i.e.
    boundaries at n * (1 - sqrt(1 - g / world)) (all-pairs sweep: row i meets n - 1 - i partners)."""
    if balance == "rows":
        return [partition(n, world, r) for r in range(world)]
    cuts = [int(round(n * (1.0 - (1.0 - g / world) ** 0.5))) for g in range(world)] + [n]
    cuts = [min(max(c, 0), n) for c in cuts]
    for g in range(1, world + 1):
        cuts[g] = max(cuts[g], cuts[g - 1])
    return [(cuts[g], cuts[g + 1]) for g in range(world)]


def graph_balance(thr):
    """how graph ROWS are cut into per-GPU blocks by the paths that work row by row (bdg_graph_edges_rows_dev): the
    neighbourhood probes of thr 1 do the same 176 look-ups for every row ("rows"); the q-gram join and the sweep (thr >= 2 when
    forced, thr >= 3 by default) let row i walk what lies BEHIND it, i.e. work ~ n - i ("pairs").  The deletion-variant joins
    that serve thr 1 / 2 by default are cut by 14- / 15-mer groups instead: bdg_graph_edges_part_dev does either."""
    return "rows" if thr <= 1 else "pairs"


def graph_parts_sharded(part_fn, ranks_sorted, thr, qgram_T):
    """Run part_fn(ranks_sorted, part, nparts, thr, qgram_T) -> structured edge array (bdg_graph_edges_part_dev: one of
    nparts disjoint shares of the edge list, cut by the library) with part = this rank, and return, on rank 0, all edges
    sorted by (a, b) (None elsewhere).  No collective on the data path: the gather of the edge lists is the only exchange."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mine = part_fn(ranks_sorted, rank, world, thr, qgram_T)
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine.tobytes(), gathered, dst=0)
        if rank != 0:
            return None
        mine = np.concatenate([np.frombuffer(b, dtype=mine.dtype) for b in gathered])
    return mine[np.lexsort((mine["b"], mine["a"]))]


def graph_edges_sharded(edges_fn, ranks_sorted, thr, qgram_T, balance="rows"):
    """Run edges_fn(ranks_sorted, row_lo, row_hi, thr, qgram_T) -> structured edge array on this rank's row block
    and return, on rank 0, all edges sorted by (a, b) (None elsewhere).  No collective on the data path: the gather
    of the edge lists is the only exchange."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = graph_row_blocks(len(ranks_sorted), world, balance)[rank]
    mine = edges_fn(ranks_sorted, lo, hi, thr, qgram_T)
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine.tobytes(), gathered, dst=0)
        if rank != 0:
            return None
        mine = np.concatenate([np.frombuffer(b, dtype=mine.dtype) for b in gathered])
    return mine[np.lexsort((mine["b"], mine["a"]))]
