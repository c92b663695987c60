"""The N>1 path on CPU: world_size 2 over gloo.  Partitioning, the rank-ordered gather and the
max-over-ranks clock.  (The per-shard arithmetic is the oracle here -- no GPU in this tier.)"""
import os
import socket
import time

import numpy as np
import torch.multiprocessing as mp

from badger_amd import dist as bdist, synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from oracle import pyoracle as orc
    bdist.init(backend="gloo")
    wl = synth.make_whitelist(500)
    bases, off = synth.make_reads(301, wl, seed=5)
    b, o = bases.numpy(), off.numpy().astype(np.uint64)
    recs = bdist.extract_sharded(lambda bb, oo, u: orc.extract_batch(bb, oo, u, threads=1), b, o, 12)
    t = bdist.timed(lambda: time.sleep(0.05 * (rank + 1)), 2)
    # graph rows: every rank holds all ranks and emits the edges whose smaller rank lies in its row block
    rng = np.random.default_rng(9)
    base = rng.integers(0, 1 << 32, 40, dtype=np.uint64)
    near = base[rng.integers(0, 40, 400)] ^ (rng.integers(1, 4, 400).astype(np.uint64) << (2 * rng.integers(0, 16, 400).astype(np.uint64)))
    ranks = np.unique(np.concatenate([base, near]).astype(np.uint32))

    def rows_fn(rs, lo, hi, thr, T):
        e = orc.graph_edges(rs, thr, T)
        keep = (e["a"] >= rs[lo]) & (e["a"] <= rs[hi - 1]) if hi > lo else np.zeros(len(e), bool)
        return e[keep]

    edges = bdist.graph_edges_sharded(rows_fn, ranks, 1, 5, balance="pairs")

    def part_fn(rs, part, nparts, thr, T):                # (shares cut by something other than rows, as the deletion-variant join cuts them)
        e = orc.graph_edges(rs, thr, T)
        return e[(e["a"].astype(np.uint64) * np.uint64(31) + e["b"].astype(np.uint64)) % np.uint64(nparts) == np.uint64(part)]

    by_parts = bdist.graph_parts_sharded(part_fn, ranks, 1, 5)
    if rank == 0:
        whole = orc.extract_batch(b, o, 12, threads=1)
        all_edges = orc.graph_edges(ranks, 1, 5)
        all_edges = all_edges[np.lexsort((all_edges["b"], all_edges["a"]))]
        graph_ok = len(edges) == len(all_edges) and bool((edges == all_edges).all()) and len(edges) > 50
        graph_ok = graph_ok and len(by_parts) == len(all_edges) and bool((by_parts == all_edges).all())
        q.put((bool((recs == whole).all()) and graph_ok, len(recs), t))
    bdist.barrier()


def test_partition_covers_everything():
    for n in (0, 1, 7, 8, 100, 1000003):
        for w in (1, 2, 3, 8):
            parts = [bdist.partition(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1


def test_graph_row_blocks_cover_everything():
    for n in (0, 1, 2, 255, 256, 1000, 500000):
        for w in (1, 2, 3, 8):
            for balance in ("rows", "pairs"):
                blocks = bdist.graph_row_blocks(n, w, balance)
                assert blocks[0][0] == 0 and blocks[-1][1] == n and len(blocks) == w
                assert all(blocks[i][1] == blocks[i + 1][0] and blocks[i][0] <= blocks[i][1] for i in range(w - 1))
    # equal pairs: the row blocks grow towards the end, the pair counts stay within a few percent
    n, w = 500000, 8
    pairs = [sum(n - 1 - i for i in (lo, hi - 1)) * (hi - lo) / 2 for lo, hi in bdist.graph_row_blocks(n, w, "pairs")]
    assert max(pairs) / min(pairs) < 1.02


def test_world2_gloo_sharded_extract_and_clock():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, n, t = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same and n == 301
    assert t >= 0.19          # max over ranks: rank 1 sleeps 2 x 0.1 s


def test_pair_balanced_blocks_balance_the_joins_work():
    """thr >= 2: the q-gram join's work for row i is the number of candidate entries behind it in its 11 buckets
    (index.py:77-93 walks exactly those).  Blocks of equal pair counts must give every GPU the same share of that within
    5 %; equal rows would give the first of 8 GPUs almost twice its share."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n = 60000
    ranks = bench.observed_barcodes(n, synth.make_whitelist(20000))
    assert bdist.graph_balance(1) == "rows" and bdist.graph_balance(2) == "pairs" and bdist.graph_balance(3) == "pairs"
    # entries behind row i: for each of its 11 six-mers, the rows > i holding that six-mer (with multiplicity)
    r = ranks.astype(np.uint64)
    keys = np.stack([(r >> np.uint64(2 * p)) & np.uint64(0xFFF) for p in range(11)], axis=1).astype(np.int64)       # [n, 11]
    flat = keys.ravel()
    order = np.argsort(flat, kind="stable")                       # sorted by six-mer, rows ascending inside a bucket
    pos = np.empty(len(flat), dtype=np.int64)
    pos[order] = np.arange(len(flat))
    ends = np.searchsorted(flat[order], np.arange(4097))          # bucket k ends at ends[k + 1]
    behind = (ends[flat + 1] - pos - 1).reshape(n, 11).sum(axis=1)
    total = behind.sum()
    for world in (2, 4, 8):
        share = [behind[lo:hi].sum() / total for lo, hi in bdist.graph_row_blocks(n, world, "pairs")]
        assert max(share) * world < 1.05 and min(share) * world > 0.95, (world, share)
        rows = [behind[lo:hi].sum() / total for lo, hi in bdist.graph_row_blocks(n, world, "rows")]
        assert max(rows) * world > 1.4


def test_hashed_group_shares_balance_the_deletion_variant_join():
    """thr 2 on several GPUs: bdg_graph_edges_part_dev gives a rank the 14-mer groups whose multiplicative hash falls into
    its part (graph_kernels.hip d2_part).  Restated here on bench.py's barcodes: index entries and meeting pairs per part
    stay within 3 % of an even share for 2, 4 and 8 parts."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n = 60000
    r = bench.observed_barcodes(n, synth.make_whitelist(20000)).astype(np.uint64)
    one = np.uint64(1)
    keys = []
    for p in range(16):
        for q in range(p + 1, 16):
            lo = r & ((one << np.uint64(2 * p)) - one)
            mid = (r >> np.uint64(2 * p + 2)) & ((one << np.uint64(2 * (q - p - 1))) - one)
            hi = (r >> np.uint64(2 * q + 2)) if q < 15 else np.zeros_like(r)
            keys.append(((lo | (mid << np.uint64(2 * p)) | (hi << np.uint64(2 * q - 2))) << np.uint64(32)) | np.arange(n, dtype=np.uint64))
    k = np.unique(np.concatenate(keys))                               # a row holds a 14-mer once
    assert 60 * n < len(k) < 90 * n
    groups, size = np.unique((k >> np.uint64(32)).astype(np.uint32), return_counts=True)
    for world in (2, 4, 8):
        part = (((groups.astype(np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) * np.uint64(world)) >> np.uint64(32)
        entries = np.bincount(part.astype(np.int64), weights=size, minlength=world)
        pairs = np.bincount(part.astype(np.int64), weights=size * (size - 1) / 2, minlength=world)
        assert entries.max() / entries.sum() * world < 1.03 and entries.min() / entries.sum() * world > 0.97
        assert pairs.max() / pairs.sum() * world < 1.03 and pairs.min() / pairs.sum() * world > 0.97


def test_timed_settles_waiting_work_inside_the_clock():
    """bench.py's batch pipelining leaves a step's whitelist match waiting until the next step begins: the clock must not
    start with one waiting from the warm-up, and must not stop before the last step's has been queued and finished."""
    from badger_amd import dist as bdist
    log = []
    t = bdist.timed(lambda: log.append("step"), 3, None, settle=lambda: log.append("settle"))
    assert log == ["settle", "step", "step", "step", "settle"] and t >= 0.0
    log.clear()
    bdist.timed(lambda: log.append("step"), 2, None)
    assert log == ["step", "step"]
