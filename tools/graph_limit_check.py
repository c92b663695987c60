#!/usr/bin/env python3
"""One-off check near the row limit of the joins (2^25 rows): the deletion-variant join and the q-gram join on ROWS random
distinct 16-mers (default 30,000,000; more than 2^31 index entries), thr 2 - the same edge list is required.  Builder tool."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from badger_amd import _native  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    r = torch.randint(0, 1 << 32, (int(n * 1.02),), device=dev, dtype=torch.int64, generator=g)
    r = torch.unique(r)[:n]                                    # ascending, distinct
    assert len(r) == n
    d_ranks = (r & 0xFFFFFFFF).to(torch.int32) if False else torch.where(r >= (1 << 31), r - (1 << 32), r).to(torch.int32)
    del r
    ctx = _native.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    cap = 16 * n
    d_edges = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
    d_n = torch.zeros(1, dtype=torch.int64, device=dev)
    keys = {}
    for algo in ((5, 3) if n < (1 << 25) else (5, 55)):            # (beyond the q-gram join's 2^25 rows: the join in other rounds against itself)
        if algo == 55:
            os.environ["BADGER_AMD_D2_ROUNDS"] = "5"
            algo = 5
            tag = 55
        else:
            tag = algo
        ctx.graph_set_algo(algo)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.graph_edges_rows_dev(d_ranks, n, 0, n, 2, 4, d_edges, cap, d_n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ne = int(d_n[0])
        if ne > cap:
            raise SystemExit("edge capacity too small: %d" % ne)
        e = d_edges[:ne].to(torch.int64) & 0xFFFFFFFF
        # (a, b) as one signed 64-bit key (a - 2^31 in the high half), the distance beside it
        kk, order = torch.sort(((e[:, 0] - (1 << 31)) << 32) | e[:, 1])
        keys[tag] = kk
        dists = globals().setdefault("_dists", {})
        dists[tag] = e[:, 2][order]
        print(json.dumps({"rows": n, "algo": algo, "rounds": os.environ.get("BADGER_AMD_D2_ROUNDS", "auto") if algo == 5 else None,
                          "seconds": round(dt, 3), "edges": ne}), flush=True)
        if tag == 5:                                                 # no repeats; a sample of the edges by the oracle's S and dmin
            k = keys[5]
            assert bool((k[1:] != k[:-1]).all()), "an edge is listed twice"
            from oracle import pyoracle as orc
            idx = torch.randint(0, ne, (2000,), device=dev)
            for a, b, d in e[idx].cpu().tolist():
                assert orc.qgram_S(a, b) >= 4 and orc.dmin3(a, b) == d and d <= 2, (a, b, d)
        del e
    other = [t for t in keys if t != 5][0]
    same = keys[other].shape == keys[5].shape and bool((keys[other] == keys[5]).all()) and bool((_dists[other] == _dists[5]).all())
    print(json.dumps({"rows": n, "same_edges": same, "against": "q-gram join" if other == 3 else "the join in 5 rounds"}))
    if not same:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
