#!/usr/bin/env python3
"""K3 at thr 2 over a range of row counts: the q-gram join (algo 3) against the deletion-variant join (algo 5), same
edge lists required.  Barcodes: every whitelist entry a cell, observed with substitutions and a deletion (so the number
of variants per cell stays near what a run of that many reads shows).  One JSON object per line.  Builder tool.
usage: graph_sizes.py [rows ...]   (default 2000 ... 16000000)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from badger_amd import _native, synth  # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [2000, 10000, 50000, 100000, 250000, 500000, 1000000, 2000000, 4000000, 8000000, 16000000]
    limit_qjoin = float(os.environ.get("GRAPH_SIZES_QJOIN_MAX", "4000000"))
    dev = torch.device("cuda", 0)
    wl = synth.make_whitelist(737280)
    ctx = _native.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    thr, T = 2, 4
    pair = (3, 5)                                    # q-gram join against the deletion-variant join
    if os.environ.get("GRAPH_SIZES_THR") == "1":     # thr 1: the neighbourhood probes against the one-deletion join
        thr, T, pair = 1, 5, (2, 6)
    for n in sizes:
        ranks = bench.observed_barcodes(n, wl, seed=3, n_cells=len(wl))
        d_ranks = torch.from_numpy(ranks.view(np.int32)).to(dev)
        cap = 24 * n
        d_edges = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
        d_n = torch.zeros(1, dtype=torch.int64, device=dev)
        out = {"rows": n}
        lists = {}
        for algo in pair:
            if algo == pair[0] and n > limit_qjoin:
                continue
            ctx.graph_set_algo(algo)
            best = None
            for rep in range(5 if n <= 1000000 else 3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.graph_edges_rows_dev(d_ranks, n, 0, n, thr, T, d_edges, cap, d_n)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            ne = int(d_n[0])
            if ne > cap:
                raise SystemExit("edge capacity too small at %d rows: %d" % (n, ne))
            e = d_edges[:ne].to(torch.int64) & 0xFFFFFFFF
            key = ((e[:, 0] - (1 << 31)) << 32) | e[:, 1]                 # (a, b) as one signed 64-bit key; the distance follows from the pair
            lists[algo] = torch.sort(key).values
            out["algo%d_ms" % algo] = round(best * 1e3, 2)
            out["edges"] = ne
            ctx.profile(True)
            ctx.profile_reset()
            ctx.graph_edges_rows_dev(d_ranks, n, 0, n, thr, T, d_edges, cap, d_n)
            torch.cuda.synchronize()
            out["algo%d_kernels_ms" % algo] = {k: round(v[1] / max(1, v[0]), 3) for k, v in ctx.profile_read().items() if v[0]}
            ctx.profile(False)
        if len(lists) == 2:
            out["same_edges"] = bool(lists[pair[0]].shape == lists[pair[1]].shape and bool((lists[pair[0]] == lists[pair[1]]).all()))
        ctx.graph_set_algo(0)
        print(json.dumps(out), flush=True)
        del d_edges, d_ranks, lists


if __name__ == "__main__":
    main()
