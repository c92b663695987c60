#!/usr/bin/env python3
"""bdg_distinct_dev on made-up records of different shapes (uniform ranks, the bench's dense reads, hot barcodes), per kernel:
where the time of the distinct count goes.  One JSON line per case."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from badger_amd import _native, synth  # noqa: E402


def records_of(ranks):
    recs = np.zeros(len(ranks), dtype=_native.REC_DTYPE)
    recs["bc_rank"] = ranks
    recs["valid"] = 1
    recs["flags"] = _native.FLAG_RANK_OK | _native.FLAG_BC16
    return recs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    dev = torch.device("cuda", 0)
    ctx = _native.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(1)
    wl = synth.make_whitelist(737280)
    cells = wl[rng.permutation(len(wl))[:5000]]
    sizes = rng.lognormal(0.0, 1.0, 5000)
    pick = rng.choice(5000, n, p=sizes / sizes.sum())
    dense = cells[pick].astype(np.uint64)
    for _ in range(2):
        hit = rng.random(n) < 0.35
        dense = np.where(hit, dense ^ (rng.integers(1, 4, n).astype(np.uint64) << (2 * rng.integers(0, 16, n).astype(np.uint64))), dense)
    cases = {"uniform": rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32),
             "dense_cells": dense.astype(np.uint32),
             "exact_copies": cells[pick].astype(np.uint32)}
    for name, ranks in cases.items():
        recs = records_of(ranks)
        d_recs = torch.from_numpy(recs.view(np.int32).reshape(-1, 8).copy()).to(dev)
        uq, ct, fi = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3))
        dn = torch.zeros(2, dtype=torch.int32, device=dev)
        for _ in range(3):
            ctx.distinct_dev(d_recs, n, uq, ct, fi, dn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.distinct_dev(d_recs, n, uq, ct, fi, dn)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(5):
            ctx.distinct_dev(d_recs, n, uq, ct, fi, dn)
        torch.cuda.synchronize()
        kernels = {k: round(v[1] / max(1, v[0]), 4) for k, v in ctx.profile_read().items() if v[0]}
        ctx.profile(False)
        wu, wc = np.unique(ranks, return_counts=True)
        nu = int(dn[0])
        ok = nu == len(wu) and bool((uq[:nu].cpu().numpy().view(np.uint32) == wu).all()) and bool((ct[:nu].cpu().numpy() == wc).all())
        top = np.bincount(ranks >> 22, minlength=1024)
        print(json.dumps({"case": name, "records": n, "distinct": nu, "ms": round(ms, 4), "kernels_ms": kernels, "ok": ok,
                          "largest_multiplicity": int(wc.max()), "largest_top10_bucket": int(top.max())}))


if __name__ == "__main__":
    main()
