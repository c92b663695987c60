#!/usr/bin/env python3
"""Randomised differential test of K3's default paths against the oracle's whole edge lists: barcode sets from several
generators (uniform, cells with substitution / insertion / deletion clouds of different mixes, low-complexity repeats,
sets dense around few cells), sizes between 60 K and 700 K rows, thr 1 and 2 (one-deletion / deletion-variant joins above their
row limits, probes below).  One line per case; exit code 1 on any mismatch.  Builder tool (minutes of oracle time).
usage: join_fuzz.py [cases] [seed]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from badger_amd import _native  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402


def edit_cloud(rng, cells, m, p_sub, p_ins, p_del):
    """m observed barcodes: a cell with per-letter substitutions / insertions / deletions, cut or padded to 16 letters"""
    r = cells[rng.integers(0, len(cells), m)].astype(np.uint64)
    for _ in range(3):
        hit = rng.random(m) < p_sub * 16 / 3
        r = np.where(hit, r ^ (rng.integers(1, 4, m).astype(np.uint64) << (2 * rng.integers(0, 16, m).astype(np.uint64))), r)
    for _ in range(2):
        hit = rng.random(m) < p_del * 16 / 2
        pos = rng.integers(0, 16, m).astype(np.uint64)
        low = (np.uint64(1) << (2 * pos)) - np.uint64(1)
        d = (r & low) | ((r >> np.uint64(2)) & ~low & np.uint64(0xFFFFFFFF)) | (rng.integers(0, 4, m).astype(np.uint64) << np.uint64(30))
        r = np.where(hit, d, r)
        hit = rng.random(m) < p_ins * 16 / 2
        pos = rng.integers(0, 16, m).astype(np.uint64)
        low = (np.uint64(1) << (2 * pos)) - np.uint64(1)
        i = (r & low) | (rng.integers(0, 4, m).astype(np.uint64) << (2 * pos)) | ((r & ~low) << np.uint64(2))
        r = np.where(hit, i & np.uint64(0xFFFFFFFF), r)
    return np.unique((r & np.uint64(0xFFFFFFFF)).astype(np.uint32))


def low_complexity(rng, m):
    out = np.zeros(m, dtype=np.uint64)
    for k in range(m):
        unit = rng.integers(0, 4, int(rng.integers(1, 5)))
        s = np.resize(unit, 16).copy()
        for _ in range(int(rng.integers(0, 4))):
            s[int(rng.integers(0, 16))] = rng.integers(0, 4)
        out[k] = sum(int(c) << (2 * i) for i, c in enumerate(s))
    return np.unique(out.astype(np.uint32))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda", 0)
    ctx = _native.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    cores = len(os.sched_getaffinity(0))
    bad = 0
    for case in range(cases):
        rng = np.random.default_rng(seed * 1000 + case)
        kind = case % 4
        n_want = int(rng.integers(60000, 700000))
        if kind == 0:
            ranks = np.unique(rng.integers(0, 1 << 32, n_want, dtype=np.uint64).astype(np.uint32))
            what = "uniform"
        elif kind == 1:
            cells = rng.integers(0, 1 << 32, int(rng.integers(2000, 200000)), dtype=np.uint64)
            mix = (float(rng.random() * 0.06), float(rng.random() * 0.04), float(rng.random() * 0.04))
            ranks = edit_cloud(rng, cells, 2 * n_want, *mix)[:n_want]
            what = "clouds of %d cells, sub/ins/del %.3f/%.3f/%.3f" % ((len(cells),) + mix)
        elif kind == 2:
            cells = rng.integers(0, 1 << 32, int(rng.integers(20, 400)), dtype=np.uint64)
            ranks = edit_cloud(rng, cells, 3 * n_want, 0.08, 0.03, 0.03)[:n_want]
            what = "dense around %d cells" % len(cells)
        else:
            ranks = np.unique(np.concatenate([low_complexity(rng, 30000), edit_cloud(rng, low_complexity(rng, 3000).astype(np.uint64), n_want, 0.04, 0.02, 0.02)]))[:n_want]
            what = "low complexity"
        ranks = np.sort(ranks)
        row = {"case": case, "what": what, "rows": int(len(ranks))}
        for thr in (1, 2):
            T = orc.qgram_threshold(thr)
            e = ctx.graph_edges(ranks, thr, T)
            w, _, _ = orc.graph_edges_sampled(ranks, thr, 1, T, threads=cores, cap=len(e) + 1)
            same = len(e) == len(w) and (len(e) == 0 or bool((e == w).all()))
            row["thr%d" % thr] = {"edges": int(len(w)), "same": same}
            bad += 0 if same else 1
        print(json.dumps(row), flush=True)
    print("JOIN FUZZ %s" % ("OK" if bad == 0 else "FAILED (%d)" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
