#!/usr/bin/env python3
"""Large randomized parity run on the GPU box (not part of the test suite: minutes of oracle time).
For several seeds / error profiles: extraction of N synthetic reads on the GPU vs the C oracle on all host cores,
then nearest16 of the extracted barcodes vs the oracle on a sample, and graph edges (thr 1, 2, 3) of a slice and of all of the
distinct barcodes vs the oracle.  Prints one line per case; exit code 1 on any mismatch.

    python tools/stress_parity.py [--reads 1000000] [--cases 6] [--n-rate 0.001]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from badger_amd import _native, synth  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

PROFILES = [dict(p_sub=0.03, p_ins=0.02, p_del=0.03), dict(p_sub=0.0, p_ins=0.0, p_del=0.0),
            dict(p_sub=0.08, p_ins=0.04, p_del=0.04), dict(p_sub=0.01, p_ins=0.06, p_del=0.01),
            dict(p_sub=0.01, p_ins=0.01, p_del=0.08), dict(p_sub=0.15, p_ins=0.05, p_del=0.05)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1000000)
    ap.add_argument("--cases", type=int, default=len(PROFILES))
    ap.add_argument("--n-rate", type=float, default=0.0, help="probability of replacing a base by N")
    ap.add_argument("--seed-base", type=int, default=100, help="case k uses seed seed-base + k")
    ap.add_argument("--no-polya-rule", action="store_true", help="the reference's second strand rule (find_barcode_umi_no_polya)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cores = len(os.sched_getaffinity(0))
    wl = synth.make_whitelist(737280)
    ctx = _native.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.whitelist_load(wl)
    rule = 1 if args.no_polya_rule else 0
    ctx.extract_set_strand_rule(rule)
    bad = 0
    for case in range(args.cases):
        prof = PROFILES[case % len(PROFILES)]
        n = args.reads
        bases, off = synth.make_reads(n, wl, seed=args.seed_base + case, device=dev, **prof)
        total = int(off[-1])
        if args.n_rate > 0:                                   # sprinkle N: exercises the exact per-byte path of the scan at scale
            g = torch.Generator(device=dev)
            g.manual_seed(1000 + case)
            bases[torch.rand(total, generator=g, device=dev) < args.n_rate] = ord("N")
        bases = torch.cat([bases, torch.zeros(64, dtype=torch.uint8, device=dev)])
        recs = torch.zeros((n, 8), dtype=torch.int32, device=dev)
        for _ in range(2):
            ctx.extract_batch_dev(bases, off.contiguous(), n, total, 12, recs)
            rc, _, _ = ctx.extract_status()
            if rc != _native.E_CAPACITY:
                break
        assert rc == 0
        got = recs.cpu().numpy().view(_native.REC_DTYPE).reshape(-1)
        t0 = time.time()
        want = orc.extract_batch(bases[:total].cpu().numpy(), off.cpu().numpy().astype(np.uint64), 12, threads=cores, rule=rule)
        t_orc = time.time() - t0
        mism = int((got != want).sum())
        # nearest16 on a sample of the extracted barcodes
        ok = (got["flags"] & 2) != 0
        q = got["bc_rank"][ok][:200000]
        d_q = torch.from_numpy(q.view(np.int32)).to(dev)
        bi = torch.zeros(len(q), dtype=torch.int32, device=dev)
        be = torch.zeros(len(q), dtype=torch.uint8, device=dev)
        bt = torch.zeros(len(q), dtype=torch.int16, device=dev)
        ctx.nearest16_dev(d_q, len(q), 2, bi, be, bt)
        torch.cuda.synchronize()
        # every call against the oracle's probe form (pinned to its exhaustive scan in the test suite), 96 against the scan itself
        wi, we, wt = orc.nearest16(q, wl, 2, threads=cores, probe=True)
        nm = int((bi.cpu().numpy().view(np.uint32) != wi).sum() + (be.cpu().numpy() != we).sum()
                 + (bt.cpu().numpy().view(np.uint16) != wt).sum())
        sel = np.random.default_rng(case).integers(0, len(q), 96)
        xi, xe, xt = orc.nearest16(q[sel], wl, 2, threads=cores)
        nm += int((wi[sel] != xi).sum() + (we[sel] != xe).sum() + (wt[sel] != xt).sum())
        # graph edges: of a slice of 60,000 of the distinct barcodes (thr 1: neighbourhood probes, thr 2: deletion-variant join,
        # thr 3: q-gram join) and of ALL of them (some 600,000; thr 1: one-deletion join, thr 2: deletion-variant join) against
        # the oracle's whole lists
        every = np.unique(got["bc_rank"][ok])
        gm, lists = 0, 0
        for ranks, thrs in ((every[:60000], (1, 2, 3)), (every, (1, 2))):
            for thr in thrs:
                T = orc.qgram_threshold(thr)
                e = ctx.graph_edges(ranks, thr, T)
                w, _, _ = orc.graph_edges_sampled(ranks, thr, 1, T, threads=cores, cap=len(e) + 1)
                gm += int(len(e) != len(w) or (len(e) and (e != w).any()))
                lists += 1
        print("case %d %s: %d reads, valid %.3f, extract mismatches %d (oracle %.1f s), nearest mismatches %d / 200000, graph mismatching lists %d / %d"
              % (case, prof, n, float(got["valid"].mean()), mism, t_orc, nm, gm, lists), flush=True)
        bad += mism + nm + gm
    print("STRESS PARITY %s" % ("OK" if bad == 0 else "FAILED (%d)" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
