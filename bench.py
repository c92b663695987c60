#!/usr/bin/env python3
"""Headline benchmark: barcode calls/s against the 737K-entry whitelist (BASELINE.json).

One step = one pass of the hot path over one device-resident batch of synthetic ONT reads:
  K1 extract (k_scan_reads, k_sw_clusters x3, k_strict_filter, k_finalize_reads)  -> 32-byte record per read
  K2 nearest16 (probe path, max_ed 2) of every extracted barcode against the whitelist
A "call" is one read taken through both.  Reads are sharded per GPU (weak scaling, no
collective on the data path); the process group is only used for the barrier and the
max-over-ranks clock.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--whitelist W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from badger_amd import _native, synth  # noqa: E402
from badger_amd import dist as bdist  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(bases_dev, off_dev, wl, n_extract, n_nearest):
    """The CPU oracle (a port of the reference algorithm) timed on this box's host cores,
    on a bounded sample of the same workload.  Only the checker is timed here; nothing the
    product path produces depends on it."""
    from oracle import pyoracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_extract = min(n_extract, off_dev.numel() - 1)
    end = int(off_dev[n_extract])
    b = bases_dev[:end].cpu().numpy()
    o = off_dev[:n_extract + 1].cpu().numpy().astype(np.uint64)
    orc.extract_batch(b[:int(o[64])], o[:65], 12, threads=cores)          # warm
    t0 = time.perf_counter()
    recs = orc.extract_batch(b, o, 12, threads=cores)
    t_ext = (time.perf_counter() - t0) / n_extract
    q = recs["bc_rank"][(recs["flags"] & 2) != 0][:n_nearest]
    t0 = time.perf_counter()
    orc.nearest16(q, wl, 2, threads=cores)
    t_near = (time.perf_counter() - t0) / max(1, len(q))
    return {"value": 1.0 / (t_ext + t_near), "unit": "calls/s", "cores": cores, "kind": "port",
            "sample": "%d reads through oracle extract_batch (%.0f reads/s) + %d barcodes through the exhaustive "
                      "Levenshtein scan of the %d-entry whitelist the reference's postprocessing loop does "
                      "(%.1f calls/s); OpenMP over all %d cores" % (n_extract, 1.0 / t_ext, len(q), len(wl), 1.0 / t_near, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1000000, help="reads per GPU (weak scaling)")
    ap.add_argument("--whitelist", type=int, default=737280)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, local_rank, world = bdist.env_rank()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("BADGER_DIST_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 path with ranks sharing a GPU
    if local_rank >= ndev and backend == "nccl":
        raise SystemExit("rank %d has no GPU (%d visible)" % (local_rank, ndev))
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    bdist.init(backend=backend, device=dev)           # RCCL; only the barrier and the clock use it

    # ---- inputs, resident in HBM before the clock starts
    wl = synth.make_whitelist(args.whitelist)
    bases, off = synth.make_reads(args.reads, wl, seed=1 + rank, device=dev)
    n = args.reads
    total_bytes = int(off[-1])
    pad = torch.zeros(64, dtype=torch.uint8, device=dev)
    bases = torch.cat([bases, pad])[:total_bytes + 64]
    off_u = off.to(torch.int64).contiguous()          # same bits as uint64
    recs = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    best_idx = torch.zeros(n, dtype=torch.int32, device=dev)
    best_ed = torch.zeros(n, dtype=torch.uint8, device=dev)
    n_ties = torch.zeros(n, dtype=torch.int16, device=dev)

    ctx = _native.Context(local_dev)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    ctx.whitelist_load(wl)

    def step():
        ctx.extract_batch_dev(bases, off_u, n, total_bytes, 12, recs)
        ctx.nearest16_recs_dev(recs, n, 2, best_idx, best_ed, n_ties)      # every record's barcode against the whitelist

    for _ in range(max(1, args.warmup)):
        step()
    rc, bad, nwin = ctx.extract_status()
    if rc == _native.E_CAPACITY:                       # window queue grown: one more warm-up pass
        step()
        rc, bad, nwin = ctx.extract_status()
    if rc != 0:
        raise SystemExit("extract failed: rc=%d bad_read=%d" % (rc, bad))
    stats = ctx.extract_counters()
    ctx.profile(True)
    ctx.profile_reset()

    elapsed = bdist.timed(step, args.steps, dev)    # barrier + sync, K steps, sync + barrier, max over ranks
    prof = ctx.profile_read()
    ctx.profile(False)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n / (elapsed / args.steps)
        # roofline of the dominant kernel: algorithmic bytes of the unit it serves / its own launch time
        k1 = ("k_scan_reads", "k_sw_clusters", "k_sw_requeued", "k_sw_survivors", "k_strict_filter", "k_finalize_reads")
        per_launch_ms = {k: v[1] / max(1, v[0]) for k, v in prof.items()}
        dom = max(per_launch_ms, key=per_launch_ms.get)
        k1_bytes = total_bytes + 40 * n                # sum(L_i) + 8 (offset) + 32 (record) per read
        k2_bytes = 11 * n + 4 * len(wl)                # 4 (query) + 7 (idx, ed, ties) per call + whitelist once
        alg = k1_bytes if dom in k1 else k2_bytes
        achieved = alg / (per_launch_ms[dom] * 1e-3) / 1e9
        # from the committed rocprofv3 PMC passes (tools/pmc_profile.sh -> tools/summarize_profile.py): HBM bytes and
        # vector instructions of one launch of that kernel
        traffic, valu = None, None
        for name in ("traffic.json", "valu.json"):
            f = os.path.join(ROOT, "profiles", name)
            if os.path.exists(f):
                try:
                    v = json.load(open(f)).get(dom)
                except Exception:
                    v = None
                if name == "traffic.json":
                    traffic = v
                else:
                    valu = v
        # the path is integer-issue bound, so say how close the kernel is to THAT ceiling:
        # 1024 SIMDs, one wave64 VALU instruction per 4 clocks each, at the 2.4 GHz peak clock
        issue_peak = 1024 * 2.4e9 / 4.0
        int_issue = None if valu is None else {"valu_insts_per_launch": valu, "achieved": valu / (per_launch_ms[dom] * 1e-3),
                                               "peak": issue_peak, "unit": "wave-instr/s",
                                               "frac": valu / (per_launch_ms[dom] * 1e-3) / issue_peak}
        line = {
            "metric": "barcode calls/sec vs 737K 10x whitelist", "value": value, "unit": "calls/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "%d synthetic ONT reads per GPU (mean %.0f bp) vs %d-entry whitelist: K1 extract + K2 nearest16(max_ed=2)"
                                   % (n, total_bytes / n, len(wl)),
                       "reads_per_gpu": n, "whitelist": len(wl), "sw_windows_per_step": int(nwin), "pipeline_counts": stats,
                       "parallelism": "reads sharded per GPU, no collectives"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": per_launch_ms[dom], "int_issue": int_issue},
            "kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(per_launch_ms.items())},
        }
        if not args.no_cpu_baseline and world == 1:       # the CPU leg runs at N = 1 only (256 host threads would fight the other ranks)
            line["cpu_baseline"] = cpu_baseline(bases, off_u, wl, 100000, 48)
        print(json.dumps(line))
    if world > 1:
        bdist.barrier(dev)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
