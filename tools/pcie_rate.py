#!/usr/bin/env python3
"""Extraction rate when the boundary hands over HOST buffers (what the C ABI's bdg_extract_batch and
bdg_extract_submit / bdg_extract_collect do), PCIe transfer included: 1M synthetic reads (1.05 GB)
(a) one pageable call, (b) chunks of 100,000 reads from pinned memory, two in flight per context.
Compared with the device-resident rate of bench.py this is the cost of the link, never the bench's `value`.
Prints one JSON object per line.  Builder tool."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from badger_amd import _native, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    wl = synth.make_whitelist(737280)
    bases, off = synth.make_reads(n, wl, seed=1, device="cuda")
    bases, off = bases.cpu().numpy(), off.cpu().numpy().astype(np.uint64)
    total = int(off[-1])
    ctx = _native.Context(0)
    ref = ctx.extract_batch(bases, off, 12)                      # warm: workspaces, code objects
    for label, reps in (("pageable, one bdg_extract_batch call", 3),):
        t0 = time.perf_counter()
        for _ in range(reps):
            got = ctx.extract_batch(bases, off, 12)
        dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"path": label, "reads": n, "bytes": total, "s": round(dt, 4), "reads_per_s": round(n / dt),
                          "GB_per_s": round(total / dt / 1e9, 2), "equal_to_first_call": bool((got == ref).all())}), flush=True)
    # pinned chunks, two in flight
    chunk = 100000
    pieces = []
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        pb = torch.from_numpy(bases[int(off[a]):int(off[b])].copy()).pin_memory()
        pad = torch.zeros(int(pb.numel()) + 64, dtype=torch.uint8).pin_memory()
        pad[:pb.numel()] = pb
        po = torch.from_numpy((off[a:b + 1] - off[a]).astype(np.int64)).pin_memory()
        pieces.append((pad, po, b - a))
    # the link alone: the same pinned chunks copied with torch, nothing else
    dst = torch.empty(int(max(p[0].numel() for p in pieces)), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for pb, _, _ in pieces:
        dst[:pb.numel()].copy_(pb, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"path": "torch copy of the pinned chunks alone (no kernels)", "bytes": total, "s": round(dt, 4), "GB_per_s": round(total / dt / 1e9, 2)}), flush=True)
    t_submit = 0.0
    for reps in (1, 5):                       # the first pass (pages touched by the device for the first time) is not the rate
        t_submit = 0.0
        t0 = time.perf_counter()
        for _ in range(reps):
            out, inflight = [], []
            for k, (pb, po, m) in enumerate(pieces):
                if len(inflight) == 2:
                    s, mm = inflight.pop(0)
                    out.append(ctx.extract_collect(s, mm))
                ts = time.perf_counter()
                ctx.extract_submit(k % 2, pb.data_ptr(), po.data_ptr(), m, 12)
                t_submit += time.perf_counter() - ts
                inflight.append((k % 2, m))
            for s, mm in inflight:
                out.append(ctx.extract_collect(s, mm))
        dt = (time.perf_counter() - t0) / reps
        got = np.concatenate(out)
        if reps == 1:
            continue
        print(json.dumps({"path": "pinned chunks of %d reads, bdg_extract_submit / collect, two in flight" % chunk, "reads": n, "bytes": total,
                          "s": round(dt, 4), "reads_per_s": round(n / dt), "GB_per_s": round(total / dt / 1e9, 2),
                          "host_time_inside_submit_s": round(t_submit / reps, 4),
                          "equal_to_first_call": bool((got == ref).all())}), flush=True)


if __name__ == "__main__":
    main()
