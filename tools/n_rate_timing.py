#!/usr/bin/env python3
"""Extraction time of 1M synthetic reads as a function of the fraction of bases replaced by N (GPU box)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from badger_amd import _native, synth
dev = torch.device("cuda", 0)
wl = synth.make_whitelist(737280)
ctx = _native.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
n = 1000000
bases0, off = synth.make_reads(n, wl, seed=1, device=dev)
total = int(off[-1])
recs = torch.zeros((n, 8), dtype=torch.int32, device=dev)
for rate in (0.0, 1e-6, 1e-5, 1e-4, 1e-3):
    bases = bases0.clone()
    if rate > 0:
        g = torch.Generator(device=dev); g.manual_seed(7)
        bases[torch.rand(total, generator=g, device=dev) < rate] = ord("N")
    bases = torch.cat([bases, torch.zeros(64, dtype=torch.uint8, device=dev)])
    for _ in range(3):
        ctx.extract_batch_dev(bases, off.contiguous(), n, total, 12, recs)
    ctx.extract_status()
    ctx.profile(True); ctx.profile_reset()
    for _ in range(5):
        ctx.extract_batch_dev(bases, off.contiguous(), n, total, 12, recs)
    torch.cuda.synchronize()
    prof = ctx.profile_read(); ctx.profile(False)
    print("N rate %g: " % rate + ", ".join("%s %.3f" % (k, v[1] / max(1, v[0])) for k, v in sorted(prof.items())), flush=True)
