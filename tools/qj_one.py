#!/usr/bin/env python3
"""One configuration of the q-gram join, three launches (for rocprofv3)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from badger_amd import _native, synth
from bench_ops import observed_barcodes
dev = torch.device("cuda", 0)
ctx = _native.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
n = 500000
ranks = observed_barcodes(n, synth.make_whitelist(737280))
d_ranks = torch.from_numpy(ranks.astype(np.int64)).to(dev).to(torch.int32)
cap = 64 * n
d_edges = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
d_n = torch.zeros(1, dtype=torch.int64, device=dev)
ctx.graph_set_algo(3)
for _ in range(3):
    ctx.graph_edges_dev(d_ranks, n, 2, 4, d_edges, cap, d_n)
torch.cuda.synchronize()
print(int(d_n[0]))
