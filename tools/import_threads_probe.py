#!/usr/bin/env python3
"""bdg_import_stage1_tsv on a stage-1 TSV with 4 / 8 / 16 / 32 parser threads (BADGER_AMD_IMPORT_THREADS), best of three.
usage: import_threads_probe.py <stage-1 TSV>.  Builder tool."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from badger_amd import _native
_native.PRELOAD_TORCH = False
_native.load()
p = sys.argv[1]
for th in ("4", "8", "16", "32"):
    os.environ["BADGER_AMD_IMPORT_THREADS"] = th
    best = 9
    for rep in range(3):
        t0 = time.perf_counter(); ids, r, u = _native.import_stage1_tsv(p, 16); dt = time.perf_counter() - t0
        best = min(best, dt); del ids, r, u
    print("threads", th, round(best, 3), "s")
