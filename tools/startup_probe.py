#!/usr/bin/env python3
"""Where the fixed cost of a command-line run goes on the GPU box: interpreter start, imports, library load, context
creation, first launches, process exit.  One JSON line.  Builder tool."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, os
t0 = time.perf_counter()
import numpy as np
t1 = time.perf_counter()
sys.path.insert(0, %r)
from badger_amd import _native
_native.PRELOAD_TORCH = False
_native.load()
t2 = time.perf_counter()
ctx = _native.Context(0)
t3 = time.perf_counter()
from badger_amd import synth
b = np.frombuffer(b"ACGT" * 300, dtype=np.uint8).copy()
off = np.array([0, 600, 1200], dtype=np.uint64)
ctx.extract_batch(b, off, 12)
t4 = time.perf_counter()
ctx.extract_batch(b, off, 12)
t5 = time.perf_counter()
mode = sys.argv[1]
print("%%f %%f %%f %%f %%f %%f" %% (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, time.time()), flush=True)
if mode == "_exit":
    os._exit(0)
''' % ROOT


def run(mode):
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", CHILD, mode], capture_output=True, text=True)
    t1 = time.time()
    f = [float(x) for x in p.stdout.split()]
    return {"mode": mode, "wall_s": round(t1 - t0, 3), "import_numpy": round(f[0], 3), "load_library": round(f[1], 3), "bdg_init": round(f[2], 3),
            "first_extract": round(f[3], 3), "second_extract": round(f[4], 3), "exit_s": round(t1 - f[5], 3),
            "before_python_runs": round(t1 - t0 - (t1 - f[5]) - sum(f[:5]), 3)}


t0 = time.time()
subprocess.run([sys.executable, "-c", "pass"])
bare = time.time() - t0
for mode in ("normal", "normal", "_exit"):
    d = run(mode)
    d["bare_interpreter_s"] = round(bare, 3)
    print(json.dumps(d), flush=True)
