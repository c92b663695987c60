#!/usr/bin/env python3
"""End-to-end wall time of the stage-2 CLI (python -m badger_amd.badger) on one MI355X box: 1M synthetic reads,
(a) from the stage-1 TSV (host route), (b) from the FASTQ itself (extraction + device hand-off), thresholds 1 and 2.
Prints one JSON object per line.  S2_EXTRA="-hs": more flags for badger.py.  Builder tool."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from badger_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    tmp = os.environ.get("TMPDIR", "/tmp")
    wl = synth.make_whitelist(737280)
    import numpy as np
    from cli_throughput import helper
    L = helper(tmp)
    fq = os.path.join(tmp, "s2_reads.fastq")
    if os.path.exists(fq):
        os.remove(fq)
    for k in range(0, n, 1000000):                 # slabs of 1 M reads, seeds 1, 2, ... (as tools/cli_throughput.py makes them)
        m = min(1000000, n - k)
        tb, to = synth.make_reads(m, wl, seed=1 + k // 1000000, device="cuda")
        b, o = tb.cpu().numpy(), to.cpu().numpy().astype(np.uint64)
        assert L.fq_append(fq.encode(), b.ctypes.data, o.ctypes.data, m, k, b"read_") > 0
    wlf = os.path.join(tmp, "s2_wl.txt")
    with open(wlf, "w") as f:
        f.write("\n".join(synth.rank_to_str(r) for r in wl) + "\n")
    tsv = os.path.join(tmp, "s2_stage1.tsv")
    t0 = time.perf_counter()
    subprocess.check_call([sys.executable, "-m", "badger_amd.extract_raw_barcodes", "--mode", "tenX_v3", "-i", fq, "-o", tsv, "-t", "1"],
                          cwd=ROOT, stdout=subprocess.DEVNULL)
    print(json.dumps({"stage": 1, "reads": n, "wall_s": round(time.perf_counter() - t0, 2)}), flush=True)
    if os.environ.get("S2_PROFILE"):               # where the time of one run goes (cProfile, top of the cumulative list)
        prof = os.path.join(tmp, "s2.prof")
        subprocess.check_call([sys.executable, "-m", "cProfile", "-o", prof, "-m", "badger_amd.badger", "-r", fq, "-d", "tenX_v3", "-l", wlf,
                               "-c", "5000", "-t", os.environ["S2_PROFILE"], "-o", os.path.join(tmp, "s2_prof_out")], cwd=ROOT, stdout=subprocess.DEVNULL)
        import pstats
        pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
        return
    outs = {}
    for thr in ("1", "2"):
        for label, reads in (("tsv", tsv), ("fastq", fq)):
            prefix = os.path.join(tmp, "s2_out_%s_%s" % (label, thr))
            timing = os.path.join(tmp, "s2_timing.jsonl")
            best = None
            for rep in range(2):
                if os.path.exists(timing):
                    os.remove(timing)
                t0 = time.perf_counter()
                subprocess.check_call([sys.executable, "-m", "badger_amd.badger", "-r", reads, "-d", "tenX_v3", "-l", wlf, "-c", "5000",
                                       "-t", thr, "-tr", "16", "-o", prefix] + os.environ.get("S2_EXTRA", "").split(), cwd=ROOT, stdout=subprocess.DEVNULL,
                                      env=dict(os.environ, BADGER_AMD_STAGE2_TIMING=timing))
                wall = time.perf_counter() - t0
                if best is None or wall < best[0]:
                    best = (wall, json.loads(open(timing).read().strip().split("\n")[-1]))
            wall, phases = best
            outs[(label, thr)] = open(prefix + "_output_file.tsv").read()
            assigned = sum(1 for l in outs[(label, thr)].split("\n")[1:] if l and not l.endswith("*"))
            print(json.dumps({"stage": 2, "input": label, "threshold": int(thr), "reads": n, "wall_s": round(wall, 2),
                              "reads_per_s": round(n / wall), "assigned": assigned, "phases_s": phases}), flush=True)
        print(json.dumps({"threshold": int(thr), "tsv_route_equals_device_route": outs[("tsv", thr)] == outs[("fastq", thr)]}), flush=True)


if __name__ == "__main__":
    main()
