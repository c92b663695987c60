#!/usr/bin/env python3
"""End-to-end wall time of the stage-2 CLI (python -m badger_amd.badger) on one MI355X box: 1M synthetic reads,
(a) from the stage-1 TSV (host route), (b) from the FASTQ itself (extraction + device hand-off), thresholds 1 and 2.
Prints one JSON object per line.  Builder tool."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from badger_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    tmp = os.environ.get("TMPDIR", "/tmp")
    wl = synth.make_whitelist(737280)
    bases, off = synth.make_reads(n, wl, seed=1, device="cuda")
    seqs = synth.reads_to_list(bases.cpu(), off.cpu())
    fq = os.path.join(tmp, "s2_reads.fastq")
    with open(fq, "w") as f:
        for a in range(0, n, 50000):
            f.write("".join("@read_%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in zip(range(a, a + 50000), seqs[a:a + 50000])))
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
            t0 = time.perf_counter()
            subprocess.check_call([sys.executable, "-m", "badger_amd.badger", "-r", reads, "-d", "tenX_v3", "-l", wlf, "-c", "5000",
                                   "-t", thr, "-o", prefix], cwd=ROOT, stdout=subprocess.DEVNULL)
            wall = time.perf_counter() - t0
            outs[(label, thr)] = open(prefix + "_output_file.tsv").read()
            assigned = sum(1 for l in outs[(label, thr)].split("\n")[1:] if l and not l.endswith("*"))
            print(json.dumps({"stage": 2, "input": label, "threshold": int(thr), "reads": n, "wall_s": round(wall, 2),
                              "reads_per_s": round(n / wall), "assigned": assigned}), flush=True)
        print(json.dumps({"threshold": int(thr), "tsv_route_equals_device_route": outs[("tsv", thr)] == outs[("fastq", thr)]}), flush=True)


if __name__ == "__main__":
    main()
