#!/usr/bin/env python3
"""One steady-state step of the pipelined bench loop as a timeline: which kernel ran when, on which queue.
Reads the kernel trace of `rocprofv3 --kernel-trace -- python3 bench.py --no-graph --no-cpu-baseline --no-ramp --steps 30 --warmup 5`
(csv) and prints, for one k_scan_reads launch in the middle of the run up to the next one, every kernel's start relative to that
scan's start and its duration.  usage: pipeline_timeline.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if "::k_" not in name:
            continue
        short = re.search(r"::(k_[a-z0-9_]+)", name).group(1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?")))
rows.sort()
scans = [i for i, r in enumerate(rows) if r[2] == "k_scan_reads"]
a, b = scans[len(scans) // 2], scans[len(scans) // 2 + 1]
t0 = rows[a][0]
print("# start_us  dur_us  queue  kernel   (one step: from a k_scan_reads launch to the next; step = %.1f us)" % ((rows[b][0] - t0) / 1e3))
for s, e, n, q in rows[a:b + 1]:
    print("%9.1f %7.1f  %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
