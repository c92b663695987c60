#!/usr/bin/env python3
"""Condense a tools/pmc_profile.sh output directory into profiles/<tag>_summary.json and
profiles/traffic.json (HBM bytes per launch per kernel, what bench.py reports as roofline.traffic) and
profiles/valu.json (vector instructions per launch, bench.py's roofline.int_issue).

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced streaming reads, so it is
doubled for the streaming kernels (k_scan_reads); WRITE_SIZE is exact for 16-byte stores.
For gather-dominated kernels the doubling is uncalibrated and reported as a bracket [x1, x2]."""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
more = sys.argv[3:]              # further pass directories (--config 3 / 5 runs): their kernels join traffic.json / valu.json / lds.json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# kernels -> the names the library times them under (ScopedKernelTimer; bench.py's kernels_ms tables use those).  Several
# kernels under one timer are one entry here: their per-launch figures add up (each runs once per pass).
TIMERS = (("k_d2_pairs_w<1", "k_d1_pairs"), ("k_d2_pairs<1", "k_d1_pairs"), ("k_d2_pairs_w<2", "k_d2_pairs"), ("k_d2_pairs<2", "k_d2_pairs"),
          ("k_d2_rows<false", "k_d2_count"), ("k_d2_rows<true", "k_d2_emit"), ("k_d1_rows<false", "k_d1_count"), ("k_d1_rows<true", "k_d1_emit"),
          ("k_part_colscan", "k_%s_scan"), ("k_part_bases", "k_%s_scan"), ("k_part_split<unsigned int>", "k_%s_split"),
          ("k_distinct_rows", "k_distinct_rows"), ("k_part_split<unsigned long", "k_distinct_rows"),
          ("k_distinct_offsets", "k_distinct_finish"), ("k_distinct_compact", "k_distinct_finish"))
SUB = {"": "d2"}                               # which join the shared partition kernels of a pass directory belong to


def kname(full, sub=""):
    i = full.find("k_")
    name = full[i:full.find("(", i)] if "(" in full[i:] else full[i:]
    for prefix, timer in TIMERS:
        if name.startswith(prefix):
            return timer % ("d1" if sub == "_c3" else "d2") if "%s" in timer else timer
    return name.split("<", 1)[0]              # (template arguments of a kernel are not part of its name here)


summary = collections.defaultdict(dict)


def take_dir(one, sub):
    """one pmc_profile.sh output directory; a kernel that an earlier directory named keeps that directory's figures"""
    known = {k for k in summary if "calls" in summary[k]}
    for f in glob.glob(os.path.join(one, "stats", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            k = kname(r["Name"], sub)
            if k in known:
                continue
            s = summary[k]
            s["calls"] = max(s.get("calls", 0), int(r["Calls"]))
            s["avg_ns"] = s.get("avg_ns", 0.0) + float(r["AverageNs"])     # (kernels under one timer: one after the other)
            s["min_ns"] = s.get("min_ns", 0.0) + float(r["MinNs"])
            s["max_ns"] = s.get("max_ns", 0.0) + float(r["MaxNs"])
    for d in sorted(glob.glob(os.path.join(one, "pmc*"))):
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if len(files) != 1:          # (one process per pass; more means leftovers of an earlier run in the same directory)
            sys.exit("%s holds %d counter files, expected 1" % (d, len(files)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        raw = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in files:
            for r in csv.DictReader(open(f)):
                raw[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        # per timer name: the sum over its kernels of each kernel's mean per launch (kept as a one-element list per kernel)
        for full, cs in raw.items():
            for c, v in cs.items():
                agg[kname(full, sub)][c].append((len(v), sum(v) / len(v)))
        # the pass as committed: per kernel and counter, launches and mean per launch
        with open(os.path.join(ROOT, "profiles", "%s%s_%s.csv" % (tag, sub, os.path.basename(d))), "w") as out:
            out.write("kernel,counter,launches,mean_per_launch\n")
            for k in sorted(agg):
                for c in sorted(agg[k]):
                    out.write("%s,%s,%d,%.1f\n" % (k, c, max(n for n, _ in agg[k][c]), sum(m for _, m in agg[k][c])))
        for k, cs in agg.items():
            for c, v in cs.items():
                summary[k].setdefault(c, sum(m for _, m in v))


take_dir(src, "")
for one in more:
    take_dir(one, "_" + os.path.basename(one.rstrip("/")).split("_")[-1])
STREAMING = {"k_scan_reads"}
traffic = {}
for k, s in summary.items():
    if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
        rd1, wr = s["FETCH_SIZE"] * 1024.0, s["WRITE_SIZE"] * 1024.0
        s["hbm_bytes_fetch_x1"] = rd1 + wr
        s["hbm_bytes_fetch_x2"] = 2 * rd1 + wr
        traffic[k] = s["hbm_bytes_fetch_x2"] if k in STREAMING else s["hbm_bytes_fetch_x1"]
sys.path.insert(0, ROOT)
from badger_amd import _native  # noqa: E402  (dlopen only: which build of the library the counters belong to)
lib = _native.load().bdg_version().decode()
parts = lib.split()
# (the counters belong to the device code; "summarised_with" only says which build of the library ran this script)
meta = {"kernels": parts[parts.index("kernels") + 1] if "kernels" in parts[:-1] else lib, "tag": tag, "summarised_with": lib}
traffic["_meta"] = meta
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(summary, open(os.path.join(ROOT, "profiles", tag + "_summary.json"), "w"), indent=1, sort_keys=True)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
valu = {k: s["SQ_INSTS_VALU"] for k, s in summary.items() if "SQ_INSTS_VALU" in s}
valu["_meta"] = meta
json.dump(valu, open(os.path.join(ROOT, "profiles", "valu.json"), "w"), indent=1, sort_keys=True)
# LDS-array cycles per launch (SQ_LDS_IDX_ACTIVE: all cycles, SQ_LDS_BANK_CONFLICT: the extra ones), MI355X_MICROARCH.md "LDS"
lds = {k: {"cycles": s["SQ_LDS_IDX_ACTIVE"], "conflict_cycles": s.get("SQ_LDS_BANK_CONFLICT")} for k, s in summary.items() if "SQ_LDS_IDX_ACTIVE" in s}
lds["_meta"] = meta
json.dump(lds, open(os.path.join(ROOT, "profiles", "lds.json"), "w"), indent=1, sort_keys=True)
for k in sorted(summary):
    s = summary[k]
    print("%-26s avg %8.1f us  VALU %6.1fM  hbm %s" % (k, s.get("avg_ns", 0) / 1e3, s.get("SQ_INSTS_VALU", 0) / 1e6,
                                                     ("%.3f GB" % (traffic[k] / 1e9)) if k in traffic else "-"))
print("library:", meta["summarised_with"])
