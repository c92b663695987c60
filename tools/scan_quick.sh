#!/bin/bash
# Runs on the GPU box: a short bench line plus one PMC pass, then the instruction counts of k_scan_reads (iteration helper).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/scanq
rm -rf $OUT; mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"], 4), "parity", d.get("parity_sample"), {k: v for k, v in d["kernels_ms_per_step"].items()})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_scan_reads" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for c, d in sorted(acc.items()):
    v = [sum(x) for x in d.values()]
    print(c, round(sum(v) / len(v) / 1e6, 2), "M per launch")
PY
rm -rf $OUT/pmc
