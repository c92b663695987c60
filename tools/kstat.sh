#!/bin/bash
# usage: kstat.sh <tag>   (GPU box) kernel durations of a short bench run, parity ignored
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstat_$1; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/log.txt 2>&1
f=$(find $OUT -name "*_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]; i=n.find("k_")
    if i>=0 and int(r["Calls"])>20: print(n[i:n.find("(",i)], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
find $OUT -name "*.csv" -delete
