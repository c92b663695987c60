#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of tools/bench_ops.py (K3 all paths, K2 alone, distinct counting).
# usage: tools/ops_profile.sh <outdir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py > $OUT/ops.jsonl 2> $OUT/stats.log
for f in $(find $OUT/stats -name "*.csv"); do
  head -1 $f > $f.filtered; grep -E "::k_|hipcub|rocprim" $f >> $f.filtered; mv $f.filtered $f
done
find $OUT/stats -name "*_kernel_trace.csv" -delete
du -sh $OUT
