#!/bin/bash
# Runs on the GPU box: per-kernel stats + PMC passes (separate runs, as the guide prescribes).
# usage: tools/pmc_profile.sh <outdir-under-gpurun_out> [extra bench.py arguments, e.g. --config 5]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the stats pass runs bench.py as the driver does (with its clock ramp: the average duration is the steady state's);
# the counter passes skip the ramp (counters are per launch and do not depend on the clock) to stay small
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline $* > $OUT/stats.log 2>&1
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-ramp --no-cpu-baseline $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- $CMD > $OUT/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc5 -- $CMD > $OUT/pmc5.log 2>&1
# keep only our kernels (names start with "(anonymous namespace)::k_") to stay under the merge limit
for d in stats pmc1 pmc2 pmc3 pmc4 pmc5; do
  for f in $(find $OUT/$d -name "*.csv"); do
    head -1 $f > $f.filtered; grep "::k_" $f >> $f.filtered; mv $f.filtered $f
  done
done
du -sh $OUT
