#!/bin/bash
# Runs on the GPU box: kernel stats + PMC passes (separate runs, as MI355X_MICROARCH.md prescribes) of an arbitrary
# python command.   usage: tools/prof_cmd.sh <outdir-under-gpurun_out> <script.py> [args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $SCRIPT "$@" > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc1 -- python3 $SCRIPT "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc2 -- python3 $SCRIPT "$@" > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $SCRIPT "$@" > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $SCRIPT "$@" > $OUT/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc5 -- python3 $SCRIPT "$@" > $OUT/pmc5.log 2>&1
for d in stats pmc1 pmc2 pmc3 pmc4 pmc5; do
  for f in $(find $OUT/$d -name "*.csv"); do
    head -1 $f > $f.filtered; grep "::k_" $f >> $f.filtered; mv $f.filtered $f
  done
done
du -sh $OUT
