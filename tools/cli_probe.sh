#!/bin/bash
# Experiments around the stage-1 command line on a prepared input (tools/cli_throughput.py leaves its files in $TMPDIR):
# wall clock and the pipeline's own breakdown for a few settings of the tuning knobs.
IN=${1:-$TMPDIR/cli_reads.fastq}
OUT=$TMPDIR/cli_probe.tsv
run() {
    rm -f $TMPDIR/probe_timing.jsonl
    local t0=$(date +%s.%N)
    env "$@" BADGER_AMD_STAGE1_TIMING=$TMPDIR/probe_timing.jsonl python -m badger_amd.extract_raw_barcodes --mode tenX_v3 -i $IN -o $OUT -t 16 > /dev/null 2> $TMPDIR/probe.err
    local t1=$(date +%s.%N)
    echo "$* wall $(python3 -c "print(round($t1 - $t0, 3))") $(grep -h "pinned allocations\|stage1:" $TMPDIR/probe.err | tr '\n' ' ') $(cat $TMPDIR/probe_timing.jsonl | cut -c1-400)"
}
run A=1
run BADGER_AMD_INGEST_DEBUG=1
run BADGER_AMD_SEGMENT_MB=8
run BADGER_AMD_SEGMENT_MB=24
run BADGER_AMD_SEGMENT_MB=32
run BADGER_AMD_INFLIGHT=3
run BADGER_AMD_INFLIGHT=4
run BADGER_AMD_FORMAT_THREADS=3
run BADGER_AMD_FORMAT_THREADS=6
