#!/bin/bash
# CPU-only: the read ingest (reader threads + assembler, badger_amd/csrc/ingest.cpp) under ThreadSanitizer and under
# AddressSanitizer + UBSan: a BGZF file, a BGZF file with a plain gzip member appended, a BGZF file with a damaged block, the
# same reads as plain FASTQ, a FASTQ with a malformed record in the middle, a BAM file, and the reads as one plain gzip stream
# (whole and cut off: the parallel inflate of csrc/pgunzip.cpp) - each with one segment and with
# segments of 20,011 and 700 bytes (the parallel path, its fall-back and the record stitching), each read to the end twice
# and closed in mid-file once.
# GPU sanitizers are not available on the pool; this covers the host threads of the path.
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=${TMPDIR:-/tmp}/bdg_sanitize; mkdir -p $T
python3 - "$ROOT" "$T" <<'PY'
import gzip, sys
import numpy as np
root, t = sys.argv[1], sys.argv[2]
sys.path.insert(0, root + "/tests"); sys.path.insert(0, root)
from test_ingest import _bgzf
rng = np.random.default_rng(3)
rnd = lambda k: "".join("ACGTN"[i] for i in rng.integers(0, 5, k))
raw = "".join("@r%d w\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in ((i, rnd(int(rng.integers(0, 4000)))) for i in range(3000))).encode()
open(t + "/a.fastq.gz", "wb").write(_bgzf(raw, 4000))
half = raw.index(b"@r1500 ")
open(t + "/b.fastq.gz", "wb").write(_bgzf(raw[:half], eof_marker=False) + gzip.compress(raw[half:]))
bad = bytearray(_bgzf(raw)); bad[len(bad) // 2] ^= 0x55
open(t + "/c.fastq.gz", "wb").write(bytes(bad))
open(t + "/d.fastq", "wb").write(raw)
cut = raw.index(b"@r2000 ")
open(t + "/e.fastq", "wb").write(raw[:cut] + b"@x\nACGT\nIIII\n" + raw[cut:])
import bamio
recs = [("q%d" % i, [4, 0, 16, 256, 2048][i % 5], rnd(int(rng.integers(1, 900))).replace("N", "A"), [], b"") for i in range(2000)]
open(t + "/f.bam", "wb").write(bamio.bgzf(bamio.bam_raw(recs), block=3000))
open(t + "/g.fastq.gz", "wb").write(gzip.compress(raw, 6))                       # one plain gzip stream: the parallel inflate
open(t + "/h.fastq.gz", "wb").write(gzip.compress(raw, 6)[:400000])               # the same cut off
PY
FLAGS="-O1 -g -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/include -I$ROOT/badger_amd/csrc"
LIBS="-L/opt/rocm/lib -lamdhip64 -lz -lpthread -ldl -Wl,-rpath,/opt/rocm/lib"
g++ $FLAGS -fsanitize=thread $ROOT/tools/sanitize/ingest_driver.cpp $ROOT/badger_amd/csrc/ingest.cpp $ROOT/badger_amd/csrc/pgunzip.cpp -o $T/drv_tsan $LIBS 2>/dev/null
g++ $FLAGS -fsanitize=address,undefined $ROOT/tools/sanitize/ingest_driver.cpp $ROOT/badger_amd/csrc/ingest.cpp $ROOT/badger_amd/csrc/pgunzip.cpp -o $T/drv_asan $LIBS 2>/dev/null
rc=0
export BADGER_AMD_GUNZIP_MIN_KB=0 BADGER_AMD_GUNZIP_CHUNK_KB=16      # (plain gzip members go through pgunzip.cpp, many chunks)
for f in a.fastq.gz b.fastq.gz c.fastq.gz d.fastq e.fastq f.bam g.fastq.gz h.fastq.gz; do for t in 0 1 3; do for seg in 0 20011 700; do
    for drv in drv_tsan drv_asan; do
        out=$($T/$drv $T/$f $t $seg 2>&1) || true
        if echo "$out" | grep -q "Sanitizer\|runtime error"; then echo "FAIL $drv $f threads=$t segment=$seg"; echo "$out" | head -20; rc=1; fi
    done
done; done; done
[ $rc = 0 ] && echo "ingest sanitizers: clean (tsan, asan+ubsan; 8 files x 3 thread counts x 3 segment sizes)"
exit $rc
