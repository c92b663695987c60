#!/usr/bin/env python3
"""End-to-end throughput of the stage-1 CLI (python -m badger_amd.extract_raw_barcodes) on one MI355X box:
1M synthetic reads as FASTQ (plain and gzipped) -> TSV, wall clock of the whole process, output checked
byte-for-byte against the rows the CPU oracle's records give.  Also the parser alone (no GPU work) for scale.
Prints one JSON object per line.  Builder tool (the numbers go to BASELINE.md / DESIGN.md), not the bench contract."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from badger_amd import _native, synth  # noqa: E402
from badger_amd.barcode_extraction.barcode_callers import record_to_row  # noqa: E402


def main():
    from oracle import pyoracle as orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    tmp = os.environ.get("TMPDIR", "/tmp")
    wl = synth.make_whitelist(737280)
    bases, off = synth.make_reads(n, wl, seed=1, device="cuda")
    bases, off = bases.cpu(), off.cpu()
    seqs = synth.reads_to_list(bases, off)
    fq = os.path.join(tmp, "cli_reads.fastq")
    t0 = time.perf_counter()
    with open(fq, "w") as f:
        for a in range(0, n, 50000):
            f.write("".join("@read_%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in zip(range(a, a + 50000), seqs[a:a + 50000])))
    subprocess.check_call("gzip -1 -k -f %s" % fq, shell=True)
    # the same reads as BGZF (what bgzip writes: 64 KiB gzip members that state their size), written here with zlib
    import struct
    import zlib
    bgz = os.path.join(tmp, "cli_reads_bgzf.fastq.gz")
    with open(fq, "rb") as src, open(bgz, "wb") as dst:
        while True:
            piece = src.read(65280)
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = co.compress(piece) + co.flush()
            dst.write(b"\x1f\x8b\x08\x04\0\0\0\0\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(body) + 8 - 1))
            dst.write(body + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
            if not piece:
                break
    sizes = {"fastq": os.path.getsize(fq), "fastq.gz": os.path.getsize(fq + ".gz"), "bgzf.fastq.gz": os.path.getsize(bgz)}
    cores = len(os.sched_getaffinity(0))
    recs = orc.extract_batch(bases.numpy(), off.numpy().astype(np.uint64), 12, threads=cores)
    want = "#read_id\tbarcode\tUMI\tBC_score\tvalid_UMI\tstrand\tpolyT_start\tR1_end\n" + \
           "".join(record_to_row("read_%d" % i, s, r) + "\n" for i, (s, r) in enumerate(zip(seqs, recs)))
    print(json.dumps({"prepared_s": round(time.perf_counter() - t0, 1), "reads": n, "bytes": sizes}), flush=True)
    for path in (fq, fq + ".gz", bgz):
        # parser alone
        t0 = time.perf_counter()
        ing = _native.Ingest(path, 100000, 4)
        got = 0
        while True:
            ch = ing.next()
            if ch.n == 0:
                break
            got += ch.n
            ing.release(ch)
        ing.close()
        t_parse = time.perf_counter() - t0
        for threads in ("1", "16"):
            out = os.path.join(tmp, "cli_out_%s.tsv" % threads)
            t0 = time.perf_counter()
            subprocess.check_call([sys.executable, "-m", "badger_amd.extract_raw_barcodes", "--mode", "tenX_v3", "-i", path,
                                   "-o", out, "-t", threads], cwd=ROOT, stdout=subprocess.DEVNULL)
            wall = time.perf_counter() - t0
            text = open(out).read()
            if threads == "1":
                same = text == want
            else:
                same = "".join(l + "\n" for l in text.split("\n")[:-1] if not l.startswith("#")) == want.split("\n", 1)[1]
            print(json.dumps({"input": os.path.basename(path), "threads_flag": threads, "reads": n, "wall_s": round(wall, 2),
                              "reads_per_s": round(n / wall), "input_MB_per_s": round(os.path.getsize(path) / wall / 1e6),
                              "parser_alone_s": round(t_parse, 2), "parser_alone_reads_per_s": round(got / t_parse),
                              "tsv_identical_to_oracle_rows": same}), flush=True)


if __name__ == "__main__":
    main()
