#!/usr/bin/env python3
"""End-to-end throughput of the stage-1 command line (python -m badger_amd.extract_raw_barcodes) on one MI355X box, at a
size where fixed costs stop dominating: N synthetic reads (default 12.5 M = one GPU's share of BASELINE config 4; made
in 1 M-read slabs with seeds 1, 2, ...) as plain FASTQ, BGZF, BAM (--forms=...,bam) and (first CLI_GZ_SLABS slabs) plain gzip -> TSV.  Wall clock of the
whole process, where its time went (BADGER_AMD_STAGE1_TIMING: the native pipeline's own breakdown), the readers alone,
and the checks: the rows of the first slab equal the rows the CPU oracle's records give; every input form and both file
shapes give the same rows (sha256 over the non-header lines).  One JSON object per line.
Builder tool (the numbers go to DESIGN.md / profiles/), not the bench contract.

    python tools/cli_throughput.py [reads] [--gpus-rehearsal]
"""
import ctypes as C
import hashlib
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

SLAB = 1000000


def helper(tmp):
    import tempfile
    so = os.path.join(tempfile.mkdtemp(dir="/tmp"), "libfastx_tools.so")        # (not under TMPDIR: /dev/shm is mounted noexec)
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tools", "native", "fastx_tools.c"), "-lz"])
    L = C.CDLL(so)
    L.fq_append.restype = C.c_int64
    L.fq_append.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p]
    L.bam_raw_append.restype = C.c_int64
    L.bam_raw_append.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p]
    L.bgzf_compress_file.restype = C.c_int64
    L.bgzf_compress_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    return L


def rows_digest(path):
    """sha256 over the non-header lines, and their number"""
    h, n = hashlib.sha256(), 0
    with open(path, "rb") as f:
        for line in f:
            if not line.startswith(b"#"):
                h.update(line)
                n += 1
    return h.hexdigest()[:16], n


def run_cli(path, out, threads, timing, extra=()):
    if os.path.exists(timing):
        os.remove(timing)
    env = dict(os.environ, BADGER_AMD_STAGE1_TIMING=timing)
    t0 = time.perf_counter()
    subprocess.check_call([sys.executable, "-m", "badger_amd.extract_raw_barcodes", "--mode", "tenX_v3", "-i", path, "-o", out,
                           "-t", str(threads)] + list(extra), cwd=ROOT, stdout=subprocess.DEVNULL, env=env)
    wall = time.perf_counter() - t0
    br = json.loads(open(timing).read().strip().split("\n")[-1])
    return wall, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in br.items() if k.startswith("seconds") or k in ("chunks", "out_bytes")}


GZ_SLABS = int(os.environ.get("CLI_GZ_SLABS", "1"))


def main():
    from oracle import pyoracle as orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12500000
    tmp = os.environ.get("TMPDIR", "/tmp")
    L = helper(tmp)
    wl = synth.make_whitelist(737280)
    fq = os.path.join(tmp, "cli_reads.fastq")
    if os.path.exists(fq):
        os.remove(fq)
    t0 = time.perf_counter()
    first_rows = None
    done = 0
    while done < n:
        k = min(SLAB, n - done)
        tb, to = synth.make_reads(k, wl, seed=1 + done // SLAB, device="cuda")
        tb, to = tb.cpu(), to.cpu()
        bases, off = tb.numpy(), to.numpy().astype(np.uint64)
        if done < GZ_SLABS * SLAB:                      # the plain-gzip form holds the first GZ_SLABS slabs (gzip -1 of a slab takes ~10 s)
            gz1 = os.path.join(tmp, "cli_reads_1m.fastq")
            if done == 0 and os.path.exists(gz1):
                os.remove(gz1)
            assert L.fq_append(gz1.encode(), bases.ctypes.data, off.ctypes.data, k, done, b"read_") > 0
        if done == 0:
            k0 = min(k, 200000)
            recs = orc.extract_batch(bases[:int(off[k0])], off[:k0 + 1], 12, threads=16)
            seqs = synth.reads_to_list(tb[:int(off[k0])], to[:k0 + 1])
            first_rows = "".join(record_to_row("read_%d" % i, s, r) + "\n" for i, (s, r) in enumerate(zip(seqs, recs))).encode()
        assert L.fq_append(fq.encode(), bases.ctypes.data, off.ctypes.data, k, done, b"read_") > 0
        if "--forms=" in " ".join(sys.argv) and "bam" in [a for a in sys.argv if a.startswith("--forms=")][0]:
            assert L.bam_raw_append(os.path.join(tmp, "cli_reads.rawbam").encode(), bases.ctypes.data, off.ctypes.data, k, done, b"read_") > 0
        done += k
    forms = "plain,bgzf,gz"
    for a in sys.argv[1:]:
        if a.startswith("--forms="):
            forms = a.split("=", 1)[1]
    forms = forms.split(",")
    bgz = os.path.join(tmp, "cli_reads_bgzf.fastq.gz")
    sizes = {"fastq": os.path.getsize(fq)}
    if "bgzf" in forms:
        assert L.bgzf_compress_file(fq.encode(), bgz.encode(), 1) > 0
        sizes["bgzf.fastq.gz"] = os.path.getsize(bgz)
    bam = os.path.join(tmp, "cli_reads.bam")
    if "bam" in forms:
        assert L.bgzf_compress_file(os.path.join(tmp, "cli_reads.rawbam").encode(), bam.encode(), 1) > 0
        os.remove(os.path.join(tmp, "cli_reads.rawbam"))
        sizes["bam"] = os.path.getsize(bam)
    if "gz" in forms:
        subprocess.check_call("gzip -1 -k -f %s" % gz1, shell=True)
        sizes["1m.fastq.gz"] = os.path.getsize(gz1 + ".gz")
    print(json.dumps({"prepared_s": round(time.perf_counter() - t0, 1), "reads": n, "bytes": sizes,
                      "cores": len(os.sched_getaffinity(0)), "cpu_count": os.cpu_count()}), flush=True)
    timing = os.path.join(tmp, "cli_timing.jsonl")
    digests = {}
    inputs = [(fq, n)] * ("plain" in forms) + [(bgz, n)] * ("bgzf" in forms) + [(bam, n)] * ("bam" in forms) + [(gz1 + ".gz", min(n, GZ_SLABS * SLAB))] * ("gz" in forms)
    for path, nreads in inputs:
        for threads in (0, 1) if path == bgz else (0, 1, 4, 8, 16):
            # the readers alone (pageable buffers: no GPU, no pinning)
            t0 = time.perf_counter()
            ing = _native.Ingest(path, 100000, 8, pinned=False, inflate_threads=threads)
            got = 0
            while True:
                ch = ing.next()
                if ch.n == 0:
                    break
                got += ch.n
                ing.release(ch)
            ing.close()
            dt = time.perf_counter() - t0
            print(json.dumps({"input": os.path.basename(path), "readers_alone": True, "reader_threads": threads or "auto",
                              "reads": got, "wall_s": round(dt, 3), "reads_per_s": round(got / dt)}), flush=True)
        for tflag in ("1", "16"):
            out = os.path.join(tmp, "cli_out.tsv")
            best = None
            for rep in range(2):                        # second run: the page cache holds the input, the clocks are up
                wall, br = run_cli(path, out, tflag, timing)
                if best is None or wall < best[0]:
                    best = (wall, br)
            wall, br = best
            dg, nrows = rows_digest(out)
            digests.setdefault(nreads, set()).add(dg)
            head = open(out, "rb").read(len(first_rows) + 4096)
            body = b"".join(l for l in head.split(b"\n", 1)[1].splitlines(True) if not l.startswith(b"#")) if head else b""
            print(json.dumps({"input": os.path.basename(path), "threads_flag": tflag, "reads": nreads, "rows": nrows, "wall_s": round(wall, 3),
                              "reads_per_s": round(nreads / wall), "input_MB_per_s": round(os.path.getsize(path) / wall / 1e6),
                              "page_cache": "warm (the tool wrote the input seconds earlier: the file is read from memory, not from storage)",
                              "pipeline": br, "first_rows_equal_oracle": body[:len(first_rows)] == first_rows, "rows_sha256_16": dg}), flush=True)
    print(json.dumps({"every_form_and_shape_gave_the_same_rows": {str(k): len(v) == 1 for k, v in digests.items()}}), flush=True)
    if "--gpus-rehearsal" in sys.argv:
        # three contexts on this one device standing in for three devices: chunk k on context k mod 3
        env = dict(os.environ, BADGER_AMD_CONTEXTS_ON_ONE_DEVICE="1")
        for g in (1, 3):
            t0 = time.perf_counter()
            subprocess.check_call([sys.executable, "-m", "badger_amd.extract_raw_barcodes", "--mode", "tenX_v3", "-i", fq, "-o",
                                   os.path.join(tmp, "cli_out_g.tsv"), "-t", "16", "--gpus", str(g)], cwd=ROOT, stdout=subprocess.DEVNULL, env=env)
            wall = time.perf_counter() - t0
            print(json.dumps({"gpus_flag": g, "contexts_on_one_device": True, "reads": n, "wall_s": round(wall, 3), "reads_per_s": round(n / wall),
                              "rows_sha256_16": rows_digest(os.path.join(tmp, "cli_out_g.tsv"))[0]}), flush=True)


if __name__ == "__main__":
    main()
