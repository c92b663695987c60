#!/usr/bin/env python3
"""Stage 1 CLI: raw barcode extraction, same flags and TSV as the reference's
extract_raw_barcodes.py (reference extract_raw_barcodes.py:360-391), MI355X underneath.

    python -m badger_amd.extract_raw_barcodes --mode tenX_v3 -i reads.fq.gz -o out.tsv [-t N]

What differs from the reference, by design:
  * reads are cut into chunks of 100,000 (READ_CHUNK_SIZE, reference :32) and each chunk is
    ONE batch on the GPU; `--threads` no longer buys CPU parallelism.  It is kept
    because it selects the reference's two output shapes: threads == 1 writes one header
    and a tab-separated .stats (reference :162-173); threads > 1 writes a header per chunk
    and a space-separated .stats (reference :243-259) -- here always in input order, where
    the reference concatenates chunks in completion order.
  * every input format (FASTA / FASTQ / SAM, optionally gzipped or BGZF, and BAM) goes through the native pipeline
    (bdg_stage1_run): reader threads parse segments of the input into pinned chunks, chunk k+1 is on its way to / on
    the GPU while the rows of chunk k are formatted and written by further native threads.  `--threads` is the number
    of reader threads (1: one sequential reader).  No pysam, no Bio.
  * new optional flag: --gpus N.  Chunk k goes to device k mod N; submission is asynchronous, so
    the N devices work at the same time; rows are written in input order.
"""
import argparse
import gzip
import logging
import os
import sys
from collections import defaultdict, deque
from traceback import print_exc

from . import _native

from .barcode_extraction.barcode_callers import (ReadStats, TenXBarcodeExtractorV2, TenXBarcodeExtractorV3,
                                                 record_to_row)

logger = logging.getLogger("BarcodeGraph")

READ_CHUNK_SIZE = 100000
BARCODE_CALLING_MODES = {"tenX_v2": TenXBarcodeExtractorV2, "tenX_v3": TenXBarcodeExtractorV3}


# ----------------------------------------------------------------------------- readers
def _fasta_records(handle):
    rid, parts = None, []
    for line in handle:
        line = line.rstrip("\r\n")
        if line.startswith(">"):
            if rid is not None:
                yield rid, "".join(parts)
            fields = line[1:].split()
            rid, parts = (fields[0] if fields else ""), []
        elif rid is not None:
            parts.append(line.strip())
    if rid is not None:
        yield rid, "".join(parts)


def _fastq_records(handle):
    while True:
        head = handle.readline()
        if not head:
            return
        head = head.rstrip("\r\n")
        if not head:
            continue
        if not head.startswith("@"):
            raise ValueError("malformed FASTQ record header: %r" % head[:50])
        seq = handle.readline().rstrip("\r\n")
        plus = handle.readline()
        qual = handle.readline()
        if not plus.startswith("+") or len(qual.rstrip("\r\n")) != len(seq):
            raise ValueError("malformed FASTQ record %r" % head[:50])
        fields = head[1:].split()
        yield (fields[0] if fields else ""), seq


def _native_records(path, skip_secondary):
    """(read_id, sequence) through the native reader (SAM / BAM: the library's own decoder, no pysam)"""
    ing = _native.Ingest(path, READ_CHUNK_SIZE, 4, pinned=False, skip_secondary=skip_secondary)
    try:
        while True:
            ch = ing.next()
            if ch.n == 0:
                return
            recs = _native.chunk_reads(ch)
            ing.release(ch)
            yield from recs
    finally:
        ing.close()


def open_reads(input_file, skip_secondary=True):
    """-> iterator of (read_id, sequence); None for an unknown extension (reference :80-97,181-197)."""
    fname, ext = os.path.splitext(os.path.basename(input_file))
    ext = ext.lower()
    handle = None
    if ext in (".gz", ".gzip"):
        handle = gzip.open(input_file, "rt")
        fname, ext = os.path.splitext(fname)
        ext = ext.lower()
    if ext in (".fq", ".fastq"):
        return _fastq_records(handle or open(input_file))
    if ext in (".fa", ".fasta"):
        return _fasta_records(handle or open(input_file))
    if ext in (".bam", ".sam"):
        return _native_records(input_file, skip_secondary)
    return None


def read_chunks(records, size=READ_CHUNK_SIZE):
    chunk = []
    for rec in records:
        chunk.append(rec)
        if len(chunk) >= size:
            yield chunk
            chunk = []
    yield chunk                      # the reference also yields the trailing (possibly empty) chunk


def _ext(input_file):
    fname, ext = os.path.splitext(os.path.basename(input_file))
    if ext.lower() in (".gz", ".gzip"):
        fname, ext = os.path.splitext(fname)
    return ext.lower()


def is_fastx(input_file):
    """True for [gzipped] FASTA / FASTQ by extension (reference :80-97)"""
    return _ext(input_file) in (".fq", ".fastq", ".fa", ".fasta")


def is_native_input(input_file):
    """True for every format the reference reads (:80-97): [gzipped] FASTA / FASTQ, BAM, SAM - all parsed natively"""
    return _ext(input_file) in (".fq", ".fastq", ".fa", ".fasta", ".bam", ".sam")


def chunk_read_ids(ch):
    """read ids of an ingest chunk, in order"""
    import ctypes as C
    n = ch.n
    if not n:
        return []
    off = _native.np.ctypeslib.as_array(C.cast(ch.id_off, C.POINTER(C.c_uint64)), shape=(n + 1,)).tolist()
    o0 = off[0]
    text = C.string_at(ch.ids + o0, off[n] - o0).decode("ascii", "replace")
    return [text[off[i] - o0:off[i + 1] - o0] for i in range(n)]


def run_fastx_pipeline(input_file, detectors, on_chunk, chunk_size=None, inflate_threads=0, ids_only=False, skip_secondary=False,
                       segment_bytes=0):
    """file -> native reader threads -> pinned chunks -> GPU(s) -> native row formatter -> on_chunk(rows, recs), in
    input order, driven from Python (callers that want the rows or ids in Python; file-to-file runs use
    _native.stage1_run, which keeps everything native).  Two chunks per device are in flight (bdg_extract_submit /
    bdg_extract_collect), chunk k on device k mod N.  A chunk holds at most chunk_size reads and never spans two parse
    segments of the input, so chunks may be shorter in the middle of a large file.  ids_only: the caller wants the read
    ids and the records, not the TSV text (stage 2 from read input): on_chunk(list of ids, recs).  Returns the number of
    reads."""
    chunk_size = chunk_size or READ_CHUNK_SIZE
    ng = len(detectors)
    ing = _native.Ingest(input_file, chunk_size, ring_chunks=2 * ng + 2, inflate_threads=inflate_threads,
                         skip_secondary=skip_secondary, segment_bytes=segment_bytes)
    inflight = deque()

    def finish(item):
        det, slot, ch = item
        try:
            recs = det._ctx().extract_collect(slot, ch.n)
        except _native.BadgerHipError as e:
            if e.code == _native.E_BADBASE:
                raise KeyError(str(e))      # the reference raises KeyError in reverese_complement
            raise
        rows = chunk_read_ids(ch) if ids_only else _native.format_rows(ch, recs)[0]
        ing.release(ch)
        on_chunk(rows, recs)

    total = 0
    try:
        k = 0
        while True:
            ch = ing.next()
            if ch.n == 0:
                break
            if len(inflight) >= 2 * ng:
                finish(inflight.popleft())
            det, slot = detectors[k % ng], (k // ng) % 2
            det._ctx().extract_submit(slot, ch.bases, ch.off, ch.n, det.UMI_LEN_10X)
            inflight.append((det, slot, ch))
            k, total = k + 1, total + ch.n
        while inflight:
            finish(inflight.popleft())
    finally:
        # chunks still in flight after an error: wait for the GPU before the pinned buffers go away
        for det, slot, ch in inflight:
            try:
                det._ctx().extract_collect(slot, ch.n)
            except Exception:
                pass
        ing.close()
    return total


# ----------------------------------------------------------------------------- handlers
class FileReadHandler:
    def __init__(self, outfile):
        self.output_table = outfile
        self.output_file = open(outfile, "w")

    def add_header(self, header):
        self.output_file.write(header + "\n")

    def add_read(self, barcode_result):
        self.output_file.write(str(barcode_result) + "\n")

    def add_rows(self, rows):
        if rows:
            self.output_file.write("\n".join(rows) + "\n")

    def add_text(self, rows_bytes):
        """rows as the native formatter delivers them: one "\n"-terminated line per read"""
        if rows_bytes:
            self.output_file.write(rows_bytes.decode("ascii"))

    def dump_stats(self, read_stat):
        with open(self.output_table + ".stats", "w") as f:
            f.write(str(read_stat))

    def close(self):
        if not self.output_file.closed:
            self.output_file.close()

    def __del__(self):
        self.close()


class ListReadHandler:
    def __init__(self):
        self.read_storage = []

    def add_header(self, header):
        pass

    def add_read(self, r):
        self.read_storage.append((r.read_id, r.barcode, r.UMI))

    def add_rows(self, rows):
        for row in rows:
            f = row.split("\t")
            self.read_storage.append((f[0], f[1], f[2]))

    def add_text(self, rows_bytes):
        if rows_bytes:
            self.add_rows(rows_bytes.decode("ascii").split("\n")[:-1])

    def dump_stats(self, read_stat):
        pass


class BarcodeCaller:
    """Same seam as the reference's BarcodeCaller (reference :71-128): process_chunk() takes
    list[(read_id, seq)], feeds the handler one row per read in input order, updates read_stat."""

    def __init__(self, barcode_detector, read_handler):
        self.barcode_detector = barcode_detector
        self.read_handler = read_handler
        self.read_handler.add_header(barcode_detector.result_type().header())
        self.read_stat = ReadStats()

    def process_chunk(self, read_chunk):
        if not read_chunk:
            return
        recs = self.barcode_detector.extract_records([s for _, s in read_chunk])
        self.read_handler.add_rows([record_to_row(rid, s, r) for (rid, s), r in zip(read_chunk, recs)])
        self.read_stat.add_records(recs)

    def process(self, input_file, skip_secondary=False, threads=0):
        """every read of the file through the detector's device into the handler (reference :78-118).  The handler gets
        TSV text (add_text) or, if it says ids_only, the read ids (add_ids)."""
        logger.info("Processing " + input_file)
        if is_native_input(input_file):
            ids_only = getattr(self.read_handler, "ids_only", False)

            def on_chunk(rows, recs):
                if ids_only:
                    self.read_handler.add_ids(rows)
                else:
                    self.read_handler.add_text(rows)
                self.read_stat.add_records(recs)
            run_fastx_pipeline(input_file, [self.barcode_detector], on_chunk, ids_only=ids_only, inflate_threads=threads,
                               skip_secondary=skip_secondary)
        else:
            logger.error("Unknown file format " + input_file)
        logger.info("Finished " + input_file)


# ----------------------------------------------------------------------------- drivers
def _detectors(mode, gpus):
    gpus = max(1, gpus)
    if os.environ.get("BADGER_AMD_CONTEXTS_ON_ONE_DEVICE") == "1":
        # rehearsal of --gpus N on a one-GPU box: N independent contexts (own streams, workspaces, staging) on device 0
        return [BARCODE_CALLING_MODES[mode](device=0, instance=g) for g in range(gpus)]
    if gpus > 1:
        have = _native.device_count()
        if gpus > have:
            raise SystemExit("--gpus %d: this node shows %d device(s)" % (gpus, have))
    return [BARCODE_CALLING_MODES[mode](device=g) for g in range(gpus)]


def _stats_lines(res):
    """ReadStats.__str__ (barcode_callers.py:138-143) from the native run's counters: the attribute lines come in the order
    in which the first read showing each was met (a dict's insertion order in the reference)."""
    lines = [("Total reads", res.reads), ("Barcode detected", res.barcodes), ("Reliable UMI", 0)]
    attrs = []
    if res.polyt:
        attrs.append((res.first_polyt, 0, "PolyT detected", res.polyt))
    if res.r1:
        attrs.append((res.first_r1, 1, "R1 detected", res.r1))       # one read adds "PolyT detected" before "R1 detected"
    return lines + [(name, v) for _, _, name, v in sorted(attrs)]


def _run_native(args, header_every, threads, skip_secondary):
    if not is_native_input(args.input):
        logger.error("Unknown file format " + args.input)
        sys.exit(-1)
    detectors = _detectors(args.mode, getattr(args, "gpus", 1))
    logger.info("Barcode caller created")
    res = _native.stage1_run([d._ctx() for d in detectors], args.input, args.output, detectors[0].result_type().header(),
                             detectors[0].UMI_LEN_10X, threads=threads, header_every=header_every, skip_secondary=skip_secondary)
    timing = os.environ.get("BADGER_AMD_STAGE1_TIMING")
    if timing:                                   # where the run's time went (tools/cli_throughput.py reads it)
        import json
        with open(timing, "a") as f:
            f.write(json.dumps({k: getattr(res, k) for k, _ in res._fields_}) + "\n")
    return res


def process_single_thread(args):
    """one header on top, tab-separated .stats (reference :162-173); every SAM / BAM record is used (:110-118)"""
    logger.info("Processing " + args.input)
    res = _run_native(args, 0, 1, False)
    with open(args.output + ".stats", "w") as f:
        for k, v in _stats_lines(res):
            f.write("%s:\t%d\n" % (k, v))
            logger.info("%s:\t%d" % (k, v))
    logger.info("Finished barcode calling")


def process_in_parallel(args):
    """a header in front of every READ_CHUNK_SIZE reads (+ one for the trailing chunk) and a space-separated merged .stats:
    the reference's parallel-mode file shape (:131-159,243-259), rows in input order; secondary and supplementary SAM / BAM
    records are skipped (:144-145).  Chunk k goes to device k mod N (all N work concurrently)."""
    logger.info("Processing " + args.input)
    res = _run_native(args, READ_CHUNK_SIZE, args.threads, True)
    with open(args.output + ".stats", "w") as out_stats:
        for k, v in _stats_lines(res):
            logger.info("%s: %d" % (k, v))
            out_stats.write("%s: %d\n" % (k, v))
    logger.info("Finished barcode calling")


def extract_barcodes_single_thread(input_file, mode, device=0):
    logger.info("Extracting from " + input_file)
    handler = ListReadHandler()
    BarcodeCaller(BARCODE_CALLING_MODES[mode](device=device), handler).process(input_file)
    logger.info("Finished barcode extraction")
    return handler.read_storage


def extract_barcodes_in_parallel(input_file, mode, threads, device=0):
    logger.info("Extracting from " + input_file)
    if not is_native_input(input_file):
        logger.error("Unknown file format " + input_file)
        sys.exit(-1)
    handler = ListReadHandler()
    BarcodeCaller(BARCODE_CALLING_MODES[mode](device=device), handler).process(input_file, skip_secondary=True, threads=threads)
    logger.info("Finished barcode extraction")
    return handler.read_storage


class IdListHandler:
    """collects the read ids only (stage 2 takes the barcodes from the device records)"""

    def __init__(self):
        self.read_ids = []

    def add_header(self, header):
        pass

    def add_read(self, r):
        self.read_ids.append(r.read_id)

    def add_rows(self, rows):
        self.read_ids.extend(row.split("\t", 1)[0] for row in rows)

    def add_text(self, rows_bytes):
        if rows_bytes:
            self.read_ids.extend(line.split("\t", 1)[0] for line in rows_bytes.decode("ascii").split("\n")[:-1])

    ids_only = True              # FASTX input: the pipeline hands over the ids themselves, no TSV text is formatted

    def add_ids(self, ids):
        self.read_ids.extend(ids)

    def dump_stats(self, read_stat):
        pass


def extract_read_ids(input_file, mode, device=0, skip_secondary=False, threads=0):
    """run the extraction (records stay with the context if it keeps them) and return the read ids in order"""
    logger.info("Extracting from " + input_file)
    handler = IdListHandler()
    BarcodeCaller(BARCODE_CALLING_MODES[mode](device=device), handler).process(input_file, skip_secondary=skip_secondary, threads=threads)
    logger.info("Finished barcode extraction")
    return handler.read_ids


def set_logger(logger_instance):
    logger_instance.setLevel(logging.INFO)
    if not logger_instance.handlers:
        ch = logging.StreamHandler(sys.stdout)
        ch.setLevel(logging.INFO)
        ch.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
        logger_instance.addHandler(ch)


def parse_args(sys_argv):
    p = argparse.ArgumentParser(formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--output", "-o", type=str, help="output prefix name", required=True)
    p.add_argument("--mode", type=str, help="mode to be used", choices=BARCODE_CALLING_MODES.keys(), default="double")
    p.add_argument("--input", "-i", type=str, help="input reads in [gzipped] FASTA, FASTQ, BAM, SAM", required=True)
    p.add_argument("--threads", "-t", type=int, help="threads to use (16)", default=16)
    p.add_argument("--tmp_dir", type=str, help="folder for temporary files (unused: no temporary files are written)")
    p.add_argument("--gpus", type=int, default=1, help="number of MI355X devices of this node to shard chunks over")
    return p.parse_args(sys_argv)


def main(sys_argv):
    args = parse_args(sys_argv)
    set_logger(logger)
    if args.mode not in BARCODE_CALLING_MODES:
        raise KeyError(args.mode)          # the reference's default 'double' is not a valid key either
    if args.threads == 1:
        process_single_thread(args)
    else:
        process_in_parallel(args)


if __name__ == "__main__":
    _native.PRELOAD_TORCH = False            # nothing on this command line's path imports torch: skip its start-up cost
    try:
        main(sys.argv[1:])
        # every file is written and closed: leave without the interpreter's and the HIP runtime's tear-down (0.2 s of a run
        # that takes about a second for 12.5 M reads; tools/startup_probe.py)
        logging.shutdown()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    except SystemExit:
        raise
    except:  # noqa: E722  (same catch-all and exit code as the reference :383-391)
        print_exc()
        sys.exit(-1)
