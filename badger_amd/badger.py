#!/usr/bin/env python3
"""Stage 2 CLI: barcode correction, same flags and output file as the reference's badger.py
(reference badger.py:23-47,62-132), with the edit-distance graph built on the MI355X.

    python -m badger_amd.badger -r out.tsv -d tenX_v3 -l whitelist.txt -c 5000 [-t 1] [-hs]

--stats and --ground_truth drive the reference's offline evaluation module (stats.py), which
is outside the accelerated path; the flags are accepted and rejected with a message.
"""
import argparse
import logging
import os
import sys
from io import StringIO
from traceback import print_exc

from . import _native
from .barcode_graph import BarcodeGraph
from .extract_raw_barcodes import BARCODE_CALLING_MODES, is_native_input

logger = logging.getLogger("BarcodeGraph")


def parse_args(args):
    p = argparse.ArgumentParser(formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--threshold", "-t", help="Maximal accepted difference between barcodes", type=int, dest="threshold", default=1)
    p.add_argument("--reads", "-r", help="read in FASTQ/FASTA (can be gzipped), BAM or TSV from barcode extraction",
                   type=str, dest="reads", required=True)
    p.add_argument("--ground_truth", type=str, default=None,
                   help="File connecting each observed barcode to its read ID containing true barcode, only used for statistics")
    p.add_argument("--barcode_list", "-l", type=str, dest="barcode_list", default=None,
                   help="List of all possible barcodes for the used method, helps identify correct barcodes")
    p.add_argument("--data_type", "-d", choices=BARCODE_CALLING_MODES.keys(), type=str,
                   help="Type of single cell sequencing data in the input")
    p.add_argument("--true_barcodes", type=str, default=None,
                   help="List of all true barcodes of the input data, for example obtained from short read data")
    p.add_argument("--n_cells", "-c", help="expected number of cell associated barcodes", type=int, default=5000)
    p.add_argument("--output", "-o", help="File prefix for output files", type=str, default="OUT")
    p.add_argument("--interval", "-i", default=25, type=int,
                   help="Percentage by which the number of cells is allowed to differ from estimated cell number, default 25%%")
    p.add_argument("--stats", "-s", action="store_true", default=False,
                   help="if set, true barcode statistics are run instead of barcode calling.")
    p.add_argument("--threads", "-tr", dest="threads", default=1, type=int)
    p.add_argument("--high_sens", "-hs", action="store_true", default=False,
                   help="if set, Badger is run in high sensitivity mode. This increases recall but decreases precision")
    p.add_argument("--device", type=int, default=0, help="MI355X device index")
    p.add_argument("--gpus", type=int, default=0,
                   help="devices the edge build is shared over (one share of the edge list each, no exchange between them); "
                        "default: as many as --threads asks for and the node has (the reference's -tr N fans compare_chunk "
                        "out over N processes, barcode_graph.py:164-189), at least 1")
    return p.parse_args(args)


def edge_build_gpus(args):
    """--gpus N, or -tr N mapped onto the devices that exist"""
    if args.gpus > 0:
        return args.gpus
    if args.threads > 1:
        if os.environ.get("BADGER_AMD_CONTEXTS_ON_ONE_DEVICE") == "1":
            return args.threads
        return max(1, min(args.threads, _native.device_count()))
    return 1


def set_logger(logger_instance):
    logger_instance.setLevel(logging.INFO)
    if not logger_instance.handlers:
        h = logging.StreamHandler(stream=sys.stdout)
        h.setLevel(logging.INFO)
        h.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
        logger_instance.addHandler(h)
    logger_instance.info("Starting")


_PANDAS_NA = ("", "NA", "NaN", "nan", "N/A", "NULL", "null", "None")


def import_tsv(path, bc_len):
    """Stage-1 TSV -> (read_assignment, barcodes) the way reference badger.py:91-111 reads it with pandas.read_csv:
    repeated header rows are skipped, a missing or empty barcode field counts as '*' (a row that ends before the barcode
    column stays a read), blank lines are skipped, a field in double quotes loses them, an id pandas takes for a missing
    value becomes the empty string its to_csv writes; 17-character barcodes lose their last base in read_assignment
    (graph_construction trims its own copy).  The command line uses the native form (bdg_import_stage1_tsv); this is its
    checker and the host-side API."""
    read_assignment, barcodes = [], []
    with open(path) as f:
        first = f.readline()
        if not first:
            raise ValueError("%s is empty" % path)
        header = first.rstrip("\n").rstrip("\r").split("\t")
        ci, cb = header.index("#read_id"), header.index("barcode")
        for line in f:
            line = line.rstrip("\n").rstrip("\r")
            if not line:
                continue
            fields = [x[1:-1] if len(x) >= 2 and x[0] == x[-1] == '"' else x for x in line.split("\t")]
            rid = fields[ci] if ci < len(fields) else ""
            bc = fields[cb] if cb < len(fields) else "*"
            if rid in _PANDAS_NA:
                rid = ""
            if rid == "#read_id" or bc == "barcode":
                continue
            if bc in _PANDAS_NA:                          # pandas reads these as missing
                bc = "*"
            if bc != "*":
                barcodes.append(bc)
            read_assignment.append((rid, bc[:-1] if len(bc) == bc_len + 1 else bc))
    return read_assignment, barcodes


def load_true_barcodes(path):
    """reference badger.py:73-80"""
    vals = [l.rstrip("\n").split("\t")[0] for l in open(path) if l.strip()]
    if vals and vals[0][-1] == "1":
        vals = [v[:-2] for v in vals]
    return set(vals)


def main(args):
    import time
    t_marks = [("start", time.perf_counter())]

    def mark(name):
        t_marks.append((name, time.perf_counter()))

    args = parse_args(args)
    set_logger(logger)
    if args.data_type and args.data_type.startswith("tenX"):
        bc_len = 16
    else:
        logger.error("Please specify the type of single cell data used. Options are tenX_v2 and tenX_v3.")
        sys.exit(-3)
    if args.stats or args.ground_truth is not None:
        logger.error("--stats / --ground_truth run the reference's offline evaluation module, which this build does not carry")
        sys.exit(-4)
    # the device context comes up (0.15 - 0.3 s of runtime start) while this thread reads the barcode lists
    import threading
    def _warm():
        try:
            _native.default_context(args.device)
        except Exception:
            pass                                  # (it shows again, as the exception it is, at the first real use)
    warm = threading.Thread(target=_warm, daemon=True)
    warm.start()
    true_barcodes = load_true_barcodes(args.true_barcodes) if args.true_barcodes else None
    barcode_list = None
    if args.barcode_list:
        from .common import BarcodeRanks
        barcode_list = BarcodeRanks.from_file(args.barcode_list, bc_len)      # (the reference keeps a set of the lines, :82-88)

    from .stage2 import Stage2, observed_from_strings
    warm.join()
    st2 = Stage2(args.threshold, device=args.device)
    if args.reads.endswith("tsv"):
        read_ids, obs_rank, usable = _native.import_stage1_tsv(args.reads, bc_len)      # (import_tsv below, natively)
        logger.info("Imported barcodes from file")
        logger.info("Initializing Graph")
        # the observed barcodes go to the device as records (bdg_keep_observed): from here on the TSV route is the route of
        # read input - counting, edges, clustering and the per-read assignment run there
        ctx = _native.default_context(args.device)
        ctx.keep_observed(obs_rank, usable)
        mark("import")
        st2.count_device(ctx)
        mark("count")
        st2.build_edges(ctx, on_device=True, gpus=edge_build_gpus(args))
        from_device = ctx
    elif is_native_input(args.reads):
        # FASTA / FASTQ / SAM / BAM: the records of every chunk stay on the device (stage 1 -> stage 2 hand-off without host
        # strings): counting and the edge build run there, the host only gets the per-read ranks for the output file.
        # Like the reference (:112-117) one thread keeps every SAM / BAM record, several skip secondary / supplementary ones.
        ctx = _native.default_context(args.device)
        ctx.extract_keep_records(True)
        logger.info("Extracting from " + args.reads)
        read_ids = _native.IdStore()
        umi_len = BARCODE_CALLING_MODES[args.data_type](device=args.device).UMI_LEN_10X
        try:
            # (-tr 1 is one sequential reader, compressed input as one gzip stream - the reference's single-thread shape,
            # as extract_raw_barcodes.process_single_thread asks for it)
            _native.stage1_collect(ctx, args.reads, umi_len, read_ids, threads=args.threads,
                                   skip_secondary=args.threads != 1)
        except BaseException:
            ctx.extract_keep_records(False)
            raise
        mark("extract")
        logger.info("Finished barcode extraction")
        logger.info("Initializing Graph")
        st2.count_device(ctx)
        mark("count")
        st2.build_edges(ctx, on_device=True, gpus=edge_build_gpus(args))
        from_device = ctx
    else:
        logger.error("Unknown file format " + args.reads)
        sys.exit(-1)
    mark("reads_and_graph")
    logger.info("Graph construction done")
    st2.cluster(true_barcodes, barcode_list, args.n_cells, bc_len, args.interval)
    mark("cluster")
    logger.info("Clustering done")
    st2.output_file_from_device(read_ids, from_device, args.output, args.high_sens)
    from_device.extract_keep_records(False)
    disconnected = st2.disconnected()  # (counted where the edges are, before they are given back)
    st2.release_device()
    mark("output")
    print(disconnected)                # "disconnected" count (reference :131-132)
    timing = os.environ.get("BADGER_AMD_STAGE2_TIMING")
    if timing:                                   # where the run's time went (tools/stage2_throughput.py reads it)
        import json
        with open(timing, "a") as f:
            f.write(json.dumps({b[0]: round(b[1] - a[1], 4) for a, b in zip(t_marks, t_marks[1:])}) + "\n")


if __name__ == "__main__":
    _native.PRELOAD_TORCH = False            # this command line allocates through the library (bdg_mem_alloc): no torch start-up
    try:
        main(sys.argv[1:])
        logging.shutdown()                   # (files are closed: skip the interpreter's and the HIP runtime's tear-down)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    except (SystemExit, KeyboardInterrupt):
        raise
    except:  # noqa: E722  (same catch-all as the reference :177-196)
        if logger.handlers:
            buf = StringIO()
            print_exc(file=buf)
            logger.critical("Barcode Graph failed" + buf.getvalue())
        else:
            sys.stderr.write("Barcode Graph failed")
            print_exc()
        sys.exit(-1)
