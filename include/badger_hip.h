/*
 * badger_hip.h -- C ABI of libbadger_hip.so, the MI355X (gfx950) drop-in for the
 * data-parallel hot path of algbio/Badger.
 *
 * The reference is pure Python and has no FFI of its own; each entry point below
 * replaces a Python call site (reference-relative file:line) and is what a ctypes
 * binding inside the reference would call (INTEGRATION.md shows the stubs):
 *
 *   bdg_extract_batch*   BarcodeCaller.process_chunk -> TenXBarcodeExtractor.find_barcode_umi
 *                        extract_raw_barcodes.py:120-128, barcode_callers.py:165-229,
 *                        barcode_extraction/common.py:10-51,85-114, kmer_indexer.py:49-75
 *   bdg_graph_edges*     BarcodeGraph.graph_construction / compare_chunk
 *                        barcode_graph.py:75-111,207-249, index.py:29-35,77-93
 *   bdg_nearest16*       loop body of BarcodeGraph.postprocessing
 *                        barcode_graph.py:376-384 (argmin editdistance.eval over the centers)
 *
 * Conventions: plain pointers and sizes, little-endian integers, no exceptions
 * cross the boundary.  Return 0 = OK, <0 = error (BDG_E_*); bdg_last_error()
 * gives a message.  One bdg_ctx per device, used by one host thread at a time;
 * different contexts are independent (one per GPU, no collectives).
 * The library never keeps a caller pointer past the call.
 *
 * Two families:
 *   host-buffer calls   (bdg_extract_batch, bdg_nearest16, bdg_graph_edges):
 *       pointers are host memory; the call copies in, runs, copies out, returns
 *       when the results are in the caller's buffers.
 *   device-resident calls (*_dev): pointers are device memory on the context's
 *       device; kernels are enqueued on the context's stream (bdg_set_stream)
 *       and the call returns without synchronising unless stated.
 */
#ifndef BADGER_HIP_H
#define BADGER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDG_OK            0
#define BDG_E_ARG        -1   /* bad argument */
#define BDG_E_HIP        -2   /* HIP runtime error */
#define BDG_E_NOMEM      -3   /* device allocation failed */
#define BDG_E_CAPACITY   -4   /* output capacity too small (graph edges: see *n_edges) */
#define BDG_E_BADBASE    -5   /* a read holds a byte outside "ACGTN" (reference: KeyError,
                                 barcode_extraction/common.py:34-38) */
#define BDG_E_FORMAT     -6   /* malformed FASTA / FASTQ / SAM / BAM input (reference: ValueError from Bio.SeqIO / pysam) */
#define BDG_E_NOSEQ      -7   /* a SAM / BAM record without a sequence (reference: query_sequence is None and
                                 find_barcode_umi raises TypeError on it, barcode_callers.py:183) */

typedef struct bdg_ctx bdg_ctx;

/* One record per read: everything BarcodeCaller needs to print the TSV row
 * (barcode_callers.py:40-42,91-93) without string data leaving the device.
 * Coordinates are in the coordinates of the strand the result was taken from
 * (flags & BDG_FLAG_REV: the reverse complement of the read). */
typedef struct bdg_extract_rec {
    int32_t  polyT;      /* polyT_start column, -1 = none                          */
    int32_t  r1_end;     /* R1_end column, -1 = none                               */
    int32_t  bc_start;   /* barcode = strand_seq[bc_start : bc_start+16] (Python slice) */
    int32_t  umi_start;  /* UMI     = strand_seq[umi_start : umi_end]              */
    int32_t  umi_end;
    uint32_t bc_rank;    /* rank() of the barcode (common.py:21-25) if BDG_FLAG_RANK_OK */
    int8_t   r1_score;   /* Smith-Waterman score of the accepted R1 alignment, else 0 */
    int8_t   strand;     /* printed strand column: +1 '+', -1 '-', 0 '.'           */
    uint8_t  valid;      /* 1: barcode detected (is_valid()), 0: row prints "*"    */
    uint8_t  flags;      /* BDG_FLAG_*                                             */
    uint32_t reserved;
} bdg_extract_rec;       /* 32 bytes */

#define BDG_FLAG_REV      1u  /* result comes from reverese_complement(read)          */
#define BDG_FLAG_RANK_OK  2u  /* bc_rank is valid: 16 in-range ACGT bases             */
#define BDG_FLAG_BC16     4u  /* the barcode slice holds 16 bases (some may be N)     */
#define BDG_FLAG_INCOMPLETE 8u /* the batch overflowed an internal queue: this record (like every record of the
                                  batch) is a placeholder with valid = 0; bdg_extract_status() returns BDG_E_CAPACITY */

typedef struct bdg_edge {
    uint32_t a;          /* rank, a < b */
    uint32_t b;
    uint32_t dist;       /* min(ed(a,b), ed(a[:-1],b), ed(a,b[:-1]))  barcode_graph.py:243 */
} bdg_edge;              /* 12 bytes */

/* Per-kernel device time, collected with HIP events on the context's stream when
 * profiling is enabled (bdg_profile_enable). */
typedef struct bdg_kernel_time {
    char     name[48];
    uint64_t launches;
    double   total_ms;
} bdg_kernel_time;

/* ---- context ---------------------------------------------------------- */
int  bdg_init(int device_id, bdg_ctx** out);
void bdg_free(bdg_ctx* ctx);
const char* bdg_last_error(bdg_ctx* ctx);     /* ctx-local, valid until the next call; ctx may be NULL */
const char* bdg_version(void);
/* Number of devices this process may open (hipGetDeviceCount; 0 when there is none or the runtime fails).  What the
 * reference's "-t threads" sizing becomes for "--gpus N" (extract_raw_barcodes.py:366, :208-214). */
int  bdg_device_count(void);
/* The arithmetic of the deletion-variant joins' index entries (csrc/dj_codec.hpp: key mixing and its inverse, deleting and
 * re-inserting letters, entry encode / decode), run on the HOST on `rounds` random barcodes from `seed`: 0 when every
 * identity holds, else the number of the first check that failed.  The same functions are what the kernels compile; no GPU
 * is needed (CPU test tier).  Nothing in the reference corresponds (its buckets are Python dicts, index.py:29-35). */
int  bdg_selftest_dj_codec(uint64_t seed, uint32_t rounds);
/* Plain device buffers on the context's device, for hosts that bring no allocator of their own (the command lines of
 * this package run without torch): zero-filled allocation, release, and copies to and from host memory on the
 * context's stream (both return when the copy is done).  Callers that do have one (torch tensors, hipMalloc of their own) pass those pointers to the
 * *_dev entry points just the same. */
int  bdg_mem_alloc(bdg_ctx* ctx, uint64_t bytes, void** d_out);
int  bdg_mem_free(bdg_ctx* ctx, void* d_ptr);
int  bdg_mem_to_host(bdg_ctx* ctx, void* dst, const void* d_src, uint64_t bytes);
int  bdg_mem_from_host(bdg_ctx* ctx, void* d_dst, const void* src, uint64_t bytes);   /* returns when src may be reused */
/* Use `hip_stream` (a hipStream_t) for all later work of this context.  NULL is the device's
 * default (null) stream -- which is what torch.cuda.current_stream().cuda_stream is unless the
 * caller made its own.  Until this is called the context works on a private non-blocking stream.
 * Lets the caller order the library's kernels with its own work and time with its own events. */
int  bdg_set_stream(bdg_ctx* ctx, void* hip_stream);
int  bdg_synchronize(bdg_ctx* ctx);            /* waits for everything the context has queued (both streams, see below) */
/* Batch pipelining.  With overlap on, bdg_nearest16_recs_dev runs on an auxiliary stream of the context, ordered behind the
 * extraction that wrote d_recs - and it is not queued at once: it waits for the NEXT bdg_extract_batch_dev, which queues it
 * behind its own scan kernel, so that the whitelist match of batch i (gathers) runs beside the alignment kernels of batch
 * i + 1 (integer issue, hardly any memory traffic) and not beside the scan, which streams the reads at the memory's rate
 * and loses more to the gathers than they gain.  A match still waiting is queued by bdg_synchronize(), by the next match,
 * and before the whitelist changes.  The caller alternates between two record / result buffers (an extraction waits for
 * the match queued last, i.e. it never overwrites records a match has yet to read); results of a match are complete
 * after bdg_synchronize() - not after a device-wide synchronisation alone, which does not know about a waiting match. */
int  bdg_set_overlap(bdg_ctx* ctx, int on);
int  bdg_profile_enable(bdg_ctx* ctx, int on);
/* Time only the kernel of this name (NULL or "": every kernel again).  A pair of events around a kernel costs a few
 * microseconds and keeps the next launch from being queued behind it early; timing all eight kernels of a step adds ~6 % to
 * the step, timing one adds nothing measurable (bench.py times the dominant kernel in its timed region, all of them in a
 * separate pass). */
int  bdg_profile_only(bdg_ctx* ctx, const char* kernel);
int  bdg_profile_reset(bdg_ctx* ctx);
/* Synchronises, then writes up to cap entries; returns the number of kernels known. */
int  bdg_profile_read(bdg_ctx* ctx, bdg_kernel_time* out, int cap);

/* ---- B-E: barcode extraction ------------------------------------------ */
/* n reads as one concatenated ASCII buffer + n+1 offsets (off[0]=0 not required,
 * off non-decreasing).  umi_len 10 (tenX_v2) or 12 (tenX_v3), barcode_callers.py:156. */
int  bdg_extract_batch(bdg_ctx* ctx, const uint8_t* bases, const uint64_t* off, uint32_t n,
                       uint32_t umi_len, bdg_extract_rec* out);
/* Device-resident form.  d_bases must be 16-byte aligned and readable up to
 * total_bytes rounded up to 16 (any hipMalloc/torch allocation is).  Asynchronous;
 * a bad base is reported by the next bdg_extract_status().
 * Queue overflow: the alignment candidates of a batch pass through internal queues sized from the batch's byte
 * count.  If one overflows (adapter-dense input), EVERY record of the batch is written as a placeholder
 * {valid 0, flags BDG_FLAG_INCOMPLETE}: later device-side consumers of d_out (bdg_nearest16_recs_dev,
 * bdg_distinct_dev) then find nothing to use instead of half-aligned records, bdg_extract_status() returns
 * BDG_E_CAPACITY and has grown the workspace; call bdg_extract_batch_dev again (loop while BDG_E_CAPACITY:
 * the second pass measures what the first could not; bdg_extract_batch does exactly that). */
int  bdg_extract_batch_dev(bdg_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_off, uint32_t n,
                           uint64_t total_bytes, uint32_t umi_len, bdg_extract_rec* d_out);
/* Synchronises and returns BDG_OK, BDG_E_BADBASE (read index in *bad_read) or
 * BDG_E_CAPACITY (see above). n_windows: Smith-Waterman windows evaluated. */
int  bdg_extract_status(bdg_ctx* ctx, uint64_t* bad_read, uint64_t* n_windows);
/* Entries per segment of the internal queues for the next launches (16 bytes each, 3 queues x 8 segments);
 * 0 = automatic (total_bytes / 48 per queue).  An overflow still grows it.  For callers that must bound the
 * workspace, and for tests of the overflow path. */
int  bdg_extract_set_queue_capacity(bdg_ctx* ctx, uint64_t entries_per_segment);
/* Which of the reference's two strand rules picks a read's result from its forward and reverse-complement results:
 * BDG_STRAND_RULE_DEFAULT    TenXBarcodeExtractor.find_barcode_umi (barcode_callers.py:165-179): both valid -> the higher
 *                            r1_score, ties to the reverse strand; one valid -> that one; none -> the forward result;
 * BDG_STRAND_RULE_NO_POLYA   TenXBarcodeExtractor.find_barcode_umi_no_polya (barcode_callers.py:231-248): the forward
 *                            result if valid, else the reverse one if valid, else the more informative of the two (neither
 *                            carries a score then, which leaves the reverse result).
 * Holds for the launches that follow.  The reference's second rule forms the reverse complement only when the forward
 * result is invalid, so a byte outside 'ACGTN' raises there only then; this library reports BDG_E_BADBASE for every
 * read holding one, under either rule. */
#define BDG_STRAND_RULE_DEFAULT  0
#define BDG_STRAND_RULE_NO_POLYA 1
int  bdg_extract_set_strand_rule(bdg_ctx* ctx, int rule);
/* Pipeline statistics of the last extraction (synchronises): out[0] 6-mer hits, [1] clusters aligned
 * (queue A), [2] hits sent to the strict filter (queue B), [3] of those skipped because the
 * relaxed search had already succeeded, [4] filter survivors, [5] hits re-queued from clusters,
 * [6] alignments run, [7] clusters the hits of [2] arrived in. */
int  bdg_extract_counters(bdg_ctx* ctx, uint64_t out[8]);

/* Pipelined form of bdg_extract_batch for a stream of chunks (extract_raw_barcodes.py:131-159: chunks of 100,000
 * reads): submit() enqueues the H2D copy of a chunk (pinned host memory makes it asynchronous), the kernels and the
 * D2H copy of the records on the context's stream and returns; collect() waits for that chunk, reruns it if a queue
 * overflowed, and hands over its records.  `slot` (0 .. BDG_SLOTS-1) names one of the context's staging sets: a
 * chunk may be submitted to a free slot while earlier ones are still in flight, so the device works on chunk k+1
 * while the host formats chunk k.  Collect in submission order.  bases / off must stay valid until collect(). */
#define BDG_SLOTS 4
int  bdg_extract_submit(bdg_ctx* ctx, uint32_t slot, const uint8_t* bases, const uint64_t* off, uint32_t n, uint32_t umi_len);
int  bdg_extract_collect(bdg_ctx* ctx, uint32_t slot, bdg_extract_rec* out);

/* Stage-1 -> stage-2 hand-off on the device (badger.py:112-121 extracts and then builds the graph in one process):
 * while `on`, bdg_extract_collect also appends each chunk's records to a device-side array, in collection order.
 * bdg_kept_records gives that array (device pointer, valid until the next collect / keep_records call) for
 * bdg_distinct_dev -> bdg_graph_edges_dev, so the barcodes never pass through host strings.  Turning it on starts
 * an empty array; turning it off frees it. */
int  bdg_extract_keep_records(bdg_ctx* ctx, int on);
int  bdg_kept_records(bdg_ctx* ctx, const bdg_extract_rec** d_recs, uint64_t* n);
/* The same hand-off for barcodes that come out of a stage-1 TSV (badger.py:91-111, bdg_import_stage1_tsv): n reads, rank[i]
 * the rank of read i's barcode where usable[i] != 0 (host arrays).  They become the kept records (valid, bc_rank, flags; the
 * other fields empty), replacing what was kept before, so that bdg_distinct_dev, the edge build, bdg_cluster_dev and
 * bdg_assign_reads_dev serve the TSV route as they serve read input.  Synchronises. */
int  bdg_keep_observed(bdg_ctx* ctx, const uint32_t* rank, const uint8_t* usable, uint64_t n);
/* The kept records copied to host memory (the first min(n, cap) of them); synchronises.  output_file
 * (barcode_graph.py:388-410) needs every read's observed barcode once more, as a rank. */
int  bdg_kept_records_to_host(bdg_ctx* ctx, bdg_extract_rec* out, uint64_t cap);

/* ---- read ingest and row output (host side) --------------------------------------------- */
/* [gzipped / BGZF] FASTA / FASTQ / SAM and BAM -> chunks of at most chunk_reads reads {concatenated bases, offsets, ids}
 * in pinned host memory (pinned = 0: pageable, for hosts without a GPU), in file order.  Replaces the reference's record
 * loops over Bio.SeqIO.parse / pysam.AlignmentFile (extract_raw_barcodes.py:78-118,131-150).  Format by extension like the
 * reference (:80-97,181-197): .fa .fasta .fq .fastq .sam .bam, optionally + .gz / .gzip; anything else returns BDG_E_ARG.
 * Record semantics: Bio.SeqIO's (id = first word of the header; FASTA sequence = its lines joined; FASTQ = four-line
 * records) and pysam's (query_name, query_sequence).  The text is parsed by several threads, one segment of the input
 * each (csrc/ingest.cpp says how and why the result equals a one-thread parse); a chunk never spans two segments, so a
 * chunk may be shorter than chunk_reads in the middle of a large file. */
typedef struct bdg_ingest bdg_ingest;
typedef struct bdg_ingest_chunk {
    uint32_t        id;           /* for bdg_ingest_release */
    uint32_t        n;            /* reads in the chunk; 0 = end of input */
    const uint8_t*  bases;        /* concatenated ASCII; read i is bases[off[i] .. off[i+1]); 64 readable bytes behind the end */
    const uint64_t* off;          /* n + 1 offsets into bases (off[0] is 0 only for the first chunk cut from a segment) */
    uint64_t        total_bytes;  /* off[n] - off[0] */
    const char*     ids;          /* concatenated read ids; id i is ids[id_off[i] .. id_off[i+1]) */
    const uint64_t* id_off;       /* n + 1 offsets into ids */
} bdg_ingest_chunk;
typedef struct bdg_ingest_opts {
    uint32_t chunk_reads;         /* reads per chunk at most (the reference's READ_CHUNK_SIZE = 100000) */
    uint32_t ring_chunks;         /* chunks the caller may hold at once (taken and not yet released), >= 2 */
    int32_t  pinned;              /* bases in pinned host memory (hipHostMalloc) */
    uint32_t threads;             /* threads that inflate and parse: 0 = min(12, cores); 1 = one, and every compressed input is read as
                                     the sequential gzip stream it is for gzip.open in the reference (extract_raw_barcodes.py:86-87) */
    uint64_t segment_bytes;       /* text per parse segment (0 = 16 MiB: measured best of 16 / 24 / 32 / 64 / 128 end to end) */
    int32_t  skip_secondary;      /* SAM / BAM: drop secondary and supplementary records (flag 0x100 / 0x800) like the reference's
                                     chunk reader (:144-145); its single-thread loop keeps them (:110-118) */
    uint32_t reserved;
} bdg_ingest_opts;
int  bdg_ingest_open(const char* path, uint32_t chunk_reads, uint32_t ring_chunks, int pinned, bdg_ingest** out);
/* The same with the number of reader threads stated (bdg_ingest_opts.threads).  What the reference's "-t threads" buys on
 * the input side. */
int  bdg_ingest_open_mt(const char* path, uint32_t chunk_reads, uint32_t ring_chunks, int pinned, uint32_t threads,
                        bdg_ingest** out);
int  bdg_ingest_open_ex(const char* path, const bdg_ingest_opts* opts, bdg_ingest** out);
/* Blocks until the next chunk is parsed.  The chunk's memory stays untouched until bdg_ingest_release(id); at most
 * ring_chunks chunks can be held.  BDG_E_FORMAT: malformed record (bdg_ingest_error says where; the chunks in front of it
 * have been delivered); BDG_E_NOSEQ: a SAM / BAM record without a sequence. */
int  bdg_ingest_next(bdg_ingest* g, bdg_ingest_chunk* out);
int  bdg_ingest_release(bdg_ingest* g, uint32_t id);
const char* bdg_ingest_error(bdg_ingest* g);
uint64_t bdg_ingest_reads(bdg_ingest* g);      /* reads in the chunks made so far (all of them once bdg_ingest_next has returned n = 0) */
void bdg_ingest_close(bdg_ingest* g);
/* TSV rows of a chunk (TenXBarcodeDetectionResult.__str__, barcode_callers.py:40-42,91-93,117-119), one line per read,
 * "\n"-terminated, into out[cap].  Returns the bytes written, or the bytes needed if cap is too small (nothing
 * written then; call with out = NULL to size), or < 0.  counts (may be NULL): reads, barcodes detected, polyT
 * detected, R1 detected (ReadStats, barcode_callers.py:122-143). */
int64_t bdg_format_rows(const bdg_ingest_chunk* chunk, const bdg_extract_rec* recs, char* out, uint64_t cap, uint64_t counts[4]);

/* Stage 1 from file to file in native threads: readers -> GPU(s) -> row formatters -> one writer, rows in input order
 * (extract_raw_barcodes.py:162-173 process_single_thread, :176-261 process_in_parallel).  Chunk k goes to context k mod
 * n_ctx, two chunks in flight per context.  header: the column line without its newline.  Returns BDG_E_BADBASE (reference:
 * KeyError), BDG_E_FORMAT (ValueError), BDG_E_NOSEQ (TypeError) with the rows of the chunks in front of the failure
 * written, like the reference's loop; the message is bdg_last_error(ctxs[0]). */
typedef struct bdg_stage1_opts {
    uint32_t umi_len;             /* 10 (tenX_v2) or 12 (tenX_v3) */
    uint32_t threads;             /* reader threads (bdg_ingest_opts.threads) */
    uint32_t format_threads;      /* 0 = 4 */
    uint32_t header_every;        /* 0: the header once, on top (the reference's single-thread file shape); N: in front of every N
                                     reads and once more when the input ends on a multiple of N - the reference's parallel shape,
                                     one header per READ_CHUNK_SIZE chunk including the trailing empty one (:131-150,243-246) */
    uint32_t chunk_reads;         /* reads per GPU batch at most (0 = 100000) */
    int32_t  skip_secondary;      /* bdg_ingest_opts.skip_secondary */
    uint64_t segment_bytes;       /* bdg_ingest_opts.segment_bytes */
} bdg_stage1_opts;
typedef struct bdg_stage1_result {
    uint64_t reads, barcodes, polyt, r1;      /* ReadStats: total, barcode detected, polyT detected, R1 detected */
    uint64_t first_polyt, first_r1;           /* index of the first read showing each attribute (~0: none): the order of the .stats lines */
    uint64_t bad_read;                        /* BDG_E_BADBASE: index of the read, ~0 if unknown */
    uint64_t chunks, out_bytes;
    double   seconds_total;
    double   seconds_wait_parse;              /* this thread waiting for the readers */
    double   seconds_submit;                  /* ... queueing copies and kernels */
    double   seconds_wait_gpu;                /* ... waiting for a chunk's records */
    double   seconds_wait_format;             /* ... waiting for the formatters / the writer to take a chunk */
    double   seconds_format, seconds_write;   /* busy time of the formatter threads (summed) / of the writer */
} bdg_stage1_result;
int  bdg_stage1_run(bdg_ctx* const* ctxs, uint32_t n_ctx, const char* in_path, const char* out_path, const char* header,
                    const bdg_stage1_opts* opts, bdg_stage1_result* res);

/* ---- B-N: nearest whitelist barcode ----------------------------------- */
/* Per query: the whitelist entry with the smallest Levenshtein distance (ties ->
 * lowest whitelist index), that distance, and how many entries share it.  If the
 * distance exceeds max_ed: best_idx 0xFFFFFFFF, best_ed 0xFF, n_ties 0.
 * Queries and whitelist are rank-packed 16-mers (common.py:21-25). */
int  bdg_nearest16(bdg_ctx* ctx, const uint32_t* q, uint32_t nq, const uint32_t* wl, uint32_t nw,
                   uint32_t max_ed, uint32_t* best_idx, uint8_t* best_ed, uint16_t* n_ties);
/* Copy a whitelist (host memory, any order, distinct) to the device and build its
 * lookup index; kept in the context until replaced.  Indices reported later refer
 * to the caller's order. */
int  bdg_whitelist_load(bdg_ctx* ctx, const uint32_t* wl, uint32_t nw);
int  bdg_nearest16_dev(bdg_ctx* ctx, const uint32_t* d_q, uint32_t nq, uint32_t max_ed,
                       uint32_t* d_best_idx, uint8_t* d_best_ed, uint16_t* d_n_ties);
/* The same for the barcodes of a batch of extraction records, straight from bdg_extract_batch_dev's output (query i =
 * d_recs[i].bc_rank): the per-read step of the pipeline without a gather in between.  A record without
 * BDG_FLAG_RANK_OK (invalid read, or a barcode holding N) reports idx 0xFFFFFFFF, ed 255, ties 0. */
int  bdg_nearest16_recs_dev(bdg_ctx* ctx, const bdg_extract_rec* d_recs, uint32_t n, uint32_t max_ed,
                            uint32_t* d_best_idx, uint8_t* d_best_ed, uint16_t* d_n_ties);
/* algorithm: 0 = automatic, 1 = force the exhaustive Myers scan, 2 = force the
 * neighbourhood-probe path (max_ed <= 2 only). Results are identical. */
int  bdg_nearest16_set_algo(bdg_ctx* ctx, int algo);
/* Device memory held by the neighbourhood-probe index of the loaded whitelist, in bytes: 0 until a call takes the probe path
 * (automatic mode takes the exhaustive scan while nw * nq stays small, e.g. stage 2's --high_sens pass against ~5,000 centres:
 * no index is ever built then); the deletion-variant part is added by the first probe call with max_ed = 2. */
uint64_t bdg_nearest16_index_bytes(bdg_ctx* ctx);

/* ---- B-G: edit-distance graph ----------------------------------------- */
/* ranks: distinct rank-packed 16-mers, any order.  Writes up to cap edges (a<b)
 * with S(a,b) >= qgram_T (index.py:77-93) and dist <= thr, sorted by (a,b); the
 * total found goes to *n_edges.  Returns BDG_E_CAPACITY if *n_edges > cap (the
 * first cap edges in sorted order are still written). */
int  bdg_graph_edges(bdg_ctx* ctx, const uint32_t* ranks, uint32_t n, uint32_t thr, int32_t qgram_T,
                     bdg_edge* out, uint64_t cap, uint64_t* n_edges);
/* Device-resident form: d_ranks sorted ascending and distinct; edges are written
 * unsorted; *d_n_edges (device, 8 bytes) receives the total.  Asynchronous. */
int  bdg_graph_edges_dev(bdg_ctx* ctx, const uint32_t* d_ranks, uint32_t n, uint32_t thr, int32_t qgram_T,
                         bdg_edge* d_out, uint64_t cap, uint64_t* d_n_edges);
/* Row block of the same computation, for sharding over GPUs (the reference splits the rows over
 * processes the same way, barcode_graph.py:177-190 compare_chunk): only edges (a, b), a < b, whose smaller rank a is
 * d_ranks[i] with row_begin <= i < row_end are produced.  The row blocks of a partition of [0, n) give disjoint edge
 * lists whose union is the full list. */
int  bdg_graph_edges_rows_dev(bdg_ctx* ctx, const uint32_t* d_ranks, uint32_t n, uint32_t row_begin, uint32_t row_end,
                              uint32_t thr, int32_t qgram_T, bdg_edge* d_out, uint64_t cap, uint64_t* d_n_edges);
/* 0 automatic (thr 1: neighbourhood probes below 100,000 rows, the one-deletion join from there on; thr 2: deletion-variant
 * join from 10,000 rows on, the q-gram join below; thr >= 3: q-gram join), 1 all-pairs scan,
 * 2 neighbourhood probes (thr = 1 only), 3 q-gram join (the device form of QGramIndex, index.py:29-35,77-93; any thr),
 * 4 the same with every candidate verified in closed form (the join's fallback for slices its table cannot take; for tests),
 * 5 deletion-variant join (thr <= 2: rows that share a 14-mer left by two deletions meet; same dmin and S tests; work linear
 * in n where the q-gram join's is quadratic; any n - a large input is taken in rounds over shares of the 14-mers; a single
 * round never waits for the host), 6 the same over the 15-mers left by one deletion (thr <= 1).  All give identical edge lists. */
int  bdg_graph_set_algo(bdg_ctx* ctx, int algo);
/* What the device-resident graph calls cannot return because they do not wait: waits for the context's stream, then
 * BDG_OK, or BDG_E_CAPACITY when a deletion-variant join of the last call met an input it could not group (more index
 * entries in one round than 32 bits address - only possible beyond 35 M rows; a bucket of variants beyond every split).
 * The edge list of such a call is incomplete.  bdg_graph_edges asks by itself.  Same contract as bdg_extract_status for
 * the extraction (the reference has no such state: compare_chunk, barcode_graph.py:75-111, raises where it fails). */
int  bdg_graph_status(bdg_ctx* ctx);
/* One of nparts disjoint shares of the edge list (compare_in_parallel's fan-out, barcode_graph.py:164-189, over GPUs: every
 * device holds the whole sorted array and calls this with its own part; the union over the parts is the list of
 * bdg_graph_edges_dev).  Which edges a part holds is the library's choice, made so that the parts cost the same: blocks of
 * rows for the paths that work row by row (equal rows for the probes, equal pair counts for the q-gram join and the scan:
 * what bdg_graph_edges_rows_dev takes explicitly), shares of the 14-mer / 15-mer groups for the deletion-variant joins. */
int  bdg_graph_edges_part_dev(bdg_ctx* ctx, const uint32_t* d_ranks, uint32_t n, uint32_t part, uint32_t nparts,
                              uint32_t thr, int32_t qgram_T, bdg_edge* d_out, uint64_t cap, uint64_t* d_n_edges);
/* Distinct-barcode counting of a batch on the device (BarcodeGraph.index_bc_single_thread,
 * barcode_graph.py:192-204): from n extraction records, the distinct barcodes of records with a
 * full 16-base ACGT barcode, ascending in d_uniq, with their multiplicity (d_count) and the index of
 * the first record showing them (d_first; sorting by it gives the reference's counts order).  The
 * three outputs need room for n entries.  d_n[0] = number of distinct barcodes, d_n[1] = records
 * whose 16-base barcode holds a non-ACGT base (the reference raises KeyError on those, common.py:21-25).
 * Asynchronous. */
int  bdg_distinct_dev(bdg_ctx* ctx, const bdg_extract_rec* d_recs, uint32_t n,
                      uint32_t* d_uniq, uint32_t* d_count, uint32_t* d_first, uint32_t* d_n);
/* d_rows[i] = position of d_values[i * stride_words] in the ascending array d_sorted[0..n), 0xFFFFFFFF when it is not
 * there (asynchronous on the context's stream).  With d_values = the edge array of bdg_graph_edges_dev and stride 3 this
 * turns the edges' ranks (the reference keys `edges` by rank, barcode_graph.py:245-247) into indices of the distinct
 * barcode arrays of bdg_distinct_dev, which is what clustering on arrays needs (offset 0: a, offset 1: b). */
int  bdg_rows_of_dev(bdg_ctx* ctx, const uint32_t* d_sorted, uint32_t n, const uint32_t* d_values, uint64_t m,
                     uint32_t stride_words, uint32_t* d_rows);

/* The two clustering levels of BarcodeGraph.cluster (barcode_graph.py:279-301) on the device.  d_ea / d_eb: the m edges
 * as positions in the distinct-barcode array of nu entries (bdg_rows_of_dev).  d_owner [nu], in: c at every centre's own
 * position c, -2 elsewhere; out: the position of the centre a barcode belongs to, -1 where two centres met on one level
 * (the reference's (-1, -1)), -2 unclustered.  Asynchronous. */
int  bdg_cluster_dev(bdg_ctx* ctx, const uint32_t* d_ea, const uint32_t* d_eb, uint64_t m, uint32_t nu, int32_t* d_owner);
/* Per read of d_recs: the barcode its observed barcode was corrected to (assign_by_cluster + the loop of output_file,
 * barcode_graph.py:322-329,388-410): d_uniq [nu] ascending distinct barcodes, d_assigned [nu] / d_has [nu] what each was
 * assigned to (has = 0: nothing, the row prints '*').  Asynchronous. */
int  bdg_assign_reads_dev(bdg_ctx* ctx, const bdg_extract_rec* d_recs, uint64_t n, const uint32_t* d_uniq, uint32_t nu,
                          const uint32_t* d_assigned, const uint8_t* d_has, uint32_t* d_out_rank, uint8_t* d_out_has);
/* How many of the nu distinct barcodes appear in the m edges given as positions (d_ea, d_eb: what bdg_rows_of_dev made of
 * the edge array) or in d_extra (n_extra positions: the centres that were observed): the barcodes that are keys of the
 * reference's `edges` dict when badger.py prints len(counts) - len(edges) (:131-132) - without the positions leaving the
 * device.  Synchronises; *count is a host variable. */
int  bdg_touched_count_dev(bdg_ctx* ctx, const uint32_t* d_ea, const uint32_t* d_eb, uint64_t m, uint32_t nu,
                           const uint32_t* d_extra, uint32_t n_extra, uint64_t* count);

/* ---- stage 2's read-side plumbing on the host (badger.py:112-121,129; barcode_graph.py:388-410) -------------- */
/* Read ids of a run, kept natively (12 bytes per read instead of a Python string each). */
typedef struct bdg_idstore bdg_idstore;
bdg_idstore* bdg_idstore_new(void);
void     bdg_idstore_free(bdg_idstore* s);
uint64_t bdg_idstore_count(const bdg_idstore* s);
/* n ids as one concatenated buffer + n + 1 offsets (off[0] need not be 0) */
int      bdg_idstore_append(bdg_idstore* s, const char* ids, const uint64_t* off, uint64_t n);
/* id i: pointer into the store (valid until the next append) and its length */
int      bdg_idstore_get(const bdg_idstore* s, uint64_t i, const char** p, uint32_t* len);
/* Stage 1 without the TSV: every read of in_path through the context (its records stay on the device if the context keeps
 * them, bdg_extract_keep_records), the read ids into `ids`.  What badger.py does with read input (:112-117).  opts as for
 * bdg_stage1_run (header_every / format_threads unused); res->reads = reads seen. */
int  bdg_stage1_collect(bdg_ctx* ctx, const char* in_path, const bdg_stage1_opts* opts, bdg_idstore* ids, bdg_stage1_result* res);
/* A stage-1 TSV as badger.py reads it (:91-111): the read ids into `ids`; per read the rank of its observed barcode (a
 * barcode of bc_len + 1 letters loses the last) and whether it has one of bc_len letters.  *rank / *usable are malloc'd
 * arrays of *n entries: release with bdg_host_free.  BDG_E_FORMAT: no "#read_id" / "barcode" column; BDG_E_BADBASE: a
 * usable barcode holds a letter outside ACGT (reference: KeyError from rank()), *bad_line = its line. */
int  bdg_import_stage1_tsv(const char* path, uint32_t bc_len, bdg_idstore* ids, uint32_t** rank, uint8_t** usable, uint64_t* n, uint64_t* bad_line);
void bdg_host_free(void* p);
/* "<readID>\t<barcode>\n" per read under the header "readID\tbarcode" (output_file, barcode_graph.py:406-410): rank[i]
 * spelled out (common.py:27-38) where has[i] != 0, '*' elsewhere.  n must equal the store's count. */
int  bdg_write_assignments(const bdg_idstore* ids, const uint32_t* rank, const uint8_t* has, uint64_t n, const char* path);

#ifdef __cplusplus
}
#endif
#endif
