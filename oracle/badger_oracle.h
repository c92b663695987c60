/*
 * badger_oracle.h -- CPU restatement of the algbio/Badger barcode-calling hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker for the HIP
 * product path (badger_amd/csrc).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - control flow, slicing, k-mer hits, q-gram filter, rank/unrank, edge sets:
 *     pinned against the reference's own Python modules run in the build
 *     container (tools/gen_golden.py -> tests/golden/).
 *   - Levenshtein distance: uniquely defined (unit cost); pinned by definition.
 *   - Smith-Waterman SCORE: uniquely defined.  Smith-Waterman COORDINATES
 *     (ref_end/read_end/read_begin tie rules of the third-party `ssw-py`
 *     library, absent from /root/reference and from this image): restated
 *     from the published SSW algorithm, PARITY UNPINNED against ssw-py itself.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root).
 */
#ifndef BADGER_ORACLE_H
#define BADGER_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same 32-byte layout as bdg_extract_rec in include/badger_hip.h. */
typedef struct orc_extract_rec {
    int32_t  polyT;
    int32_t  r1_end;
    int32_t  bc_start;
    int32_t  umi_start;
    int32_t  umi_end;
    uint32_t bc_rank;
    int8_t   r1_score;
    int8_t   strand;
    uint8_t  valid;
    uint8_t  flags;
    uint32_t reserved;
} orc_extract_rec;

#define ORC_FLAG_REV      1u   /* result comes from the reverse-complement strand */
#define ORC_FLAG_RANK_OK  2u   /* bc_rank holds rank() of a full 16-base ACGT barcode */
#define ORC_FLAG_BC16     4u   /* the barcode slice holds 16 bases (some may be N) */

typedef struct orc_edge {
    uint32_t a;      /* a < b (ranks) */
    uint32_t b;
    uint32_t dist;
} orc_edge;

/* barcode_extraction/common.py:10-31 */
int  orc_find_polyt_start(const char* seq, int len, int window_size, int polya_count);
/* barcode_extraction/common.py:34-39 ; returns 0, or -1 on a byte outside "ACGTN " */
int  orc_revcomp(const char* seq, int len, char* out);
/* barcode_extraction/kmer_indexer.py:20-27,49-75 for the single pattern R1, k=6.
 * Writes ascending k-mer start positions; returns the count (cap = max written). */
int  orc_kmer_hits(const char* seq, int len, int32_t* pos, int cap);
/* ssw.AlignmentMgr(match_score=1, mismatch_penalty=1).align(gap_open=1, gap_extension=1)
 * as called at barcode_extraction/common.py:42-51 (read = pattern, reference = window).
 * out[5] = {reference_start, reference_end, read_start, read_end, optimal_score},
 * 0-based inclusive, relative to the window. */
void orc_sw_align(const char* pattern, int plen, const char* ref, int rlen, int32_t out[5]);
/* barcode_extraction/common.py:85-114.  Returns 1 when a position was found.
 * out[3] = {start_pos, end_pos(+leftover), score}. */
int  orc_detect_exact_positions(const char* seq, int start, int end,
                                const int32_t* hits, int nhits,
                                int min_score, int start_delta, int end_delta,
                                int32_t out[3]);
/* barcode_callers.py:165-229 (find_barcode_umi on one read). Returns 0 or -1 (bad base). */
int  orc_extract_read(const char* seq, int len, int umi_len, orc_extract_rec* rec);
/* The same with the strand rule chosen: ORC_RULE_DEFAULT = find_barcode_umi (:165-179), ORC_RULE_NO_POLYA =
 * find_barcode_umi_no_polya (:231-248: forward result if valid, else reverse if valid, else the more informative).
 * The reference forms the reverse complement only when the forward result is invalid, so there a bad base raises
 * only then; here it is reported for every read (stricter on invalid input, identical on valid input). */
#define ORC_RULE_DEFAULT  0
#define ORC_RULE_NO_POLYA 1
int  orc_extract_read_rule(const char* seq, int len, int umi_len, int rule, orc_extract_rec* rec);
int64_t orc_extract_batch_rule(const uint8_t* bases, const uint64_t* off, uint32_t n,
                               uint32_t umi_len, int rule, orc_extract_rec* out, int threads);
/* extract_raw_barcodes.py:126-128 over a concatenated batch; OpenMP over reads.
 * Returns 0, or -(index+1) of the first read holding a bad base. */
int64_t orc_extract_batch(const uint8_t* bases, const uint64_t* off, uint32_t n,
                          uint32_t umi_len, orc_extract_rec* out, int threads);

/* common.py:21-38 */
uint32_t orc_rank16(const char* seq);
void     orc_unrank16(uint32_t rk, char* out16);
/* editdistance.eval (unit-cost Levenshtein), plain DP. */
int  orc_levenshtein(const char* a, int la, const char* b, int lb);
/* Same value via Myers' bit-vector algorithm on rank-packed 16-mers (CPU baseline). */
int  orc_lev16_packed(uint32_t a, int la, uint32_t b, int lb);
/* barcode_graph.py:243  min(ed(a,b), ed(a[:-1],b), ed(a,b[:-1])) */
int  orc_dmin3(uint32_t a, uint32_t b);
/* index.py:77-93 closed form  S(a,b) = #{(p,p'): a[p:p+6]==b[p':p'+6]} */
int  orc_qgram_S(uint32_t a, uint32_t b);
/* index.py:19-24 */
int  orc_qgram_threshold(int threshold, int bc_len, int q);

/* barcode_graph.py:207-249 restricted to the edge set: distinct ranks in, edges
 * (a<b) with S>=qgram_T and dmin<=thr out, sorted by (a,b).  Bucket method of
 * index.py:29-35,77-93.  Returns total edges found (may exceed cap). */
uint64_t orc_graph_edges(const uint32_t* ranks, uint32_t n, uint32_t thr, int32_t qgram_T,
                         orc_edge* out, uint64_t cap, int threads);
/* The same for every row_stride-th row of the sorted array only (row i against every j > i of the whole array, index over
 * all n rows): a bounded sample of the full-size job for bench.py's cpu_baseline and parity check.  t[0] / t[1] receive the
 * seconds spent on the index / in the row loop (t may be NULL). */
uint64_t orc_graph_edges_sampled(const uint32_t* ranks, uint32_t n, uint32_t thr, int32_t qgram_T, uint32_t row_stride,
                                 orc_edge* out, uint64_t cap, int threads, double* t);
/* Same edge set by brute force over all pairs (cross-check of the bucket method). */
uint64_t orc_graph_edges_brute(const uint32_t* ranks, uint32_t n, uint32_t thr, int32_t qgram_T,
                               orc_edge* out, uint64_t cap);

/* barcode_graph.py:376-384 loop body as an operator: per query the whitelist entry of
 * minimal Levenshtein distance (ties -> lowest whitelist index), its distance and the
 * number of entries at that distance; distance > max_ed reports idx 0xFFFFFFFF,
 * ed 0xFF, ties 0. */
void orc_nearest16(const uint32_t* q, uint32_t nq, const uint32_t* wl, uint32_t nw,
                   uint32_t max_ed, uint32_t* best_idx, uint8_t* best_ed, uint16_t* n_ties,
                   int threads);

/* Same answers for max_ed <= 2 by enumerating the query's edit neighbourhood against a hash set of the
 * whitelist (larger max_ed falls through to orc_nearest16).  Not the reference's algorithm: the CPU baseline of
 * the GPU path's algorithm class and a fast checker for large samples; pinned by equality with orc_nearest16. */
void orc_nearest16_probe(const uint32_t* q, uint32_t nq, const uint32_t* wl, uint32_t nw,
                         uint32_t max_ed, uint32_t* best_idx, uint8_t* best_ed, uint16_t* n_ties,
                         int threads);

#ifdef __cplusplus
}
#endif
#endif
