/*
 * badger_oracle.c -- CPU restatement of the algbio/Badger barcode-calling hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see badger_oracle.h).  Plain C, no device code.
 * It restates WHAT the reference computes; it shares no code with it (the
 * reference is Python).  Citations are reference-relative file:line.
 */
#define _POSIX_C_SOURCE 199309L                /* clock_gettime */
#include "badger_oracle.h"

#include <stdlib.h>
#include <time.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* barcode_callers.py:154 */
static const char R1[] = "CTACACGACGCTCTTCCGATCT";
#define R1_LEN 22
#define KMER 6
#define BC_LEN 16

/* ------------------------------------------------------------------ */
/* barcode_extraction/common.py:10-31                                  */
/* ------------------------------------------------------------------ */
int orc_find_polyt_start(const char* seq, int len, int window_size, int polya_count)
{
    if (len < window_size) return -1;                       /* :13-14 */
    int i = 0;
    int a_count = 0;
    for (int k = 0; k < window_size; ++k) a_count += (seq[k] == 'T');   /* :16 */
    while (i < len - window_size) {                         /* :17 */
        if (a_count >= polya_count) break;                  /* :18-19 */
        int first_base_a = seq[i] == 'T';
        int new_base_a = (i + window_size < len) && seq[i + window_size] == 'T';
        if (first_base_a && !new_base_a) a_count -= 1;
        else if (!first_base_a && new_base_a) a_count += 1;
        i += 1;
    }
    if (i >= len - window_size) return -1;                  /* :28-29 */
    /* :31  i + max(0, seq[i:].find('TTT')) */
    int off = 0;
    for (int k = i; k + 2 < len; ++k) {
        if (seq[k] == 'T' && seq[k + 1] == 'T' && seq[k + 2] == 'T') { off = k - i; break; }
    }
    return i + off;
}

/* ------------------------------------------------------------------ */
/* barcode_extraction/common.py:34-39                                  */
/* ------------------------------------------------------------------ */
int orc_revcomp(const char* seq, int len, char* out)
{
    for (int i = 0; i < len; ++i) {
        char c = seq[len - 1 - i], r;
        switch (c) {
        case 'A': r = 'T'; break;
        case 'C': r = 'G'; break;
        case 'G': r = 'C'; break;
        case 'T': r = 'A'; break;
        case 'N': r = 'N'; break;
        case ' ': r = ' '; break;
        default: return -1;          /* KeyError in the reference */
        }
        out[i] = r;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* barcode_extraction/kmer_indexer.py:14-32 (index over [R1]) and      */
/* :49-75 (get_occurrences, single pattern => the top-hits filter is a */
/* no-op): every 6-mer start `pos` whose 6-mer equals an R1 6-mer, one */
/* entry per matching R1 6-mer (R1's 17 six-mers are distinct).        */
/* ------------------------------------------------------------------ */
int orc_kmer_hits(const char* seq, int len, int32_t* pos, int cap)
{
    int n = 0;
    for (int p = 0; p + KMER <= len; ++p) {
        for (int q = 0; q + KMER <= R1_LEN; ++q) {
            if (memcmp(seq + p, R1 + q, KMER) == 0) {
                if (n < cap) pos[n] = p;
                n++;
            }
        }
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* ssw.AlignmentMgr(match_score=1, mismatch_penalty=1)                 */
/*    .align(gap_open=1, gap_extension=1)                              */
/* Third-party (PyPI ssw-py, unpinned; C library: Complete-Striped-    */
/* Smith-Waterman).  Published algorithm restated:                     */
/*  - H(i,j) = max(0, H(i-1,j-1)+s, E, F); gap of k bases costs        */
/*    open + (k-1)*ext = k here, so E/F collapse to H(neigh)-1.        */
/*  - s = +1 equal bases, -1 different, 0 if either is N (SSW's        */
/*    "ambiguous base: no penalty" matrix row; unverified vs ssw-py).  */
/*  - end cell: first reference column (left->right) at which the      */
/*    running maximum strictly increases to its final value; within    */
/*    it the smallest read index holding that value.                   */
/*  - begin cell: same scan over ref[0..ref_end] / read[0..read_end]   */
/*    both reversed, stopped at the first column whose maximum equals  */
/*    the forward score.                                               */
/* ------------------------------------------------------------------ */
static int sw_score(char p, char r)
{
    if (p == 'N' || r == 'N') return 0;
    return p == r ? 1 : -1;
}

/* Forward scan. read = pattern (rows), ref = window (columns).
 * If terminate > 0 the scan stops after the first column whose max equals it. */
static void sw_scan(const char* read, int m, const char* ref, int n, int rev,
                    int terminate, int* o_max, int* o_end_ref, int* o_end_read)
{
    int hprev[64], hcur[64], hbest[64];
    for (int i = 0; i < m; ++i) { hprev[i] = 0; hbest[i] = 0; }
    int max = 0, end_ref = -1;
    for (int jj = 0; jj < n; ++jj) {
        int j = rev ? n - 1 - jj : jj;          /* column visiting order */
        char rc = ref[j];
        int colmax = 0;
        for (int i = 0; i < m; ++i) {
            char pc = rev ? read[m - 1 - i] : read[i];
            int diag = (i > 0 ? hprev[i - 1] : 0) + sw_score(pc, rc);
            int up = (i > 0 ? hcur[i - 1] : 0) - 1;
            int left = hprev[i] - 1;
            int h = 0;
            if (diag > h) h = diag;
            if (up > h) h = up;
            if (left > h) h = left;
            hcur[i] = h;
            if (h > colmax) colmax = h;
        }
        if (colmax > max) {
            max = colmax;
            end_ref = j;
            for (int i = 0; i < m; ++i) hbest[i] = hcur[i];
        }
        for (int i = 0; i < m; ++i) hprev[i] = hcur[i];
        if (terminate > 0 && colmax == terminate) break;
    }
    int end_read = m - 1;
    for (int i = 0; i < m; ++i) {
        if (hbest[i] == max && i < end_read) end_read = i;
    }
    *o_max = max; *o_end_ref = end_ref; *o_end_read = end_read;
}

void orc_sw_align(const char* pattern, int plen, const char* ref, int rlen, int32_t out[5])
{
    int score, ref_end, read_end;
    sw_scan(pattern, plen, ref, rlen, 0, 0, &score, &ref_end, &read_end);
    int ref_begin = -1, read_begin = -1;
    if (score > 0 && ref_end >= 0) {
        int s2, rb, rr;
        sw_scan(pattern, read_end + 1, ref, ref_end + 1, 1, score, &s2, &rb, &rr);
        ref_begin = rb;
        read_begin = read_end - rr;
    }
    out[0] = ref_begin; out[1] = ref_end; out[2] = read_begin; out[3] = read_end; out[4] = score;
}

/* ------------------------------------------------------------------ */
/* barcode_extraction/common.py:85-114                                 */
/* ------------------------------------------------------------------ */
int orc_detect_exact_positions(const char* seq, int start, int end,
                               const int32_t* hits, int nhits,
                               int min_score, int start_delta, int end_delta,
                               int32_t out[3])
{
    out[0] = -1; out[1] = -1; out[2] = 0;
    if (nhits <= 0) return 0;                                           /* :87-88 */
    int have = 0, start_pos = 0, end_pos = 0, pattern_start = 0, pattern_end = 0, score = 0;
    /* :91-94 last_potential_pos is never updated => every hit is aligned */
    for (int h = 0; h < nhits; ++h) {
        int match_position = hits[h];
        int potential_start = start + match_position - R1_LEN + KMER;   /* :96 */
        if (potential_start < start) potential_start = start;           /* :97 */
        int potential_end = start + match_position + R1_LEN + 1;        /* :98 */
        if (potential_end > end) potential_end = end;                   /* :99 */
        int32_t a[5];
        orc_sw_align(R1, R1_LEN, seq + potential_start, potential_end - potential_start, a);
        if (a[4] < min_score) continue;                                 /* :49-50 */
        if (a[4] > score) {                                             /* :102-103 */
            have = 1;
            start_pos = potential_start + a[0];
            end_pos = potential_start + a[1];
            pattern_start = a[2];
            pattern_end = a[3];
            score = a[4];
        }
    }
    if (!have) return 0;                                                /* :105-106 */
    if (start_delta >= 0 && pattern_start > start_delta) return 0;      /* :108-109 */
    if (end_delta >= 0 && R1_LEN - pattern_end - 1 > end_delta) return 0; /* :110-111 */
    int leftover = R1_LEN - pattern_end - 1;                            /* :113 */
    out[0] = start_pos; out[1] = end_pos + leftover; out[2] = score;    /* :114 */
    return 1;
}

/* ------------------------------------------------------------------ */
/* barcode_callers.py:181-229  _find_barcode_umi_fwd                   */
/* ------------------------------------------------------------------ */
typedef struct {
    int valid, polyT, r1, r1_score, bc_start, umi_start, umi_end;
} strand_res;

static void extract_strand(const char* s, int len, int umi_len, int32_t* hits, strand_res* r)
{
    r->valid = 0; r->r1 = -1; r->r1_score = 0; r->bc_start = -1; r->umi_start = -1; r->umi_end = -1;
    int polyt = orc_find_polyt_start(s, len, 16, 12);                   /* :183 */
    int found = 0;
    int32_t o[3];
    if (polyt != -1) {                                                  /* :186-192 relaxed */
        int nh = orc_kmer_hits(s, polyt + 1, hits, len + 1);
        found = orc_detect_exact_positions(s, 0, polyt + 1, hits, nh, 9, -1, 4, o);
    }
    if (!found) {                                                       /* :195-202 strict */
        int nh = orc_kmer_hits(s, len, hits, len + 1);
        found = orc_detect_exact_positions(s, 0, len, hits, nh, 17, 1, 1, o);
    }
    r->polyT = polyt;
    if (!found) return;                                                 /* :204-205 */
    int r1_end = o[1], r1_score = o[2];
    if (polyt != -1 && polyt - r1_end < BC_LEN) return;                 /* :208-209 */
    if (polyt == -1 || polyt - r1_end > BC_LEN + umi_len + 10) {        /* :211-218 */
        int presumable = r1_end + BC_LEN + umi_len;
        int ss = presumable - 4;
        int se = presumable + 10; if (se > len) se = len;
        int sl = (ss < len && se > ss) ? se - ss : 0;
        polyt = sl > 0 ? orc_find_polyt_start(s + ss, sl, 5, 5) : -1;
        if (polyt != -1) polyt += ss;
    }
    int barcode_start = r1_end + 1;                                     /* :220 */
    int barcode_end = r1_end + BC_LEN;
    int umi_start = barcode_end + 1;                                    /* :224 */
    int umi_end = polyt - 1;
    if (umi_end - umi_start <= 5) umi_end = umi_start + umi_len - 1;    /* :226-227 */
    r->valid = 1; r->polyT = polyt; r->r1 = r1_end; r->r1_score = r1_score;
    r->bc_start = barcode_start; r->umi_start = umi_start; r->umi_end = umi_end + 1;
}

static int code_of(char c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; }
    return -1;
}

/* barcode_callers.py:165-179 find_barcode_umi (rule 0) and :231-248 find_barcode_umi_no_polya (rule 1) */
int orc_extract_read_rule(const char* seq, int len, int umi_len, int rule, orc_extract_rec* rec)
{
    char* rc = (char*)malloc((size_t)len + 1);
    int32_t* hits = (int32_t*)malloc(sizeof(int32_t) * ((size_t)len + 2));
    memset(rec, 0, sizeof(*rec));
    if (orc_revcomp(seq, len, rc) != 0) { free(rc); free(hits); return -1; }
    strand_res f, v;
    extract_strand(seq, len, umi_len, hits, &f);                        /* :166 */
    extract_strand(rc, len, umi_len, hits, &v);                         /* :170-171 */
    int use_rev;
    if (rule == ORC_RULE_NO_POLYA) {
        /* :234-237 forward result if valid; :244-245 else reverse if valid; :247 neither is valid: both carry
         * r1_score 0 (:204-209 build the result without a score), so "forward if more informative" never holds */
        use_rev = f.valid ? 0 : v.valid ? 1 : !(f.r1_score > v.r1_score);
    } else if (v.valid && f.valid) use_rev = !(f.r1_score > v.r1_score);   /* :175-176 */
    else if (v.valid) use_rev = 1;                                      /* :177-178 */
    else use_rev = 0;                                                   /* :179 */
    const strand_res* c = use_rev ? &v : &f;
    const char* s = use_rev ? rc : seq;
    rec->polyT = c->polyT;
    rec->r1_end = c->r1;
    rec->bc_start = c->bc_start;
    rec->umi_start = c->umi_start;
    rec->umi_end = c->umi_end;
    rec->r1_score = (int8_t)c->r1_score;
    rec->strand = c->polyT != -1 ? (use_rev ? -1 : 1) : 0;              /* :167-168,172-173 */
    rec->valid = (uint8_t)c->valid;
    rec->flags = use_rev ? ORC_FLAG_REV : 0;
    rec->bc_rank = 0;
    if (c->valid && c->bc_start + BC_LEN <= len) {
        uint32_t rk = 0; int ok = 1;
        for (int i = 0; i < BC_LEN; ++i) {
            int cd = code_of(s[c->bc_start + i]);
            if (cd < 0) { ok = 0; break; }
            rk |= (uint32_t)cd << (2 * i);
        }
        rec->flags |= ORC_FLAG_BC16;
        if (ok) { rec->bc_rank = rk; rec->flags |= ORC_FLAG_RANK_OK; }
    }
    free(rc); free(hits);
    return 0;
}

int orc_extract_read(const char* seq, int len, int umi_len, orc_extract_rec* rec)
{
    return orc_extract_read_rule(seq, len, umi_len, ORC_RULE_DEFAULT, rec);
}

int64_t orc_extract_batch(const uint8_t* bases, const uint64_t* off, uint32_t n,
                          uint32_t umi_len, orc_extract_rec* out, int threads)
{
    return orc_extract_batch_rule(bases, off, n, umi_len, ORC_RULE_DEFAULT, out, threads);
}

int64_t orc_extract_batch_rule(const uint8_t* bases, const uint64_t* off, uint32_t n,
                               uint32_t umi_len, int rule, orc_extract_rec* out, int threads)
{
    int64_t bad = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        int rcode = orc_extract_read_rule((const char*)bases + off[i], (int)(off[i + 1] - off[i]),
                                          (int)umi_len, rule, &out[i]);
        if (rcode != 0) {
#pragma omp critical
            { if (bad == 0 || -(i + 1) > bad) bad = -(i + 1); }
        }
    }
    return bad;
}

/* ------------------------------------------------------------------ */
/* common.py:21-38                                                     */
/* ------------------------------------------------------------------ */
uint32_t orc_rank16(const char* seq)
{
    uint32_t rk = 0;
    for (int i = 0; i < BC_LEN; ++i) rk += (uint32_t)code_of(seq[i]) << (2 * i);   /* RANK*4^i */
    return rk;
}

void orc_unrank16(uint32_t rk, char* out16)
{
    static const char U[4] = { 'A', 'C', 'G', 'T' };
    for (int i = 0; i < BC_LEN; ++i) { out16[i] = U[rk % 4]; rk /= 4; }
}

/* editdistance.eval: unit-cost Levenshtein (third-party PyPI `editdistance`, unpinned;
 * the value is uniquely defined).  Call sites barcode_graph.py:96,243,379. */
int orc_levenshtein(const char* a, int la, const char* b, int lb)
{
    int prev[65], cur[65];
    for (int j = 0; j <= lb; ++j) prev[j] = j;
    for (int i = 1; i <= la; ++i) {
        cur[0] = i;
        for (int j = 1; j <= lb; ++j) {
            int c = prev[j - 1] + (a[i - 1] != b[j - 1]);
            if (prev[j] + 1 < c) c = prev[j] + 1;
            if (cur[j - 1] + 1 < c) c = cur[j - 1] + 1;
            cur[j] = c;
        }
        memcpy(prev, cur, sizeof(int) * (size_t)(lb + 1));
    }
    return prev[lb];
}

/* Myers 1999 / Hyyro 2001 bit-vector Levenshtein on rank-packed strings
 * (2 bits per base, base i in bits [2i,2i+1]); la, lb <= 16. */
int orc_lev16_packed(uint32_t a, int la, uint32_t b, int lb)
{
    if (la == 0) return lb;
    uint32_t peq[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < la; ++i) peq[(a >> (2 * i)) & 3] |= 1u << i;
    uint32_t mask = la >= 32 ? 0xFFFFFFFFu : ((1u << la) - 1u);
    uint32_t top = 1u << (la - 1);
    uint32_t pv = mask, mv = 0;
    int score = la;
    for (int j = 0; j < lb; ++j) {
        uint32_t eq = peq[(b >> (2 * j)) & 3];
        uint32_t xv = eq | mv;
        uint32_t xh = ((((eq & pv) + pv) ^ pv) | eq) & mask;
        uint32_t ph = (mv | ~(xh | pv)) & mask;
        uint32_t mh = pv & xh;
        if (ph & top) score++;
        else if (mh & top) score--;
        ph = ((ph << 1) | 1u) & mask;
        mh = (mh << 1) & mask;
        pv = (mh | ~(xv | ph)) & mask;
        mv = ph & xv;
    }
    return score;
}

/* barcode_graph.py:243 (and :96) */
int orc_dmin3(uint32_t a, uint32_t b)
{
    int d0 = orc_lev16_packed(a, 16, b, 16);
    int d1 = orc_lev16_packed(a, 15, b, 16);   /* barcode[:-1] vs sequence */
    int d2 = orc_lev16_packed(a, 16, b, 15);   /* barcode vs sequence[:-1] */
    int d = d0;
    if (d1 < d) d = d1;
    if (d2 < d) d = d2;
    return d;
}

/* index.py:29-35 + :77-93: distances[j] = sum over the 11 q-grams of `barcode`
 * (with multiplicity) of index[q][j]  ==  number of (p,p') with equal 6-grams. */
int orc_qgram_S(uint32_t a, uint32_t b)
{
    int s = 0;
    for (int p = 0; p + KMER <= BC_LEN; ++p) {
        uint32_t qa = (a >> (2 * p)) & 0xFFFu;
        for (int q = 0; q + KMER <= BC_LEN; ++q) s += qa == ((b >> (2 * q)) & 0xFFFu);
    }
    return s;
}

/* index.py:19-24 */
int orc_qgram_threshold(int threshold, int bc_len, int q)
{
    int t = bc_len - q + 1 - q * threshold;
    if (t <= 0) t = 4;
    return t;
}

/* ------------------------------------------------------------------ */
/* barcode_graph.py:207-249 -- edge set                                */
/* ------------------------------------------------------------------ */
static int cmp_u32(const void* x, const void* y)
{
    uint32_t a = *(const uint32_t*)x, b = *(const uint32_t*)y;
    return a < b ? -1 : a > b;
}

static int cmp_edge(const void* x, const void* y)
{
    const orc_edge* a = (const orc_edge*)x; const orc_edge* b = (const orc_edge*)y;
    if (a->a != b->a) return a->a < b->a ? -1 : 1;
    if (a->b != b->b) return a->b < b->b ? -1 : 1;
    return 0;
}

typedef struct { orc_edge* v; uint64_t n, cap; } edge_vec;

static void ev_push(edge_vec* e, uint32_t a, uint32_t b, uint32_t d)
{
    if (e->n == e->cap) {
        e->cap = e->cap ? e->cap * 2 : 1024;
        e->v = (orc_edge*)realloc(e->v, e->cap * sizeof(orc_edge));
    }
    e->v[e->n].a = a; e->v[e->n].b = b; e->v[e->n].dist = d; e->n++;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* The edges of every row_stride-th row of the sorted array (row i against all j > i of the WHOLE array), index over all
 * n rows: the same work per row as the full job, on a bounded sample of rows.  t[0] = seconds spent building the index,
 * t[1] = seconds in the row loop (NULL: not wanted). */
uint64_t orc_graph_edges_sampled(const uint32_t* ranks, uint32_t n, uint32_t thr, int32_t qgram_T, uint32_t row_stride,
                                 orc_edge* out, uint64_t cap, int threads, double* t)
{
    if (t) t[0] = t[1] = 0.0;
    if (n == 0) return 0;
    if (row_stride < 1) row_stride = 1;
    const double t_begin = now_s();
    if (threads < 1) threads = 1;
    uint32_t* r = (uint32_t*)malloc(sizeof(uint32_t) * n);
    memcpy(r, ranks, sizeof(uint32_t) * n);
    qsort(r, n, sizeof(uint32_t), cmp_u32);
    /* QGramIndex buckets (index.py:29-35): one entry per (q-gram occurrence, barcode). */
    const int NB = 4096, NQ = BC_LEN - KMER + 1;
    uint64_t* bstart = (uint64_t*)calloc((size_t)NB + 1, sizeof(uint64_t));
    for (uint32_t i = 0; i < n; ++i)
        for (int p = 0; p < NQ; ++p) bstart[((r[i] >> (2 * p)) & 0xFFFu) + 1]++;
    for (int g = 0; g < NB; ++g) bstart[g + 1] += bstart[g];
    uint32_t* bent = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n * NQ);
    uint64_t* fill = (uint64_t*)malloc(sizeof(uint64_t) * NB);
    memcpy(fill, bstart, sizeof(uint64_t) * NB);
    for (uint32_t i = 0; i < n; ++i)
        for (int p = 0; p < NQ; ++p) bent[fill[(r[i] >> (2 * p)) & 0xFFFu]++] = i;   /* ascending i */
    free(fill);

    edge_vec* per = (edge_vec*)calloc((size_t)threads, sizeof(edge_vec));
    const double t_indexed = now_s();
    const int64_t n_sampled = ((int64_t)n + row_stride - 1) / row_stride;
#pragma omp parallel num_threads(threads)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        uint16_t* acc = (uint16_t*)calloc(n, sizeof(uint16_t));
        uint32_t* touched = (uint32_t*)malloc(sizeof(uint32_t) * n);
#pragma omp for schedule(dynamic, 256)
        for (int64_t ii = 0; ii < n_sampled; ++ii) {
            uint32_t i = (uint32_t)(ii * row_stride), nt = 0;
            /* get_close (index.py:77-93): entries j > number */
            for (int p = 0; p < NQ; ++p) {
                uint32_t g = (r[i] >> (2 * p)) & 0xFFFu;
                uint64_t lo = bstart[g], hi = bstart[g + 1];
                while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (bent[mid] <= i) lo = mid + 1; else hi = mid; }
                for (uint64_t e = lo; e < bstart[g + 1]; ++e) {
                    uint32_t j = bent[e];
                    if (acc[j]++ == 0) touched[nt++] = j;
                }
            }
            for (uint32_t t = 0; t < nt; ++t) {
                uint32_t j = touched[t];
                int s = acc[j]; acc[j] = 0;
                if (s >= qgram_T) {                                     /* index.py:90-92 */
                    int d = orc_dmin3(r[i], r[j]);                      /* barcode_graph.py:243 */
                    if (d <= (int)thr) ev_push(&per[tid], r[i], r[j], (uint32_t)d);
                }
            }
        }
        free(acc); free(touched);
    }
    if (t) { t[0] = t_indexed - t_begin; t[1] = now_s() - t_indexed; }
    uint64_t total = 0;
    for (int w = 0; w < threads; ++w) total += per[w].n;
    orc_edge* all = (orc_edge*)malloc(sizeof(orc_edge) * (total ? total : 1));
    uint64_t k = 0;
    for (int w = 0; w < threads; ++w) {
        if (per[w].n) memcpy(all + k, per[w].v, per[w].n * sizeof(orc_edge));
        k += per[w].n; free(per[w].v);
    }
    qsort(all, total, sizeof(orc_edge), cmp_edge);
    for (uint64_t e = 0; e < total && e < cap; ++e) out[e] = all[e];
    free(all); free(per); free(bent); free(bstart); free(r);
    return total;
}

uint64_t orc_graph_edges(const uint32_t* ranks, uint32_t n, uint32_t thr, int32_t qgram_T,
                         orc_edge* out, uint64_t cap, int threads)
{
    return orc_graph_edges_sampled(ranks, n, thr, qgram_T, 1, out, cap, threads, NULL);
}

uint64_t orc_graph_edges_brute(const uint32_t* ranks, uint32_t n, uint32_t thr, int32_t qgram_T,
                               orc_edge* out, uint64_t cap)
{
    if (n == 0) return 0;
    uint32_t* r = (uint32_t*)malloc(sizeof(uint32_t) * n);
    memcpy(r, ranks, sizeof(uint32_t) * n);
    qsort(r, n, sizeof(uint32_t), cmp_u32);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        for (uint32_t j = i + 1; j < n; ++j) {
            int s = orc_qgram_S(r[i], r[j]);
            if (s < 1 || s < qgram_T) continue;     /* only barcodes sharing a q-gram are ever listed */
            int d = orc_dmin3(r[i], r[j]);
            if (d > (int)thr) continue;
            if (total < cap) { out[total].a = r[i]; out[total].b = r[j]; out[total].dist = (uint32_t)d; }
            total++;
        }
    }
    free(r);
    return total;
}

/* ------------------------------------------------------------------ */
/* barcode_graph.py:376-384 loop body as an operator                   */
/* ------------------------------------------------------------------ */
void orc_nearest16(const uint32_t* q, uint32_t nq, const uint32_t* wl, uint32_t nw,
                   uint32_t max_ed, uint32_t* best_idx, uint8_t* best_ed, uint16_t* n_ties,
                   int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)nq; ++i) {
        int best = 255; uint32_t idx = 0xFFFFFFFFu; uint32_t ties = 0;
        for (uint32_t w = 0; w < nw; ++w) {
            int d = orc_lev16_packed(q[i], 16, wl[w], 16);
            if (d < best) { best = d; idx = w; ties = 1; }              /* strict <: first minimum wins */
            else if (d == best) ties++;
        }
        if (best > (int)max_ed) { best = 255; idx = 0xFFFFFFFFu; ties = 0; }
        best_idx[i] = idx; best_ed[i] = (uint8_t)best;
        n_ties[i] = (uint16_t)(ties > 0xFFFFu ? 0xFFFFu : ties);
    }
}

/* ------------------------------------------------------------------ */
/* The same operator for max_ed <= 2 by neighbourhood enumeration: the */
/* query, its 48 substitution neighbours, its 1080 double substitutions */
/* and its 1024 delete-one-insert-one variants are looked up in a hash  */
/* set of the whitelist (equal-length strings at Levenshtein distance 1 */
/* differ by one substitution; at distance 2 by two substitutions or    */
/* one deletion plus one insertion).  NOT the reference's algorithm     */
/* (barcode_graph.py:376-384 scans every center): it is the CPU         */
/* baseline of the same algorithm class as the GPU path, and it lets    */
/* the tests check large samples.  Pinned by equality with              */
/* orc_nearest16 (tests/test_oracle_golden.py).                         */
/* ------------------------------------------------------------------ */
typedef struct { uint32_t* slot; uint32_t mask; int shift; const uint32_t* wl; } wl_set;

static inline int wl_find(const wl_set* s, uint32_t key, uint32_t* idx)
{
    uint32_t h = (key * 0x9E3779B1u) >> s->shift;
    for (;;) {
        uint32_t v = s->slot[h];
        if (v == 0) return 0;
        if (s->wl[v - 1] == key) { *idx = v - 1; return 1; }
        h = (h + 1) & s->mask;
    }
}

void orc_nearest16_probe(const uint32_t* q, uint32_t nq, const uint32_t* wl, uint32_t nw,
                         uint32_t max_ed, uint32_t* best_idx, uint8_t* best_ed, uint16_t* n_ties,
                         int threads)
{
    if (threads < 1) threads = 1;
    if (max_ed > 2 || nw == 0) { orc_nearest16(q, nq, wl, nw, max_ed, best_idx, best_ed, n_ties, threads); return; }
    int bits = 4;
    while ((1ull << bits) < 2ull * nw) ++bits;
    wl_set set;
    set.slot = (uint32_t*)calloc((size_t)1 << bits, sizeof(uint32_t));
    set.mask = (uint32_t)((1ull << bits) - 1); set.shift = 32 - bits; set.wl = wl;
    for (uint32_t w = 0; w < nw; ++w) {                       /* distinct entries (the ABI requires it) */
        uint32_t h = (wl[w] * 0x9E3779B1u) >> set.shift;
        while (set.slot[h]) h = (h + 1) & set.mask;
        set.slot[h] = w + 1;
    }
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)nq; ++i) {
        const uint32_t a = q[i];
        uint32_t idx = 0xFFFFFFFFu, found; int best = 255; uint32_t ties = 0;
        if (wl_find(&set, a, &found)) { best = 0; idx = found; ties = 1; }
        if (best == 255 && max_ed >= 1) {
            for (int p = 0; p < 16; ++p)
                for (uint32_t x = 1; x < 4; ++x)
                    if (wl_find(&set, a ^ (x << (2 * p)), &found)) { best = 1; ties++; if (found < idx) idx = found; }
        }
        if (best == 255 && max_ed >= 2) {
            uint32_t seen[256]; uint32_t ns = 0; int overflow = 0;
            for (int p = 0; p < 16 && !overflow; ++p)
                for (int r = p + 1; r < 16; ++r)
                    for (uint32_t x = 1; x < 4; ++x)
                        for (uint32_t y = 1; y < 4; ++y)
                            if (wl_find(&set, a ^ (x << (2 * p)) ^ (y << (2 * r)), &found)) {
                                if (ns < 256) seen[ns++] = found; else overflow = 1;    /* distinct strings: no duplicates here */
                            }
            for (int p = 0; p < 16 && !overflow; ++p) {
                const uint32_t lm = p ? ((1u << (2 * p)) - 1u) : 0u;
                const uint32_t d15 = ((a & lm) | ((a >> 2) & ~lm)) & 0x3FFFFFFFu;                 /* base p deleted */
                for (int k = 0; k < 16; ++k) {
                    const uint32_t km = k ? ((1u << (2 * k)) - 1u) : 0u;
                    for (uint32_t c = 0; c < 4; ++c) {
                        const uint32_t b = (d15 & km) | (c << (2 * k)) | ((d15 & ~km) << 2);   /* letter c inserted at k */
                        if (b == a || !wl_find(&set, b, &found)) continue;
                        int dup = 0;
                        for (uint32_t t = 0; t < ns; ++t) if (seen[t] == found) { dup = 1; break; }
                        if (!dup) { if (ns < 256) seen[ns++] = found; else overflow = 1; }
                    }
                }
            }
            if (overflow) {                                    /* a crowd of neighbours: settle it exhaustively */
                orc_nearest16(&a, 1, wl, nw, max_ed, &best_idx[i], &best_ed[i], &n_ties[i], 1);
                continue;
            }
            if (ns) { best = 2; ties = ns; for (uint32_t t = 0; t < ns; ++t) if (seen[t] < idx) idx = seen[t]; }
        }
        if (best > (int)max_ed) { best = 255; idx = 0xFFFFFFFFu; ties = 0; }
        best_idx[i] = idx; best_ed[i] = (uint8_t)best;
        n_ties[i] = (uint16_t)(ties > 0xFFFFu ? 0xFFFFu : ties);
    }
    free(set.slot);
}
