// K3: edit-distance graph over the distinct barcodes (reference
// BarcodeGraph.graph_construction / compare_chunk, barcode_graph.py:75-111,207-249,
// with the q-gram candidate filter of QGramIndex.get_close, index.py:77-93).
//
// An edge (a<b) exists iff  S(a,b) >= T  and  dmin(a,b) <= thr  where
//   S    = #{(p,p') : a[p:p+6] == b[p':p'+6]}        (what index.py accumulates)
//   dmin = min(ed(a,b), ed(a[:-1],b), ed(a,b[:-1]))  (barcode_graph.py:243)
//
// k_graph_scan: tiled all-pairs sweep over the sorted rank array.  Each lane owns one
// row barcode; column tiles (rank + letter-count signature) are staged in LDS and
// broadcast.  A pair survives the sweep only if the L1 distance of the letter counts
// (one v_sad_u8) is <= 2*thr+1, a bound every pair with dmin <= thr meets; survivors are
// compacted per wave into an LDS queue and verified 64 at a time, so the expensive part
// (one Myers pass yielding D[16][16], D[15][16], D[16][15]; then S by 21 shifted XORs)
// always runs with full lanes.
//
// k_graph_probe (thr = 1): instead of sweeping pairs, every barcode enumerates the 16-mers
// that can have dmin <= 1 with it and looks them up in the sorted array (membership bitmap,
// then a prefix directory); see graph_probe_candidate() below.
//
// k_graph_qjoin_w (thr >= 3, cross-check at thr 2): the device form of the reference's QGramIndex, see further down.
// k_d1_rows / k_d2_rows + k_part_* + k_d2_pairs_w (thr 1 from 100 K rows, thr 2 from 10 K): the deletion-variant joins, see
// "Deletion-variant join" below.  No sort or scan library anywhere: every grouping is csrc/bdg_partition.hpp.
#include "bdg_common.hpp"
#include "bdg_partition.hpp"
#include "dj_codec.hpp"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int GT = 2048;     // column tile

__device__ __forceinline__ uint32_t lanes_below_u64(unsigned long long m, int lane)     // set bits of m in the lanes below this one
{
    (void)lane;
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint32_t letter_sig(uint32_t r)
{
    // byte k = number of bases with rank code k
    const uint32_t lo = r & 0x55555555u, hi = (r >> 1) & 0x55555555u;
    const uint32_t c3 = __popc(lo & hi), c1 = __popc(lo & ~hi), c2 = __popc(hi & ~lo);
    const uint32_t c0 = 16u - c1 - c2 - c3;
    return c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
}

// One Myers pass, pattern a (rows), text b (columns) -> min of D[16][16], D[15][16], D[16][15].
// The vectors are kept SPREAD: row i at bit 2i, where a's 2-bit codes put it.  The equality vector of a column then needs no
// per-letter match vectors (16 x 4 compares to build them): it is two three-input operations on a's two bit planes and the
// column's code bits spread over the word, as in k_strict_filter.  The one addition of the recurrence must carry from bit 2i
// to bit 2i + 2: pv keeps every odd bit set (its update, mh | ~(xv | ph), sets them by itself since xv and ph have none), so
// a carry passes through; xh and mh then hold carries in their odd bits, which nothing reads.  293 vector instructions where
// the form with match vectors took 511 (round 4; checked against the edit distance on the host, tools/myers_spread_check.py).
__device__ __forceinline__ uint32_t dmin3(uint32_t a, uint32_t b)
{
    constexpr uint32_t EVEN = 0x55555555u;
    const uint32_t P0 = a & EVEN, P1 = (a >> 1) & EVEN;
    uint32_t pv = 0xFFFFFFFFu, mv = 0u, score = 16u, score15 = 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)b, 2 * j, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int)b, 2 * j + 1, 1);
        const uint32_t t1 = __builtin_amdgcn_bitop3_b32(m0, P0, EVEN, 0x82);              // ~(m0 ^ P0) & EVEN
        const uint32_t eq = __builtin_amdgcn_bitop3_b32(t1, m1, P1, 0x90);                // t1 & ~(m1 ^ P1)
        const uint32_t xv = eq | mv;
        const uint32_t xh = __builtin_amdgcn_bitop3_b32((eq & pv) + pv, pv, eq, 0xBE);     // (((eq & pv) + pv) ^ pv) | eq
        uint32_t ph = __builtin_amdgcn_bitop3_b32(mv, xh, pv, 0xF1);                      // mv | ~(xh | pv)
        uint32_t mh = pv & xh;
        score += (ph >> 30) & 1u;
        score -= (mh >> 30) & 1u;
        ph = (ph << 2) | 1u;
        mh = mh << 2;
        pv = __builtin_amdgcn_bitop3_b32(mh, xv, ph, 0xF1);                               // mh | ~(xv | ph)
        mv = ph & xv;
        if (j == 14) score15 = score;              // D[16][15] = ed(a, b[:-1])
    }
    // D[15][16] = D[16][16] - (vertical delta of the last row in the last column)
    const uint32_t d1516 = score - ((pv >> 30) & 1u) + ((mv >> 30) & 1u);   // ed(a[:-1], b)
    uint32_t d = score < score15 ? score : score15;
    return d < d1516 ? d : d1516;
}

// S(a,b): matching 6-gram position pairs, diagonal by diagonal.
__device__ __forceinline__ uint32_t qgram_S(uint32_t a, uint32_t b)
{
    uint32_t s = 0;
#pragma unroll
    for (int sh = -10; sh <= 10; ++sh) {
        // compare a[p] with b[p+sh]
        const int len = 16 - (sh < 0 ? -sh : sh);
        const uint32_t x = sh >= 0 ? (a ^ (b >> (2 * sh))) : ((a >> (-2 * sh)) ^ b);
        uint32_t z = ~(x | (x >> 1)) & 0x55555555u;
        z &= len >= 16 ? 0xFFFFFFFFu : ((1u << (2 * len)) - 1u);
        const uint32_t z2 = z & (z >> 2);
        const uint32_t z4 = z2 & (z2 >> 4);          // runs of 4
        const uint32_t z6 = z4 & (z2 >> 8);          // runs of 6
        s += __popc(z6);
    }
    return s;
}

__global__ __launch_bounds__(256)
void k_graph_sig(const uint32_t* __restrict__ ranks, uint32_t n, uint32_t* __restrict__ sig)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) sig[i] = letter_sig(ranks[i]);
}

// Edge output.  Edges are staged per wave in LDS; a block reserves output slots with ONE atomic when it ends (returning
// atomics on one address complete ~11 ns apart device-wide, so one per edge - or per wave step - would bound the kernel).
constexpr uint32_t ECAP = 128;                   // staged edges per wave

template <uint32_t CAP>
struct EdgeStageT { static constexpr uint32_t cap = CAP; uint32_t a[CAP], b[CAP]; uint8_t d[CAP]; };
using EdgeStage = EdgeStageT<ECAP>;

template <class Stage>
__device__ __forceinline__ void edge_copy_out(const Stage& st, uint32_t n, unsigned long long base, int lane,
                                              bdg_edge* __restrict__ out, uint64_t cap)
{
    for (uint32_t i = (uint32_t)lane; i < n; i += 64u) {
        const unsigned long long k = base + i;
        if (k < cap) { out[k].a = st.a[i]; out[k].b = st.b[i]; out[k].dist = st.d[i]; }
    }
}
// wave-wide: lanes with `want` append their edge; a full stage is written out with the wave's own reservation
template <class Stage>
__device__ __forceinline__ void edge_push(bool want, uint32_t a, uint32_t b, uint32_t d, Stage& st, uint32_t& n, int lane,
                                          bdg_edge* __restrict__ out, uint64_t cap, unsigned long long* n_edges)
{
    const unsigned long long m = __ballot(want);
    if (!m) return;
    const uint32_t cnt = (uint32_t)__popcll(m);
    if (n + cnt > Stage::cap) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(n_edges, (unsigned long long)n);
        base = __shfl(base, 0);
        edge_copy_out(st, n, base, lane, out, cap);
        n = 0;
        __builtin_amdgcn_wave_barrier();
    }
    if (want) {
        const uint32_t at = n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        st.a[at] = a; st.b[at] = b; st.d[at] = (uint8_t)d;
    }
    n += cnt;
}
// block-wide, every thread: one reservation for the waves' stages (NW waves per block)
template <int NW = 4, class Stage = EdgeStage>
__device__ __forceinline__ void edge_finish(Stage* stages, uint32_t n, uint32_t* s_cnt, unsigned long long* s_base,
                                            bdg_edge* __restrict__ out, uint64_t cap, unsigned long long* n_edges)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wv] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += s_cnt[w];
        *s_base = tot ? atomicAdd(n_edges, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    unsigned long long base = *s_base;
    for (int w = 0; w < wv; ++w) base += s_cnt[w];
    edge_copy_out(stages[wv], n, base, lane, out, cap);
}

__device__ __forceinline__ void verify_pair(bool on, uint32_t a, uint32_t b, uint32_t thr, int32_t T,
                                            EdgeStage& st, uint32_t& ne, int lane,
                                            bdg_edge* out, uint64_t cap, unsigned long long* n_edges)
{
    const uint32_t d = on ? dmin3(a, b) : 99u;
    const bool close = d <= thr;
    if (__ballot(close)) {
        const bool edge = close && (int32_t)qgram_S(a, b) >= T;
        edge_push(edge, a, b, d, st, ne, lane, out, cap, n_edges);
    }
}

__global__ __launch_bounds__(256)
void k_graph_scan(const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ sig, uint32_t n,
                  uint32_t row_begin, uint32_t row_end,
                  uint32_t thr, int32_t T, bdg_edge* __restrict__ out, uint64_t cap,
                  unsigned long long* __restrict__ n_edges)
{
    __shared__ uint32_t s_r[GT], s_s[GT];
    __shared__ uint32_t s_qa[4][128], s_qb[4][128];
    __shared__ EdgeStage s_edges[4];
    __shared__ uint32_t s_ecnt[4];
    __shared__ unsigned long long s_ebase;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t ne = 0;
    // triangular load balance: block k takes row tile k from the front and the matching one from the back
    const uint32_t tile0 = row_begin / 256u, ntiles = (row_end + 255u) / 256u;      // row tiles [tile0, ntiles) of this block of rows
    const uint32_t lim = 2u * thr + 1u;
    for (uint32_t pass = 0; pass < 2; ++pass) {
        const uint32_t tile = pass == 0 ? tile0 + blockIdx.x : ntiles - 1u - blockIdx.x;
        if (pass == 1 && tile <= tile0 + blockIdx.x) break;  // middle tile handled once
        if (tile >= ntiles) break;
        const uint32_t i = tile * 256u + tid;
        const bool row = i >= row_begin && i < row_end;
        const uint32_t a = row ? ranks[i] : 0u, sa = row ? sig[i] : 0u;
        uint32_t qn = 0;
        for (uint32_t j0 = tile * 256u; j0 < n; j0 += GT) {
            const uint32_t tn = n - j0 < (uint32_t)GT ? n - j0 : (uint32_t)GT;
            __syncthreads();
            for (uint32_t k = tid; k < tn; k += 256u) { s_r[k] = ranks[j0 + k]; s_s[k] = sig[j0 + k]; }
            __syncthreads();
            for (uint32_t k = 0; k < tn; ++k) {
                const uint32_t b = s_r[k], sb = s_s[k];
                const bool ok = row && (j0 + k > i) && __builtin_amdgcn_sad_u8(sa, sb, 0u) <= lim;
                const unsigned long long m = __ballot(ok);
                if (m) {
                    if (ok) {
                        const uint32_t idx = qn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        s_qa[wv][idx] = a; s_qb[wv][idx] = b;
                    }
                    qn += (uint32_t)__popcll(m);
                    if (qn >= 64u) {
                        __builtin_amdgcn_wave_barrier();
                        qn -= 64u;
                        const uint32_t pa = s_qa[wv][qn + lane], pb = s_qb[wv][qn + lane];
                        __builtin_amdgcn_wave_barrier();
                        verify_pair(true, pa, pb, thr, T, s_edges[wv], ne, lane, out, cap, n_edges);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (qn) {
            const bool on = (uint32_t)lane < qn;
            const uint32_t pa = on ? s_qa[wv][lane] : 0u, pb = on ? s_qb[wv][lane] : 0u;
            verify_pair(on, pa, pb, thr, T, s_edges[wv], ne, lane, out, cap, n_edges);
        }
        __syncthreads();
    }
    edge_finish(s_edges, ne, s_ecnt, &s_ebase, out, cap, n_edges);
}

// ---------------------------------------------------------------------------
// thr = 1 neighbourhood probes.
// For 16-mers a != b, dmin(a,b) <= 1 iff one of
//   (1) ed(a,b)       <= 1 : b is a with one substitution                       (48 candidates)
//   (2) ed(a[:15],b)  <= 1 : b is a[:15] with one base inserted                 (16 slots x 4)
//   (3) ed(a,b[:15])  <= 1 : b[:15] is a with one base deleted, b[15] free      (16 x 4)
//       or b[:15] == a[:15]... (that is a substitution of the last base: case 1)
// (distance-0 prefixes: ed(a[:15], b) = 0 is impossible (lengths differ); the
//  insert/delete cases already cover ed = 1, the only achievable value <= 1.)
// Each candidate b > a found in the sorted array is verified with dmin3 + S like any pair,
// so duplicates among the cases only cost a lookup; an edge is emitted by the FIRST
// candidate slot that produces b (lower slots are checked for equality).
// ---------------------------------------------------------------------------
constexpr int NPROBE = 48 + 64 + 64;

__device__ __forceinline__ uint32_t lowm(int bases) { return bases >= 16 ? 0xFFFFFFFFu : ((1u << (2 * bases)) - 1u); }

__device__ __forceinline__ uint32_t graph_probe_candidate(uint32_t a, int t)
{
    if (t < 48) {
        const int pos = t / 3; const uint32_t x = 1u + (uint32_t)(t % 3);
        return a ^ (x << (2 * pos));
    }
    if (t < 112) {              // insert letter c at slot sl of a[:15]
        const int u = t - 48, sl = u >> 2; const uint32_t c = (uint32_t)u & 3u;
        const uint32_t d = a & lowm(15);
        const uint32_t sm = lowm(sl);
        return (d & sm) | (c << (2 * sl)) | ((d & ~sm) << 2);
    }
    {                           // delete base i of a, append letter c
        const int u = t - 112, i = u >> 2; const uint32_t c = (uint32_t)u & 3u;
        const uint32_t lm = lowm(i);
        const uint32_t d = (a & lm) | ((a >> 2) & ~lm);          // 15 bases
        return (d & lowm(15)) | (c << 30);
    }
}

// Is t the lowest slot whose candidate equals b = graph_probe_candidate(a, t)?  (closed form of "no u < t yields b")
//   substitutions (t < 48) are pairwise distinct and come first;
//   any later candidate at Hamming distance 1 from a repeats a substitution;
//   inserting c at slot sl repeats slot sl-1 iff the base before the slot is c (and only then: equal strings force c' = c
//   and a run of c between the two slots);
//   deleting base i repeats i-1 iff a[i] == a[i-1]; a deletion candidate also repeats an insertion candidate iff
//   removing one base of b yields a[:15].
__device__ __forceinline__ bool graph_probe_first(uint32_t a, uint32_t b, int t)
{
    if (t < 48) return true;
    const uint32_t x = a ^ b;
    if (__popc((x | (x >> 1)) & 0x55555555u) == 1) return false;
    if (t < 112) {
        const int u = t - 48, sl = u >> 2; const uint32_t c = (uint32_t)u & 3u;
        return sl == 0 || ((a >> (2 * (sl - 1))) & 3u) != c;
    }
    const int i = (t - 112) >> 2;
    if (i > 0 && (((a >> (2 * i)) ^ (a >> (2 * i - 2))) & 3u) == 0u) return false;
    const uint32_t a15 = a & lowm(15);
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) {
        const uint32_t lm = lowm(sl);
        if ((((b & lm) | ((b >> 2) & ~lm)) & lowm(15)) == a15) return false;
    }
    return true;
}

// Is a candidate one of the ranks?  One that passed the membership bitmap is looked up through a prefix directory over the
// same top bits (dir[b] = first row whose rank >> shift is >= b): the range, then its one or two rows, where a binary search
// over the sorted array made about log2(n) dependent round trips (k_graph_probe does this for four candidates at a time).

// membership bitmap over the top `32 - shift` bits of the ranks (most candidates die here on one L2 hit) and the directory
// over the same bits: rows with equal top bits are neighbours in the sorted array, the first of them fills the directory
// entries since the previous row's bucket
__global__ __launch_bounds__(256)
void k_graph_bitmap(const uint32_t* __restrict__ ranks, uint32_t n, int shift, uint32_t* __restrict__ bitmap, uint32_t* __restrict__ dir)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = ranks[i] >> shift;
    const uint32_t nb = 0xFFFFFFFFu >> shift;                                // last bucket
    const uint32_t first = i ? (ranks[i - 1] >> shift) + 1u : 0u;            // buckets (previous row's, b] start at row i
    if (i == 0 || first <= b) atomicOr(&bitmap[b >> 5], 1u << (b & 31u));
    for (uint32_t q = first; q <= b; ++q) dir[q] = i;
    if (i == n - 1) for (uint32_t q = b + 1u; q <= nb + 1u; ++q) dir[q] = n;
}

__global__ __launch_bounds__(256)
void k_graph_probe(const uint32_t* __restrict__ ranks, uint32_t n, uint32_t row_begin, uint32_t row_end, int32_t T,
                   const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ dir, int shift,
                   bdg_edge* __restrict__ out, uint64_t cap, unsigned long long* __restrict__ n_edges)
{
    __shared__ EdgeStage s_edges[4];
    __shared__ uint32_t s_ecnt[4];
    __shared__ unsigned long long s_ebase;
    // 4 lanes per barcode: lane sub-index s takes candidates s, s+4, ...
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t i = row_begin + (gid >> 2); const int sub = (int)(gid & 3u);
    const bool on = i < row_end;
    const uint32_t a = on ? ranks[i] : 0u;
    uint32_t ne = 0;
    // Four candidates per round: their bitmap words are loaded together, then the directory ranges of those that passed,
    // then the rows; a round costs three round trips to L2 / memory instead of up to twelve.  (NPROBE = 176 = 11 rounds of
    // 4 candidates for each of the 4 lanes of a barcode: same trip count in every lane.)
    static_assert(NPROBE % 16 == 0, "rounds of four candidates per lane");
    for (int t0 = sub; t0 < NPROBE; t0 += 16) {
        uint32_t b[4], word[4]; bool cand[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            b[u] = graph_probe_candidate(a, t0 + 4 * u);
            cand[u] = on && b[u] > a;
            word[u] = bitmap[(cand[u] ? b[u] >> shift : 0u) >> 5];
        }
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t bb = b[u] >> shift;
            cand[u] = cand[u] && ((word[u] >> (bb & 31u)) & 1u) != 0;
            const uint32_t q = cand[u] ? bb : 0u;
            lo[u] = dir[q]; hi[u] = dir[q + 1];
            if (!cand[u]) hi[u] = lo[u] = 0u;
        }
        uint32_t first_row[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) first_row[u] = ranks[lo[u] < hi[u] ? lo[u] : 0u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bool edge = cand[u] && lo[u] < hi[u] && first_row[u] == b[u];
            for (uint32_t k = lo[u] + 1u; k < hi[u]; ++k) edge = edge || (cand[u] && ranks[k] == b[u]);   // buckets of several rows: rare
            if (edge) edge = graph_probe_first(a, b[u], t0 + 4 * u);         // de-duplicate: only the lowest slot producing b emits
            uint32_t d = 0;
            if (edge) { d = dmin3(a, b[u]); edge = d <= 1u && (int32_t)qgram_S(a, b[u]) >= T; }
            edge_push(edge, a, b[u], d, s_edges[wv], ne, lane, out, cap, n_edges);
        }
    }
    edge_finish(s_edges, ne, s_ecnt, &s_ebase, out, cap, n_edges);
}


// ---------------------------------------------------------------------------
// thr >= 2: q-gram join (the device form of QGramIndex, index.py:29-35,77-93).
//
// The reference keeps 4096 buckets {rank: count} and, for every barcode, sums the counts of all later
// barcodes over its 11 six-mers: distances[j] = S(i, j); candidates are the j with S >= T.  Here:
//   k_qj_count / k_part_colscan / k_part_bases / k_qj_place   one entry row << 4 | position per barcode and position (11 n
//                entries) in its six-mer's bucket, rows ascending inside it (a stable counting sort on the 12 key bits), where
//                each (row, position) landed (pos_of), where each bucket starts,
//   k_qj_split   for every bucket the first entry at or past each multiple of W rows (so that a row can take its
//                candidates in slices of W rows without searching),
//   k_graph_qjoin one block per row i: for each slice of later rows, every entry of the 11 bucket tails is one unit
//                of S(i, j); the block keeps S(i, j) in one byte of LDS per row j of the slice (direct addressing, no keys)
//                - the count IS the reference's statistic - and the add that lifts a byte to T lists j; listed rows are
//                verified with one Myers pass (dmin3) 64 at a time.  Nothing is computed for the ~99 % of candidate
//                pairs that share a single six-mer by chance, except one LDS atomic.
// A slice takes any number of entries.  If a row's list of candidates overflows (dense single-cell inputs), the slice's rows
// are read back from the counters instead.  bdg_graph_set_algo(ctx, 4) verifies every entry by itself in closed form
// (qgram_S + "is this the first matching position pair"): the cross-check of the tests.
// ---------------------------------------------------------------------------
constexpr int QJ_NQ = 11;                        // six-mers per 16-mer
constexpr uint32_t QJ_W = 32768;                 // rows per slice = bytes of LDS counters per block

// The index: every (row, position) entry in its six-mer's bucket, rows ascending inside a bucket - a stable counting sort on
// the 12 key bits, in the two runs of csrc/bdg_partition.hpp (round 3 called hipCUB's radix sort here).
// k_qj_count: a tile of rows per block, how many entries it has for each of the 4,096 six-mers (one column of the
// buckets x tiles matrix; k_part_colscan / k_part_bases turn it into places: tiles ascend inside a bucket).
__global__ __launch_bounds__(256)
void k_qj_count(const uint32_t* __restrict__ ranks, uint32_t n, uint32_t rows_per_tile, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_h[4096];
    for (uint32_t i = threadIdx.x; i < 4096u; i += 256u) s_h[i] = 0u;
    __syncthreads();
    const uint32_t row0 = blockIdx.x * rows_per_tile;
    const uint32_t row1 = n - row0 < rows_per_tile ? n : row0 + rows_per_tile;
    for (uint32_t row = row0 + threadIdx.x; row < row1; row += 256u) {
        const uint32_t r = ranks[row];
#pragma unroll
        for (uint32_t p = 0; p < (uint32_t)QJ_NQ; ++p) atomicAdd(&s_h[(r >> (2u * p)) & 0xFFFu], 1u);     // barcode[p:p+6] (index.py:31-33)
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 4096u; i += 256u) hist[(size_t)i * gridDim.x + blockIdx.x] = s_h[i];
}

// k_qj_place: one WAVE per tile walks the tile's entries in (row, position) order, 64 at a time, and gives every entry the
// next place of its bucket: inside a group of 64 the entries of one six-mer are told apart by twelve ballots (one per key
// bit: the lanes that agree with this one on every bit), an entry's place is the bucket's cursor plus the number of such
// lanes below it, and the last of them moves the cursor on.  Rows therefore ascend inside every bucket, positions inside
// a row.  vals[place] = row << 4 | position, pos_of[row * 11 + position] = place, bucket_off[q] = where bucket q starts.
__global__ __launch_bounds__(64)
void k_qj_place(const uint32_t* __restrict__ ranks, uint32_t n, uint32_t rows_per_tile, const uint32_t* __restrict__ hist,
                const unsigned long long* __restrict__ base, uint32_t* __restrict__ vals, uint32_t* __restrict__ pos_of,
                uint32_t* __restrict__ bucket_off /* [4097] */)
{
    __shared__ uint32_t s_cur[4096];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4096u; i += 64u) {
        const uint32_t b = (uint32_t)base[i];
        s_cur[i] = b + hist[(size_t)i * gridDim.x + blockIdx.x];
        if (blockIdx.x == 0) bucket_off[i] = b;
    }
    if (blockIdx.x == 0 && lane == 0) bucket_off[4096] = (uint32_t)base[4096];
    __builtin_amdgcn_wave_barrier();
    const uint32_t row0 = blockIdx.x * rows_per_tile;
    const uint32_t row1 = n - row0 < rows_per_tile ? n : row0 + rows_per_tile;
    const unsigned long long g0 = (unsigned long long)row0 * QJ_NQ, g1 = (unsigned long long)row1 * QJ_NQ;
    for (unsigned long long gb = g0; gb < g1; gb += 64ull) {
        const unsigned long long g = gb + lane;
        const bool on = g < g1;
        const uint32_t row = on ? (uint32_t)(g / QJ_NQ) : 0u, p = on ? (uint32_t)(g % QJ_NQ) : 0u;
        const uint32_t key = on ? (ranks[row] >> (2u * p)) & 0xFFFu : 0u;
        unsigned long long peers = __ballot(on);
#pragma unroll
        for (uint32_t bit = 0; bit < 12u; ++bit) {
            const unsigned long long m = __ballot((key >> bit) & 1u);
            peers &= ((key >> bit) & 1u) ? m : ~m;
        }
        if (on) {
            const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
            const uint32_t at = s_cur[key] + below;
            vals[at] = (row << 4) | p;
            pos_of[g] = at;
        }
        __builtin_amdgcn_wave_barrier();                              // (every lane has read its cursor)
        if (on && (peers >> lane) == 1ull) s_cur[key] += (uint32_t)__popcll(peers);      // the group's highest lane
        __builtin_amdgcn_wave_barrier();
    }
}

// split[q * (G + 1) + g] = first entry of bucket q whose row is >= g * W
__global__ __launch_bounds__(256)
void k_qj_split(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ bucket_off, uint32_t G, uint32_t W,
                uint32_t* __restrict__ split)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= 4096u * (G + 1u)) return;
    const uint32_t q = t / (G + 1u), g = t % (G + 1u);
    uint32_t lo = bucket_off[q], hi = bucket_off[q + 1];
    const uint64_t want = (uint64_t)g * W;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint64_t)(vals[mid] >> 4) < want) lo = mid + 1; else hi = mid;
    }
    split[t] = lo;
}

// first matching six-mer position pair (p in a, p' in b), lexicographically: p * 16 + p'; 0xFFFFFFFF if none
__device__ __forceinline__ uint32_t qgram_first_match(uint32_t a, uint32_t b)
{
    uint32_t best = 0xFFFFFFFFu;
#pragma unroll
    for (int sh = -10; sh <= 10; ++sh) {
        const int len = 16 - (sh < 0 ? -sh : sh);
        const uint32_t x = sh >= 0 ? (a ^ (b >> (2 * sh))) : ((a >> (-2 * sh)) ^ b);
        uint32_t z = ~(x | (x >> 1)) & 0x55555555u;
        z &= len >= 16 ? 0xFFFFFFFFu : ((1u << (2 * len)) - 1u);
        const uint32_t z2 = z & (z >> 2);
        const uint32_t z4 = z2 & (z2 >> 4);
        const uint32_t z6 = z4 & (z2 >> 8);
        if (z6) {
            const uint32_t p = (uint32_t)__builtin_ctz(z6) >> 1;           // index in the unshifted operand
            const uint32_t pa = sh >= 0 ? p : p + (uint32_t)(-sh), pb = sh >= 0 ? p + (uint32_t)sh : p;
            const uint32_t key = pa * 16u + pb;
            best = key < best ? key : best;
        }
    }
    return best;
}

constexpr uint32_t QJ_GMAX = 64;      // slices whose bounds are kept in LDS at a time (a row walks its slices in groups of this many)
constexpr uint32_t QJ_HCAP = 1024;    // rows j with S >= T waiting for their Myers pass (verified once per row)

// One block per row i.  The later rows are taken in slices of W consecutive rows; within a slice S(i, j) lives in ONE BYTE
// of LDS per row j (S <= 121 fits), addressed directly by j - slice start: counting a candidate entry is a single returning
// ds_add, and the add that lifts a byte to T lists j for the Myers pass (a count passes T exactly once).  No keys, no probing,
// no capacity to overflow: every slice takes any number of entries.
template <uint32_t W>
__global__ __launch_bounds__(256)
void k_graph_qjoin(const uint32_t* __restrict__ ranks, uint32_t n, uint32_t row_begin, uint32_t row_end,
                   const uint32_t* __restrict__ vals, const uint32_t* __restrict__ pos_of,
                   const uint32_t* __restrict__ split, uint32_t G,
                   uint32_t thr, int32_t T, int closed_form,
                   bdg_edge* __restrict__ out, uint64_t cap, unsigned long long* __restrict__ n_edges)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[W / 4];   // S(i, j) for the rows j of the current slice, one byte each
    __shared__ uint32_t s_split[QJ_NQ][QJ_GMAX + 1];                 // bounds of the row's 11 bucket tails, slice by slice
    __shared__ uint32_t s_hit[QJ_HCAP];
    __shared__ uint32_t s_nh, s_over;
    __shared__ EdgeStage s_edges[4];
    __shared__ uint32_t s_ecnt[4];
    __shared__ unsigned long long s_ebase;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t ne = 0;
    for (uint32_t k = tid; k < W / 4; k += 256u) s_cnt[k] = 0u;
    const uint32_t Tc = T < 1 ? 1u : (uint32_t)T;

    for (uint32_t i = row_begin + blockIdx.x; i < row_end; i += gridDim.x) {
        const uint32_t a = ranks[i];
        auto verify = [&](bool on, uint32_t j) {
            uint32_t b = 0, d = 99u;
            if (on) { b = ranks[j]; d = dmin3(a, b); }
            edge_push(on && d <= thr, a, b, d, s_edges[wv], ne, lane, out, cap, n_edges);
        };
        // one Myers pass per listed row, full lanes (block-wide; the list is empty afterwards)
        auto flush_hits = [&](uint32_t nh) {
            for (uint32_t h0 = 0; h0 < nh; h0 += 256u) {
                const uint32_t h = h0 + (uint32_t)tid;
                verify(h < nh, h < nh ? s_hit[h] : 0u);
            }
            __syncthreads();
            if (tid == 0) s_nh = 0u;
            __syncthreads();
        };
        uint32_t nh0 = 0;                                            // list length when the current slice began
        __syncthreads();                                             // previous row done with s_split / s_hit
        if (tid == 0) { s_nh = 0u; s_over = 0u; }
        const uint32_t g0 = i / W;                                   // row i lies in slice g0, its candidates in slices g0 .. G-1
        for (uint32_t gg = g0; gg < G; gg += QJ_GMAX) {
            // the bounds of up to QJ_GMAX slices in one round trip
            const uint32_t ng = G - gg < QJ_GMAX ? G - gg : QJ_GMAX;
            __syncthreads();
            for (uint32_t t = tid; t < QJ_NQ * (ng + 1u); t += 256u) {
                const uint32_t sg = t / (ng + 1u), gi = t % (ng + 1u);
                uint32_t v = split[(size_t)((a >> (2u * sg)) & 0xFFFu) * (G + 1u) + gg + gi];
                if (gg + gi == g0) { const uint32_t tail = pos_of[(size_t)i * QJ_NQ + sg] + 1u; v = v > tail ? v : tail; }   // behind row i's own entry
                s_split[sg][gi] = v;
            }
            __syncthreads();
            for (uint32_t gi = 0; gi < ng; ++gi) {
                uint32_t total = 0;
#pragma unroll
                for (int sg = 0; sg < QJ_NQ; ++sg) {
                    const uint32_t lo = s_split[sg][gi], hi = s_split[sg][gi + 1];
                    total += hi > lo ? hi - lo : 0u;
                }
                if (total == 0) continue;
                const uint32_t base_row = (gg + gi) * W;
                // this wave's share of the slice: the runs of six-mers wv, wv + 4, wv + 8 as one flat list (entry t lies at
                // vals[t + off]), eight loads in flight per lane
                const uint32_t loA = s_split[wv][gi], hiA = s_split[wv][gi + 1];
                const uint32_t loB = s_split[wv + 4][gi], hiB = s_split[wv + 4][gi + 1];
                const bool hasC = wv + 8 < QJ_NQ;
                const uint32_t loC = hasC ? s_split[hasC ? wv + 8 : 0][gi] : 0u, hiC = hasC ? s_split[hasC ? wv + 8 : 0][gi + 1] : 0u;
                const uint32_t lenA = hiA > loA ? hiA - loA : 0u, lenB = hiB > loB ? hiB - loB : 0u, lenC = hiC > loC ? hiC - loC : 0u;
                const uint32_t eB = lenA + lenB, wt = eB + lenC;
                const uint32_t offA = loA, offB = loB - lenA, offC = loC - eB;
                if (closed_form) {
                    // cross-check path: every entry verified by itself; a pair is reported by its first matching position pair only
                    for (uint32_t t0 = 0; t0 < wt; t0 += 64u) {
                        const uint32_t t = t0 + (uint32_t)lane;
                        uint32_t b = 0, d = 99u; bool on = false;
                        if (t < wt) {
                            const uint32_t sgi = t < lenA ? (uint32_t)wv : (t < eB ? (uint32_t)wv + 4u : (uint32_t)wv + 8u);
                            const uint32_t v = vals[t + (t < lenA ? offA : (t < eB ? offB : offC))];
                            const uint32_t j = v >> 4;
                            if (j > i) {
                                b = ranks[j];
                                on = qgram_first_match(a, b) == sgi * 16u + (v & 15u) && qgram_S(a, b) >= Tc;
                                if (on) d = dmin3(a, b);
                            }
                        }
                        edge_push(on && d <= thr, a, b, d, s_edges[wv], ne, lane, out, cap, n_edges);
                    }
                    continue;
                }
                for (uint32_t t0 = 0; t0 < wt; t0 += 512u) {
                    uint32_t jv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const uint32_t t = t0 + 64u * (uint32_t)u + (uint32_t)lane;
                        const uint32_t o = t < lenA ? offA : (t < eB ? offB : offC);
                        jv[u] = t < wt ? vals[t + o] >> 4 : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const uint32_t j = jv[u];
                        if (j > i) {                                              // (nothing, or row i's own repeat of the six-mer, otherwise)
                            const uint32_t off = j - base_row, sh = (off & 3u) * 8u;
                            const uint32_t old = atomicAdd(&s_cnt[off >> 2], 1u << sh);
                            if (((old >> sh) & 0xFFu) + 1u == Tc) {               // this entry lifts S(i, j) to T: list j
                                const uint32_t at = atomicAdd(&s_nh, 1u);
                                if (at < QJ_HCAP) s_hit[at] = j; else s_over = 1u;
                            }
                        }
                    }
                }
                __syncthreads();
                const bool over = s_over != 0u;
                const uint32_t nh1 = s_nh;
                if (over) {
                    // the list overflowed during this slice: drop what the slice listed and take its rows from the counters
                    // instead (each lane clears the words it has read: no other wave touches them before the barrier below)
                    __syncthreads();                                              // everyone has read s_over / s_nh
                    if (tid == 0) { s_nh = nh0; s_over = 0u; }
                    for (uint32_t k0 = (uint32_t)wv * 64u; k0 < W / 4; k0 += 256u) {
                        const uint32_t w = s_cnt[k0 + lane];
                        s_cnt[k0 + lane] = 0u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool hit = ((w >> (8 * q)) & 0xFFu) >= Tc;
                            if (__ballot(hit)) verify(hit, base_row + 4u * (k0 + (uint32_t)lane) + (uint32_t)q);
                        }
                    }
                } else {
                    nh0 = nh1;                                                    // (< QJ_HCAP: the list did not overflow)
                    for (uint32_t k = (uint32_t)tid * 4u; k < W / 4; k += 1024u) *reinterpret_cast<uint4*>(&s_cnt[k]) = make_uint4(0u, 0u, 0u, 0u);
                }
                __syncthreads();
                if (nh0 > QJ_HCAP / 2u) { flush_hits(nh0); nh0 = 0u; }
            }
        }
        __syncthreads();
        flush_hits(s_nh < QJ_HCAP ? s_nh : QJ_HCAP);
    }
    __syncthreads();
    edge_finish(s_edges, ne, s_ecnt, &s_ebase, out, cap, n_edges);
}

// ---------------------------------------------------------------------------
// k_graph_qjoin_w: the same join with ONE WAVE per row and no block barrier at all (round 3; k_graph_qjoin above spent its
// time in three __syncthreads and a 32 KB clear per slice of ~900 entries).  A wave keeps S(i, j) for a slice of WQ rows in
// WQ bytes of LDS that only it touches, walks the row's 11 bucket tails with 11 cursors - a tail is sorted by row, so the
// entries of a slice are the next ones behind the cursor: every lane loads entry cursor + lane of every tail (11 coalesced
// loads in flight), a ballot says how many of them belong to the slice - and needs neither the per-slice bounds table nor a
// search.  The loads of the NEXT slice are issued before the counters of this one are touched (the cursors move as soon as
// the ballots are in), the counters a lane has touched are cleared by that lane (a byte store each) instead of clearing the
// slice, and slices without entries are skipped (the next slice is the one of the smallest unconsumed row).  Rows whose
// counter reaches T are listed per wave and verified 64 at a time.
// ---------------------------------------------------------------------------
constexpr uint32_t QW_HCAP = 128;                // listed rows per wave
constexpr uint32_t QW_SENT = 0x0FFFFFFFu;        // "no entry": a row beyond every slice (rows are < 2^25)

// ROWS: rows per slice = bytes of LDS counters per wave; WAVES: waves per block (the block's static LDS stays below 64 KB);
// T_GE2: T >= 2, which lets a spare word (always 0) stand in for "no entry" without a check
template <uint32_t ROWS, int WAVES, bool T_GE2>
__global__ __launch_bounds__(64 * WAVES)
void k_graph_qjoin_w(const uint32_t* __restrict__ ranks, uint32_t n, uint32_t row_begin, uint32_t row_end,
                     const uint32_t* __restrict__ vals, const uint32_t* __restrict__ pos_of, const uint32_t* __restrict__ bucket_off,
                     uint32_t thr, int32_t T,
                     bdg_edge* __restrict__ out, uint64_t cap, unsigned long long* __restrict__ n_edges)
{
    // per wave: the counters, 64 spare words behind them (lanes without an entry add 0 / store there: every LDS operation
    // below is unconditional, so the compiler issues a slice's eleven of each kind together and waits once), the list
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[WAVES][ROWS / 4 + 64];
    __shared__ uint32_t s_hit[WAVES][QW_HCAP];
    __shared__ EdgeStage s_edges[WAVES];
    __shared__ uint32_t s_ecnt[WAVES];
    __shared__ unsigned long long s_ebase;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* const cnt = s_cnt[wv];
    uint8_t* const cnt8 = reinterpret_cast<uint8_t*>(cnt);
    uint32_t* const hits = s_hit[wv];
    const uint32_t spare = ROWS + 4u * (uint32_t)lane;             // byte offset of this lane's spare word
    uint32_t ne = 0;
    for (uint32_t k = (uint32_t)lane; k < ROWS / 4 + 64; k += 64u) cnt[k] = 0u;
    __builtin_amdgcn_wave_barrier();
    const uint32_t Tc = T < 1 ? 1u : (uint32_t)T;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * (uint32_t)WAVES + (uint32_t)wv), nwaves = gridDim.x * (uint32_t)WAVES;
    const uint32_t m_last = n * (uint32_t)QJ_NQ - 1u;              // last entry of vals: where a lane without an entry loads from

    for (uint32_t i = row_begin + wave_id; i < row_end; i += nwaves) {
        const uint32_t a = __builtin_amdgcn_readfirstlane(ranks[i]);
        uint32_t nh = 0;
        auto flush_hits = [&]() {
            for (uint32_t h0 = 0; h0 < nh; h0 += 64u) {
                const uint32_t h = h0 + (uint32_t)lane;
                const bool on = h < nh;
                const uint32_t b = ranks[on ? hits[h] : i];
                const uint32_t d = on ? dmin3(a, b) : 99u;
                edge_push(on && d <= thr, a, b, d, s_edges[wv], ne, lane, out, cap, n_edges);
            }
            nh = 0;
            __builtin_amdgcn_wave_barrier();
        };
        // the 11 tails: behind row i's own entry of each of its six-mers, to the end of that six-mer's bucket
        uint32_t my_cur = 0, my_end = 0;
        if (lane < QJ_NQ) { my_cur = pos_of[(size_t)i * QJ_NQ + lane] + 1u; my_end = bucket_off[((a >> (2 * lane)) & 0xFFFu) + 1u]; }
        uint32_t cur[QJ_NQ], end[QJ_NQ];
#pragma unroll
        for (int b = 0; b < QJ_NQ; ++b) { cur[b] = (uint32_t)__builtin_amdgcn_readlane((int)my_cur, b); end[b] = (uint32_t)__builtin_amdgcn_readlane((int)my_end, b); }
        uint32_t jv[QJ_NQ];
        uint32_t base_row = QW_SENT;                                   // the slice starts at the smallest row not yet counted
        {
            uint32_t raw[QJ_NQ];
#pragma unroll
            for (int b = 0; b < QJ_NQ; ++b) { const uint32_t t = cur[b] + (uint32_t)lane; raw[b] = vals[t < m_last ? t : m_last]; }
#pragma unroll
            for (int b = 0; b < QJ_NQ; ++b) {
                jv[b] = cur[b] + (uint32_t)lane < end[b] ? raw[b] >> 4 : QW_SENT;
                const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)jv[b], 0);
                base_row = f < base_row ? f : base_row;
            }
        }
        // A slice is [base_row, slice_end): at most ROWS rows, and cut short where a tail has more entries in it than the 64
        // loaded (then the slice ends at that tail's last loaded row, whose entries wait for the next slice): every entry
        // of a slice is in registers when it is counted, so the lanes can clear exactly what they touched.  A tail holds a
        // row at most 11 times, so a slice always gets past its first row.
        while (base_row != QW_SENT) {
            uint32_t slice_end = base_row + ROWS;
#pragma unroll
            for (int b = 0; b < QJ_NQ; ++b) { const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)jv[b], 63); slice_end = l < slice_end ? l : slice_end; }
            uint32_t c[QJ_NQ], next_min = QW_SENT;
#pragma unroll
            for (int b = 0; b < QJ_NQ; ++b) {
                c[b] = (uint32_t)__popcll(__ballot(jv[b] < slice_end));         // (sorted by row: the first c[b] lanes; lane 63 never)
                cur[b] += c[b];
                const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)jv[b], (int)c[b]);
                next_min = f < next_min ? f : next_min;
            }
            // what comes next is known: its loads fly while this slice is counted
            uint32_t raw[QJ_NQ];
#pragma unroll
            for (int b = 0; b < QJ_NQ; ++b) { const uint32_t t = cur[b] + (uint32_t)lane; raw[b] = vals[t < m_last ? t : m_last]; }
            uint32_t old[QJ_NQ], boff[QJ_NQ];
#pragma unroll
            for (int b = 0; b < QJ_NQ; ++b) {
                const bool on = (uint32_t)lane < c[b] && jv[b] > i;               // (j == i: row i's own repeat of the six-mer)
                boff[b] = on ? jv[b] - base_row : spare;
                old[b] = atomicAdd(&cnt[boff[b] >> 2], on ? 1u << ((boff[b] & 3u) * 8u) : 0u);
            }
            uint32_t hitmask = 0;
#pragma unroll
            for (int b = 0; b < QJ_NQ; ++b) {                                      // this entry lifts S(i, j) to T: list j
                const bool lifts = __builtin_amdgcn_ubfe(old[b], (boff[b] & 3u) * 8u, 8u) == Tc - 1u;
                hitmask |= (lifts && (T_GE2 || boff[b] < ROWS)) ? 1u << b : 0u;
            }
            if (__ballot(hitmask != 0u)) {
                // listed rows go to the wave's list in rounds of what it still holds (one round unless a slice lists more than
                // QW_HCAP rows): lane l writes its hits at the positions its prefix count gives
                const uint32_t mine = (uint32_t)__popc(hitmask);
                const uint32_t incl = wave_incl_scan(mine);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                uint32_t taken = 0;
                for (;;) {
                    const uint32_t room = QW_HCAP - nh;
                    uint32_t k = incl - mine;                                      // index of this lane's next hit among the iteration's hits
#pragma unroll
                    for (int b = 0; b < QJ_NQ; ++b) {
                        if ((hitmask >> b) & 1u) {
                            if (k >= taken && k - taken < room) hits[nh + k - taken] = jv[b];
                            ++k;
                        }
                    }
                    const uint32_t now = total - taken < room ? total - taken : room;
                    nh += now; taken += now;
                    __builtin_amdgcn_wave_barrier();
                    if (taken == total) break;
                    flush_hits();
                }
            }
#pragma unroll
            for (int b = 0; b < QJ_NQ; ++b) cnt8[boff[b]] = 0;                    // every lane clears what it touched
#pragma unroll
            for (int b = 0; b < QJ_NQ; ++b) jv[b] = cur[b] + (uint32_t)lane < end[b] ? raw[b] >> 4 : QW_SENT;
            base_row = next_min;
        }
        flush_hits();
    }
    __syncthreads();
    edge_finish<WAVES>(s_edges, ne, s_ecnt, &s_ebase, out, cap, n_edges);
}


// ---------------------------------------------------------------------------
// Deletion-variant join (thr <= 2, large n).  The q-gram join above follows the reference's index literally: a row walks
// the tails of its 11 six-mer buckets, which grow with n, so its work is quadratic (500 K rows: 11 ms; 9.5 M rows: 6.9 s).
// The edge set itself does not need that walk:  dmin(a,b) <= 2  implies that a and b share a 14-mer that is left when two
// letters are deleted from each (two substitutions: delete the two positions; one insertion + one deletion: delete the
// odd letter of each, then any common letter; the forms through a[:-1] / b[:-1] delete the last letter and one or two
// more - a[:-1] minus one letter is a minus two).  So: every row emits its <= 120 distinct two-deletion 14-mers as 32-bit
// entries (dj_codec.hpp), the entries are GROUPED by 14-mer - two bucket levels through HBM, the rest in LDS
// (bdg_partition.hpp, k_d2_pairs_w; round 3 sorted them) - and only rows that meet in a group are
// verified - by the same dmin and the same S (in closed form) as everywhere else, so the filter stays the reference's.
// A pair shares several 14-mers; it is reported from exactly one group, named by a rule that looks at the two barcodes only
// (k_d2_pairs).  A row's entries are made distinct when they are emitted (of equal 14-mers the first deletion pair stays),
// so a group holds a row once and the rule names one pair of entries.
// Work is linear in n (about 71 entries per row) plus the pairs that meet.
// ---------------------------------------------------------------------------
constexpr int D2_NPAIR = 120;                    // C(16, 2) deletion pairs

// deletion pair t -> (p << 4 | q), p < q, in the order p = 0 (q = 1..15), p = 1 (q = 2..15), ...
struct D2Table { uint8_t pq[D2_NPAIR]; };
constexpr D2Table d2_make_table()
{
    D2Table t{};
    int k = 0;
    for (int p = 0; p < 16; ++p) for (int q = p + 1; q < 16; ++q) t.pq[k++] = (uint8_t)(p << 4 | q);
    return t;
}
__constant__ D2Table d2_table = d2_make_table();

// r without its bases p and q (p < q): a 14-mer in 28 bits
__device__ __forceinline__ uint32_t d2_key(uint32_t r, uint32_t p, uint32_t q)
{
    const uint32_t lo = r & ((1u << (2u * p)) - 1u);
    const uint32_t mid = (r >> (2u * p + 2u)) & ((1u << (2u * (q - p - 1u))) - 1u);
    const uint32_t hi = (uint32_t)((unsigned long long)r >> (2u * q + 2u));          // (q = 15: nothing)
    return lo | (mid << (2u * p)) | (hi << (2u * q - 2u));
}

// which of nparts shares of the 14-mers k belongs to (multiplicative hash: even whatever the barcodes look like)
__device__ __forceinline__ uint32_t d2_part(uint32_t k, uint32_t nparts)
{
    return (uint32_t)(((unsigned long long)(k * 2654435761u) * nparts) >> 32);
}

// One wave per row at a time (rows interleaved over the resident waves): lane l holds deletion pairs l and 64 + l.
// A pair is dropped when an earlier one of the row gives the same 14-mer.  Letters inside a run are interchangeable, so only
// the first letter of a run (or the first two, for two deletions in one run) need to be deleted: that alone removes most
// repeats (and, of equal 14-mers, keeps the earliest pair of the table).  The rest
// is settled exactly through a table of 1024 slots in LDS: a one-to-one mixing of the 28 key bits (multiply, shift-xor,
// multiply, all mod 2^28) gives 10 bits that name the slot and 18 that are a fingerprint, and the slot takes the minimum of
// (pair index << 18 | fingerprint).  A lane
// that finds its own value won; one that finds its fingerprint under an earlier pair is a repeat; one that finds another
// fingerprint lost the slot to a different 14-mer - so did every other holder of its own 14-mer, and those few lanes
// (about three a row) compare among themselves.
constexpr uint32_t D2_SLOTS = 1024;
constexpr unsigned long long D2_ROUND_ENTRIES = 1000000000ull;      // index entries (estimated at 72 / 12 a row) per round of the join
constexpr int DJ_THREADS = 512;                  // k_d2_pairs: threads of a block,
constexpr uint32_t DJ_CAP = 2048;                // entries of a fine bucket it holds in LDS at a time (>= the largest group: 1920),
constexpr uint32_t DJ_ECAPW = 256;               // edges a wave stages before it reserves output (128: 39 K reservations on one address, the pair walk 0.64 ms instead of 0.55)

struct D2Row { uint32_t k0, k1, z0, z1; bool keep0, keep1; };      // the lane's two 14-mers, their mixed keys, which stay

__device__ __forceinline__ void d2_tab_init(uint32_t* __restrict__ tab, int lane)
{
    uint4* const tab4 = reinterpret_cast<uint4*>(tab);
#pragma unroll
    for (uint32_t i = 0; i < D2_SLOTS / 4u / 64u; ++i) tab4[i * 64u + (uint32_t)lane] = make_uint4(~0u, ~0u, ~0u, ~0u);
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ D2Row d2_row(uint32_t r, int lane, uint32_t pq0, uint32_t pq1, uint32_t* __restrict__ tab,
                                        uint32_t part, uint32_t nparts)
{
    D2Row o;
    const bool has1 = lane < D2_NPAIR - 64;
    o.k0 = d2_key(r, pq0 >> 4, pq0 & 15u);
    o.k1 = has1 ? d2_key(r, pq1 >> 4, pq1 & 15u) : 0xFFFFFFFFu;
    const uint32_t diff = r ^ (r << 2);
    const uint32_t first = ((diff | (diff >> 1)) & 0x55555554u) | 1u;                   // bit 2x: base x starts a run
    auto canonical = [&](uint32_t pq) {
        const uint32_t p = pq >> 4, q = pq & 15u;
        const bool fq = (first >> (2u * q)) & 1u, fp = (first >> (2u * p)) & 1u;
        return fp && (fq || p + 1u == q);                                                // (p + 1 == q and q not first: same run)
    };
    bool keep0 = canonical(pq0), keep1 = has1 && canonical(pq1);
    // (the table is all ones when a row begins: d2_tab_init once, then every row puts back what it took)
    // (a row's 14-mers agree in their low letters whenever both deletions lie behind them: the slot must come from all 28 bits)
    // (the same mixed key names the entry's bucket afterwards, dj_codec.hpp)
    const uint32_t x0 = djc::mix<28>(o.k0), x1 = djc::mix<28>(o.k1 & 0x0FFFFFFFu);
    o.z0 = x0; o.z1 = x1;
    const uint32_t mine0 = (uint32_t)lane << 18 | (x0 & 0x3FFFFu), mine1 = (uint32_t)(64 + lane) << 18 | (x1 & 0x3FFFFu);
    const bool put0 = keep0, put1 = keep1;
    if (put0) atomicMin(&tab[x0 >> 18], mine0);
    if (put1) atomicMin(&tab[x1 >> 18], mine1);
    __builtin_amdgcn_wave_barrier();
    const uint32_t got0 = tab[x0 >> 18], got1 = tab[x1 >> 18];
    __builtin_amdgcn_wave_barrier();
    // the slots this row used, all ones again: two 4-byte stores per lane where clearing the table took four 16-byte ones -
    // the LDS arrays were the busiest unit of both row passes (31 M cycles a pass, profiles/r04_d: 60 % of the pass)
    if (put0) tab[x0 >> 18] = ~0u;
    if (put1) tab[x1 >> 18] = ~0u;
    const bool same0 = ((got0 ^ mine0) & 0x3FFFFu) == 0u, same1 = ((got1 ^ mine1) & 0x3FFFFu) == 0u;
    const bool lost0 = keep0 && !same0, lost1 = keep1 && !same1;         // the slot went to another 14-mer
    keep0 = keep0 && (got0 == mine0 || !same0);
    keep1 = keep1 && (got1 == mine1 || !same1);
    const unsigned long long u0 = __ballot(lost0), u1 = __ballot(lost1);
    for (unsigned long long w = u0; w; w &= w - 1ull) {
        const int src = __builtin_ctzll(w);
        const uint32_t ks = (uint32_t)__builtin_amdgcn_readlane((int)o.k0, src);
        if (lost0 && lane > src && o.k0 == ks) keep0 = false;
        if (lost1 && o.k1 == ks) keep1 = false;
    }
    for (unsigned long long w = u1; w; w &= w - 1ull) {
        const int src = __builtin_ctzll(w);
        const uint32_t ks = (uint32_t)__builtin_amdgcn_readlane((int)o.k1, src);
        if (lost1 && lane > src && o.k1 == ks) keep1 = false;
    }
    // a share of the work (one GPU of several): the 14-mers whose hash falls into this part - a group is whole or absent
    if (nparts > 1u) {
        keep0 = keep0 && d2_part(o.k0, nparts) == part;
        keep1 = keep1 && d2_part(o.k1, nparts) == part;
    }
    o.keep0 = keep0; o.keep1 = keep1;
    return o;
}

// r without its base p: a 15-mer in 30 bits
__device__ __forceinline__ uint32_t d1_key(uint32_t r, uint32_t p)
{
    const uint32_t lo = r & ((1u << (2u * p)) - 1u);
    const uint32_t hi = (uint32_t)((unsigned long long)r >> (2u * p + 2u));
    return lo | (hi << (2u * p));
}

// x[:15] against y without its letter `del`, equal but for one substituted letter at place `sub` of x[:15]?  late: the
// substitution lies at or behind the deleted letter's place (then y's letter is y[sub + 1]), else in front of it.
// Straight-line code (no branch: the reporting rule runs for every meeting, 64 different pairs a wave, and every branch it
// had was taken by some lane - 22 of them cost more than the arithmetic); without a relation: del = 1, sub = 0, late = false.
__device__ __forceinline__ bool d2_shifted(uint32_t x, uint32_t y, uint32_t& del, uint32_t& sub, bool& late)
{
    const uint32_t u = x & 0x3FFFFFFFu;
    const uint32_t x0 = (u ^ y) & 0x3FFFFFFFu, x1 = (u ^ (y >> 2)) & 0x3FFFFFFFu;
    const uint32_t nz0 = (x0 | (x0 >> 1)) & 0x15555555u;               // place p: x[p] != y[p]
    const uint32_t nz1 = (x1 | (x1 >> 1)) & 0x15555555u;               // place p: x[p] != y[p + 1]
    const uint32_t f0 = nz0 ? (uint32_t)__builtin_ctz(nz0) >> 1 : 15u;
    const uint32_t behind = nz1 & ~((1u << (2u * f0)) - 1u);
    const bool ok_late = __popc(behind) == 1;
    const uint32_t sub_late = behind ? (uint32_t)__builtin_ctz(behind) >> 1 : 0u;
    const uint32_t rest = nz0 & (nz0 - 1u);
    const uint32_t j = rest ? (uint32_t)__builtin_ctz(rest) >> 1 : 15u;
    const bool ok_early = nz0 != 0u && (nz1 & ~((1u << (2u * j)) - 1u)) == 0u;
    const bool ok = ok_late || ok_early;
    late = ok_late;
    del = ok_late ? f0 : (ok_early ? j : 1u);
    sub = ok_late ? sub_late : (ok_early ? f0 : 0u);
    return ok;
}

// Which of the 14-mers a pair shares reports it.  A function of the two barcodes alone (a = the lower row), so that every
// group the pair meets in decides alike, and always one of the shared 14-mers:
//   1. at most two differing letters: the 14-mer without them (one differing letter: without it and letter 0, or 1);
//   2. a without letter i == b without letter j for some i, j (one insertion + one deletion, which includes the forms through
//      a[:-1] / b[:-1] alone): that 15-mer without its first letter.  With lcp / lcs the common prefix / suffix of a and b,
//      i <= j needs i <= lcp, j >= 15 - lcs and a[x + 1] == b[x] for x in [i, j): the narrowest such interval decides;
//      j < i likewise with the roles swapped;
//   3. a[:-1] against b without a letter (or b[:-1] against a without one), equal but for one substituted letter - what is
//      left of the forms through a[:-1] / b[:-1]: the 14-mer without that letter and the dropped / deleted one.
// These are all the ways to dmin(a, b) <= 2: ed(a, b) <= 2 between two 16-mers is at most two substitutions (1) or one
// insertion and one deletion (2); ed(a[:-1], b) <= 2 between a 15-mer and a 16-mer is one insertion (2, with i = 15) or one
// insertion and one substitution (3) - and the tests behind 2 and 3 find the relation whenever it exists (the narrowest
// interval; the two places the shift can sit relative to the substituted letter).  So a pair that none of them names is no
// edge and is not even verified.  Returns 1: k is that 14-mer; 0: it is not, or there is no such relation.
__device__ __forceinline__ int d2_reports(uint32_t a, uint32_t b, uint32_t k)
{
    // every relation's 14-mer is computed, the first relation that holds (in the order above) names the reporter: no branch
    const uint32_t x = a ^ b;                                          // (a != b)
    const uint32_t nz = (x | (x >> 1)) & 0x55555555u;
    const uint32_t h = (uint32_t)__popc(nz);
    const uint32_t lcp = (uint32_t)__builtin_ctz(nz) >> 1, lcs = (uint32_t)__builtin_clz(nz) >> 1;
    // 1. at most two differing letters
    const bool c1 = h <= 2u;
    const uint32_t s1 = lcp, s2 = h == 2u ? 15u - lcs : (s1 == 0u ? 1u : 0u);
    const uint32_t k1 = d2_key(a, s1 < s2 ? s1 : s2, s1 < s2 ? s2 : s1);
    // 2. one insertion + one deletion
    const uint32_t far = 15u - lcs;                                    // first position from which on the tails agree
    const uint32_t near = lcp < far ? lcp : far;
    const uint32_t span = ((1u << (2u * far)) - 1u) & ~((1u << (2u * near)) - 1u);      // letters near .. far - 1
    const bool c2a = (((a >> 2) ^ b) & span) == 0u;                    // i = near <= j = far
    const bool c2b = (((b >> 2) ^ a) & span) == 0u;                    // j = near < i = far
    const uint32_t k2 = d1_key(a, c2a ? near : far) >> 2;
    // 3. the forms through a[:-1] / b[:-1] with one more edit: x[:15] equals y without one letter but for one substituted
    //    letter (the shift of the deleted letter sits either in front of the substitution or behind it): the 14-mer without
    //    the substituted letter and the dropped last one / the deleted one
    uint32_t del_a, sub_a, del_b, sub_b; bool late_a, late_b;
    const bool c3a = d2_shifted(a, b, del_a, sub_a, late_a);
    const bool c3b = d2_shifted(b, a, del_b, sub_b, late_b);
    const uint32_t p3 = c3a ? sub_a : (late_b ? del_b : sub_b), q3 = c3a ? 15u : (late_b ? sub_b + 1u : del_b);
    const uint32_t k3 = d2_key(a, p3, q3);
    const uint32_t want = c1 ? k1 : ((c2a || c2b) ? k2 : k3);
    return (c1 || c2a || c2b || c3a || c3b) && want == k ? 1 : 0;     // none of the relations holds: dmin(a, b) > 2, no edge
}

// thr 1: dmin(a, b) <= 1 means one substituted letter, or a without letter i == b without letter j (which covers the forms
// through a[:-1] / b[:-1]); either way the two share the 15-mer that is left, and the narrowest (i, j) names one of them
// (for a substitution at s: i = j = s).  1: k is that 15-mer; 0: it is not, or the pair is no edge at all.
__device__ __forceinline__ int d1_reports(uint32_t a, uint32_t b, uint32_t k)
{
    const uint32_t x = a ^ b;                                          // (a != b)
    const uint32_t nz = (x | (x >> 1)) & 0x55555555u;
    const uint32_t lcp = (uint32_t)__builtin_ctz(nz) >> 1, lcs = (uint32_t)__builtin_clz(nz) >> 1;
    const uint32_t far = 15u - lcs, near = lcp < far ? lcp : far;
    const uint32_t span = ((1u << (2u * far)) - 1u) & ~((1u << (2u * near)) - 1u);
    if ((((a >> 2) ^ b) & span) == 0u) return d1_key(a, near) == k ? 1 : 0;
    if ((((b >> 2) ^ a) & span) == 0u) return d1_key(a, far) == k ? 1 : 0;
    return 0;
}

// ---- level 1 of the grouping (bdg_partition.hpp): a tile of rows per block, run twice.  EMIT = false: how many entries
// the tile has for each coarse bucket (the top l1 bits of the variant's mixed key); EMIT = true: the entries, each at its
// bucket's cursor (the tile's place inside the bucket, from the counts of all tiles).  The two runs see the same rows and
// drop the same repeats, so the places are exact: no atomic on global memory, nothing to size by guessing.
// thr <= 2: one wave per row at a time (d2_row), up to 64 consecutive rows per coalesced load of their barcodes.
template <bool EMIT, uint32_t NB1CAP>
__global__ __launch_bounds__(256)
void k_d2_rows(const uint32_t* __restrict__ ranks, uint32_t n, uint32_t rows_per_tile, uint32_t part, uint32_t nparts, uint32_t l1,
               uint32_t* __restrict__ hist /* [tiles][nb1]: the tile's place inside each bucket */, uint32_t* __restrict__ tot,
               const unsigned long long* __restrict__ base, const uint32_t* __restrict__ geom,
               uint32_t* __restrict__ ent, ulonglong2* __restrict__ kept /* per row: which of its 120 deletion pairs stay */)
{
    __shared__ uint32_t s_tab[EMIT ? 1 : 4][D2_SLOTS];
    __shared__ uint32_t s_h[NB1CAP];                                   // (1024: eight blocks a compute unit; 4096 for inputs of millions of rows)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t nb1 = 1u << l1, zb = 28u - l1;
    const uint32_t row0 = blockIdx.x * rows_per_tile;
    const uint32_t row1 = n - row0 < rows_per_tile ? n : row0 + rows_per_tile;
    if (EMIT && (geom[bdgpart::G_FLAGS] & 1u)) return;                 // (more entries than the caller can index: it cuts smaller)
    for (uint32_t i = threadIdx.x; i < nb1; i += 256u) s_h[i] = EMIT ? (uint32_t)base[i] + hist[(size_t)i * gridDim.x + blockIdx.x] : 0u;
    __syncthreads();
    const uint32_t pq0 = d2_table.pq[lane], pq1 = d2_table.pq[lane < D2_NPAIR - 64 ? 64 + lane : 0];
    if (!EMIT) d2_tab_init(s_tab[wv], lane);
    for (uint32_t chunk = row0 + (uint32_t)wv * 64u; chunk < row1; chunk += 256u) {
        const uint32_t rows = row1 - chunk < 64u ? row1 - chunk : 64u;
        const uint32_t mine = (uint32_t)lane < rows ? ranks[chunk + (uint32_t)lane] : 0u;
        if (EMIT) {
            // the second run does not settle the repeats again: the first left every row's two 64-bit masks of the pairs that stay
            const ulonglong2 km = (uint32_t)lane < rows ? kept[chunk + (uint32_t)lane] : make_ulonglong2(0ull, 0ull);
            for (uint32_t i = 0; i < rows; ++i) {
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)i);
                const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)km.x, (int)i), a1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(km.x >> 32), (int)i);
                const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)km.y, (int)i), b1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(km.y >> 32), (int)i);
                const bool keep0 = ((lane < 32 ? a0 : a1) >> (lane & 31)) & 1u, keep1 = ((lane < 32 ? b0 : b1) >> (lane & 31)) & 1u;
                if (keep0) {
                    const uint32_t z = djc::mix<28>(d2_key(r, pq0 >> 4, pq0 & 15u));
                    ent[atomicAdd(&s_h[z >> zb], 1u)] = djc::enc2(z, zb, (uint32_t)lane, r, pq0);
                }
                if (keep1) {
                    const uint32_t z = djc::mix<28>(d2_key(r, pq1 >> 4, pq1 & 15u));
                    ent[atomicAdd(&s_h[z >> zb], 1u)] = djc::enc2(z, zb, 64u + (uint32_t)lane, r, pq1);
                }
            }
        } else {
            unsigned long long m0 = 0, m1 = 0;
            for (uint32_t i = 0; i < rows; ++i) {
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)i);
                const D2Row o = d2_row(r, lane, pq0, pq1, s_tab[wv], part, nparts);
                if (o.keep0) atomicAdd(&s_h[o.z0 >> zb], 1u);
                if (o.keep1) atomicAdd(&s_h[o.z1 >> zb], 1u);
                const unsigned long long k0 = __ballot(o.keep0), k1 = __ballot(o.keep1);
                if ((uint32_t)lane == i) { m0 = k0; m1 = k1; }
            }
            if ((uint32_t)lane < rows) kept[chunk + (uint32_t)lane] = make_ulonglong2(m0, m1);
        }
    }
    if (!EMIT) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb1; i += 256u) hist[(size_t)i * gridDim.x + blockIdx.x] = s_h[i];      // (one row per bucket: k_part_colscan)
    }
}

// thr <= 1: the one-deletion 15-mers of a row.  Deleting any letter of a run gives the same 15-mer, so the first letter of
// every run is deleted - exactly the distinct ones (about 12 of 16 on random barcodes).  One thread a row.
template <bool EMIT, uint32_t NB1CAP>
__global__ __launch_bounds__(256)
void k_d1_rows(const uint32_t* __restrict__ ranks, uint32_t n, uint32_t rows_per_tile, uint32_t part, uint32_t nparts, uint32_t l1,
               uint32_t* __restrict__ hist, uint32_t* __restrict__ tot, const unsigned long long* __restrict__ base, const uint32_t* __restrict__ geom,
               uint32_t* __restrict__ ent)
{
    __shared__ uint32_t s_h[NB1CAP];
    const uint32_t nb1 = 1u << l1, zb = 30u - l1;
    const uint32_t row0 = blockIdx.x * rows_per_tile;
    const uint32_t row1 = n - row0 < rows_per_tile ? n : row0 + rows_per_tile;
    if (EMIT && (geom[bdgpart::G_FLAGS] & 1u)) return;
    for (uint32_t i = threadIdx.x; i < nb1; i += 256u) s_h[i] = EMIT ? (uint32_t)base[i] + hist[(size_t)i * gridDim.x + blockIdx.x] : 0u;
    __syncthreads();
    for (uint32_t row = row0 + threadIdx.x; row < row1; row += 256u) {
        const uint32_t r = ranks[row];
        const uint32_t diff = r ^ (r << 2);
        const uint32_t first = ((diff | (diff >> 1)) & 0x55555554u) | 1u;
#pragma unroll
        for (uint32_t p = 0; p < 16u; ++p) {
            if (!((first >> (2u * p)) & 1u)) continue;
            const uint32_t k = d1_key(r, p);
            if (nparts > 1u && d2_part(k, nparts) != part) continue;
            const uint32_t z = djc::mix<30>(k);
            if (EMIT) ent[atomicAdd(&s_h[z >> zb], 1u)] = djc::enc1(z, zb, p, r);
            else atomicAdd(&s_h[z >> zb], 1u);
        }
    }
    if (!EMIT) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb1; i += 256u) hist[(size_t)i * gridDim.x + blockIdx.x] = s_h[i];      // (one row per bucket: k_part_colscan)
    }
}

// ---- the consumer, first form: one fine bucket per WAVE, no block barrier anywhere.
// A fine bucket holds the entries whose mixed keys share their top l1 + l2 bits - whole groups, about a hundred entries, at
// most WCAP (a larger one is listed for the block kernel below).  The wave finishes the grouping in its own stretch of LDS:
// a counting pass over WBIN bins named by the next key bits (the LDS atomic that counts also gives the entry its place inside
// the bin; no order is needed), then every entry meets the entries behind it in its bin.  A bin is mostly one group; where
// two variants share a bin the meeting ends at one compare.  Lanes stay full whatever the group sizes: the meetings of the
// whole bucket are numbered through (entry p owns the slots [before_p, before_p + L_p)), the wave takes 64 slots at a time,
// and a slot finds its owner without a search - every owner marks its FIRST slot with its place, and a running maximum
// over the marks (six DPP steps and the carry of the step before) is the owner of every slot.  Which group reports a pair is
// a function of the two barcodes (d2_reports / d1_reports); only meetings that would report are verified (the same Myers
// dmin3 and the same S >= T as on every other path), 64 at a time out of a per-wave queue.  The next bucket's entries are
// loaded into registers before this one is walked.
template <int NDEL, uint32_t WCAP, uint32_t ECAPW>
__global__ __launch_bounds__(256)
void k_d2_pairs_w(const uint32_t* __restrict__ ent, const uint32_t* __restrict__ fstart, uint32_t* __restrict__ geom, uint32_t l1,
                  const uint32_t* __restrict__ ranks, uint32_t row_begin, uint32_t row_end, uint32_t thr, int32_t T,
                  bdg_edge* __restrict__ out, uint64_t cap, unsigned long long* __restrict__ n_edges, uint32_t* __restrict__ ovf)
{
    constexpr int KB = NDEL == 2 ? 28 : 30;
    constexpr uint32_t PER = WCAP / 64u, WBIN = WCAP, LBIN = 31u - (uint32_t)__builtin_clz(WCAP), CH = 256u;
    static_assert(PER % 4u == 0u && (WCAP & (WCAP - 1u)) == 0u, "WCAP: 256, 512, ...");
    __shared__ unsigned long long s_kv[4][WCAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_bin[4][WBIN + 4];
    __shared__ __attribute__((aligned(16))) uint16_t s_before[4][WCAP];      // (a bucket of 256 has fewer than 2^15 meeting slots)
    __shared__ uint16_t s_mark[4][CH];
    static_assert(WCAP <= 256u, "16-bit slot numbers");
    __shared__ EdgeStageT<ECAPW> stages[4];
    __shared__ uint32_t s_qa[4][128], s_qb[4][128];
    __shared__ uint32_t s_ma[4][128], s_mb[4][128], s_mk[4][128];
    __shared__ uint32_t s_cnt[4];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t ne = 0, qn = 0;
    const uint32_t flags = geom[bdgpart::G_FLAGS];
    const uint32_t l2 = geom[bdgpart::G_L2];
    const uint32_t nfb = (flags & 1u) ? 0u : 1u << (l1 + l2);
    const uint32_t zb = (uint32_t)KB - l1, rb = zb - l2;
    const uint32_t lb = rb < LBIN ? rb : LBIN, bin_sh = rb - lb, bin_mask = (1u << lb) - 1u;
    const uint32_t rank_lo = ranks[row_begin], rank_hi = ranks[row_end - 1u];       // (row_begin < row_end: the launcher's check)
    auto verify = [&](uint32_t a, uint32_t b, bool act) {
        const uint32_t d = act ? dmin3(a, b) : 99u;
        bool edge = d <= thr;
        if (__ballot(edge)) edge = edge && (int32_t)qgram_S(a, b) >= T;
        edge_push(edge, a, b, d, stages[wv], ne, lane, out, cap, n_edges);
    };
    uint32_t mn = 0;                                                   // meetings waiting for the reporting rule
    auto report = [&](uint32_t from, bool act) {
        const uint32_t a = s_ma[wv][from + lane], b = s_mb[wv][from + lane], kk = s_mk[wv][from + lane];
        const int rep = act ? (NDEL == 2 ? d2_reports(a, b, kk) : d1_reports(a, b, kk)) : 0;
        const unsigned long long mq = __ballot(rep != 0);
        if (rep != 0) {
            const uint32_t at = qn + lanes_below_u64(mq, lane);
            s_qa[wv][at] = a; s_qb[wv][at] = b;
        }
        qn += (uint32_t)__popcll(mq);
        __builtin_amdgcn_wave_barrier();
        if (qn >= 64u) {
            qn -= 64u;
            verify(s_qa[wv][qn + lane], s_qb[wv][qn + lane], true);
            __builtin_amdgcn_wave_barrier();
        }
    };
    const uint32_t GW = gridDim.x * 4u;
    uint32_t fb = blockIdx.x * 4u + (uint32_t)wv;
    uint32_t start = 0, cnt = 0, e[PER];
    auto fetch = [&](uint32_t f, uint32_t& s, uint32_t& c, uint32_t (&ee)[PER]) {
        s = 0; c = 0;
        if (f < nfb) { s = fstart[f]; c = fstart[f + 1u] - s; }
        s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s); c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) ee[j] = (c <= WCAP && j * 64u + (uint32_t)lane < c) ? ent[s + j * 64u + (uint32_t)lane] : 0u;
    };
    fetch(fb, start, cnt, e);
    while (fb < nfb) {
        uint32_t nstart, ncnt, nx[PER];
        fetch(fb + GW, nstart, ncnt, nx);                               // (travels while this bucket is walked)
        if (cnt > WCAP) {
            if (lane == 0) ovf[1u + atomicAdd(&geom[bdgpart::G_OVF], 1u)] = fb;
        } else if (cnt >= 2u) {
            const uint32_t b1 = fb >> l2;
            uint4* const bin4 = reinterpret_cast<uint4*>(s_bin[wv]);
            uint2* const bef4 = reinterpret_cast<uint2*>(s_before[wv]);        // (four 16-bit words)
#pragma unroll
            for (uint32_t i = 0; i < PER / 4u; ++i) bin4[i * 64u + (uint32_t)lane] = make_uint4(0u, 0u, 0u, 0u);
            __builtin_amdgcn_wave_barrier();
            uint32_t bn[PER], rk[PER];
#pragma unroll
            for (uint32_t j = 0; j < PER; ++j) {
                bn[j] = (e[j] >> bin_sh) & bin_mask; rk[j] = 0;
                if (j * 64u + (uint32_t)lane < cnt) rk[j] = atomicAdd(&s_bin[wv][bn[j]], 1u);
            }
            __builtin_amdgcn_wave_barrier();
            {   // the bins' starts: a lane takes PER consecutive bins
                uint4 c4[PER / 4u];
                uint32_t sum = 0;
#pragma unroll
                for (uint32_t i = 0; i < PER / 4u; ++i) { c4[i] = bin4[(uint32_t)lane * (PER / 4u) + i]; sum += c4[i].x + c4[i].y + c4[i].z + c4[i].w; }
                uint32_t run = wave_incl_scan(sum) - sum;
#pragma unroll
                for (uint32_t i = 0; i < PER / 4u; ++i) {
                    uint4 o;
                    o.x = run; run += c4[i].x; o.y = run; run += c4[i].y; o.z = run; run += c4[i].z; o.w = run; run += c4[i].w;
                    bin4[(uint32_t)lane * (PER / 4u) + i] = o;
                }
                if (lane == 63) s_bin[wv][WBIN] = run;                 // (= cnt)
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (uint32_t j = 0; j < PER; ++j) {
                if (j * 64u + (uint32_t)lane >= cnt) continue;
                const uint32_t pos = s_bin[wv][bn[j]] + rk[j], end = s_bin[wv][bn[j] + 1u];
                uint32_t k, r;
                if (NDEL == 2) djc::dec2(e[j], b1, zb, d2_table.pq[(e[j] >> zb) & 127u], k, r); else djc::dec1(e[j], b1, zb, k, r);
                s_kv[wv][pos] = (unsigned long long)k << 32 | r;
                s_before[wv][pos] = (uint16_t)(end - pos - 1u);         // (for now: L, the entries behind this one in its bin)
            }
            __builtin_amdgcn_wave_barrier();
            // the meetings numbered through: a lane takes PER consecutive places
            uint32_t L[PER], bf[PER], sum = 0;
#pragma unroll
            for (uint32_t i = 0; i < PER / 4u; ++i) {
                const uint2 v = bef4[(uint32_t)lane * (PER / 4u) + i];
                L[4u * i] = v.x & 0xFFFFu; L[4u * i + 1u] = v.x >> 16; L[4u * i + 2u] = v.y & 0xFFFFu; L[4u * i + 3u] = v.y >> 16;
            }
#pragma unroll
            for (uint32_t i = 0; i < PER; ++i) { if ((uint32_t)lane * PER + i >= cnt) L[i] = 0u; sum += L[i]; }
            const uint32_t incl = wave_incl_scan(sum);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total) {
                uint32_t run = incl - sum;
#pragma unroll
                for (uint32_t i = 0; i < PER; ++i) { bf[i] = run; run += L[i]; }
#pragma unroll
                for (uint32_t i = 0; i < PER / 4u; ++i) bef4[(uint32_t)lane * (PER / 4u) + i] = make_uint2(bf[4u * i] | bf[4u * i + 1u] << 16, bf[4u * i + 2u] | bf[4u * i + 3u] << 16);
                uint32_t carry = 0;
                for (uint32_t cb = 0; cb < total; cb += CH) {
#pragma unroll
                    for (uint32_t i = 0; i < CH / 128u; ++i) reinterpret_cast<uint32_t*>(s_mark[wv])[i * 64u + (uint32_t)lane] = 0u;
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (uint32_t i = 0; i < PER; ++i) if (L[i] && bf[i] - cb < CH) s_mark[wv][bf[i] - cb] = (uint16_t)((uint32_t)lane * PER + i + 1u);
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t x0 = 0; x0 < CH && cb + x0 < total; x0 += 64u) {
                        const uint32_t x = cb + x0 + (uint32_t)lane;
                        const bool act = x < total;
                        uint32_t own = wave_incl_max((uint32_t)s_mark[wv][x0 + (uint32_t)lane]);
                        own = own > carry ? own : carry;
                        carry = (uint32_t)__builtin_amdgcn_readlane((int)own, 63);
                        uint32_t a = 0, b = 0, kk = 0;
                        bool on = false;
                        if (act) {
                            const uint32_t p1 = own - 1u, p2 = p1 + 1u + (x - (uint32_t)s_before[wv][p1]);
                            const unsigned long long kv1 = s_kv[wv][p1], kv2 = s_kv[wv][p2];
                            kk = (uint32_t)(kv1 >> 32);
                            if (kk == (uint32_t)(kv2 >> 32)) {         // (the same variant, not just the same bin)
                                const uint32_t v1 = (uint32_t)kv1, v2 = (uint32_t)kv2;
                                const bool lower = v1 < v2;            // (a row has one entry per variant: v1 != v2)
                                a = lower ? v1 : v2; b = lower ? v2 : v1;
                                on = a >= rank_lo && a <= rank_hi;
                            }
                        }
                        // real meetings wait in a queue of their own, so that the reporting rule - the dearest part of a
                        // meeting, and both of its branches run whenever a wave holds both kinds - always sees 64 of them
                        // (a third of the slots are bin neighbours of another variant or the tail of a bucket's last step)
                        const unsigned long long mm = __ballot(on);
                        if (on) {
                            const uint32_t at = mn + lanes_below_u64(mm, lane);
                            s_ma[wv][at] = a; s_mb[wv][at] = b; s_mk[wv][at] = kk;
                        }
                        mn += (uint32_t)__popcll(mm);
                        __builtin_amdgcn_wave_barrier();
                        if (mn >= 64u) { mn -= 64u; report(mn, true); }
                    }
                    __builtin_amdgcn_wave_barrier();                     // (the marks are rewritten next)
                }
            }
            __builtin_amdgcn_wave_barrier();                             // (the bucket's arrays are rewritten next)
        }
        fb += GW; start = nstart; cnt = ncnt;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) e[j] = nx[j];
    }
    if (mn) { const uint32_t left = mn; mn = 0; report(0u, (uint32_t)lane < left); }
    if (qn) {
        const bool act = (uint32_t)lane < qn;
        verify(act ? s_qa[wv][lane] : 0u, act ? s_qb[wv][lane] : 1u, act);
    }
    edge_finish<4>(stages, ne, s_cnt, &s_base, out, cap, n_edges);
}

// ---- the consumer, second form: one fine bucket per BLOCK, for the buckets the wave kernel listed as too large for a wave
// (list != nullptr) or for all of them (list == nullptr: a cross-check).
// The block finishes the grouping in LDS: a counting pass over NBIN bins named by the next key bits (an LDS atomic gives an
// entry its place inside its bin, no order is needed), then every entry meets the entries behind it in its bin.  A bin is
// mostly one group; where two variants share a bin the meeting ends at one compare.  The walk itself is round 3's: a wave
// takes 64 consecutive places, lane l's entry meets the L_l entries behind it, and the wave walks the SUM of the meetings
// 64 at a time (a meeting's owner is found in the running sums), so lanes stay full whatever the group sizes; which group
// reports a pair is a function of the two barcodes (d2_reports / d1_reports); only meetings that would report are verified
// (the same Myers dmin3 and the same S >= T as on every other path), 64 at a time out of a per-wave queue.
// A bucket larger than CAP (the hash spreads keys evenly, so this is for adversarial inputs) is taken in shares of the low
// key bits, each share through the same code; a group never exceeds 1920 (thr <= 2) / 64 (thr <= 1) entries, CAP >= 2048.
template <int NDEL, int THREADS, uint32_t CAP, uint32_t ECAPW>
__global__ __launch_bounds__(THREADS)
void k_d2_pairs(const uint32_t* __restrict__ ent, const uint32_t* __restrict__ fstart, uint32_t* __restrict__ geom, uint32_t l1,
                const uint32_t* __restrict__ ranks, uint32_t row_begin, uint32_t row_end, uint32_t thr, int32_t T,
                bdg_edge* __restrict__ out, uint64_t cap, unsigned long long* __restrict__ n_edges, const uint32_t* __restrict__ list)
{
    constexpr int KB = NDEL == 2 ? 28 : 30;
    constexpr int NW = THREADS / 64;
    constexpr uint32_t NBIN = CAP, LBIN = 31u - (uint32_t)__builtin_clz(CAP), PER = NBIN / THREADS;
    static_assert((CAP & (CAP - 1u)) == 0u && CAP >= 2048u && NBIN % THREADS == 0u, "CAP: a power of two that holds the largest group");
    __shared__ uint32_t s_k[CAP], s_v[CAP];
    __shared__ uint32_t s_bin[NBIN + 1];
    __shared__ EdgeStageT<ECAPW> stages[NW];
    __shared__ uint32_t s_qa[NW][128], s_qb[NW][128];
    __shared__ uint32_t s_incl[NW][64];
    __shared__ uint32_t s_cnt[NW], s_w[NW + 1];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t ne = 0, qn = 0;
    const uint32_t flags = geom[bdgpart::G_FLAGS];
    const uint32_t l2 = geom[bdgpart::G_L2];
    const uint32_t nfb = (flags & 1u) ? 0u : 1u << (l1 + l2);
    const uint32_t zb = (uint32_t)KB - l1;                              // key bits an entry carries
    const uint32_t rb = zb - l2;                                        // ... of which these are not spelled by the fine bucket
    const uint32_t lb = rb < LBIN ? rb : LBIN, bin_sh = rb - lb, bin_mask = (1u << lb) - 1u;
    const uint32_t rank_lo = ranks[row_begin], rank_hi = ranks[row_end - 1u];       // (row_begin < row_end: the launcher's check)
    auto verify = [&](uint32_t a, uint32_t b, bool act) {
        const uint32_t d = act ? dmin3(a, b) : 99u;
        bool edge = d <= thr;
        if (__ballot(edge)) edge = edge && (int32_t)qgram_S(a, b) >= T;
        edge_push(edge, a, b, d, stages[wv], ne, lane, out, cap, n_edges);
    };
    // the entries of [start, start + cnt) whose low sb key bits spell `share` (sb = 0: all of them), at most CAP: into the
    // bins, then the walk
    auto process = [&](uint32_t start, uint32_t cnt, uint32_t b1, uint32_t sb, uint32_t share) {
        const uint32_t smask = (1u << sb) - 1u;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) s_bin[threadIdx.x * PER + j] = 0u;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cnt; i += THREADS) {
            const uint32_t e = ent[start + i];
            if ((e & smask) == share) atomicAdd(&s_bin[(e >> bin_sh) & bin_mask], 1u);
        }
        __syncthreads();
        uint32_t c[PER], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) { c[j] = s_bin[threadIdx.x * PER + j]; sum += c[j]; }
        uint32_t held;
        uint32_t run = bdgpart::block_excl_scan<THREADS>(sum, s_w, held);
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) { s_bin[threadIdx.x * PER + j] = run; run += c[j]; }
        if (threadIdx.x == 0) s_bin[NBIN] = held;
        __syncthreads();
        // (held <= CAP: the caller's check)  second pass: decode, and place every entry - the bins' starts count up to their ends
        for (uint32_t i = threadIdx.x; i < cnt; i += THREADS) {
            const uint32_t e = ent[start + i];
            if ((e & smask) != share) continue;
            const uint32_t at = atomicAdd(&s_bin[(e >> bin_sh) & bin_mask], 1u);
            uint32_t k, r;
            if (NDEL == 2) djc::dec2(e, b1, zb, d2_table.pq[(e >> zb) & 127u], k, r); else djc::dec1(e, b1, zb, k, r);
            s_k[at] = k; s_v[at] = r;
        }
        __syncthreads();
        // now s_bin[b] = END of bin b (= start of bin b + 1); an entry's place inside its bin is its place minus the start
        for (uint32_t wbase = (uint32_t)wv * 64u; wbase < held; wbase += (uint32_t)NW * 64u) {
            const uint32_t pos = wbase + (uint32_t)lane;
            const bool have = pos < held;
            const uint32_t k = have ? s_k[pos] : 0u;
            const uint32_t bin = (djc::mix<KB>(k) >> bin_sh) & bin_mask;
            const uint32_t L = have ? s_bin[bin] - pos - 1u : 0u;       // entries behind this one in its bin
            const uint32_t incl = wave_incl_scan(L);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total == 0u) continue;
            s_incl[wv][lane] = incl;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t x0 = 0; x0 < total; x0 += 64u) {
                const uint32_t x = x0 + (uint32_t)lane;
                const bool act = x < total;
                uint32_t o = 0;                                        // owner: the number of lanes whose running sum is <= x
#pragma unroll
                for (uint32_t s = 32; s >= 1; s >>= 1) if (s_incl[wv][o + s - 1u] <= x) o += s;
                o = act ? o : 0u;
                const uint32_t before = o ? s_incl[wv][o - 1u] : 0u;
                const uint32_t p1 = wbase + o, p2 = p1 + (x - before) + 1u;
                uint32_t a = 0, b = 0, kk = 0;
                bool on = false;
                if (act) {
                    kk = s_k[p1];
                    if (kk == s_k[p2]) {                               // (the same variant, not just the same bin)
                        const uint32_t v1 = s_v[p1], v2 = s_v[p2];
                        const bool lower = v1 < v2;                    // (a row has one entry per variant: v1 != v2)
                        a = lower ? v1 : v2; b = lower ? v2 : v1;
                        on = a >= rank_lo && a <= rank_hi;
                    }
                }
                const int rep = on ? (NDEL == 2 ? d2_reports(a, b, kk) : d1_reports(a, b, kk)) : 0;
                const unsigned long long mq = __ballot(rep != 0);
                if (rep != 0) {
                    const uint32_t at = qn + lanes_below_u64(mq, lane);
                    s_qa[wv][at] = a; s_qb[wv][at] = b;
                }
                qn += (uint32_t)__popcll(mq);
                __builtin_amdgcn_wave_barrier();
                if (qn >= 64u) {
                    qn -= 64u;
                    verify(s_qa[wv][qn + lane], s_qb[wv][qn + lane], true);
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __builtin_amdgcn_wave_barrier();                             // (the window's running sums are rewritten next)
        }
        __syncthreads();                                                 // (the bucket's arrays are rewritten next)
    };
    const uint32_t nwork = (flags & 1u) ? 0u : (list ? geom[bdgpart::G_OVF] : nfb);
    for (uint32_t w = blockIdx.x; w < nwork; w += gridDim.x) {
        const uint32_t fb = list ? list[1u + w] : w;
        const uint32_t start = fstart[fb], cnt = fstart[fb + 1u] - start;
        if (cnt < 2u) continue;
        const uint32_t b1 = fb >> l2;
        if (cnt <= CAP) { process(start, cnt, b1, 0u, 0u); continue; }
        // cold path: shares by the low sb key bits, sb grown until every share fits (counted first: a share is walked once)
        uint32_t sb = 1;
        while ((cnt >> sb) > CAP / 2u && sb < LBIN) ++sb;
        bool fits = false;
        for (; sb <= LBIN && !fits; ++sb) {
            for (uint32_t i = threadIdx.x; i < (1u << sb); i += THREADS) s_k[i] = 0u;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < cnt; i += THREADS) atomicAdd(&s_k[ent[start + i] & ((1u << sb) - 1u)], 1u);
            __syncthreads();
            uint32_t worst = 0;
            for (uint32_t i = threadIdx.x; i < (1u << sb); i += THREADS) worst = s_k[i] > worst ? s_k[i] : worst;
            fits = __syncthreads_or(worst > CAP) == 0;
        }
        if (!fits) { if (threadIdx.x == 0) atomicOr(&geom[bdgpart::G_FLAGS], 2u); continue; }       // (reported by the launcher's caller)
        --sb;
        for (uint32_t share = 0; share < (1u << sb); ++share) process(start, cnt, b1, sb, share);
    }
    if (qn) {
        const bool act = (uint32_t)lane < qn;
        verify(act ? s_qa[wv][lane] : 0u, act ? s_qb[wv][lane] : 1u, act);
    }
    edge_finish<NW>(stages, ne, s_cnt, &s_base, out, cap, n_edges);
}

}  // namespace

// ---------------------------------------------------------------------------
static uint32_t d2_pairs_blocks_per_cu()
{
    static const uint32_t v = [] { const char* e = getenv("BADGER_AMD_D2_PAIRS_BLOCKS"); const int x = e ? atoi(e) : 4; return (uint32_t)(x < 1 ? 1 : (x > 8 ? 8 : x)); }();
    return v;                                                          // (k_d2_pairs_w: 36 KB of LDS a block of 4 waves at 256 entries a wave)
}

// once per context: the device's compute units and how many blocks of the join kernels a unit holds (resident grids are sized from these)
static int graph_props(bdg_ctx* ctx)
{
    if (!ctx->g_cus) {
        hipDeviceProp_t prop;
        BDG_HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
        int per_cu = 0, per_cu_w = 0;
        BDG_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_graph_qjoin<QJ_W>, 256, 0));
        // slice size per wave, measured at 500 K rows / thr 2: 16 K rows x 2 waves per block 11.1 ms, 8 K x 4 12.2, 32 K x 1 15.8, 4 K x 4 18.8
        ctx->g_qjw_variant = 1;
        if (const char* e = getenv("BADGER_AMD_QJ_VARIANT")) ctx->g_qjw_variant = atoi(e);      // (for measurements)
        if (ctx->g_qjw_variant < 0 || ctx->g_qjw_variant > 3) ctx->g_qjw_variant = 1;
        switch (ctx->g_qjw_variant) {
        case 1:  BDG_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_w, k_graph_qjoin_w<16384, 2, true>, 128, 0)); break;
        case 2:  BDG_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_w, k_graph_qjoin_w<32768, 1, true>, 64, 0)); break;
        case 3:  BDG_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_w, k_graph_qjoin_w<4096, 4, true>, 256, 0)); break;
        default: BDG_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_w, k_graph_qjoin_w<8192, 4, true>, 256, 0)); break;
        }
        ctx->g_cus = prop.multiProcessorCount; ctx->g_qj_per_cu = per_cu < 1 ? 1 : per_cu; ctx->g_qjw_per_cu = per_cu_w < 1 ? 1 : per_cu_w;
    }
    return BDG_OK;
}

// what the join kernels of the last launch reported (waits for the stream): bit 0 - a round held more entries than can be
// indexed, bit 1 - a fine bucket could not be taken apart (2048 shares of its low key bits, one of them above the LDS capacity)
int bdg_graph_join_flags(bdg_ctx* ctx, uint32_t* flags)
{
    *flags = 0;
    if (!ctx->g_dj_geom) return BDG_OK;
    BDG_HIP_TRY(ctx, hipMemcpyAsync(flags, ctx->g_dj_geom + bdgpart::G_FLAGS, 4, hipMemcpyDeviceToHost, ctx->stream));
    BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BDG_OK;
}
int bdg_graph_flags_error(bdg_ctx* ctx, uint32_t flags)
{
    if (flags & 1u) return bdg_fail(ctx, BDG_E_CAPACITY, "deletion-variant join: a round holds more entries than 32 bits index (set BADGER_AMD_D2_ROUNDS higher)");
    return bdg_fail(ctx, BDG_E_CAPACITY, "deletion-variant join: a bucket of variants could not be taken apart");
}

// which path bdg_graph_launch takes: 1 all-pairs sweep, 2 neighbourhood probes, 3 / 4 q-gram join, 5 / 6 deletion-variant join
// over 14-mers (thr <= 2) / 15-mers (thr <= 1)
int bdg_graph_plan(const bdg_ctx* ctx, uint32_t n, uint32_t thr)
{
    if (ctx->graph_algo) return ctx->graph_algo;
    if (thr == 1) return n >= ctx->g_d1_min_rows ? 6 : 2;
    if (thr == 2 && n >= ctx->g_d2_min_rows) return 5;                 // (any n: large inputs are taken in rounds)
    if (thr >= 2 && n < (1u << 25)) return 3;
    return 1;
}

// part / nparts: only the deletion-variant join looks at them (its share of the 14-mer groups); the other paths share by rows
int bdg_graph_launch(bdg_ctx* ctx, const uint32_t* d_ranks, uint32_t n, uint32_t row_begin, uint32_t row_end,
                     uint32_t thr, int32_t qgram_T, bdg_edge* d_out, uint64_t cap, uint64_t* d_n_edges, uint32_t part, uint32_t nparts)
{
    hipStream_t st = ctx->stream;
    ctx->g_dj_geom = nullptr;                                          // (bdg_graph_status speaks about this call)
    BDG_HIP_TRY(ctx, hipMemsetAsync(d_n_edges, 0, 8, st));
    if (n < 2 || row_begin >= row_end) return BDG_OK;
    if (thr > 16) return bdg_fail(ctx, BDG_E_ARG, "thr must be <= 16");
    if (qgram_T < 1) return bdg_fail(ctx, BDG_E_ARG, "qgram_T must be >= 1 (index.py:22-24 never yields less)");
    const int plan = bdg_graph_plan(ctx, n, thr);
    const bool probe = plan == 2;
    if (ctx->graph_algo == 2 && thr != 1) return bdg_fail(ctx, BDG_E_ARG, "probe path needs thr == 1");
    // q-gram join: any thr; the automatic choice for thr >= 3 (row << 4 | position and (j + 1) << 7 | count must fit 32 bits)
    const bool qjoin = plan == 3 || plan == 4;
    if ((ctx->graph_algo == 3 || ctx->graph_algo == 4) && n >= (1u << 25)) return bdg_fail(ctx, BDG_E_ARG, "q-gram join needs n < 2^25");
    int rc;
    // deletion-variant join: thr <= 2 only (what makes it complete); the automatic choice for thr 2
    const bool d2join = plan == 5 || plan == 6;
    const bool one_deletion = plan == 6;
    if (ctx->graph_algo == 5 && thr > 2) return bdg_fail(ctx, BDG_E_ARG, "deletion-variant join needs thr <= 2");
    if (ctx->graph_algo == 6 && thr > 1) return bdg_fail(ctx, BDG_E_ARG, "the one-deletion join needs thr <= 1");
    if (nparts == 0 || part >= nparts) return bdg_fail(ctx, BDG_E_ARG, "part outside [0, nparts)");
    if (d2join) {
        // entries: about 71 per row on random barcodes, 120 at most (one deletion: 12, 16 at most).  They are grouped by
        // their variant in two bucket levels and one pass in LDS (bdg_partition.hpp, k_d2_pairs) - no sort, no count the host
        // waits for: the buffers hold the most a round can emit, the fine bucket count is chosen on the device from the true
        // total.  A large input is taken in several rounds, each over its share of the variant groups (the same cut that
        // gives several GPUs their parts), so that a round's entries can be indexed with 32 bits whatever n is.
        auto* cnt = reinterpret_cast<unsigned long long*>(d_n_edges);
        if ((rc = graph_props(ctx))) return rc;
        const unsigned long long per_row_max = one_deletion ? 16ull : 120ull, per_row_est = one_deletion ? 12ull : 72ull;
        // rounds: so that a round expects at most D2_ROUND_ENTRIES entries.  Its buffers hold whatever it can emit - every
        // entry of every row - up to what 32 bits index; only beyond that (thr 2: from 35 M rows on) can a round's share of
        // the groups fail to fit, which the device reports (geom flags) and the loop below asks after each such round.
        uint32_t rounds = (uint32_t)(((unsigned long long)n * per_row_est / nparts + D2_ROUND_ENTRIES - 1) / D2_ROUND_ENTRIES);
        if (const char* e = getenv("BADGER_AMD_D2_ROUNDS")) rounds = (uint32_t)std::max(1, atoi(e));      // (for tests)
        if (rounds < 1) rounds = 1;
        if ((unsigned long long)nparts * rounds > 0xFFFFFFFFull) return bdg_fail(ctx, BDG_E_ARG, "too many parts");
        const uint32_t keybits = one_deletion ? 30u : 28u;
        const unsigned long long cap_ent = std::min((unsigned long long)n * per_row_max, 0xFFFFFFF0ull);
        const bool may_overflow = (unsigned long long)n * per_row_max > cap_ent;
        for (uint32_t round = 0; round < rounds; ++round) {
            const uint32_t sub = part * rounds + round, nsub = nparts * rounds;
            const unsigned long long est = (unsigned long long)n * per_row_est / nsub + 1ull;
            // Coarse buckets: as few as leave the second level (at most 4,096 sub-buckets each) able to cut fine buckets of
            // `target` entries - every further coarse bucket is one more cursor the second row pass scatters its 4-byte
            // stores over (4 M rows: 9 bits instead of 11 take that pass from 2.7 to 1.7 ms and the join from 7.2 to 6.6;
            // rounds 3-4 sized them for at most 256 K entries each)
            uint32_t fine_bits = 0;
            while (fine_bits < 24u && (est >> fine_bits) > 160ull) ++fine_bits;
            uint32_t l1 = fine_bits > 20u ? std::min(12u, fine_bits - 12u) : 8u;
            if (const char* e = getenv("BADGER_AMD_DJ_L1")) l1 = (uint32_t)std::min(12, std::max(8, atoi(e)));      // (for measurements)
            const uint32_t nb1 = 1u << l1;
            // sub-buckets: as many as bring a fine bucket to `target` entries if the round emitted every row's maximum, at most
            // 4096 (what k_part_split counts in LDS) and at most what the key has bits for
            uint32_t l2_max = 0;
            while (l2_max < 12u && l2_max < keybits - l1 && ((cap_ent / nsub + 1ull) >> (l1 + l2_max)) > 64ull) ++l2_max;
            // the pair walk: a wave per fine bucket of at most wcap entries (1: the block kernel for every bucket, a cross-check)
            static const int pairs_form = [] { const char* e = getenv("BADGER_AMD_DJ_WCAP"); const int x = e ? atoi(e) : 256; return x == 1 ? 1 : 256; }();
            uint32_t target = pairs_form == 1 ? DJ_CAP / 2u : (uint32_t)pairs_form * 5u / 8u;      // entries a fine bucket should hold at most on average (80 .. 160 of a wave's 256)
            if (const char* e = getenv("BADGER_AMD_DJ_TARGET")) target = (uint32_t)std::max(1, atoi(e));          // (for tests: oversize buckets)
            if (const char* e = getenv("BADGER_AMD_DJ_L2MAX")) l2_max = (uint32_t)std::min((int)l2_max, std::max(0, atoi(e)));
            uint32_t tiles_want = (uint32_t)ctx->g_cus * 8u;
            uint32_t rows_per_tile = ((n + tiles_want - 1u) / tiles_want + 63u) & ~63u;
            if (rows_per_tile < 64u) rows_per_tile = 64u;
            const uint32_t ntiles = (n + rows_per_tile - 1u) / rows_per_tile;
            // workspace: hist [ntiles][nb1] | tot [nb1] | geom | base u64 [nb1 + 1] | fstart [(nb1 << l2_max) + 1]; entries twice
            const size_t w_hist = (size_t)ntiles * nb1, w_fstart = ((size_t)nb1 << l2_max) + 1;
            const size_t small_bytes = 4 * (w_hist + nb1 + bdgpart::G_WORDS + 2 * w_fstart) + 8 * ((size_t)nb1 + 1) + 64;
            if ((rc = bdg_reserve(ctx, ctx->g_sig, small_bytes))) return rc;            // (the sweep's signature buffer is free here)
            auto* base = static_cast<unsigned long long*>(ctx->g_sig.p);
            auto* hist = reinterpret_cast<uint32_t*>(base + nb1 + 1);
            auto* tot = hist + w_hist;
            auto* geom = tot + nb1;
            auto* fstart = geom + bdgpart::G_WORDS;
            auto* ovf = fstart + w_fstart;                                          // [0] unused, then the buckets left to the block kernel
            if ((rc = bdg_reserve(ctx, ctx->g_qj, 4ull * 2ull * (cap_ent + 64) + (one_deletion ? 0ull : 16ull * n)))) return rc;
            auto* e_a = static_cast<uint32_t*>(ctx->g_qj.p);
            auto* e_b = e_a + cap_ent + 64;
            auto* kept = reinterpret_cast<ulonglong2*>(e_b + cap_ent + 64);                 // (thr 2: the pairs each row keeps, from the first run to the second)
            {
                ScopedKernelTimer tm(ctx, one_deletion ? "k_d1_count" : "k_d2_count");
#define BDG_ROWS(EMIT) do { \
                    if (one_deletion) { if (l1 <= 10u) hipLaunchKernelGGL((k_d1_rows<EMIT, 1024>), dim3(ntiles), dim3(256), 0, st, d_ranks, n, rows_per_tile, sub, nsub, l1, hist, tot, base, geom, e_a); \
                                        else hipLaunchKernelGGL((k_d1_rows<EMIT, 4096>), dim3(ntiles), dim3(256), 0, st, d_ranks, n, rows_per_tile, sub, nsub, l1, hist, tot, base, geom, e_a); } \
                    else { if (l1 <= 10u) hipLaunchKernelGGL((k_d2_rows<EMIT, 1024>), dim3(ntiles), dim3(256), 0, st, d_ranks, n, rows_per_tile, sub, nsub, l1, hist, tot, base, geom, e_a, kept); \
                           else hipLaunchKernelGGL((k_d2_rows<EMIT, 4096>), dim3(ntiles), dim3(256), 0, st, d_ranks, n, rows_per_tile, sub, nsub, l1, hist, tot, base, geom, e_a, kept); } } while (0)
                BDG_ROWS(false);
            }
            {
                ScopedKernelTimer tm(ctx, one_deletion ? "k_d1_scan" : "k_d2_scan");
                hipLaunchKernelGGL(bdgpart::k_part_colscan, dim3(nb1), dim3(256), 0, st, hist, ntiles, nb1, tot);
                hipLaunchKernelGGL(bdgpart::k_part_bases, dim3(1), dim3(1024), 0, st, tot, nb1, target, l2_max, cap_ent, base, geom);
            }
            {
                ScopedKernelTimer tm(ctx, one_deletion ? "k_d1_emit" : "k_d2_emit");
                BDG_ROWS(true);
#undef BDG_ROWS
            }
            {
                ScopedKernelTimer tm(ctx, one_deletion ? "k_d1_split" : "k_d2_split");
                hipLaunchKernelGGL(bdgpart::k_part_split<uint32_t>, dim3(nb1), dim3(1024), 0, st, e_a, e_b, base, geom, nb1, keybits - l1, fstart);
            }
            {
                ScopedKernelTimer tm(ctx, one_deletion ? "k_d1_pairs" : "k_d2_pairs");
#define BDG_PAIRS_W(NDEL, WCAP) hipLaunchKernelGGL((k_d2_pairs_w<NDEL, WCAP, DJ_ECAPW>), dim3(wgrid), dim3(256), 0, st, e_b, fstart, geom, l1, \
                                                   d_ranks, row_begin, row_end, thr, qgram_T, d_out, cap, cnt, ovf)
#define BDG_PAIRS_B(NDEL, GRID, LIST) hipLaunchKernelGGL((k_d2_pairs<NDEL, DJ_THREADS, DJ_CAP, DJ_ECAPW>), dim3(GRID), dim3(DJ_THREADS), 0, st, e_b, fstart, geom, l1, \
                                                         d_ranks, row_begin, row_end, thr, qgram_T, d_out, cap, cnt, LIST)
                const uint32_t bgrid = (uint32_t)ctx->g_cus * 3u;                   // (53 KB of LDS a block of 8 waves: three fit a compute unit)
                if (pairs_form == 1) {
                    if (one_deletion) BDG_PAIRS_B(1, bgrid, (const uint32_t*)nullptr); else BDG_PAIRS_B(2, bgrid, (const uint32_t*)nullptr);
                } else {
                    const uint32_t wgrid = (uint32_t)ctx->g_cus * d2_pairs_blocks_per_cu();
                    if (one_deletion) BDG_PAIRS_W(1, 256); else BDG_PAIRS_W(2, 256);
                    // what the waves left: buckets beyond their capacity (none on any data met so far; the launch is a few microseconds)
                    if (one_deletion) BDG_PAIRS_B(1, (uint32_t)ctx->g_cus, ovf); else BDG_PAIRS_B(2, (uint32_t)ctx->g_cus, ovf);
                }
#undef BDG_PAIRS_W
#undef BDG_PAIRS_B
            }
            BDG_HIP_TRY(ctx, hipGetLastError());
            ctx->g_dj_geom = geom;                                                // (bdg_graph_status: what the device reported)
            if (may_overflow || rounds > 1u) {                                    // (the next round rewrites the report)
                uint32_t flags = 0;
                if ((rc = bdg_graph_join_flags(ctx, &flags))) return rc;
                if (flags) return bdg_graph_flags_error(ctx, flags);
            }
        }
        return BDG_OK;
    }
    if (qjoin) {
        const size_t m = (size_t)n * QJ_NQ;
        const uint32_t W = QJ_W;
        const uint32_t G = (n + W - 1) / W;
        const bool closed_form = ctx->graph_algo == 4;
        // tiles of rows for the two runs of the counting sort: at most ~512 of them (the place run gives a tile to one wave)
        uint32_t rows_per_tile = ((n + 511u) / 512u + 63u) & ~63u;
        if (rows_per_tile < 64u) rows_per_tile = 64u;
        const uint32_t ntiles = (n + rows_per_tile - 1u) / rows_per_tile;
        // workspace: base u64 [4097] | v_out [m] | pos_of [m] | bucket_off [4097] | split | hist [4096][ntiles] | tot [4096] | geom
        const size_t words = 2 * m + 4097 + 4096ull * (G + 1) + 4096ull * ntiles + 4096 + bdgpart::G_WORDS + 64;
        if ((rc = bdg_reserve(ctx, ctx->g_qj, 8ull * 4098 + sizeof(uint32_t) * words + 256))) return rc;
        auto* base = static_cast<unsigned long long*>(ctx->g_qj.p);
        auto* v_out = reinterpret_cast<uint32_t*>(base + 4098);
        auto* pos_of = v_out + m;
        auto* bucket_off = pos_of + m;
        auto* split = bucket_off + 4097;
        auto* hist = split + 4096ull * (G + 1);
        auto* tot = hist + 4096ull * ntiles;
        auto* geom = tot + 4096;
        {
            ScopedKernelTimer tm(ctx, "k_qj_build");
            hipLaunchKernelGGL(k_qj_count, dim3(ntiles), dim3(256), 0, st, d_ranks, n, rows_per_tile, hist);
            hipLaunchKernelGGL(bdgpart::k_part_colscan, dim3(4096), dim3(256), 0, st, hist, ntiles, 4096u, tot);
            hipLaunchKernelGGL(bdgpart::k_part_bases, dim3(1), dim3(1024), 0, st, tot, 4096u, 1u, 0u, (unsigned long long)m, base, geom);
            hipLaunchKernelGGL(k_qj_place, dim3(ntiles), dim3(64), 0, st, d_ranks, n, rows_per_tile, hist, base, v_out, pos_of, bucket_off);
            if (closed_form) hipLaunchKernelGGL(k_qj_split, dim3((4096u * (G + 1) + 255) / 256), dim3(256), 0, st, v_out, bucket_off, G, W, split);
        }
        if ((rc = graph_props(ctx))) return rc;
        if (closed_form) {
            ScopedKernelTimer tm(ctx, "k_graph_qjoin");
            uint32_t grid = (uint32_t)ctx->g_qj_per_cu * (uint32_t)ctx->g_cus;          // resident grid, rows interleaved
            if (grid > row_end - row_begin) grid = row_end - row_begin;
            hipLaunchKernelGGL(k_graph_qjoin<QJ_W>, dim3(grid), dim3(256), 0, st, d_ranks, n, row_begin, row_end, v_out, pos_of, split, G,
                               thr, qgram_T, 1, d_out, cap, reinterpret_cast<unsigned long long*>(d_n_edges));
        } else {
            ScopedKernelTimer tm(ctx, "k_graph_qjoin_w");
            const int var = ctx->g_qjw_variant;
            const uint32_t waves = qgram_T < 2 ? 4u : (var == 1 ? 2u : (var == 2 ? 1u : 4u));
            uint32_t grid = (uint32_t)ctx->g_qjw_per_cu * (uint32_t)ctx->g_cus;         // resident grid, one row per wave at a time, rows interleaved
            const uint32_t want = (row_end - row_begin + waves - 1u) / waves;
            if (grid > want) grid = want;
            auto* ne = reinterpret_cast<unsigned long long*>(d_n_edges);
#define BDG_QJW(ROWS, WAVES, GE2) hipLaunchKernelGGL((k_graph_qjoin_w<ROWS, WAVES, GE2>), dim3(grid), dim3(64 * WAVES), 0, st, d_ranks, n, \
                                                     row_begin, row_end, v_out, pos_of, bucket_off, thr, qgram_T, d_out, cap, ne)
            if (qgram_T < 2) BDG_QJW(8192, 4, false);
            else if (var == 1) BDG_QJW(16384, 2, true);
            else if (var == 2) BDG_QJW(32768, 1, true);
            else if (var == 3) BDG_QJW(4096, 4, true);
            else BDG_QJW(8192, 4, true);
#undef BDG_QJW
        }
        BDG_HIP_TRY(ctx, hipGetLastError());
        return BDG_OK;
    }
    if (probe) {
        int bbits = 16;
        while (bbits < 27 && (1u << (bbits - 4)) < n) ++bbits;           // ~16 bits per barcode
        const size_t bm_bytes = (size_t(1) << bbits) / 8;
        if ((rc = bdg_reserve(ctx, ctx->g_sig, bm_bytes))) return rc;       // (the scan path's signature buffer is free here)
        auto* bitmap = static_cast<uint32_t*>(ctx->g_sig.p);
        BDG_HIP_TRY(ctx, hipMemsetAsync(bitmap, 0, bm_bytes, st));
        if ((rc = bdg_reserve(ctx, ctx->g_qj, sizeof(uint32_t) * ((size_t(1) << bbits) + 2)))) return rc;     // (the q-gram join's workspace is free here)
        auto* dir = static_cast<uint32_t*>(ctx->g_qj.p);
        {
            ScopedKernelTimer tm(ctx, "k_graph_bitmap");
            hipLaunchKernelGGL(k_graph_bitmap, dim3((n + 255) / 256), dim3(256), 0, st, d_ranks, n, 32 - bbits, bitmap, dir);
        }
        ScopedKernelTimer tm(ctx, "k_graph_probe");
        const uint64_t threads = 4ull * (row_end - row_begin);
        hipLaunchKernelGGL(k_graph_probe, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, st, d_ranks, n, row_begin, row_end, qgram_T,
                           bitmap, dir, 32 - bbits, d_out, cap, reinterpret_cast<unsigned long long*>(d_n_edges));
        BDG_HIP_TRY(ctx, hipGetLastError());
        return BDG_OK;
    }
    if ((rc = bdg_reserve(ctx, ctx->g_sig, sizeof(uint32_t) * (size_t)n))) return rc;
    auto* sig = static_cast<uint32_t*>(ctx->g_sig.p);
    {
        ScopedKernelTimer tm(ctx, "k_graph_sig");
        hipLaunchKernelGGL(k_graph_sig, dim3((n + 255) / 256), dim3(256), 0, st, d_ranks, n, sig);
    }
    {
        ScopedKernelTimer tm(ctx, "k_graph_scan");
        const uint32_t ntiles = (row_end + 255u) / 256u - row_begin / 256u;
        hipLaunchKernelGGL(k_graph_scan, dim3((ntiles + 1) / 2), dim3(256), 0, st, d_ranks, sig, n, row_begin, row_end, thr, qgram_T,
                           d_out, cap, reinterpret_cast<unsigned long long*>(d_n_edges));
    }
    BDG_HIP_TRY(ctx, hipGetLastError());
    return BDG_OK;
}
