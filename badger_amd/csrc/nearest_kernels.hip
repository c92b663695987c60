// K2: nearest whitelist barcode under Levenshtein distance (operator form of the loop
// body of BarcodeGraph.postprocessing, reference barcode_graph.py:376-384, whose
// distance is editdistance.eval on 16-character strings).
//
// Two device paths with identical results (ties -> lowest caller index, tie count):
//   scan   k_nearest_scan: exhaustive.  One query per lane (its four 16-bit match
//          vectors live in registers), whitelist tiles staged in LDS and broadcast to
//          the wave, Myers/Hyyro bit-vector distance per pair.  Any max_ed.
//   probe  k_nearest_pairs + k_nearest_delins (+ k_nearest_scan for the rare query whose hit list overflows), max_ed <= 2:
//          instead of the whitelist the places are visited where a neighbour can sit.  An entry within Hamming distance 2
//          shares two whole 4-base blocks with the query: six tables of buckets keyed by a block pair (pass 1, which
//          settles distance 0 and 1 and the two-substitution neighbours).  Equal-length strings at distance 2 that are
//          not two substitutions apart are one deletion plus one insertion apart, i.e. share a 15-base deletion variant:
//          a 2^30-bit map of the whitelist's variants, then the owners of a variant that is present (pass 2).  Details
//          at PairTables / delmap_index below.
#include "bdg_common.hpp"
#include "bdg_partition.hpp"

#include <algorithm>
#include <numeric>

namespace {

constexpr uint32_t NONE_IDX = 0xFFFFFFFFu;

struct WlIndex {
    const uint32_t* sorted;
    const uint32_t* orig;
    const uint32_t* prefix;
    const uint32_t* bitmap;
    uint32_t n;
    int pshift;     // 32 - pbits
    int bshift;     // 32 - bbits
};

__device__ __forceinline__ bool wl_lookup(const WlIndex& ix, uint32_t key, uint32_t& orig)
{
    const uint32_t b = key >> ix.bshift;
    if (!((ix.bitmap[b >> 5] >> (b & 31u)) & 1u)) return false;
    const uint32_t p = key >> ix.pshift;
    const uint32_t lo = ix.prefix[p], hi = ix.prefix[p + 1];
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t v = ix.sorted[k];
        if (v == key) { orig = ix.orig[k]; return true; }
        if (v > key) break;
    }
    return false;
}

// ---- exhaustive scan -------------------------------------------------------
constexpr int SCAN_TILE = 4096;

__global__ __launch_bounds__(256)
void k_nearest_scan(const uint32_t* __restrict__ q, uint32_t qstride, int recs,
                    const uint32_t* __restrict__ qlist, uint32_t nq_host,
                    const uint32_t* __restrict__ d_nq,
                    const uint32_t* __restrict__ wl_sorted, const uint32_t* __restrict__ wl_orig, uint32_t nw,
                    uint32_t max_ed, uint32_t* __restrict__ best_idx, uint8_t* __restrict__ best_ed,
                    uint16_t* __restrict__ n_ties)
{
    __shared__ uint32_t s_rank[SCAN_TILE];
    __shared__ uint32_t s_orig[SCAN_TILE];
    const uint32_t nq = d_nq ? *d_nq : nq_host;          // list length may live on the device (overflow list)
  for (uint32_t slot0 = blockIdx.x * 256u; slot0 < nq; slot0 += gridDim.x * 256u) {
    const uint32_t slot = slot0 + threadIdx.x;
    const bool active = slot < nq;
    // queries: a plain array (qstride 1) or the bc_rank field of extraction records (qstride 8 words; a record whose
    // barcode is not 16 ACGT bases has no query and reports "nothing within max_ed")
    const uint32_t qi = active ? (qlist ? qlist[slot] : slot) : 0u;
    const uint32_t qq = active ? q[(size_t)qi * qstride] : 0u;
    const bool usable = !recs || !active || ((q[(size_t)qi * qstride + 1] >> 24) & BDG_FLAG_RANK_OK) != 0;
    // The Myers vectors stay SPREAD - row i of the query at bit 2i, where its 2-bit codes are - and a column's equality vector
    // is two three-input operations on the query's two bit planes and the entry's code bits (scalars: the entry is the same in
    // every lane); the addition carries through odd bits that pv keeps set (graph_kernels.hip, dmin3: the same statements).
    constexpr uint32_t EVEN = 0x55555555u;
    const uint32_t P0 = qq & EVEN, P1 = (qq >> 1) & EVEN;
    uint32_t best = 255u, bidx = NONE_IDX, ties = 0u;
    for (uint32_t t0 = 0; t0 < nw; t0 += SCAN_TILE) {
        const uint32_t tn = nw - t0 < (uint32_t)SCAN_TILE ? nw - t0 : (uint32_t)SCAN_TILE;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < tn; k += 256u) { s_rank[k] = wl_sorted[t0 + k]; s_orig[k] = wl_orig[t0 + k]; }
        __syncthreads();
        for (uint32_t k = 0; k < tn; ++k) {
            const uint32_t t = __builtin_amdgcn_readfirstlane(s_rank[k]);
            const uint32_t o = __builtin_amdgcn_readfirstlane(s_orig[k]);
            uint32_t pv = 0xFFFFFFFFu, mv = 0u, score = 16u;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t m0 = (uint32_t)((int32_t)(t << (31 - 2 * j)) >> 31), m1 = (uint32_t)((int32_t)(t << (30 - 2 * j)) >> 31);
                const uint32_t t1 = __builtin_amdgcn_bitop3_b32(P0, m0, EVEN, 0x82);             // ~(P0 ^ m0) & EVEN
                const uint32_t eq = __builtin_amdgcn_bitop3_b32(t1, P1, m1, 0x90);               // t1 & ~(P1 ^ m1)
                const uint32_t xv = eq | mv;
                const uint32_t xh = __builtin_amdgcn_bitop3_b32((eq & pv) + pv, pv, eq, 0xBE);    // (((eq & pv) + pv) ^ pv) | eq
                uint32_t ph = __builtin_amdgcn_bitop3_b32(mv, xh, pv, 0xF1);                     // mv | ~(xh | pv)
                uint32_t mh = pv & xh;
                score += (ph >> 30) & 1u;
                score -= (mh >> 30) & 1u;
                ph = (ph << 2) | 1u;
                mh = mh << 2;
                pv = __builtin_amdgcn_bitop3_b32(mh, xv, ph, 0xF1);                              // mh | ~(xv | ph)
                mv = ph & xv;
            }
            const bool better = score < best, same = score == best;
            bidx = better ? o : ((same && o < bidx) ? o : bidx);
            ties = better ? 1u : (same ? ties + 1u : ties);
            best = better ? score : best;
        }
    }
    if (active) {
        if (best > max_ed || !usable) { best = 255u; bidx = NONE_IDX; ties = 0u; }
        best_idx[qi] = bidx; best_ed[qi] = (uint8_t)best; n_ties[qi] = (uint16_t)(ties > 0xFFFFu ? 0xFFFFu : ties);
    }
  }
}

// ---- probe path ----------------------------------------------------------------
// Pass 1 (k_nearest_pairs), one query per lane.  A 16-mer is four blocks of four bases.  A
// whitelist entry within Hamming distance 2 of the query agrees with it on at least two whole
// blocks, so it sits in one of six buckets keyed by a pair of blocks (6 tables, 65,536 buckets
// each, ~nw/65536 entries per bucket).  Scanning those six buckets settles distance 0 and 1
// completely (equal-length strings at Levenshtein distance 1 differ by one substitution) and
// finds every distance-2 entry that is two substitutions away.  An entry is counted in the
// first pair of agreeing blocks only, so ties are exact.
// Pass 2 (k_nearest_delins), 4 lanes per query that still has no hit below distance 2: the
// remaining distance-2 entries are one deletion + one insertion away, i.e. share a 15-mer
// deletion variant with the query.  Lane g probes del(q, 4g .. 4g+3) in a 2^30-bit map of all
// deletion variants of the whitelist (one 64-bit word, see delmap_index); only on a hit are the
// owners of the variant looked up.
struct PairTables {
    // Blocks of 16 words (64 bytes, one memory sector): word 0 = entries in the block (<= 30) | next block of the bucket << 8
    // (0 = none), words 1..15 = 30 entries of 16 bits: the two 8-bit blocks of the rank that are NOT the bucket's key (the key
    // spells the other two), lower block in the low byte.  Block p * 65536 + key is the head of bucket `key` of table p; longer
    // buckets continue in blocks appended behind the 6 * 65536 heads.  `idx` holds the caller index of each entry, 32 words
    // per block (read only on a hit).  One sector per bucket probe; a bucket of the 737,280-entry list (11 entries on
    // average) never chains, one of a 4.9 M list (75) takes 3 blocks.
    const uint32_t* rank;
    const uint32_t* idx;
    const uint32_t* delmap;  // 2^30 bits
    uint32_t nw;
};
constexpr uint32_t PAIR_BLOCK_ENTRIES = 30;

__device__ __forceinline__ uint32_t pair_key(uint32_t r, int p)
{
    // pairs (0,1) (0,2) (0,3) (1,2) (1,3) (2,3); block k = bits [8k, 8k+8)
    const int bi = p < 3 ? 0 : (p < 5 ? 1 : 2);
    const int bj = p < 3 ? p + 1 : (p < 5 ? p - 1 : 3);
    return ((r >> (8 * bi)) & 0xFFu) | (((r >> (8 * bj)) & 0xFFu) << 8);
}

// Scan order of the six pair tables: (0,1) = table 0 and (2,3) = table 5 first.  An entry within Hamming distance 1 agrees with
// the query on three blocks, and every three of the four blocks contain (0,1) or (2,3): after those two tables every
// distance-0/1 entry has been seen, and a query that found one needs nothing else.
// canonical_pair: first table, in scan order, whose two blocks agree (x = query ^ entry); an entry is counted there only.
__device__ __forceinline__ int canonical_pair(uint32_t x)
{
    const bool c0 = (x & 0xFFu) == 0, c1 = (x & 0xFF00u) == 0, c2 = (x & 0xFF0000u) == 0, c3 = (x & 0xFF000000u) == 0;
    return (c0 && c1) ? 0 : (c2 && c3) ? 5 : (c0 && c2) ? 1 : (c0 && c3) ? 2 : (c1 && c2) ? 3 : 4;
}

__device__ __forceinline__ uint32_t hamming16(uint32_t x)
{
    return __popc((x | (x >> 1)) & 0x55555555u);
}

// list of the queries pass 2 must look at: LSH segments of nq slots, one counter (own 128-byte line) per segment; a block
// reserves its slots with ONE atomic (same-address atomics complete ~11 ns apart, whoever issues them)
constexpr int LSH = 8;
constexpr int CTR_N3 = LSH * 32;           // uint32 index of the overflow-list counter
constexpr size_t NCTR_BYTES = (LSH + 1) * 128;

__global__ __launch_bounds__(256)
void k_nearest_pairs(const uint32_t* __restrict__ q, uint32_t qstride, int recs, uint32_t nq, PairTables pt, uint32_t max_ed,
                     uint32_t* __restrict__ best_idx, uint8_t* __restrict__ best_ed, uint16_t* __restrict__ n_ties,
                     uint2* __restrict__ list2, uint32_t* __restrict__ counters)
{
    __shared__ uint32_t s_wcnt[4], s_base;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool on = i < nq;
    bool need2 = false;
    uint32_t qq = 0;
    bool usable = on;
    if (on) {
        qq = q[(size_t)i * qstride];
        if (recs) usable = ((q[(size_t)i * qstride + 1] >> 24) & BDG_FLAG_RANK_OK) != 0;      // flags byte of the record
    }
    if (on && !usable) { best_idx[i] = NONE_IDX; best_ed[i] = 255; n_ties[i] = 0; }
    if (usable) {
        uint32_t best = 3u, bidx = NONE_IDX, ties = 0u;
        auto scan_bucket = [&](int p) {
            // the two blocks outside the key: (k, l) = the complement of pair p's (i, j)
            const int bk = p < 3 ? (p == 0 ? 2 : 1) : (p < 5 ? 0 : 0);
            const int bl = p < 3 ? (p == 2 ? 2 : 3) : (p == 3 ? 3 : (p == 4 ? 2 : 1));
            const uint32_t qrest = ((qq >> (8 * bk)) & 0xFFu) | (((qq >> (8 * bl)) & 0xFFu) << 8);
            uint32_t blk = (uint32_t)p * 65536u + pair_key(qq, p);
            do {
                const uint4* rb = reinterpret_cast<const uint4*>(pt.rank + (size_t)blk * 16u);
                const uint4 q0 = rb[0], q1 = rb[1], q2 = rb[2], q3 = rb[3];          // the whole sector, four independent loads
                const uint32_t wr[16] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
                const uint32_t cnt = wr[0] & 0xFFu;
#pragma unroll
                for (uint32_t u = 0; u < PAIR_BLOCK_ENTRIES; ++u) {
                    const uint32_t e = (wr[1 + (u >> 1)] >> (16 * (u & 1u))) & 0xFFFFu;
                    const uint32_t xr = e ^ qrest;                                     // differences in the two blocks outside the key
                    const uint32_t h = __popc((xr | (xr >> 1)) & 0x5555u);
                    if (u < cnt && h <= 2u && h <= best) {
                        const uint32_t x = ((xr & 0xFFu) << (8 * bk)) | ((xr >> 8) << (8 * bl));   // query ^ entry (zero in the key blocks)
                        if (canonical_pair(x) == p) {
                            const uint32_t wo = pt.idx[(size_t)blk * 32u + u];
                            if (h < best) { best = h; bidx = wo; ties = 1u; }
                            else { ties++; bidx = wo < bidx ? wo : bidx; }
                        }
                    }
                }
                blk = wr[0] >> 8;
            } while (blk);
        };
        scan_bucket(0);
        if (best != 0u) {                                  // an exact match sits in table (0,1) and nothing can tie with it
            scan_bucket(5);
            if (best > 1u) {                               // otherwise all entries within distance 1 have been seen
#pragma unroll
                for (int p = 1; p <= 4; ++p) scan_bucket(p);
            }
        }
        if (best > max_ed) { best = 255u; bidx = NONE_IDX; ties = 0u; }
        best_idx[i] = bidx; best_ed[i] = (uint8_t)(best == 3u ? 255u : best);
        n_ties[i] = (uint16_t)(ties > 0xFFFFu ? 0xFFFFu : ties);
        need2 = max_ed >= 2u && (best == 2u || best == 3u || best == 255u);
    }
    // block-wide reservation in this block's list segment
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(need2);
    if (lane == 0) s_wcnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    const uint32_t seg = blockIdx.x % LSH;
    if (threadIdx.x == 0) {
        const uint32_t tot = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        s_base = tot ? atomicAdd(&counters[seg * 32], tot) : 0u;
    }
    __syncthreads();
    if (need2) {
        uint32_t at = s_base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; ++w) at += s_wcnt[w];
        list2[(size_t)seg * nq + at] = make_uint2(i, qq);                 // pass 2 gets the query with its index: one load less on its chain
    }
}

__device__ __forceinline__ uint32_t low_mask(int bases) { return bases >= 16 ? 0xFFFFFFFFu : ((1u << (2 * bases)) - 1u); }

// Deletion variants of the whitelist, twice: as a 2^30-bit membership map (the cheap first question) and as (variant, entry)
// pairs to be sorted by variant (the second: WHICH entries own a variant).  Deleting a base inside a run of equal bases
// gives the same variant as deleting its left neighbour: only the first of a run is emitted (key 0xFFFFFFFF sorts the
// others to the end).
// directory over the top 25 of the 30 variant bits (128 MB): 0.35 entries per bucket.  A wave waits for the longest bucket among
// its lanes' hits, one round trip to memory per entry: with 22 bits (2.8 entries per bucket) k_nearest_delins took 0.120 ms per
// 1M calls, with 24 bits 0.089, 25 bits 0.084, 27 bits 0.080
constexpr int DV_DIR_SHIFT = 5;
constexpr uint32_t DV_DIR_N = 1u << (30 - DV_DIR_SHIFT);

// The membership map exists four times, each copy addressed by a different permutation of the variant's bits.  The deletion
// variants i = 4g .. 4g+3 of one 16-mer differ from each other only in the bases 4g .. 4g+2 (deleting base i or base i' > i
// changes the bases in between), i.e. in the six bits [8g, 8g+6): copy g has those six bits as the bit number inside a 64-bit
// word, so the four probes of a group of lanes read ONE 8-byte word (one memory sector per group instead of one per lane).
__device__ __forceinline__ uint32_t delmap_index(uint32_t d, int g)
{
    const uint32_t lowm = (1u << (8 * g)) - 1u;
    return ((d >> (8 * g)) & 63u) | ((d & lowm) << 6) | (d & ~((lowm << 6) | 63u));
}
constexpr uint32_t DELMAP_WORDS = 1u << 25;              // 2^30 bits per copy

// PLACE = false: the map bits and how many variants fall into each directory bucket (dir[bucket + 2]: an inclusive scan
// then leaves every bucket's start at dir[bucket + 1] and its end at dir[bucket + 2]).  PLACE = true: every variant's entry
// {variant, rank, caller index, 0} at the next free place of its bucket (an atomic add on dir[bucket + 1], which thereby
// moves on to the bucket's end = the next bucket's start: afterwards dir[b] is where bucket b starts and dir[b + 1] where
// it ends, the form the look-ups read).  A counting sort on the directory key: inside a bucket - 0.35 entries on average -
// the look-up compares every entry anyway, so no order is needed there (round 3 radix-sorted the pairs with hipCUB).
template <bool PLACE>
__global__ __launch_bounds__(256)
void k_build_delmap(const uint32_t* __restrict__ wl, const uint32_t* __restrict__ orig, uint32_t nw, uint32_t* __restrict__ delmap,
                    uint32_t* __restrict__ dir, uint4* __restrict__ ent)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t w = g >> 4; const int i = (int)(g & 15u);
    if (w >= nw) return;
    const uint32_t r = wl[w];
    const uint32_t lm = low_mask(i);
    const uint32_t d = ((r & lm) | ((r >> 2) & ~lm)) & 0x3FFFFFFFu;
    const bool dup = i > 0 && (((r >> (2 * i)) ^ (r >> (2 * i - 2))) & 3u) == 0u;
    if (dup) return;
    if (PLACE) {
        ent[atomicAdd(&dir[(d >> DV_DIR_SHIFT) + 1], 1u)] = make_uint4(d, r, orig[w], 0u);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) { const uint32_t x = delmap_index(d, c); atomicOr(&delmap[c * DELMAP_WORDS + (x >> 5)], 1u << (x & 31u)); }
        atomicAdd(&dir[(d >> DV_DIR_SHIFT) + 2], 1u);
    }
}

__global__ __launch_bounds__(256)
void k_nearest_delins(const uint2* __restrict__ list2,
                      uint32_t nq, const uint32_t* counters, const uint32_t* __restrict__ delmap,
                      const uint4* __restrict__ dv_ent, const uint32_t* __restrict__ dv_dir,
                      uint32_t* __restrict__ best_idx, uint8_t* __restrict__ best_ed, uint16_t* __restrict__ n_ties,
                      uint32_t* __restrict__ list3, uint32_t* counters_out)
{
    uint32_t n2 = 0;                                                              // virtual length: LSH * longest segment
#pragma unroll
    for (int k = 0; k < LSH; ++k) { const uint32_t c = counters[k * 32]; n2 = c > n2 ? c : n2; }
    n2 *= LSH;
    // Four lanes per query: lane g of a query owns its deletion variants 4g .. 4g+3, whose map bits lie in one 64-bit word.
    const int lane = threadIdx.x & 63, sub = lane & 3, grp = lane >> 2;
    const uint32_t wave_slot0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 16u;    // first query slot of this wave
    const uint32_t ngroups = gridDim.x * 64u;                                     // a multiple of LSH: a group stays in its segment
    const unsigned long long gmask = 0xFull << (4 * grp);
    const uint32_t seg = (wave_slot0 + (uint32_t)grp) % LSH;
    const uint32_t seg_cnt = counters[seg * 32];
    const uint2* seg_list = list2 + (size_t)seg * nq;
    // Two loads lead to a group's answer before any look-up: its list entry {index, query} and the deletion-map word of
    // its variants.  They are issued two and one iterations ahead, so an iteration starts with both in registers.
    auto fetch = [&](uint32_t s0) -> uint2 {                                      // entry s / LSH of segment s % LSH
        const uint32_t s = s0 + (uint32_t)grp;
        return (s < n2 && s / LSH < seg_cnt) ? seg_list[s / LSH] : make_uint2(NONE_IDX, 0u);
    };
    auto variant = [&](uint32_t qq, int t) -> uint32_t {                          // deletion variant 4 * sub + t
        const uint32_t lm = low_mask(4 * sub + t);
        return ((qq & lm) | ((qq >> 2) & ~lm)) & 0x3FFFFFFFu;
    };
    const uint2* const my_map = reinterpret_cast<const uint2*>(delmap + (size_t)sub * DELMAP_WORDS);
    auto map_word = [&](uint32_t qq) -> uint2 { return my_map[delmap_index(variant(qq, 0), sub) >> 6]; };
    uint2 e1 = fetch(wave_slot0), e2 = fetch(wave_slot0 + ngroups);
    uint2 w1 = e1.x != NONE_IDX ? map_word(e1.y) : make_uint2(0u, 0u);
    for (uint32_t s0 = wave_slot0; s0 < n2; s0 += ngroups) {                      // wave-uniform loop bound
        const uint32_t qi = e1.x, qq = e1.y;
        const unsigned long long word = ((unsigned long long)w1.y << 32) | w1.x;
        const bool on = qi != NONE_IDX;
        e1 = e2;
        w1 = e1.x != NONE_IDX ? map_word(e1.y) : make_uint2(0u, 0u);
        e2 = fetch(s0 + 2u * ngroups);
        // A variant that occurs in the whitelist: WHICH entries own it (they are the re-insertions of one base into the
        // variant)?  directory -> the few sorted {variant, rank, caller index} entries of its bucket.  Entries within Hamming
        // distance 2 were pass 1's.  Equal neighbours give equal variants: the first of a run stands for all.
        uint32_t found[4] = { 0, 0, 0, 0 }; int nf = 0; bool overflow = false;
        auto take = [&](const uint4 en, uint32_t d) __attribute__((always_inline)) {
            if (en.x != d) return;
            if (hamming16(en.y ^ qq) <= 2u) return;
            const uint32_t oo = en.z;
            const bool dup = (nf > 0 && found[0] == oo) || (nf > 1 && found[1] == oo) ||
                             (nf > 2 && found[2] == oo) || (nf > 3 && found[3] == oo);
            if (!dup) {
                if (nf < 4) { found[0] = nf == 0 ? oo : found[0]; found[1] = nf == 1 ? oo : found[1];
                              found[2] = nf == 2 ? oo : found[2]; found[3] = nf == 3 ? oo : found[3]; ++nf; }
                else overflow = true;
            }
        };
        uint32_t dvar[4]; bool hit[4]; bool any_hit = false;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = 4 * sub + t;
            dvar[t] = variant(qq, t);
            const bool dup_del = i > 0 && (((qq >> (2 * i)) ^ (qq >> (2 * i - 2))) & 3u) == 0u;
            hit[t] = on && !dup_del && ((word >> (delmap_index(dvar[t], sub) & 63u)) & 1ull);
            any_hit = any_hit || hit[t];
        }
        if (__ballot(any_hit)) {
            // The look-ups behind the hits of a lane's four variants run side by side, not one after the other: first every
            // directory range, then the first DV_AHEAD entries of every range (a directory bucket holds 0.35 entries on
            // average), all of them unconditional loads (index 0 stands in where there is nothing to load); what a longer
            // bucket holds beyond that is walked afterwards.
            constexpr int DV_AHEAD = 3;
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t b = hit[t] ? (dvar[t] >> DV_DIR_SHIFT) : 0u;
                lo[t] = dv_dir[b]; hi[t] = dv_dir[b + 1];
                if (!hit[t]) hi[t] = lo[t] = 0u;
            }
            uint4 en[4][DV_AHEAD];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < DV_AHEAD; ++e) en[t][e] = dv_ent[lo[t] + (uint32_t)e < hi[t] ? lo[t] + (uint32_t)e : 0u];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int e = 0; e < DV_AHEAD; ++e) if (lo[t] + (uint32_t)e < hi[t]) take(en[t][e], dvar[t]);
                for (uint32_t k = lo[t] + DV_AHEAD; k < hi[t]; ++k) take(dv_ent[k], dvar[t]);
            }
        }
        // merge inside the query's four lanes: distinct hits, lowest caller index
        const bool any_over = (__ballot(overflow) & gmask) != 0;
        uint32_t add = 0, midx = NONE_IDX;
        int pending = nf;
        while (true) {
            const unsigned long long bal_all = __ballot(pending > 0);
            if (!bal_all) break;
            const unsigned long long bal = bal_all & gmask;
            const int src = bal ? __builtin_ctzll(bal) : lane;
            const uint32_t v = __shfl(found[0], src);
            if (bal) {
                ++add; midx = v < midx ? v : midx;
                const bool h0 = pending > 0 && found[0] == v, h1 = pending > 1 && found[1] == v;
                const bool h2 = pending > 2 && found[2] == v, h3 = pending > 3 && found[3] == v;
                if (h0) { found[0] = found[1]; found[1] = found[2]; found[2] = found[3]; }
                else if (h1) { found[1] = found[2]; found[2] = found[3]; }
                else if (h2) { found[2] = found[3]; }
                if (h0 || h1 || h2 || h3) --pending;
            }
        }
        if (on && sub == 0) {
            if (any_over) list3[atomicAdd(&counters_out[CTR_N3], 1u)] = qi;
            else if (add) {
                const uint32_t cur_ed = best_ed[qi];
                uint32_t t = add, bi = midx;
                if (cur_ed == 2u) { t += n_ties[qi]; const uint32_t o2 = best_idx[qi]; bi = o2 < bi ? o2 : bi; }
                best_idx[qi] = bi; best_ed[qi] = 2; n_ties[qi] = (uint16_t)(t > 0xFFFFu ? 0xFFFFu : t);
            }
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------
static uint64_t wl_fingerprint(const uint32_t* wl, uint32_t nw)
{
    uint64_t h = 0xCBF29CE484222325ull ^ nw;                 // FNV-1a over the words, in caller order
    for (uint32_t i = 0; i < nw; ++i) { h ^= wl[i]; h *= 0x100000001B3ull; }
    return h ? h : 1;
}

int bdg_whitelist_load_impl(bdg_ctx* ctx, const uint32_t* wl, uint32_t nw)
{
    // the same list again (bdg_nearest16 is called per batch with the same centres): everything is still in place
    const uint64_t fp = nw ? wl_fingerprint(wl, nw) : 0;
    if (nw && ctx->w_n == nw && ctx->w_fp == fp && ctx->w_host_sorted.size() == nw) {
        // (the fingerprint only says "probably": a collision must not match later queries against the old list)
        bool same = true;
        for (uint32_t i = 0; i < nw && same; ++i) same = wl[ctx->w_host_order[i]] == ctx->w_host_sorted[i];
        if (same) return BDG_OK;
    }
    // nothing is published until every table of the new list is complete: a failure below leaves "no whitelist loaded"
    { const int rcd = bdg_launch_deferred_match(ctx, false); if (rcd) return rcd; }      // a match that waits meant the old list
    if (ctx->aux_pending) { BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->aux_stream)); ctx->aux_pending = false; }
    ctx->w_n = 0; ctx->w_fp = 0; ctx->w_probe_ready = false; ctx->w_delins_ready = false;
    if (nw == 0) return BDG_OK;
    std::vector<uint32_t> order(nw);
    std::iota(order.begin(), order.end(), 0u);
    bool sorted_in = true;
    for (uint32_t i = 1; i < nw; ++i) if (wl[i - 1] >= wl[i]) { sorted_in = false; break; }
    if (!sorted_in) std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return wl[a] < wl[b] || (wl[a] == wl[b] && a < b); });
    std::vector<uint32_t> srt(nw);
    for (uint32_t i = 0; i < nw; ++i) srt[i] = wl[order[i]];
    for (uint32_t i = 1; i < nw; ++i)
        if (srt[i] == srt[i - 1]) return bdg_fail(ctx, BDG_E_ARG, "whitelist entries must be distinct");
    int pbits = 8;
    while (pbits < 20 && (1u << pbits) < nw) ++pbits;
    const int bbits = pbits + 4 > 24 ? 24 : pbits + 4;
    std::vector<uint32_t> prefix((size_t(1) << pbits) + 1, 0u), bitmap(size_t(1) << (bbits - 5), 0u);
    for (uint32_t i = 0; i < nw; ++i) {
        prefix[(srt[i] >> (32 - pbits)) + 1]++;
        const uint32_t b = srt[i] >> (32 - bbits);
        bitmap[b >> 5] |= 1u << (b & 31u);
    }
    for (size_t p = 0; p < (size_t(1) << pbits); ++p) prefix[p + 1] += prefix[p];
    int rc;
    if ((rc = bdg_reserve(ctx, ctx->w_sorted, sizeof(uint32_t) * nw))) return rc;
    if ((rc = bdg_reserve(ctx, ctx->w_orig, sizeof(uint32_t) * nw))) return rc;
    if ((rc = bdg_reserve(ctx, ctx->w_prefix, sizeof(uint32_t) * prefix.size()))) return rc;
    if ((rc = bdg_reserve(ctx, ctx->w_bitmap, sizeof(uint32_t) * bitmap.size()))) return rc;
    BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    BDG_HIP_TRY(ctx, hipMemcpy(ctx->w_sorted.p, srt.data(), sizeof(uint32_t) * nw, hipMemcpyHostToDevice));
    BDG_HIP_TRY(ctx, hipMemcpy(ctx->w_orig.p, order.data(), sizeof(uint32_t) * nw, hipMemcpyHostToDevice));
    BDG_HIP_TRY(ctx, hipMemcpy(ctx->w_prefix.p, prefix.data(), sizeof(uint32_t) * prefix.size(), hipMemcpyHostToDevice));
    BDG_HIP_TRY(ctx, hipMemcpy(ctx->w_bitmap.p, bitmap.data(), sizeof(uint32_t) * bitmap.size(), hipMemcpyHostToDevice));
    ctx->w_pbits = pbits; ctx->w_bbits = bbits;
    ctx->w_host_sorted.swap(srt);
    ctx->w_host_order.swap(order);
    ctx->w_n = nw; ctx->w_fp = fp;            // the scan path is usable from here; the probe index is built on first use
    return BDG_OK;
}

// The neighbourhood-probe index, built when the probe path first needs it on a whitelist, not by bdg_whitelist_load (a call with
// max_ed > 2, a forced scan or a small job never needs it): the block-pair tables for pass 1 (25 + 50 MB at 737 K entries),
// and - only once a call with max_ed = 2 reaches pass 2 - the deletion-variant maps and the sorted variant entries.
static int build_pair_tables(bdg_ctx* ctx)
{
    const uint32_t nw = ctx->w_n;
    const std::vector<uint32_t>& srt = ctx->w_host_sorted;
    const std::vector<uint32_t>& order = ctx->w_host_order;
    int rc;
    // six block-pair tables as chains of 16-word blocks of 30 16-bit entries (see PairTables)
    std::vector<uint32_t> prank(6ull * 65536ull * 16ull, 0u), pidx(6ull * 65536ull * 32ull, 0u);
    for (int p = 0; p < 6; ++p) {
        const int bi = p < 3 ? 0 : (p < 5 ? 1 : 2);
        const int bj = p < 3 ? p + 1 : (p < 5 ? p - 1 : 3);
        int bk = -1, bl = -1;
        for (int b = 0; b < 4; ++b) if (b != bi && b != bj) { if (bk < 0) bk = b; else bl = b; }
        auto key = [&](uint32_t r) { return ((r >> (8 * bi)) & 0xFFu) | (((r >> (8 * bj)) & 0xFFu) << 8); };
        auto rest = [&](uint32_t r) { return ((r >> (8 * bk)) & 0xFFu) | (((r >> (8 * bl)) & 0xFFu) << 8); };
        std::vector<uint32_t> tail(65536);                 // block currently being filled, per bucket
        for (uint32_t k = 0; k < 65536u; ++k) tail[k] = (uint32_t)p * 65536u + k;
        for (uint32_t i = 0; i < nw; ++i) {                // ascending rank inside a bucket
            const uint32_t k = key(srt[i]);
            uint32_t blk = tail[k];
            uint32_t cnt = prank[(size_t)blk * 16] & 0xFFu;
            if (cnt == PAIR_BLOCK_ENTRIES) {               // chain a fresh block
                const uint32_t nb = (uint32_t)(prank.size() / 16);
                if (nb >= (1u << 24)) return bdg_fail(ctx, BDG_E_ARG, "whitelist too large for the pair tables");
                prank.resize(prank.size() + 16, 0u); pidx.resize(pidx.size() + 32, 0u);
                prank[(size_t)blk * 16] |= nb << 8;
                tail[k] = blk = nb; cnt = 0;
            }
            prank[(size_t)blk * 16 + 1 + (cnt >> 1)] |= rest(srt[i]) << (16 * (cnt & 1u));
            pidx[(size_t)blk * 32 + cnt] = order[i];
            prank[(size_t)blk * 16] = (prank[(size_t)blk * 16] & ~0xFFu) | (cnt + 1u);
        }
    }
    if ((rc = bdg_reserve(ctx, ctx->w_pent, sizeof(uint32_t) * (prank.size() + pidx.size())))) return rc;
    ctx->w_pwords = prank.size();
    BDG_HIP_TRY(ctx, hipMemcpy(ctx->w_pent.p, prank.data(), sizeof(uint32_t) * prank.size(), hipMemcpyHostToDevice));
    BDG_HIP_TRY(ctx, hipMemcpy(static_cast<uint32_t*>(ctx->w_pent.p) + prank.size(), pidx.data(), sizeof(uint32_t) * pidx.size(), hipMemcpyHostToDevice));
    ctx->w_probe_ready = true;
    return BDG_OK;
}

static int build_delins_index(bdg_ctx* ctx)
{
    const uint32_t nw = ctx->w_n;
    int rc;
    if ((rc = bdg_reserve(ctx, ctx->w_delmap, size_t(4) << 27))) return rc;          // four copies of 2^30 bits
    BDG_HIP_TRY(ctx, hipMemsetAsync(ctx->w_delmap.p, 0, size_t(4) << 27, ctx->stream));
    {
        // deletion variants: map bits, the directory (how many variants per bucket -> where each bucket starts), the entries
        const size_t npairs = 16ull * nw;
        if (npairs >= (size_t(1) << 31)) return bdg_fail(ctx, BDG_E_ARG, "whitelist too large");
        const size_t ndir = (size_t)DV_DIR_N + 3;
        const uint32_t nscan = (uint32_t)((ndir + bdgpart::SCAN_SPAN - 1) / bdgpart::SCAN_SPAN);
        if ((rc = bdg_reserve(ctx, ctx->w_dv, sizeof(uint32_t) * (4 * npairs + ndir + nscan + 8)))) return rc;
        auto* dv_ent = static_cast<uint4*>(ctx->w_dv.p);
        auto* dv_dir = static_cast<uint32_t*>(ctx->w_dv.p) + 4 * npairs;
        auto* sums = dv_dir + ndir;
        const auto* srt = static_cast<const uint32_t*>(ctx->w_sorted.p);
        const auto* org = static_cast<const uint32_t*>(ctx->w_orig.p);
        const dim3 grid((uint32_t)((npairs + 255) / 256));
        BDG_HIP_TRY(ctx, hipMemsetAsync(dv_dir, 0, sizeof(uint32_t) * ndir, ctx->stream));
        hipLaunchKernelGGL(k_build_delmap<false>, grid, dim3(256), 0, ctx->stream, srt, org, nw, static_cast<uint32_t*>(ctx->w_delmap.p), dv_dir, dv_ent);
        hipLaunchKernelGGL(bdgpart::k_scan_blocks, dim3(nscan), dim3(1024), 0, ctx->stream, dv_dir, (unsigned long long)ndir, sums);
        hipLaunchKernelGGL(bdgpart::k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, sums, nscan);
        hipLaunchKernelGGL(bdgpart::k_scan_add, dim3(nscan), dim3(1024), 0, ctx->stream, dv_dir, (unsigned long long)ndir, sums);
        hipLaunchKernelGGL(k_build_delmap<true>, grid, dim3(256), 0, ctx->stream, srt, org, nw, static_cast<uint32_t*>(ctx->w_delmap.p), dv_dir, dv_ent);
        BDG_HIP_TRY(ctx, hipGetLastError());
    }
    BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->w_delins_ready = true;
    return BDG_OK;
}

// Pair evaluations below which the exhaustive scan is taken when the probe index does not exist yet: building the index costs
// tens of milliseconds (and up to 0.9 GB of tables), the scan does about 3 x 10^11 pair evaluations a second.  Stage 2's
// --high_sens pass (a few hundred thousand unassigned barcodes against ~5,000 centres, barcode_graph.py:370-385) stays far
// below it; a whitelist-sized list (737 K entries) crosses it with 5,500 queries.
constexpr uint64_t SCAN_PAIR_EVALS_MAX = 4000000000ull;

int bdg_nearest16_launch(bdg_ctx* ctx, const uint32_t* d_q, uint32_t qstride, int recs, uint32_t nq, uint32_t max_ed,
                         uint32_t* d_best_idx, uint8_t* d_best_ed, uint16_t* d_n_ties)
{
    if (nq == 0) return BDG_OK;
    if (ctx->w_n == 0) return bdg_fail(ctx, BDG_E_ARG, "no whitelist loaded (bdg_whitelist_load)");
    hipStream_t st = ctx->launch_stream ? ctx->launch_stream : ctx->stream;
    const auto* srt = static_cast<const uint32_t*>(ctx->w_sorted.p);
    const auto* org = static_cast<const uint32_t*>(ctx->w_orig.p);
    // automatic: the probe path when it applies (max_ed <= 2) and either its index exists already or the job is large enough to
    // pay for building it
    const bool built = ctx->w_probe_ready && (max_ed < 2 || ctx->w_delins_ready);
    const bool probe = ctx->n16_algo == 2 || (ctx->n16_algo == 0 && max_ed <= 2 && (built || (uint64_t)ctx->w_n * nq > SCAN_PAIR_EVALS_MAX));
    if (ctx->n16_algo == 2 && max_ed > 2) return bdg_fail(ctx, BDG_E_ARG, "probe path needs max_ed <= 2");
    if (!probe) {
        ScopedKernelTimer tm(ctx, "k_nearest_scan");
        hipLaunchKernelGGL(k_nearest_scan, dim3((nq + 255) / 256), dim3(256), 0, st, d_q, qstride, recs, (const uint32_t*)nullptr, nq,
                           (const uint32_t*)nullptr, srt, org, ctx->w_n, max_ed, d_best_idx, d_best_ed, d_n_ties);
        BDG_HIP_TRY(ctx, hipGetLastError());
        return BDG_OK;
    }
    int rc;
    if (!ctx->w_probe_ready && (rc = build_pair_tables(ctx))) return rc;
    if (max_ed >= 2 && !ctx->w_delins_ready && (rc = build_delins_index(ctx))) return rc;
    if ((rc = bdg_reserve(ctx, ctx->n_list, sizeof(uint32_t) * (2ull * LSH + 1ull) * nq))) return rc;
    if ((rc = bdg_reserve(ctx, ctx->n_counters, NCTR_BYTES))) return rc;
    auto* list2 = static_cast<uint2*>(ctx->n_list.p);                          // LSH segments of nq {index, query} entries
    auto* list3 = reinterpret_cast<uint32_t*>(list2 + (size_t)LSH * nq);       // overflow list, nq indices
    auto* counters = static_cast<uint32_t*>(ctx->n_counters.p);
    BDG_HIP_TRY(ctx, hipMemsetAsync(counters, 0, NCTR_BYTES, st));
    PairTables pt{ static_cast<const uint32_t*>(ctx->w_pent.p), static_cast<const uint32_t*>(ctx->w_pent.p) + ctx->w_pwords,
                   static_cast<const uint32_t*>(ctx->w_delmap.p), ctx->w_n };
    {
        ScopedKernelTimer tm(ctx, "k_nearest_pairs");
        hipLaunchKernelGGL(k_nearest_pairs, dim3((nq + 255) / 256), dim3(256), 0, st, d_q, qstride, recs, nq, pt, max_ed,
                           d_best_idx, d_best_ed, d_n_ties, list2, counters);
    }
    if (max_ed >= 2) {
        {
            ScopedKernelTimer tm(ctx, "k_nearest_delins");
            const uint32_t grid = std::min<uint32_t>((nq + 63) / 64, 256u * 8u);
            const size_t npairs = 16ull * ctx->w_n;
            hipLaunchKernelGGL(k_nearest_delins, dim3(grid), dim3(256), 0, st, list2, nq, counters, pt.delmap,
                               static_cast<const uint4*>(ctx->w_dv.p), static_cast<const uint32_t*>(ctx->w_dv.p) + 4 * npairs,
                               d_best_idx, d_best_ed, d_n_ties, list3, counters);
        }
        // queries whose hit list overflowed (one lane found more than 4 distinct entries): exhaustive
        // scan of just those; the list length stays on the device, so no host round trip
        {
            ScopedKernelTimer tm(ctx, "k_nearest_scan_overflow");
            hipLaunchKernelGGL(k_nearest_scan, dim3(64), dim3(256), 0, st, d_q, qstride, recs, list3, 0u, counters + CTR_N3,
                               srt, org, ctx->w_n, max_ed, d_best_idx, d_best_ed, d_n_ties);
        }
    }
    BDG_HIP_TRY(ctx, hipGetLastError());
    return BDG_OK;
}
