// Grouping by key without a sort library: bucket partition in two levels, exact and without one atomic on global memory.
//
// What the reference does with dict buckets (index.py:29-35, barcode_graph.py:192-204) needs GROUPS of equal keys, and
// the distinct-barcode count needs them in ascending order; neither needs a full radix sort of the entries through HBM.
//
//   level 1  the producer runs twice over tiles of its input (a tile = what one block handles): the first run counts, per
//            tile, how many entries go to each of NB1 coarse buckets (histogram in LDS, one column of a buckets x tiles
//            matrix per tile); k_part_colscan turns the matrix's rows into every tile's write position inside each
//            bucket, k_part_bases the row totals into the buckets' places (and picks NB2 from the sum, on the device:
//            the host never waits for a count); the second run writes each entry to its place (cursor per bucket in LDS).
//   level 2  k_part_split: one block per coarse bucket reads it twice (the second time out of L2), counts NB2 sub-buckets
//            in LDS and writes the entries grouped by sub-bucket into a second buffer, with the start of every fine
//            bucket in `fstart`.
//   then     a consumer takes one fine bucket (about a thousand entries) into LDS and finishes the grouping there.
//
// Every entry is written twice and read three times (the radix sort this replaces: ten times each).  gfx950 only.
#pragma once

#include "bdg_common.hpp"

namespace bdgpart {

constexpr uint32_t NB1_MAX = 4096;     // coarse buckets (the producer's LDS histogram)
constexpr uint32_t NB2_MAX = 4096;     // sub-buckets of one coarse bucket (k_part_split's LDS histograms)

// geom[]: what the device decides and the later kernels read
enum { G_L2 = 0, G_M_LO = 1, G_M_HI = 2, G_FLAGS = 3, G_OVF = 4 /* a consumer's list of buckets left to a second kernel */, G_WORDS = 8 };

#if defined(__HIPCC__)

// exclusive prefix sums over the values the threads of a block hold (one each); every thread calls; s_w: one word per wave + 1
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w, uint32_t& total)
{
    constexpr int NW = THREADS / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const uint32_t x = s_w[w]; all += x; if (w < wv) before += x; }
    total = all;
    __syncthreads();                                   // (s_w is free again)
    return before + incl - v;
}

// Where a tile's entries go inside a coarse bucket.  The counting run leaves the tiles' counts as a matrix with one ROW PER
// BUCKET (hist[b * ntiles + t]: a block stores and later loads its column with scattered 4-byte words, a few microseconds'
// worth), so that the scan below reads and writes whole rows: one block per bucket, a thread takes consecutive tiles.
// (Two other forms were measured.  Tile-major rows with a scan that strides through them: 14 us for 256 buckets, 50 us
// for 1,024 - as long as the kernels around it at config 3 and in the distinct count.  No matrix at all - the counting run
// ending with one returning atomic add per tile and bucket on the bucket's total: invisible behind config 5's long counting
// run, but 0.5 - 2 M adds on a few hundred addresses behind a short one cost 50 - 100 us.)
static __global__ __launch_bounds__(256)
void k_part_colscan(uint32_t* __restrict__ hist, uint32_t ntiles, uint32_t nb1, uint32_t* __restrict__ tot)
{
    __shared__ uint32_t s_w[5];
    uint32_t* const row = hist + (size_t)blockIdx.x * ntiles;
    const uint32_t per = (ntiles + 255u) / 256u;
    const uint32_t t0 = threadIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    uint32_t sum = 0;
    for (uint32_t t = t0; t < t1; ++t) sum += row[t];
    uint32_t total;
    uint32_t run = block_excl_scan<256>(sum, s_w, total);
    for (uint32_t t = t0; t < t1; ++t) { const uint32_t c = row[t]; row[t] = run; run += c; }
    if (threadIdx.x == 0) tot[blockIdx.x] = total;
}

// base[b] = where coarse bucket b starts (64-bit sums: the total is checked, not assumed), base[nb1] = m, the number of
// entries; geom: log2 of the sub-bucket count that brings a fine bucket to about `target` entries (at most l2_max), m, and
// flag bit 0 when m does not fit the limit the caller can index (the caller then cuts its work smaller).
static __global__ __launch_bounds__(1024)
void k_part_bases(const uint32_t* __restrict__ tot, uint32_t nb1, uint32_t target, uint32_t l2_max, unsigned long long limit,
                  unsigned long long* __restrict__ base, uint32_t* __restrict__ geom)
{
    __shared__ unsigned long long s_w[17];
    constexpr uint32_t PT = NB1_MAX / 1024u;           // consecutive buckets per thread
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long c[PT], v = 0;
#pragma unroll
    for (uint32_t j = 0; j < PT; ++j) { const uint32_t b = threadIdx.x * PT + j; c[j] = b < nb1 ? tot[b] : 0ull; v += c[j]; }
    unsigned long long incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) { const unsigned long long o = __shfl_up(incl, s); if (lane >= s) incl += o; }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    unsigned long long before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const unsigned long long x = s_w[w]; all += x; if (w < wv) before += x; }
    unsigned long long run = before + incl - v;
#pragma unroll
    for (uint32_t j = 0; j < PT; ++j) { const uint32_t b = threadIdx.x * PT + j; if (b < nb1) base[b] = run; run += c[j]; }
    if (threadIdx.x == 0) {
        base[nb1] = all;
        uint32_t l2 = 0;
        while (l2 < l2_max && (all >> l2) > (unsigned long long)nb1 * target) ++l2;
        geom[G_L2] = l2; geom[G_M_LO] = (uint32_t)all; geom[G_M_HI] = (uint32_t)(all >> 32);
        geom[G_FLAGS] = all > limit ? 1u : 0u;
        geom[G_OVF] = 0u;
    }
}

// Inclusive prefix sums of n 32-bit words in place (n up to 2^32; sums must fit 32 bits): 16,384 words per block - every
// block scans its stretch and leaves its total, one block scans the totals, every block adds what lies in front of it.
// (The directory of the whitelist's deletion variants, 2^25 words, once per whitelist: hipCUB's scan stood here.)
constexpr uint32_t SCAN_SPAN = 16384;
static __global__ __launch_bounds__(1024)
void k_scan_blocks(uint32_t* __restrict__ a, unsigned long long n, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t s_w[17];
    const unsigned long long i0 = (unsigned long long)blockIdx.x * SCAN_SPAN + threadIdx.x * 16ull;
    uint32_t v[16], sum = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { v[j] = i0 + j < n ? a[i0 + j] : 0u; sum += v[j]; }
    uint32_t total;
    uint32_t run = block_excl_scan<1024>(sum, s_w, total);
#pragma unroll
    for (int j = 0; j < 16; ++j) { run += v[j]; if (i0 + j < n) a[i0 + j] = run; }
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
static __global__ __launch_bounds__(1024)
void k_scan_sums(uint32_t* __restrict__ sums, uint32_t nblocks)       // exclusive, one block
{
    __shared__ uint32_t s_w[17];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 1024u) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < nblocks ? sums[b] : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan<1024>(v, s_w, total);
        if (b < nblocks) sums[b] = carry + ex;
        carry += total;
    }
}
static __global__ __launch_bounds__(1024)
void k_scan_add(uint32_t* __restrict__ a, unsigned long long n, const uint32_t* __restrict__ sums)
{
    const uint32_t add = sums[blockIdx.x];
    if (add == 0u) return;
    const unsigned long long i0 = (unsigned long long)blockIdx.x * SCAN_SPAN + threadIdx.x * 16ull;
#pragma unroll
    for (int j = 0; j < 16; ++j) if (i0 + j < n) a[i0 + j] += add;
}

// One block per coarse bucket: its entries grouped by sub-bucket (bits [sh_top - l2, sh_top) of an entry) into `out`,
// the places of the fine buckets into fstart[b << l2 | j], fstart[nb1 << l2] = m.  Any bucket size: the bucket is streamed
// twice, first for the sub-buckets' sizes, then in rounds of 64 KB that are grouped inside LDS before they leave, so that
// what a wave stores is runs of neighbours (a store of 64 scattered words is 64 transactions at the L2, and 35 M of those
// cost more than the rest of the pass: measured 0.31 ms against 0.10 at 35.6 M entries).
template <class E>
__global__ __launch_bounds__(1024)
void k_part_split(const E* __restrict__ in, E* __restrict__ out, const unsigned long long* __restrict__ base,
                  const uint32_t* __restrict__ geom, uint32_t nb1, uint32_t sh_top, uint32_t* __restrict__ fstart)
{
    constexpr uint32_t CHK = 65536u / sizeof(E), PT = CHK / 1024u, PB = NB2_MAX / 1024u;
    __shared__ E s_stage[CHK];
    __shared__ uint32_t s_g[NB2_MAX];                 // where each sub-bucket's next entry goes
    __shared__ uint32_t s_c[NB2_MAX];                 // per round: the sub-bucket's count, then its start inside the round
    __shared__ uint32_t s_w[17];
    if (geom[G_FLAGS] & 1u) return;
    const uint32_t l2 = geom[G_L2], nb2 = 1u << l2, sh = sh_top - l2, mask = nb2 - 1u;
    // a thread owns PB consecutive sub-buckets (all of them beyond nb2 stay empty)
    auto scan_counts = [&](uint32_t (&v)[PB], uint32_t& total) -> uint32_t {
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < PB; ++j) { const uint32_t b = threadIdx.x * PB + j; v[j] = b < nb2 ? s_c[b] : 0u; sum += v[j]; }
        return block_excl_scan<1024>(sum, s_w, total);
    };
    for (uint32_t b = blockIdx.x; b < nb1; b += gridDim.x) {
        const unsigned long long s = base[b];
        const uint32_t c = (uint32_t)(base[b + 1] - s);
        for (uint32_t j = threadIdx.x; j < nb2; j += 1024u) s_c[j] = 0u;
        __syncthreads();
        // (four loads in flight per thread: a block streams its bucket alone on its compute unit)
        for (uint32_t i = threadIdx.x; i < c; i += 4096u) {
            E e[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) e[u] = i + u * 1024u < c ? in[s + i + u * 1024u] : E(0);
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) if (i + u * 1024u < c) atomicAdd(&s_c[(uint32_t)(e[u] >> sh) & mask], 1u);
        }
        __syncthreads();
        {
            uint32_t v[PB], total;
            uint32_t at = (uint32_t)s + scan_counts(v, total);
#pragma unroll
            for (uint32_t j = 0; j < PB; ++j) {
                const uint32_t sb = threadIdx.x * PB + j;
                if (sb < nb2) { s_g[sb] = at; fstart[((size_t)b << l2) + sb] = at; }
                at += v[j];
            }
            if (b == nb1 - 1u && threadIdx.x == 0) fstart[(size_t)nb1 << l2] = (uint32_t)base[nb1];
        }
        for (uint32_t c0 = 0; c0 < c; c0 += CHK) {
            const uint32_t nc = c - c0 < CHK ? c - c0 : CHK;
            for (uint32_t j = threadIdx.x; j < nb2; j += 1024u) s_c[j] = 0u;
            __syncthreads();
            E e[PT];
            uint32_t rk[PT];
#pragma unroll
            for (uint32_t u = 0; u < PT; ++u) e[u] = u * 1024u + threadIdx.x < nc ? in[s + c0 + u * 1024u + threadIdx.x] : E(0);
#pragma unroll
            for (uint32_t u = 0; u < PT; ++u) rk[u] = u * 1024u + threadIdx.x < nc ? atomicAdd(&s_c[(uint32_t)(e[u] >> sh) & mask], 1u) : 0u;
            __syncthreads();
            uint32_t v[PB], total;
            uint32_t cs = scan_counts(v, total);
#pragma unroll
            for (uint32_t j = 0; j < PB; ++j) { const uint32_t sb = threadIdx.x * PB + j; if (sb < nb2) s_c[sb] = cs; cs += v[j]; }
            __syncthreads();
#pragma unroll
            for (uint32_t u = 0; u < PT; ++u) if (u * 1024u + threadIdx.x < nc) s_stage[s_c[(uint32_t)(e[u] >> sh) & mask] + rk[u]] = e[u];
            __syncthreads();
#pragma unroll
            for (uint32_t u = 0; u < PT; ++u) {
                const uint32_t i = u * 1024u + threadIdx.x;
                if (i < nc) {
                    const E x = s_stage[i];
                    const uint32_t j = (uint32_t)(x >> sh) & mask;
                    out[s_g[j] + (i - s_c[j])] = x;
                }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < PB; ++j) { const uint32_t sb = threadIdx.x * PB + j; if (sb < nb2) s_g[sb] += v[j]; }
        }
        __syncthreads();
    }
}

#endif  // __HIPCC__

}  // namespace bdgpart
