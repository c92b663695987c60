// Distinct-barcode counting on the device (reference BarcodeGraph.index_bc_single_thread,
// barcode_graph.py:192-204): from the extraction records of a batch, the distinct 16-base
// barcodes ascending, how often each occurs and where it occurs first (the reference's `counts` dict is in
// first-occurrence order, which get_cluster_centers depends on, barcode_graph.py:253-255).
//
// No sort library: the records' (rank, index) pairs are grouped by the rank's top bits in the two bucket levels of
// bdg_partition.hpp (exact places from counting passes, no atomics on global memory), and a block then sorts one fine
// bucket - about a thousand pairs - inside LDS as 64-bit words rank << 32 | index (bitonic network), so that equal ranks
// stand together with their smallest index first: a run's length is the multiplicity, its first element the first
// occurrence.  Buckets ascend with the rank, so the runs of bucket after bucket are the answer in order; two small passes
// (sums of the buckets' run counts, copy) close the gaps.  A bucket larger than the LDS capacity - a barcode seen tens of
// thousands of times, or an adversarial input - is taken apart by the next key bits, eight at a time, with a run of one
// repeated rank reduced in a streaming pass whatever its length.
#include "bdg_common.hpp"
#include "bdg_partition.hpp"

namespace {

constexpr uint32_t DS_SLOTS = 2048;              // slots of the hash table a block counts one bucket's ranks in,
constexpr uint32_t DS_DCAP = 1536;               // different ranks it may hold,
constexpr uint32_t DS_CAP = 1024;                // pairs a fine bucket should hold on average (about 600 different ranks on bench data)
constexpr int DS_THREADS = 256;

// level 1 (bdg_partition.hpp), run twice over tiles of records: counts per coarse bucket (the rank's top l1 bits), then the
// pairs at their places.  Records without a usable barcode take no part; out_n[1] counts the 16-base barcodes with a non-ACGT base.
template <bool EMIT>
__global__ __launch_bounds__(256)
void k_distinct_rows(const bdg_extract_rec* __restrict__ recs, uint32_t n, uint32_t per_tile, uint32_t l1,
                     uint32_t* __restrict__ hist, uint32_t* __restrict__ tot, const unsigned long long* __restrict__ base,
                     const uint32_t* __restrict__ geom, unsigned long long* __restrict__ ent, uint32_t* __restrict__ out_n)
{
    __shared__ uint32_t s_h[bdgpart::NB1_MAX];
    const uint32_t nb1 = 1u << l1, sh = 32u - l1;
    const uint32_t row0 = blockIdx.x * per_tile;
    const uint32_t row1 = n - row0 < per_tile ? n : row0 + per_tile;
    if (EMIT && (geom[bdgpart::G_FLAGS] & 1u)) return;
    for (uint32_t i = threadIdx.x; i < nb1; i += 256u) s_h[i] = EMIT ? (uint32_t)base[i] + hist[(size_t)i * gridDim.x + blockIdx.x] : 0u;
    __syncthreads();
    for (uint32_t i = row0 + threadIdx.x; i < row1; i += 256u) {
        const bdg_extract_rec r = recs[i];
        const bool ok = r.valid && (r.flags & BDG_FLAG_RANK_OK);
        if (ok) {
            if (EMIT) ent[atomicAdd(&s_h[r.bc_rank >> sh], 1u)] = (unsigned long long)r.bc_rank << 32 | i;
            else atomicAdd(&s_h[r.bc_rank >> sh], 1u);
        } else if (!EMIT && r.valid && (r.flags & BDG_FLAG_BC16)) atomicAdd(&out_n[1], 1u);
    }
    if (!EMIT) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb1; i += 256u) hist[(size_t)i * gridDim.x + blockIdx.x] = s_h[i];      // (one row per bucket: k_part_colscan)
    }
}

// One fine bucket per block at a time: the runs (rank, how often, first index) of its pairs, written in ascending order
// to the bucket's own stretch of the temporary arrays, their number to nuniq[bucket].
// Neither a comparison network (a bitonic sort of a thousand pairs is 55 phases with a block barrier each) nor anything
// quadratic in how often a barcode repeats (a cell's exact barcode comes a thousand times in a million reads): the pairs
// stream through a hash table in LDS - one atomic compare-and-swap finds or claims the rank's slot, an add counts, a minimum
// keeps the first index - so that a bucket may hold any number of pairs as long as its DIFFERENT ranks fit the table.
// The different ranks are then put in order: a counting pass over bins named by the rank's next bits (order-preserving,
// about one rank per bin), and inside a bin a rank's place is the number of smaller ones.  A bucket with more different
// ranks than the table takes (a device-side choice of the bucket count from the number of pairs normally prevents it) is
// taken apart by the next two key bits, each part through the same code.
__global__ __launch_bounds__(DS_THREADS)
void k_distinct_buckets(const unsigned long long* __restrict__ ent, const uint32_t* __restrict__ fstart,
                        const unsigned long long* __restrict__ base, const uint32_t* __restrict__ geom, uint32_t l1,
                        uint32_t* __restrict__ t_uniq, uint32_t* __restrict__ t_count, uint32_t* __restrict__ t_first,
                        uint32_t* __restrict__ nuniq)
{
    constexpr uint32_t SLOTS = DS_SLOTS, NBIN = DS_SLOTS, LBIN = 31u - (uint32_t)__builtin_clz(DS_SLOTS), SPER = SLOTS / DS_THREADS;
    __shared__ uint32_t s_hk[SLOTS], s_hc[SLOTS], s_hf[SLOTS];      // the table: rank, how often, first index
    __shared__ uint32_t s_bin[NBIN + 1];
    __shared__ unsigned long long s_e[DS_DCAP + DS_THREADS];         // rank << 32 | slot, grouped by bin
    __shared__ uint32_t st_bits[14], st_val[14], st_next[14];
    __shared__ uint32_t s_w[DS_THREADS / 64 + 1];
    __shared__ uint32_t s_new;
    if (geom[bdgpart::G_FLAGS] & 1u) return;
    const uint32_t l2 = fstart ? geom[bdgpart::G_L2] : 0u;
    const uint32_t nfb = 1u << (l1 + l2);
    for (uint32_t fb = blockIdx.x; fb < nfb; fb += gridDim.x) {
        const uint32_t start = fstart ? fstart[fb] : (uint32_t)base[fb];
        const uint32_t cnt = (fstart ? fstart[fb + 1u] : (uint32_t)base[fb + 1u]) - start;
        uint32_t emitted = 0;
        // the pairs of [start, start + cnt) whose rank has the top `bits` bits `val`: their runs, ascending - unless they hold
        // more than DS_DCAP different ranks (returns false, nothing written)
        auto group_emit = [&](uint32_t bits, uint32_t val) -> bool {
            const uint32_t EMPTY = val == (uint32_t)((1ull << bits) - 1ull) ? 0u : 0xFFFFFFFFu;      // (a rank that does not have this prefix)
#pragma unroll
            for (uint32_t j = 0; j < SPER; ++j) { const uint32_t i = j * DS_THREADS + threadIdx.x; s_hk[i] = EMPTY; s_hc[i] = 0u; s_hf[i] = 0xFFFFFFFFu; }
            if (threadIdx.x == 0) s_new = 0u;
            __syncthreads();
            for (uint32_t i0 = threadIdx.x; i0 < cnt; i0 += 4u * DS_THREADS) {
                unsigned long long e[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) e[u] = i0 + u * DS_THREADS < cnt ? ent[start + i0 + u * DS_THREADS] : 0ull;
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    if (i0 + u * DS_THREADS >= cnt || (uint32_t)(e[u] >> (64u - bits)) != val) continue;
                    if (*(volatile uint32_t*)&s_new > DS_DCAP) continue;           // (too many different ranks: the attempt is void)
                    const uint32_t k = (uint32_t)(e[u] >> 32);
                    uint32_t h = (k * 0x9E3779B1u) >> (32u - LBIN);
                    for (;;) {
                        const uint32_t old = atomicCAS(&s_hk[h], EMPTY, k);
                        if (old == EMPTY) atomicAdd(&s_new, 1u);
                        if (old == EMPTY || old == k) { atomicAdd(&s_hc[h], 1u); atomicMin(&s_hf[h], (uint32_t)e[u]); break; }
                        h = (h + 1u) & (SLOTS - 1u);                                // (never full: at most DS_DCAP + a block's threads claim a slot)
                    }
                }
            }
            __syncthreads();
            const uint32_t D = s_new;
            if (D > DS_DCAP) { __syncthreads(); return false; }
            // the different ranks in order: bins by the next key bits, then by comparison inside a bin
            const uint32_t rb = 32u - bits, lb = rb < LBIN ? rb : LBIN, bsh = rb - lb, bmask = (1u << lb) - 1u;
#pragma unroll
            for (uint32_t j = 0; j < SPER; ++j) s_bin[threadIdx.x * SPER + j] = 0u;
            __syncthreads();
            uint32_t key[SPER], rk[SPER];
#pragma unroll
            for (uint32_t j = 0; j < SPER; ++j) {
                key[j] = s_hk[j * DS_THREADS + threadIdx.x]; rk[j] = 0;
                if (key[j] != EMPTY) rk[j] = atomicAdd(&s_bin[(key[j] >> bsh) & bmask], 1u);
            }
            __syncthreads();
            uint32_t c[SPER], sum = 0;
#pragma unroll
            for (uint32_t j = 0; j < SPER; ++j) { c[j] = s_bin[threadIdx.x * SPER + j]; sum += c[j]; }
            uint32_t held;
            uint32_t run = bdgpart::block_excl_scan<DS_THREADS>(sum, s_w, held);
#pragma unroll
            for (uint32_t j = 0; j < SPER; ++j) { s_bin[threadIdx.x * SPER + j] = run; run += c[j]; }
            if (threadIdx.x == 0) s_bin[NBIN] = held;
            __syncthreads();
#pragma unroll
            for (uint32_t j = 0; j < SPER; ++j)
                if (key[j] != EMPTY) s_e[s_bin[(key[j] >> bsh) & bmask] + rk[j]] = (unsigned long long)key[j] << 32 | (j * DS_THREADS + threadIdx.x);
            __syncthreads();
            for (uint32_t pos = threadIdx.x; pos < D; pos += DS_THREADS) {
                const unsigned long long e = s_e[pos];
                const uint32_t k = (uint32_t)(e >> 32), slot = (uint32_t)e, b = (k >> bsh) & bmask;
                const uint32_t bs = s_bin[b], be = s_bin[b + 1u];
                uint32_t smaller = 0;
                for (uint32_t i = bs; i < be; ++i) smaller += (uint32_t)(s_e[i] >> 32) < k ? 1u : 0u;
                const size_t o = (size_t)start + emitted + bs + smaller;
                t_uniq[o] = k; t_count[o] = s_hc[slot]; t_first[o] = s_hf[slot];
            }
            emitted += D;
            __syncthreads();
            return true;
        };
        if (cnt == 0u) { if (threadIdx.x == 0) nuniq[fb] = 0u; continue; }
        if (group_emit(l1 + l2, fb)) { if (threadIdx.x == 0) nuniq[fb] = emitted; continue; }
        // cold path: the bucket is taken apart by the next key bits, two at a time (ascending, so the output stays in order).
        // The stack lives in LDS (indexing a private array by the depth would put it into scratch memory); thread 0 writes it
        // between two barriers, everybody reads it behind them.
        int sp = 0;
        __syncthreads();
        if (threadIdx.x == 0) { st_bits[0] = l1 + l2; st_val[0] = fb; st_next[0] = 0u; }
        while (sp >= 0) {
            __syncthreads();
            const uint32_t bits = st_bits[sp], val = st_val[sp], ch = st_next[sp];
            __syncthreads();
            const uint32_t w = 32u - bits < 2u ? 32u - bits : 2u;             // (bits < 32: more than one rank shares this prefix)
            if (ch >= (1u << w)) { --sp; continue; }
            if (threadIdx.x == 0) st_next[sp] = ch + 1u;
            const uint32_t cb = bits + w, cv = (val << w) | ch;
            if (group_emit(cb, cv)) continue;
            if (threadIdx.x == 0) { st_bits[sp + 1] = cb; st_val[sp + 1] = cv; st_next[sp + 1] = 0u; }
            ++sp;
        }
        if (threadIdx.x == 0) nuniq[fb] = emitted;
    }
}

// off[b] = runs in the buckets before b (one block; the bucket count is a power of two, read from geom); out_n[0] = all runs
__global__ __launch_bounds__(1024)
void k_distinct_offsets(const uint32_t* __restrict__ nuniq, const uint32_t* __restrict__ geom, uint32_t l1, int have_l2,
                        uint32_t* __restrict__ off, uint32_t* __restrict__ out_n)
{
    __shared__ uint32_t s_w[17];
    if (geom[bdgpart::G_FLAGS] & 1u) return;
    const uint32_t nfb = 1u << (l1 + (have_l2 ? geom[bdgpart::G_L2] : 0u));
    const uint32_t per = (nfb + 1023u) / 1024u;
    const uint32_t b0 = threadIdx.x * per, b1 = b0 + per < nfb ? b0 + per : nfb;
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; ++b) sum += nuniq[b];
    uint32_t total;
    uint32_t run = bdgpart::block_excl_scan<1024>(sum, s_w, total);
    for (uint32_t b = b0; b < b1; ++b) { off[b] = run; run += nuniq[b]; }
    if (threadIdx.x == 0) out_n[0] = total;
}

// the runs of bucket after bucket, side by side
__global__ __launch_bounds__(256)
void k_distinct_compact(const uint32_t* __restrict__ fstart, const unsigned long long* __restrict__ base, const uint32_t* __restrict__ geom,
                        uint32_t l1, const uint32_t* __restrict__ nuniq, const uint32_t* __restrict__ off,
                        const uint32_t* __restrict__ t_uniq, const uint32_t* __restrict__ t_count, const uint32_t* __restrict__ t_first,
                        uint32_t* __restrict__ uniq, uint32_t* __restrict__ count, uint32_t* __restrict__ first)
{
    if (geom[bdgpart::G_FLAGS] & 1u) return;
    const uint32_t nfb = 1u << (l1 + (fstart ? geom[bdgpart::G_L2] : 0u));
    // a wave a bucket (a bucket holds about a thousand pairs, fewer runs)
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t fb = blockIdx.x * 4u + (threadIdx.x >> 6); fb < nfb; fb += gridDim.x * 4u) {
        const uint32_t start = fstart ? fstart[fb] : (uint32_t)base[fb];
        const uint32_t k = nuniq[fb], o = off[fb];
        for (uint32_t j = lane; j < k; j += 64u) { uniq[o + j] = t_uniq[start + j]; count[o + j] = t_count[start + j]; first[o + j] = t_first[start + j]; }
    }
}

// Observed barcodes that come from a stage-1 TSV (badger.py:91-111) instead of from an extraction: the records the
// rest of stage 2 reads (bc_rank, valid, flags) for them, so that counting and assignment run the same device code.
__global__ __launch_bounds__(256)
void k_records_of_observed(const uint32_t* __restrict__ rank, const uint8_t* __restrict__ usable, uint64_t n,
                           bdg_extract_rec* __restrict__ recs)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const bool ok = usable[i] != 0;
    bdg_extract_rec r;
    r.polyT = -1; r.r1_end = -1; r.bc_start = 0; r.umi_start = 0; r.umi_end = 0;
    r.bc_rank = ok ? rank[i] : 0u;
    r.r1_score = 0; r.strand = 0;
    r.valid = ok ? 1 : 0;
    r.flags = ok ? (uint8_t)(BDG_FLAG_RANK_OK | BDG_FLAG_BC16) : (uint8_t)0;
    r.reserved = 0;
    recs[i] = r;
}

// rows[i] = position of values[i * stride] in the ascending array sorted[0..n), NONE when absent: what turns an edge's
// two ranks (the reference keys its edges dict by rank, barcode_graph.py:245-247) into indices of the distinct-barcode
// arrays the stage-2 array code works on.
__global__ __launch_bounds__(256)
void k_rows_of(const uint32_t* __restrict__ sorted, uint32_t n, const uint32_t* __restrict__ values, uint64_t m,
               uint32_t stride, uint32_t* __restrict__ rows)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256ull + threadIdx.x;
    if (i >= m) return;
    const uint32_t v = values[i * stride];
    uint32_t lo = 0, hi = n;                     // first position with sorted[pos] >= v
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (sorted[mid] < v) lo = mid + 1; else hi = mid;
    }
    rows[i] = (lo < n && sorted[lo] == v) ? lo : 0xFFFFFFFFu;
}

}  // namespace

int bdg_records_of_observed_launch(bdg_ctx* ctx, const uint32_t* d_rank, const uint8_t* d_usable, uint64_t n, bdg_extract_rec* d_recs)
{
    if (n == 0) return BDG_OK;
    ScopedKernelTimer tm(ctx, "k_records_of_observed");
    hipLaunchKernelGGL(k_records_of_observed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_rank, d_usable, n, d_recs);
    BDG_HIP_TRY(ctx, hipGetLastError());
    return BDG_OK;
}

int bdg_rows_of_launch(bdg_ctx* ctx, const uint32_t* d_sorted, uint32_t n, const uint32_t* d_values, uint64_t m,
                       uint32_t stride, uint32_t* d_rows)
{
    if (m == 0) return BDG_OK;
    ScopedKernelTimer tm(ctx, "k_rows_of");
    hipLaunchKernelGGL(k_rows_of, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, d_sorted, n, d_values, m, stride, d_rows);
    BDG_HIP_TRY(ctx, hipGetLastError());
    return BDG_OK;
}

int bdg_distinct_launch(bdg_ctx* ctx, const bdg_extract_rec* d_recs, uint32_t n,
                        uint32_t* d_uniq, uint32_t* d_count, uint32_t* d_first, uint32_t* d_n)
{
    hipStream_t st = ctx->stream;
    BDG_HIP_TRY(ctx, hipMemsetAsync(d_n, 0, 8, st));
    if (n == 0) return BDG_OK;
    if (!ctx->g_cus) {
        hipDeviceProp_t prop;
        BDG_HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->g_cus_distinct = prop.multiProcessorCount;
    }
    const uint32_t cus = (uint32_t)(ctx->g_cus ? ctx->g_cus : ctx->g_cus_distinct);
    // geometry: coarse buckets by the rank's top l1 bits; below about a million records they are the fine buckets already
    // (no second level), beyond that the device picks the sub-bucket count from the number of usable records
    uint32_t l1 = 8;
    while (l1 < 10u && (n >> l1) > DS_CAP) ++l1;
    const bool two_levels = (n >> l1) > DS_CAP;
    const uint32_t nb1 = 1u << l1, l2_max = two_levels ? 10u : 0u;
    uint32_t per_tile = ((n + cus * 8u - 1u) / (cus * 8u) + 255u) & ~255u;
    const uint32_t ntiles = (n + per_tile - 1u) / per_tile;
    const size_t nfb_max = (size_t)nb1 << l2_max;
    // workspace: base u64 [nb1 + 1] | pairs u64 [n] twice | hist [ntiles][nb1] | tot | geom | fstart | nuniq | off | temporary runs 3 x [n]
    const size_t w_hist = (size_t)ntiles * nb1;
    const size_t need = 8 * ((size_t)nb1 + 1 + 2 * (size_t)n) + 4 * (w_hist + nb1 + bdgpart::G_WORDS + (nfb_max + 1) + 2 * nfb_max + 3 * (size_t)n) + 256;
    int rc;
    if ((rc = bdg_reserve(ctx, ctx->g_tmp1, need))) return rc;
    auto* base = static_cast<unsigned long long*>(ctx->g_tmp1.p);
    auto* e_a = base + nb1 + 1;
    auto* e_b = e_a + n;
    auto* hist = reinterpret_cast<uint32_t*>(e_b + n);
    auto* tot = hist + w_hist;
    auto* geom = tot + nb1;
    auto* fstart = geom + bdgpart::G_WORDS;
    auto* nuniq = fstart + nfb_max + 1;
    auto* off = nuniq + nfb_max;
    auto* t_uniq = off + nfb_max;
    auto* t_count = t_uniq + n;
    auto* t_first = t_count + n;
    {
        ScopedKernelTimer tm(ctx, "k_distinct_rows");
        hipLaunchKernelGGL(k_distinct_rows<false>, dim3(ntiles), dim3(256), 0, st, d_recs, n, per_tile, l1, hist, tot, base, geom, e_a, d_n);
        hipLaunchKernelGGL(bdgpart::k_part_colscan, dim3(nb1), dim3(256), 0, st, hist, ntiles, nb1, tot);
        hipLaunchKernelGGL(bdgpart::k_part_bases, dim3(1), dim3(1024), 0, st, tot, nb1, DS_CAP, l2_max, (unsigned long long)n, base, geom);
        hipLaunchKernelGGL(k_distinct_rows<true>, dim3(ntiles), dim3(256), 0, st, d_recs, n, per_tile, l1, hist, tot, base, geom, e_a, d_n);
        if (two_levels) hipLaunchKernelGGL(bdgpart::k_part_split<unsigned long long>, dim3(nb1), dim3(1024), 0, st, e_a, e_b, base, geom, nb1, 64u - l1, fstart);
    }
    {
        ScopedKernelTimer tm(ctx, "k_distinct_buckets");
        const uint32_t grid = (uint32_t)std::min<size_t>(nfb_max, (size_t)cus * 6u);
        hipLaunchKernelGGL(k_distinct_buckets, dim3(grid), dim3(DS_THREADS), 0, st, two_levels ? e_b : e_a, two_levels ? fstart : nullptr, base, geom, l1,
                           t_uniq, t_count, t_first, nuniq);
    }
    {
        ScopedKernelTimer tm(ctx, "k_distinct_finish");
        hipLaunchKernelGGL(k_distinct_offsets, dim3(1), dim3(1024), 0, st, nuniq, geom, l1, two_levels ? 1 : 0, off, d_n);
        const uint32_t grid = (uint32_t)std::min<size_t>((nfb_max + 3) / 4, (size_t)cus * 8u);
        hipLaunchKernelGGL(k_distinct_compact, dim3(grid), dim3(256), 0, st, two_levels ? fstart : nullptr, base, geom, l1, nuniq, off,
                           t_uniq, t_count, t_first, d_uniq, d_count, d_first);
    }
    BDG_HIP_TRY(ctx, hipGetLastError());
    return BDG_OK;
}

// ---------------------------------------------------------------------------
// The two clustering levels of BarcodeGraph.cluster (barcode_graph.py:279-301) over the edge array that is already on the
// device.  The reference walks breadth-first from every centre, two levels deep, and marks a barcode reached by two
// different centres on one level as nobody's - whatever the visiting order (the order-independence
// argument) - so a level is: every edge (u, v), u expanding on this level and v still unclustered, offers owner[u] to v;
// v takes it if every offer on this level names the same centre.  "The same" = minimum equals maximum: two atomics per
// offer, no sort, no set.
// ---------------------------------------------------------------------------
namespace {

constexpr int32_t OWN_NONE = -2, OWN_CONFLICT = -1;

__global__ __launch_bounds__(256)
void k_cluster_offer(const uint32_t* __restrict__ ea, const uint32_t* __restrict__ eb, uint64_t m, int level,
                     const int32_t* __restrict__ owner, const uint8_t* __restrict__ reached,
                     int32_t* __restrict__ lo, int32_t* __restrict__ hi)
{
    const uint64_t e = (uint64_t)blockIdx.x * 256ull + threadIdx.x;
    if (e >= m) return;
    const uint32_t a = ea[e], b = eb[e];
    const int32_t oa = owner[a], ob = owner[b];
    // level 1: the centres expand (owner[c] == c); level 2: the barcodes level 1 gave to exactly one centre
    const bool xa = level == 1 ? oa == (int32_t)a : reached[a] != 0;
    const bool xb = level == 1 ? ob == (int32_t)b : reached[b] != 0;
    if (xa && ob == OWN_NONE) { atomicMin(&lo[b], oa); atomicMax(&hi[b], oa); }
    if (xb && oa == OWN_NONE) { atomicMin(&lo[a], ob); atomicMax(&hi[a], ob); }
}

__global__ __launch_bounds__(256)
void k_cluster_apply(uint32_t nu, int32_t* __restrict__ owner, uint8_t* __restrict__ reached,
                     int32_t* __restrict__ lo, int32_t* __restrict__ hi)
{
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= nu) return;
    const int32_t l = lo[v], h = hi[v];
    uint8_t r = 0;
    if (h >= 0) { owner[v] = l == h ? l : OWN_CONFLICT; r = l == h ? 1 : 0; }
    reached[v] = r;
    lo[v] = 0x7FFFFFFF; hi[v] = -1;
}

__global__ __launch_bounds__(256)
void k_cluster_init(uint32_t nu, uint8_t* __restrict__ reached, int32_t* __restrict__ lo, int32_t* __restrict__ hi)
{
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= nu) return;
    reached[v] = 0; lo[v] = 0x7FFFFFFF; hi[v] = -1;
}

// per read: position of its barcode in the distinct array -> what that barcode was corrected to (assign_by_cluster +
// output_file, barcode_graph.py:322-329,388-410): rank and "has one"
__global__ __launch_bounds__(256)
void k_assign_reads(const bdg_extract_rec* __restrict__ recs, uint64_t n, const uint32_t* __restrict__ uniq, uint32_t nu,
                    const uint32_t* __restrict__ assigned, const uint8_t* __restrict__ has,
                    uint32_t* __restrict__ out_rank, uint8_t* __restrict__ out_has)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const bdg_extract_rec r = recs[i];
    uint32_t rank = 0; uint8_t ok = 0;
    if (r.valid && (r.flags & BDG_FLAG_RANK_OK)) {
        uint32_t lo = 0, hi = nu;
        while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (uniq[mid] < r.bc_rank) lo = mid + 1; else hi = mid; }
        if (lo < nu && uniq[lo] == r.bc_rank && has[lo]) { rank = assigned[lo]; ok = 1; }
    }
    out_rank[i] = rank; out_has[i] = ok;
}

}  // namespace

int bdg_cluster_launch(bdg_ctx* ctx, const uint32_t* d_ea, const uint32_t* d_eb, uint64_t m, uint32_t nu, int32_t* d_owner)
{
    if (nu == 0) return BDG_OK;
    int rc;
    if ((rc = bdg_reserve(ctx, ctx->g_tmp0, (size_t)nu * 9 + 256))) return rc;
    auto* lo = static_cast<int32_t*>(ctx->g_tmp0.p);
    auto* hi = lo + nu;
    auto* reached = reinterpret_cast<uint8_t*>(hi + nu);
    hipStream_t st = ctx->stream;
    ScopedKernelTimer tm(ctx, "k_cluster_levels");
    hipLaunchKernelGGL(k_cluster_init, dim3((nu + 255) / 256), dim3(256), 0, st, nu, reached, lo, hi);
    for (int level = 1; level <= 2; ++level) {
        if (m) hipLaunchKernelGGL(k_cluster_offer, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d_ea, d_eb, m, level, d_owner, reached, lo, hi);
        hipLaunchKernelGGL(k_cluster_apply, dim3((nu + 255) / 256), dim3(256), 0, st, nu, d_owner, reached, lo, hi);
    }
    BDG_HIP_TRY(ctx, hipGetLastError());
    return BDG_OK;
}

// how many of the nu distinct barcodes have an edge or are listed in `extra` (the centres that were observed): what the
// reference's `len(self.edges)` counts (badger.py:131-132), without the edge positions leaving the device
namespace {
__global__ __launch_bounds__(256)
void k_touch(const uint32_t* __restrict__ ea, const uint32_t* __restrict__ eb, uint64_t m, const uint32_t* __restrict__ extra, uint32_t n_extra,
             uint32_t nu, uint8_t* __restrict__ flags)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256ull + threadIdx.x;
    if (i < m) { const uint32_t a = ea[i], b = eb[i]; if (a < nu) flags[a] = 1; if (b < nu) flags[b] = 1; }
    if (i < n_extra) { const uint32_t c = extra[i]; if (c < nu) flags[c] = 1; }
}
__global__ __launch_bounds__(256)
void k_count_flags(const uint8_t* __restrict__ flags, uint32_t nu, unsigned long long* __restrict__ total)
{
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256ull + threadIdx.x; i < nu; i += (uint64_t)gridDim.x * 256ull) c += flags[i] ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, (unsigned long long)c);
}
}  // namespace

int bdg_touched_count_launch(bdg_ctx* ctx, const uint32_t* d_ea, const uint32_t* d_eb, uint64_t m, uint32_t nu,
                             const uint32_t* d_extra, uint32_t n_extra, uint64_t* count)
{
    *count = 0;
    if (nu == 0) return BDG_OK;
    int rc;
    if ((rc = bdg_reserve(ctx, ctx->g_tmp1, (size_t)nu + 64))) return rc;
    auto* total = static_cast<unsigned long long*>(ctx->g_tmp1.p);
    auto* flags = reinterpret_cast<uint8_t*>(total + 1);
    hipStream_t st = ctx->stream;
    BDG_HIP_TRY(ctx, hipMemsetAsync(ctx->g_tmp1.p, 0, (size_t)nu + 8, st));
    const uint64_t work = m > n_extra ? m : n_extra;
    {
        ScopedKernelTimer tm(ctx, "k_touch");
        if (work) hipLaunchKernelGGL(k_touch, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, d_ea, d_eb, m, d_extra, n_extra, nu, flags);
        unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)nu + 255) / 256, 4096);
        hipLaunchKernelGGL(k_count_flags, dim3(grid), dim3(256), 0, st, flags, nu, total);
    }
    BDG_HIP_TRY(ctx, hipGetLastError());
    unsigned long long h = 0;
    BDG_HIP_TRY(ctx, hipMemcpyAsync(&h, total, 8, hipMemcpyDeviceToHost, st));
    BDG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *count = h;
    return BDG_OK;
}

int bdg_assign_reads_launch(bdg_ctx* ctx, const bdg_extract_rec* d_recs, uint64_t n, const uint32_t* d_uniq, uint32_t nu,
                            const uint32_t* d_assigned, const uint8_t* d_has, uint32_t* d_out_rank, uint8_t* d_out_has)
{
    if (n == 0) return BDG_OK;
    ScopedKernelTimer tm(ctx, "k_assign_reads");
    hipLaunchKernelGGL(k_assign_reads, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_recs, n, d_uniq, nu, d_assigned, d_has, d_out_rank, d_out_has);
    BDG_HIP_TRY(ctx, hipGetLastError());
    return BDG_OK;
}
