// Distinct-barcode counting on the device (reference BarcodeGraph.index_bc_single_thread,
// barcode_graph.py:192-204): from the extraction records of a batch, the distinct 16-base
// barcodes, how often each occurs and where it occurs first (the reference's `counts` dict is in
// first-occurrence order, which get_cluster_centers depends on, barcode_graph.py:253-255).
// Stable LSD radix sort of (33-bit key = unusable << 32 | rank, value = read index) + run-length
// encode, both from hipCUB; the first element of each run is its first occurrence because the
// sort is stable.
#include "bdg_common.hpp"

#include <hipcub/hipcub.hpp>

namespace {

__global__ __launch_bounds__(256)
void k_distinct_keys(const bdg_extract_rec* __restrict__ recs, uint32_t n,
                     unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals,
                     uint32_t* __restrict__ out_n /* [0] n_uniq (later), [1] barcodes of 16 bases holding a non-ACGT base */)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const bdg_extract_rec r = recs[i];
    const bool ok = r.valid && (r.flags & BDG_FLAG_RANK_OK);
    keys[i] = ok ? (unsigned long long)r.bc_rank : (1ull << 32);
    vals[i] = i;
    if (r.valid && !ok && (r.flags & BDG_FLAG_BC16)) atomicAdd(&out_n[1], 1u);
}

__global__ __launch_bounds__(256)
void k_distinct_finish(const unsigned long long* __restrict__ ukeys, const uint32_t* __restrict__ ucounts,
                       const uint32_t* __restrict__ uoffsets, const uint32_t* __restrict__ sorted_vals,
                       const uint32_t* __restrict__ n_runs,
                       uint32_t* __restrict__ uniq, uint32_t* __restrict__ count, uint32_t* __restrict__ first,
                       uint32_t* __restrict__ out_n)
{
    const uint32_t nr = *n_runs;
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j == 0) out_n[0] = (nr && ukeys[nr - 1] >= (1ull << 32)) ? nr - 1 : nr;     // the unusable run sorts last
    if (j >= nr || ukeys[j] >= (1ull << 32)) return;
    uniq[j] = (uint32_t)ukeys[j];
    count[j] = ucounts[j];
    first[j] = sorted_vals[uoffsets[j]];
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
    size_t t_sort = 0, t_rle = 0, t_scan = 0;
    unsigned long long* kp = nullptr; uint32_t* vp = nullptr;
    BDG_HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, t_sort, kp, kp, vp, vp, (int)n, 0, 33, st));
    BDG_HIP_TRY(ctx, hipcub::DeviceRunLengthEncode::Encode(nullptr, t_rle, kp, kp, vp, vp, (int)n, st));
    BDG_HIP_TRY(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, t_scan, vp, vp, (int)n, st));
    const size_t t_max = std::max(t_sort, std::max(t_rle, t_scan));
    // workspace: keys in/out (8n each), vals in/out (4n each), unique keys (8n), counts (4n), offsets (4n), n_runs, temp
    const size_t need = 8ull * n * 3 + 4ull * n * 4 + 256 + t_max;
    int rc;
    if ((rc = bdg_reserve(ctx, ctx->g_tmp1, need))) return rc;
    char* base = static_cast<char*>(ctx->g_tmp1.p);
    auto* k_in = reinterpret_cast<unsigned long long*>(base);
    auto* k_out = k_in + n;
    auto* u_keys = k_out + n;
    auto* v_in = reinterpret_cast<uint32_t*>(u_keys + n);
    auto* v_out = v_in + n;
    auto* u_cnt = v_out + n;
    auto* u_off = u_cnt + n;
    auto* n_runs = u_off + n;
    void* temp = reinterpret_cast<char*>(n_runs) + 256;
    {
        ScopedKernelTimer tm(ctx, "k_distinct_keys");
        hipLaunchKernelGGL(k_distinct_keys, dim3((n + 255) / 256), dim3(256), 0, st, d_recs, n, k_in, v_in, d_n);
    }
    {
        ScopedKernelTimer tm(ctx, "hipcub_sort_rle_scan");
        size_t t = t_max;
        BDG_HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(temp, t, k_in, k_out, v_in, v_out, (int)n, 0, 33, st));
        t = t_max;
        BDG_HIP_TRY(ctx, hipcub::DeviceRunLengthEncode::Encode(temp, t, k_out, u_keys, u_cnt, n_runs, (int)n, st));
        t = t_max;
        BDG_HIP_TRY(ctx, hipcub::DeviceScan::ExclusiveSum(temp, t, u_cnt, u_off, (int)n, st));
    }
    {
        ScopedKernelTimer tm(ctx, "k_distinct_finish");
        hipLaunchKernelGGL(k_distinct_finish, dim3((n + 255) / 256), dim3(256), 0, st, u_keys, u_cnt, u_off, v_out, n_runs,
                           d_uniq, d_count, d_first, d_n);
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
