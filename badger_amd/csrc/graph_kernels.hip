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
// that can have dmin <= 1 with it and looks them up in the sorted array (binary search);
// see graph_probe_candidates() below.
#include "bdg_common.hpp"

#include <algorithm>

namespace {

constexpr int GT = 2048;     // column tile

__device__ __forceinline__ uint32_t letter_sig(uint32_t r)
{
    // byte k = number of bases with rank code k
    const uint32_t lo = r & 0x55555555u, hi = (r >> 1) & 0x55555555u;
    const uint32_t c3 = __popc(lo & hi), c1 = __popc(lo & ~hi), c2 = __popc(hi & ~lo);
    const uint32_t c0 = 16u - c1 - c2 - c3;
    return c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
}

// One Myers pass, pattern a (rows), text b (columns) -> min of D[16][16], D[15][16], D[16][15].
__device__ __forceinline__ uint32_t dmin3(uint32_t a, uint32_t b)
{
    uint32_t peq[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t c = (a >> (2 * i)) & 3u;
        peq[0] |= (c == 0u ? 1u : 0u) << i; peq[1] |= (c == 1u ? 1u : 0u) << i;
        peq[2] |= (c == 2u ? 1u : 0u) << i; peq[3] |= (c == 3u ? 1u : 0u) << i;
    }
    uint32_t pv = 0xFFFFu, mv = 0u, score = 16u, score15 = 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t c = (b >> (2 * j)) & 3u;
        const uint32_t eq = (c & 2u) ? ((c & 1u) ? peq[3] : peq[2]) : ((c & 1u) ? peq[1] : peq[0]);
        const uint32_t xv = eq | mv;
        const uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
        uint32_t ph = mv | ~(xh | pv);
        uint32_t mh = pv & xh;
        score += (ph >> 15) & 1u;
        score -= (mh >> 15) & 1u;
        ph = (ph << 1) | 1u;
        mh = mh << 1;
        pv = mh | ~(xv | ph);
        mv = ph & xv;
        if (j == 14) score15 = score;              // D[16][15] = ed(a, b[:-1])
    }
    // D[15][16] = D[16][16] - (vertical delta of the last row in the last column)
    const uint32_t d1516 = score - ((pv >> 15) & 1u) + ((mv >> 15) & 1u);   // ed(a[:-1], b)
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
constexpr uint32_t ECAP = 256;                   // staged edges per wave

struct EdgeStage { uint32_t a[ECAP], b[ECAP]; uint8_t d[ECAP]; };

__device__ __forceinline__ void edge_copy_out(const EdgeStage& st, uint32_t n, unsigned long long base, int lane,
                                              bdg_edge* __restrict__ out, uint64_t cap)
{
    for (uint32_t i = (uint32_t)lane; i < n; i += 64u) {
        const unsigned long long k = base + i;
        if (k < cap) { out[k].a = st.a[i]; out[k].b = st.b[i]; out[k].dist = st.d[i]; }
    }
}
// wave-wide: lanes with `want` append their edge; a full stage is written out with the wave's own reservation
__device__ __forceinline__ void edge_push(bool want, uint32_t a, uint32_t b, uint32_t d, EdgeStage& st, uint32_t& n, int lane,
                                          bdg_edge* __restrict__ out, uint64_t cap, unsigned long long* n_edges)
{
    const unsigned long long m = __ballot(want);
    if (!m) return;
    const uint32_t cnt = (uint32_t)__popcll(m);
    if (n + cnt > ECAP) {
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
// block-wide, every thread: one reservation for the four waves' stages
__device__ __forceinline__ void edge_finish(EdgeStage* stages, uint32_t n, uint32_t* s_cnt, unsigned long long* s_base,
                                            bdg_edge* __restrict__ out, uint64_t cap, unsigned long long* n_edges)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wv] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
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

__device__ __forceinline__ bool sorted_find(const uint32_t* __restrict__ ranks, uint32_t n, uint32_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t v = ranks[mid];
        if (v < key) lo = mid + 1; else hi = mid;
    }
    return lo < n && ranks[lo] == key;
}

// membership bitmap over the top `32 - shift` bits of the ranks: most candidates die here on one L2 hit instead of a binary search
__global__ __launch_bounds__(256)
void k_graph_bitmap(const uint32_t* __restrict__ ranks, uint32_t n, int shift, uint32_t* __restrict__ bitmap)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = ranks[i] >> shift;
    atomicOr(&bitmap[b >> 5], 1u << (b & 31u));
}

__global__ __launch_bounds__(256)
void k_graph_probe(const uint32_t* __restrict__ ranks, uint32_t n, uint32_t row_begin, uint32_t row_end, int32_t T,
                   const uint32_t* __restrict__ bitmap, int shift,
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
    for (int t = sub; t < NPROBE; t += 4) {                 // same trip count in every lane
        const uint32_t b = graph_probe_candidate(a, t);
        const uint32_t bb = b >> shift;
        bool edge = on && b > a && ((bitmap[bb >> 5] >> (bb & 31u)) & 1u) != 0 && sorted_find(ranks, n, b);
        if (edge) edge = graph_probe_first(a, b, t);         // de-duplicate: only the lowest slot producing b emits
        uint32_t d = 0;
        if (edge) { d = dmin3(a, b); edge = d <= 1u && (int32_t)qgram_S(a, b) >= T; }
        edge_push(edge, a, b, d, s_edges[wv], ne, lane, out, cap, n_edges);
    }
    edge_finish(s_edges, ne, s_ecnt, &s_ebase, out, cap, n_edges);
}

}  // namespace

// ---------------------------------------------------------------------------
int bdg_graph_launch(bdg_ctx* ctx, const uint32_t* d_ranks, uint32_t n, uint32_t row_begin, uint32_t row_end,
                     uint32_t thr, int32_t qgram_T, bdg_edge* d_out, uint64_t cap, uint64_t* d_n_edges)
{
    hipStream_t st = ctx->stream;
    BDG_HIP_TRY(ctx, hipMemsetAsync(d_n_edges, 0, 8, st));
    if (n < 2 || row_begin >= row_end) return BDG_OK;
    if (thr > 16) return bdg_fail(ctx, BDG_E_ARG, "thr must be <= 16");
    if (qgram_T < 1) return bdg_fail(ctx, BDG_E_ARG, "qgram_T must be >= 1 (index.py:22-24 never yields less)");
    const bool probe = ctx->graph_algo == 2 || (ctx->graph_algo == 0 && thr == 1);
    if (ctx->graph_algo == 2 && thr != 1) return bdg_fail(ctx, BDG_E_ARG, "probe path needs thr == 1");
    int rc;
    if (probe) {
        int bbits = 16;
        while (bbits < 27 && (1u << (bbits - 4)) < n) ++bbits;           // ~16 bits per barcode
        const size_t bm_bytes = (size_t(1) << bbits) / 8;
        if ((rc = bdg_reserve(ctx, ctx->g_sig, bm_bytes))) return rc;       // (the scan path's signature buffer is free here)
        auto* bitmap = static_cast<uint32_t*>(ctx->g_sig.p);
        BDG_HIP_TRY(ctx, hipMemsetAsync(bitmap, 0, bm_bytes, st));
        {
            ScopedKernelTimer tm(ctx, "k_graph_bitmap");
            hipLaunchKernelGGL(k_graph_bitmap, dim3((n + 255) / 256), dim3(256), 0, st, d_ranks, n, 32 - bbits, bitmap);
        }
        ScopedKernelTimer tm(ctx, "k_graph_probe");
        const uint64_t threads = 4ull * (row_end - row_begin);
        hipLaunchKernelGGL(k_graph_probe, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, st, d_ranks, n, row_begin, row_end, qgram_T,
                           bitmap, 32 - bbits, d_out, cap, reinterpret_cast<unsigned long long*>(d_n_edges));
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
