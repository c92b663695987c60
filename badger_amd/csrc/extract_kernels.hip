// K1: per-read barcode extraction on gfx950.
//
// Replaces TenXBarcodeExtractor.find_barcode_umi (reference barcode_callers.py:165-229)
// and everything under it: find_polyt_start (barcode_extraction/common.py:10-31),
// reverese_complement (:34-39), KmerIndexer.get_occurrences over [R1]
// (kmer_indexer.py:49-75), detect_exact_positions (:85-114) and the SSW local
// alignment behind align_pattern_ssw (:42-51).
//
// Launches per batch, no host round trip in between:
//   k_scan_reads     persistent waves take tasks of 16 reads and stream them as 16-byte vectors, 63 per step, straight from
//                    the ASCII buffer (through LDS-DMA, two steps ahead): polyT start of both strands (T-windows of the
//                    reverse complement are A-windows of the read) and every R1 6-mer hit of both strands (7-mer probe
//                    table in LDS).  The hits of a vector -
//                    or of two neighbouring vectors cut through one adapter copy - form one CLUSTER
//                    {read, (first hit << 1) | strand, offset mask}; queue A takes clusters whose first hit lies left of
//                    polyT (relaxed search applies), queue B the other hits one by one.
//   k_sw_clusters    queue A, two clusters per lane (packed 16-bit halves): ONE 22-row Smith-Waterman pass over the union
//                    of the cluster's windows (all start at the first hit's window start, so the first hit's strict and
//                    relaxed windows are column prefixes of the union: their results are snapshots of the running key).
//                    If the union cannot beat the first hit, no later hit of the cluster can replace it as "first
//                    strictly best" (common.py:102-103) and they are skipped; otherwise they are re-queued (queue C).
//   k_strict_filter  queue B only matters if an alignment reaches score 17, which implies semi-global edit distance <= 5:
//                    Myers' 22-bit search per hit, after dropping read-strands the relaxed search has already decided;
//                    survivors join queue C.
//   k_sw_singles     the same alignment kernel over queue C: re-queued hits and filter survivors, one by one.
//   k_finalize_reads one lane per read: delta checks, reverse pass for strict hits, polyT re-search, barcode/UMI
//                    slicing, strand choice, 32-byte record.
// Every queue and every hot counter exists NSH times (see "Counters" below).
//
// Integer-only; bit-exact to oracle/badger_oracle.c.
#include "bdg_common.hpp"

namespace {

constexpr int R1_LEN = 22;
constexpr char R1[R1_LEN + 1] = "CTACACGACGCTCTTCCGATCT";   // barcode_callers.py:154
constexpr int KMER = 6;
constexpr int BC_LEN = 16;

// internal base code = (ascii >> 1) & 3 : A0 C1 T2 G3 ; complement = code ^ 2
constexpr uint32_t icode(char c) { return (uint32_t(c) >> 1) & 3u; }

constexpr int KEY_SHIFT = 11;                 // score | (63-col) << 5 | (31-row)
constexpr int32_t ONE = 1 << KEY_SHIFT;

// Counters.  One returning atomic on a single address completes every ~11 ns on this part, whoever issues it, so every hot
// counter exists NSH times (blocks use shard blockIdx % NSH) and each copy owns a 128-byte line.
constexpr int NSH = 8;
enum { K_TASK = 0,      // next task of the shard (k_scan_reads)
       K_NAB = 1,       // queue A count | queue B count << 32
       K_NC = 2, K_ND = 3,
       K_STAT = 4,      // S_* words below
       K_LINES = 5 };
enum { S_BADREAD = 0 /* max of ~read index, 0 = none */, S_NWINDOWS = 1, S_NKEPT = 2, S_NHITS = 3, S_NSKIPPED = 4, S_NBHITS = 5, S_N = 6 };
constexpr int SH_WORDS = K_LINES * 16;                       // 64-bit words per shard
constexpr size_t COUNTER_BYTES = (size_t)NSH * SH_WORDS * 8;

__device__ __forceinline__ unsigned long long* ctr(unsigned long long* c, uint32_t shard, int kind)
{
    return c + shard * SH_WORDS + kind * 16;
}

constexpr uint32_t HOLE_R = 0xFFFFFFFFu;   // unused queue slot

// queue entry: {read, (pos << 1) | strand, offset mask of the hits (bit 0 = first hit), unused}
typedef uint4 QEnt;

// A queue is NSH segments of `seg` entries, one per counter shard.  Consumers walk a virtual index g: entry g / NSH of
// segment g % NSH, a hole where that segment is shorter than the longest one.
__device__ __forceinline__ uint64_t queue_counts(const unsigned long long* counters, int kind, int half, uint64_t seg,
                                                 uint32_t* s_cnt /* [NSH], shared */)
{
    if (threadIdx.x < NSH) {
        unsigned long long c = *ctr(const_cast<unsigned long long*>(counters), threadIdx.x, kind);
        if (kind == K_NAB) c = half ? c >> 32 : c & 0xFFFFFFFFull;
        s_cnt[threadIdx.x] = (uint32_t)(c < seg ? c : seg);
    }
    __syncthreads();
    uint32_t mx = 0;
#pragma unroll
    for (int k = 0; k < NSH; ++k) mx = s_cnt[k] > mx ? s_cnt[k] : mx;
    return (uint64_t)mx * NSH;
}
__device__ __forceinline__ QEnt queue_fetch(const QEnt* __restrict__ q, uint64_t seg, const uint32_t* s_cnt, uint64_t g, uint64_t nq)
{
    QEnt e = make_uint4(0xFFFFFFFFu, 0, 0, 0);
    const uint32_t sh = (uint32_t)g & (NSH - 1);
    const uint64_t idx = g / NSH;
    if (g < nq && idx < s_cnt[sh]) e = q[sh * seg + idx];
    return e;
}
// per-wave LDS staging of {read, (pos << 1) | strand} single-hit entries: one reservation per flush
__device__ __forceinline__ void stage_flush(uint2* buf, uint32_t& n, int lane, QEnt* __restrict__ q, uint64_t seg, uint32_t shard,
                                            unsigned long long* cnt)
{
    if (n == 0) return;
    unsigned long long gb = 0;
    if (lane == 0) gb = atomicAdd(cnt, (unsigned long long)n);
    gb = __shfl(gb, 0);
    for (uint32_t i = (uint32_t)lane; i < n; i += 64u) {
        const unsigned long long idx = gb + i;
        if (idx < seg) { const uint2 e = buf[i]; q[shard * seg + idx] = make_uint4(e.x, e.y, 1u, 0u); }
    }
    n = 0;
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t range_mask16(int32_t lo, int32_t hi)
{
    const int l = min(max(lo, 0), 16), h = min(max(hi, 0), 16);
    return h > l ? (((1u << h) - 1u) & ~((1u << l) - 1u)) : 0u;
}

// ---------------------------------------------------------------------------
// k_scan_reads
// Autonomous waves, no block barrier after the probe table is in LDS.  A wave repeatedly takes a TASK
// of 16 consecutive reads from a sharded counter (the next task is taken, and its offsets are
// loaded, while the current one is processed).  The task's reads are laid end to end as a stream of
// 16-byte vectors; each step the wave's lanes take 63 consecutive vectors of that stream (lane 63
// repeats as lane 0 of the next step: it only feeds lane 62's look-ahead), so short tails of one
// read and the head of the next share a step.  The vectors reach the wave through LDS
// (global_load_lds_dwordx4, two steps ahead).  polyT candidates are parked and evaluated 64 at a time.
// Lane-vectors with hits are staged in a per-wave LDS region and become clusters at the end of the
// task, written to the global queues with ONE packed 64-bit reservation (A count | B count << 32).
// ---------------------------------------------------------------------------
constexpr int TASK_READS = 16;
constexpr uint32_t WENT = 256;             // per-wave staging (2 KiB)
constexpr int TASK_SHARDS = NSH;

struct TaskTab {                           // per wave, double buffered
    uint4    rd[TASK_READS];               // per read {A, -B, L, L - 5}: vector `slot` of the task lies at byte base + A + 16 * slot
                                           // (A is relative, modulo 2^32) and starts at read position 16 * slot - B
    uint32_t pend[TASK_READS];             // vectors of reads 0..j inclusive
    uint32_t base_lo, base_hi;             // byte offset of the task's first vector (0 for a task without vectors)
    uint32_t last_off;                     // offset of the task's last vector from there
};

__device__ __forceinline__ uint32_t find_read(const TaskTab& t, uint32_t slot)
{
    // number of reads whose vectors end at or before `slot` (pend is non-decreasing), at most TASK_READS - 1
    uint32_t j = t.pend[7] <= slot ? 8u : 0u;
    j += t.pend[j + 3] <= slot ? 4u : 0u;
    j += t.pend[j + 1] <= slot ? 2u : 0u;
    j += t.pend[j] <= slot ? 1u : 0u;
    return j;
}

// 16 bytes of the read stream, non-temporal: every byte is used once
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lanes_below(unsigned long long b)      // set bits of b in the lanes below this one
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
}
__device__ __forceinline__ uint4 ld_stream16(const uint8_t* p)
{
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t spread16(uint32_t x)      // bit k -> bit 2k
{
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    return (x | (x << 1)) & 0x55555555u;
}
__device__ __forceinline__ uint32_t gather_even(uint32_t x)   // bit 2k -> bit k
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0x0000FFFFu;
}

// Hit word of a vector as the probes deliver it ("Z form"): byte i = forward hits of positions 4i..4i+3 in the low nibble,
// reverse hits of the same positions in the high nibble.
__device__ __forceinline__ uint32_t z_low_nibbles(uint32_t z)      // low nibbles of the four bytes -> 16 bits
{
    uint32_t x = z & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0x0000FFFFu;
}
__device__ __forceinline__ uint32_t z_from_mask16(uint32_t m)      // 16 bits -> the same bits in both nibbles of each byte
{
    uint32_t x = m & 0xFFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    return x | (x << 4);
}

constexpr int SCAN_WAVES = 8;              // waves per block (they share the 16 KiB probe table)
constexpr int SCAN_BLOCKS_PER_CU = 2;      // (LDS)

__global__ __launch_bounds__(64 * SCAN_WAVES, SCAN_BLOCKS_PER_CU)
void k_scan_reads(const uint8_t* __restrict__ bases, uint64_t total_rounded,
                  const uint64_t* __restrict__ off, uint32_t n,
                  const uint8_t* __restrict__ kmer7,
                  int32_t* __restrict__ polyt,
                  QEnt* __restrict__ qa_all, QEnt* __restrict__ qb_all, uint64_t qcap /* per segment */,
                  unsigned long long* __restrict__ counters,
                  unsigned long long* __restrict__ keys)
{
    // 7-mer code (14 bits, base p in bits 0-1) -> bit0/1: bases p..p+5 / p+1..p+6 are an R1 6-mer, bit4/5: same for
    // the reverse complement of one.  One probe answers two positions.
    // All of the block's LDS is one object with the probe table first, i.e. at LDS address 0: a probe's address is its index.
    struct Shared {
        uint8_t  kmer[16384];
        uint8_t  buf[SCAN_WAVES][2][1024];   // per wave: the vectors of the next two steps (LDS-DMA)
        uint2    ent[SCAN_WAVES][WENT];      // per-wave staging of lane-vectors with hits (see emit)
        uint32_t ringr[SCAN_WAVES][32];      // read index of each ring slot
        TaskTab  tab[SCAN_WAVES][2];
        int32_t  pt[SCAN_WAVES][32][2];      // polyT of the reads of the wave's last tasks (ring)
        uint2    cand[SCAN_WAVES][192];      // pending T/A-rich candidates: .x = flags of the lane's 16 bases (T even bits, A odd), .y = of the next 16
        uint32_t candi[SCAN_WAVES][192];     // (p0 + 16) << 6 | ring << 1 | type
        uint32_t ptmin[SCAN_WAVES][32][2];   // per ring slot and strand: min of (window start << 5 | offset), 0xFFFFFFFF = none
        int32_t  ringL[SCAN_WAVES][32];      // read length of each ring slot
    };
    __shared__ __attribute__((aligned(16))) Shared sh;
    auto& s_kmer = sh.kmer; auto& s_buf = sh.buf; auto& s_ent = sh.ent; auto& s_ringr = sh.ringr; auto& s_tab = sh.tab;
    auto& s_pt = sh.pt; auto& s_cand = sh.cand; auto& s_candi = sh.candi; auto& s_ptmin = sh.ptmin; auto& s_ringL = sh.ringL;
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 1024 / (64 * SCAN_WAVES); ++k)
        reinterpret_cast<uint4*>(s_kmer)[tid + 64 * SCAN_WAVES * k] = reinterpret_cast<const uint4*>(kmer7)[tid + 64 * SCAN_WAVES * k];
    __syncthreads();

    const int lane = tid & 63, wv = tid >> 6;
    const uint32_t ntasks = (n + TASK_READS - 1) / TASK_READS;
    const uint32_t shard = blockIdx.x % TASK_SHARDS;
    unsigned int* const task_ctr = reinterpret_cast<unsigned int*>(ctr(counters, shard, K_TASK));
    unsigned long long* const nab = ctr(counters, shard, K_NAB);
    unsigned long long* const stat = ctr(counters, shard, K_STAT);
    QEnt* const qa = qa_all + shard * qcap;                 // this shard's segments
    QEnt* const qb = qb_all + shard * qcap;
    uint2* ent = s_ent[wv];
    uint32_t nent = 0, nhits_stat = 0;

    // take a task: k-th grab of this shard is task k * TASK_SHARDS + shard
    auto grab = [&]() -> uint32_t {
        uint32_t k = 0;
        if (lane == 0) k = atomicAdd(task_ctr, 1u);
        k = __builtin_amdgcn_readfirstlane(k);          // (wave-uniform for the compiler too: the task loop branches on scalars)
        const unsigned long long t = (unsigned long long)k * TASK_SHARDS + shard;
        return t < ntasks ? (uint32_t)t : 0xFFFFFFFFu;
    };
    // offsets of a task -> table (lanes 0..16 load them; neighbours, prefix sums and maxima go through DPP)
    auto load_tab = [&](uint32_t task, TaskTab& tb) {
        const uint64_t r0 = (uint64_t)task * TASK_READS;
        const uint32_t nr = (uint32_t)(n - r0 < TASK_READS ? n - r0 : TASK_READS);
        uint64_t o = 0;
        if ((uint32_t)lane <= nr) o = off[r0 + lane];
        // (lanes 0 .. 16 hold the task's offsets: neighbours, prefix sums and maxima over them go through DPP, not LDS)
        const uint64_t o1 = ((uint64_t)wave_shl1((uint32_t)(o >> 32)) << 32) | wave_shl1((uint32_t)o);
        int64_t L = (uint32_t)lane < nr ? (int64_t)(o1 - o) : 0;
        bool bad = false;
        // the vectors of a task are addressed relative to its first one with 32 bits: true for any valid offset array (16 reads
        // below 2^26 bases each); offsets that jump further are corrupt
        const uint32_t o0_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)o), o0_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(o >> 32));
        const uint64_t T0 = (((uint64_t)o0_hi << 32) | o0_lo) & ~15ull;          // (the builtin returns int: unsigned before it is widened)
        if (L < 0 || L >= (1ll << 26) || o + (uint64_t)L > total_rounded || o - T0 >= (1ull << 30)) { bad = (uint32_t)lane < nr; L = 0; }
        if (bad) atomicMax(&stat[S_BADREAD], ~(unsigned long long)(r0 + lane));        // corrupt offsets: report, never loop on them
        // every vector of a validated read lies inside [0, total_rounded): no bounds checks on the loads below
        uint32_t nv = L > 0 ? (uint32_t)(((o & 15ull) + (uint64_t)L + 15ull) >> 4) : 0u;
        uint32_t incl = nv;
        static_assert(TASK_READS == 16, "the sixteen reads of a task are one DPP row");
#define BDG_ROW_SHR(x, d) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x110 + (d) /* row_shr:d */, 0xF, 0xF, true))
        incl += BDG_ROW_SHR(incl, 1); incl += BDG_ROW_SHR(incl, 2); incl += BDG_ROW_SHR(incl, 4); incl += BDG_ROW_SHR(incl, 8);
        if (lane < TASK_READS) {
            const uint32_t excl = incl - nv;
            const uint32_t A = nv ? (uint32_t)((o & ~15ull) - T0) - 16u * excl : 0u;
            tb.rd[lane] = make_uint4(A, (uint32_t)(-(int32_t)(16u * excl + (uint32_t)(o & 15ull))), (uint32_t)L, (uint32_t)((int32_t)L - (KMER - 1)));
            tb.pend[lane] = incl;
            if (lane == TASK_READS - 1) { tb.base_lo = incl ? (uint32_t)T0 : 0u; tb.base_hi = incl ? (uint32_t)(T0 >> 32) : 0u; }
        }
        // offset of the last vector: lanes behind the task's end re-read it (a valid address that costs no select)
        uint32_t lastv = lane < TASK_READS && nv ? (uint32_t)((o & ~15ull) - T0) + 16u * (nv - 1u) : 0u;
        lastv = max(lastv, BDG_ROW_SHR(lastv, 1)); lastv = max(lastv, BDG_ROW_SHR(lastv, 2));
        lastv = max(lastv, BDG_ROW_SHR(lastv, 4)); lastv = max(lastv, BDG_ROW_SHR(lastv, 8));
#undef BDG_ROW_SHR
        if (lane == TASK_READS - 1) tb.last_off = lastv;             // (maximum over lanes 0 .. 15)
    };
    auto tab_base = [&](const TaskTab& tb) -> uint64_t {           // wave-uniform byte offset of the task's first vector
        const uint32_t lo = __builtin_amdgcn_readfirstlane(tb.base_lo), hi = __builtin_amdgcn_readfirstlane(tb.base_hi);
        return ((uint64_t)hi << 32) | lo;
    };
    // Staged item: one lane-vector with hits, {forward hit mask | reverse hit mask << 16, (p0 + 16) << 5 | ring slot}.
    // Items become queue entries 64 at a time once the polyT of their reads is known: per strand one cluster
    // {read, (first hit << 1) | strand, offset mask}; a cluster whose first hit lies left of polyT goes to queue A
    // (relaxed search applies), any other is split into single hits for the filter (queue B).  One packed reservation
    // (A count | B count << 32) per call.  force_a: polyT not known yet (staging overflow), everything to queue A.
    // An item is staged in Z form and unclipped; to_plain gives {forward hits | reverse hits << 16} of the 6-mer starts whose
    // six bases lie inside the read (the item's read length comes from the ring).
    auto to_plain = [&](uint2 it) -> uint32_t {
        const int32_t p0 = (int32_t)(it.y >> 5) - 16;
        const int32_t L = s_ringL[wv][it.y & 31u];
        const int32_t lo = max(-p0, 0), hv = min(max(L - (KMER - 1) - p0, 0), 16);
        const uint32_t valid = __builtin_amdgcn_ubfe(0xFFFFFFFFu << lo, 0, hv);
        return (z_low_nibbles(it.x) & valid) | ((z_low_nibbles(it.x >> 4) & valid) << 16);
    };
    auto emit = [&](bool from_lds, uint2 mine, bool mine_on, uint32_t n_items, bool force_a) __attribute__((always_inline)) {
        constexpr int MAXC = (int)(WENT / 64u);
        // Two items that are neighbouring vectors of one read and whose hits of a strand span at most 17 positions (one
        // adapter copy cut by the vector boundary) form ONE cluster: its union window still fits the 56 columns of
        // k_sw_clusters and one alignment serves both.  The item holding the cluster's first hit in strand order
        // absorbs the other; an item takes part in one merge only (cond(h,h+1) && !cond(h-1,h) - a missed merge
        // costs time, never correctness).
        auto span_ok = [](uint2 a, uint2 b, int sh) -> bool {
            const uint32_t ha = (a.x >> sh) & 0xFFFFu, hb = (b.x >> sh) & 0xFFFFu;
            const uint32_t M = ha | (hb << 16);
            return ha != 0 && hb != 0 && b.y - a.y == (16u << 5) && (31 - __builtin_clz(M)) - __builtin_ctz(M) <= 17;
        };
        // Pass 1, 64 items at a time: the lane's (up to) two clusters, kept in registers across the reservation:
        // {read, forward (position << 1, offset mask), reverse (position << 1 | 1, offset mask), af | ar << 1 | bf << 2 | br << 3}
        uint32_t c_r[MAXC], c_fy[MAXC], c_fz[MAXC], c_ry[MAXC], c_rz[MAXC], c_fl[MAXC];
        const uint32_t nchunk = from_lds ? (n_items + 63u) / 64u : 1u;
        uint32_t acc = 0;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            c_r[c] = c_fy[c] = c_fz[c] = c_ry[c] = c_rz[c] = c_fl[c] = 0u;
            if ((uint32_t)c < nchunk) {
                uint2 it = mine; bool on = mine_on;
                if (from_lds) { const uint32_t h = (uint32_t)c * 64u + (uint32_t)lane; on = h < n_items; it = on ? ent[h] : make_uint2(0u, 0u); }
                it.x = on ? to_plain(it) : 0u;
                nhits_stat += __popc(it.x);
                uint32_t hF = it.x & 0xFFFFu, hR = it.x >> 16;
                const uint32_t ring = it.y & 31u;
                const int32_t p0 = (int32_t)(it.y >> 5) - 16;
                int32_t p0r = p0;
                if (from_lds) {
                    // neighbours inside the 64-item chunk by DPP (a pair cut by the chunk boundary is simply not merged)
                    const uint2 nx = make_uint2(wave_shl1(it.x), wave_shl1(it.y));
                    const uint32_t cF = span_ok(it, nx, 0) ? 1u : 0u, cR = span_ok(it, nx, 16) ? 1u : 0u;     // cond(h, h+1)
                    const uint32_t cFp = wave_shr1(cF), cFp2 = wave_shr1(cFp), cRp = wave_shr1(cR), cRn = wave_shl1(cR);
                    const uint32_t xp = wave_shr1(it.x);
                    // forward strand: the earlier item leads
                    if (cFp && !cFp2) hF = 0;                                                // absorbed by h-1
                    else if (cF && !cFp) hF |= (nx.x & 0xFFFFu) << 16;                      // absorbs h+1
                    // reverse strand: the later item leads
                    if (cR && !cRn) hR = 0;                                                  // absorbed by h+1
                    else if (cRp && !cR) { hR = (xp >> 16) | (hR << 16); p0r = p0 - 16; }   // absorbs h-1
                }
                const int32_t L = s_ringL[wv][ring];
                const int32_t ptF = s_pt[wv][ring][0], ptR = s_pt[wv][ring][1];
                const int k0 = hF ? __builtin_ctz(hF) : 0;
                const int32_t posF = p0 + k0;
                const bool af = hF != 0 && (force_a || (ptF >= 0 && posF + KMER <= ptF + 1));
                const bool bf = hF != 0 && !af;
                const int k1 = hR ? 31 - __builtin_clz(hR) : 0;
                const int32_t posR = L - KMER - (p0r + k1);
                const bool ar = hR != 0 && (force_a || (ptR >= 0 && posR + KMER <= ptR + 1));
                const bool br = hR != 0 && !ar;
                c_r[c] = s_ringr[wv][ring];
                c_fy[c] = (uint32_t)posF << 1; c_fz[c] = hF >> k0;
                c_ry[c] = ((uint32_t)posR << 1) | 1u; c_rz[c] = __brev(hR) >> (31 - k1);
                c_fl[c] = (af ? 1u : 0u) | (ar ? 2u : 0u) | (bf ? 4u : 0u) | (br ? 8u : 0u);
                acc += (af ? 1u : 0u) + (ar ? 1u : 0u) + (((bf ? 1u : 0u) + (br ? 1u : 0u)) << 16);
            }
        }
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(acc), 63);
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(nab, (unsigned long long)(tot & 0xFFFFu) | ((unsigned long long)(tot >> 16) << 32));
        unsigned long long gA = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base), gB = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        // Pass 2: positions inside the reservation, entries written
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if ((uint32_t)c < nchunk) {
                const uint32_t fl = c_fl[c];
                const uint32_t mine_cnt = (fl & 1u) + ((fl >> 1) & 1u) + ((((fl >> 2) & 1u) + ((fl >> 3) & 1u)) << 16);
                const uint32_t incl = wave_incl_scan(mine_cnt);
                const uint32_t excl = incl - mine_cnt;
                unsigned long long ga = gA + (excl & 0xFFFFu), gb = gB + (excl >> 16);
                const QEnt ef = make_uint4(c_r[c], c_fy[c], c_fz[c], 0u), er = make_uint4(c_r[c], c_ry[c], c_rz[c], 0u);
                if (fl & 1u) { if (ga < qcap) qa[ga] = ef; ++ga; }
                if (fl & 2u) { if (ga < qcap) qa[ga] = er; }
                if (fl & 4u) { if (gb < qcap) qb[gb] = ef; ++gb; }
                if (fl & 8u) { if (gb < qcap) qb[gb] = er; }
                const uint32_t ctot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                gA += ctot & 0xFFFFu; gB += ctot >> 16;
            }
        }
    };
    // Only whole groups of 64 staged items are turned into clusters (every lane busy); the rest waits for the next task's
    // items - but never longer: the ring keeps a read's polyT for one more task only.
    uint32_t carried = 0;                   // staged items that belong to the previous task
    auto flush = [&](bool all) {
        uint32_t nf = all ? nent : (nent & ~63u);
        if (nf < carried) nf = nent;
        if (nf) {
            emit(true, make_uint2(0u, 0u), false, nf, false);
            const uint32_t rem = nent - nf;                       // < 64 (or 0)
            if (rem) {
                const uint2 t = (uint32_t)lane < rem ? ent[nf + lane] : make_uint2(0u, 0u);
                __builtin_amdgcn_wave_barrier();
                if ((uint32_t)lane < rem) ent[lane] = t;
                __builtin_amdgcn_wave_barrier();
            }
            nent = rem;
        }
        carried = nent;
    };

    // polyT: the forward strand wants the FIRST 16-window with >= 12 T, the reverse strand the LAST with >= 12 A
    // (= the first T-window of the reverse complement).  Both are order-independent reductions, so candidate lanes
    // only park their flags here; 64 parked lanes are evaluated together (16 shifted popcounts each) and the
    // result is reduced with one LDS atomic min on (window start << 5 | offset of the first TTT, common.py:31).
    uint32_t ncand = 0;
    auto eval_cands = [&](uint32_t first, uint32_t count) {
        if ((uint32_t)lane < count) {
            const uint2 c = s_cand[wv][first + lane];
            const uint32_t info = s_candi[wv][first + lane];
            const int32_t p0 = (int32_t)(info >> 6) - 16; const uint32_t ring = (info >> 1) & 31u, typ = info & 1u;
            uint32_t m = gather_even(c.x >> typ) | (gather_even(c.y >> typ) << 16);     // flags of the 32 bases from p0 on
            const int32_t L = s_ringL[wv][ring];
            // The reverse strand wants the LAST A-window, which is the first window of the same 31 bases read backwards
            // (base p0 + 30 - x at position x): window start k becomes 15 - k, the position of the window on the reverse
            // strand, L - 16 - (p0 + k), becomes (L - 31 - p0) + (15 - k), and the first 'TTT' behind the start is the same
            // search.  One code path serves both kinds.
            if (typ) m = __brev(m) >> 1;
            const int32_t P = typ ? L - 31 - p0 : p0;                   // strand position of the window with start 0
            // The 16 window counts at once, bit-sliced: after the level of width w, bit k of (s_i) holds bit i of the number
            // of flags among m[k .. k + w); a level adds the counts of two half windows (a ripple adder of bitwise operations:
            // sum = a ^ b ^ carry, carry = majority).  31 instructions for all 16 starts.
            auto xor3 = [](uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); };
            auto maj = [](uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); };
            const uint32_t a0 = m ^ (m >> 1), a1 = m & (m >> 1);                                         // width 2
            const uint32_t b0 = a0 >> 2, b1 = a1 >> 2, c0 = a0 & b0;
            const uint32_t s0 = a0 ^ b0, s1 = xor3(a1, b1, c0), s2 = maj(a1, b1, c0);                    // width 4: 0..4
            const uint32_t d0 = s0 >> 4, d1 = s1 >> 4, d2 = s2 >> 4, e0 = s0 & d0, e1 = maj(s1, d1, e0);
            const uint32_t t0 = s0 ^ d0, t1 = xor3(s1, d1, e0), t2 = xor3(s2, d2, e1), t3 = maj(s2, d2, e1);   // width 8: 0..8
            const uint32_t f0 = t0 >> 8, f1 = t1 >> 8, f2 = t2 >> 8, f3 = t3 >> 8;
            const uint32_t g0 = t0 & f0, g1 = maj(t1, f1, g0), g2 = maj(t2, f2, g1);
            const uint32_t u2 = xor3(t2, f2, g1), u3 = xor3(t3, f3, g2), u4 = maj(t3, f3, g2);           // width 16: 0..16 (bits 0, 1 not needed)
            // >= 12 (int(16 * 0.75), common.py:11) = 16, or 8 + 4 + ...
            uint32_t q = __builtin_amdgcn_bitop3_b32(u4, u3, u2, 0xF8) &                                 // u4 | (u3 & u2)
                         range_mask16(-P, L - 16 - P);                  // window starts 0 <= p < L-16 (common.py:17,28)
            if (q) {
                const int k = __builtin_ctz(q);
                const uint32_t tt = (m & (m >> 1) & (m >> 2)) >> k;     // 'TTT' starts, common.py:31
                atomicMin(&s_ptmin[wv][ring][typ], ((uint32_t)(P + k) << 5) | (tt ? (uint32_t)__builtin_ctz(tt) : 0u));
            }
        }
    };

    // The vectors of a step reach the wave through LDS: global_load_lds_dwordx4 writes lane l's 16 bytes to buffer + 16 l
    // without passing through registers, so the loads of the next TWO steps stay in flight while a step is worked on (a load
    // into registers would have to be copied when the steps rotate, and a copy waits for its load; one step of work does not
    // cover the memory latency at four waves per SIMD).  The compiler does not see these loads: the waits are placed here.
    // Loads complete in order, so "at most one younger load outstanding" means the older one has landed.
    const uint32_t buf_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&s_buf[wv][0][0];
    auto dma_issue = [&](uint32_t g, const uint8_t* gptr) {            // step g's vector of this lane -> buffer g & 1
        const uint32_t base = __builtin_amdgcn_readfirstlane(buf_lds + (g & 1u) * 1024u);
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"                       // (the buffer's previous contents have been read)
                     "s_mov_b32 m0, %0\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off nt" :: "s"(base), "v"(gptr) : "memory", "m0");
    };
    auto dma_wait = [&](bool younger_in_flight) {
        if (younger_in_flight) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const uint32_t kmer_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)&s_kmer[0];

    int cur = 0;
    uint32_t tseq = 0;                      // ring slot of read j of the current task: (tseq & 1) * TASK_READS + j
    uint32_t task = grab();
    if (task != 0xFFFFFFFFu) load_tab(task, s_tab[wv][cur]);
    uint32_t g = 0;                         // steps this wave has done
    bool have_v = false, have_n = false;    // the loads of step g / g + 1 have been issued
    const uint32_t lane_z = lane < 63 ? 0xFFFFFFFFu : 0u;      // lane 63 only feeds lane 62's look-ahead: it stages no hits
    const uint32_t pt_thr = lane < 63 ? 12u : 99u;             // ... and parks no polyT candidates
    while (task != 0xFFFFFFFFu) {
        const uint32_t next_task = grab();
        if (next_task != 0xFFFFFFFFu) load_tab(next_task, s_tab[wv][cur ^ 1]);
        __builtin_amdgcn_wave_barrier();
        const TaskTab& tb = s_tab[wv][cur];
        const TaskTab& nt = s_tab[wv][cur ^ 1];
        const uint64_t r0 = (uint64_t)task * TASK_READS;
        const uint32_t nr = (uint32_t)(n - r0 < TASK_READS ? n - r0 : TASK_READS);
        const uint32_t nslots = __builtin_amdgcn_readfirstlane(tb.pend[TASK_READS - 1]);
        const uint32_t niter = (nslots + 62u) / 63u;
        const uint32_t nnslots = next_task != 0xFFFFFFFFu ? __builtin_amdgcn_readfirstlane(nt.pend[TASK_READS - 1]) : 0u;
        const uint32_t nniter = (nnslots + 62u) / 63u;
        // Read of a slot = number of boundaries (pend[k], k < TASK_READS - 1) at or below it.  The lanes of a step hold 64
        // consecutive slots starting at `lo`, and steps are mapped in order: with kb = number of boundaries at or below lo only
        // the boundaries inside the 64 slots are compared (one per step on average).  Leaves kb = boundaries at or below lo + 63.
        uint32_t kb = 0;
        const uint32_t pend_lane = tb.pend[lane & (TASK_READS - 1)];          // boundary k in lane k: read back with v_readlane
        auto map_step = [&](uint32_t lo, uint32_t slot) -> uint32_t {
            uint32_t jv = kb;
            while (kb < TASK_READS - 1) {
                const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)pend_lane, (int)kb);
                if (b > lo + 63u) break;
                jv += b <= slot ? 1u : 0u;
                kb = __builtin_amdgcn_readfirstlane(kb + 1u);
            }
            return jv;
        };
        // address of this lane's vector in step s of this task (jj = read of the lane's slot) / of the next task; a lane
        // behind the task's end gets the task's last vector (a valid address that costs no select)
        const uint32_t last_off = __builtin_amdgcn_readfirstlane(tb.last_off), nlast_off = __builtin_amdgcn_readfirstlane(nt.last_off);
        const uint64_t tbase = tab_base(tb);
        const uint64_t nbase = next_task != 0xFFFFFFFFu ? tab_base(nt) : 0ull;
        auto addr_cur = [&](uint32_t s, uint32_t jj, uint32_t& negB) -> const uint8_t* {      // (also: -B of the lane's read)
            const uint32_t sl = s * 63u + (uint32_t)lane;
            const uint2 ab = *reinterpret_cast<const uint2*>(&tb.rd[jj]);
            negB = ab.y;
            return bases + tbase + min(ab.x + 16u * sl, last_off);
        };
        auto addr_next = [&](uint32_t s) -> const uint8_t* {
            const uint32_t sl = s * 63u + (uint32_t)lane;
            return bases + nbase + min(nt.rd[find_read(nt, sl)].x + 16u * sl, nlast_off);
        };
        const uint32_t ring0 = (tseq & 1u) * TASK_READS;
        if (lane < 2 * TASK_READS) s_ptmin[wv][ring0 + (lane >> 1)][lane & 1] = 0xFFFFFFFFu;
        if (lane < TASK_READS) { s_ringr[wv][ring0 + lane] = (uint32_t)(r0 + lane); s_ringL[wv][ring0 + lane] = (int32_t)tb.rd[lane].z; }
        __builtin_amdgcn_wave_barrier();
        uint32_t j = map_step(0u, (uint32_t)lane), jn = map_step(63u, 63u + (uint32_t)lane);
        uint32_t nb = 0, nbn = 0;               // -B of read j / jn (the vector's position in its read is 16 * slot - B)
        {
            const uint8_t* const a0 = addr_cur(0u, j, nb);
            const uint8_t* const a1 = addr_cur(1u, jn, nbn);
            if (niter) {                        // (start of the batch, or behind a task without vectors: nothing is in flight)
                if (!have_v) { dma_issue(g, a0); have_v = true; }
                if (!have_n) {
                    if (niter > 1u) { dma_issue(g + 1u, a1); have_n = true; }
                    else if (nniter) { dma_issue(g + 1u, addr_next(0u)); have_n = true; }
                }
            }
        }

        for (uint32_t it = 0; it < niter; ++it, ++g) {
            const uint32_t slot_lo = it * 63u;
            const uint32_t slot = slot_lo + (uint32_t)lane;
            dma_wait(have_n);
            __builtin_amdgcn_s_setprio(1);              // (steps before per-task work of other waves: measured, 1.5 %)
            const uint4 v = *reinterpret_cast<const uint4*>(&s_buf[wv][g & 1u][16 * lane]);

            // 2-bit codes.  byte & 6 is a perfect hash of "ACTG" (twice the code: 0, 2, 4, 6); v_perm maps it back to the expected
            // letter, v_dot4 packs four of them into twice a byte of codes.  A byte that is not its expected letter (N, a bad
            // base, the zero padding behind the last read) sends the whole wave through the exact per-byte path.
            const uint32_t words[4] = { v.x, v.y, v.z, v.w };
            uint32_t sel[4], diff = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                sel[d] = words[d] & 0x06060606u;
                diff = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_perm(0x00470054u, 0x00430041u, sel[d]), words[d], diff, 0xBE);   // (a ^ b) | c
            }
            asm volatile("" : "+v"(diff));                          // (one compare of the OR, not four compares)
            // where the step after next comes from
            uint32_t jnn = 0, nbnn = 0;
            const uint8_t* pf = bases;
            bool have_nn = false;
            if (it + 2u < niter) { jnn = map_step(slot_lo + 126u, slot + 126u); pf = addr_cur(it + 2u, jnn, nbnn); have_nn = true; }
            else if (it + 2u - niter < nniter) { pf = addr_next(it + 2u - niter); have_nn = true; }
            if (have_nn) dma_issue(g, pf);                          // the buffer just read takes it
            // Position of the vector in its read.  Nothing is clipped to the read here: a 6-mer start is checked against the
            // read when the item becomes a cluster (to_plain); a polyT window start is valid only if its 16 bases lie inside the
            // read (eval_cands), so flags of bytes outside it never count, and the first 'TTT' behind a valid start lies inside
            // its window (>= 12 T among 16 leave a run of three).  A lane behind the task's last vector sits at p0 >= L of the
            // last read: it has neither.
            const int32_t p0 = (int32_t)(slot << 4) + (int32_t)nb;
            const uint32_t where = ((uint32_t)(p0 + 16) << 5) | (ring0 + j);     // position and ring slot, as staged and parked
            uint32_t zm = lane_z;                                   // 6-mer starts that may hit, Z form
            uint32_t codes, bad = 0;
            uint32_t T_s, A_s;                                      // T = code 2, A = code 0: flags on the even bits
            const bool exact = __ballot(diff != 0) != 0;
            if (!exact) {
                uint32_t c[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) c[d] = __builtin_amdgcn_udot4(sel[d], 0x40100401u, 0u, false);      // 2 * (four codes)
                codes = ((((c[3] << 8) + c[2]) << 8) + c[1]) << 7 | (c[0] >> 1);      // (the doubled bytes do not overlap: bit 0 of each is 0)
                T_s = (codes >> 1) & ~codes & 0x55555555u;
                A_s = ~((codes >> 1) | codes) & 0x55555555u;
            } else {
                uint32_t fm = 0x55555555u;                                       // positions whose T / A flag counts
                const int32_t L = (int32_t)tb.rd[j].z;
                uint32_t nN = 0, nbad = 0;
                codes = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t b = (words[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                    const bool isN = b == 'N';
                    const bool ok = b == 'A' || b == 'C' || b == 'G' || b == 'T';
                    nN |= (isN ? 1u : 0u) << k;
                    nbad |= ((isN || ok) ? 0u : 1u) << k;
                    codes |= ((b >> 1) & 3u) << (2 * k);
                }
                const int32_t lo_t = max(-p0, 0);                                // read bases of this vector: [lo_t, hi)
                const int32_t hi = min(max(L - p0, 1), 16);
                const bool same_next = lane < 63 && p0 + 16 < L;                 // the next lane holds the next 16 bases of the same read
                const uint32_t rm16 = (0xFFFFu >> (16 - hi)) & (0xFFFFu << lo_t);
                bad = slot < nslots ? nbad & rm16 : 0u;                          // (a lane behind the task's end owns no base)
                nN = (nN | ~rm16) & 0xFFFFu;                                     // out-of-read behaves like N
                fm &= ~spread16(nN | nbad);                                      // neither T nor A
                uint32_t N1 = wave_shl1(nN);
                if (!same_next) N1 = 0xFFFFu;
                const uint32_t N32 = nN | (N1 << 16);
                uint32_t nvm = N32 | (N32 >> 1);
                nvm |= nvm >> 2;
                nvm |= N32 >> 4; nvm |= N32 >> 5;
                zm &= z_from_mask16(~nvm);                                       // 6-mers touching an N never match
                T_s = (codes >> 1) & ~codes & fm;
                A_s = ~((codes >> 1) | codes) & fm;
            }

            // look-ahead from the next lane (one DPP move each)
            const uint32_t TA = T_s | (A_s << 1);
            const uint32_t TA1 = wave_shl1(TA);
            const uint32_t codes1 = wave_shl1(codes);

            // R1 6-mer hits of both strands: 8 probes of the 7-mer table, two positions each.  (Issued by hand: the compiler
            // masks every sub-dword LDS load although ds_read_u8 zero-extends.)
            uint32_t e[8];
            {
                uint32_t ad[8];
                const uint32_t hi = __builtin_amdgcn_alignbit(codes1, codes, 20);      // codes of positions 10 .. 25
#pragma unroll
                for (int k = 0; k < 16; k += 2)
                    ad[k >> 1] = kmer_lds + (k <= 8 ? __builtin_amdgcn_ubfe(codes, 2 * k, 14) : __builtin_amdgcn_ubfe(hi, 2 * k - 20, 14));
                asm volatile("ds_read_u8 %0, %8\n\tds_read_u8 %1, %9\n\tds_read_u8 %2, %10\n\tds_read_u8 %3, %11\n\t"
                             "ds_read_u8 %4, %12\n\tds_read_u8 %5, %13\n\tds_read_u8 %6, %14\n\tds_read_u8 %7, %15"
                             : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]), "=&v"(e[4]), "=&v"(e[5]), "=&v"(e[6]), "=&v"(e[7])
                             : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]));
            }

            // polyT windows.  A 16-window with >= 12 T needs >= 12 T among the 31 bases a lane can see, which few lanes
            // have: those lanes park their flags; evaluation happens 64 lanes at a time (eval_cands).
            {
                // (lane 63 never parks: its vector is lane 0's of the next step.  A lane behind the task's end may: it sits at
                // p0 >= L, where eval_cands finds no valid window start.)
                const bool cT = __popc(T_s) + __popc(TA1 & 0x15555555u) >= pt_thr;
                const bool cA = __popc(A_s) + __popc(TA1 & 0x2AAAAAAAu) >= pt_thr;
                const unsigned long long bT = __builtin_amdgcn_ballot_w64(cT), bA = __builtin_amdgcn_ballot_w64(cA), bc = bT | bA;
                if (bc) {
                    // one entry per lane: its flags of both kinds, the kind to evaluate in bit 0 of the info word
                    const uint32_t info = where << 1;
                    if (cT || cA) { const uint32_t at = ncand + lanes_below(bc);
                                    s_cand[wv][at] = make_uint2(TA, TA1); s_candi[wv][at] = info | (cT ? 0u : 1u); }
                    ncand += (uint32_t)__popcll(bc);
                    const unsigned long long bb = bT & bA;                         // rich in both (AT repeats): a second entry
                    if (bb) {
                        if (cT && cA) { const uint32_t at = ncand + lanes_below(bb);
                                        s_cand[wv][at] = make_uint2(TA, TA1); s_candi[wv][at] = info | 1u; }
                        ncand += (uint32_t)__popcll(bb);
                    }
                }
            }
            if (exact && __ballot(bad != 0)) {
                if (bad != 0) atomicMax(&stat[S_BADREAD], ~(unsigned long long)(r0 + j));
            }
            // Lanes with hits stage {Z, position, ring slot}; they become queue entries at the next flush.  A table entry holds
            // the forward hits of its two positions in bits 0-1 and the reverse hits in bits 4-5: two probes make a byte of Z.
            {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]));
                const uint32_t b0 = e[0] | (e[1] << 2), b1 = e[2] | (e[3] << 2), b2 = e[4] | (e[5] << 2), b3 = e[6] | (e[7] << 2);
                const uint32_t z = ((b0 | (b1 << 8)) | ((b2 | (b3 << 8)) << 16)) & zm;
                const unsigned long long bal = __ballot(z != 0);
                if (bal) {
                    const uint32_t cnt = (uint32_t)__popcll(bal);
                    const uint2 item = make_uint2(z, where);
                    if (nent + cnt <= WENT) {
                        if (z) ent[nent + lanes_below(bal)] = item;
                        nent += cnt;
                    } else {
                        emit(false, item, z != 0, 0u, true);       // staging full (pathological reads): straight to queue A
                    }
                }
            }
            j = jn; jn = jnn; nb = nbn; nbn = nbnn;
            have_v = have_n; have_n = have_nn;
            __builtin_amdgcn_wave_barrier();
            while (ncand >= 64u) { ncand -= 64u; eval_cands(ncand, 64u); }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_wave_barrier();
        if (ncand) { eval_cands(0u, ncand); ncand = 0; }
        __builtin_amdgcn_wave_barrier();
        if (lane < 2 * TASK_READS) {
            const uint32_t pk = s_ptmin[wv][ring0 + (lane >> 1)][lane & 1];
            const int32_t pv = pk == 0xFFFFFFFFu ? -1 : (int32_t)((pk >> 5) + (pk & 31u));
            s_pt[wv][ring0 + (lane >> 1)][lane & 1] = pv;
            if ((uint32_t)lane < 2 * nr) {
                polyt[2 * r0 + lane] = pv;
                keys[2 * r0 + lane] = 0ull;                              // best relaxed / strict alignment of the read-strand:
                keys[2ull * n + 2 * r0 + lane] = 0ull;                   // none yet (k_sw_clusters runs after this kernel)
            }
        }
        __builtin_amdgcn_wave_barrier();
        ++tseq;
        flush(false);                           // the task's polyT is known: its staged items become clusters
        task = next_task;
        cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (no load of a step is left in flight: every issued step was done)
    flush(true);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nhits_stat += __shfl_xor(nhits_stat, d);
    if (lane == 0 && nhits_stat) atomicAdd(&stat[S_NHITS], (unsigned long long)nhits_stat);
}

// ---------------------------------------------------------------------------
// Window loader: 4*NW characters of strand `strand` starting at strand position x0,
// walking in direction dir, as raw forward bytes (char k = byte k of w[]).
// Characters outside the read are garbage; the caller masks them (k >= n).
// ---------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };

template <int NW>
__device__ __forceinline__ void load_block(const uint8_t* __restrict__ bases, uint64_t total_rounded,
                                           uint64_t rs, int64_t L, int strand, int64_t x0, int dir,
                                           uint32_t (&w)[NW])
{
    const int64_t f0 = strand ? (L - 1 - x0) : x0;
    const bool asc = (dir > 0) != (strand != 0);
    const int64_t a = (int64_t)rs + (asc ? f0 : f0 - (4 * NW - 1));
    const int64_t a_al = a & ~3ll;
    const uint32_t sh = (uint32_t)(a - a_al);
    // NW+1 dwords from a dword-aligned address: 16-byte loads (global_load_dwordx4 only needs dword alignment; one
    // wide load per lane costs the address unit far less than four narrow ones), per-dword guarded loads at the buffer edges
    constexpr int NQ = (NW + 1 + 3) / 4;
    uint32_t t[4 * NQ];
    if (a_al >= 0 && (uint64_t)a_al + 16ull * NQ <= total_rounded) {
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
            const U4 v4 = *reinterpret_cast<const U4*>(bases + a_al + 16 * qd);
            t[4 * qd] = v4.x; t[4 * qd + 1] = v4.y; t[4 * qd + 2] = v4.z; t[4 * qd + 3] = v4.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NW + 1; ++i) {
            const int64_t idx = a_al + 4 * i;
            t[i] = (idx >= 0 && (uint64_t)idx + 4 <= total_rounded)
                       ? *reinterpret_cast<const uint32_t*>(bases + idx) : 0u;
        }
    }
    uint32_t u[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) u[i] = __builtin_amdgcn_alignbyte(t[i + 1], t[i], sh);
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = asc ? u[i] : __builtin_bswap32(u[NW - 1 - i]);
}

template <int NW>
__device__ __forceinline__ bool block_has_N(const uint32_t (&w)[NW])
{
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint32_t x = w[i] ^ 0x4E4E4E4Eu;
        any |= (x - 0x01010101u) & ~x & 0x80808080u;
    }
    return any != 0;
}

__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(v, d); v = o > v ? o : v; }
    return __builtin_amdgcn_readfirstlane(v);
}

// ---------------------------------------------------------------------------
// Smith-Waterman of R1 (rows) against the first characters of a window (columns), +1/-1/-1 linear gaps, N scores 0.
// The result is the maximum over all cells of the SSW key
//   (H + 1) << 11 | (63 - col) << 5 | (31 - row)
// i.e. the best score, its first column, and the smallest row in that column - the end cell SSW reports (see
// oracle/badger_oracle.c, sw_scan); unpk() takes the 2048 off again (0: no cell with a positive score).  The running key after
// column n1 - 1 / n2 - 1 is returned in snap1 / snap2: the result for the window made of the first n1 / n2 columns.
//
// TWO alignments per lane, one per 16-bit half, so that add / subtract / max run as packed 16-bit instructions on two cells
// at once.  Cells hold G = (H + 1) << 10 as unsigned 16-bit: H - 1 is never negative in that form (G - 1024 >= 0) and nothing
// exceeds 25 << 10 = 0x6400.  Below 0x7C00 an unsigned 16-bit pattern is also a finite non-negative half float whose
// order is the order of the integers, so the three-way maximum of a cell is ONE v_pk_maximum3_f16 (gfx950; kernels run
// with 16-bit denormals kept, and every cell is a multiple of 0x0400, the smallest normal number, anyway) where integer
// instructions need two v_pk_max_u16.  The SSW key is put together per COLUMN: the rows contribute G << 10 | (31-row) to a
// column maximum (again one three-way maximum per two rows), the column adds its number and moves G up one bit.
// Eight instructions per pair of cells in round 3's form, 5.5 here.  (Rounds 1-3 also had a one-alignment 32-bit form for the
// reverse pass of k_finalize_reads; that pass runs through this one now, SINGLE.)
// ---------------------------------------------------------------------------
constexpr uint32_t code_plane(int bit)
{
    uint32_t m = 0;
    for (int i = 0; i < R1_LEN; ++i) if ((icode(R1[i]) >> bit) & 1u) m |= 1u << i;
    return m;
}
constexpr uint32_t R1_P0 = code_plane(0), R1_P1 = code_plane(1);
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)); }
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b)      // v_pk_sub_u16 clamp: max(a - b, 0) per half
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max3_below_7c00(uint32_t a, uint32_t b, uint32_t c)      // v_pk_maximum3_f16; every half < 0x7C00
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(
        __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)), __builtin_bit_cast(f16x2, c)));
}

__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c)      // a * b + c per half, b the same in every lane
{
    uint32_t r;             // (written out: the compiler turns a multiplication by a power of two into a shift and an addition)
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

// The window of one cluster, made ready for sw_block2: the strand's complement folded into the bytes (ASCII bit 2 = bit 1
// of the 2-bit code) and every place behind the window's end turned into 'N', which matches nothing.
template <int NW>
__device__ __forceinline__ void sw_window(uint32_t (&w)[NW], int n, uint32_t comp)
{
    const uint32_t flip = comp ? 0x04040404u : 0u;
#pragma unroll
    for (int d = 0; d < NW; ++d) {
        const int left = n - 4 * d;
        const uint32_t keep = left >= 4 ? 0xFFFFFFFFu : (left <= 0 ? 0u : (1u << (8 * left)) - 1u);
        w[d] = ((w[d] ^ flip) & keep) | (0x4E4E4E4Eu & ~keep);
    }
}

// win: the two windows (sw_window) of this lane in LDS but for their first dwords (firstA / firstB): dword d >= 1 of cluster A
// at win[(d - 1) * 64], of cluster B at win[(NW - 1 + d - 1) * 64] (13 + 13 words a lane: with the re-queue staging a block of
// k_sw_clusters stays below 32 KB, five blocks a compute unit).  The loop takes one dword of each per four columns; held in
// registers the 2 x NW words had to be
// moved down one place per round).  n1 / n2: the two window lengths whose running key is wanted, both clusters packed.
// SINGLE: one alignment a lane (the low halves; the high halves see a window of 'N').  P0 / P1: the bit planes of the rows'
// codes (R1_P0 / R1_P1, or a lane's own: the reverse pass of k_finalize_reads aligns R1[end_read .. 0]), rowmask: the rows
// that exist.
template <int NW, bool WITH_N, bool SINGLE>
__device__ __forceinline__ uint32_t sw_block2(const uint32_t* win, uint32_t firstA, uint32_t firstB, int ndw, uint32_t n1, uint32_t n2,
                                              uint32_t P0, uint32_t P1, uint32_t rowmask,
                                              uint32_t& snap1, uint32_t& snap2)
{
    constexpr uint32_t ONE2 = 0x04000400u;       // 1 << 10 in both halves
    // Cells are kept as G - 1 = H in offset form, floored at 0 by the saturating subtract: every consumer of a cell (the
    // cell to its right, below it, and diagonally below) needs exactly that, so one v_pk_sub_u16 clamp per cell replaces
    // two subtractions and the max with 0.  (An unfloored cell only ever reaches the running key with score <= 0, where
    // the key is never used.)
    uint32_t hm[R1_LEN];
#pragma unroll
    for (int i = 0; i < R1_LEN; ++i) hm[i] = 0u;             // H = 0
    uint32_t acc = 0, s1 = 0, s2 = 0;
    uint32_t curA = firstA, curB = SINGLE ? 0x4E4E4E4Eu : firstB;
#pragma nounroll
    for (int d = 0; d < ndw; ++d) {
        const int dn = d + 1 < NW ? d + 1 : NW - 1;           // (NW >= 2)
        const uint32_t nextA = win[(dn - 1) * 64], nextB = SINGLE ? 0x4E4E4E4Eu : win[(NW - 1 + dn - 1) * 64];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = d * 4 + b;
            // Equality words (row i = bit i; bits 22.. hold anything, no row reads them) from the two bit planes of R1's codes
            // and the base's code bits spread over the word; ASCII bit 3 = 'N' or behind the window: matches nothing.
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_sbfe((int)curA, 8 * b + 1, 1), a1 = (uint32_t)__builtin_amdgcn_sbfe((int)curA, 8 * b + 2, 1);
            const uint32_t an = (uint32_t)__builtin_amdgcn_sbfe((int)curA, 8 * b + 3, 1);
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_sbfe((int)curB, 8 * b + 1, 1), b1 = (uint32_t)__builtin_amdgcn_sbfe((int)curB, 8 * b + 2, 1);
            const uint32_t bn = (uint32_t)__builtin_amdgcn_sbfe((int)curB, 8 * b + 3, 1);
            const uint32_t eA = __builtin_amdgcn_bitop3_b32(a0, P0, __builtin_amdgcn_bitop3_b32(a1, P1, an, 0xBE), 0x41) & rowmask;
            const uint32_t eB = SINGLE ? 0u : __builtin_amdgcn_bitop3_b32(b0, P0, __builtin_amdgcn_bitop3_b32(b1, P1, bn, 0xBE), 0x41);
            const uint32_t M0 = __builtin_amdgcn_perm(eB, eA, 0x05040100u);  // rows 0..15 of both
            const uint32_t M1 = __builtin_amdgcn_perm(eB, eA, 0x07060302u);  // rows 16..21 (and what lies above them)
            const uint32_t M2 = M0 >> 4;                                     // rows 12..15 at bits 8..11 of both halves
            // N scores 0.  Places behind a window's end are 'N' too: in the form without N they simply match nothing (-1, as
            // before); here they score 0, which cannot lift a cell above the best one before them - and at equal score the
            // earlier column wins.
            const uint32_t dN = WITH_N ? ((an & 0x0400u) | (bn & 0x04000000u)) : 0u;
            uint32_t diag_t = 0u;        // (H(-1, j-1) - 1) in offset form
            uint32_t upm = 0u;           // (H(-1, j) - 1) likewise
            uint32_t colkey = 0u, kprev = 0u;
#pragma unroll
            for (int i = 0; i < R1_LEN; ++i) {
                const uint32_t tl = hm[i];                                               // H(i, j-1) - 1
                // The row's match bit stays where it is in its word (bit p <= 11 of both halves) and is scaled to +2 by
                // the multiplier of a packed multiply-add: one AND and one v_pk_mad_u16 where shift, AND and add were three.
                const uint32_t src = i < 12 ? M0 : (i < 16 ? M2 : M1);
                const int p = i < 12 ? i : (i < 16 ? i - 4 : i - 16);
                uint32_t dg = pk_mad(src & (0x00010001u << p), 0x00010001u << (11 - p), diag_t);      // H(i-1,j-1) +/- 1
                if (WITH_N) dg = pk_add(dg, dN);
                const uint32_t g = pk_max3_below_7c00(dg, tl, upm);                      // max(diag, left-1, up-1); the floor comes next
                const uint32_t gm = pk_sub_sat(g, ONE2);                                 // max(H, 0) - 1 in offset form
                diag_t = tl;
                hm[i] = gm;
                upm = gm;
                const uint32_t k = g | ((uint32_t)(31 - i) * 0x00010001u);
                if (i & 1) colkey = pk_max3_below_7c00(colkey, kprev, k);
                else kprev = k;
            }
            // the column's best cell as an SSW key: G up one bit, the column number in between
            const uint32_t cj = (uint32_t)((63 - j) << 5) * 0x00010001u;
            const uint32_t key = ((colkey & 0x7C007C00u) << 1) | (colkey & 0x001F001Fu) | cj;
            acc = pk_max(acc, key);
            // the running key after column n1 - 1 / n2 - 1: all ones in the half whose length is j + 1
            const uint32_t jj = (uint32_t)(j + 1) * 0x00010001u;
            const uint32_t m1 = pk_sub(pk_min(n1 ^ jj, 0x00010001u), 0x00010001u);
            const uint32_t m2 = pk_sub(pk_min(n2 ^ jj, 0x00010001u), 0x00010001u);
            s1 = (acc & m1) | (s1 & ~m1);
            s2 = (acc & m2) | (s2 & ~m2);
        }
        curA = nextA; curB = nextB;
    }
    snap1 = s1; snap2 = s2;
    return acc;
}

// packed half -> sw_block key
__device__ __forceinline__ uint32_t unpk(uint32_t packed, int half)
{
    const uint32_t k = half ? packed >> 16 : packed & 0xFFFFu;
    return k >= (uint32_t)ONE ? k - (uint32_t)ONE : 0u;
}

__device__ __forceinline__ uint64_t make_key(uint32_t acc, uint32_t pos)
{
    const uint64_t score = acc >> KEY_SHIFT;
    const uint64_t end_ref = 63u - ((acc >> 5) & 63u);
    const uint64_t end_read = 31u - (acc & 31u);
    return (score << 43) | ((uint64_t)(0xFFFFFFFFu - pos) << 11) | (end_ref << 5) | end_read;
}

// ---------------------------------------------------------------------------
// k_strict_filter: one lane per queue-B cluster.  A local alignment of R1 with score >= 17
// (+1/-1/-1, N = 0) leaves at most 5 of the 22 pattern bases unmatched or mispaired, so R1
// occurs in the window with semi-global edit distance <= 5.  Myers' 22-bit search decides that
// at about a tenth of the cost of the alignment; only surviving hits join queue A.
// ---------------------------------------------------------------------------

__device__ __forceinline__ uint32_t myers_search(uint32_t (&w)[10], int n, uint32_t comp)
{
    // Only bit 21 of the vectors is ever read and carries only move upwards, so bits 22..31 are left to hold anything.
    // A column is 20 vector instructions (round 4; the compiler's form of the textbook statements was 29): the equality mask
    // comes from the two bit planes of R1's codes and the base's code bits spread over the word, with "matches nothing" - an
    // N, or a place behind the window's end, turned into 'N' once before the loop - told by ASCII bit 3 (set in 'N' alone
    // among ACGTN) inside the same two three-input operations; every other line of the recurrence that joins three words is
    // one v_bitop3 as well.
    uint32_t pv = 0x3FFFFFu, mv = 0u, score = R1_LEN, best = R1_LEN;
    const uint32_t flip = comp ? 0x04040404u : 0u;           // complement = 2-bit code ^ 2 = ASCII bit 2
#pragma unroll
    for (int d = 0; d < 10; ++d) {
        const int left = n - 4 * d;                          // window bases in this word and behind it
        const uint32_t keep = left >= 4 ? 0xFFFFFFFFu : (left <= 0 ? 0u : (1u << (8 * left)) - 1u);
        w[d] = ((w[d] ^ flip) & keep) | (0x4E4E4E4Eu & ~keep);
    }
#pragma nounroll
    for (int d = 0; d < 10; ++d) {                           // rolled: keeps the kernel at 8 waves per SIMD, which the gathers need
        const uint32_t cur = w[0];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)cur, 8 * b + 1, 1);      // 0 or ~0
            const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)cur, 8 * b + 2, 1);
            const uint32_t mn = (uint32_t)__builtin_amdgcn_sbfe((int)cur, 8 * b + 3, 1);
            const uint32_t off = __builtin_amdgcn_bitop3_b32(m1, R1_P1, mn, 0xBE);            // (m1 ^ P1) | mn: where the base cannot match
            const uint32_t eq = __builtin_amdgcn_bitop3_b32(m0, R1_P0, off, 0x41);            // ~((m0 ^ P0) | off)
            const uint32_t xv = eq | mv;
            const uint32_t xh = __builtin_amdgcn_bitop3_b32((eq & pv) + pv, pv, eq, 0xBE);     // (((eq & pv) + pv) ^ pv) | eq
            const uint32_t ph = __builtin_amdgcn_bitop3_b32(mv, xh, pv, 0xF1);                // mv | ~(xh | pv)
            const uint32_t mh = pv & xh;
            score += (ph >> (R1_LEN - 1)) & 1u;
            score -= (mh >> (R1_LEN - 1)) & 1u;
            const uint32_t ph1 = ph << 1, mh1 = mh << 1;                                       // search: D[0][j] = 0
            pv = __builtin_amdgcn_bitop3_b32(mh1, xv, ph1, 0xF1);                             // mh1 | ~(xv | ph1)
            mv = ph1 & xv;
            // columns past the window have eq = 0, where the score cannot go down: no need to exclude them here
            best = score < best ? score : best;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) w[i] = w[i + 1];
    }
    return best;
}

__global__ __launch_bounds__(256)
void k_strict_filter(const uint8_t* __restrict__ bases, uint64_t total_rounded,
                     const uint64_t* __restrict__ off,
                     const QEnt* __restrict__ qb, QEnt* __restrict__ qd, uint64_t seg,
                     unsigned long long* __restrict__ counters,
                     const unsigned long long* __restrict__ keys)
{
    __shared__ uint2 s_buf[4][192];          // live hits {read, (pos << 1) | strand}, compacted per wave (< 64 waiting + <= 128 new)
    __shared__ uint2 s_out[4][128];          // survivors, flushed with one reservation
    __shared__ uint32_t s_cnt[NSH];
    const uint64_t nb = queue_counts(counters, K_NAB, 1, seg, s_cnt);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t shard = blockIdx.x % NSH;
    const uint64_t stride = (uint64_t)gridDim.x * 256ull;
    uint32_t nkept = 0, nskip = 0, nbhits = 0, nbuf = 0, nout = 0;

    auto process = [&](uint2 h, bool active) {
        uint64_t rs = 0; int64_t L = 0;
        if (active) { rs = off[h.x]; L = (int64_t)(off[h.x + 1] - rs); }
        const uint32_t strand = h.y & 1u;
        const int64_t pos = (int64_t)(h.y >> 1);
        const int64_t ws = pos - (R1_LEN - KMER) > 0 ? pos - (R1_LEN - KMER) : 0;
        const int64_t we = pos + R1_LEN + 1 < L ? pos + R1_LEN + 1 : L;
        uint32_t w[10];
        load_block<10>(bases, total_rounded, rs, L, (int)strand, ws, +1, w);
        const uint32_t k = myers_search(w, active ? (int)(we - ws) : 0, strand);
        const bool keep = active && k <= 5u;
        const unsigned long long m = __ballot(keep);
        if (m) {
            const uint32_t cnt = (uint32_t)__popcll(m);
            if (nout + cnt > 128u) stage_flush(s_out[wv], nout, lane, qd, seg, shard, ctr(counters, shard, K_NC));
            if (keep) s_out[wv][nout + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = h;
            nout += cnt; nkept += cnt;
        }
    };

    for (uint64_t base = (uint64_t)blockIdx.x * 256ull + (threadIdx.x & ~63); base < nb; base += stride) {
        const QEnt e = queue_fetch(qb, seg, s_cnt, base + lane, nb);       // a cluster: first hit + offset mask of all its hits
        uint32_t mask = e.x != HOLE_R ? e.z : 0u;
        nbhits += __popc(mask);
        if (mask) {
            // The strict search only runs when the relaxed one found nothing acceptable (barcode_callers.py:195).
            // Every relaxed candidate of this read-strand has been aligned by now: if its winner passes
            // end_delta = 4, the strict result is never looked at.
            const unsigned long long kr = keys[2ull * e.x + (e.y & 1u)];
            if (kr != 0 && (R1_LEN - 1 - (int)(kr & 31u)) <= 4) { nskip += __popc(mask); mask = 0u; }
        }
        // live clusters hand over their hits, at most two per lane and round, to the compacted list
        while (__ballot(mask != 0u)) {
            const int j0 = mask ? __builtin_ctz(mask) : 0;
            const uint32_t m1 = mask & (mask - 1u);
            const int j1 = m1 ? __builtin_ctz(m1) : 0;
            const uint32_t cnt = (mask ? 1u : 0u) + (m1 ? 1u : 0u);
            mask = m1 & (m1 - 1u);
            const uint32_t incl = wave_incl_scan(cnt);
            const uint32_t at = nbuf + incl - cnt;
            if (cnt >= 1u) s_buf[wv][at] = make_uint2(e.x, e.y + ((uint32_t)j0 << 1));
            if (cnt == 2u) s_buf[wv][at + 1u] = make_uint2(e.x, e.y + ((uint32_t)j1 << 1));
            nbuf += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            __builtin_amdgcn_wave_barrier();
            while (nbuf >= 64u) {
                nbuf -= 64u;
                const uint2 h = s_buf[wv][nbuf + lane];
                __builtin_amdgcn_wave_barrier();
                process(h, true);
            }
        }
    }
    if (nbuf) {
        const bool on = (uint32_t)lane < nbuf;
        const uint2 h = on ? s_buf[wv][lane] : make_uint2(0u, 0u);
        process(h, on);
    }
    __builtin_amdgcn_wave_barrier();
    stage_flush(s_out[wv], nout, lane, qd, seg, shard, ctr(counters, shard, K_NC));
    unsigned long long* const stat = ctr(counters, shard, K_STAT);
    if (lane == 0 && nkept) atomicAdd(&stat[S_NKEPT], (unsigned long long)nkept);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nskip += __shfl_xor(nskip, d);
    if (lane == 0 && nskip) atomicAdd(&stat[S_NSKIPPED], (unsigned long long)nskip);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nbhits += __shfl_xor(nbhits, d);
    if (lane == 0 && nbhits) atomicAdd(&stat[S_NBHITS], (unsigned long long)nbhits);
}

// ---------------------------------------------------------------------------
// k_sw_clusters: one lane per cluster (detect_exact_positions loop body,
// barcode_extraction/common.py:91-103, for the relaxed and the strict search at once).
// ---------------------------------------------------------------------------
constexpr int CW = 14;                       // 56 columns: window start 16 before the first hit, hits spanning <= 17 positions, 23 after the last

struct ClusterJob {
    bool active, relaxed;
    uint32_t r, strand, mask;
    int64_t pos, ws, L;
    uint64_t rs;
    int n_s, n_r, n_u;
};

__device__ __forceinline__ ClusterJob make_job(const QEnt e,
                                               const uint64_t* __restrict__ off, const int32_t* __restrict__ polyt)
{
    ClusterJob jb;
    jb.active = e.x != HOLE_R;
    jb.r = jb.active ? e.x : 0u; jb.strand = e.y & 1u; jb.mask = jb.active ? e.z : 0u;
    jb.pos = jb.active ? (int64_t)(e.y >> 1) : 0;
    jb.rs = 0; jb.L = 0; int32_t pt = -1;
    if (jb.active) { jb.rs = off[jb.r]; jb.L = (int64_t)(off[jb.r + 1] - jb.rs); pt = polyt[2 * (uint64_t)jb.r + jb.strand]; }
    jb.ws = jb.pos - (R1_LEN - KMER) > 0 ? jb.pos - (R1_LEN - KMER) : 0;                 // common.py:96-97
    const int64_t we = jb.pos + R1_LEN + 1 < jb.L ? jb.pos + R1_LEN + 1 : jb.L;          // :98-99 (strict: end = len)
    const int64_t last = jb.pos + (jb.mask ? 31 - __builtin_clz(jb.mask) : 0);
    const int64_t wu = last + R1_LEN + 1 < jb.L ? last + R1_LEN + 1 : jb.L;              // end of the cluster's union
    // relaxed search (barcode_callers.py:186-192): hits inside sequence[0:polyT+1], end = polyT+1
    jb.relaxed = jb.active && pt >= 0 && jb.pos + KMER <= (int64_t)pt + 1;
    const int64_t we_r = jb.pos + R1_LEN + 1 < (int64_t)pt + 1 ? jb.pos + R1_LEN + 1 : (int64_t)pt + 1;
    jb.n_s = jb.active ? (int)(we - jb.ws) : 0;
    jb.n_r = jb.relaxed ? (int)(we_r - jb.ws) : 0;
    jb.n_u = jb.active ? (int)(wu - jb.ws) : 0;
    return jb;
}

// keys of the first hit; returns the other hits of the cluster when the union beats it (they must be aligned one by one)
__device__ __forceinline__ uint32_t finish_job(const ClusterJob& jb, uint32_t acc_s, uint32_t acc_r, uint32_t acc_u,
                                               uint32_t n_reads, unsigned long long* __restrict__ keys)
{
    const uint32_t score_s = acc_s >> KEY_SHIFT;
    if (jb.active && score_s >= 17u)                                                    // barcode_callers.py:200
        atomicMax(&keys[2ull * n_reads + 2ull * jb.r + jb.strand], (unsigned long long)make_key(acc_s, (uint32_t)jb.pos));
    if (jb.relaxed && (acc_r >> KEY_SHIFT) >= 9u)                                        // barcode_callers.py:191
        atomicMax(&keys[2ull * jb.r + jb.strand], (unsigned long long)make_key(acc_r, (uint32_t)jb.pos));
    // Every window of the cluster lies inside the union, so no later hit scores above the union.
    // If the union does not beat the first hit, none of them can replace it (strictly greater is
    // required, common.py:102); otherwise align them one by one (queue C, second launch).
    return (jb.active && (acc_u >> KEY_SHIFT) > score_s) ? (jb.mask & ~1u) : 0u;
}

constexpr uint32_t REQ_CAP = 128;            // per-wave staging of re-queued hits (2 % of the clusters re-queue any)

__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(5, 5))) void k_sw_clusters(const uint8_t* __restrict__ bases, uint64_t total_rounded,
                   const uint64_t* __restrict__ off, uint32_t n_reads,
                   const int32_t* __restrict__ polyt,
                   const QEnt* __restrict__ q, int kind, uint64_t seg,
                   QEnt* __restrict__ qc,
                   unsigned long long* __restrict__ counters,
                   unsigned long long* __restrict__ keys)
{
    __shared__ uint2 s_out[4][REQ_CAP];
    __shared__ uint32_t s_win[4][2 * (CW - 1)][64];    // the two windows of every lane but for their first words (sw_block2)
    __shared__ uint32_t s_cnt[NSH];
    const uint64_t nq = queue_counts(counters, kind, 0, seg, s_cnt);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t shard = blockIdx.x % NSH;
    unsigned long long* const nc = ctr(counters, shard, K_NC);
    const uint64_t stride = (uint64_t)gridDim.x * 512ull;           // two clusters per lane
    uint32_t nwin = 0, nout = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * 512ull + 2ull * (threadIdx.x & ~63); base < nq; base += stride) {
        const ClusterJob ja = make_job(queue_fetch(q, seg, s_cnt, base + 2ull * lane, nq), off, polyt);
        const ClusterJob jb = make_job(queue_fetch(q, seg, s_cnt, base + 2ull * lane + 1, nq), off, polyt);
        uint32_t wa[CW], wb[CW];
        load_block<CW>(bases, total_rounded, ja.rs, ja.L, (int)ja.strand, ja.ws, +1, wa);
        load_block<CW>(bases, total_rounded, jb.rs, jb.L, (int)jb.strand, jb.ws, +1, wb);
        const bool anyN = __ballot((ja.active && block_has_N<CW>(wa)) || (jb.active && block_has_N<CW>(wb))) != 0;
        const int ndw = (wave_max(ja.n_u > jb.n_u ? ja.n_u : jb.n_u) + 3) >> 2;
        sw_window<CW>(wa, ja.n_u, ja.strand);
        sw_window<CW>(wb, jb.n_u, jb.strand);
        uint32_t* const win = &s_win[wv][0][lane];
#pragma unroll
        for (int d = 1; d < CW; ++d) { win[(d - 1) * 64] = wa[d]; win[(CW - 1 + d - 1) * 64] = wb[d]; }
        __builtin_amdgcn_wave_barrier();
        uint32_t sn_s = 0, sn_r = 0;
        const uint32_t n_s2 = (uint32_t)ja.n_s | ((uint32_t)jb.n_s << 16), n_r2 = (uint32_t)ja.n_r | ((uint32_t)jb.n_r << 16);
        const uint32_t acc = anyN ? sw_block2<CW, true, false>(win, wa[0], wb[0], ndw, n_s2, n_r2, R1_P0, R1_P1, 0xFFFFFFFFu, sn_s, sn_r)
                                  : sw_block2<CW, false, false>(win, wa[0], wb[0], ndw, n_s2, n_r2, R1_P0, R1_P1, 0xFFFFFFFFu, sn_s, sn_r);
        __builtin_amdgcn_wave_barrier();
        nwin += (ja.active ? 1u : 0u) + (jb.active ? 1u : 0u);
        uint32_t rest_a = finish_job(ja, unpk(sn_s, 0), unpk(sn_r, 0), unpk(acc, 0), n_reads, keys);
        uint32_t rest_b = finish_job(jb, unpk(sn_s, 1), unpk(sn_r, 1), unpk(acc, 1), n_reads, keys);
        if (__ballot((rest_a | rest_b) != 0)) {
            const uint32_t cnt = __popc(rest_a) + __popc(rest_b);
            uint32_t incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            const uint32_t total = __shfl(incl, 63);
            if (nout + total > REQ_CAP) stage_flush(s_out[wv], nout, lane, qc, seg, shard, nc);
            if (total <= REQ_CAP) {
                uint32_t idx = nout + incl - cnt;
                while (rest_a) {
                    const int j = __builtin_ctz(rest_a); rest_a &= rest_a - 1;
                    s_out[wv][idx++] = make_uint2(ja.r, ((uint32_t)(ja.pos + j) << 1) | ja.strand);
                }
                while (rest_b) {
                    const int j = __builtin_ctz(rest_b); rest_b &= rest_b - 1;
                    s_out[wv][idx++] = make_uint2(jb.r, ((uint32_t)(jb.pos + j) << 1) | jb.strand);
                }
                nout += total;
                __builtin_amdgcn_wave_barrier();
            } else {                       // more than the staging holds (dense repeats): straight to the queue
                unsigned long long gb = 0;
                if (lane == 63) gb = atomicAdd(nc, (unsigned long long)total);
                gb = __shfl(gb, 63);
                unsigned long long idx = gb + incl - cnt;
                while (rest_a) {
                    const int j = __builtin_ctz(rest_a); rest_a &= rest_a - 1;
                    if (idx < seg) qc[shard * seg + idx] = make_uint4(ja.r, ((uint32_t)(ja.pos + j) << 1) | ja.strand, 1u, 0u);
                    ++idx;
                }
                while (rest_b) {
                    const int j = __builtin_ctz(rest_b); rest_b &= rest_b - 1;
                    if (idx < seg) qc[shard * seg + idx] = make_uint4(jb.r, ((uint32_t)(jb.pos + j) << 1) | jb.strand, 1u, 0u);
                    ++idx;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    stage_flush(s_out[wv], nout, lane, qc, seg, shard, nc);
    // window count (statistics only)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nwin += __shfl_xor(nwin, d);
    if (lane == 0 && nwin) atomicAdd(&ctr(counters, shard, K_STAT)[S_NWINDOWS], (unsigned long long)nwin);
}

// ---------------------------------------------------------------------------
// k_finalize_reads
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t strand_byte(const uint8_t* __restrict__ rd, int64_t L, int strand, int64_t x)
{
    // character x of the strand, returned as the FORWARD byte (complement applied by the caller's test)
    return rd[strand ? (L - 1 - x) : x];
}

struct StrandRes { int32_t valid, polyT, r1, score, bc_start, umi_start, umi_end; uint32_t bc[4]; /* 16 barcode characters, strand order, forward bytes */ };

__device__ __forceinline__ StrandRes finalize_strand(const uint8_t* __restrict__ bases, uint64_t total_rounded,
                                     uint64_t rs, int64_t L, int strand, int32_t pt,
                                     uint64_t kr, uint64_t ks, int umi_len, bool active, uint32_t* win /* LDS: this lane's window words, 64 apart */)
{
    StrandRes res = { 0, pt, -1, 0, -1, -1, -1, { 0, 0, 0, 0 } };
    bool found = false;
    int64_t r1_end = 0; int32_t r1_score = 0;
    if (active && pt != -1 && kr != 0) {                       // relaxed: min_score 9, end_delta 4
        const int64_t pos = (int64_t)(0xFFFFFFFFu - (uint32_t)(kr >> 11));
        const int end_ref = (int)((kr >> 5) & 63), end_read = (int)(kr & 31);
        const int64_t ws = pos - (R1_LEN - KMER) > 0 ? pos - (R1_LEN - KMER) : 0;
        const int leftover = R1_LEN - end_read - 1;            // common.py:113
        if (leftover <= 4) { found = true; r1_end = ws + end_ref + leftover; r1_score = (int32_t)(kr >> 43); }
    }
    // strict: min_score 17, start_delta 1, end_delta 1 -> needs read_start (reverse pass)
    int end_ref_s = 0, end_read_s = 0; int64_t ws_s = 0;
    bool need_rev = false;
    if (active && !found && ks != 0) {
        const int64_t pos = (int64_t)(0xFFFFFFFFu - (uint32_t)(ks >> 11));
        end_ref_s = (int)((ks >> 5) & 63); end_read_s = (int)(ks & 31);
        ws_s = pos - (R1_LEN - KMER) > 0 ? pos - (R1_LEN - KMER) : 0;
        need_rev = (R1_LEN - end_read_s - 1) <= 1;             // common.py:110-111
    }
    if (__ballot(need_rev)) {
        // SSW's reverse pass: R1[end_read .. 0] against the read walked backwards from the end cell - the packed kernel of
        // k_sw_clusters with one alignment a lane, the rows' bit planes reversed and shifted per lane (row r = R1[end_read - r])
        uint32_t w[10];
        load_block<10>(bases, total_rounded, rs, L, strand, ws_s + end_ref_s, -1, w);
        const bool anyN = __ballot(need_rev && block_has_N<10>(w)) != 0;
        const int ncol = need_rev ? end_ref_s + 1 : 0;
        sw_window<10>(w, ncol, (uint32_t)strand);
#pragma unroll
        for (int d = 1; d < 10; ++d) win[(d - 1) * 64] = w[d];
        __builtin_amdgcn_wave_barrier();
        const int ndw = (wave_max(ncol) + 3) >> 2;
        const uint32_t p0r = __brev(R1_P0) >> (31 - end_read_s), p1r = __brev(R1_P1) >> (31 - end_read_s);
        const uint32_t rows = (2u << end_read_s) - 1u;
        uint32_t sn1, sn2;
        const uint32_t acc = anyN ? sw_block2<10, true, true>(win, w[0], 0u, ndw, 0u, 0u, p0r, p1r, rows, sn1, sn2)
                                  : sw_block2<10, false, true>(win, w[0], 0u, ndw, 0u, 0u, p0r, p1r, rows, sn1, sn2);
        __builtin_amdgcn_wave_barrier();
        if (need_rev) {
            const int rr = 31 - (int)(unpk(acc, 0) & 31u);             // (the pass always finds the forward score again: the key is never 0)
            const int read_begin = end_read_s - rr;
            if (read_begin <= 1) {                             // common.py:108-109
                found = true;
                r1_end = ws_s + end_ref_s + (R1_LEN - end_read_s - 1);
                r1_score = (int32_t)(ks >> 43);
            }
        }
    }
    if (!found) return res;                                                        // barcode_callers.py:204-205
    if (pt != -1 && (int64_t)pt - r1_end < BC_LEN) return res;                     // :208-209
    // barcode characters and the polyT re-search window, loaded together (wide loads, one latency)
    const int64_t barcode_start = r1_end + 1;
    const bool research = pt == -1 || (int64_t)pt - r1_end > BC_LEN + umi_len + 10;   // :211-218
    const int64_t presumable = r1_end + BC_LEN + umi_len;
    const int64_t ss = presumable - 4;
    const int64_t se = presumable + 10 < L ? presumable + 10 : L;
    const int64_t sl = (research && ss < L && se > ss) ? se - ss : 0;
    load_block<4>(bases, total_rounded, rs, L, strand, barcode_start, +1, res.bc);
    int64_t polyt = pt;
    if (research) {
        uint32_t w[4];
        load_block<4>(bases, total_rounded, rs, L, strand, ss, +1, w);             // sl <= 14 characters
        polyt = -1;
        const uint32_t tchar = strand ? (uint32_t)'A' : (uint32_t)'T';
        int run = 0;
#pragma unroll
        for (int x = 0; x < 14; ++x) {              // first all-T window of 5 starting at i < sl-5
            const uint32_t b = (w[x >> 2] >> (8 * (x & 3))) & 0xFFu;
            run = (x < sl && b == tchar) ? run + 1 : 0;
            const int64_t i = x - 4;
            if (polyt == -1 && run >= 5 && i < sl - 5) polyt = ss + i;
        }
    }
    const int64_t umi_start = r1_end + BC_LEN + 1;
    int64_t umi_end = polyt - 1;
    if (umi_end - umi_start <= 5) umi_end = umi_start + umi_len - 1;               // :226-227
    res.valid = 1; res.polyT = (int32_t)polyt; res.r1 = (int32_t)r1_end; res.score = r1_score;
    res.bc_start = (int32_t)barcode_start; res.umi_start = (int32_t)umi_start; res.umi_end = (int32_t)(umi_end + 1);
    return res;
}

__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(7, 7))) void k_finalize_reads(const uint8_t* __restrict__ bases, uint64_t total_rounded,
                      const uint64_t* __restrict__ off, uint32_t n,
                      const int32_t* __restrict__ polyt,
                      const unsigned long long* __restrict__ keys,
                      const unsigned long long* __restrict__ counters, uint64_t qcap,
                      uint32_t umi_len, int strand_rule, bdg_extract_rec* __restrict__ out,
                      unsigned long long* __restrict__ next_counters)
{
    // The NEXT batch's counters (the other of two sets) are cleared here, by the last kernel of this batch: a memset of its
    // own between two batches cost the stream 20 us of switching from kernels to a fill and back (round 4, kernel trace).
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < (uint32_t)(COUNTER_BYTES / 8); i += 256u) next_counters[i] = 0ull;
    const uint64_t r = (uint64_t)blockIdx.x * 256ull + threadIdx.x;
    const bool active = r < n;
    // A queue that overflowed dropped alignment candidates: no record of this batch can be trusted.  Every record is
    // then written as "not extracted, batch incomplete", so that whatever consumes the records next on the stream
    // (bdg_nearest16_recs_dev, bdg_distinct_dev) sees nothing usable; bdg_extract_status() reports BDG_E_CAPACITY.
    __shared__ uint32_t s_over;
    __shared__ uint32_t s_rwin[4][9][64];            // the reverse pass's windows but for their first words (finalize_strand)
    if (threadIdx.x == 0) s_over = 0u;
    __syncthreads();
    if (threadIdx.x < NSH) {
        const unsigned long long ab = *ctr(const_cast<unsigned long long*>(counters), threadIdx.x, K_NAB);
        const unsigned long long c = *ctr(const_cast<unsigned long long*>(counters), threadIdx.x, K_NC);
        const unsigned long long d = *ctr(const_cast<unsigned long long*>(counters), threadIdx.x, K_ND);
        if ((ab & 0xFFFFFFFFull) > qcap || (ab >> 32) > qcap || c > qcap || d > qcap) s_over = 1u;
    }
    __syncthreads();
    if (s_over) {
        if (active) {
            bdg_extract_rec rec;
            rec.polyT = -1; rec.r1_end = -1; rec.bc_start = -1; rec.umi_start = -1; rec.umi_end = -1; rec.bc_rank = 0;
            rec.r1_score = 0; rec.strand = 0; rec.valid = 0; rec.flags = BDG_FLAG_INCOMPLETE; rec.reserved = 0;
            out[r] = rec;
        }
        return;
    }
    uint64_t rs = 0; int64_t L = 0;
    int32_t ptF = -1, ptR = -1; uint64_t krF = 0, krR = 0, ksF = 0, ksR = 0;
    if (active) {
        rs = off[r]; L = (int64_t)(off[r + 1] - rs);
        ptF = polyt[2 * r]; ptR = polyt[2 * r + 1];
        krF = keys[2 * r]; krR = keys[2 * r + 1];
        ksF = keys[2ull * n + 2 * r]; ksR = keys[2ull * n + 2 * r + 1];
    }
    uint32_t* const win = &s_rwin[threadIdx.x >> 6][0][threadIdx.x & 63];
    const StrandRes f = finalize_strand(bases, total_rounded, rs, L, 0, ptF, krF, ksF, (int)umi_len, active, win);
    const StrandRes v = finalize_strand(bases, total_rounded, rs, L, 1, ptR, krR, ksR, (int)umi_len, active, win);
    if (!active) return;
    bool use_rev;
    if (strand_rule == BDG_STRAND_RULE_NO_POLYA) use_rev = !f.valid; // find_barcode_umi_no_polya, :234-247: forward if valid, else
                                                                     // reverse (valid, or neither is: both scores are 0 then, :247)
    else if (v.valid && f.valid) use_rev = !(f.score > v.score);     // barcode_callers.py:175-176
    else use_rev = v.valid != 0;                                     // :177-179
    const StrandRes c = use_rev ? v : f;
    bdg_extract_rec rec;
    rec.polyT = c.polyT; rec.r1_end = c.r1; rec.bc_start = c.bc_start;
    rec.umi_start = c.umi_start; rec.umi_end = c.umi_end;
    rec.bc_rank = 0;
    rec.r1_score = (int8_t)c.score;
    rec.strand = c.polyT != -1 ? (use_rev ? -1 : 1) : 0;             // :167-168,172-173
    rec.valid = (uint8_t)c.valid;
    rec.flags = use_rev ? BDG_FLAG_REV : 0;
    rec.reserved = 0;
    if (c.valid && (int64_t)c.bc_start + BC_LEN <= L) {
        uint32_t rk = 0; bool ok = true;
#pragma unroll
        for (int i = 0; i < BC_LEN; ++i) {
            const uint32_t b = (c.bc[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            // rank codes A0 C1 G2 T3 (common.py:11-14); the strand's letter is the complement of a reverse-strand byte
            uint32_t cd = b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : 3u;
            ok = ok && (b == 'A' || b == 'C' || b == 'G' || b == 'T');
            if (use_rev) cd = 3u - cd;
            rk |= cd << (2 * i);
        }
        rec.flags |= BDG_FLAG_BC16;
        if (ok) { rec.bc_rank = rk; rec.flags |= BDG_FLAG_RANK_OK; }
    }
    out[r] = rec;
}

// host-side table -------------------------------------------------------------
// 7-mer probe table of k_scan_reads: index = 2-bit codes of 7 consecutive bases (base p in bits 0-1), entry bit 0 / 1:
// bases p..p+5 / p+1..p+6 spell a 6-mer of R1, bit 4 / 5: the same for the reverse complement of a 6-mer of R1
// (KmerIndexer over [R1] with k = 6, kmer_indexer.py:20-27, applied to the read and to its reverse complement).
void build_kmer7_table(uint8_t* k7 /* [16384] */)
{
    static uint8_t six[4096];                       // bit 0: forward 6-mer, bit 1: reverse-complement 6-mer
    memset(six, 0, sizeof(six));
    auto comp = [](char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; };
    for (int q = 0; q + KMER <= R1_LEN; ++q) {
        uint32_t kf = 0, kr = 0;
        for (int u = 0; u < KMER; ++u) {
            kf |= icode(R1[q + u]) << (2 * u);
            kr |= icode(comp(R1[q + KMER - 1 - u])) << (2 * u);
        }
        six[kf] |= 1u;
        six[kr] |= 2u;
    }
    for (uint32_t idx = 0; idx < 16384u; ++idx) {
        const uint32_t e0 = six[idx & 4095u], e1 = six[idx >> 2];
        k7[idx] = (uint8_t)((e0 & 1u) | ((e1 & 1u) << 1) | (((e0 >> 1) & 1u) << 4) | (((e1 >> 1) & 1u) << 5));
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// host entry (called from bdg_abi.cpp)
// ---------------------------------------------------------------------------
int bdg_extract_launch(bdg_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_off, uint32_t n,
                       uint64_t total_bytes, uint32_t umi_len, bdg_extract_rec* d_out)
{
    if (n == 0) return BDG_OK;
    if (reinterpret_cast<uintptr_t>(d_bases) & 15u) return bdg_fail(ctx, BDG_E_ARG, "d_bases must be 16-byte aligned");
    if (total_bytes >= (1ull << 62)) return bdg_fail(ctx, BDG_E_ARG, "total_bytes too large");
    int rc;
    if (!ctx->x_lut.p) {
        static uint8_t t[16384];
        build_kmer7_table(t);
        if ((rc = bdg_reserve(ctx, ctx->x_lut, sizeof(t)))) return rc;
        BDG_HIP_TRY(ctx, hipMemcpy(ctx->x_lut.p, t, sizeof(t), hipMemcpyHostToDevice));
    }
    if ((rc = bdg_reserve(ctx, ctx->x_polyt, sizeof(int32_t) * 2ull * n))) return rc;
    if ((rc = bdg_reserve(ctx, ctx->x_keys, sizeof(uint64_t) * 4ull * n))) return rc;
    if ((rc = bdg_reserve(ctx, ctx->x_counters, 2 * COUNTER_BYTES))) return rc;
    // three cluster queues (A: aligned, B: filtered first, C: re-queued hits of a cluster; D, the filter's survivors, reuses A),
    // 16 B per entry, each made of NSH segments of x_hits_cap entries
    uint64_t want = (total_bytes / 48 + 4096 + NSH - 1) / NSH;
    if (want < ctx->x_hits_cap) want = ctx->x_hits_cap;
    if (ctx->x_hits_cap_fixed) want = ctx->x_hits_cap_fixed;   // the caller's choice (bdg_extract_set_queue_capacity) until an overflow grows it
    if ((rc = bdg_reserve(ctx, ctx->x_hits, sizeof(QEnt) * 3ull * NSH * want))) return rc;
    ctx->x_hits_cap = ctx->x_hits_cap_fixed ? want : ctx->x_hits.bytes / sizeof(QEnt) / (3 * NSH);
    const uint64_t qcap = ctx->x_hits_cap;                     // per segment
    ctx->x_hits_cap_launched = qcap;
    QEnt* qa = static_cast<QEnt*>(ctx->x_hits.p);
    QEnt* qb = qa + NSH * qcap;
    QEnt* qc = qb + NSH * qcap;

    hipStream_t st = ctx->stream;
    const uint64_t total_rounded = (total_bytes + 15ull) & ~15ull;
    if (ctx->x_counters_cleared != ctx->x_counters.p) {          // a fresh allocation: both sets once, here
        BDG_HIP_TRY(ctx, hipMemsetAsync(ctx->x_counters.p, 0, 2 * COUNTER_BYTES, st));
        ctx->x_counters_cleared = ctx->x_counters.p;
    }
    ctx->x_counter_set ^= 1u;                                     // this batch's set was cleared by the batch before (k_finalize_reads)
    auto* counters = static_cast<unsigned long long*>(ctx->x_counters.p) + (size_t)ctx->x_counter_set * (COUNTER_BYTES / 8);
    auto* next_counters = static_cast<unsigned long long*>(ctx->x_counters.p) + (size_t)(ctx->x_counter_set ^ 1u) * (COUNTER_BYTES / 8);
    auto* keys = static_cast<unsigned long long*>(ctx->x_keys.p);
    const auto* pt = static_cast<const int32_t*>(ctx->x_polyt.p);
    {
        ScopedKernelTimer tm(ctx, "k_scan_reads");
        const uint32_t ntasks = (n + TASK_READS - 1) / TASK_READS;
        uint32_t grid = (ntasks + SCAN_WAVES - 1) / SCAN_WAVES;
        if (grid > 256u * SCAN_BLOCKS_PER_CU) grid = 256u * SCAN_BLOCKS_PER_CU;   // persistent: as many blocks per CU as the LDS takes
        grid = (grid + TASK_SHARDS - 1) / TASK_SHARDS * TASK_SHARDS;            // every shard has a block
        hipLaunchKernelGGL(k_scan_reads, dim3(grid), dim3(64 * SCAN_WAVES), 0, st, d_bases, total_rounded, d_off, n,
                           static_cast<const uint8_t*>(ctx->x_lut.p), static_cast<int32_t*>(ctx->x_polyt.p),
                           qa, qb, qcap, counters, keys);
    }
    if (ctx->deferred.pending && (rc = bdg_launch_deferred_match(ctx, true))) return rc;      // overlap mode: the match of the batch before
    {
        ScopedKernelTimer tm(ctx, "k_sw_clusters");
        hipLaunchKernelGGL(k_sw_clusters, dim3(256 * 8), dim3(256), 0, st, d_bases, total_rounded, d_off, n, pt,
                           qa, (int)K_NAB, qcap, qc, counters, keys);
    }
    {
        // Every candidate of the relaxed search that queue A's clusters hold is aligned now.  The hits those clusters re-queued
        // (queue C) wait: the filter appends its survivors to the same queue and ONE more launch aligns both kinds (two
        // latency-bound launches otherwise).  The filter's "relaxed search already succeeded" shortcut then only sees the
        // first-hit alignments, which is all it needs: skipping is an optimisation, never a condition of correctness.
        ScopedKernelTimer tm(ctx, "k_strict_filter");
        hipLaunchKernelGGL(k_strict_filter, dim3(256 * 8), dim3(256), 0, st, d_bases, total_rounded, d_off,
                           qb, qc, qcap, counters, keys);
    }
    {
        ScopedKernelTimer tm(ctx, "k_sw_singles");
        hipLaunchKernelGGL(k_sw_clusters, dim3(256 * 2), dim3(256), 0, st, d_bases, total_rounded, d_off, n, pt,
                           qc, (int)K_NC, qcap, qa, counters, keys);      // single hits: nothing is re-queued
    }
    {
        ScopedKernelTimer tm(ctx, "k_finalize_reads");
        hipLaunchKernelGGL(k_finalize_reads, dim3((n + 255) / 256), dim3(256), 0, st, d_bases, total_rounded, d_off, n,
                           static_cast<const int32_t*>(ctx->x_polyt.p),
                           static_cast<const unsigned long long*>(ctx->x_keys.p), counters, qcap, umi_len, ctx->x_strand_rule, d_out,
                           next_counters);
    }
    BDG_HIP_TRY(ctx, hipGetLastError());
    return BDG_OK;
}

namespace {
struct CounterSums { uint64_t a_max, b_max, c_max, a, b, c, d, bad, stat[S_N]; };

void sum_counters(const uint64_t* c, CounterSums& cs)
{
    memset(&cs, 0, sizeof(cs));
    for (int sh = 0; sh < NSH; ++sh) {
        const uint64_t* w = c + (size_t)sh * SH_WORDS;
        const uint64_t a = w[K_NAB * 16] & 0xFFFFFFFFull, b = w[K_NAB * 16] >> 32, cc = w[K_NC * 16];
        cs.a += a; cs.b += b; cs.c += cc; cs.d += w[K_ND * 16];
        cs.a_max = a > cs.a_max ? a : cs.a_max; cs.b_max = b > cs.b_max ? b : cs.b_max; cs.c_max = cc > cs.c_max ? cc : cs.c_max;
        const uint64_t* st = w + K_STAT * 16;
        cs.bad = st[S_BADREAD] > cs.bad ? st[S_BADREAD] : cs.bad;                 // max of ~index = smallest index
        for (int k = 1; k < S_N; ++k) cs.stat[k] += st[k];
    }
}

int read_counters(bdg_ctx* ctx, CounterSums& cs)
{
    std::vector<uint64_t> c(COUNTER_BYTES / 8);
    BDG_HIP_TRY(ctx, hipMemcpyAsync(c.data(), bdg_extract_counters_now(ctx), COUNTER_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    sum_counters(c.data(), cs);
    return BDG_OK;
}

// verdict on one batch from its counters (qcap: the queue capacity that batch ran with); grows the workspace on overflow
int judge_counters(bdg_ctx* ctx, const CounterSums& cs, uint64_t qcap, uint64_t* bad_read, uint64_t* n_windows)
{
    const uint64_t bad = cs.bad ? ~cs.bad : ~0ull;
    if (bad_read) *bad_read = bad;
    if (n_windows) *n_windows = cs.stat[S_NWINDOWS];
    if (cs.a_max > qcap || cs.b_max > qcap || cs.c_max > qcap) {
        // Queues A and B counted every entry they were offered, so their true sizes are known now; queue C (hits re-queued
        // by clusters of A) was fed from a truncated A and may be under-counted: leave room, and the caller loops.
        uint64_t want = cs.a_max > cs.b_max ? cs.a_max : cs.b_max;
        want = (want > cs.c_max ? want : cs.c_max);
        want += want / 2 + 4096;           // queue D (filter survivors) never exceeds queue B
        if (want > ctx->x_hits_cap) ctx->x_hits_cap = want;            // next launch reserves this much per segment
        ctx->x_hits_cap_fixed = 0;
        return bdg_fail(ctx, BDG_E_CAPACITY, "window queue overflow: rerun the batch (workspace grown)");
    }
    if (bad != ~0ull)
        return bdg_fail(ctx, BDG_E_BADBASE, "read " + std::to_string(bad) + " holds a byte outside 'ACGTN'");
    return BDG_OK;
}
}  // namespace

size_t bdg_extract_counter_bytes() { return COUNTER_BYTES; }
// the counters of the batch launched last (one of the two sets)
const void* bdg_extract_counters_now(const bdg_ctx* ctx)
{
    return static_cast<const char*>(ctx->x_counters.p) + (size_t)ctx->x_counter_set * COUNTER_BYTES;
}

// the same verdict from a host copy of the counters taken right behind a batch (bdg_extract_submit / collect)
int bdg_extract_judge_host(bdg_ctx* ctx, const void* host_counters, uint64_t qcap, uint64_t* bad_read, uint64_t* n_windows)
{
    CounterSums cs;
    sum_counters(static_cast<const uint64_t*>(host_counters), cs);
    return judge_counters(ctx, cs, qcap, bad_read, n_windows);
}

int bdg_extract_status_impl(bdg_ctx* ctx, uint64_t* bad_read, uint64_t* n_windows)
{
    if (!ctx->x_counters.p) { if (bad_read) *bad_read = ~0ull; if (n_windows) *n_windows = 0; return BDG_OK; }
    CounterSums cs;
    int rc;
    if ((rc = read_counters(ctx, cs))) return rc;
    return judge_counters(ctx, cs, ctx->x_hits_cap_launched, bad_read, n_windows);
}

int bdg_extract_counters_impl(bdg_ctx* ctx, uint64_t out[8])
{
    memset(out, 0, sizeof(uint64_t) * 8);
    if (!ctx->x_counters.p) return BDG_OK;
    CounterSums cs;
    int rc;
    if ((rc = read_counters(ctx, cs))) return rc;
    out[0] = cs.stat[S_NHITS]; out[1] = cs.a; out[2] = cs.stat[S_NBHITS]; out[3] = cs.stat[S_NSKIPPED];
    out[4] = cs.stat[S_NKEPT]; out[5] = cs.c - cs.stat[S_NKEPT]; out[6] = cs.stat[S_NWINDOWS]; out[7] = cs.b;    // queue C = re-queued + survivors
    return BDG_OK;
}
