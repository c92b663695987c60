// Shared host-side plumbing of libbadger_hip.so: context, error handling,
// per-kernel HIP-event timing, grow-only device workspaces.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/badger_hip.h"

struct DevBuf {
    void*  p = nullptr;
    size_t bytes = 0;
};

struct KTimer {
    std::string name;
    uint64_t launches = 0;
    double total_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct bdg_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // bdg_set_overlap: the whitelist match of a batch's records runs on aux_stream, ordered behind the extraction that wrote
    // them, so that it overlaps the extraction of the NEXT batch on `stream` (its alignment kernels: see DeferredMatch)
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_main = nullptr, ev_aux[2] = { nullptr, nullptr };
    uint64_t aux_count = 0;                 // matches queued on aux_stream so far
    bool overlap = false, aux_pending = false;
    // The match is not queued when it is asked for but behind the NEXT extraction's scan (bdg_launch_deferred_match): beside
    // the scan (which streams the reads at the memory's rate) its gathers cost more than they hide, beside the alignment
    // kernels that follow (integer issue, almost no memory traffic) they are nearly free.
    struct DeferredMatch { bool pending = false; const uint32_t* q = nullptr; uint32_t n = 0, max_ed = 0;
                           uint32_t* idx = nullptr; uint8_t* ed = nullptr; uint16_t* ties = nullptr; } deferred;
    hipEvent_t ev_scan = nullptr;
    hipStream_t launch_stream = nullptr;    // where kernels (and their timing events) currently go: stream, or aux_stream
    std::string err;
    bool profiling = false;
    std::string profile_only;               // non-empty: only this kernel is timed (bdg_profile_only)
    std::vector<KTimer> timers;
    std::vector<hipEvent_t> event_pool;

    // ---- extraction workspace (extract_kernels.hip)
    DevBuf x_lut;        // 7-mer probe table of k_scan_reads (16 KiB)
    DevBuf x_polyt;      // int32 [2n]
    DevBuf x_keys;       // uint64 [4n]  relaxed[2n] | strict[2n]
    DevBuf x_hits;       // uint64 [hits_cap]
    DevBuf x_counters;   // two sets of the extraction's counters: a batch uses one and its last kernel clears the other for the next
    uint32_t x_counter_set = 0;          // the set of the batch launched last
    void* x_counters_cleared = nullptr;  // the allocation both sets of which have been cleared once
    uint64_t x_hits_cap = 0;
    uint64_t x_hits_cap_fixed = 0;     // bdg_extract_set_queue_capacity (0 = automatic)
    int x_strand_rule = 0;             // bdg_extract_set_strand_rule
    uint64_t x_hits_cap_launched = 0;  // capacity the last launch ran with
    void* x_counters_host = nullptr;   // pinned mirror
    // host-buffer staging
    DevBuf s_in0, s_in1, s_out0;
    // pipelined chunks (bdg_extract_submit / collect)
    struct Slot {
        DevBuf d_bases, d_off, d_recs;
        void* h_recs = nullptr; size_t h_recs_bytes = 0;     // pinned
        void* h_off = nullptr;  size_t h_off_bytes = 0;      // pinned, offsets rebased to 0
        void* h_counters = nullptr;                          // pinned snapshot of the batch's counters
        hipEvent_t done = nullptr;
        uint32_t n = 0, umi_len = 0; uint64_t total = 0, qcap = 0;
        bool busy = false;
    } slots[BDG_SLOTS];
    // records of every collected chunk, kept on the device in submission order (bdg_extract_keep_records)
    bool keep_records = false;
    DevBuf x_allrecs; uint64_t x_allrecs_n = 0;

    // ---- whitelist index (nearest_kernels.hip)
    DevBuf w_sorted;     // uint32 [nw] ranks ascending
    DevBuf w_orig;       // uint32 [nw] caller index of sorted entry
    DevBuf w_prefix;     // uint32 [2^pbits + 1] offsets by top bits
    DevBuf w_bitmap;     // uint32 [2^bbits / 32] membership of top bbits
    DevBuf w_pent;       // block-pair tables: rank blocks (w_pwords words), then caller-index blocks of the same shape
    size_t w_pwords = 0;
    DevBuf w_delmap;     // 2^30 bits: every 15-mer deletion variant of the whitelist
    DevBuf w_dv;         // the same variants as (variant, sorted-whitelist position) pairs sorted by variant, + directory
    uint32_t w_n = 0;        // 0: no whitelist loaded (set last, after every table of the list is complete)
    uint64_t w_fp = 0;       // fingerprint of the caller's list: the same list again is not rebuilt
    bool w_probe_ready = false;                       // pair tables built (on first use of the probe path)
    bool w_delins_ready = false;                      // deletion-variant maps and entries built (first probe call with max_ed = 2)
    std::vector<uint32_t> w_host_sorted, w_host_order;  // host copy the probe index is built from
    int w_pbits = 0, w_bbits = 0;
    bool w_identity = false;
    int n16_algo = 0;
    DevBuf n_list;       // uint32 [(8 + 1) * nq] level-2 query list (8 segments) + overflow list
    DevBuf n_counters;   // one 128-byte line per list segment + one for the overflow list

    // ---- graph workspace (graph_kernels.hip)
    int graph_algo = 0;
    DevBuf g_sig;        // uint32 [n] letter-count signatures
    DevBuf g_tmp0, g_tmp1, g_cnt;
    DevBuf g_qj;         // q-gram join: sorted (six-mer, row) entries, inverse positions, bucket and slice starts
    uint32_t g_d1_min_rows = 100000;    // thr 1: from this many rows on the one-deletion join runs instead of the neighbourhood probes (BADGER_AMD_D1_MIN_ROWS)
    uint32_t g_d2_min_rows = 10000;     // thr 2: from this many rows on the deletion-variant join runs instead of the q-gram join (BADGER_AMD_D2_MIN_ROWS)
    int g_cus_distinct = 0;             // compute units (bdg_distinct_dev asks by itself when no graph call has)
    uint32_t* g_dj_geom = nullptr;      // deletion-variant joins: the device-side report of the last launch (bdg_graph_status)
    int g_cus = 0, g_qj_per_cu = 0, g_qjw_per_cu = 0, g_qjw_variant = 0;   // compute units and resident blocks per unit of the join kernels (asked once)
};

#define BDG_HIP_TRY(ctx, expr)                                                           \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);              \
            return e_ == hipErrorOutOfMemory ? BDG_E_NOMEM : BDG_E_HIP;                  \
        }                                                                                \
    } while (0)

// Grow-only device buffer.
int bdg_reserve(bdg_ctx* ctx, DevBuf& b, size_t bytes);
const void* bdg_extract_counters_now(const bdg_ctx* ctx);   // the counters of the extraction launched last (extract_kernels.hip)
int bdg_launch_deferred_match(bdg_ctx* ctx, bool behind_scan);   // overlap mode: queue the waiting whitelist match now (bdg_abi.cpp)

// Event-bracketed launch bookkeeping.
int  bdg_timer_id(bdg_ctx* ctx, const char* name);
void bdg_timer_begin(bdg_ctx* ctx, int id);
void bdg_timer_end(bdg_ctx* ctx, int id);

struct ScopedKernelTimer {
    bdg_ctx* ctx; int id;
    ScopedKernelTimer(bdg_ctx* c, const char* name) : ctx(c), id(-1) {
        if (c->profiling && (c->profile_only.empty() || c->profile_only == name)) { id = bdg_timer_id(c, name); bdg_timer_begin(c, id); }
    }
    ~ScopedKernelTimer() { if (id >= 0) bdg_timer_end(ctx, id); }
};

static inline int bdg_fail(bdg_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

// ---- wave-wide data movement through DPP (gfx9 controls; behaviour on gfx950 checked by tools/ubench/dpp_check.hip) ----
#if defined(__HIPCC__)
// lane i <- lane i + 1; lane 63 and lanes whose source is inactive read 0
__device__ __forceinline__ uint32_t wave_shl1(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
}
// lane i <- lane i - 1; lane 0 and lanes whose source is inactive read 0
__device__ __forceinline__ uint32_t wave_shr1(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
}
// inclusive prefix sum over the wave, six VALU instructions (row_shr 1,2,4,8 then row_bcast 15 / 31); all lanes active
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true);
    return x;
}
// inclusive prefix maximum over the wave (values >= 0 as unsigned: an absent source reads 0), same six steps; all lanes active
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x)
{
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true));
    return x;
}
#endif
