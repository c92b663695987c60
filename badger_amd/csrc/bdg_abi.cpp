// C ABI of libbadger_hip.so (include/badger_hip.h): context management, host-buffer
// wrappers (H2D, launch, D2H) and the device-resident entry points.
#include "bdg_common.hpp"
#include "dj_codec.hpp"

#include <cstddef>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <mutex>

// launchers in the kernel translation units
int bdg_extract_launch(bdg_ctx*, const uint8_t*, const uint64_t*, uint32_t, uint64_t, uint32_t, bdg_extract_rec*);
int bdg_extract_status_impl(bdg_ctx*, uint64_t*, uint64_t*);
int bdg_extract_counters_impl(bdg_ctx*, uint64_t*);
int bdg_extract_judge_host(bdg_ctx*, const void*, uint64_t, uint64_t*, uint64_t*);
size_t bdg_extract_counter_bytes();
int bdg_whitelist_load_impl(bdg_ctx*, const uint32_t*, uint32_t);
int bdg_nearest16_launch(bdg_ctx*, const uint32_t*, uint32_t, int, uint32_t, uint32_t, uint32_t*, uint8_t*, uint16_t*);
int bdg_graph_launch(bdg_ctx*, const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t, int32_t, bdg_edge*, uint64_t, uint64_t*, uint32_t part = 0, uint32_t nparts = 1);
int bdg_graph_plan(const bdg_ctx*, uint32_t, uint32_t);
int bdg_graph_join_flags(bdg_ctx*, uint32_t*);
int bdg_graph_flags_error(bdg_ctx*, uint32_t);
int bdg_distinct_launch(bdg_ctx*, const bdg_extract_rec*, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*);
int bdg_records_of_observed_launch(bdg_ctx*, const uint32_t*, const uint8_t*, uint64_t, bdg_extract_rec*);
int bdg_rows_of_launch(bdg_ctx*, const uint32_t*, uint32_t, const uint32_t*, uint64_t, uint32_t, uint32_t*);
int bdg_cluster_launch(bdg_ctx*, const uint32_t*, const uint32_t*, uint64_t, uint32_t, int32_t*);
int bdg_assign_reads_launch(bdg_ctx*, const bdg_extract_rec*, uint64_t, const uint32_t*, uint32_t, const uint32_t*, const uint8_t*, uint32_t*, uint8_t*);
int bdg_touched_count_launch(bdg_ctx*, const uint32_t*, const uint32_t*, uint64_t, uint32_t, const uint32_t*, uint32_t, uint64_t*);

static thread_local std::string g_err_noctx;

int bdg_reserve(bdg_ctx* ctx, DevBuf& b, size_t bytes)
{
    if (bytes <= b.bytes && b.p) return BDG_OK;
    if (bytes < 256) bytes = 256;
    if (b.p) {
        // the buffer may still be in use by queued work
        BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->aux_pending) { BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->aux_stream)); ctx->aux_pending = false; }
        BDG_HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr; b.bytes = 0;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ctx->err = "hipMalloc(" + std::to_string(bytes) + " bytes): " + hipGetErrorString(e);
        return BDG_E_NOMEM;
    }
    b.p = p; b.bytes = bytes;
    return BDG_OK;
}

int bdg_timer_id(bdg_ctx* ctx, const char* name)
{
    for (size_t i = 0; i < ctx->timers.size(); ++i) if (ctx->timers[i].name == name) return (int)i;
    ctx->timers.emplace_back();
    ctx->timers.back().name = name;
    return (int)ctx->timers.size() - 1;
}

static hipEvent_t take_event(bdg_ctx* ctx)
{
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void bdg_timer_begin(bdg_ctx* ctx, int id)
{
    hipEvent_t a = take_event(ctx), b = take_event(ctx);
    (void)hipEventRecord(a, ctx->launch_stream ? ctx->launch_stream : ctx->stream);
    ctx->timers[id].pending.emplace_back(a, b);
}

void bdg_timer_end(bdg_ctx* ctx, int id)
{
    (void)hipEventRecord(ctx->timers[id].pending.back().second, ctx->launch_stream ? ctx->launch_stream : ctx->stream);
    ctx->timers[id].launches++;
}

int bdg_nearest16_launch(bdg_ctx*, const uint32_t*, uint32_t, int, uint32_t, uint32_t, uint32_t*, uint8_t*, uint16_t*);

int bdg_launch_deferred_match(bdg_ctx* ctx, bool behind_scan)
{
    if (!ctx->deferred.pending) return BDG_OK;
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));          // (bdg_synchronize of one context among several devices' contexts)
    const bdg_ctx::DeferredMatch d = ctx->deferred;
    ctx->deferred.pending = false;
    // behind the extraction that wrote the records: one event on the main stream, recorded NOW - behind the next extraction's
    // scan when that has just been queued (which is behind the records' extraction in stream order), else behind whatever
    // the main stream holds so far.  (Recording one when the match was asked for as well put a second marker between two
    // batches: 6 us a step.)
    hipEvent_t after = behind_scan ? ctx->ev_scan : ctx->ev_main;
    BDG_HIP_TRY(ctx, hipEventRecord(after, ctx->stream));
    BDG_HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_stream, after, 0));
    ctx->launch_stream = ctx->aux_stream;
    ctx->aux_pending = true;
    const int rc = bdg_nearest16_launch(ctx, d.q, 8u, 1, d.n, d.max_ed, d.idx, d.ed, d.ties);
    ctx->launch_stream = nullptr;
    BDG_HIP_TRY(ctx, hipEventRecord(ctx->ev_aux[ctx->aux_count & 1], ctx->aux_stream));
    ctx->aux_count++;
    return rc;
}

static int sync_all(bdg_ctx* ctx)
{
    const int rcd = bdg_launch_deferred_match(ctx, false);
    if (rcd) return rcd;
    BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->aux_pending) { BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->aux_stream)); ctx->aux_pending = false; }
    return BDG_OK;
}

static int collect_timers(bdg_ctx* ctx)
{
    int rc0 = sync_all(ctx);
    if (rc0) return rc0;
    for (auto& t : ctx->timers) {
        for (auto& pr : t.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) t.total_ms += ms;
            ctx->event_pool.push_back(pr.first);
            ctx->event_pool.push_back(pr.second);
        }
        t.pending.clear();
    }
    return BDG_OK;
}

extern "C" {

#ifndef BDG_KERNEL_HASH
#define BDG_KERNEL_HASH "unhashed"
#endif
#ifndef BDG_HOST_HASH
#define BDG_HOST_HASH "unhashed"
#endif
const char* bdg_version(void) { return "badger_hip 0.3 (gfx950) kernels " BDG_KERNEL_HASH " host " BDG_HOST_HASH; }

int bdg_selftest_dj_codec(uint64_t seed, uint32_t rounds)
{
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 16); };
    for (uint32_t t = 0, k = 0; t < 16; ++t) for (uint32_t q = t + 1; q < 16; ++q, ++k) if (djc::pair_of(k) != (t << 4 | q)) return 1;
    for (uint32_t it = 0; it < rounds; ++it) {
        const uint32_t r = it == 0 ? 0u : (it == 1 ? 0xFFFFFFFFu : rnd());
        const uint32_t k28 = rnd() & 0x0FFFFFFFu, k30 = rnd() & 0x3FFFFFFFu;
        if (djc::unmix<28>(djc::mix<28>(k28)) != k28 || djc::mix<28>(k28) >> 28) return 2;
        if (djc::unmix<30>(djc::mix<30>(k30)) != k30 || djc::mix<30>(k30) >> 30) return 3;
        if (djc::mix<28>(djc::unmix<28>(k28)) != k28 || djc::mix<30>(djc::unmix<30>(k30)) != k30) return 4;
        for (uint32_t p = 0; p < 16; ++p) {
            const uint32_t k = djc::del1(r, p);
            if (k >> 30 || djc::ins1(k, p, (r >> (2 * p)) & 3u) != r) return 5;
            for (uint32_t l1 = 8; l1 <= 10; ++l1) {
                const uint32_t zb = 30 - l1, z = djc::mix<30>(k), e = djc::enc1(z, zb, p, r);
                uint32_t k2, r2;
                djc::dec1(e, z >> zb, zb, k2, r2);
                if (k2 != k || r2 != r || (e >> (zb + 6))) return 6;
            }
        }
        for (uint32_t t = 0; t < 120; ++t) {
            const uint32_t pq = djc::pair_of(t), p = pq >> 4, q = pq & 15u;
            const uint32_t k = djc::del2(r, p, q);
            if (k >> 28 || djc::ins2(k, p, q, (r >> (2 * p)) & 3u, (r >> (2 * q)) & 3u) != r) return 7;
            if (djc::del1(djc::del1(r, q), p) != k) return 8;
            for (uint32_t l1 = 8; l1 <= 10; ++l1) {
                const uint32_t zb = 28 - l1, z = djc::mix<28>(k), e = djc::enc2(z, zb, t, r, pq);
                uint32_t k2, r2;
                djc::dec2(e, z >> zb, zb, pq, k2, r2);
                if (k2 != k || r2 != r || (e >> (zb + 11))) return 9;
            }
        }
    }
    return 0;
}

int bdg_device_count(void)
{
    int ndev = 0;
    return hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 ? ndev : 0;
}

int bdg_init(int device_id, bdg_ctx** out)
{
    if (!out) return BDG_E_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { g_err_noctx = "no HIP device available"; return BDG_E_HIP; }
    if (device_id < 0 || device_id >= ndev) { g_err_noctx = "device_id out of range"; return BDG_E_ARG; }
    if (hipSetDevice(device_id) != hipSuccess) { g_err_noctx = "hipSetDevice failed"; return BDG_E_HIP; }
    bdg_ctx* ctx = new bdg_ctx();
    ctx->device = device_id;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx; g_err_noctx = "hipStreamCreate failed"; return BDG_E_HIP;
    }
    ctx->stream = ctx->own_stream;
    if (const char* e2 = getenv("BADGER_AMD_D2_MIN_ROWS")) ctx->g_d2_min_rows = (uint32_t)strtoul(e2, nullptr, 10);      // (for measurements)
    if (const char* e1 = getenv("BADGER_AMD_D1_MIN_ROWS")) ctx->g_d1_min_rows = (uint32_t)strtoul(e1, nullptr, 10);
    *out = ctx;
    return BDG_OK;
}

int bdg_mem_alloc(bdg_ctx* ctx, uint64_t bytes, void** d_out)
{
    if (!ctx || !d_out) return BDG_E_ARG;
    *d_out = nullptr;
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); return bdg_fail(ctx, BDG_E_NOMEM, "device allocation failed"); }
    hipError_t e = hipMemsetAsync(p, 0, bytes ? bytes : 1, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(p); return bdg_fail(ctx, BDG_E_HIP, hipGetErrorString(e)); }
    *d_out = p;
    return BDG_OK;
}

int bdg_mem_free(bdg_ctx* ctx, void* d_ptr)
{
    if (!ctx) return BDG_E_ARG;
    if (!d_ptr) return BDG_OK;
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = sync_all(ctx);                                      // work that still uses the buffer, on either stream
    if (rc) return rc;
    BDG_HIP_TRY(ctx, hipFree(d_ptr));
    return BDG_OK;
}

int bdg_mem_to_host(bdg_ctx* ctx, void* dst, const void* d_src, uint64_t bytes)
{
    if (!ctx || (bytes && (!dst || !d_src))) return BDG_E_ARG;
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (bytes) BDG_HIP_TRY(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BDG_OK;
}

int bdg_mem_from_host(bdg_ctx* ctx, void* d_dst, const void* src, uint64_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !src))) return BDG_E_ARG;
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (bytes) BDG_HIP_TRY(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BDG_OK;
}

void bdg_free(bdg_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
    if (ctx->ev_main) (void)hipEventDestroy(ctx->ev_main);
    if (ctx->ev_scan) (void)hipEventDestroy(ctx->ev_scan);
    for (hipEvent_t e : ctx->ev_aux) if (e) (void)hipEventDestroy(e);
    DevBuf* bufs[] = { &ctx->x_lut, &ctx->x_polyt, &ctx->x_keys, &ctx->x_hits, &ctx->x_counters, &ctx->s_in0,
                       &ctx->s_in1, &ctx->s_out0, &ctx->w_sorted, &ctx->w_orig, &ctx->w_prefix, &ctx->w_bitmap, &ctx->w_pent, &ctx->w_delmap, &ctx->w_dv,
                       &ctx->n_list, &ctx->n_counters, &ctx->g_sig, &ctx->g_tmp0, &ctx->g_tmp1, &ctx->g_cnt, &ctx->g_qj, &ctx->x_allrecs };
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    for (auto& sl : ctx->slots) {
        for (DevBuf* b : { &sl.d_bases, &sl.d_off, &sl.d_recs }) if (b->p) (void)hipFree(b->p);
        if (sl.h_recs) (void)hipHostFree(sl.h_recs);
        if (sl.h_off) (void)hipHostFree(sl.h_off);
        if (sl.h_counters) (void)hipHostFree(sl.h_counters);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    for (auto& t : ctx->timers) for (auto& pr : t.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* bdg_last_error(bdg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err_noctx.c_str(); }

int bdg_set_stream(bdg_ctx* ctx, void* hip_stream)
{
    if (!ctx) return BDG_E_ARG;
    int rc = sync_all(ctx);
    if (rc) return rc;
    ctx->stream = static_cast<hipStream_t>(hip_stream);      // NULL is the device's default (null) stream
    return BDG_OK;
}

int bdg_synchronize(bdg_ctx* ctx)
{
    if (!ctx) return BDG_E_ARG;
    return sync_all(ctx);
}

int bdg_set_overlap(bdg_ctx* ctx, int on)
{
    if (!ctx) return BDG_E_ARG;
    int rc = sync_all(ctx);
    if (rc) return rc;
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (on && !ctx->aux_stream) {
        // (same priority as the main stream: measured against the lowest and the highest one, tools/ov_prio_probe.sh)
        BDG_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
        // (device-scope release: these events order kernels of two streams of one device; the default, a release to the
        // system, writes the caches back and kept the next kernel waiting 12 us behind the scan)
        BDG_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming | hipEventReleaseToDevice));
        BDG_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_scan, hipEventDisableTiming | hipEventReleaseToDevice));
        BDG_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_aux[0], hipEventDisableTiming | hipEventReleaseToDevice));
        BDG_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_aux[1], hipEventDisableTiming | hipEventReleaseToDevice));
    }
    ctx->aux_count = 0;
    ctx->overlap = on != 0;
    return BDG_OK;
}

int bdg_profile_enable(bdg_ctx* ctx, int on) { if (!ctx) return BDG_E_ARG; ctx->profiling = on != 0; return BDG_OK; }

int bdg_profile_only(bdg_ctx* ctx, const char* kernel)
{
    if (!ctx) return BDG_E_ARG;
    ctx->profile_only = kernel ? kernel : "";
    return BDG_OK;
}

int bdg_profile_reset(bdg_ctx* ctx)
{
    if (!ctx) return BDG_E_ARG;
    int rc = collect_timers(ctx);
    for (auto& t : ctx->timers) { t.launches = 0; t.total_ms = 0.0; }
    return rc;
}

int bdg_profile_read(bdg_ctx* ctx, bdg_kernel_time* out, int cap)
{
    if (!ctx) return BDG_E_ARG;
    int rc = collect_timers(ctx);
    if (rc) return rc;
    int n = (int)ctx->timers.size();
    for (int i = 0; i < n && i < cap && out; ++i) {
        memset(&out[i], 0, sizeof(out[i]));
        strncpy(out[i].name, ctx->timers[i].name.c_str(), sizeof(out[i].name) - 1);
        out[i].launches = ctx->timers[i].launches;
        out[i].total_ms = ctx->timers[i].total_ms;
    }
    return n;
}

// ---- extraction -----------------------------------------------------------
int bdg_extract_batch_dev(bdg_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_off, uint32_t n,
                          uint64_t total_bytes, uint32_t umi_len, bdg_extract_rec* d_out)
{
    if (!ctx) return BDG_E_ARG;
    if (n && (!d_bases || !d_off || !d_out)) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    if (umi_len == 0 || umi_len > 64) return bdg_fail(ctx, BDG_E_ARG, "umi_len out of range");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // overlap mode: the caller alternates between two record buffers, so this extraction may overwrite what the match before
    // the last one read: stay at most one match ahead
    // (the match of the batch before this one is still waiting - it goes behind this extraction's scan; the one before it is
    // the last one queued)
    if (ctx->overlap && ctx->aux_count >= 1) BDG_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_aux[(ctx->aux_count - 1) & 1], 0));
    return bdg_extract_launch(ctx, d_bases, d_off, n, total_bytes, umi_len, d_out);
}

int bdg_extract_status(bdg_ctx* ctx, uint64_t* bad_read, uint64_t* n_windows)
{
    if (!ctx) return BDG_E_ARG;
    return bdg_extract_status_impl(ctx, bad_read, n_windows);
}

int bdg_extract_set_queue_capacity(bdg_ctx* ctx, uint64_t entries_per_segment)
{
    if (!ctx) return BDG_E_ARG;
    if (entries_per_segment > (1ull << 32)) return bdg_fail(ctx, BDG_E_ARG, "queue capacity too large");
    ctx->x_hits_cap_fixed = entries_per_segment;
    if (entries_per_segment == 0) ctx->x_hits_cap = 0;
    return BDG_OK;
}

int bdg_extract_set_strand_rule(bdg_ctx* ctx, int rule)
{
    if (!ctx) return BDG_E_ARG;
    if (rule != BDG_STRAND_RULE_DEFAULT && rule != BDG_STRAND_RULE_NO_POLYA) return bdg_fail(ctx, BDG_E_ARG, "unknown strand rule");
    ctx->x_strand_rule = rule;
    return BDG_OK;
}

int bdg_extract_counters(bdg_ctx* ctx, uint64_t out[8])
{
    if (!ctx || !out) return BDG_E_ARG;
    return bdg_extract_counters_impl(ctx, out);
}

int bdg_extract_batch(bdg_ctx* ctx, const uint8_t* bases, const uint64_t* off, uint32_t n,
                      uint32_t umi_len, bdg_extract_rec* out)
{
    if (!ctx) return BDG_E_ARG;
    if (n == 0) return BDG_OK;
    if (!bases || !off || !out) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    if (umi_len == 0 || umi_len > 64) return bdg_fail(ctx, BDG_E_ARG, "umi_len out of range");
    for (uint32_t i = 0; i < n; ++i)
        if (off[i + 1] < off[i]) return bdg_fail(ctx, BDG_E_ARG, "offsets must be non-decreasing");
    for (uint32_t i = 0; i < n; ++i)
        if (off[i + 1] - off[i] >= (1ull << 26)) return bdg_fail(ctx, BDG_E_ARG, "read longer than 2^26 bases");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // ship only the byte range the offsets reference, rebased to 0
    const uint64_t lo = off[0], hi = off[n], total = hi - lo;
    int rc;
    if ((rc = bdg_reserve(ctx, ctx->s_in0, total + 64))) return rc;
    if ((rc = bdg_reserve(ctx, ctx->s_in1, sizeof(uint64_t) * ((size_t)n + 1)))) return rc;
    if ((rc = bdg_reserve(ctx, ctx->s_out0, sizeof(bdg_extract_rec) * (size_t)n))) return rc;
    std::vector<uint64_t> rel((size_t)n + 1);
    for (uint32_t i = 0; i <= n; ++i) rel[i] = off[i] - lo;
    hipStream_t st = ctx->stream;
    if (total) BDG_HIP_TRY(ctx, hipMemcpyAsync(ctx->s_in0.p, bases + lo, total, hipMemcpyHostToDevice, st));
    BDG_HIP_TRY(ctx, hipMemcpyAsync(ctx->s_in1.p, rel.data(), sizeof(uint64_t) * rel.size(), hipMemcpyHostToDevice, st));
    // A queue overflow grows the workspace from what the failed pass could count; the hits re-queued by clusters are only
    // known once queue A is complete, so a second overflow is possible: loop (each pass at least 1.5 x the last one).
    for (int attempt = 0; attempt < 8; ++attempt) {
        rc = bdg_extract_launch(ctx, static_cast<const uint8_t*>(ctx->s_in0.p), static_cast<const uint64_t*>(ctx->s_in1.p),
                                n, total, umi_len, static_cast<bdg_extract_rec*>(ctx->s_out0.p));
        if (rc) return rc;
        uint64_t bad = 0, nwin = 0;
        rc = bdg_extract_status_impl(ctx, &bad, &nwin);
        if (rc != BDG_E_CAPACITY) break;
    }
    if (rc) return rc;
    BDG_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->s_out0.p, sizeof(bdg_extract_rec) * (size_t)n, hipMemcpyDeviceToHost, st));
    BDG_HIP_TRY(ctx, hipStreamSynchronize(st));
    return BDG_OK;
}

// ---- pipelined chunks ---------------------------------------------------------
static int pinned_reserve(bdg_ctx* ctx, void*& p, size_t& have, size_t want)
{
    if (want <= have && p) return BDG_OK;
    if (p) { int rc = sync_all(ctx); if (rc) return rc; (void)hipHostFree(p); p = nullptr; have = 0; }
    want += want / 4 + 4096;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; return bdg_fail(ctx, BDG_E_NOMEM, "hipHostMalloc failed"); }
    have = want;
    return BDG_OK;
}

static int slot_enqueue(bdg_ctx* ctx, bdg_ctx::Slot& sl)
{
    int rc = bdg_extract_launch(ctx, static_cast<const uint8_t*>(sl.d_bases.p), static_cast<const uint64_t*>(sl.d_off.p),
                                sl.n, sl.total, sl.umi_len, static_cast<bdg_extract_rec*>(sl.d_recs.p));
    if (rc) return rc;
    sl.qcap = ctx->x_hits_cap_launched;
    hipStream_t st = ctx->stream;
    BDG_HIP_TRY(ctx, hipMemcpyAsync(sl.h_recs, sl.d_recs.p, sizeof(bdg_extract_rec) * (size_t)sl.n, hipMemcpyDeviceToHost, st));
    BDG_HIP_TRY(ctx, hipMemcpyAsync(sl.h_counters, bdg_extract_counters_now(ctx), bdg_extract_counter_bytes(), hipMemcpyDeviceToHost, st));
    BDG_HIP_TRY(ctx, hipEventRecord(sl.done, st));
    return BDG_OK;
}

// where bdg_extract_submit's time goes (BADGER_AMD_INGEST_DEBUG; printed by bdg_stage1_run).  Summed only when that variable is
// set, and atomically: the ABI lets different host threads drive different contexts, and they all pass here.
std::atomic<double> g_submit_t[6];
static const bool g_submit_debug = getenv("BADGER_AMD_INGEST_DEBUG") != nullptr;
static inline void submit_add(int i, double v) { double o = g_submit_t[i].load(std::memory_order_relaxed); while (!g_submit_t[i].compare_exchange_weak(o, o + v, std::memory_order_relaxed)) {} }
static inline double submit_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int bdg_extract_submit(bdg_ctx* ctx, uint32_t slot, const uint8_t* bases, const uint64_t* off, uint32_t n, uint32_t umi_len)
{
    const double T0 = submit_now();
    if (!ctx || slot >= BDG_SLOTS) return BDG_E_ARG;
    bdg_ctx::Slot& sl = ctx->slots[slot];
    if (sl.busy) return bdg_fail(ctx, BDG_E_ARG, "slot still in flight: collect it first");
    if (n && (!bases || !off)) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    if (umi_len == 0 || umi_len > 64) return bdg_fail(ctx, BDG_E_ARG, "umi_len out of range");
    sl.n = n; sl.umi_len = umi_len; sl.total = 0;
    if (n == 0) { sl.busy = true; return BDG_OK; }
    for (uint32_t i = 0; i < n; ++i) {
        if (off[i + 1] < off[i]) return bdg_fail(ctx, BDG_E_ARG, "offsets must be non-decreasing");
        if (off[i + 1] - off[i] >= (1ull << 26)) return bdg_fail(ctx, BDG_E_ARG, "read longer than 2^26 bases");
    }
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t lo = off[0], total = off[n] - lo;
    sl.total = total;
    int rc;
    if ((rc = bdg_reserve(ctx, sl.d_bases, total + 64))) return rc;
    if ((rc = bdg_reserve(ctx, sl.d_off, sizeof(uint64_t) * ((size_t)n + 1)))) return rc;
    if ((rc = bdg_reserve(ctx, sl.d_recs, sizeof(bdg_extract_rec) * (size_t)n))) return rc;
    if ((rc = pinned_reserve(ctx, sl.h_recs, sl.h_recs_bytes, sizeof(bdg_extract_rec) * (size_t)n))) return rc;
    if ((rc = pinned_reserve(ctx, sl.h_off, sl.h_off_bytes, sizeof(uint64_t) * ((size_t)n + 1)))) return rc;
    if (!sl.h_counters) {
        if (hipHostMalloc(&sl.h_counters, bdg_extract_counter_bytes(), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError(); sl.h_counters = nullptr; return bdg_fail(ctx, BDG_E_NOMEM, "hipHostMalloc failed");
        }
    }
    if (!sl.done) BDG_HIP_TRY(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    const double T1 = submit_now();
    uint64_t* rel = static_cast<uint64_t*>(sl.h_off);
    for (uint32_t i = 0; i <= n; ++i) rel[i] = off[i] - lo;
    hipStream_t st = ctx->stream;
    const double T2 = submit_now();
    // (one copy on one stream runs at the link's rate here: 56.6 GB/s for 32 MB from pinned memory, tools/hip_first_calls.py; two
    // halves on two streams, which gained 10 % in round 2, gain nothing any more)
    if (total) BDG_HIP_TRY(ctx, hipMemcpyAsync(sl.d_bases.p, bases + lo, total, hipMemcpyHostToDevice, st));
    BDG_HIP_TRY(ctx, hipMemcpyAsync(sl.d_off.p, rel, sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, st));
    const double T3 = submit_now();
    if ((rc = slot_enqueue(ctx, sl))) return rc;
    sl.busy = true;
    const double T4 = submit_now();
    if (g_submit_debug) { submit_add(0, T1 - T0); submit_add(1, T2 - T1); submit_add(2, T3 - T2); submit_add(3, T4 - T3); submit_add(4, 1.0); }
    return BDG_OK;
}

int bdg_extract_collect(bdg_ctx* ctx, uint32_t slot, bdg_extract_rec* out)
{
    if (!ctx || slot >= BDG_SLOTS) return BDG_E_ARG;
    bdg_ctx::Slot& sl = ctx->slots[slot];
    if (!sl.busy) return bdg_fail(ctx, BDG_E_ARG, "nothing submitted to this slot");
    sl.busy = false;
    if (sl.n == 0) return BDG_OK;
    if (!out) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = BDG_OK;
    for (int attempt = 0; attempt < 8; ++attempt) {
        BDG_HIP_TRY(ctx, hipEventSynchronize(sl.done));
        uint64_t bad = 0, nwin = 0;
        rc = bdg_extract_judge_host(ctx, sl.h_counters, sl.qcap, &bad, &nwin);
        if (rc != BDG_E_CAPACITY) break;
        // this chunk overflowed a queue: run it again (its input is still on the device) behind whatever is queued
        int rc2 = slot_enqueue(ctx, sl);
        if (rc2) return rc2;
    }
    if (rc) return rc;
    memcpy(out, sl.h_recs, sizeof(bdg_extract_rec) * (size_t)sl.n);
    if (ctx->keep_records) {
        // append the chunk's records to the device-side array (grown by copying: earlier chunks stay)
        const size_t have = sizeof(bdg_extract_rec) * (size_t)ctx->x_allrecs_n, add = sizeof(bdg_extract_rec) * (size_t)sl.n;
        if (have + add > ctx->x_allrecs.bytes) {
            DevBuf nb;
            size_t want = (have + add) * 2;
            if (want < (size_t(64) << 20)) want = size_t(64) << 20;
            if ((rc = bdg_reserve(ctx, nb, want))) return rc;
            if (have) BDG_HIP_TRY(ctx, hipMemcpyAsync(nb.p, ctx->x_allrecs.p, have, hipMemcpyDeviceToDevice, ctx->stream));
            BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->x_allrecs.p) (void)hipFree(ctx->x_allrecs.p);
            ctx->x_allrecs = nb;
        }
        BDG_HIP_TRY(ctx, hipMemcpyAsync(static_cast<char*>(ctx->x_allrecs.p) + have, sl.d_recs.p, add, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->x_allrecs_n += sl.n;
    }
    return BDG_OK;
}

int bdg_extract_keep_records(bdg_ctx* ctx, int on)
{
    if (!ctx) return BDG_E_ARG;
    ctx->keep_records = on != 0;
    ctx->x_allrecs_n = 0;
    if (!on && ctx->x_allrecs.p) {
        BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->x_allrecs.p);
        ctx->x_allrecs = DevBuf();
    }
    return BDG_OK;
}

int bdg_keep_observed(bdg_ctx* ctx, const uint32_t* rank, const uint8_t* usable, uint64_t n)
{
    if (!ctx) return BDG_E_ARG;
    if (n && (!rank || !usable)) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    if (n >= (1ull << 32)) return bdg_fail(ctx, BDG_E_ARG, "more than 2^32 - 1 reads");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = bdg_extract_keep_records(ctx, 1);                       // (an empty array; what was kept before is dropped)
    if (rc || n == 0) return rc;
    if ((rc = bdg_reserve(ctx, ctx->x_allrecs, sizeof(bdg_extract_rec) * n))) return rc;
    // the two host arrays through the scratch buffer (pageable memory: the copies return when the data has left it)
    if ((rc = bdg_reserve(ctx, ctx->g_tmp1, 5 * n + 16))) return rc;
    uint32_t* const d_rank = static_cast<uint32_t*>(ctx->g_tmp1.p);
    uint8_t* const d_usable = reinterpret_cast<uint8_t*>(d_rank + n);
    BDG_HIP_TRY(ctx, hipMemcpyAsync(d_rank, rank, 4 * n, hipMemcpyHostToDevice, ctx->stream));
    BDG_HIP_TRY(ctx, hipMemcpyAsync(d_usable, usable, n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = bdg_records_of_observed_launch(ctx, d_rank, d_usable, n, static_cast<bdg_extract_rec*>(ctx->x_allrecs.p)))) return rc;
    BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));             // (the scratch buffer is free for the next user)
    ctx->x_allrecs_n = n;
    return BDG_OK;
}

int bdg_kept_records_to_host(bdg_ctx* ctx, bdg_extract_rec* out, uint64_t cap)
{
    if (!ctx || (cap && !out)) return BDG_E_ARG;
    const uint64_t n = ctx->x_allrecs_n < cap ? ctx->x_allrecs_n : cap;
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n) BDG_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->x_allrecs.p, sizeof(bdg_extract_rec) * n, hipMemcpyDeviceToHost, ctx->stream));
    BDG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BDG_OK;
}

int bdg_kept_records(bdg_ctx* ctx, const bdg_extract_rec** d_recs, uint64_t* n)
{
    if (!ctx || !d_recs || !n) return BDG_E_ARG;
    *d_recs = static_cast<const bdg_extract_rec*>(ctx->x_allrecs.p);
    *n = ctx->x_allrecs_n;
    return BDG_OK;
}

// ---- nearest ----------------------------------------------------------------
int bdg_whitelist_load(bdg_ctx* ctx, const uint32_t* wl, uint32_t nw)
{
    if (!ctx) return BDG_E_ARG;
    if (nw && !wl) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bdg_whitelist_load_impl(ctx, wl, nw);
}

int bdg_nearest16_set_algo(bdg_ctx* ctx, int algo)
{
    if (!ctx || algo < 0 || algo > 2) return BDG_E_ARG;
    ctx->n16_algo = algo;
    return BDG_OK;
}

uint64_t bdg_nearest16_index_bytes(bdg_ctx* ctx)
{
    if (!ctx || ctx->w_n == 0) return 0;
    uint64_t b = 0;
    if (ctx->w_probe_ready) b += ctx->w_pent.bytes;
    if (ctx->w_delins_ready) b += ctx->w_delmap.bytes + ctx->w_dv.bytes;
    return b;
}

int bdg_nearest16_dev(bdg_ctx* ctx, const uint32_t* d_q, uint32_t nq, uint32_t max_ed,
                      uint32_t* d_best_idx, uint8_t* d_best_ed, uint16_t* d_n_ties)
{
    if (!ctx) return BDG_E_ARG;
    if (nq && (!d_q || !d_best_idx || !d_best_ed || !d_n_ties)) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bdg_nearest16_launch(ctx, d_q, 1u, 0, nq, max_ed, d_best_idx, d_best_ed, d_n_ties);
}

int bdg_nearest16_recs_dev(bdg_ctx* ctx, const bdg_extract_rec* d_recs, uint32_t n, uint32_t max_ed,
                           uint32_t* d_best_idx, uint8_t* d_best_ed, uint16_t* d_n_ties)
{
    if (!ctx) return BDG_E_ARG;
    if (n && (!d_recs || !d_best_idx || !d_best_ed || !d_n_ties)) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    static_assert(sizeof(bdg_extract_rec) == 32 && offsetof(bdg_extract_rec, bc_rank) == 20 && offsetof(bdg_extract_rec, flags) == 27,
                  "record layout the strided query reads");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->overlap) {
        // not queued yet: it goes behind the next extraction's scan (or behind everything queued so far, at the next
        // synchronisation, whitelist change or match)
        int rc = bdg_launch_deferred_match(ctx, false);
        if (rc) return rc;
        if (ctx->w_n == 0) return bdg_fail(ctx, BDG_E_ARG, "no whitelist loaded (bdg_whitelist_load)");
        ctx->deferred.pending = true;
        ctx->deferred.q = reinterpret_cast<const uint32_t*>(d_recs) + 5; ctx->deferred.n = n; ctx->deferred.max_ed = max_ed;
        ctx->deferred.idx = d_best_idx; ctx->deferred.ed = d_best_ed; ctx->deferred.ties = d_n_ties;
        return BDG_OK;
    }
    return bdg_nearest16_launch(ctx, reinterpret_cast<const uint32_t*>(d_recs) + 5, 8u, 1, n, max_ed, d_best_idx, d_best_ed, d_n_ties);
}

int bdg_nearest16(bdg_ctx* ctx, const uint32_t* q, uint32_t nq, const uint32_t* wl, uint32_t nw,
                  uint32_t max_ed, uint32_t* best_idx, uint8_t* best_ed, uint16_t* n_ties)
{
    if (!ctx) return BDG_E_ARG;
    if (nq == 0) return BDG_OK;
    if (!q || !best_idx || !best_ed || !n_ties || (nw && !wl)) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (nw == 0) {
        for (uint32_t i = 0; i < nq; ++i) { best_idx[i] = 0xFFFFFFFFu; best_ed[i] = 0xFF; n_ties[i] = 0; }
        return BDG_OK;
    }
    int rc = bdg_whitelist_load_impl(ctx, wl, nw);
    if (rc) return rc;
    const size_t bq = sizeof(uint32_t) * (size_t)nq;
    if ((rc = bdg_reserve(ctx, ctx->s_in0, bq))) return rc;
    if ((rc = bdg_reserve(ctx, ctx->s_out0, bq * 2 + 64))) return rc;     // idx u32 | ties u16 | ed u8
    hipStream_t st = ctx->stream;
    auto* d_idx = static_cast<uint32_t*>(ctx->s_out0.p);
    auto* d_ties = reinterpret_cast<uint16_t*>(d_idx + nq);
    auto* d_ed = reinterpret_cast<uint8_t*>(d_ties + nq);
    BDG_HIP_TRY(ctx, hipMemcpyAsync(ctx->s_in0.p, q, bq, hipMemcpyHostToDevice, st));
    rc = bdg_nearest16_launch(ctx, static_cast<const uint32_t*>(ctx->s_in0.p), 1u, 0, nq, max_ed, d_idx, d_ed, d_ties);
    if (rc) return rc;
    BDG_HIP_TRY(ctx, hipMemcpyAsync(best_idx, d_idx, bq, hipMemcpyDeviceToHost, st));
    BDG_HIP_TRY(ctx, hipMemcpyAsync(n_ties, d_ties, sizeof(uint16_t) * (size_t)nq, hipMemcpyDeviceToHost, st));
    BDG_HIP_TRY(ctx, hipMemcpyAsync(best_ed, d_ed, (size_t)nq, hipMemcpyDeviceToHost, st));
    BDG_HIP_TRY(ctx, hipStreamSynchronize(st));
    return BDG_OK;
}

// ---- graph --------------------------------------------------------------------
int bdg_graph_set_algo(bdg_ctx* ctx, int algo)
{
    if (!ctx || algo < 0 || algo > 6) return BDG_E_ARG;
    ctx->graph_algo = algo;
    return BDG_OK;
}

int bdg_graph_status(bdg_ctx* ctx)
{
    if (!ctx) return BDG_E_ARG;
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t flags = 0;
    int rc;
    if ((rc = bdg_graph_join_flags(ctx, &flags))) return rc;
    return flags ? bdg_graph_flags_error(ctx, flags) : BDG_OK;
}

int bdg_graph_edges_dev(bdg_ctx* ctx, const uint32_t* d_ranks, uint32_t n, uint32_t thr, int32_t qgram_T,
                        bdg_edge* d_out, uint64_t cap, uint64_t* d_n_edges)
{
    if (!ctx) return BDG_E_ARG;
    if (!d_n_edges || (n && !d_ranks) || (cap && !d_out)) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bdg_graph_launch(ctx, d_ranks, n, 0u, n, thr, qgram_T, d_out, cap, d_n_edges);
}

int bdg_graph_edges_rows_dev(bdg_ctx* ctx, const uint32_t* d_ranks, uint32_t n, uint32_t row_begin, uint32_t row_end,
                             uint32_t thr, int32_t qgram_T, bdg_edge* d_out, uint64_t cap, uint64_t* d_n_edges)
{
    if (!ctx) return BDG_E_ARG;
    if (!d_n_edges || (n && !d_ranks) || (cap && !d_out)) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    if (row_begin > row_end || row_end > n) return bdg_fail(ctx, BDG_E_ARG, "row block outside [0, n]");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bdg_graph_launch(ctx, d_ranks, n, row_begin, row_end, thr, qgram_T, d_out, cap, d_n_edges);
}

int bdg_graph_edges_part_dev(bdg_ctx* ctx, const uint32_t* d_ranks, uint32_t n, uint32_t part, uint32_t nparts,
                             uint32_t thr, int32_t qgram_T, bdg_edge* d_out, uint64_t cap, uint64_t* d_n_edges)
{
    if (!ctx) return BDG_E_ARG;
    if (!d_n_edges || (n && !d_ranks) || (cap && !d_out)) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    if (nparts == 0 || part >= nparts) return bdg_fail(ctx, BDG_E_ARG, "part outside [0, nparts)");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int plan = bdg_graph_plan(ctx, n, thr);
    if (plan == 5 || plan == 6) return bdg_graph_launch(ctx, d_ranks, n, 0u, n, thr, qgram_T, d_out, cap, d_n_edges, part, nparts);
    // the other paths share by blocks of rows: equal rows where a row's work is constant (neighbourhood probes), equal numbers
    // of (i, j > i) pairs where row i meets what lies behind it (q-gram join, sweep): cuts at n (1 - sqrt(1 - g / nparts))
    auto cut = [&](uint32_t g) -> uint32_t {
        if (g >= nparts) return n;
        if (plan == 2) return (uint32_t)((unsigned long long)n * g / nparts);
        const double c = (double)n * (1.0 - std::sqrt(1.0 - (double)g / (double)nparts));
        return c <= 0.0 ? 0u : (c >= (double)n ? n : (uint32_t)(c + 0.5));
    };
    uint32_t lo = cut(part), hi = cut(part + 1);
    if (hi < lo) hi = lo;
    return bdg_graph_launch(ctx, d_ranks, n, lo, hi, thr, qgram_T, d_out, cap, d_n_edges);
}

int bdg_graph_edges(bdg_ctx* ctx, const uint32_t* ranks, uint32_t n, uint32_t thr, int32_t qgram_T,
                    bdg_edge* out, uint64_t cap, uint64_t* n_edges)
{
    if (!ctx) return BDG_E_ARG;
    if (!n_edges || (n && !ranks) || (cap && !out)) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    *n_edges = 0;
    if (n < 2) return BDG_OK;
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> srt(ranks, ranks + n);
    std::sort(srt.begin(), srt.end());
    for (uint32_t i = 1; i < n; ++i)
        if (srt[i] == srt[i - 1]) return bdg_fail(ctx, BDG_E_ARG, "ranks must be distinct");
    int rc;
    hipStream_t st = ctx->stream;
    if ((rc = bdg_reserve(ctx, ctx->g_tmp0, sizeof(uint32_t) * (size_t)n))) return rc;
    if ((rc = bdg_reserve(ctx, ctx->g_cnt, 64))) return rc;
    BDG_HIP_TRY(ctx, hipMemcpyAsync(ctx->g_tmp0.p, srt.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, st));
    uint64_t dcap = std::max<uint64_t>(cap, 4ull * n + 1024);
    std::vector<bdg_edge> all;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if ((rc = bdg_reserve(ctx, ctx->g_tmp1, sizeof(bdg_edge) * dcap))) return rc;
        rc = bdg_graph_launch(ctx, static_cast<const uint32_t*>(ctx->g_tmp0.p), n, 0u, n, thr, qgram_T,
                              static_cast<bdg_edge*>(ctx->g_tmp1.p), dcap, static_cast<uint64_t*>(ctx->g_cnt.p));
        if (rc) return rc;
        uint64_t total = 0;
        BDG_HIP_TRY(ctx, hipMemcpyAsync(&total, ctx->g_cnt.p, 8, hipMemcpyDeviceToHost, st));
        BDG_HIP_TRY(ctx, hipStreamSynchronize(st));
        if ((rc = bdg_graph_status(ctx))) return rc;
        if (total > dcap) { dcap = total; continue; }
        all.resize(total);
        if (total) BDG_HIP_TRY(ctx, hipMemcpy(all.data(), ctx->g_tmp1.p, sizeof(bdg_edge) * total, hipMemcpyDeviceToHost));
        *n_edges = total;
        break;
    }
    std::sort(all.begin(), all.end(), [](const bdg_edge& x, const bdg_edge& y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
    const uint64_t w = std::min<uint64_t>(cap, all.size());
    if (w) memcpy(out, all.data(), sizeof(bdg_edge) * w);
    if (all.size() > cap) return bdg_fail(ctx, BDG_E_CAPACITY, "edge capacity too small");
    return BDG_OK;
}

int bdg_distinct_dev(bdg_ctx* ctx, const bdg_extract_rec* d_recs, uint32_t n,
                     uint32_t* d_uniq, uint32_t* d_count, uint32_t* d_first, uint32_t* d_n)
{
    if (!ctx) return BDG_E_ARG;
    if (!d_n || (n && (!d_recs || !d_uniq || !d_count || !d_first))) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bdg_distinct_launch(ctx, d_recs, n, d_uniq, d_count, d_first, d_n);
}

int bdg_rows_of_dev(bdg_ctx* ctx, const uint32_t* d_sorted, uint32_t n, const uint32_t* d_values, uint64_t m,
                    uint32_t stride_words, uint32_t* d_rows)
{
    if (!ctx) return BDG_E_ARG;
    if (m && (!d_values || !d_rows || (n && !d_sorted) || stride_words == 0)) return bdg_fail(ctx, BDG_E_ARG, "null pointer or zero stride");
    if (m > (1ull << 39)) return bdg_fail(ctx, BDG_E_ARG, "too many values");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bdg_rows_of_launch(ctx, d_sorted, n, d_values, m, stride_words, d_rows);
}

int bdg_cluster_dev(bdg_ctx* ctx, const uint32_t* d_ea, const uint32_t* d_eb, uint64_t m, uint32_t nu, int32_t* d_owner)
{
    if (!ctx) return BDG_E_ARG;
    if ((nu && !d_owner) || (m && (!d_ea || !d_eb))) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bdg_cluster_launch(ctx, d_ea, d_eb, m, nu, d_owner);
}

int bdg_assign_reads_dev(bdg_ctx* ctx, const bdg_extract_rec* d_recs, uint64_t n, const uint32_t* d_uniq, uint32_t nu,
                         const uint32_t* d_assigned, const uint8_t* d_has, uint32_t* d_out_rank, uint8_t* d_out_has)
{
    if (!ctx) return BDG_E_ARG;
    if (n && (!d_recs || !d_out_rank || !d_out_has || (nu && (!d_uniq || !d_assigned || !d_has)))) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bdg_assign_reads_launch(ctx, d_recs, n, d_uniq, nu, d_assigned, d_has, d_out_rank, d_out_has);
}

int bdg_touched_count_dev(bdg_ctx* ctx, const uint32_t* d_ea, const uint32_t* d_eb, uint64_t m, uint32_t nu,
                          const uint32_t* d_extra, uint32_t n_extra, uint64_t* count)
{
    if (!ctx) return BDG_E_ARG;
    if (!count || (m && (!d_ea || !d_eb)) || (n_extra && !d_extra)) return bdg_fail(ctx, BDG_E_ARG, "null pointer");
    BDG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return bdg_touched_count_launch(ctx, d_ea, d_eb, m, nu, d_extra, n_extra, count);
}

}  // extern "C"
