// Stage 1 as one native pipeline, and its row formatter:
//   bdg_format_rows  one TSV row per read from the device's 32-byte records (TenXBarcodeDetectionResult.__str__,
//                    barcode_callers.py:40-42,91-93,117-119); the barcode / UMI text is sliced from the chunk's bases, for
//                    reverse-strand results from the reverse complement (barcode_extraction/common.py:34-39).
//   bdg_stage1_run   input file -> TSV, everything between in native threads: the readers of ingest.cpp fill pinned chunks,
//                    this thread submits them to the GPU(s) (bdg_extract_submit / collect, chunk k on context k mod N, two
//                    in flight per context), a few formatter threads turn records into rows and a writer thread writes them
//                    in input order.  What the reference spreads over a ProcessPoolExecutor, temporary files and a final
//                    concatenation (extract_raw_barcodes.py:176-261) - with the two file shapes it produces:
//                    one header on top (process_single_thread, :162-173), or a header in front of every READ_CHUNK_SIZE
//                    reads plus one for the trailing, possibly empty chunk (process_in_parallel, :131-159,243-246).
#include "bdg_common.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

namespace {

inline char comp_base(char c)
{
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; }   // N -> N
}

inline char* put_int(char* o, int v)
{
    char t[16]; int k = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { t[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *o++ = '-';
    while (k) *o++ = t[--k];
    return o;
}

struct RowStats { uint64_t reads = 0, bc = 0, pt = 0, r1 = 0, first_pt = ~0ull, first_r1 = ~0ull; };

// upper bound of the text of a chunk's rows (+ the headers that fall inside it)
uint64_t rows_bound(const bdg_ingest_chunk* ch, const bdg_extract_rec* recs, uint64_t g0, uint32_t header_every, size_t header_len)
{
    uint64_t need = 0;
    for (uint32_t i = 0; i < ch->n; ++i) {
        const uint64_t L = ch->off[i + 1] - ch->off[i];
        need += (ch->id_off[i + 1] - ch->id_off[i]) + 64 + (recs[i].valid ? 16 + std::min<uint64_t>(L, (uint64_t)std::max(0, recs[i].umi_end - recs[i].umi_start)) : 2);
    }
    if (header_every) need += (ch->n / header_every + 2) * (header_len + 1);
    (void)g0;
    return need;
}

// rows of a chunk whose first read is read g0 of the input; header_every > 0: the header line goes in front of every read
// whose index is a multiple of it
char* write_rows(const bdg_ingest_chunk* ch, const bdg_extract_rec* recs, char* o, uint64_t g0, uint32_t header_every,
                 const char* header, size_t header_len, RowStats& st)
{
    constexpr uint32_t AHEAD = 12;              // a row needs one or two lines of its read's bases, nowhere near the last row's: ask early
    for (uint32_t i = 0; i < ch->n; ++i) {
        if (header_every && (g0 + i) % header_every == 0) { memcpy(o, header, header_len); o += header_len; *o++ = '\n'; }
        if (i + AHEAD < ch->n) {
            const bdg_extract_rec& f = recs[i + AHEAD];
            if (f.valid) {
                const uint64_t a = ch->off[i + AHEAD], b = ch->off[i + AHEAD + 1];
                const uint8_t* q = (f.flags & BDG_FLAG_REV) ? ch->bases + b - 1 - (uint64_t)std::min<int64_t>(f.umi_end, (int64_t)(b - a)) : ch->bases + a + (uint64_t)std::max(f.bc_start, 0);
                __builtin_prefetch(q); __builtin_prefetch(q + 40);
            }
        }
        const bdg_extract_rec& r = recs[i];
        const uint8_t* seq = ch->bases + ch->off[i];
        const int64_t L = (int64_t)(ch->off[i + 1] - ch->off[i]);
        const size_t idl = (size_t)(ch->id_off[i + 1] - ch->id_off[i]);
        memcpy(o, ch->ids + ch->id_off[i], idl); o += idl;
        *o++ = '\t';
        const bool rev = (r.flags & BDG_FLAG_REV) != 0;
        auto slice = [&](int64_t a, int64_t b) {                       // Python slice s[a:b] of the strand's text (a, b >= 0)
            a = std::min<int64_t>(std::max<int64_t>(a, 0), L); b = std::min<int64_t>(std::max<int64_t>(b, 0), L);
            if (rev) for (int64_t x = a; x < b; ++x) *o++ = comp_base((char)seq[L - 1 - x]);
            else if (b > a) { memcpy(o, seq + a, (size_t)(b - a)); o += b - a; }
        };
        if (r.valid) {
            slice(r.bc_start, (int64_t)r.bc_start + 16); *o++ = '\t';
            slice(r.umi_start, r.umi_end);
            memcpy(o, "\t0\tFalse\t", 9); o += 9;
            ++st.bc;
        } else {
            memcpy(o, "*\t*\t-1\tFalse\t", 13); o += 13;
        }
        *o++ = r.strand > 0 ? '+' : (r.strand < 0 ? '-' : '.');
        *o++ = '\t';
        o = put_int(o, r.polyT); *o++ = '\t';
        o = put_int(o, r.valid ? r.r1_end : -1);
        *o++ = '\n';
        if (r.polyT != -1) { ++st.pt; if (st.first_pt == ~0ull) st.first_pt = g0 + i; }
        if (r.valid && r.r1_end != -1) { ++st.r1; if (st.first_r1 == ~0ull) st.first_r1 = g0 + i; }
    }
    st.reads += ch->n;
    return o;
}

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Job {
    uint64_t seq = 0, g0 = 0;
    bdg_ingest_chunk ch;
    bdg_ctx* ctx = nullptr; uint32_t slot = 0;
    std::vector<bdg_extract_rec> recs;
    std::vector<char> text; size_t text_len = 0;
    RowStats st;
};

struct Pipeline {
    bdg_ingest* ing = nullptr;
    int fd = -1;
    std::string header;
    uint32_t header_every = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job*> to_format;
    std::map<uint64_t, Job*> formatted;
    uint64_t next_write = 0, outstanding = 0;
    bool closing = false, write_failed = false;
    RowStats total;
    double t_format = 0, t_write = 0;
    uint64_t out_bytes = 0;

    void format_loop()
    {
        for (;;) {
            Job* j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return closing || !to_format.empty(); });
                if (to_format.empty()) return;
                j = to_format.front(); to_format.pop_front();
            }
            const double t0 = now_s();
            j->text.resize((size_t)rows_bound(&j->ch, j->recs.data(), j->g0, header_every, header.size()));
            char* e = write_rows(&j->ch, j->recs.data(), j->text.data(), j->g0, header_every, header.data(), header.size(), j->st);
            j->text_len = (size_t)(e - j->text.data());
            bdg_ingest_release(ing, j->ch.id);
            std::vector<bdg_extract_rec>().swap(j->recs);
            const double dt = now_s() - t0;
            {
                std::lock_guard<std::mutex> lk(mu);
                formatted[j->seq] = j;
                --outstanding;
                t_format += dt;
            }
            cv.notify_all();
        }
    }
    void write_loop()
    {
        for (;;) {
            Job* j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return (closing && to_format.empty() && outstanding == 0 && formatted.empty()) || formatted.count(next_write); });
                auto it = formatted.find(next_write);
                if (it == formatted.end()) return;
                j = it->second; formatted.erase(it); ++next_write;
            }
            const double t0 = now_s();
            const char* p = j->text.data(); size_t left = j->text_len;
            bool bad = false;
            while (left) {
                const ssize_t w = ::write(fd, p, left);
                if (w < 0) { if (errno == EINTR) continue; bad = true; break; }
                p += w; left -= (size_t)w;
            }
            const double dt = now_s() - t0;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (bad) write_failed = true;
                total.reads += j->st.reads; total.bc += j->st.bc; total.pt += j->st.pt; total.r1 += j->st.r1;
                total.first_pt = std::min(total.first_pt, j->st.first_pt); total.first_r1 = std::min(total.first_r1, j->st.first_r1);
                t_write += dt; out_bytes += j->text_len;
            }
            delete j;
        }
    }
};

bool write_all(int fd, const char* p, size_t n)
{
    while (n) {
        const ssize_t w = ::write(fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        p += w; n -= (size_t)w;
    }
    return true;
}

}  // namespace

extern "C" {

int64_t bdg_format_rows(const bdg_ingest_chunk* ch, const bdg_extract_rec* recs, char* out, uint64_t cap, uint64_t counts[4])
{
    if (!ch || (ch->n && (!recs || !ch->bases || !ch->off || !ch->ids || !ch->id_off))) return BDG_E_ARG;
    const uint64_t need = rows_bound(ch, recs, 0, 0, 0);
    if (!out || need > cap) return (int64_t)need;
    RowStats st;
    char* e = write_rows(ch, recs, out, 0, 0, nullptr, 0, st);
    if (counts) { counts[0] = ch->n; counts[1] = st.bc; counts[2] = st.pt; counts[3] = st.r1; }
    return (int64_t)(e - out);
}

int bdg_stage1_run(bdg_ctx* const* ctxs, uint32_t n_ctx, const char* in_path, const char* out_path, const char* header,
                   const bdg_stage1_opts* o, bdg_stage1_result* res)
{
    if (!ctxs || n_ctx == 0 || !ctxs[0] || !in_path || !out_path || !header || !o || !res) return BDG_E_ARG;
    bdg_ctx* const c0 = ctxs[0];
    memset(res, 0, sizeof(*res));
    res->first_polyt = res->first_r1 = res->bad_read = ~0ull;
    if (o->umi_len == 0 || o->umi_len > 64) return bdg_fail(c0, BDG_E_ARG, "umi_len out of range");
    const double t_start = now_s();
    uint32_t fthreads = o->format_threads ? std::min(o->format_threads, 32u) : 4u;
    if (!o->format_threads) if (const char* e = getenv("BADGER_AMD_FORMAT_THREADS")) { const long v = atol(e); if (v > 0 && v <= 32) fthreads = (uint32_t)v; }
    uint32_t per_ctx = 2;                                        // chunks in flight per context (BDG_SLOTS >= 2)
    if (const char* e = getenv("BADGER_AMD_INFLIGHT")) { const long v = atol(e); if (v >= 1 && v <= BDG_SLOTS) per_ctx = (uint32_t)v; }
    bdg_ingest_opts io;
    memset(&io, 0, sizeof(io));
    io.chunk_reads = o->chunk_reads ? o->chunk_reads : 100000u;
    io.ring_chunks = per_ctx * n_ctx + 2 * fthreads + 4;         // views this pipeline holds at once
    io.pinned = 1; io.threads = o->threads; io.segment_bytes = o->segment_bytes; io.skip_secondary = o->skip_secondary;
    const uint64_t max_outstanding = 2 * fthreads + 2;           // collected chunks waiting for / in the formatters
    Pipeline P;
    int rc = bdg_ingest_open_ex(in_path, &io, &P.ing);
    if (rc) return bdg_fail(c0, rc, std::string("cannot read ") + in_path + " (unknown extension or unreadable file)");
    P.fd = ::open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (P.fd < 0) { bdg_ingest_close(P.ing); return bdg_fail(c0, BDG_E_ARG, std::string("cannot write ") + out_path); }
    P.header = header; P.header_every = o->header_every;
    bool ok_io = true;
    if (!o->header_every) ok_io = write_all(P.fd, (P.header + "\n").data(), P.header.size() + 1);
    std::vector<std::thread> fmt;
    for (uint32_t i = 0; i < fthreads; ++i) fmt.emplace_back(&Pipeline::format_loop, &P);
    std::thread writer(&Pipeline::write_loop, &P);

    std::deque<Job*> inflight;
    double t_parse_wait = 0, t_gpu_wait = 0, t_fmt_wait = 0, t_submit = 0;
    std::string err; uint64_t bad_read = ~0ull;
    auto collect = [&](Job* j) -> int {
        j->recs.resize(j->ch.n);
        const double t0 = now_s();
        int r = bdg_extract_collect(j->ctx, j->slot, j->recs.data());
        t_gpu_wait += now_s() - t0;
        if (r) {
            err = bdg_last_error(j->ctx);
            if (r == BDG_E_BADBASE) { uint64_t b = ~0ull, w = 0; (void)bdg_extract_status(j->ctx, &b, &w); if (b != ~0ull) bad_read = j->g0 + b; }
            bdg_ingest_release(P.ing, j->ch.id);
            delete j;
            return r;
        }
        const double t1 = now_s();
        {
            std::unique_lock<std::mutex> lk(P.mu);
            P.cv.wait(lk, [&] { return P.outstanding < max_outstanding; });
            ++P.outstanding;
            P.to_format.push_back(j);
        }
        P.cv.notify_all();
        t_fmt_wait += now_s() - t1;
        return BDG_OK;
    };
    uint64_t k = 0, g0 = 0;
    while (rc == BDG_OK) {
        bdg_ingest_chunk ch;
        const double t0 = now_s();
        rc = bdg_ingest_next(P.ing, &ch);
        t_parse_wait += now_s() - t0;
        if (rc) { err = bdg_ingest_error(P.ing); break; }
        if (ch.n == 0) break;
        if (inflight.size() >= (size_t)per_ctx * n_ctx) { Job* j = inflight.front(); inflight.pop_front(); if ((rc = collect(j))) { bdg_ingest_release(P.ing, ch.id); break; } }
        Job* j = new Job();
        j->seq = k; j->g0 = g0; j->ch = ch; j->ctx = ctxs[k % n_ctx]; j->slot = (uint32_t)((k / n_ctx) % per_ctx);
        const double t1 = now_s();
        rc = bdg_extract_submit(j->ctx, j->slot, ch.bases, ch.off, ch.n, o->umi_len);
        t_submit += now_s() - t1;
        if (rc) { err = bdg_last_error(j->ctx); bdg_ingest_release(P.ing, ch.id); delete j; break; }
        inflight.push_back(j);
        g0 += ch.n; ++k;
    }
    while (!inflight.empty()) {
        Job* j = inflight.front(); inflight.pop_front();
        if (rc == BDG_OK) rc = collect(j);
        else { j->recs.resize(j->ch.n); (void)bdg_extract_collect(j->ctx, j->slot, j->recs.data()); bdg_ingest_release(P.ing, j->ch.id); delete j; }   // wait for the GPU before the pinned buffers go
    }
    { std::lock_guard<std::mutex> lk(P.mu); P.closing = true; }
    P.cv.notify_all();
    for (auto& t : fmt) t.join();
    writer.join();
    // rows of the chunks before a failure are in the file, like in the reference's loop
    if (rc == BDG_OK && o->header_every && g0 % o->header_every == 0) ok_io = write_all(P.fd, (P.header + "\n").data(), P.header.size() + 1) && ok_io;
    if (::close(P.fd) != 0) ok_io = false;
    const double t_close0 = now_s();
    bdg_ingest_close(P.ing);
    if (getenv("BADGER_AMD_INGEST_DEBUG")) {
        extern std::atomic<double> g_submit_t[6];
        fprintf(stderr, "stage1: reader closed in %.3f s; %d submits: reserve %.3f s, offsets %.3f s, copies %.3f s, launches + D2H %.3f s\n", now_s() - t_close0,
                (int)g_submit_t[4].load(), g_submit_t[0].load(), g_submit_t[1].load(), g_submit_t[2].load(), g_submit_t[3].load());
    }
    res->reads = P.total.reads; res->barcodes = P.total.bc; res->polyt = P.total.pt; res->r1 = P.total.r1;
    res->first_polyt = P.total.first_pt; res->first_r1 = P.total.first_r1; res->bad_read = bad_read;
    res->chunks = k; res->out_bytes = P.out_bytes;
    res->seconds_total = now_s() - t_start; res->seconds_wait_parse = t_parse_wait; res->seconds_wait_gpu = t_gpu_wait;
    res->seconds_wait_format = t_fmt_wait; res->seconds_submit = t_submit; res->seconds_format = P.t_format; res->seconds_write = P.t_write;
    if (rc) return bdg_fail(c0, rc, err);
    if (!ok_io || P.write_failed) return bdg_fail(c0, BDG_E_ARG, std::string("write error on ") + out_path);
    return BDG_OK;
}

// ---- stage 2's read-side plumbing ------------------------------------------------------------------------------------
}  // extern "C"

struct bdg_idstore {
    std::vector<char> text;
    std::vector<uint64_t> off{ 0 };
};

extern "C" {

bdg_idstore* bdg_idstore_new(void) { return new bdg_idstore(); }
void bdg_idstore_free(bdg_idstore* s) { delete s; }
uint64_t bdg_idstore_count(const bdg_idstore* s) { return s ? s->off.size() - 1 : 0; }

int bdg_idstore_append(bdg_idstore* s, const char* ids, const uint64_t* off, uint64_t n)
{
    if (!s || (n && (!ids || !off))) return BDG_E_ARG;
    if (!n) return BDG_OK;
    const uint64_t lo = off[0], bytes = off[n] - lo, base = s->text.size();
    s->text.insert(s->text.end(), ids + lo, ids + lo + bytes);
    if (s->off.capacity() < s->off.size() + n) s->off.reserve(std::max<size_t>(s->off.size() + n, 2 * s->off.capacity()));   // (never to the exact size: appends come one id at a time, too)
    for (uint64_t i = 1; i <= n; ++i) s->off.push_back(base + (off[i] - lo));
    return BDG_OK;
}

int bdg_idstore_get(const bdg_idstore* s, uint64_t i, const char** p, uint32_t* len)
{
    if (!s || !p || !len || i + 1 >= s->off.size()) return BDG_E_ARG;
    *p = s->text.data() + s->off[i]; *len = (uint32_t)(s->off[i + 1] - s->off[i]);
    return BDG_OK;
}

int bdg_stage1_collect(bdg_ctx* ctx, const char* in_path, const bdg_stage1_opts* o, bdg_idstore* ids, bdg_stage1_result* res)
{
    if (!ctx || !in_path || !o || !ids || !res) return BDG_E_ARG;
    memset(res, 0, sizeof(*res));
    res->first_polyt = res->first_r1 = res->bad_read = ~0ull;
    if (o->umi_len == 0 || o->umi_len > 64) return bdg_fail(ctx, BDG_E_ARG, "umi_len out of range");
    const double t_start = now_s();
    bdg_ingest_opts io;
    memset(&io, 0, sizeof(io));
    io.chunk_reads = o->chunk_reads ? o->chunk_reads : 100000u;
    io.ring_chunks = 4; io.pinned = 1; io.threads = o->threads; io.segment_bytes = o->segment_bytes; io.skip_secondary = o->skip_secondary;
    bdg_ingest* ing = nullptr;
    int rc = bdg_ingest_open_ex(in_path, &io, &ing);
    if (rc) return bdg_fail(ctx, rc, std::string("cannot read ") + in_path + " (unknown extension or unreadable file)");
    struct Fly { bdg_ingest_chunk ch; uint32_t slot; uint64_t g0; };
    std::deque<Fly> inflight;
    std::vector<bdg_extract_rec> recs;
    std::string err; uint64_t bad_read = ~0ull, k = 0, g0 = 0;
    auto collect = [&](const Fly& f) -> int {
        recs.resize(f.ch.n);
        const double t0 = now_s();
        int r = bdg_extract_collect(ctx, f.slot, recs.data());
        res->seconds_wait_gpu += now_s() - t0;
        if (r) {
            err = bdg_last_error(ctx);
            if (r == BDG_E_BADBASE) { uint64_t b = ~0ull, w = 0; (void)bdg_extract_status(ctx, &b, &w); if (b != ~0ull) bad_read = f.g0 + b; }
        }
        bdg_ingest_release(ing, f.ch.id);
        return r;
    };
    while (rc == BDG_OK) {
        bdg_ingest_chunk ch;
        const double t0 = now_s();
        rc = bdg_ingest_next(ing, &ch);
        res->seconds_wait_parse += now_s() - t0;
        if (rc) { err = bdg_ingest_error(ing); break; }
        if (ch.n == 0) break;
        if (inflight.size() >= 2) { const Fly f = inflight.front(); inflight.pop_front(); if ((rc = collect(f))) { bdg_ingest_release(ing, ch.id); break; } }
        const double t1 = now_s();
        (void)bdg_idstore_append(ids, ch.ids, ch.id_off, ch.n);
        const uint32_t slot = (uint32_t)(k % 2);
        rc = bdg_extract_submit(ctx, slot, ch.bases, ch.off, ch.n, o->umi_len);
        res->seconds_submit += now_s() - t1;
        if (rc) { err = bdg_last_error(ctx); bdg_ingest_release(ing, ch.id); break; }
        inflight.push_back(Fly{ ch, slot, g0 });
        g0 += ch.n; ++k;
    }
    while (!inflight.empty()) {
        const Fly f = inflight.front(); inflight.pop_front();
        const int r = collect(f);                              // (after a failure: still wait for the GPU before the pinned buffers go)
        if (rc == BDG_OK) rc = r;
    }
    bdg_ingest_close(ing);
    res->reads = g0; res->chunks = k; res->bad_read = bad_read; res->seconds_total = now_s() - t_start;
    if (rc) return bdg_fail(ctx, rc, err);
    return BDG_OK;
}

// Stage-1 TSV -> read ids + observed barcodes, the way badger.py:91-111 takes it in through pandas: the columns "#read_id" and
// "barcode" by the first line's names, repeated header rows skipped (:104,107), an empty / NA barcode is '*', a barcode of
// bc_len + 1 letters loses its last one (:108-109).  usable[i] = the read has a barcode of bc_len letters; its rank
// (common.py:21-25) or BDG_E_BADBASE for a letter outside ACGT (the reference's rank() raises KeyError).
int bdg_import_stage1_tsv(const char* path, uint32_t bc_len, bdg_idstore* ids, uint32_t** rank_out, uint8_t** usable_out, uint64_t* n_out, uint64_t* bad_line)
{
    if (!path || !ids || !rank_out || !usable_out || !n_out || bc_len == 0 || bc_len > 16) return BDG_E_ARG;
    *rank_out = nullptr; *usable_out = nullptr; *n_out = 0;
    if (bad_line) *bad_line = 0;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return BDG_E_ARG;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { ::close(fd); return BDG_E_ARG; }
    const size_t size = (size_t)sb.st_size;
    if (size == 0) { ::close(fd); return BDG_E_FORMAT; }                       // (pandas: EmptyDataError "No columns to parse from file")
    void* const map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) return BDG_E_ARG;
    const char* const begin = static_cast<const char*>(map);
    const char* const end = begin + size;

    // the first line names the columns
    int ci = -1, cb = -1;
    const char* body;
    {
        const char* nl = static_cast<const char*>(memchr(begin, '\n', size));
        const char* le = nl ? nl : end;
        body = nl ? nl + 1 : end;
        if (le > begin && le[-1] == '\r') --le;
        int col = 0;
        for (const char* q = begin;; ++col) {
            const char* t = static_cast<const char*>(memchr(q, '\t', (size_t)(le - q)));
            const size_t l = (size_t)((t ? t : le) - q);
            if (l == 8 && memcmp(q, "#read_id", 8) == 0 && ci < 0) ci = col;
            if (l == 7 && memcmp(q, "barcode", 7) == 0 && cb < 0) cb = col;
            if (!t) break;
            q = t + 1;
        }
        if (ci < 0 || cb < 0) { munmap(map, size); return BDG_E_FORMAT; }
    }

    // the lines behind it, in ranges cut at line ends: one thread per range, results joined in file order
    struct Part {
        const char* lo; const char* hi;
        std::vector<uint32_t> ranks; std::vector<uint8_t> usable; std::vector<char> text; std::vector<uint32_t> idlen;
        uint64_t lines = 0, bad = 0;                     // lines seen; 1-based line (inside the range) of the first bad letter
    };
    const size_t body_bytes = (size_t)(end - body);
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));       // (12.5 M rows: 0.45 / 0.17 / 0.12 s with 4 / 16 / 32 threads)
    if (const char* e = getenv("BADGER_AMD_IMPORT_THREADS")) nt = (unsigned)std::max(1, atoi(e));
    nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, body_bytes >> 20));        // a megabyte per thread at least
    std::vector<Part> parts(nt);
    {
        const char* at = body;
        for (unsigned k = 0; k < nt; ++k) {
            parts[k].lo = at;
            const char* want = k + 1 == nt ? end : body + body_bytes / nt * (k + 1);
            if (want < at) want = at;
            if (want < end) { const char* nl = static_cast<const char*>(memchr(want, '\n', (size_t)(end - want))); want = nl ? nl + 1 : end; }
            parts[k].hi = at = want;
        }
    }
    auto parse = [&](Part& pt) {
        static const char* const na[] = { "", "NA", "NaN", "nan", "N/A", "NULL", "null", "None" };      // what pandas reads as missing
        const size_t bytes = (size_t)(pt.hi - pt.lo);
        pt.ranks.reserve(bytes / 48); pt.usable.reserve(bytes / 48); pt.idlen.reserve(bytes / 48); pt.text.reserve(bytes / 3);
        const char* p = pt.lo;
        while (p < pt.hi) {
            const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(pt.hi - p)));
            const char* le = nl ? nl : pt.hi;
            const char* next = nl ? nl + 1 : pt.hi;
            if (le > p && le[-1] == '\r') --le;
            ++pt.lines;
            const char* fs[2] = { nullptr, nullptr }; size_t fl[2] = { 0, 0 };
            int col = 0;
            for (const char* q = p;; ++col) {
                const char* t = static_cast<const char*>(memchr(q, '\t', (size_t)(le - q)));
                const size_t l = (size_t)((t ? t : le) - q);
                if (col == ci) { fs[0] = q; fl[0] = l; }
                if (col == cb) { fs[1] = q; fl[1] = l; }
                if (!t || (fs[0] && fs[1])) break;
                q = t + 1;
            }
            const bool blank = le == p;
            p = next;
            // pandas.read_csv as badger.py:92 calls it: a blank line is skipped; a row that ends before the barcode column
            // has no barcode (NaN -> '*', :95) and stays a read; one that ends before the id column has the id NaN, which
            // to_csv writes as an empty field; a field in double quotes loses them; an id spelled like a missing value
            // ("NA", "NaN", ...) is NaN as well
            if (blank) continue;
            static const char none_field[] = "*";
            if (!fs[1]) { fs[1] = none_field; fl[1] = 1; }
            if (!fs[0]) { fs[0] = none_field; fl[0] = 0; }
            for (int f = 0; f < 2; ++f) if (fl[f] >= 2 && fs[f][0] == '"' && fs[f][fl[f] - 1] == '"') { ++fs[f]; fl[f] -= 2; }
            if (fl[0] <= 4) for (const char* t : na) if (strlen(t) == fl[0] && memcmp(t, fs[0], fl[0]) == 0) { fl[0] = 0; break; }
            if ((fl[0] == 8 && memcmp(fs[0], "#read_id", 8) == 0) || (fl[1] == 7 && memcmp(fs[1], "barcode", 7) == 0)) continue;
            size_t L = fl[1];
            bool none = L == 1 && fs[1][0] == '*';
            if (!none && L <= 4) for (const char* t : na) if (strlen(t) == L && memcmp(t, fs[1], L) == 0) { none = true; break; }
            if (!none && L == (size_t)bc_len + 1) L = bc_len;
            uint32_t r = 0; uint8_t ok = 0;
            if (!none && L == bc_len) {
                ok = 1;
                for (uint32_t i = 0; i < bc_len; ++i) {
                    uint32_t c;
                    switch (fs[1][i]) { case 'A': c = 0; break; case 'C': c = 1; break; case 'G': c = 2; break; case 'T': c = 3; break;
                                        default: pt.bad = pt.lines; return; }
                    r |= c << (2 * i);
                }
            }
            pt.text.insert(pt.text.end(), fs[0], fs[0] + fl[0]);
            pt.idlen.push_back((uint32_t)fl[0]);
            pt.ranks.push_back(r); pt.usable.push_back(ok);
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nt; ++k) th.emplace_back([&, k] { parse(parts[k]); });
        parse(parts[0]);
        for (auto& t : th) t.join();
    }
    munmap(map, size);
    uint64_t lines_before = 1;                                                     // (the header line)
    size_t n = 0, text_bytes = 0;
    for (const Part& pt : parts) {
        if (pt.bad) { if (bad_line) *bad_line = lines_before + pt.bad; return BDG_E_BADBASE; }
        lines_before += pt.lines; n += pt.ranks.size(); text_bytes += pt.text.size();
    }
    *rank_out = static_cast<uint32_t*>(malloc(sizeof(uint32_t) * (n ? n : 1)));
    *usable_out = static_cast<uint8_t*>(malloc(n ? n : 1));
    if (!*rank_out || !*usable_out) { free(*rank_out); free(*usable_out); *rank_out = nullptr; *usable_out = nullptr; return BDG_E_NOMEM; }
    ids->text.reserve(ids->text.size() + text_bytes);
    ids->off.reserve(ids->off.size() + n);
    size_t at = 0;
    for (Part& pt : parts) {
        const size_t m = pt.ranks.size();
        if (m) { memcpy(*rank_out + at, pt.ranks.data(), sizeof(uint32_t) * m); memcpy(*usable_out + at, pt.usable.data(), m); }
        at += m;
        uint64_t o = ids->text.size();
        ids->text.insert(ids->text.end(), pt.text.begin(), pt.text.end());
        for (const uint32_t l : pt.idlen) { o += l; ids->off.push_back(o); }
        pt = Part();                                                               // (its memory goes back before the next one is copied)
    }
    *n_out = n;
    return BDG_OK;
}

void bdg_host_free(void* p) { free(p); }

int bdg_write_assignments(const bdg_idstore* ids, const uint32_t* rank, const uint8_t* has, uint64_t n, const char* path)
{
    if (!ids || !path || (n && (!rank || !has)) || n != bdg_idstore_count(ids)) return BDG_E_ARG;
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return BDG_E_ARG;
    bool ok = write_all(fd, "readID\tbarcode\n", 15);
    // a row's length is known before it is written (id + tab + 16 letters or '*' + newline): the rows are cut into ranges,
    // every range knows its place in the file, and a thread formats and pwrite()s its range by itself
    unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("BADGER_AMD_WRITE_THREADS")) nt = (unsigned)std::max(1, atoi(e));
    nt = (unsigned)std::min<uint64_t>(nt, std::max<uint64_t>(1, n >> 16));                 // 65,536 rows per thread at least
    std::vector<uint64_t> lo(nt + 1), at(nt + 1);
    for (unsigned k = 0; k <= nt; ++k) lo[k] = n * k / nt;
    at[0] = 15;
    {
        std::vector<uint64_t> bytes(nt, 0);
        std::vector<std::thread> th;
        auto size_of = [&](unsigned k) {
            uint64_t b = ids->off[lo[k + 1]] - ids->off[lo[k]] + 2 * (lo[k + 1] - lo[k]);
            for (uint64_t i = lo[k]; i < lo[k + 1]; ++i) b += has[i] ? 16 : 1;
            bytes[k] = b;
        };
        for (unsigned k = 1; k < nt; ++k) th.emplace_back(size_of, k);
        size_of(0);
        for (auto& t : th) t.join();
        for (unsigned k = 0; k < nt; ++k) at[k + 1] = at[k] + bytes[k];
    }
    std::atomic<bool> good{ ok };
    auto write_range = [&](unsigned k) {
        std::vector<char> buf;
        buf.reserve(size_t(8) << 20);
        uint64_t pos = at[k];
        auto flush = [&]() {
            size_t done = 0;
            while (done < buf.size()) {
                const ssize_t w = pwrite(fd, buf.data() + done, buf.size() - done, (off_t)(pos + done));
                if (w <= 0) { good = false; return; }
                done += (size_t)w;
            }
            pos += buf.size(); buf.clear();
        };
        for (uint64_t i = lo[k]; i < lo[k + 1] && good; ++i) {
            const size_t idl = (size_t)(ids->off[i + 1] - ids->off[i]);
            const size_t a = buf.size();
            buf.resize(a + idl + 19);
            char* o = buf.data() + a;
            memcpy(o, ids->text.data() + ids->off[i], idl); o += idl;
            *o++ = '\t';
            if (has[i]) { const uint32_t r = rank[i]; for (int b = 0; b < 16; ++b) *o++ = "ACGT"[(r >> (2 * b)) & 3u]; }   // unrank, common.py:27-38
            else *o++ = '*';
            *o++ = '\n';
            buf.resize((size_t)(o - buf.data()));
            if (buf.size() > (size_t(8) << 20) - 4096) flush();
        }
        if (good && !buf.empty()) flush();
    };
    if (ok) {
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nt; ++k) th.emplace_back(write_range, k);
        write_range(0);
        for (auto& t : th) t.join();
    }
    ok = good;
    if (::close(fd) != 0) ok = false;
    return ok ? BDG_OK : BDG_E_ARG;
}


}  // extern "C"
