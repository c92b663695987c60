// Read ingest and row output of stage 1, the host side of SURVEY 8f-3 / 8f-4:
//   bdg_ingest_*     [gzipped] FASTA / FASTQ -> chunks of reads {concatenated bases, offsets, read ids} in pinned host
//                    memory, parsed by a background thread into a ring of chunks (reference extract_raw_barcodes.py:78-98
//                    format sniffing, :131-150 chunks of READ_CHUNK_SIZE = 100,000 reads); record semantics are those of
//                    Bio.SeqIO's "fasta" / "fastq" readers as the reference uses them: id = first word of the header,
//                    FASTA sequence = its lines joined, FASTQ = four-line records.
//   bdg_format_rows  one TSV row per read from the device's 32-byte records (TenXBarcodeDetectionResult.__str__,
//                    barcode_callers.py:40-42,91-93,117-119); the barcode / UMI text is sliced from the chunk's bases, for
//                    reverse-strand results from the reverse complement (barcode_extraction/common.py:34-39).
// Plain C++ (zlib for .gz); the only HIP calls are hipHostMalloc / hipHostFree for the pinned buffers.
#include "bdg_common.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace {

struct Chunk {
    uint8_t*  bases = nullptr;  size_t bases_cap = 0;
    uint64_t* off = nullptr;    size_t off_cap = 0;      // n + 1 entries
    char*     ids = nullptr;    size_t ids_cap = 0;
    uint64_t* id_off = nullptr; size_t id_off_cap = 0;   // n + 1 entries
    uint32_t  n = 0;
    uint64_t  bases_bytes = 0, ids_bytes = 0;
    int state = 0;              // 0 free, 1 filled, 2 handed to the consumer
    bool bad = false;           // the parser failed while filling this chunk
    bool nomem = false;         // ... because a buffer could not be allocated (not because of the input)
};

void* pinned_alloc(size_t bytes, bool pinned)
{
    void* p = nullptr;
    if (pinned) {
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess) return p;
        (void)hipGetLastError();
        return nullptr;
    }
    return malloc(bytes);
}
void pinned_free(void* p, bool pinned)
{
    if (!p) return;
    if (pinned) (void)hipHostFree(p); else free(p);
}

template <typename T>
bool grow(T*& p, size_t& cap, size_t used, size_t want, bool pinned)
{
    if (want <= cap) return true;
    size_t ncap = std::max(want, cap + cap / 2);
    T* q = static_cast<T*>(pinned_alloc(ncap * sizeof(T), pinned));
    if (!q) return false;
    if (used) memcpy(q, p, used * sizeof(T));
    pinned_free(p, pinned);
    p = q; cap = ncap;
    return true;
}

// ---- BGZF: gzip members of <= 64 KiB that state their own compressed size (the 'BC' extra field, SAM specification
// section 4.1; what bgzip / htslib write).  gzip.open() in the reference (extract_raw_barcodes.py:86-87) reads them as the
// multi-member gzip files they are, one core; here a dispatcher thread cuts the file into blocks, a pool inflates them and
// the parser takes the results in file order.  A member without the field (bgzip output with a plain gzip file appended)
// is inflated by the dispatcher itself, in sequence, to the end of the file.
struct BgzfBlock {
    std::vector<uint8_t> in, out;
    size_t out_len = 0;
    int state = 0;                 // 0 free, 1 waiting for a worker, 2 inflated, 3 failed
};

struct BgzfReader {
    FILE* f = nullptr;
    std::vector<BgzfBlock> q;      // ring indexed by block number % size
    uint64_t issued = 0, claimed = 0, taken = 0;   // blocks queued by the dispatcher / claimed by workers / consumed
    size_t pos = 0;                // consumer's position inside block `taken`
    bool eof = false, failed = false, stop = false;
    std::string err;
    std::mutex mu;
    std::condition_variable cv;
    std::thread dispatcher;
    std::vector<std::thread> workers;

    static bool is_bgzf_header(const uint8_t* h, size_t n)
    {
        return n >= 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 &&
               h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
    }
    void fail(const std::string& m)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) { failed = true; err = m; }
        eof = true;
        cv.notify_all();
    }
    // slot for the next block, once the consumer has drained it; nullptr when asked to stop
    BgzfBlock* next_slot()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || issued - taken < q.size(); });
        return stop ? nullptr : &q[issued % q.size()];
    }
    void publish(BgzfBlock* b, int state)
    {
        { std::lock_guard<std::mutex> lk(mu); b->state = state; ++issued; }
        cv.notify_all();
    }
    // the rest of the file as ordinary (multi-member) gzip, starting with the `have` bytes already read
    void sequential_tail(const uint8_t* head, size_t have)
    {
        z_stream z; memset(&z, 0, sizeof(z));
        if (inflateInit2(&z, 15 + 16) != Z_OK) return fail("zlib: inflateInit2 failed");
        std::vector<uint8_t> in(1u << 18);
        memcpy(in.data(), head, have);
        z.next_in = in.data(); z.avail_in = (uInt)have;
        bool member_open = true;
        for (;;) {
            if (z.avail_in == 0) {
                const size_t got = fread(in.data(), 1, in.size(), f);
                if (got == 0) {
                    if (member_open) { inflateEnd(&z); return fail("gzip: unexpected end of file"); }
                    break;
                }
                z.next_in = in.data(); z.avail_in = (uInt)got;
            }
            if (!member_open) {                                   // between members: another one, or trailing bytes (ignored, like gzread)
                if (z.avail_in < 2) {                             // the magic may straddle two reads
                    uint8_t keep = z.next_in[0];
                    in[0] = keep;
                    const size_t got = fread(in.data() + 1, 1, in.size() - 1, f);
                    z.next_in = in.data(); z.avail_in = (uInt)(got + 1);
                    if (got == 0) break;
                }
                if (z.next_in[0] != 0x1f || z.next_in[1] != 0x8b) break;
                inflateReset(&z); member_open = true;
            }
            BgzfBlock* b = next_slot();
            if (!b) { inflateEnd(&z); return; }
            b->out.resize(1u << 16);
            z.next_out = b->out.data(); z.avail_out = (uInt)b->out.size();
            const int rc = inflate(&z, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) { inflateEnd(&z); return fail(std::string("gzip: ") + (z.msg ? z.msg : "corrupt data")); }
            if (rc == Z_STREAM_END) member_open = false;
            b->out_len = b->out.size() - z.avail_out;
            publish(b, 2);
        }
        inflateEnd(&z);
    }
    void dispatch_loop()
    {
        uint8_t h[18];
        for (;;) {
            const size_t got = fread(h, 1, sizeof(h), f);
            if (got == 0) break;                                                   // clean end of file
            if (got < 2 || h[0] != 0x1f || h[1] != 0x8b) break;                   // trailing bytes behind the last member: ignored
            if (!is_bgzf_header(h, got)) { sequential_tail(h, got); break; }
            const size_t total = ((size_t)h[16] | ((size_t)h[17] << 8)) + 1;       // whole member, header and trailer included
            if (total < 18 + 8) return fail("BGZF: impossible block size");
            BgzfBlock* b = next_slot();
            if (!b) return;
            b->in.resize(total - 18);
            if (fread(b->in.data(), 1, b->in.size(), f) != b->in.size()) return fail("BGZF: unexpected end of file");
            publish(b, 1);
        }
        { std::lock_guard<std::mutex> lk(mu); eof = true; }
        cv.notify_all();
    }
    void work_loop()
    {
        z_stream z; memset(&z, 0, sizeof(z));
        if (inflateInit2(&z, -15) != Z_OK) return fail("zlib: inflateInit2 failed");
        for (;;) {
            BgzfBlock* b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || claimed < issued || (eof && claimed == issued); });
                if (stop || claimed == issued) break;
                b = &q[claimed % q.size()];
                ++claimed;
                if (b->state != 1) continue;                                       // inflated by the dispatcher already
            }
            const size_t n = b->in.size();
            const uint8_t* t = b->in.data() + n - 8;                               // CRC32, ISIZE
            const uint32_t crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
            const uint32_t isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
            int state = 2;
            if (isize > (1u << 16)) state = 3;
            else {
                b->out.resize(isize ? isize : 1);
                inflateReset(&z);
                z.next_in = b->in.data(); z.avail_in = (uInt)(n - 8);
                z.next_out = b->out.data(); z.avail_out = isize;
                const int rc = inflate(&z, Z_FINISH);
                if (rc != Z_STREAM_END || z.avail_out != 0 || z.avail_in != 0 ||
                    (uint32_t)crc32(crc32(0L, Z_NULL, 0), b->out.data(), isize) != crc) state = 3;
                b->out_len = isize;
            }
            { std::lock_guard<std::mutex> lk(mu); b->state = state; }
            cv.notify_all();
        }
        inflateEnd(&z);
    }
    // like gzread: up to cap bytes, 0 at the end, -1 on a corrupt file
    int read(char* dst, size_t cap)
    {
        size_t done = 0;
        std::unique_lock<std::mutex> lk(mu);
        while (done < cap) {
            cv.wait(lk, [&] { return (taken < issued && q[taken % q.size()].state >= 2) || (eof && taken == issued); });
            if (taken == issued) break;
            BgzfBlock& b = q[taken % q.size()];
            if (b.state == 3) { if (!failed) { failed = true; err = "BGZF: corrupt block (size or checksum)"; } break; }
            const size_t k = std::min(cap - done, b.out_len - pos);
            if (k) {
                lk.unlock();                                   // the block is ours until `taken` moves
                memcpy(dst + done, b.out.data() + pos, k);
                lk.lock();
            }
            done += k; pos += k;
            if (pos == b.out_len) { b.state = 0; pos = 0; ++taken; cv.notify_all(); }
        }
        if (done == 0 && failed) return -1;
        return (int)done;
    }
    void start(FILE* file, unsigned threads)
    {
        f = file;
        if (threads < 1) threads = 1;
        q.resize(64 * (size_t)threads);
        dispatcher = std::thread(&BgzfReader::dispatch_loop, this);
        for (unsigned i = 0; i < threads; ++i) workers.emplace_back(&BgzfReader::work_loop, this);
    }
    ~BgzfReader()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        if (dispatcher.joinable()) dispatcher.join();
        for (auto& w : workers) if (w.joinable()) w.join();
        if (f) fclose(f);
    }
};

}  // namespace

struct bdg_ingest {
    gzFile gz = nullptr;
    BgzfReader* bgzf = nullptr;
    int format = 0;             // 0 FASTA, 1 FASTQ
    uint32_t chunk_reads = 100000;
    bool pinned = true;
    std::vector<Chunk> ring;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    size_t head = 0;            // next chunk the consumer takes
    bool done = false, stop = false, failed = false;
    std::string err;
    // line reader
    std::vector<char> buf; size_t bpos = 0, blen = 0;
    std::string carry;
    uint64_t line_no = 0;

    bool fill()
    {
        int got;
        if (bgzf) {
            got = bgzf->read(buf.data(), buf.size());
            if (got < 0) { err = "read error: " + bgzf->err; failed = true; return false; }
        } else {
            got = gzread(gz, buf.data(), (unsigned)buf.size());
            if (got < 0) { int e = 0; err = std::string("read error: ") + gzerror(gz, &e); failed = true; return false; }
        }
        bpos = 0; blen = (size_t)got;
        return got > 0;
    }
    // next line without its terminator ("\n" or "\r\n"); false at end of file
    bool next_line(const char*& p, size_t& len)
    {
        carry.clear();
        bool have = false;
        for (;;) {
            if (bpos == blen) { if (!fill()) break; }
            const char* s = buf.data() + bpos;
            const char* nl = static_cast<const char*>(memchr(s, '\n', blen - bpos));
            if (nl) {
                const size_t l = (size_t)(nl - s);
                bpos += l + 1;
                if (have || !carry.empty()) { carry.append(s, l); p = carry.data(); len = carry.size(); }
                else { p = s; len = l; }
                ++line_no;
                while (len && (p[len - 1] == '\r' || p[len - 1] == '\n')) --len;
                return true;
            }
            carry.append(s, blen - bpos); have = true;
            bpos = blen;
        }
        if (failed) return false;
        if (have || !carry.empty()) {
            p = carry.data(); len = carry.size(); ++line_no;
            while (len && (p[len - 1] == '\r' || p[len - 1] == '\n')) --len;
            return true;
        }
        return false;
    }
};

namespace {

bool chunk_begin_read(bdg_ingest* g, Chunk& c, const char* id, size_t idlen)
{
    // id = first whitespace-delimited word of the header (Bio.SeqIO)
    size_t a = 0;
    while (a < idlen && (id[a] == ' ' || id[a] == '\t')) ++a;
    size_t b = a;
    while (b < idlen && id[b] != ' ' && id[b] != '\t' && id[b] != '\r') ++b;
    if (!grow(c.ids, c.ids_cap, (size_t)c.ids_bytes, (size_t)c.ids_bytes + (b - a) + 1, false)) return false;
    if (!grow(c.id_off, c.id_off_cap, (size_t)c.n + 1, (size_t)c.n + 2, false)) return false;
    if (!grow(c.off, c.off_cap, (size_t)c.n + 1, (size_t)c.n + 2, g->pinned)) return false;
    memcpy(c.ids + c.ids_bytes, id + a, b - a);
    c.ids_bytes += b - a;
    return true;
}
bool chunk_append_bases(bdg_ingest* g, Chunk& c, const char* s, size_t len)
{
    if (!len) return true;
    if (!grow(c.bases, c.bases_cap, (size_t)c.bases_bytes, (size_t)c.bases_bytes + len + 64, g->pinned)) return false;
    memcpy(c.bases + c.bases_bytes, s, len);
    c.bases_bytes += len;
    return true;
}
void chunk_end_read(Chunk& c)
{
    ++c.n;
    c.off[c.n] = c.bases_bytes;
    c.id_off[c.n] = c.ids_bytes;
}

void strip(const char*& p, size_t& len)
{
    while (len && (*p == ' ' || *p == '\t' || *p == '\r')) { ++p; --len; }
    while (len && (p[len - 1] == ' ' || p[len - 1] == '\t' || p[len - 1] == '\r')) --len;
}

// parser thread: fills free chunks in ring order
void parse_loop(bdg_ingest* g)
{
    size_t tail = 0;
    bool in_record = false;            // FASTA: a header has been seen and its record is open
    std::string pending_id; bool have_pending = false;     // FASTA header that closed the previous chunk's last record
    bool eof = false;
    while (!eof) {
        Chunk* c;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait(lk, [&] { return g->stop || g->ring[tail].state == 0; });
            if (g->stop) return;
            c = &g->ring[tail];
        }
        c->n = 0; c->bases_bytes = 0; c->ids_bytes = 0; c->bad = false; c->nomem = false;
        // sized for a typical chunk up front (pinned allocations are slow), grown on demand
        bool ok = grow(c->off, c->off_cap, 0, (size_t)g->chunk_reads + 2, g->pinned) && grow(c->id_off, c->id_off_cap, 0, (size_t)g->chunk_reads + 2, false) &&
                  grow(c->bases, c->bases_cap, 0, (size_t)g->chunk_reads * 1200 + 64, g->pinned) && grow(c->ids, c->ids_cap, 0, (size_t)g->chunk_reads * 40 + 64, false);
        if (ok) { c->off[0] = 0; c->id_off[0] = 0; }
        const char* p; size_t len;
        if (ok && g->format == 0) {
            if (have_pending) { ok = chunk_begin_read(g, *c, pending_id.data(), pending_id.size()); have_pending = false; in_record = true; }
            while (ok) {
                if (!g->next_line(p, len)) { eof = true; break; }
                if (len && p[0] == '>') {
                    if (in_record) chunk_end_read(*c);
                    if (c->n >= g->chunk_reads) { pending_id.assign(p + 1, len - 1); have_pending = true; in_record = false; break; }
                    ok = chunk_begin_read(g, *c, p + 1, len - 1);
                    in_record = true;
                } else if (in_record) {
                    strip(p, len);
                    ok = chunk_append_bases(g, *c, p, len);
                }
            }
            if (eof && in_record) { chunk_end_read(*c); in_record = false; }
        } else if (ok) {
            while (ok && c->n < g->chunk_reads) {
                if (!g->next_line(p, len)) { eof = true; break; }
                if (len == 0) continue;                                  // blank line between records
                if (p[0] != '@') { g->err = "malformed FASTQ record header at line " + std::to_string(g->line_no); g->failed = true; break; }
                ok = chunk_begin_read(g, *c, p + 1, len - 1);
                if (!ok) break;
                const uint64_t b0 = c->bases_bytes;
                // (a read error inside a record keeps its own message: next_line has set `failed` then)
                auto malformed = [&](const char* what) { if (!g->failed) { g->err = std::string(what) + " at line " + std::to_string(g->line_no); g->failed = true; } };
                if (!g->next_line(p, len)) { malformed("truncated FASTQ record"); break; }
                ok = chunk_append_bases(g, *c, p, len);
                const uint64_t slen = c->bases_bytes - b0;
                if (!g->next_line(p, len) || len == 0 || p[0] != '+') { malformed("malformed FASTQ record (no '+' line)"); break; }
                if (!g->next_line(p, len) || len != slen) { malformed("malformed FASTQ record (quality length)"); break; }
                chunk_end_read(*c);
            }
        }
        if (!ok && !g->failed) { g->err = "out of (pinned) host memory while reading"; g->failed = true; c->nomem = true; }
        if (g->failed) { eof = true; c->bad = true; }
        {
            std::lock_guard<std::mutex> lk(g->mu);
            c->state = 1;
            if (eof) g->done = true;
        }
        g->cv.notify_all();
        tail = (tail + 1) % g->ring.size();
    }
}

bool ends_with_ci(const std::string& s, const char* suf)
{
    const size_t n = strlen(suf);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; ++i) if (tolower((unsigned char)s[s.size() - n + i]) != suf[i]) return false;
    return true;
}

}  // namespace

extern "C" {

int bdg_ingest_open(const char* path, uint32_t chunk_reads, uint32_t ring_chunks, int pinned, bdg_ingest** out)
{
    return bdg_ingest_open_mt(path, chunk_reads, ring_chunks, pinned, 0, out);
}

int bdg_ingest_open_mt(const char* path, uint32_t chunk_reads, uint32_t ring_chunks, int pinned, uint32_t inflate_threads,
                       bdg_ingest** out)
{
    if (!out) return BDG_E_ARG;
    *out = nullptr;
    if (!path || chunk_reads == 0) return BDG_E_ARG;
    std::string name(path);
    if (ends_with_ci(name, ".gz")) name.resize(name.size() - 3);
    else if (ends_with_ci(name, ".gzip")) name.resize(name.size() - 5);
    int format;
    if (ends_with_ci(name, ".fq") || ends_with_ci(name, ".fastq")) format = 1;
    else if (ends_with_ci(name, ".fa") || ends_with_ci(name, ".fasta")) format = 0;
    else return BDG_E_ARG;                                   // unknown extension (BAM / SAM are the caller's business)
    // BGZF (blocked gzip) is inflated by a pool of threads; anything else goes through zlib's reader, one thread
    BgzfReader* bgzf = nullptr;
    gzFile gz = nullptr;
    if (inflate_threads != 1) {
        FILE* f = fopen(path, "rb");
        if (!f) return BDG_E_ARG;
        uint8_t h[18];
        const size_t got = fread(h, 1, sizeof(h), f);
        if (BgzfReader::is_bgzf_header(h, got)) {
            rewind(f);
            unsigned t = inflate_threads ? inflate_threads : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
            bgzf = new BgzfReader();
            bgzf->start(f, t);
        } else fclose(f);
    }
    if (!bgzf) {
        gz = gzopen(path, "rb");                             // plain files are read through unchanged
        if (!gz) return BDG_E_ARG;
        gzbuffer(gz, 1u << 20);
    }
    bdg_ingest* g = new bdg_ingest();
    g->gz = gz; g->bgzf = bgzf; g->format = format; g->chunk_reads = chunk_reads; g->pinned = pinned != 0;
    g->ring.resize(ring_chunks < 2 ? 2 : ring_chunks);
    g->buf.resize(4u << 20);
    g->worker = std::thread(parse_loop, g);
    *out = g;
    return BDG_OK;
}

int bdg_ingest_next(bdg_ingest* g, bdg_ingest_chunk* out)
{
    if (!g || !out) return BDG_E_ARG;
    memset(out, 0, sizeof(*out));
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [&] { return g->ring[g->head].state == 1 || (g->done && g->ring[g->head].state != 1); });
    Chunk& c = g->ring[g->head];
    if (c.state != 1) return BDG_OK;                                          // end of file: n = 0
    if (c.bad) return c.nomem ? BDG_E_NOMEM : BDG_E_FORMAT;                   // chunks before the failure were good
    c.state = 2;
    out->id = (uint32_t)g->head;
    out->n = c.n; out->bases = c.bases; out->off = c.off; out->total_bytes = c.bases_bytes;
    out->ids = c.ids; out->id_off = c.id_off;
    g->head = (g->head + 1) % g->ring.size();
    return BDG_OK;
}

int bdg_ingest_release(bdg_ingest* g, uint32_t id)
{
    if (!g || id >= g->ring.size()) return BDG_E_ARG;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        if (g->ring[id].state == 2) g->ring[id].state = 0;
    }
    g->cv.notify_all();
    return BDG_OK;
}

const char* bdg_ingest_error(bdg_ingest* g) { return g ? g->err.c_str() : "no reader"; }

void bdg_ingest_close(bdg_ingest* g)
{
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->stop = true;
    }
    g->cv.notify_all();
    if (g->worker.joinable()) g->worker.join();
    for (Chunk& c : g->ring) {
        pinned_free(c.bases, g->pinned); pinned_free(c.off, g->pinned);
        free(c.ids); free(c.id_off);
    }
    if (g->gz) gzclose(g->gz);
    delete g->bgzf;
    delete g;
}

// ---- rows -------------------------------------------------------------------------------------------------------------
static inline char comp_base(char c)
{
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; }   // N -> N
}

static inline char* put_int(char* o, int v)
{
    char t[16]; int k = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { t[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *o++ = '-';
    while (k) *o++ = t[--k];
    return o;
}

int64_t bdg_format_rows(const bdg_ingest_chunk* ch, const bdg_extract_rec* recs, char* out, uint64_t cap, uint64_t counts[4])
{
    if (!ch || (ch->n && (!recs || !ch->bases || !ch->off || !ch->ids || !ch->id_off))) return BDG_E_ARG;
    uint64_t need = 0, n_bc = 0, n_pt = 0, n_r1 = 0;
    for (uint32_t i = 0; i < ch->n; ++i) {
        const uint64_t L = ch->off[i + 1] - ch->off[i];
        need += (ch->id_off[i + 1] - ch->id_off[i]) + 64 + (recs[i].valid ? 16 + std::min<uint64_t>(L, (uint64_t)std::max(0, recs[i].umi_end - recs[i].umi_start)) : 2);
    }
    if (!out || need > cap) return (int64_t)need;
    char* o = out;
    for (uint32_t i = 0; i < ch->n; ++i) {
        const bdg_extract_rec& r = recs[i];
        const uint8_t* seq = ch->bases + ch->off[i];
        const int64_t L = (int64_t)(ch->off[i + 1] - ch->off[i]);
        const size_t idl = (size_t)(ch->id_off[i + 1] - ch->id_off[i]);
        memcpy(o, ch->ids + ch->id_off[i], idl); o += idl;
        *o++ = '\t';
        const bool rev = (r.flags & BDG_FLAG_REV) != 0;
        auto slice = [&](int64_t a, int64_t b) {                       // Python slice s[a:b] of the strand's text (a, b >= 0)
            a = std::min<int64_t>(std::max<int64_t>(a, 0), L); b = std::min<int64_t>(std::max<int64_t>(b, 0), L);
            for (int64_t x = a; x < b; ++x) *o++ = rev ? comp_base((char)seq[L - 1 - x]) : (char)seq[x];
        };
        if (r.valid) {
            slice(r.bc_start, (int64_t)r.bc_start + 16); *o++ = '\t';
            slice(r.umi_start, r.umi_end);
            memcpy(o, "\t0\tFalse\t", 9); o += 9;
            ++n_bc;
        } else {
            memcpy(o, "*\t*\t-1\tFalse\t", 13); o += 13;
        }
        *o++ = r.strand > 0 ? '+' : (r.strand < 0 ? '-' : '.');
        *o++ = '\t';
        o = put_int(o, r.polyT); *o++ = '\t';
        o = put_int(o, r.valid ? r.r1_end : -1);
        *o++ = '\n';
        if (r.polyT != -1) ++n_pt;
        if (r.valid && r.r1_end != -1) ++n_r1;
    }
    if (counts) { counts[0] = ch->n; counts[1] = n_bc; counts[2] = n_pt; counts[3] = n_r1; }
    return (int64_t)(o - out);
}

}  // extern "C"
