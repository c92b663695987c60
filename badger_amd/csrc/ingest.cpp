// Read ingest of stage 1, the host side of SURVEY 8f-3:
//   bdg_ingest_*     [gzipped / BGZF] FASTA / FASTQ / SAM and BAM -> chunks of reads {concatenated bases, offsets, read ids}
//                    in pinned host memory (reference extract_raw_barcodes.py:78-98 format sniffing, :100-118 / :131-150 the
//                    record loops).  Record semantics are those of the readers the reference uses: Bio.SeqIO "fasta" /
//                    "fastq" (id = first word of the header, FASTA sequence = its lines joined, FASTQ = four-line records)
//                    and pysam.AlignmentFile (query_name, query_sequence; the chunk reader skips secondary and
//                    supplementary records, :144-145, the single-thread loop does not, :110-118).
//
// How it is parallel.  The reference parses with one Python process and parallelises the per-read work behind it; here the
// per-read work is a GPU's, so the parser itself has to deliver > 10 M reads/s.  The decompressed text is cut into SEGMENTS
// (16 MiB by default) that worker threads parse independently:
//   * plain file: the file is mapped, a segment is a byte range of the mapping;
//   * BGZF (bgzip / htslib: gzip members of <= 64 KiB that state their own size, SAM specification 4.1): a segment is a run
//     of members, inflated by the worker that parses it (libdeflate when the system has it, else zlib);
//   * plain gzip: one stream without an index, inflated by several threads all the same (pgunzip.hpp); a segment is the
//     next 16 MiB of its output.
// A worker finds the first record start inside its segment (FASTQ: a line starting with '@' whose line + 2 starts with '+'
// and whose lines + 1 / + 3 have equal lengths; FASTA: a line starting with '>'; SAM: any line), parses whole records from
// there into a pinned chunk and reports what it could not own: the bytes before that start (HEAD) and the unfinished record
// at its end (TAIL).  One assembler thread walks the segments in file order and runs the sequential parser over TAIL(k-1) +
// HEAD(k): if that ends exactly on a record boundary where segment k's worker started, the worker's records are by
// induction the ones a sequential parse would have produced (the guess above only decides whether the fast path is taken,
// never what is parsed); if not - or the worker met a malformed record - the assembler parses on sequentially from there, so
// malformed input fails with the same message and line number as a one-thread parse.  BAM records are length-prefixed
// binary without a resynchronisation mark: a reader guesses a record start from the fixed fields of three records in a row.
// Chunks handed out never span two segments and hold at most chunk_reads reads; a file smaller than one segment gives the
// reference's exact READ_CHUNK_SIZE chunking.
// Plain C++ (zlib; libdeflate by dlopen); the only HIP calls are hipHostMalloc / hipHostFree for the pinned buffers.
#include "bdg_common.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "ingest_internal.hpp"
#include "pgunzip.hpp"

namespace {

// Pinned buffers are kept when a reader closes and handed to the next one (hipHostMalloc / hipHostFree of some tens of
// 32 MB buffers are a sizeable part of a one-second run): a process-wide cache of at most PINNED_CACHE_MAX bytes, never
// returned before the process ends.
struct PinnedCache {
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> idle;
    std::vector<std::pair<void*, size_t>> live;          // size of every pinned buffer handed out
    size_t idle_bytes = 0;
    double alloc_s = 0; uint64_t allocs = 0, alloc_bytes = 0;      // hipHostMalloc calls so far (BADGER_AMD_INGEST_DEBUG prints them)
};
constexpr size_t PINNED_CACHE_MAX = size_t(3) << 30;
PinnedCache& pinned_cache() { static PinnedCache* c = new PinnedCache(); return *c; }    // (leaked on purpose: no teardown order problems)

void* pinned_alloc(size_t bytes, bool pinned)
{
    if (!pinned) return malloc(bytes);
    PinnedCache& pc = pinned_cache();
    {
        std::lock_guard<std::mutex> lk(pc.mu);
        size_t best = pc.idle.size();
        for (size_t i = 0; i < pc.idle.size(); ++i)
            if (pc.idle[i].second >= bytes && pc.idle[i].second <= 2 * bytes + (1u << 20) && (best == pc.idle.size() || pc.idle[i].second < pc.idle[best].second)) best = i;
        if (best != pc.idle.size()) {
            const std::pair<void*, size_t> e = pc.idle[best];
            pc.idle.erase(pc.idle.begin() + (long)best);
            pc.idle_bytes -= e.second;
            pc.live.push_back(e);
            return e.first;
        }
    }
    void* p = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(pc.mu);
    pc.live.emplace_back(p, bytes);
    pc.alloc_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); pc.allocs++; pc.alloc_bytes += bytes;
    return p;
}
void pinned_free(void* p, bool pinned)
{
    if (!p) return;
    if (!pinned) { free(p); return; }
    PinnedCache& pc = pinned_cache();
    {
        std::lock_guard<std::mutex> lk(pc.mu);
        for (size_t i = 0; i < pc.live.size(); ++i) {
            if (pc.live[i].first != p) continue;
            const std::pair<void*, size_t> e = pc.live[i];
            pc.live[i] = pc.live.back(); pc.live.pop_back();
            if (pc.idle_bytes + e.second <= PINNED_CACHE_MAX) { pc.idle.push_back(e); pc.idle_bytes += e.second; return; }
            break;
        }
    }
    (void)hipHostFree(p);
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

enum { F_FASTA = 0, F_FASTQ = 1, F_SAM = 2, F_BAM = 3 };

// ---- inflate: libdeflate when the system has it (about three times zlib's speed on BGZF members), else zlib -----------
struct Deflate {
    typedef void* (*alloc_fn)(void);
    typedef int (*decomp_fn)(void*, const void*, size_t, void*, size_t, size_t*);
    typedef void (*free_fn)(void*);
    typedef uint32_t (*crc_fn)(uint32_t, const void*, size_t);
    alloc_fn alloc = nullptr; decomp_fn decomp = nullptr; free_fn release = nullptr; crc_fn crc = nullptr;
    Deflate()
    {
        const char* off = getenv("BADGER_AMD_NO_LIBDEFLATE");
        if (off && off[0] == '1') return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (alloc_fn)dlsym(h, "libdeflate_alloc_decompressor");
        decomp = (decomp_fn)dlsym(h, "libdeflate_deflate_decompress");
        release = (free_fn)dlsym(h, "libdeflate_free_decompressor");
        crc = (crc_fn)dlsym(h, "libdeflate_crc32");
        if (!alloc || !decomp || !release || !crc) alloc = nullptr;
    }
    bool have() const { return alloc != nullptr; }
};
const Deflate& deflate_lib() { static Deflate d; return d; }

struct Inflater {                       // one per worker thread
    void* ld = nullptr;
    z_stream z; bool z_open = false;
    ~Inflater() { if (ld) deflate_lib().release(ld); if (z_open) inflateEnd(&z); }
    // one raw deflate stream of known inflated size; checks size and CRC32
    bool raw(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out, uint32_t crc)
    {
        const Deflate& L = deflate_lib();
        if (L.have()) {
            if (!ld && !(ld = L.alloc())) return false;
            size_t got = 0;
            if (L.decomp(ld, in, n_in, out, n_out, &got) != 0 || got != n_out) return false;
            return L.crc(0, out, n_out) == crc;
        }
        if (!z_open) { memset(&z, 0, sizeof(z)); if (inflateInit2(&z, -15) != Z_OK) return false; z_open = true; }
        else inflateReset(&z);
        uint8_t dummy = 0;
        z.next_in = const_cast<uint8_t*>(in); z.avail_in = (uInt)n_in;
        z.next_out = n_out ? out : &dummy; z.avail_out = (uInt)n_out;
        const int rc = inflate(&z, Z_FINISH);
        if (rc != Z_STREAM_END || z.avail_out != 0 || z.avail_in != 0) return false;
        return (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, (uInt)n_out) == crc;
    }
};

struct BgzfRef { const uint8_t* in; uint32_t n_in, n_out, crc; };

// ---- a segment of the decompressed input --------------------------------------------------------------------------------
struct Segment {
    uint64_t seq = 0;
    const uint8_t* data = nullptr; size_t len = 0;
    uint8_t* own = nullptr; size_t own_cap = 0;          // text of a compressed source (kept for reuse)
    std::vector<BgzfRef> blocks;                         // BGZF members to inflate into `own`
    bool first = false;                                  // starts at a record boundary (the file's first segment)
    int state = 0;                                       // 0 free, 1 being produced, 2 ready for the assembler
    bool failed = false; std::string err;                // the source failed inside this segment
    // the worker's parse (parsed = false: the assembler parses the whole segment itself)
    bool parsed = false, bad = false;
    IngestChunk* chunk = nullptr;
    size_t head_len = 0, tail_off = 0;
    uint64_t lines = 0;                                  // lines inside [head_len, tail_off)
    bool reserve(size_t n)
    {
        if (n <= own_cap) return true;
        free(own);
        own = static_cast<uint8_t*>(malloc(n + 64)); own_cap = own ? n : 0;
        return own != nullptr;
    }
    ~Segment() { free(own); }
};

// ---- the byte source ----------------------------------------------------------------------------------------------------
struct Source {
    int fd = -1; const uint8_t* map = nullptr; size_t size = 0, pos = 0;
    bool compressed = false, bgzf_parallel = true;
    size_t seg_bytes = size_t(16) << 20;
    z_stream z; bool z_init = false, member_open = false;
    uint64_t produced = 0;
    // a plain gzip member of some size is inflated by several threads (pgunzip.hpp); small ones and threads = 1 take zlib
    std::unique_ptr<PGunzip> pgz; unsigned gz_threads = 1;
    static constexpr size_t PGZ_MIN_BYTES = size_t(1) << 20;

    static size_t pgz_min()
    {
        const char* e = getenv("BADGER_AMD_GUNZIP_MIN_KB");                 // (for tests: small files through the parallel path)
        return e ? (size_t)std::max(0, atoi(e)) << 10 : PGZ_MIN_BYTES;
    }
    static bool is_bgzf_header(const uint8_t* h, size_t n)
    {
        return n >= 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 &&
               h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
    }
    bool open(const char* path, std::string& err)
    {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) { err = "cannot open the file"; return false; }
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { err = "not a regular file"; return false; }
        size = (size_t)st.st_size;
        if (size) {
            void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { err = "cannot map the file"; return false; }
            map = static_cast<const uint8_t*>(m);
            (void)madvise(m, size, MADV_SEQUENTIAL);
        }
        compressed = size >= 2 && map[0] == 0x1f && map[1] == 0x8b;
        return true;
    }
    ~Source()
    {
        pgz.reset();                                         // (its threads read the mapping)
        if (z_init) inflateEnd(&z);
        if (map && size > unmapped) munmap(const_cast<uint8_t*>(map) + unmapped, size - unmapped);
        if (fd >= 0) close(fd);
    }
    // The next segment in file order; false at the end of the input (s.failed: the input is damaged).  Called by one thread
    // at a time.  A BGZF segment still has to be inflated (materialise, any thread).
    bool claim(Segment& s)
    {
        s.data = nullptr; s.len = 0; s.blocks.clear(); s.failed = false; s.err.clear(); s.first = produced == 0;
        if (!compressed) {
            if (pos >= size) return false;
            s.data = map + pos; s.len = std::min(seg_bytes, size - pos);
            pos += s.len; ++produced;
            return true;
        }
        for (;;) {
            if (!member_open) {
                if (size - pos < 2 || map[pos] != 0x1f || map[pos + 1] != 0x8b) return false;  // the end, or trailing bytes that are no member: ignored
                if (bgzf_parallel && is_bgzf_header(map + pos, size - pos)) {
                    size_t out = 0;
                    while (out < seg_bytes && is_bgzf_header(map + pos, size - pos)) {
                        const size_t total = ((size_t)map[pos + 16] | ((size_t)map[pos + 17] << 8)) + 1;   // whole member, header and trailer included
                        if (total < 18 + 8) { s.failed = true; s.err = "BGZF: impossible block size"; break; }
                        if (total > size - pos) { s.failed = true; s.err = "BGZF: unexpected end of file"; break; }
                        const uint8_t* t = map + pos + total - 8;                               // CRC32, ISIZE
                        BgzfRef b;
                        b.in = map + pos + 18; b.n_in = (uint32_t)(total - 18 - 8);
                        b.crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
                        b.n_out = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
                        if (b.n_out > (1u << 16)) { s.failed = true; s.err = "BGZF: corrupt block (size or checksum)"; break; }
                        s.blocks.push_back(b);
                        out += b.n_out; pos += total;
                    }
                    s.len = out; ++produced;
                    return true;
                }
                if (!z_init) {
                    memset(&z, 0, sizeof(z));
                    if (inflateInit2(&z, 15 + 16) != Z_OK) { s.failed = true; s.err = "zlib: inflateInit2 failed"; ++produced; return true; }
                    z_init = true;
                } else inflateReset(&z);
                member_open = true;
            }
            // sequential stream: up to seg_bytes of text, across members
            if (!s.reserve(seg_bytes)) { s.failed = true; s.err = "out of memory"; ++produced; return true; }
            size_t out = 0;
            while (out < seg_bytes) {
                if (!member_open) {
                    if (size - pos < 2 || map[pos] != 0x1f || map[pos + 1] != 0x8b) break;
                    if (bgzf_parallel && is_bgzf_header(map + pos, size - pos)) break;          // the next claim takes the blocks
                    inflateReset(&z); member_open = true;
                }
                if (!pgz && gz_threads > 1 && z.total_in == 0 && size - pos >= pgz_min())       // at a member's first byte
                    pgz.reset(new PGunzip(map + pos, size - pos, gz_threads, 0));
                if (pgz) {
                    out += pgz->read(s.own + out, seg_bytes - out);
                    if (pgz->failed()) { s.failed = true; s.err = pgz->error(); pgz.reset(); break; }
                    if (pgz->at_member_end()) { pos += pgz->consumed(); pgz.reset(); member_open = false; }
                    continue;
                }
                const size_t in_av = std::min<size_t>(size - pos, 1u << 30), out_av = std::min<size_t>(seg_bytes - out, 1u << 30);
                if (in_av == 0) { s.failed = true; s.err = "gzip: unexpected end of file"; break; }     // (gzip.open raises EOFError)
                z.next_in = const_cast<uint8_t*>(map + pos); z.avail_in = (uInt)in_av;
                z.next_out = s.own + out; z.avail_out = (uInt)out_av;
                const int rc = inflate(&z, Z_NO_FLUSH);
                pos += in_av - z.avail_in; out += out_av - z.avail_out;
                if (rc == Z_STREAM_END) member_open = false;
                else if (rc != Z_OK && rc != Z_BUF_ERROR) { s.failed = true; s.err = std::string("gzip: ") + (z.msg ? z.msg : "corrupt data"); break; }
            }
            s.data = s.own; s.len = out;
            if (out || s.failed) { ++produced; return true; }
            // nothing came out (empty members): look at what follows
        }
    }
    // The assembler has read the last byte of a segment.  A mapped plain file gives its pages back now, piece by piece and
    // beside the run: tearing down the whole mapping at the end costs 0.35 s for a 25 GB file (6 M page-table entries),
    // on the one thread everybody waits for.
    void done_with(const Segment& s)
    {
        if (compressed || !s.data || !s.len || !map) return;
        // (segments are finished in file order: everything in front of this one's end is done; the page its end shares with
        // the next segment stays.  Only the part still mapped may ever be unmapped again - the hole can be somebody else's by then.)
        const uintptr_t page = 4096, base = reinterpret_cast<uintptr_t>(map);
        const uintptr_t end = (reinterpret_cast<uintptr_t>(s.data) + s.len) & ~(page - 1);
        if (end > base + unmapped) { (void)munmap(reinterpret_cast<void*>(base + unmapped), end - (base + unmapped)); unmapped = end - base; }
    }
    size_t unmapped = 0;                 // bytes at the front of the mapping that have been given back
    // BGZF: inflate the segment's members (any thread); text before a damaged member is kept
    static void materialise(Segment& s, Inflater& inf)
    {
        if (s.blocks.empty()) return;
        size_t at = 0;
        if (!s.reserve(s.len ? s.len : 1)) { s.failed = true; s.err = "out of memory"; s.len = 0; return; }
        for (const BgzfRef& b : s.blocks) {
            if (!inf.raw(b.in, b.n_in, s.own + at, b.n_out, b.crc)) { s.failed = true; s.err = "BGZF: corrupt block (size or checksum)"; break; }
            at += b.n_out;
        }
        s.data = s.own; s.len = at;
    }
};

// ---- parsers --------------------------------------------------------------------------------------------------------------
struct ChunkSink { virtual IngestChunk* chunk_full(IngestChunk* full) = 0; virtual ~ChunkSink() {} };

// htslib's seq_nt16_table followed by seq_nt16_str: what a SAM text SEQ becomes on its way through pysam's query_sequence
// (sam_parse1 packs it into 4-bit codes, query_sequence unpacks them).  htslib is not part of the reference tree; this is
// its published table.
const char NT16_STR[17] = "=ACMGRSVTWYHKDBN";
uint8_t nt16_code(uint8_t c)
{
    switch (c) {
        case '=': return 0;
        case 'A': case 'a': case '0': return 1;  case 'C': case 'c': case '1': return 2;  case 'M': case 'm': return 3;
        case 'G': case 'g': case '2': return 4;  case 'R': case 'r': return 5;  case 'S': case 's': return 6;
        case 'V': case 'v': return 7;            case 'T': case 't': case '3': return 8;  case 'W': case 'w': return 9;
        case 'Y': case 'y': return 10;           case 'H': case 'h': return 11; case 'K': case 'k': return 12;
        case 'D': case 'd': return 13;           case 'B': case 'b': return 14; default: return 15;
    }
}

struct Parser {
    int format = F_FASTQ;
    bool pinned = false, skip_secondary = false;
    uint32_t limit = 0;                 // reads per chunk (0 = no limit: a worker's segment chunk)
    ChunkSink* sink = nullptr;
    IngestChunk* c = nullptr;
    // state of the record in progress
    int st = 0;                         // FASTQ: 0 header expected, 1 sequence, 2 '+', 3 quality; FASTA: 1 = a record is open
    uint64_t slen = 0;
    uint64_t rec_bases0 = 0, rec_ids0 = 0;
    size_t rec_off = 0, consumed = 0;   // offsets into the buffer of the current feed(): the open record's header line / behind the last whole line
    std::string partial;                // a line (BAM: an item) that the last feed() left unfinished
    uint64_t line_no = 0;
    bool failed = false, nomem = false, noseq = false;
    std::string err;
    // BAM
    int bst = 0; uint64_t skip = 0; uint32_t refs_left = 0, need = 8;
    size_t bam_mark = 0, bam_end = 0;   // offsets into the buffer of the current feed(): where the record in progress starts / what has been looked at
    uint8_t sam_map[256]; uint16_t bam_pair[256];

    void init(int fmt, bool pin, bool skip2, uint32_t lim, ChunkSink* sk)
    {
        format = fmt; pinned = pin; skip_secondary = skip2; limit = lim; sink = sk;
        for (int i = 0; i < 256; ++i) {
            sam_map[i] = (uint8_t)NT16_STR[nt16_code((uint8_t)i)];
            bam_pair[i] = (uint16_t)((uint8_t)NT16_STR[i >> 4] | ((uint16_t)(uint8_t)NT16_STR[i & 15] << 8));
        }
    }
    bool fail(const std::string& m) { if (!failed) { failed = true; err = m; } return false; }
    bool oom() { nomem = true; return fail("out of (pinned) host memory while reading"); }

    bool begin_read(const char* id, size_t idlen, bool first_word)
    {
        size_t a = 0, b = idlen;
        if (first_word) {               // id = first whitespace-delimited word of the header (Bio.SeqIO)
            while (a < idlen && (id[a] == ' ' || id[a] == '\t')) ++a;
            b = a;
            while (b < idlen && id[b] != ' ' && id[b] != '\t' && id[b] != '\r') ++b;
        }
        rec_bases0 = c->bases_bytes; rec_ids0 = c->ids_bytes;
        if (!grow(c->ids, c->ids_cap, (size_t)c->ids_bytes, (size_t)c->ids_bytes + (b - a) + 1, false)) return oom();
        if (!grow(c->id_off, c->id_off_cap, (size_t)c->n + 1, (size_t)c->n + 2, false)) return oom();
        if (!grow(c->off, c->off_cap, (size_t)c->n + 1, (size_t)c->n + 2, false)) return oom();
        memcpy(c->ids + c->ids_bytes, id + a, b - a);
        c->ids_bytes += b - a;
        return true;
    }
    uint8_t* bases_room(size_t len)
    {
        if (!grow(c->bases, c->bases_cap, (size_t)c->bases_bytes, (size_t)c->bases_bytes + len + 64, pinned)) { oom(); return nullptr; }
        return c->bases + c->bases_bytes;
    }
    bool append_bases(const char* s, size_t len)
    {
        if (!len) return true;
        uint8_t* d = bases_room(len);
        if (!d) return false;
        memcpy(d, s, len);
        c->bases_bytes += len;
        return true;
    }
    void end_read()
    {
        ++c->n;
        c->off[c->n] = c->bases_bytes;
        c->id_off[c->n] = c->ids_bytes;
        if (limit && c->n >= limit && sink) c = sink->chunk_full(c);
    }
    void rollback_open() { c->bases_bytes = rec_bases0; c->ids_bytes = rec_ids0; }
    // the record in progress moves to an empty chunk (the assembler leaves the fast path in the middle of a record)
    bool move_open_record(IngestChunk* to)
    {
        IngestChunk* from = c;
        const uint64_t nb = from->bases_bytes - rec_bases0, ni = from->ids_bytes - rec_ids0;
        c = to;
        if (!grow(to->ids, to->ids_cap, 0, (size_t)ni + 1, false) || !grow(to->id_off, to->id_off_cap, 0, 2, false) ||
            !grow(to->off, to->off_cap, 0, 2, false) || !grow(to->bases, to->bases_cap, 0, (size_t)nb + 64, pinned)) return oom();
        memcpy(to->ids, from->ids + rec_ids0, ni); memcpy(to->bases, from->bases + rec_bases0, nb);
        to->n = 0; to->off[0] = 0; to->id_off[0] = 0; to->bases_bytes = nb; to->ids_bytes = ni;
        from->bases_bytes = rec_bases0; from->ids_bytes = rec_ids0;
        rec_bases0 = 0; rec_ids0 = 0;
        return true;
    }
    bool record_open() const { return format == F_FASTQ ? st != 0 : (format == F_FASTA ? st == 1 : false); }

    static void rstrip(const char* p, size_t& len) { while (len && (p[len - 1] == ' ' || p[len - 1] == '\t' || p[len - 1] == '\r' || p[len - 1] == '\n')) --len; }
    static void strip(const char*& p, size_t& len)
    {
        while (len && (*p == ' ' || *p == '\t' || *p == '\r')) { ++p; --len; }
        rstrip(p, len);
    }

    bool line(const char* p, size_t len, size_t at)
    {
        ++line_no;
        while (len && (p[len - 1] == '\r' || p[len - 1] == '\n')) --len;
        switch (format) {
        case F_FASTQ:
            switch (st) {
            case 0:
                if (len == 0) return true;                               // blank line between records
                if (p[0] != '@') return fail("malformed FASTQ record header at line " + std::to_string(line_no));
                if (!begin_read(p + 1, len - 1, true)) return false;
                rec_off = at; st = 1;
                return true;
            case 1:
                rstrip(p, len);                                          // (Bio's FastqGeneralIterator strips both lines)
                if (!append_bases(p, len)) return false;
                slen = len; st = 2;
                return true;
            case 2:
                if (len == 0 || p[0] != '+') return fail("malformed FASTQ record (no '+' line) at line " + std::to_string(line_no));
                st = 3;
                return true;
            default:
                rstrip(p, len);
                if (len != slen) return fail("malformed FASTQ record (quality length) at line " + std::to_string(line_no));
                st = 0;
                end_read();
                return true;
            }
        case F_FASTA:
            if (len && p[0] == '>') {
                if (st == 1) end_read();
                if (!begin_read(p + 1, len - 1, true)) return false;
                rec_off = at; st = 1;
            } else if (st == 1) {
                strip(p, len);
                if (!append_bases(p, len)) return false;
            }
            return true;
        default: {                                                       // SAM
            if (len == 0 || p[0] == '@') return true;                    // header lines
            const char* f[11]; size_t fl[11]; int nf = 0;
            const char* q = p; const char* const e = p + len;
            while (nf < 11) {
                const char* t = static_cast<const char*>(memchr(q, '\t', (size_t)(e - q)));
                f[nf] = q; fl[nf] = (size_t)((t ? t : e) - q); ++nf;
                if (!t) break;
                q = t + 1;
            }
            if (nf < 11) return fail("malformed SAM record (fewer than 11 fields) at line " + std::to_string(line_no));
            unsigned flag = 0;
            if (fl[1] == 0) return fail("malformed SAM record (FLAG) at line " + std::to_string(line_no));
            for (size_t i = 0; i < fl[1]; ++i) { if (f[1][i] < '0' || f[1][i] > '9') return fail("malformed SAM record (FLAG) at line " + std::to_string(line_no)); flag = flag * 10 + (unsigned)(f[1][i] - '0'); }
            if (skip_secondary && (flag & 0x900u)) return true;          // extract_raw_barcodes.py:144-145
            if (fl[9] == 1 && f[9][0] == '*') { noseq = true; return fail("record without a sequence (SEQ '*') at line " + std::to_string(line_no)); }
            if (!begin_read(f[0], fl[0], false)) return false;
            uint8_t* d = bases_room(fl[9]);
            if (!d) return false;
            for (size_t i = 0; i < fl[9]; ++i) d[i] = sam_map[(uint8_t)f[9][i]];
            c->bases_bytes += fl[9];
            end_read();
            return true;
        }
        }
    }

    // text formats: whole lines of [p, p + n); an unfinished last line waits in `partial`
    bool feed_text(const char* p, size_t n)
    {
        size_t i = 0;
        consumed = 0;
        if (!partial.empty()) {
            const char* nl = static_cast<const char*>(memchr(p, '\n', n));
            if (!nl) { partial.append(p, n); return true; }
            partial.append(p, (size_t)(nl - p));
            i = (size_t)(nl - p) + 1;
            const bool ok = line(partial.data(), partial.size(), 0);
            partial.clear();
            consumed = i;
            if (!ok) return false;
        }
        while (i < n) {
            const char* s = p + i;
            const char* nl = static_cast<const char*>(memchr(s, '\n', n - i));
            if (!nl) { partial.assign(s, n - i); break; }
            const size_t l = (size_t)(nl - s);
            if (!line(s, l, i)) return false;
            i += l + 1;
            consumed = i;
        }
        return true;
    }

    // BAM: magic, l_text, text, n_ref, references, then records of block_size bytes each (SAM specification 4.2)
    bool bam_item(const uint8_t* it)
    {
        auto u32 = [](const uint8_t* q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
        switch (bst) {
        case 0:
            if (memcmp(it, "BAM\1", 4) != 0) return fail("not a BAM file (magic)");
            skip = u32(it + 4); bst = 1; need = 4;
            return true;
        case 1: refs_left = u32(it); bst = refs_left ? 2 : 4; need = 4; return true;
        case 2: { const uint32_t l = u32(it); if (l > (1u << 20)) return fail("malformed BAM header (reference name)"); bst = 3; need = l + 4; return true; }
        case 3: --refs_left; bst = refs_left ? 2 : 4; need = 4; return true;
        case 4: {
            const uint32_t bs = u32(it);
            if (bs < 32 || bs > (1u << 30)) return fail("malformed BAM record (block size)");
            bst = 5; need = bs;
            return true;
        }
        default: {
            const uint32_t bs = need;
            bst = 4; need = 4;
            ++line_no;                                                   // (record number, for messages)
            const uint32_t l_name = it[8], n_cig = (uint32_t)it[12] | ((uint32_t)it[13] << 8), flag = (uint32_t)it[14] | ((uint32_t)it[15] << 8);
            const uint32_t l_seq = u32(it + 16);
            const uint64_t seq_at = 32ull + l_name + 4ull * n_cig;
            if (l_seq > (1u << 30) || seq_at + (l_seq + 1) / 2 + (uint64_t)l_seq > bs) return fail("malformed BAM record " + std::to_string(line_no));
            if (skip_secondary && (flag & 0x900u)) return true;
            if (l_seq == 0) { noseq = true; return fail("record without a sequence (l_seq 0), record " + std::to_string(line_no)); }
            size_t nl = 0;
            while (nl < l_name && it[32 + nl]) ++nl;
            if (!begin_read(reinterpret_cast<const char*>(it) + 32, nl, false)) return false;
            uint8_t* d = bases_room((size_t)l_seq + 2);
            if (!d) return false;
            const uint8_t* s = it + seq_at;
            for (uint32_t k = 0; k < (l_seq + 1) / 2; ++k) { const uint16_t pr = bam_pair[s[k]]; d[2 * k] = (uint8_t)pr; d[2 * k + 1] = (uint8_t)(pr >> 8); }
            c->bases_bytes += l_seq;
            end_read();
            return true;
        }
        }
    }
    bool feed_bam(const uint8_t* p, size_t n)
    {
        size_t i = 0;
        bam_mark = 0; bam_end = n;
        for (;;) {
            if (skip) { const size_t k = (size_t)std::min<uint64_t>(skip, n - i); i += k; skip -= k; if (skip) return true; }
            if (bst == 4 && partial.empty()) bam_mark = i;               // a record's size word starts here
            const uint8_t* it;
            bool from_partial = false;
            if (!partial.empty()) {
                const size_t take = std::min<size_t>((size_t)need - partial.size(), n - i);
                partial.append(reinterpret_cast<const char*>(p) + i, take); i += take;
                if (partial.size() < need) return true;
                it = reinterpret_cast<const uint8_t*>(partial.data()); from_partial = true;
            } else if (n - i >= need) { it = p + i; i += need; }
            else { if (n - i) partial.assign(reinterpret_cast<const char*>(p) + i, n - i); return true; }
            const bool ok = bam_item(it);
            if (from_partial) partial.clear();
            if (!ok) return false;
        }
    }
    bool feed(const uint8_t* p, size_t n)
    {
        if (failed) return false;
        if (n == 0) return true;
        return format == F_BAM ? feed_bam(p, n) : feed_text(reinterpret_cast<const char*>(p), n);
    }
    // end of the input
    bool finish()
    {
        if (failed) return false;
        if (format == F_BAM) {
            if (bst == 4 && partial.empty() && !skip) return true;       // between two records
            return fail(bst == 0 && partial.empty() ? "empty BAM file" : "truncated BAM file");
        }
        if (!partial.empty()) {                                          // last line without a newline
            const bool ok = line(partial.data(), partial.size(), 0);
            partial.clear();
            if (!ok) return false;
        }
        if (format == F_FASTQ && st != 0) {
            return fail(std::string(st == 2 ? "malformed FASTQ record (no '+' line)" : "truncated FASTQ record") + " at line " + std::to_string(line_no));
        }
        if (format == F_FASTA && st == 1) { st = 0; end_read(); }
        return true;
    }
};

// first offset of [0, n) at which a record starts, by the format's local evidence (see the file header); n if none is found
// BAM has no mark to look for: a record start is a place where the fixed fields are possible (block size against the lengths
// it must hold, reference and position >= -1, a printable NUL-terminated name) and where the same holds for the two records
// that would follow.  The assembler proves or refutes the guess like any other.
bool bam_record_plausible(const uint8_t* d, size_t n, size_t i, size_t* next)
{
    if (n - i < 36) return false;
    auto u32 = [&](size_t at) { return (uint32_t)d[at] | ((uint32_t)d[at + 1] << 8) | ((uint32_t)d[at + 2] << 16) | ((uint32_t)d[at + 3] << 24); };
    const uint32_t bs = u32(i);
    if (bs < 32 || bs > (1u << 29)) return false;
    if ((int32_t)u32(i + 4) < -1 || (int32_t)u32(i + 8) < -1 || (int32_t)u32(i + 24) < -1 || (int32_t)u32(i + 28) < -1) return false;
    const uint32_t l_name = d[i + 12], n_cig = (uint32_t)d[i + 16] | ((uint32_t)d[i + 17] << 8), l_seq = u32(i + 20);
    if (l_name == 0 || l_seq > (1u << 29)) return false;
    if (32ull + l_name + 4ull * n_cig + (l_seq + 1) / 2 + (uint64_t)l_seq > bs) return false;
    const size_t name = i + 36;
    if (name + l_name <= n) {
        for (uint32_t k = 0; k + 1 < l_name; ++k) if (d[name + k] < '!' || d[name + k] > '~') return false;
        if (d[name + l_name - 1] != 0) return false;
    }
    *next = i + 4 + (size_t)bs;
    return true;
}
size_t resync_bam(const uint8_t* d, size_t n)
{
    for (size_t i = 0; i + 36 <= n; ++i) {
        size_t a, b, c;
        if (!bam_record_plausible(d, n, i, &a)) continue;
        if (a + 36 <= n) { if (!bam_record_plausible(d, n, a, &b)) continue; if (b + 36 <= n && !bam_record_plausible(d, n, b, &c)) continue; }
        return i;
    }
    return n;
}

size_t resync(int format, const uint8_t* d, size_t n)
{
    if (format == F_BAM) return resync_bam(d, n);
    const char* p = reinterpret_cast<const char*>(d);
    size_t i = 0;
    int tries = 0;
    for (;;) {
        const char* nl = static_cast<const char*>(memchr(p + i, '\n', n - i));
        if (!nl) return n;
        i = (size_t)(nl - p) + 1;
        if (i >= n) return n;
        if (format == F_SAM) return i;
        if (format == F_FASTA) { if (p[i] == '>') return i; continue; }
        if (p[i] != '@') continue;
        // FASTQ: '@' opens a header or a quality line; decide by the three lines behind it
        size_t e[4]; size_t q = i; bool whole = true;
        for (int k = 0; k < 4; ++k) {
            const char* t = static_cast<const char*>(memchr(p + q, '\n', n - q));
            if (!t) { whole = false; break; }
            e[k] = (size_t)(t - p); q = e[k] + 1;
        }
        if (!whole) return n;
        auto len_of = [&](size_t a, size_t b) { size_t l = b - a; while (l && (p[a + l - 1] == '\r' || p[a + l - 1] == ' ' || p[a + l - 1] == '\t')) --l; return l; };
        const size_t s1 = e[0] + 1, s2 = e[1] + 1, s3 = e[2] + 1;
        if (p[s2] == '+' && len_of(s1, e[1]) == len_of(s3, e[3]) && (q >= n || p[q] == '@' || p[q] == '\n' || p[q] == '\r')) return i;
        if (++tries > 64) return n;
    }
}

}  // namespace

// ---- the reader ---------------------------------------------------------------------------------------------------------
struct View { IngestChunk* chunk; uint32_t i0, n; };

struct bdg_ingest : ChunkSink {
    Source src;
    int format = F_FASTQ;
    uint32_t chunk_reads = 100000;
    bool pinned = true, skip_secondary = false;
    unsigned n_workers = 1;
    std::vector<Segment> ring;
    std::vector<IngestChunk*> all_chunks, pool;
    std::mutex mu, claim_mu;
    std::condition_variable cv;
    uint64_t next_seq = 0, asm_seq = 0;          // segments claimed / consumed by the assembler
    bool src_eof = false, stop = false, done = false, failed = false, nomem = false, noseq = false, sequential = false;
    std::string err;
    std::deque<View> ready;
    std::vector<View> held;                      // views the consumer holds, by id
    std::vector<std::thread> workers;
    std::thread assembler;
    Parser seqp;                                 // the assembler's sequential parser
    uint64_t total_reads = 0;

    IngestChunk* pool_get()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !pool.empty(); });
        if (stop) return nullptr;
        IngestChunk* c = pool.back(); pool.pop_back();
        c->n = 0; c->bases_bytes = 0; c->ids_bytes = 0; c->views = 0;
        return c;
    }
    void pool_put(IngestChunk* c)
    {
        { std::lock_guard<std::mutex> lk(mu); pool.push_back(c); }
        cv.notify_all();
    }
    bool chunk_prepare(IngestChunk* c, size_t reads, size_t bases, size_t ids)
    {
        const bool ok = grow(c->off, c->off_cap, 0, reads + 2, false) && grow(c->id_off, c->id_off_cap, 0, reads + 2, false) &&
                        grow(c->bases, c->bases_cap, 0, bases + 64, pinned) && grow(c->ids, c->ids_cap, 0, ids + 64, false);
        if (ok) { c->off[0] = 0; c->id_off[0] = 0; }
        return ok;
    }
    // a finished chunk -> views of at most chunk_reads reads for the consumer
    void emit(IngestChunk* c)
    {
        if (!c) return;
        if (c->n == 0) { pool_put(c); return; }
        {
            std::lock_guard<std::mutex> lk(mu);
            total_reads += c->n;
            for (uint32_t i0 = 0; i0 < c->n; i0 += chunk_reads) {
                ready.push_back(View{ c, i0, std::min(chunk_reads, c->n - i0) });
                ++c->views;
            }
        }
        cv.notify_all();
    }
    // sequential mode: the parser's chunk is full
    IngestChunk* chunk_full(IngestChunk* full) override
    {
        emit(full);
        IngestChunk* c = pool_get();
        if (c && !chunk_prepare(c, chunk_reads, (size_t)chunk_reads * 64, (size_t)chunk_reads * 16)) { seqp.oom(); }
        if (!c) seqp.fail("reader stopped");                         // (closing: the parse is abandoned)
        return c ? c : full;
    }

    void worker_loop()
    {
        Inflater inf;
        Parser wp;
        wp.init(format, pinned, skip_secondary, 0, nullptr);
        const bool can_parse = true;
        for (;;) {
            IngestChunk* c = nullptr;
            bool want_parse;
            { std::lock_guard<std::mutex> lk(mu); want_parse = can_parse && !sequential; }
            if (want_parse && !(c = pool_get())) return;
            Segment* s = nullptr;
            {
                std::lock_guard<std::mutex> ck(claim_mu);
                uint64_t seq;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || src_eof || next_seq - asm_seq < ring.size(); });
                    if (stop || src_eof) { lk.unlock(); if (c) pool_put(c); return; }
                    seq = next_seq;
                    s = &ring[seq % ring.size()];
                    s->state = 1;
                }
                s->seq = seq; s->parsed = false; s->bad = false; s->chunk = nullptr; s->lines = 0;
                const bool got = src.claim(*s);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (got) ++next_seq; else { src_eof = true; s->state = 0; }
                }
                if (!got) { cv.notify_all(); if (c) pool_put(c); return; }
            }
            Source::materialise(*s, inf);
            if (c && !s->failed) {
                // parse the segment's own records
                const auto T0 = std::chrono::steady_clock::now();
                const size_t first = s->first ? 0 : resync(format, s->data, s->len);
                s->head_len = first; s->tail_off = s->len;
                if (first < s->len) {
                    const size_t span = s->len - first;
                    // (sized by the segment size, not by this segment's span: the same request every time, so a buffer that served one
                    // segment serves them all - spans differ by a few bytes and each larger one would re-pin the buffer)
                    const size_t room = std::max(span, std::min(src.seg_bytes, src.compressed ? src.seg_bytes : src.size));
                    const bool ok = chunk_prepare(c, room / 512 + 1024, format == F_FASTQ ? room / 2 + 65536 : room + 65536, room / 32 + 4096);
                    wp.c = c; wp.st = 0; wp.partial.clear(); wp.failed = false; wp.nomem = false; wp.noseq = false; wp.err.clear(); wp.line_no = 0;
                    wp.bst = s->first ? 0 : 4; wp.need = s->first ? 8 : 4; wp.skip = 0; wp.refs_left = 0;      // (BAM: the header, or between two records)
                    if (!ok || !wp.feed(s->data + first, span)) s->bad = true;            // the assembler parses it again, in sequence, and reports
                    else if (format == F_BAM) {
                        if (wp.bst < 4) s->bad = true;                                   // (a header longer than a segment: left to the assembler)
                        else {
                            // what the last record in progress has not finished goes back: from its size word on
                            s->tail_off = first + ((wp.bst == 5 || !wp.partial.empty()) ? wp.bam_mark : wp.bam_end);
                            s->lines = wp.line_no;                                        // (records, for messages)
                        }
                        wp.partial.clear();
                    } else {
                        if (wp.record_open()) { wp.rollback_open(); s->tail_off = first + wp.rec_off; }
                        else s->tail_off = first + wp.consumed;
                        // lines inside [head_len, tail_off) = whole lines seen - whole lines of the tail
                        uint64_t nl = 0;
                        const uint8_t* a = s->data + s->tail_off; const uint8_t* const e = s->data + first + wp.consumed;
                        while (a < e) { const void* t = memchr(a, '\n', (size_t)(e - a)); if (!t) break; ++nl; a = static_cast<const uint8_t*>(t) + 1; }
                        s->lines = wp.line_no - nl;
                    }
                    s->parsed = true;
                    if (getenv("BADGER_AMD_INGEST_DEBUG")) fprintf(stderr, "ingest: segment %llu parsed in %.3f s (%u reads, head %zu)\n", (unsigned long long)s->seq,
                                                                   std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count(), c->n, s->head_len);
                    s->chunk = c; c = nullptr;
                }
            }
            if (c) pool_put(c);
            { std::lock_guard<std::mutex> lk(mu); s->state = 2; }
            cv.notify_all();
        }
    }

    void set_failed(const Parser& p)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) { failed = true; err = p.err; nomem = p.nomem; noseq = p.noseq; }
    }
    void set_failed(const std::string& m)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) { failed = true; err = m; }
    }

    void assemble()
    {
        seqp.init(format, pinned, skip_secondary, 0, this);
        bool seq_mode = false;
        // (seqp.c is the chunk the sequential parser appends to: on the fast path the previous segment's chunk, which so
        // receives the record cut by the segment boundary)
        auto fresh = [&]() -> IngestChunk* {
            IngestChunk* c = pool_get();
            if (c && !chunk_prepare(c, 1024, 1 << 16, 1 << 12)) { seqp.oom(); pool_put(c); return nullptr; }
            return c;
        };
        auto enter_sequential = [&]() -> bool {
            // whatever the fast path has finished is handed out; an open record moves to a fresh chunk
            { std::lock_guard<std::mutex> lk(mu); sequential = true; }
            seq_mode = true;
            seqp.limit = chunk_reads;
            IngestChunk* old = seqp.c;
            IngestChunk* nc = fresh();
            if (!nc) return false;
            if (old && seqp.record_open()) { if (!seqp.move_open_record(nc)) { pool_put(nc); return false; } }
            else seqp.c = nc;
            emit(old);
            return true;
        };
        bool ok = true;
        if (seq_mode) { seqp.limit = chunk_reads; std::lock_guard<std::mutex> lk(mu); sequential = true; }
        if (!(seqp.c = fresh())) ok = false;
        while (ok) {
            Segment* s;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || (asm_seq < next_seq && ring[asm_seq % ring.size()].state == 2) || (src_eof && asm_seq == next_seq); });
                if (stop) { ok = false; break; }
                if (asm_seq == next_seq) break;
                s = &ring[asm_seq % ring.size()];
            }
            if (s->failed || seq_mode || !s->parsed) {
                if (s->chunk) { pool_put(s->chunk); s->chunk = nullptr; }
                ok = seqp.feed(s->data, s->len);
                // (text before a damaged spot is still parsed: a sequential reader would have delivered it)
                if (ok && s->failed) { seqp.fail("read error: " + s->err); ok = false; }
                // A segment no reader could start in (a record longer than a segment, wrapped FASTQ lines, ...) went into the
                // chunk in front of it; where that happens segment after segment the chunk must not grow without bound.
                if (ok && !seq_mode && seqp.c->n >= chunk_reads) {
                    IngestChunk* old = seqp.c;
                    IngestChunk* nc = fresh();
                    if (!nc) ok = false;
                    else if (seqp.record_open()) { if (!seqp.move_open_record(nc)) { pool_put(nc); ok = false; } }
                    else seqp.c = nc;
                    if (ok) emit(old);
                }
            } else {
                ok = seqp.feed(s->data, s->head_len);
                if (ok && s->head_len < s->len) {
                    // (BAM: between two records - or nothing read yet and this is the file's first segment, whose reader took the header)
                    const bool boundary = seqp.partial.empty() && (format != F_FASTQ || seqp.st == 0) &&
                                          (format != F_BAM || (seqp.bst == 4 && !seqp.skip) || (s->first && seqp.bst == 0));
                    if (!boundary || s->bad) {
                        if (getenv("BADGER_AMD_INGEST_DEBUG")) fprintf(stderr, "ingest: segment %llu leaves the fast path (boundary %d, bad %d, st %d, partial %zu, head %zu, tail %zu, len %zu)\n",
                                                                       (unsigned long long)s->seq, (int)boundary, (int)s->bad, seqp.st, seqp.partial.size(), s->head_len, s->tail_off, s->len);
                        pool_put(s->chunk); s->chunk = nullptr;
                        ok = enter_sequential() && seqp.feed(s->data + s->head_len, s->len - s->head_len);
                    } else {
                        if (format == F_FASTA && seqp.st == 1) { seqp.st = 0; seqp.end_read(); }
                        emit(seqp.c);
                        seqp.c = s->chunk; s->chunk = nullptr;
                        seqp.line_no += s->lines;
                        if (format == F_BAM) { seqp.bst = 4; seqp.need = 4; seqp.skip = 0; seqp.refs_left = 0; }
                        ok = seqp.feed(s->data + s->tail_off, s->len - s->tail_off);
                    }
                }
            }
            if (!ok) break;
            src.done_with(*s);
            { std::lock_guard<std::mutex> lk(mu); s->state = 0; ++asm_seq; }
            cv.notify_all();
        }
        if (ok) ok = seqp.finish();
        if (ok) emit(seqp.c);
        else {
            // every whole read in front of the failure is delivered (the reference's loop has processed them by then)
            if (seqp.c) { if (seqp.record_open()) seqp.rollback_open(); emit(seqp.c); }
            if (seqp.failed) set_failed(seqp); else set_failed("reader stopped");
        }
        seqp.c = nullptr;
        { std::lock_guard<std::mutex> lk(mu); done = true; }
        cv.notify_all();
    }
};

namespace {

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

int bdg_ingest_open_mt(const char* path, uint32_t chunk_reads, uint32_t ring_chunks, int pinned, uint32_t threads, bdg_ingest** out)
{
    bdg_ingest_opts o;
    memset(&o, 0, sizeof(o));
    o.chunk_reads = chunk_reads; o.ring_chunks = ring_chunks; o.pinned = pinned; o.threads = threads;
    return bdg_ingest_open_ex(path, &o, out);
}

int bdg_ingest_open_ex(const char* path, const bdg_ingest_opts* o, bdg_ingest** out)
{
    if (!out) return BDG_E_ARG;
    *out = nullptr;
    if (!path || !o || o->chunk_reads == 0) return BDG_E_ARG;
    std::string name(path);
    if (ends_with_ci(name, ".gz")) name.resize(name.size() - 3);
    else if (ends_with_ci(name, ".gzip")) name.resize(name.size() - 5);
    int format;
    if (ends_with_ci(name, ".fq") || ends_with_ci(name, ".fastq")) format = F_FASTQ;
    else if (ends_with_ci(name, ".fa") || ends_with_ci(name, ".fasta")) format = F_FASTA;
    else if (ends_with_ci(name, ".bam")) format = F_BAM;
    else if (ends_with_ci(name, ".sam")) format = F_SAM;
    else return BDG_E_ARG;                                   // unknown extension (extract_raw_barcodes.py:196-197)
    bdg_ingest* g = new bdg_ingest();
    std::string err;
    if (!g->src.open(path, err)) { delete g; return BDG_E_ARG; }
    if (o->segment_bytes) g->src.seg_bytes = std::max<uint64_t>(o->segment_bytes, 64);
    else if (const char* e = getenv("BADGER_AMD_SEGMENT_MB")) { const long mb = atol(e); if (mb > 0 && mb <= 4096) g->src.seg_bytes = (size_t)mb << 20; }
    g->src.bgzf_parallel = o->threads != 1;                  // 1: every compressed input as one sequential gzip stream (what gzip.open does)
    g->src.gz_threads = o->threads ? o->threads : std::min(12u, std::max(1u, std::thread::hardware_concurrency()));
    // pysam opens BAM and SAM by content; so does this, as far as the first bytes of an uncompressed file tell
    if (!g->src.compressed && (format == F_SAM || format == F_BAM))
        format = g->src.size >= 4 && memcmp(g->src.map, "BAM\1", 4) == 0 ? F_BAM : F_SAM;
    g->format = format; g->chunk_reads = o->chunk_reads; g->pinned = o->pinned != 0; g->skip_secondary = o->skip_secondary != 0;
    unsigned t = o->threads ? o->threads : std::min(12u, std::max(1u, std::thread::hardware_concurrency()));
    if (t > 64) t = 64;
    g->n_workers = t;
    g->ring.resize((size_t)t + 2);
    const size_t hold = std::max<uint32_t>(o->ring_chunks, 2);
    const size_t n_chunks = hold + g->ring.size() + t + 2;
    for (size_t i = 0; i < n_chunks; ++i) { g->all_chunks.push_back(new IngestChunk()); g->pool.push_back(g->all_chunks.back()); }
    g->held.resize(hold + 64, View{ nullptr, 0, 0 });
    for (unsigned i = 0; i < t; ++i) g->workers.emplace_back(&bdg_ingest::worker_loop, g);
    g->assembler = std::thread(&bdg_ingest::assemble, g);
    *out = g;
    return BDG_OK;
}

int bdg_ingest_next(bdg_ingest* g, bdg_ingest_chunk* out)
{
    if (!g || !out) return BDG_E_ARG;
    memset(out, 0, sizeof(*out));
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [&] { return !g->ready.empty() || g->done; });
    if (g->ready.empty()) {
        if (g->failed) return g->nomem ? BDG_E_NOMEM : (g->noseq ? BDG_E_NOSEQ : BDG_E_FORMAT);     // chunks before the failure were good
        return BDG_OK;                                                                               // end of input: n = 0
    }
    size_t id = 0;
    while (id < g->held.size() && g->held[id].chunk) ++id;
    if (id == g->held.size()) return BDG_E_ARG;                                                      // more chunks held than ring_chunks allows
    const View v = g->ready.front(); g->ready.pop_front();
    g->held[id] = v;
    const IngestChunk& c = *v.chunk;
    out->id = (uint32_t)id;
    out->n = v.n; out->bases = c.bases; out->off = c.off + v.i0; out->total_bytes = c.off[v.i0 + v.n] - c.off[v.i0];
    out->ids = c.ids; out->id_off = c.id_off + v.i0;
    return BDG_OK;
}

int bdg_ingest_release(bdg_ingest* g, uint32_t id)
{
    if (!g || id >= g->held.size()) return BDG_E_ARG;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        View& v = g->held[id];
        if (v.chunk) {
            if (--v.chunk->views == 0) g->pool.push_back(v.chunk);
            v.chunk = nullptr;
        }
    }
    g->cv.notify_all();
    return BDG_OK;
}

const char* bdg_ingest_error(bdg_ingest* g) { return g ? g->err.c_str() : "no reader"; }

uint64_t bdg_ingest_reads(bdg_ingest* g)
{
    if (!g) return 0;
    std::lock_guard<std::mutex> lk(g->mu);
    return g->total_reads;
}

void bdg_ingest_close(bdg_ingest* g)
{
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->stop = true;
    }
    g->cv.notify_all();
    for (auto& w : g->workers) if (w.joinable()) w.join();
    if (g->assembler.joinable()) g->assembler.join();
    if (getenv("BADGER_AMD_INGEST_DEBUG")) {
        PinnedCache& pc = pinned_cache();
        std::lock_guard<std::mutex> lk(pc.mu);
        fprintf(stderr, "ingest: %llu pinned allocations, %.1f MB, %.3f s inside hipHostMalloc (summed over threads); %u reader threads, %zu segments\n",
                (unsigned long long)pc.allocs, pc.alloc_bytes / 1e6, pc.alloc_s, g->n_workers, (size_t)g->next_seq);
    }
    for (IngestChunk* c : g->all_chunks) {
        pinned_free(c->bases, g->pinned);
        free(c->off); free(c->ids); free(c->id_off);
        delete c;
    }
    delete g;
}

}  // extern "C"
