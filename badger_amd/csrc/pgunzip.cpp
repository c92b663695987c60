// Parallel inflate of one plain gzip stream: see pgunzip.hpp.  Deflate as in RFC 1951, gzip framing as in RFC 1952; the
// decoder below is written for this file (16-bit output symbols, start at any bit, one block at a time), zlib only
// supplies crc32().
#include "pgunzip.hpp"

#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t WIN = 32768;
constexpr uint16_t MARK = 256;              // symbol >= MARK: the byte at place (symbol - MARK) of the 32 KiB before the chunk
constexpr int FAST = 10;                    // codes of up to FAST bits are decoded by one table look-up

enum { PG_OK = 0, PG_FINAL = 1, PG_LIMIT = 2, PG_DATA = -1, PG_EOF = -2, PG_NOTTEXT = -3, PG_NOMEM = -4 };

// ---- bits, least significant first ---------------------------------------------------------------------------------------
struct Bits {
    const uint8_t* base; const uint8_t* p; const uint8_t* end;
    uint64_t buf = 0; unsigned cnt = 0;
    Bits(const uint8_t* b, size_t n, size_t bit) : base(b), p(b + (bit >> 3)), end(b + n)
    {
        refill();
        const unsigned s = (unsigned)(bit & 7);
        if (s) { if (cnt >= s) { buf >>= s; cnt -= s; } else { buf = 0; cnt = 0; } }
    }
    inline void refill()
    {
        if (end - p >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56 && p < end) { buf |= (uint64_t)*p++ << cnt; cnt += 8; }
        }
    }
    inline uint32_t peek(unsigned n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void drop(unsigned n) { buf >>= n; cnt -= n; }
    // n <= 32 bits; false: the input ends first
    inline bool get(unsigned n, uint32_t& v)
    {
        if (cnt < n) { refill(); if (cnt < n) return false; }
        v = peek(n); drop(n);
        return true;
    }
    size_t bit_pos() const { return (size_t)(p - base) * 8 - cnt; }
    void align_byte() { drop(cnt & 7); }
};

// CRC-32 of a piece: libdeflate's (carry-less multiplication, several GB/s) when the system has the library, else zlib's
typedef uint32_t (*crc_fn)(uint32_t, const void*, size_t);
crc_fn fast_crc()
{
    static const crc_fn f = [] {
        if (getenv("BADGER_AMD_NO_LIBDEFLATE")) return (crc_fn) nullptr;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        return h ? reinterpret_cast<crc_fn>(dlsym(h, "libdeflate_crc32")) : (crc_fn) nullptr;
    }();
    return f;
}
uint32_t crc_of(const uint8_t* d, size_t n)
{
    if (const crc_fn f = fast_crc()) return f(0, d, n);
    uint32_t c = 0;
    for (size_t k = 0; k < n; k += size_t(1) << 30) c = (uint32_t)crc32(c, d + k, (uInt)std::min<size_t>(n - k, size_t(1) << 30));
    return c;
}

const uint16_t LEN_BASE[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t LEN_EXTRA[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t DIST_BASE[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193,
                                 12289, 16385, 24577 };
const uint8_t DIST_EXTRA[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

#define LEN_BASE_OF(i) LEN_BASE[i]
#define LEN_EXTRA_OF(i) LEN_EXTRA[i]
#define DIST_BASE_OF(i) DIST_BASE[i]
#define DIST_EXTRA_OF(i) DIST_EXTRA[i]

// ---- a canonical Huffman code -------------------------------------------------------------------------------------------
struct Huff {
    uint16_t fast[1 << FAST];               // (symbol << 4) | length, 0: longer than FAST bits (or no such code)
    uint16_t count[16], symbol[288];
    int n_codes = 0, max_len = 0;
    // the same look-up with what the hot loop needs next already in the entry (pack()): bits 0-3 code length, 4-7 extra bits,
    // 8-9 kind (0 literal, 1 length or distance, 2 end of block; 3 = not in the table: the careful loop takes over), 16-31
    // the literal / base length / base distance
    uint32_t packed[1 << FAST];
    void pack(bool lengths)
    {
        for (uint32_t x = 0; x < (1u << FAST); ++x) {
            const uint16_t e = fast[x];
            if (!e) { packed[x] = 3u << 8; continue; }
            const uint32_t sym = e >> 4, l = e & 15u;
            if (!lengths) packed[x] = sym > 29 ? 3u << 8 : l | (uint32_t)DIST_EXTRA_OF(sym) << 4 | 1u << 8 | (uint32_t)DIST_BASE_OF(sym) << 16;
            else if (sym < 256) packed[x] = l | sym << 16;
            else if (sym == 256) packed[x] = l | 2u << 8;
            else if (sym > 285) packed[x] = 3u << 8;
            else packed[x] = l | (uint32_t)LEN_EXTRA_OF(sym - 257) << 4 | 1u << 8 | (uint32_t)LEN_BASE_OF(sym - 257) << 16;
        }
    }
    // 0: complete; > 0: incomplete (unused code space); < 0: over-subscribed
    int build(const uint8_t* lens, int n)
    {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[lens[i]]++;
        n_codes = n - count[0];
        count[0] = 0;
        max_len = 0;
        for (int l = 15; l >= 1; --l) if (count[l]) { max_len = l; break; }
        int left = 1;
        for (int l = 1; l <= 15; ++l) { left <<= 1; left -= count[l]; if (left < 0) return left; }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i) if (lens[i]) symbol[offs[lens[i]]++] = (uint16_t)i;
        // the short codes, bit-reversed (the stream holds a code's most significant bit first)
        memset(fast, 0, sizeof(fast));
        uint32_t code = 0; int idx = 0;
        for (int l = 1; l <= 15; ++l) {
            for (int k = 0; k < count[l]; ++k, ++code, ++idx) {
                if (l > FAST) continue;
                uint32_t rev = 0;
                for (int b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
                const uint16_t e = (uint16_t)(symbol[idx] << 4 | l);
                for (uint32_t x = rev; x < (1u << FAST); x += 1u << l) fast[x] = e;
            }
            code <<= 1;
        }
        return left;
    }
    // the next symbol; < 0: PG_DATA (no such code) or PG_EOF
    inline int decode(Bits& b) const
    {
        if (b.cnt < 15) b.refill();
        const uint16_t e = fast[b.buf & ((1u << FAST) - 1)];
        if (e) {
            const unsigned l = e & 15u;
            if (b.cnt < l) return PG_EOF;
            b.drop(l);
            return e >> 4;
        }
        int code = 0, first = 0, index = 0;
        uint64_t bits = b.buf;
        for (int l = 1; l <= max_len; ++l) {
            if ((unsigned)l > b.cnt) return PG_EOF;
            code |= (int)(bits & 1); bits >>= 1;
            const int c = count[l];
            if (code - c < first) { b.drop((unsigned)l); return symbol[index + (code - first)]; }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return b.cnt < 15 && b.p >= b.end ? PG_EOF : PG_DATA;
    }
};

struct FixedCodes {
    Huff lit, dist;
    FixedCodes()
    {
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        lit.build(l, 288);
        uint8_t d[30];
        for (int i = 0; i < 30; ++i) d[i] = 5;
        dist.build(d, 30);
        lit.pack(true); dist.pack(false);
    }
};
const FixedCodes& fixed_codes() { static const FixedCodes f; return f; }

// ---- output: 16-bit symbols behind a 32 KiB window ----------------------------------------------------------------------
struct Out {
    uint16_t* d = nullptr; size_t n = 0, cap = 0;       // d[0 .. WIN) is the window in front of the data
    ~Out() { free(d); }
    bool reserve(size_t want)
    {
        if (want <= cap) return true;
        size_t c = cap ? cap : (size_t(1) << 20);
        while (c < want) c += c / 2 + 4096;
        uint16_t* nd = static_cast<uint16_t*>(realloc(d, c * sizeof(uint16_t)));
        if (!nd) return false;
        d = nd; cap = c;
        return true;
    }
    bool start_unknown()                                 // the window holds "the byte at place i"
    {
        if (!reserve(WIN + (size_t(1) << 20))) return false;
        for (uint32_t i = 0; i < WIN; ++i) d[i] = (uint16_t)(MARK + i);
        n = WIN;
        return true;
    }
    bool start_known(const uint8_t* win)                 // the window holds these bytes (oldest first)
    {
        if (!reserve(WIN + (size_t(1) << 16))) return false;
        for (uint32_t i = 0; i < WIN; ++i) d[i] = win[i];
        n = WIN;
        return true;
    }
    void steal(Out& o) { free(d); d = o.d; n = o.n; cap = o.cap; o.d = nullptr; o.n = o.cap = 0; }
};

inline bool is_text(uint32_t c) { return (c >= 32 && c < 127) || c == '\n' || c == '\r' || c == '\t'; }

// the header of a dynamic block behind its 3 type bits: the two codes
int read_dynamic(Bits& b, Huff& hl, Huff& hd)
{
    static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    uint32_t v;
    if (!b.get(14, v)) return PG_EOF;
    const int hlit = (int)(v & 31) + 257, hdist = (int)((v >> 5) & 31) + 1, hclen = (int)(v >> 10) + 4;
    if (hlit > 286 || hdist > 30) return PG_DATA;
    uint8_t cl[19];
    memset(cl, 0, sizeof(cl));
    for (int i = 0; i < hclen; ++i) { if (!b.get(3, v)) return PG_EOF; cl[order[i]] = (uint8_t)v; }
    Huff hc;
    if (hc.build(cl, 19) != 0) return PG_DATA;            // (zlib: "invalid code lengths set")
    uint8_t lens[286 + 30];
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
        const int s = hc.decode(b);
        if (s < 0) return s;
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        int rep; uint8_t val = 0;
        if (s == 16) { if (i == 0) return PG_DATA; val = lens[i - 1]; if (!b.get(2, v)) return PG_EOF; rep = 3 + (int)v; }
        else if (s == 17) { if (!b.get(3, v)) return PG_EOF; rep = 3 + (int)v; }
        else { if (!b.get(7, v)) return PG_EOF; rep = 11 + (int)v; }
        if (i + rep > total) return PG_DATA;
        while (rep--) lens[i++] = val;
    }
    if (lens[256] == 0) return PG_DATA;                   // no end-of-block code
    int left = hl.build(lens, hlit);
    if (left < 0 || (left > 0 && !(hl.n_codes == 1 && hl.max_len == 1))) return PG_DATA;
    left = hd.build(lens + hlit, hdist);
    if (left < 0 || (left > 0 && hd.max_len > 1)) return PG_DATA;      // (one distance code of one bit, or none, may leave the code incomplete: zlib's rule)
    return PG_OK;
}

// One block at b -> o.  hist: how many symbols in front of o.d[WIN] are history a distance may reach (WIN when the window is
// full or unknown, fewer at the start of a member).  text_only: a literal that is not text ends the attempt (the block finder).
int inflate_block(Bits& b, Out& o, size_t hist, bool text_only)
{
    uint32_t v;
    if (!b.get(3, v)) return PG_EOF;
    const bool final = v & 1;
    const uint32_t type = v >> 1;
    if (type == 3) return PG_DATA;
    if (type == 0) {
        b.align_byte();
        if (!b.get(32, v)) return PG_EOF;
        const uint32_t len = v & 0xFFFF;
        if ((len ^ (v >> 16)) != 0xFFFF) return PG_DATA;
        if (!o.reserve(o.n + len)) return PG_NOMEM;
        // (the buffered bits are whole bytes now)
        uint32_t i = 0;
        for (; i < len && b.cnt >= 8; ++i) { o.d[o.n + i] = (uint16_t)(b.buf & 0xFF); b.drop(8); }
        const size_t rest = len - i;
        if ((size_t)(b.end - b.p) < rest) return PG_EOF;
        for (size_t k = 0; k < rest; ++k) o.d[o.n + i + k] = b.p[k];
        b.p += rest;
        if (rest || b.cnt == 0) { b.buf = 0; b.cnt = 0; }      // (bits loaded ahead of cnt belonged to the bytes just passed)
        if (text_only) for (uint32_t k = 0; k < len; ++k) if (!is_text(o.d[o.n + k])) return PG_NOTTEXT;
        o.n += len;
        return final ? PG_FINAL : PG_OK;
    }
    Huff dl, dd;                                         // (a block's own codes, on the stack)
    const Huff* hl; const Huff* hd;
    if (type == 1) { hl = &fixed_codes().lit; hd = &fixed_codes().dist; }
    else { const int rc = read_dynamic(b, dl, dd); if (rc != PG_OK) return rc; dl.pack(true); dd.pack(false); hl = &dl; hd = &dd; }
    // The hot loop: while at least 16 input bytes lie ahead, one refill (>= 56 bits) covers two literals or a whole
    // length / distance pair (15 + 5 + 15 + 13 bits); entries say what follows without a second table; copies go eight symbols
    // at a time.  Anything unusual (a code longer than the table's 10 bits, the end of the input, a bad symbol) leaves it for
    // the careful loop below, which decides.
    constexpr uint32_t FM = (1u << FAST) - 1u;
    while (b.end - b.p >= 16) {
        if (o.n + 600 > o.cap && !o.reserve(o.n + (size_t(1) << 20))) return PG_NOMEM;
        b.refill();
        uint32_t e = hl->packed[b.buf & FM];
        uint32_t kind = (e >> 8) & 3u;
        if (kind == 0) {
            const uint32_t c0 = e >> 16;
            if (text_only && !is_text(c0)) return PG_NOTTEXT;
            o.d[o.n++] = (uint16_t)c0;
            b.drop(e & 15u);
            e = hl->packed[b.buf & FM];                  // a second literal out of the same refill, more often than not
            kind = (e >> 8) & 3u;
            if (kind == 0) {
                const uint32_t c1 = e >> 16;
                if (text_only && !is_text(c1)) return PG_NOTTEXT;
                o.d[o.n++] = (uint16_t)c1;
                b.drop(e & 15u);
                continue;
            }
            if (kind != 1) { if (kind == 2) { b.drop(e & 15u); return final ? PG_FINAL : PG_OK; } break; }
            if (b.cnt < 48) b.refill();
        }
        else if (kind == 2) { b.drop(e & 15u); return final ? PG_FINAL : PG_OK; }
        else if (kind == 3) break;
        // a length, then its distance
        const uint64_t keep_buf = b.buf; const unsigned keep_cnt = b.cnt; const uint8_t* const keep_p = b.p;
        b.drop(e & 15u);
        const uint32_t eb = (e >> 4) & 15u;
        const uint32_t len = (e >> 16) + (uint32_t)(b.buf & ((1u << eb) - 1u));
        b.drop(eb);
        const uint32_t f = hd->packed[b.buf & FM];
        if (((f >> 8) & 3u) != 1u) { b.buf = keep_buf; b.cnt = keep_cnt; b.p = keep_p; break; }        // (back in front of the length)
        b.drop(f & 15u);
        const uint32_t fb = (f >> 4) & 15u;
        const uint32_t dist = (f >> 16) + (uint32_t)(b.buf & ((1u << fb) - 1u));
        b.drop(fb);
        if (dist > (o.n - WIN) + hist) return PG_DATA;
        uint16_t* dst = o.d + o.n;
        const uint16_t* src = dst - dist;
        o.n += len;
        if (dist >= 8) {                                 // (eight symbols = 16 bytes at a time; may write up to 7 symbols past the end: room is there)
            for (uint32_t k = 0; k < len; k += 8) memcpy(dst + k, src + k, 16);
        } else if (dist == 1) {
            const uint16_t v1 = src[0];
            for (uint32_t k = 0; k < len; ++k) dst[k] = v1;
        } else {
            for (uint32_t k = 0; k < len; ++k) dst[k] = src[k];
        }
    }
    for (;;) {
        if (o.n + 260 > o.cap && !o.reserve(o.n + (size_t(1) << 20))) return PG_NOMEM;
        const int s = hl->decode(b);
        if (s < 0) return s;
        if (s < 256) {
            if (text_only && !is_text((uint32_t)s)) return PG_NOTTEXT;
            o.d[o.n++] = (uint16_t)s;
            continue;
        }
        if (s == 256) return final ? PG_FINAL : PG_OK;
        if (s > 285) return PG_DATA;
        uint32_t len = LEN_BASE[s - 257];
        if (LEN_EXTRA[s - 257]) { if (!b.get(LEN_EXTRA[s - 257], v)) return PG_EOF; len += v; }
        const int ds = hd->decode(b);
        if (ds < 0) return ds;
        if (ds > 29) return PG_DATA;
        uint32_t dist = DIST_BASE[ds];
        if (DIST_EXTRA[ds]) { if (!b.get(DIST_EXTRA[ds], v)) return PG_EOF; dist += v; }
        if (dist > (o.n - WIN) + hist) return PG_DATA;    // (zlib: "invalid distance too far back")
        uint16_t* const dst = o.d + o.n;
        const uint16_t* const src = dst - dist;
        if (dist >= len) memcpy(dst, src, len * sizeof(uint16_t));
        else for (uint32_t k = 0; k < len; ++k) dst[k] = src[k];
        o.n += len;
    }
}

// the gzip member header at p: its length, 0 if there is none, (size_t)-1 if it is cut off
size_t gzip_header(const uint8_t* p, size_t n)
{
    if (n < 10) return n >= 2 && p[0] == 0x1f && p[1] == 0x8b ? (size_t)-1 : 0;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    const uint8_t flg = p[3];
    size_t at = 10;
    if (flg & 4) { if (at + 2 > n) return (size_t)-1; at += 2 + (size_t)(p[at] | p[at + 1] << 8); if (at > n) return (size_t)-1; }
    for (int k = 0; k < 2; ++k)
        if (flg & (k ? 16 : 8)) { while (at < n && p[at]) ++at; if (at >= n) return (size_t)-1; ++at; }
    if (flg & 2) at += 2;
    return at > n ? (size_t)-1 : at;
}

// symbols a speculative chunk may hold: 64 MB of 16-bit symbols (BADGER_AMD_GUNZIP_MAX_CHUNK_KSYM: thousands, for tests)
static const size_t MAX_CHUNK_SYMBOLS = [] { const char* e = getenv("BADGER_AMD_GUNZIP_MAX_CHUNK_KSYM"); const long v = e ? atol(e) : 0; return v > 0 ? (size_t)v << 10 : size_t(32) << 20; }();

struct Chunk {
    size_t lo_bit = 0, hi_bit = 0;           // where its search starts / where it stops taking new blocks
    bool exact = false;                      // lo_bit is the start of the stream: no search, no unknown window
    int state = 0;                           // 0 to do, 1 in work, 2 done
    bool found = false, final = false;
    int stop = PG_OK;                         // why it stopped short of hi_bit, if it did
    size_t start_bit = 0, end_bit = 0;
    Out out;
};

}  // namespace

// A stretch of output in stream order: symbols of an accepted chunk that still have to become bytes (a worker does that,
// against a copy of the window as it stood in front of them), or bytes the chain inflated itself.
struct Piece {
    std::unique_ptr<Chunk> chunk;            // symbols (chunk->out), if not resolved yet
    std::vector<uint8_t> bytes;
    std::unique_ptr<uint8_t[]> win;          // the 32 KiB in front of the symbols
    int state = 0;                           // 0 symbols waiting, 1 being resolved, 2 bytes ready
    uint32_t crc = 0;                        // of bytes
    bool last = false;                       // the member ends behind this piece
};

struct PGunzipImpl {
    const uint8_t* in; size_t n_in;
    size_t chunk_bytes; unsigned nthreads;
    std::vector<std::unique_ptr<Chunk>> chunks;
    std::vector<std::thread> workers;
    std::mutex mu; std::condition_variable cv_work, cv_read;
    size_t next_take = 0;                    // next chunk to inflate
    size_t next_chain = 0;                   // next chunk the chain looks at
    size_t ahead = 0, max_pieces = 0;
    bool quit = false, chaining = false;
    std::atomic<bool> stop_all{ false };
    std::deque<std::unique_ptr<Piece>> pieces;           // validated output in stream order (front = next to read)
    // the chain: where the accepted data ends, the window behind it (only the thread that holds `chaining` touches these)
    size_t cur_bit = 0, total_out = 0;
    uint8_t win[WIN];
    bool chain_done = false;                 // the member's end was reached, or an error
    bool fail = false; std::string err;      // (set by the chain, read by the reader once the pieces before it are gone)
    uint32_t want_crc = 0; size_t used_bytes = 0;
    // the reader
    uint32_t crc = 0; size_t front_at = 0;
    bool member_end = false, reader_fail = false;
    size_t st_accepted = 0, st_seq_blocks = 0, st_seq_bytes = 0;

    // ---- workers ------------------------------------------------------------------------------------------------------
    Piece* resolvable()                       // (mu held)
    {
        for (auto& pc : pieces) if (pc->state == 0) return pc.get();
        return nullptr;
    }
    bool chain_can_move() const               // (mu held)
    {
        return !chaining && !chain_done && pieces.size() < max_pieces &&
               (next_chain >= chunks.size() || (chunks[next_chain] && chunks[next_chain]->state == 2));
    }
    void work()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            if (quit) return;
            if (Piece* pc = resolvable()) {
                pc->state = 1;
                lk.unlock();
                resolve(*pc);
                lk.lock();
                pc->state = 2;
                cv_read.notify_all(); cv_work.notify_all();
                continue;
            }
            if (chain_can_move()) {
                chaining = true;
                lk.unlock();
                chain_step();
                lk.lock();
                chaining = false;
                cv_read.notify_all(); cv_work.notify_all();
                continue;
            }
            if (!chain_done && next_take < chunks.size() && next_take < next_chain + ahead) {
                Chunk* c = chunks[next_take++].get();
                c->state = 1;
                lk.unlock();
                run(*c);
                lk.lock();
                c->state = 2;
                cv_work.notify_all();
                continue;
            }
            cv_work.wait(lk);
        }
    }

    void resolve(Piece& pc)
    {
        const Out& o = pc.chunk->out;
        const size_t n = o.n - WIN;
        pc.bytes.resize(n);
        const uint16_t* s = o.d + WIN;
        const uint8_t* w = pc.win.get();
        uint8_t* d = pc.bytes.data();
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {                     // eight symbols at a time while they are plain bytes (nearly always)
            uint64_t a, b;
            memcpy(&a, s + i, 8); memcpy(&b, s + i + 4, 8);
            if (((a | b) & 0xFF00FF00FF00FF00ull) == 0) {
                a = (a | (a >> 8)) & 0x0000FFFF0000FFFFull; a = (a | (a >> 16)) & 0xFFFFFFFFull;
                b = (b | (b >> 8)) & 0x0000FFFF0000FFFFull; b = (b | (b >> 16)) & 0xFFFFFFFFull;
                const uint64_t out = a | b << 32;
                memcpy(d + i, &out, 8);
            } else {
                for (size_t k = i; k < i + 8; ++k) { const uint16_t v = s[k]; d[k] = v < MARK ? (uint8_t)v : w[v - MARK]; }
            }
        }
        for (; i < n; ++i) { const uint16_t v = s[i]; d[i] = v < MARK ? (uint8_t)v : w[v - MARK]; }
        pc.crc = crc_of(d, n);
        pc.chunk.reset();
        pc.win.reset();
    }

    // the first block start at or behind lo_bit (and before hi_bit) whose block decodes to text; then block after block
    void run(Chunk& c)
    {
        size_t at = c.lo_bit;
        if (!c.exact) {
            bool ok = false;
            while (at < c.hi_bit && !stop_all.load(std::memory_order_relaxed)) {
                const size_t cand = next_candidate(at, c.hi_bit);
                if (cand >= c.hi_bit) break;
                {
                    Bits h(in, n_in, cand + 3);                          // the whole header first: it is cheap to refuse
                    Huff hl, hd;
                    if (read_dynamic(h, hl, hd) != PG_OK) { at = cand + 1; continue; }
                }
                if (!c.out.start_unknown()) return;
                Bits b(in, n_in, cand);
                const int rc = inflate_block(b, c.out, WIN, true);
                if (rc == PG_OK) { c.start_bit = cand; c.end_bit = b.bit_pos(); ok = true; break; }
                at = cand + 1;
            }
            if (!ok) return;
        } else {
            if (!c.out.reserve(WIN + (size_t(1) << 20))) return;
            memset(c.out.d, 0, WIN * sizeof(uint16_t));
            c.out.n = WIN;
            c.start_bit = c.end_bit = c.lo_bit;
        }
        c.found = true;
        while (c.end_bit < c.hi_bit) {
            if (stop_all.load(std::memory_order_relaxed)) { c.stop = PG_DATA; return; }
            Bits b(in, n_in, c.end_bit);
            const size_t n0 = c.out.n;
            const int rc = inflate_block(b, c.out, c.exact ? std::min<size_t>(WIN, n0 - WIN) : WIN, false);
            if (rc < 0) { c.out.n = n0; c.stop = rc; return; }          // (the chain goes on from end_bit by itself and reports what is wrong)
            c.end_bit = b.bit_pos();
            if (rc == PG_FINAL) { c.final = true; return; }
            // a chunk's output is bounded: FASTQ inflates 4 : 1 (8 M symbols a chunk), but a crafted or very repetitive stream
            // can reach 1000 : 1, and 2 * threads + 2 chunks plus as many pieces are alive at once.  Beyond the bound the chunk
            // stops short; the chain takes what it has and inflates the rest of its range block by block in constant memory,
            // as it does behind any chunk that stopped early
            if (c.out.n - WIN > MAX_CHUNK_SYMBOLS) { c.stop = PG_LIMIT; return; }
        }
    }

    // the next bit >= at where a dynamic block may start: "not final, dynamic codes, counts in range" (17 bits) and code
    // lengths of the code-length code that form a complete code (one bit in nine passes the first test, one in several
    // hundred the second: the full header and the trial decode only run for those); limit if none before it
    size_t next_candidate(size_t at, size_t limit) const
    {
        for (size_t bit = at; bit < limit; ++bit) {
            const size_t byte = bit >> 3;
            if (byte + 12 > n_in) return limit;
            uint32_t w;
            memcpy(&w, in + byte, 4);
            const uint32_t v = w >> (bit & 7);
            if ((v & 7u) != 4u) continue;                                // BFINAL 0, BTYPE 2 (bits: 0, then 0 1)
            if (((v >> 3) & 31u) > 29u || ((v >> 8) & 31u) > 29u) continue;
            const uint32_t hclen = ((v >> 13) & 15u) + 4u;
            uint64_t x;
            memcpy(&x, in + ((bit + 17) >> 3), 8);
            x >>= (bit + 17) & 7;                                         // 57 bits: up to 19 lengths of 3 bits
            uint32_t space = 0;
            for (uint32_t k = 0; k < hclen; ++k, x >>= 3) { const uint32_t l = (uint32_t)(x & 7u); if (l) space += 128u >> l; }
            if (space != 128u) continue;
            return bit;
        }
        return limit;
    }

    // ---- the chain (one thread at a time: `chaining`) -----------------------------------------------------------------
    void push(std::unique_ptr<Piece> pc)
    {
        std::lock_guard<std::mutex> lk(mu);
        pieces.push_back(std::move(pc));
    }
    void set_error(int rc)
    {
        err = rc == PG_EOF ? "gzip: unexpected end of file" : rc == PG_NOMEM ? "out of memory" : "gzip: corrupt data";
        std::lock_guard<std::mutex> lk(mu);
        fail = true; chain_done = true;
    }
    void window_take(const uint8_t* o, size_t n)
    {
        if (n >= WIN) memcpy(win, o + n - WIN, WIN);
        else { memmove(win, win + n, WIN - n); memcpy(win + WIN - n, o, n); }
    }
    // the member's last block has been taken: CRC-32 and length stand behind it (the reader compares the CRC)
    void finish_member(Piece& last)
    {
        const size_t byte = (cur_bit + 7) >> 3;
        if (byte + 8 > n_in) { set_error(PG_EOF); return; }
        const uint8_t* t = in + byte;
        want_crc = t[0] | t[1] << 8 | t[2] << 16 | (uint32_t)t[3] << 24;
        const uint32_t want_len = t[4] | t[5] << 8 | t[6] << 16 | (uint32_t)t[7] << 24;
        if (want_len != (uint32_t)total_out) {
            err = "gzip: incorrect length check";
            std::lock_guard<std::mutex> lk(mu);
            fail = true; chain_done = true;
            return;
        }
        used_bytes = byte + 8;
        last.last = true;
    }
    // an accepted chunk: its symbols wait for a worker; the window moves on by the chunk's last 32 KiB, resolved here
    void accept(std::unique_ptr<Chunk> c)
    {
        std::unique_ptr<Piece> pc(new Piece());
        const size_t n = c->out.n - WIN;
        pc->win.reset(new uint8_t[WIN]);
        memcpy(pc->win.get(), win, WIN);
        const size_t k = std::min<size_t>(n, WIN);
        uint8_t tailb[WIN];
        const uint16_t* s = c->out.d + WIN + (n - k);
        for (size_t i = 0; i < k; ++i) { const uint16_t v = s[i]; tailb[i] = v < MARK ? (uint8_t)v : pc->win[v - MARK]; }
        window_take(tailb, k);
        total_out += n;
        cur_bit = c->end_bit;
        const bool final = c->final;
        pc->chunk = std::move(c);
        ++st_accepted;
        if (final) finish_member(*pc);
        const bool ended = final || fail;
        push(std::move(pc));
        if (ended) { std::lock_guard<std::mutex> lk(mu); chain_done = true; }
    }
    // one block at cur_bit, inflated here with the window known
    bool sequential_block()
    {
        Out seq;
        if (!seq.start_known(win)) { err = "out of memory"; std::lock_guard<std::mutex> lk(mu); fail = true; chain_done = true; return false; }
        Bits b(in, n_in, cur_bit);
        const int rc = inflate_block(b, seq, std::min<size_t>(WIN, total_out), false);
        if (rc < 0) { set_error(rc); return false; }
        std::unique_ptr<Piece> pc(new Piece());
        const size_t n = seq.n - WIN;
        pc->bytes.resize(n);
        for (size_t i = 0; i < n; ++i) pc->bytes[i] = (uint8_t)seq.d[WIN + i];
        pc->crc = crc_of(pc->bytes.data(), n); pc->state = 2;
        window_take(pc->bytes.data(), n);
        total_out += n;
        cur_bit = b.bit_pos();
        ++st_seq_blocks; st_seq_bytes += n;
        if (rc == PG_FINAL) finish_member(*pc);
        const bool ended = rc == PG_FINAL || fail;
        push(std::move(pc));
        if (ended) { std::lock_guard<std::mutex> lk(mu); chain_done = true; }
        return !ended;
    }
    // look at the next chunk in order (it has been inflated, or there is none left)
    void chain_step()
    {
        if (next_chain >= chunks.size()) { sequential_block(); return; }     // (behind the last chunk's range: cannot happen, kept for safety)
        Chunk* c = chunks[next_chain].get();
        bool drop = false;
        if (c->found && c->start_bit == cur_bit) {
            if (c->stop != PG_OK && !c->final) {
                // it stopped short of its range: take what it has, then block by block from there (with the error, if there is one)
                std::unique_ptr<Chunk> part(new Chunk());
                part->out.steal(c->out); part->end_bit = c->end_bit; part->final = false;
                const size_t hi = c->hi_bit;
                accept(std::move(part));
                c->found = false; c->hi_bit = std::max(hi, cur_bit + 1);
                return;
            }
            std::unique_ptr<Chunk> own;
            {
                std::lock_guard<std::mutex> lk(mu);
                own = std::move(chunks[next_chain]);
                ++next_chain;
            }
            accept(std::move(own));
            return;
        }
        if (c->found && c->start_bit > cur_bit) {
            if (!sequential_block()) return;
            if (cur_bit > c->start_bit) c->found = false;                    // ran past it: that was no block start
        } else {
            if (cur_bit >= c->hi_bit) drop = true;
            else { if (!sequential_block()) return; if (cur_bit >= c->hi_bit) drop = true; }
        }
        if (drop) {
            std::lock_guard<std::mutex> lk(mu);
            chunks[next_chain].reset();
            ++next_chain;
        }
    }
};

PGunzip::PGunzip(const uint8_t* in, size_t n_in, unsigned threads, size_t chunk_bytes) : p(new PGunzipImpl())
{
    p->in = in; p->n_in = n_in;
    p->nthreads = threads ? threads : 1;
    if (const char* e = getenv("BADGER_AMD_GUNZIP_CHUNK_KB")) chunk_bytes = (size_t)std::max(1, atoi(e)) << 10;
    p->chunk_bytes = chunk_bytes ? chunk_bytes : (size_t(2) << 20);
    p->ahead = 2 * (size_t)p->nthreads + 2;
    p->max_pieces = 2 * (size_t)p->nthreads + 2;
    memset(p->win, 0, sizeof(p->win));
    const size_t h = gzip_header(in, n_in);
    if (h == 0) { p->reader_fail = true; p->err = "gzip: incorrect header check"; return; }
    if (h == (size_t)-1) { p->reader_fail = true; p->err = "gzip: unexpected end of file"; return; }
    p->cur_bit = h * 8;
    const size_t total_bits = n_in * 8;
    for (size_t lo = h; lo < n_in; ) {
        std::unique_ptr<Chunk> c(new Chunk());
        const size_t hi = std::min(n_in, lo == h ? (h / p->chunk_bytes + 1) * p->chunk_bytes : lo + p->chunk_bytes);
        c->lo_bit = lo * 8; c->hi_bit = std::min(total_bits, hi * 8);
        c->exact = lo == h;
        p->chunks.push_back(std::move(c));
        lo = hi;
    }
    if (p->chunks.empty()) { p->reader_fail = true; p->err = "gzip: unexpected end of file"; return; }
    for (unsigned t = 0; t < p->nthreads; ++t) p->workers.emplace_back([this] { p->work(); });
}

PGunzip::~PGunzip()
{
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->quit = true;
    }
    p->stop_all.store(true);
    p->cv_work.notify_all();
    for (auto& t : p->workers) t.join();
    if (getenv("BADGER_AMD_GUNZIP_DEBUG"))
        fprintf(stderr, "pgunzip: %zu chunks, %zu accepted, %zu blocks (%zu bytes) inflated by the chain itself\n",
                p->chunks.size(), p->st_accepted, p->st_seq_blocks, p->st_seq_bytes);
    delete p;
}

size_t PGunzip::read(uint8_t* dst, size_t cap)
{
    size_t got = 0;
    while (got < cap && !p->member_end && !p->reader_fail) {
        Piece* pc = nullptr;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_read.wait(lk, [&] { return (!p->pieces.empty() && p->pieces.front()->state == 2) || (p->pieces.empty() && p->chain_done); });
            if (p->pieces.empty()) { p->reader_fail = true; if (p->err.empty()) p->err = "gzip: corrupt data"; break; }   // (the chain stopped with an error)
            pc = p->pieces.front().get();
        }
        const size_t n = pc->bytes.size();
        const size_t k = std::min(cap - got, n - p->front_at);
        memcpy(dst + got, pc->bytes.data() + p->front_at, k);
        got += k; p->front_at += k;
        if (p->front_at == n) {
            p->crc = (uint32_t)crc32_combine(p->crc, pc->crc, (z_off_t)n);
            const bool last = pc->last;
            {
                std::lock_guard<std::mutex> lk(p->mu);
                p->pieces.pop_front();
            }
            p->front_at = 0;
            p->cv_work.notify_all();
            if (last) {
                if (p->crc != p->want_crc) { p->reader_fail = true; p->err = "gzip: incorrect data check"; }
                else p->member_end = true;
            }
        }
    }
    return got;
}

bool PGunzip::failed() const { return p->reader_fail; }
const std::string& PGunzip::error() const { return p->err; }
bool PGunzip::at_member_end() const { return p->member_end; }
size_t PGunzip::consumed() const { return p->used_bytes; }

// test hook (not part of the C ABI of include/badger_hip.h): the whole file through PGunzip, member after member, into a
// malloc'ed buffer.  0 ok, -1 error (message in err).
extern "C" int bdg_test_gunzip(const uint8_t* in, size_t n_in, unsigned threads, size_t chunk_bytes, uint8_t** out, size_t* n_out, char* err, size_t err_cap)
{
    std::vector<uint8_t> all;
    size_t at = 0;
    int rc = 0;
    while (at < n_in && n_in - at >= 2 && in[at] == 0x1f && in[at + 1] == 0x8b) {
        PGunzip g(in + at, n_in - at, threads, chunk_bytes);
        std::vector<uint8_t> buf(size_t(1) << 20);
        for (;;) {
            const size_t k = g.read(buf.data(), buf.size());
            all.insert(all.end(), buf.begin(), buf.begin() + (ptrdiff_t)k);
            if (k < buf.size()) break;
        }
        if (g.failed()) { if (err && err_cap) { strncpy(err, g.error().c_str(), err_cap - 1); err[err_cap - 1] = 0; } rc = -1; break; }
        at += g.consumed();
    }
    *n_out = all.size();
    *out = static_cast<uint8_t*>(malloc(all.size() ? all.size() : 1));
    if (*out && !all.empty()) memcpy(*out, all.data(), all.size());
    return rc;
}
