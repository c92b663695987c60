// Parallel inflate of ONE plain gzip stream (what gzip.open() reads for the reference, extract_raw_barcodes.py:80-90,137-141).
// A deflate stream has no index: a block may refer to the 32 KiB before it and starts at any bit.  The stream is cut into
// chunks of compressed bytes; a worker looks for the first block start inside its chunk (a dynamic-Huffman header that parses,
// whose block decodes to text) and inflates from there WITHOUT knowing what came before: what it produces is 16-bit symbols,
// a byte or "the byte k places into the unknown 32 KiB".  The consumer walks the chunks in order.  It knows the exact bit
// where the data it has accepted ends; a chunk is taken only if it began at exactly that bit (then it began at a true
// block boundary, by induction from the start of the stream), its symbols become bytes against the now known window, and
// whatever lies between is inflated in sequence by the same decoder.  A guess that was wrong costs time, never bytes; the
// member's CRC-32 and length are checked at its end like zlib checks them.
// Internal to the library (ingest.cpp); no HIP.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

struct PGunzipImpl;

class PGunzip {
public:
    // `in`: the file from the first byte of a gzip member on (the mapping stays valid while this object lives);
    // threads: inflating workers (>= 1); chunk_bytes: compressed bytes per chunk (0: default).
    PGunzip(const uint8_t* in, size_t n_in, unsigned threads, size_t chunk_bytes);
    ~PGunzip();
    // Up to `cap` inflated bytes of the member in stream order; 0 at the member's end (then consumed() says where it ended)
    // or on an error (failed()).  One caller at a time.
    size_t read(uint8_t* dst, size_t cap);
    bool failed() const;
    const std::string& error() const;
    bool at_member_end() const;
    size_t consumed() const;                 // bytes of `in` the member took (header and trailer included), valid at its end
private:
    PGunzipImpl* p;
};
