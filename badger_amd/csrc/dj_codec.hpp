// Deletion-variant joins (graph_kernels.hip): what an index entry is.
//
// A row (a 16-mer r) meets other rows in the groups of its deletion variants: the 14-mers r without two letters (thr <= 2)
// or the 15-mers r without one (thr <= 1).  An entry must name the variant k and the row.  Instead of the pair (k, r) -
// 64 bits, what round 3 sorted - an entry is 32 bits: the variant's MIXED key z = mix(k), a one-to-one map of the 28 / 30
// key bits whose top bits are spread evenly whatever the barcodes look like, minus the top bits that the bucket the entry
// lies in spells anyway, plus what turns k back into r: which letters were deleted and what they were (11 / 6 bits).
// Half the bytes through HBM, and the bucket number doubles as a share of the key.
//
// Plain functions, compiled for the host as well: bdg_selftest_dj_codec() runs them on the CPU (tests, no GPU needed).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define BDG_HD __host__ __device__ __forceinline__
#else
#define BDG_HD inline
#endif

namespace djc {

constexpr uint32_t MUL_A = 0x9E3779B1u, MUL_B = 0x85EBCA6Bu;

constexpr uint32_t inv_odd(uint32_t a)                // a * inv_odd(a) == 1 mod 2^32 (Newton, five doublings of the valid bits)
{
    uint32_t x = a;
    for (int i = 0; i < 5; ++i) x *= 2u - a * x;
    return x;
}
constexpr uint32_t INV_A = inv_odd(MUL_A), INV_B = inv_odd(MUL_B);
static_assert(MUL_A * INV_A == 1u && MUL_B * INV_B == 1u, "inverse mod 2^32");

// one-to-one on KB bits (KB <= 30: x ^= x >> 15 is its own inverse there); every result bit depends on every key bit
template <int KB> BDG_HD uint32_t mix(uint32_t k)
{
    constexpr uint32_t M = (1u << KB) - 1u;
    uint32_t x = (k * MUL_A) & M;
    x ^= x >> 15;
    return (x * MUL_B) & M;
}
template <int KB> BDG_HD uint32_t unmix(uint32_t z)
{
    constexpr uint32_t M = (1u << KB) - 1u;
    uint32_t x = (z * INV_B) & M;
    x ^= x >> 15;
    return (x * INV_A) & M;
}

BDG_HD uint32_t low_mask(uint32_t bits) { return bits >= 32u ? 0xFFFFFFFFu : (1u << bits) - 1u; }

// r without its base p: a 15-mer in 30 bits
BDG_HD uint32_t del1(uint32_t r, uint32_t p)
{
    const uint32_t lo = r & low_mask(2u * p);
    const uint32_t hi = (uint32_t)((unsigned long long)r >> (2u * p + 2u));
    return lo | (hi << (2u * p));
}
// the 16-mer that becomes k when its base p (letter l) is deleted
BDG_HD uint32_t ins1(uint32_t k, uint32_t p, uint32_t l)
{
    const uint32_t lo = k & low_mask(2u * p);
    const unsigned long long hi = (unsigned long long)(k >> (2u * p)) << (2u * p + 2u);
    return lo | (l << (2u * p)) | (uint32_t)hi;
}
// r without its bases p and q (p < q): a 14-mer in 28 bits
BDG_HD uint32_t del2(uint32_t r, uint32_t p, uint32_t q)
{
    const uint32_t lo = r & low_mask(2u * p);
    const uint32_t mid = (r >> (2u * p + 2u)) & low_mask(2u * (q - p - 1u));
    const uint32_t hi = (uint32_t)((unsigned long long)r >> (2u * q + 2u));          // (q = 15: nothing)
    return lo | (mid << (2u * p)) | (hi << (2u * q - 2u));
}
// the 16-mer that becomes k when its bases p (letter lp) and q (letter lq), p < q, are deleted
BDG_HD uint32_t ins2(uint32_t k, uint32_t p, uint32_t q, uint32_t lp, uint32_t lq)
{
    const uint32_t lo = k & low_mask(2u * p);
    const uint32_t mid = (k >> (2u * p)) & low_mask(2u * (q - p - 1u));
    const unsigned long long hi = (unsigned long long)(k >> (2u * q - 2u)) << (2u * q + 2u);
    return lo | (lp << (2u * p)) | (mid << (2u * p + 2u)) | (lq << (2u * q)) | (uint32_t)hi;
}

// deletion pair t of 120 -> p << 4 | q, in the order p = 0 (q = 1..15), p = 1 (q = 2..15), ...
BDG_HD uint32_t pair_of(uint32_t t)
{
    uint32_t p = 0, left = t;
    while (left >= 15u - p) { left -= 15u - p; ++p; }
    return p << 4 | (p + 1u + left);
}

// ---- entries.  l1 = log2 of the coarse bucket count; zb = KB - l1 bits of z stay in the entry
// two deletions: z[zb] | pair t << zb | letter p << (zb + 7) | letter q << (zb + 9)        (zb + 11 <= 31 bits for l1 >= 8)
BDG_HD uint32_t enc2(uint32_t z, uint32_t zb, uint32_t t, uint32_t r, uint32_t pq)
{
    const uint32_t p = pq >> 4, q = pq & 15u;
    return (z & low_mask(zb)) | (t << zb) | (((r >> (2u * p)) & 3u) << (zb + 7u)) | (((r >> (2u * q)) & 3u) << (zb + 9u));
}
// one deletion: z[zb] | p << zb | letter << (zb + 4)                                          (zb + 6 <= 28 bits)
BDG_HD uint32_t enc1(uint32_t z, uint32_t zb, uint32_t p, uint32_t r)
{
    return (z & low_mask(zb)) | (p << zb) | (((r >> (2u * p)) & 3u) << (zb + 4u));
}
// back: the entry e of coarse bucket b1 -> variant k and row barcode r
BDG_HD void dec2(uint32_t e, uint32_t b1, uint32_t zb, uint32_t pq_of_t, uint32_t& k, uint32_t& r)
{
    k = unmix<28>((b1 << zb) | (e & low_mask(zb)));
    r = ins2(k, pq_of_t >> 4, pq_of_t & 15u, (e >> (zb + 7u)) & 3u, (e >> (zb + 9u)) & 3u);
}
BDG_HD void dec1(uint32_t e, uint32_t b1, uint32_t zb, uint32_t& k, uint32_t& r)
{
    k = unmix<30>((b1 << zb) | (e & low_mask(zb)));
    r = ins1(k, (e >> zb) & 15u, (e >> (zb + 4u)) & 3u);
}

}  // namespace djc
