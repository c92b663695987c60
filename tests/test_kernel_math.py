"""The arithmetic identities a few kernels rest on, restated in numpy and checked exhaustively or on random inputs (CPU tier).
The GPU parity tests would catch a wrong kernel; these say WHY the kernels are right."""
import numpy as np


def _variant(q, i):
    """deletion variant i of a rank-packed 16-mer (nearest_kernels.hip: variant())"""
    lm = (1 << (2 * i)) - 1 if i < 16 else 0xFFFFFFFF
    return ((q & lm) | ((q >> 2) & ~lm & 0xFFFFFFFF)) & 0x3FFFFFFF


def _delmap_index(d, g):
    """nearest_kernels.hip: delmap_index()"""
    lowm = (1 << (8 * g)) - 1
    return ((d >> (8 * g)) & 63) | ((d & lowm) << 6) | (d & ~((lowm << 6) | 63) & 0xFFFFFFFF)


def test_four_deletion_variants_share_a_map_word():
    """k_nearest_delins: lane g of a query reads ONE 64-bit word of map copy g for its variants 4g .. 4g+3."""
    rng = np.random.default_rng(5)
    qs = [int(x) for x in rng.integers(0, 1 << 32, 5000)] + [0, 0xFFFFFFFF, 0x55555555, 0x12345678]
    for q in qs:
        for g in range(4):
            idx = [_delmap_index(_variant(q, 4 * g + t), g) for t in range(4)]
            assert len({i >> 6 for i in idx}) == 1
    # the permutation is a bijection on 30 bits (no two variants collide in a copy)
    for g in range(4):
        d = rng.integers(0, 1 << 30, 200000).astype(np.uint64)
        lowm = np.uint64((1 << (8 * g)) - 1)
        idx = ((d >> np.uint64(8 * g)) & np.uint64(63)) | ((d & lowm) << np.uint64(6)) | (d & ~((lowm << np.uint64(6)) | np.uint64(63)))
        assert len(np.unique(idx)) == len(np.unique(d)) and int(idx.max()) < (1 << 30)


def _window_counts_bitsliced(m):
    """k_scan_reads eval_cands: bit k of the result = (at least 12 of m[k .. k+16) are set), for k = 0 .. 15"""
    M = 0xFFFFFFFF
    xor3 = lambda a, b, c: a ^ b ^ c
    maj = lambda a, b, c: (a & b) | (a & c) | (b & c)
    a0, a1 = m ^ (m >> 1), m & (m >> 1)
    b0, b1 = a0 >> 2, a1 >> 2
    c0 = a0 & b0
    s0, s1, s2 = a0 ^ b0, xor3(a1, b1, c0), maj(a1, b1, c0)
    d0, d1, d2 = s0 >> 4, s1 >> 4, s2 >> 4
    e0 = s0 & d0
    e1 = maj(s1, d1, e0)
    t0, t1, t2, t3 = s0 ^ d0, xor3(s1, d1, e0), xor3(s2, d2, e1), maj(s2, d2, e1)
    f0, f1, f2, f3 = t0 >> 8, t1 >> 8, t2 >> 8, t3 >> 8
    g0 = t0 & f0
    g1 = maj(t1, f1, g0)
    g2 = maj(t2, f2, g1)
    u2, u3, u4 = xor3(t2, f2, g1), xor3(t3, f3, g2), maj(t3, f3, g2)
    return (u4 | (u3 & u2)) & 0xFFFF & M


def test_bit_sliced_window_counts():
    rng = np.random.default_rng(9)
    words = [int(x) for x in rng.integers(0, 1 << 32, 20000)]
    words += [int(x) | int(y) for x, y in zip(rng.integers(0, 1 << 32, 20000), rng.integers(0, 1 << 32, 20000))]      # T-rich
    words += [0, 0xFFFFFFFF, 0x0000FFFF, 0xFFFF0000, 0x0FFF0FFF, 0x7FFFFFFF]
    for m in words:
        want = 0
        for k in range(16):
            if bin((m >> k) & 0xFFFF).count("1") >= 12:
                want |= 1 << k
        assert _window_counts_bitsliced(m) == want, hex(m)


def test_last_a_window_is_the_first_window_backwards():
    """eval_cands serves the reverse strand by reversing the 31 flags: start k -> 15 - k, the strand position
    L - 16 - (p0 + k) -> (L - 31 - p0) + (15 - k), and the offset of the last 'AAA' inside the window counted from its end
    equals the offset of the first 'TTT' behind the reversed start."""
    rng = np.random.default_rng(11)
    for _ in range(20000):
        m = int(rng.integers(0, 1 << 31)) | int(rng.integers(0, 1 << 31))
        q = [k for k in range(16) if bin((m >> k) & 0xFFFF).count("1") >= 12]
        if not q:
            continue
        k = max(q)                                                   # last window (original code path)
        m3 = m & (m >> 1) & (m >> 2)
        mm = m3 & ((1 << (k + 14)) - 1)
        j = mm.bit_length() - 1 if mm else k + 13
        off = k + 13 - j
        r = int("{:032b}".format(m)[::-1], 2) >> 1                   # __brev(m) >> 1
        qr = [kk for kk in range(16) if bin((r >> kk) & 0xFFFF).count("1") >= 12]
        kr = min(qr)
        assert kr == 15 - k
        tt = (r & (r >> 1) & (r >> 2)) >> kr
        assert ((tt & -tt).bit_length() - 1 if tt else 0) == off
