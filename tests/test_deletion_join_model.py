"""The arithmetic of the deletion-variant joins (csrc/graph_kernels.hip: d2_key, d2_subseq, the run rule of the index
passes, d2_reports / d1_reports) restated in Python and checked on the CPU: every pair with dmin <= thr shares a group,
and the reporting rule - three closed-form relations that between them cover every way to dmin <= 2 - names exactly one
of the groups a pair shares, so the model's edge list is the oracle's, no edge missing and none twice; a second test throws
100,000 random pairs of every edit combination (random and low-complexity barcodes) at the same claim.  The kernels themselves are compared with the oracle on the GPU (tests/test_hip_parity.py); this
file pins the reasoning they rest on."""
import collections

import numpy as np

from badger_amd import synth
from oracle import pyoracle as orc

PQ = [(p, q) for p in range(16) for q in range(p + 1, 16)]
M32 = 0xFFFFFFFF


def ctz(x):
    return (x & -x).bit_length() - 1


def d1_key(r, p):
    return ((r & ((1 << (2 * p)) - 1)) | ((r >> (2 * p + 2)) << (2 * p))) & M32


def d2_key(r, p, q):
    lo = r & ((1 << (2 * p)) - 1)
    mid = (r >> (2 * p + 2)) & ((1 << (2 * (q - p - 1))) - 1)
    return (lo | (mid << (2 * p)) | ((r >> (2 * q + 2)) << (2 * q - 2))) & M32


def first_of_runs(r):
    d = (r ^ (r << 2)) & M32
    return ((d | (d >> 1)) & 0x55555554) | 1


def entries2(r):
    """distinct two-deletion 14-mers of r: first letters of runs only, then exact"""
    first, seen = first_of_runs(r), set()
    for p, q in PQ:
        if (first >> (2 * p)) & 1 and ((first >> (2 * q)) & 1 or p + 1 == q):
            seen.add(d2_key(r, p, q))
    return seen


def entries1(r):
    first = first_of_runs(r)
    return [d1_key(r, p) for p in range(16) if (first >> (2 * p)) & 1]


def prefix_suffix(a, b):
    x = a ^ b
    nz = (x | (x >> 1)) & 0x55555555
    return bin(nz).count("1"), ctz(nz) >> 1, (32 - nz.bit_length()) >> 1


def one_indel(a, b):
    """a without letter i == b without letter j: (i,) of a, by the narrowest interval (d2_reports rule 2)"""
    _, lcp, lcs = prefix_suffix(a, b)
    far = 15 - lcs
    near = min(lcp, far)
    span = ((1 << (2 * far)) - 1) & ~((1 << (2 * near)) - 1)
    if (((a >> 2) ^ b) & span) == 0:
        return near
    if (((b >> 2) ^ a) & span) == 0:
        return far
    return None


def shifted(x, y):
    """x[:15] against y without a letter, one substitution (d2_shifted): (deleted place of y, substituted place of x[:15], late)"""
    u = x & ((1 << 30) - 1)
    x0, x1 = (u ^ y) & ((1 << 30) - 1), (u ^ (y >> 2)) & ((1 << 30) - 1)
    nz0, nz1 = (x0 | (x0 >> 1)) & 0x15555555, (x1 | (x1 >> 1)) & 0x15555555
    f0 = (ctz(nz0) >> 1) if nz0 else 15
    behind = nz1 & ~((1 << (2 * f0)) - 1)
    if bin(behind).count("1") == 1:
        return f0, ctz(behind) >> 1, True
    if nz0:
        rest = nz0 & (nz0 - 1)
        j = (ctz(rest) >> 1) if rest else 15
        if (nz1 & ~((1 << (2 * j)) - 1)) == 0:
            return j, f0, False
    return None


def reports2(a, b, k, used):
    h, lcp, lcs = prefix_suffix(a, b)
    if h <= 2:
        used["letters"] += 1
        s1, s2 = lcp, (15 - lcs) if h == 2 else (1 if lcp == 0 else 0)
        return d2_key(a, min(s1, s2), max(s1, s2)) == k
    i = one_indel(a, b)
    if i is not None:
        used["indel"] += 1
        return (d1_key(a, i) >> 2) == k
    r = shifted(a, b)
    if r:
        used["shift"] += 1
        return d2_key(a, r[1], 15) == k
    r = shifted(b, a)
    if r:
        used["shift"] += 1
        dl, sb, late = r
        return (d2_key(a, dl, sb + 1) if late else d2_key(a, sb, dl)) == k
    used["none"] += 1                                        # none of the relations: dmin(a, b) > 2, no edge (the kernel does not verify it)
    return False


def reports1(a, b, k):
    i = one_indel(a, b)
    return i is not None and d1_key(a, i) == k


def test_keys_subsequences_and_the_run_rule():
    rng = np.random.default_rng(5)
    for _ in range(1500):
        r = int(rng.integers(0, 1 << 32))
        s = synth.rank_to_str(r)
        p, q = PQ[int(rng.integers(0, 120))]
        assert synth.rank_to_str(d2_key(r, p, q))[:14] == s[:p] + s[p + 1:q] + s[q + 1:] and d2_key(r, p, q) >> 28 == 0
        assert synth.rank_to_str(d1_key(r, p))[:15] == s[:p] + s[p + 1:]
        every2 = {s[:a] + s[a + 1:c] + s[c + 1:] for a, c in PQ}
        assert {synth.rank_to_str(k)[:14] for k in entries2(r)} == every2
        assert sorted(synth.rank_to_str(k)[:15] for k in entries1(r)) == sorted({s[:a] + s[a + 1:] for a in range(16)})


def _sets():
    from test_hip_parity import _low_complexity_barcodes, _observed_barcodes
    return {"observed": np.sort(_observed_barcodes(40, 900, 31)), "low complexity": np.sort(_low_complexity_barcodes(700, 41))}


def test_every_edge_from_exactly_one_group():
    for name, ranks in _sets().items():
        for thr, entries, reports in ((2, entries2, None), (1, entries1, reports1)):
            T = orc.qgram_threshold(thr)
            want = [(int(e["a"]), int(e["b"]), int(e["dist"])) for e in orc.graph_edges(ranks, thr, T, threads=4)]
            groups = collections.defaultdict(list)
            for i, r in enumerate(ranks):
                for k in entries(int(r)):
                    groups[k].append(i)
            used = collections.Counter()
            got = []
            for k, g in groups.items():
                assert len(set(g)) == len(g)                          # a row meets a group once
                for x in range(len(g)):
                    for y in range(x + 1, len(g)):
                        a, b = int(ranks[min(g[x], g[y])]), int(ranks[max(g[x], g[y])])
                        ok = reports2(a, b, k, used) if thr == 2 else reports(a, b, k)
                        if ok:
                            d = orc.dmin3(a, b)
                            if d <= thr and orc.qgram_S(a, b) >= T:
                                got.append((a, b, d))
            assert sorted(got) == want and len(want) > 300, (name, thr)
            if thr == 2:
                assert used["letters"] and used["indel"] and used["shift"], (name, dict(used))


def test_the_three_relations_cover_every_pair_within_distance_two():
    """random and low-complexity 16-mers with one or two edits of every kind (the barcode is cut back to 16 letters as the
    extraction does): whenever dmin <= 2 the pair shares a 14-mer, exactly one shared 14-mer reports it, and it is always
    one of the three relations that names it - what lets the kernel skip the verification of every other meeting"""
    rng = np.random.default_rng(7)
    combos = ["s", "d", "i", "ss", "sd", "si", "ds", "is", "di", "id", "dd", "ii"]

    def mutate(s, kinds):
        s = list(s)
        for kind in kinds:
            i = int(rng.integers(0, len(s)))
            if kind == "s":
                s[i] = "ACGT"[int(rng.integers(0, 4))]
            elif kind == "d":
                del s[i]
            else:
                s.insert(i, "ACGT"[int(rng.integers(0, 4))])
        return ("".join(s) + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 4)))[:16]
    used = collections.Counter()
    edges = 0
    for _ in range(100000):
        if rng.random() < 0.4:
            alphabet = "AC" if rng.random() < 0.5 else "ACG"
            unit = "".join(alphabet[int(x) % len(alphabet)] for x in rng.integers(0, 4, int(rng.integers(1, 6))))
            s = (unit * 16)[:16]
            if rng.random() < 0.5:
                s = mutate(s, "s")
        else:
            s = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 16))
        t = mutate(s, combos[int(rng.integers(0, len(combos)))])
        a, b = synth.str_to_rank(s), synth.str_to_rank(t)
        if a == b:
            continue
        a, b = min(a, b), max(a, b)
        if orc.dmin3(a, b) > 2:
            continue
        edges += 1
        shared = entries2(a) & entries2(b)
        assert shared, (s, t)
        assert sum(1 for k in shared if reports2(a, b, k, used)) == 1, (s, t)
        if orc.dmin3(a, b) <= 1:
            shared1 = set(entries1(a)) & set(entries1(b))
            assert shared1 and sum(1 for k in shared1 if reports1(a, b, k)) == 1, (s, t)
    assert edges > 60000 and used["none"] == 0 and min(used["letters"], used["indel"], used["shift"]) > 5000


# ---- the grouping that replaced the radix sort (csrc/dj_codec.hpp, csrc/bdg_partition.hpp) --------------------------------
MUL_A, MUL_B = 0x9E3779B1, 0x85EBCA6B


def mix(k, kb):
    m = (1 << kb) - 1
    x = (k * MUL_A) & m
    x ^= x >> 15
    return (x * MUL_B) & m


def test_entry_codec_of_the_shipped_library():
    """dj_codec.hpp, the very functions the kernels compile, run on the host by the library (no GPU): mixing is one-to-one
    and inverted by unmix, re-inserting the deleted letters gives the row back, encode -> decode returns (variant, row) for
    every deletion (pair) and every coarse bucket width."""
    from badger_amd import _native
    lib = _native.load()
    assert lib.bdg_selftest_dj_codec(1, 2000) == 0
    assert lib.bdg_selftest_dj_codec(20250711, 500) == 0


def test_mixed_keys_group_whole_variants_and_spread_them_evenly():
    """The buckets of the two partition levels and the bins inside a fine bucket are bit fields of the MIXED key, so a group
    (all entries of one variant) is never cut, and dense data - barcodes of a few thousand cells with errors, whose raw
    variants crowd a few thousand places - still fills the 256 coarse buckets within a few percent of each other."""
    rng = np.random.default_rng(3)
    wl = synth.make_whitelist(4000)
    cells = wl[rng.permutation(len(wl))[:300]].astype(np.uint64)
    r = cells[rng.integers(0, len(cells), 12000)]
    hit = rng.random(len(r)) < 0.5
    r = np.where(hit, r ^ (rng.integers(1, 4, len(r)).astype(np.uint64) << (2 * rng.integers(0, 16, len(r)).astype(np.uint64))), r)
    rows = np.unique(r.astype(np.uint32))
    keys = np.array([k for x in rows for k in entries2(int(x))], dtype=np.uint64)
    z = np.array([mix(int(k), 28) for k in keys], dtype=np.uint64)
    # one-to-one: as many distinct mixed keys as variants
    assert len(np.unique(z)) == len(np.unique(keys))
    coarse = np.bincount((z >> np.uint64(20)).astype(np.int64), minlength=256)
    raw = np.bincount((keys >> np.uint64(20)).astype(np.int64), minlength=256)
    assert coarse.max() < 1.35 * coarse.mean() and coarse.min() > 0.7 * coarse.mean()
    assert raw.max() > 2 * raw.mean() and raw.min() < 0.5 * raw.mean()    # (what the mixing is for)
    # bucket and bin of an entry are functions of its variant alone
    fine = (z >> np.uint64(14))
    for k in np.unique(keys)[:500]:
        assert len(np.unique(fine[keys == k])) == 1


def test_counting_pass_places_are_exact():
    """bdg_partition.hpp in numpy: the tiles' counts as a buckets x tiles matrix -> a scan along every row gives each
    (bucket, tile) run its own place, k_part_bases turns the row totals into the buckets' places; scattering with per-tile
    cursors then fills [0, m) exactly once and every bucket's range holds exactly its entries."""
    rng = np.random.default_rng(5)
    ntiles, nb1 = 37, 256
    tiles = [rng.integers(0, 1 << 28, int(rng.integers(0, 900))).astype(np.uint64) for _ in range(ntiles)]
    hist = np.stack([np.bincount((t >> np.uint64(20)).astype(np.int64), minlength=nb1) for t in tiles]).T      # [bucket][tile]
    within = np.cumsum(hist, axis=1) - hist                               # k_part_colscan: one row per bucket
    tot = hist.sum(axis=1)
    base = np.concatenate([[0], np.cumsum(tot)])                          # k_part_bases
    m = int(base[-1])
    out = np.full(m, -1, dtype=np.int64)
    for t, ent in enumerate(tiles):
        cur = base[:-1] + within[:, t]                                    # the tile's cursors
        for e in ent:
            b = int(e >> np.uint64(20))
            assert out[cur[b]] == -1
            out[cur[b]] = int(e)
            cur[b] += 1
    assert (out >= 0).all()
    for b in range(nb1):
        seg = out[base[b]:base[b + 1]]
        assert ((seg >> 20) == b).all()
    assert sorted(out.tolist()) == sorted(int(e) for t in tiles for e in t)


def test_spread_form_of_the_verifying_myers_pass():
    """dmin3 (graph_kernels.hip) keeps its bit vectors spread over the even bits and lets the addition carry through odd bits
    that pv keeps set; tools/myers_spread_check.py restates exactly those statements on the host and compares 20,000 pairs with
    the edit-distance recurrence."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "myers_spread_check.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
