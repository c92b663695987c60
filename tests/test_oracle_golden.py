"""The C oracle (oracle/badger_oracle.c) against fixtures produced by the
reference's own Python modules (tools/gen_golden.py -> tests/golden/).  CPU only."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as orc
from badger_amd import synth


@pytest.fixture(scope="module")
def prim(golden_dir):
    return json.load(open(os.path.join(golden_dir, "primitives.json")))


def test_find_polyt_start(prim):
    for s, want in prim["find_polyt_start"]:
        assert orc.find_polyt_start(s) == want, s
    for s, want in prim["find_polyt_start_w5"]:
        assert orc.find_polyt_start(s, 5, 1.0) == want, s


def test_reverse_complement(prim):
    for s, want in prim["reverese_complement"]:
        assert orc.revcomp(s) == want
    with pytest.raises(KeyError):
        orc.revcomp("ACGTacgt")


def test_kmer_hits(prim):
    for s, want in prim["get_occurrences"]:
        assert orc.kmer_hits(s) == want, s


def test_sw_align(prim):
    # fixture: align_pattern_ssw(window, 0, len, R1, 0) -> (ref_start, ref_end, read_start, read_end, score)
    for w, want in prim["align_pattern_ssw"]:
        assert list(orc.sw_align(orc.R1, w)) == want, w


def test_detect_exact_positions(prim):
    for s, start, end, hits, ms, sd, ed, want in prim["detect_exact_positions"]:
        got = orc.detect_exact_positions(s, start, end, hits, ms, sd, ed)
        assert list(got) == want, (s, ms, sd, ed)


def test_rank_unrank(prim):
    for s, rk in prim["rank"]:
        assert orc.rank16(s) == rk
        assert orc.unrank16(rk) == s
        assert synth.str_to_rank(s) == rk and synth.rank_to_str(rk) == s
    assert orc.rank16("CAAAAAAAAAAAAAAA") == 1 and orc.rank16("AAAAAAAAAAAAAAAC") == 1 << 30


def test_qgram_threshold(prim):
    for t, want in prim["qgram_threshold"]:
        assert orc.qgram_threshold(t) == want
    assert orc.qgram_threshold(1) == 5 and orc.qgram_threshold(2) == 4


def test_levenshtein(prim):
    for a, b, want in prim["editdistance"]:
        assert orc.levenshtein(a, b) == want
        if len(a) <= 16 and len(b) <= 16 and len(a) > 0:
            pa = sum("ACGT".index(c) << (2 * i) for i, c in enumerate(a))
            pb = sum("ACGT".index(c) << (2 * i) for i, c in enumerate(b))
            assert orc.lev16_packed(pa, len(a), pb, len(b)) == want


def test_myers_vs_dp_random():
    rng = np.random.default_rng(3)
    for _ in range(3000):
        a, b = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))
        if rng.random() < 0.5:       # near pair
            b = a ^ (int(rng.integers(0, 4)) << (2 * int(rng.integers(0, 16))))
            if rng.random() < 0.5:
                k = int(rng.integers(0, 15))
                lowmask = (1 << (2 * k)) - 1
                b = (b & lowmask) | ((b >> 2) & ~lowmask & 0xFFFFFFFF) | (int(rng.integers(0, 4)) << 30)
        sa, sb = synth.rank_to_str(a), synth.rank_to_str(b)
        for la, lb in ((16, 16), (15, 16), (16, 15)):
            assert orc.lev16_packed(a, la, b, lb) == orc.levenshtein(sa[:la], sb[:lb])
        assert orc.dmin3(a, b) == min(orc.levenshtein(sa, sb), orc.levenshtein(sa[:-1], sb), orc.levenshtein(sa, sb[:-1]))


def _row(rid, seq, rec):
    """Format one TSV row the way barcode_callers.py:40-42,91-93 does, from a record."""
    s = orc.revcomp(seq) if rec["flags"] & orc.FLAG_REV else seq
    if rec["valid"]:
        bc = s[rec["bc_start"]:rec["bc_start"] + 16]
        umi = s[rec["umi_start"]:rec["umi_end"]]
        score = 0
    else:
        bc, umi, score = "*", "*", -1
    strand = {1: "+", -1: "-", 0: "."}[int(rec["strand"])]
    return "%s\t%s\t%s\t%d\t%s\t%s\t%d\t%d" % (rid, bc, umi, score, False, strand, rec["polyT"], rec["r1_end"])


def test_extract_rows(golden_dir):
    ext = json.load(open(os.path.join(golden_dir, "extract_rows.json")))
    assert ext["header"] == "#read_id\tbarcode\tUMI\tBC_score\tvalid_UMI\tstrand\tpolyT_start\tR1_end"
    for umi_len, key, skey in ((12, "row_v3", "r1_score_v3"), (10, "row_v2", "r1_score_v2")):
        for r in ext["reads"]:
            rec = orc.extract_read(r["seq"], umi_len)
            assert _row(r["id"], r["seq"], rec) == r[key], r["id"]
            assert int(rec["r1_score"]) == r[skey], r["id"]


def test_extract_rows_no_polya_rule(golden_dir):
    """find_barcode_umi_no_polya (barcode_callers.py:231-248) on the same reads: rows produced by the reference."""
    ext = json.load(open(os.path.join(golden_dir, "extract_rows.json")))
    alt = json.load(open(os.path.join(golden_dir, "extract_rows_no_polya.json")))
    assert len(alt["reads"]) == len(ext["reads"])
    differ = 0
    for umi_len, key, skey in ((12, "row_v3", "r1_score_v3"), (10, "row_v2", "r1_score_v2")):
        for r, a in zip(ext["reads"], alt["reads"]):
            assert a["id"] == r["id"] and a[key] != "KeyError"
            rec = orc.extract_read(r["seq"], umi_len, rule=orc.RULE_NO_POLYA)
            assert _row(r["id"], r["seq"], rec) == a[key], r["id"]
            assert int(rec["r1_score"]) == a[skey], r["id"]
            differ += a[key] != r[key]
    assert differ >= 10          # the fixture does exercise the difference between the two rules


def test_config1_tsv_and_stats(golden_dir):
    seqs, ids = [], []
    with gzip.open(os.path.join(golden_dir, "c1_reads.fa.gz"), "rt") as f:
        for line in f:
            if line.startswith(">"):
                ids.append(line[1:].strip())
            else:
                seqs.append(line.strip())
    bases, off = synth.list_to_reads(seqs)
    recs = orc.extract_batch(bases, off, 12, threads=4)
    want = open(os.path.join(golden_dir, "c1_expected.tsv")).read().split("\n")
    assert want[0].startswith("#read_id")
    rows = [_row(i, s, r) for i, s, r in zip(ids, seqs, recs)]
    assert rows == want[1:1 + len(rows)]
    stats = dict(l.split("\t") for l in open(os.path.join(golden_dir, "c1_expected.tsv.stats")).read().strip().split("\n"))
    assert int(stats["Total reads:"]) == len(recs)
    assert int(stats["Barcode detected:"]) == int(recs["valid"].sum())
    assert int(stats["PolyT detected:"]) == int((recs["polyT"] != -1).sum())
    assert int(stats["R1 detected:"]) == int((recs["r1_end"] != -1).sum())
    # the batch entry point is order-preserving and thread-count invariant
    assert (orc.extract_batch(bases, off, 12, threads=1) == recs).all()


def _ranks_of(barcodes):
    out = []
    for s in barcodes:
        if len(s) == 17:
            s = s[:-1]
        if len(s) == 16:
            out.append(synth.str_to_rank(s))
    return out


def test_graph_edges(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "graph.json")))
    for key in ("c1_thr1", "c1_thr2", "cells60_thr1", "cells60_thr2"):
        case = g[key]
        thr = int(key[-1])
        ranks = _ranks_of(case["barcodes"])
        # counts in first-occurrence order (barcode_graph.py:192-204)
        uniq, first, cnt = np.unique(np.array(ranks, dtype=np.uint32), return_index=True, return_counts=True)
        order = np.argsort(first)
        assert [[int(uniq[i]), int(cnt[i])] for i in order] == case["counts"]
        assert orc.qgram_threshold(thr) == case["qgram_T"]
        e = orc.graph_edges(uniq, thr, case["qgram_T"], threads=4)
        got = [[int(x["a"]), int(x["b"]), int(x["dist"])] for x in e]
        assert got == case["edges"], key
        eb = orc.graph_edges(uniq, thr, case["qgram_T"], brute=True)
        assert (eb == e).all()
        # every third row of the sorted array against the whole array (bench.py's bounded form) = those rows of the list
        es, t_index, t_rows = orc.graph_edges_sampled(uniq, thr, 3, case["qgram_T"], threads=2)
        assert (es == e[np.isin(e["a"], np.sort(uniq)[::3])]).all() and t_index >= 0.0 and t_rows >= 0.0
    # the thr=2 filter is lossy: brute force without the S filter finds more
    case = g["cells60_thr2"]
    uniq = np.unique(np.array(_ranks_of(case["barcodes"]), dtype=np.uint32))
    assert len(orc.graph_edges(uniq, 2, 1, brute=True)) > len(case["edges"])


def test_get_close_is_S_statistic(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "graph.json")))
    ranks = np.unique(np.array(_ranks_of(g["cells60_thr1"]["barcodes"]), dtype=np.uint32))
    for rk, close in g["cells60_get_close_thr1"]:
        mine = sorted(int(b) for b in ranks if b > rk and orc.qgram_S(rk, int(b)) >= 5)
        assert mine == close


def test_nearest16_small():
    wl = synth.make_whitelist(500)
    rng = np.random.default_rng(5)
    q = np.concatenate([wl[:50], wl[50:100] ^ np.uint32(1), rng.integers(0, 1 << 32, 50, dtype=np.uint64).astype(np.uint32)])
    idx, ed, ties = orc.nearest16(q, wl, max_ed=16, threads=2)
    for i in range(len(q)):
        d = [orc.levenshtein(synth.rank_to_str(q[i]), synth.rank_to_str(w)) for w in wl]
        m = min(d)
        assert ed[i] == m and idx[i] == d.index(m) and ties[i] == d.count(m)
    idx2, ed2, ties2 = orc.nearest16(q, wl, max_ed=2)
    far = ed > 2
    assert (ed2[far] == 255).all() and (idx2[far] == 0xFFFFFFFF).all() and (ties2[far] == 0).all()
    assert (ed2[~far] == ed[~far]).all() and (idx2[~far] == idx[~far]).all()


def test_nearest16_probe_form_equals_exhaustive_scan():
    """The oracle's neighbourhood-probe nearest16 (CPU baseline of bench.py, large-sample checker) gives exactly what
    the exhaustive Levenshtein scan (barcode_graph.py:376-384 as an operator) gives: index, distance, tie count."""
    rng = np.random.default_rng(5)
    wl = rng.permutation(np.unique(rng.integers(0, 1 << 32, 60000, dtype=np.uint64).astype(np.uint32)))     # caller order, not sorted
    nq = 400
    src = wl[rng.integers(0, len(wl), nq)].astype(np.uint64)
    kind = rng.integers(0, 6, nq)
    q = src.copy()
    for rounds, sel in ((1, kind == 1), (2, kind == 2), (3, kind == 3)):
        for _ in range(rounds):
            q = np.where(sel, q ^ (rng.integers(1, 4, nq).astype(np.uint64) << (2 * rng.integers(0, 16, nq).astype(np.uint64))), q)
    pos = rng.integers(0, 16, nq).astype(np.uint64)
    low = (np.uint64(1) << (np.uint64(2) * pos)) - np.uint64(1)
    dele = (q & low) | ((q >> np.uint64(2)) & ~low & np.uint64(0x3FFFFFFF)) | (rng.integers(0, 4, nq).astype(np.uint64) << np.uint64(30))
    q = np.where(kind == 4, dele, q)
    q = np.where(kind == 5, rng.integers(0, 1 << 32, nq, dtype=np.uint64), q).astype(np.uint32)
    for max_ed in (0, 1, 2, 3):
        a = orc.nearest16(q, wl, max_ed, threads=4)
        b = orc.nearest16(q, wl, max_ed, threads=4, probe=True)
        assert all((x == y).all() for x, y in zip(a, b)), max_ed
    # a crowded whitelist: many entries one substitution apart, ties everywhere
    cells = rng.integers(0, 1 << 32, 20, dtype=np.uint64)
    dense = np.unique(np.concatenate([c ^ (rng.integers(0, 4, 300).astype(np.uint64) << (2 * rng.integers(0, 16, 300).astype(np.uint64)))
                                      for c in cells]).astype(np.uint32))
    dense = rng.permutation(dense)
    qd = dense[rng.integers(0, len(dense), 300)] ^ (np.uint32(1) << (2 * rng.integers(0, 16, 300)).astype(np.uint32))
    a = orc.nearest16(qd, dense, 2, threads=4)
    b = orc.nearest16(qd, dense, 2, threads=4, probe=True)
    assert all((x == y).all() for x, y in zip(a, b))
    # 42 entries one substitution (positions 0..13) off a centre that is not itself an entry; the centre with its last base
    # changed is two substitutions from each of them
    c = np.uint32(0x1B2C3D4E)
    ring = np.array([c ^ np.uint32(x << (2 * p)) for p in range(14) for x in (1, 2, 3)], dtype=np.uint32)
    for qq in (c ^ np.uint32(1 << 30), c):
        a = orc.nearest16(np.array([qq], np.uint32), ring, 2, threads=1)
        b = orc.nearest16(np.array([qq], np.uint32), ring, 2, threads=1, probe=True)
        assert all((x == y).all() for x, y in zip(a, b)) and a[2][0] == 42 and a[0][0] == 0
