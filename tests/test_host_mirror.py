"""Host-side mirror of the reference interface (CLI, readers, row/stat formatting, clustering).
CPU only: device records are supplied by the oracle, graph edges by the golden fixture."""
import gzip
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from badger_amd import badger, common, extract_raw_barcodes as erb, synth
from badger_amd.barcode_extraction import barcode_callers as bcall
from badger_amd.barcode_graph import BarcodeGraph, qgram_threshold
from oracle import pyoracle as orc


def _c1(golden_dir):
    ids, seqs = [], []
    with gzip.open(os.path.join(golden_dir, "c1_reads.fa.gz"), "rt") as f:
        for line in f:
            (ids if line.startswith(">") else seqs).append(line[1:].strip() if line.startswith(">") else line.strip())
    return ids, seqs


def test_readers_and_chunks(tmp_path, golden_dir):
    ids, seqs = _c1(golden_dir)
    got = list(erb.open_reads(os.path.join(golden_dir, "c1_reads.fa.gz")))
    assert got == list(zip(ids, seqs))
    fq = tmp_path / "x.fastq"
    fq.write_text("".join("@%s some comment\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in zip(ids[:50], seqs[:50])))
    assert list(erb.open_reads(str(fq))) == list(zip(ids[:50], seqs[:50]))
    fa = tmp_path / "y.fa"
    fa.write_text(">a desc\nACGT\nAC\n\n>b\nTTTT\n")
    assert list(erb.open_reads(str(fa))) == [("a", "ACGTAC"), ("b", "TTTT")]
    assert erb.open_reads(str(tmp_path / "z.txt")) is None
    chunks = list(erb.read_chunks(iter(range(250)), 100))
    assert [len(c) for c in chunks] == [100, 100, 50]
    assert [len(c) for c in erb.read_chunks(iter(range(200)), 100)] == [100, 100, 0]      # trailing empty chunk, as the reference


def test_rows_results_and_stats_from_records(golden_dir):
    ext = json.load(open(os.path.join(golden_dir, "extract_rows.json")))
    assert bcall.TenXBarcodeDetectionResult.header() == ext["header"]
    for r in ext["reads"]:
        rec = orc.extract_read(r["seq"], 12)
        assert bcall.record_to_row(r["id"], r["seq"], rec) == r["row_v3"]
        res = bcall.record_to_result(r["id"], r["seq"], rec)
        assert str(res) == r["row_v3"] and res.r1_score == r["r1_score_v3"]
        assert res.is_valid() == (r["row_v3"].split("\t")[1] != "*")
    ids, seqs = _c1(golden_dir)
    bases, off = synth.list_to_reads(seqs)
    recs = orc.extract_batch(bases, off, 12, threads=4)
    st = bcall.ReadStats()
    st.add_records(recs[:400])
    st.add_records(recs[400:])
    assert str(st) == open(os.path.join(golden_dir, "c1_expected.tsv.stats")).read()
    st2 = bcall.ReadStats()
    for i, s, r in zip(ids, seqs, recs):
        st2.add_read(bcall.record_to_result(i, s, r))
    assert str(st2) == str(st)
    with pytest.raises(KeyError):
        bcall.reverese_complement("ACGx")
    assert bcall.reverese_complement("AACGTN") == "NACGTT"


def test_rank_helpers():
    rng = np.random.default_rng(0)
    seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, 16)) for _ in range(100)]
    many = common.rank_many(seqs)
    for s, rk in zip(seqs, many):
        assert common.rank(s, 16) == int(rk) == orc.rank16(s) and common.unrank(int(rk), 16) == s
    with pytest.raises(KeyError):
        common.rank_many(["ACGTACGTACGTACGN"])
    assert [qgram_threshold(t, 16) for t in (0, 1, 2, 3)] == [11, 5, 4, 4]


def test_stage2_host_logic_matches_reference_output(tmp_path, golden_dir):
    """import_tsv + counting + clustering + output writer on the reference's own edge list."""
    g = json.load(open(os.path.join(golden_dir, "graph.json")))["c1_thr1"]
    ra, barcodes = badger.import_tsv(os.path.join(golden_dir, "c1_expected.tsv"), 16)
    assert barcodes == g["barcodes"] and len(ra) == 1000
    graph = BarcodeGraph(1)
    uniq = graph.index_barcodes(barcodes, 16)
    assert [[k, v] for k, v in graph.counts.items()] == g["counts"]
    assert sorted(graph.counts) == uniq.tolist()
    for a, b, d in g["edges"]:
        graph.edges[a].append(b); graph.edges[b].append(a)
        graph.dists[(a, b)] = d; graph.dists[(b, a)] = d
    wl = set(open(os.path.join(golden_dir, "c1_whitelist.txt")).read().split("\n"))
    with redirect_stdout(io.StringIO()) as out:
        graph.cluster(None, wl, 50, 16, 25)
    assert out.getvalue() == "1\n2\n"
    prefix = str(tmp_path / "o")
    graph.output_file(ra, prefix, None, 16, False)
    assert open(prefix + "_output_file.tsv").read() == open(os.path.join(golden_dir, "c1_stage2_output_file.tsv")).read()
    assert str(len(graph.counts) - len(graph.edges)) == open(os.path.join(golden_dir, "c1_stage2_stdout_tail.txt")).read().strip()


def test_import_tsv_quirks(tmp_path):
    p = tmp_path / "in.tsv"
    hdr = "#read_id\tbarcode\tUMI\tBC_score\tvalid_UMI\tstrand\tpolyT_start\tR1_end\n"
    p.write_text(hdr + "r1\tACGTACGTACGTACGT\tAAA\t0\tFalse\t+\t1\t2\n" + hdr +
                 "r2\t*\t*\t-1\tFalse\t.\t-1\t-1\n" + "r3\t\tAA\t0\tFalse\t+\t1\t2\n" +
                 "r4\tACGTACGTACGTACGTA\tAAA\t0\tFalse\t+\t1\t2\n" + "r5\tACG\tAAA\t0\tFalse\t+\t1\t2\n")
    ra, bcs = badger.import_tsv(str(p), 16)
    assert ra == [("r1", "ACGTACGTACGTACGT"), ("r2", "*"), ("r3", "*"), ("r4", "ACGTACGTACGTACGT"), ("r5", "ACG")]
    assert bcs == ["ACGTACGTACGTACGT", "ACGTACGTACGTACGTA", "ACG"]


def test_cli_argument_surface():
    a = erb.parse_args(["-o", "x", "-i", "y.fq", "--mode", "tenX_v2", "-t", "3", "--tmp_dir", "/tmp"])
    assert (a.output, a.input, a.mode, a.threads) == ("x", "y.fq", "tenX_v2", 3)
    with pytest.raises(SystemExit):
        erb.parse_args(["-o", "x", "-i", "y.fq", "--barcodes", "wl.txt"])     # the reference has no such flag either
    b = badger.parse_args(["-r", "in.tsv", "-d", "tenX_v3", "-l", "wl", "-c", "50", "-hs", "-tr", "2", "-t", "2", "-i", "10"])
    assert (b.reads, b.data_type, b.n_cells, b.high_sens, b.threads, b.threshold, b.interval) == ("in.tsv", "tenX_v3", 50, True, 2, 2, 10)
    with pytest.raises(SystemExit):
        badger.parse_args(["-r", "x", "-d", "visium"])


def test_synthetic_workload_is_pinned_and_chunk_invariant():
    """The bench / test workload is a pure function of (n, whitelist, seed): every draw is an integer hash of
    (seed, purpose, index), tables come from numpy on the host.  Pinned here on the CPU; tests/test_hip_parity.py checks
    that a GPU produces the same bytes."""
    import hashlib
    from badger_amd import synth
    wl = synth.make_whitelist(1000)
    assert hashlib.sha256(wl.tobytes()).hexdigest() == "f71ad117b257e8663d6b99525d3e3d29a56204104d4d9b0f4a45390500bca228"
    b, o = synth.make_reads(2000, wl, seed=1)
    assert int(o[-1]) == 2079674
    assert hashlib.sha256(b.numpy().tobytes()).hexdigest() == "276616e82f7c24b9e9fdc06dfe31f1c058602e3c88ac92457227b432906dcb95"
    b2, o2 = synth.make_reads(2000, wl, seed=1, chunk=333)           # chunking is an implementation detail
    assert bool((o == o2).all()) and bool((b == b2).all())
    b3, o3 = synth.make_reads(2000, wl, seed=2)
    assert int(o3[-1]) != int(o[-1])


def test_rank_packing_and_the_whitelist_filter():
    """common._pack (an OR-reduction of shifted 2-bit codes) against rank() for several lengths; rank_valid_many keeps
    exactly the strings an unrank() output can equal (right length, ACGT only), in input order."""
    from badger_amd.common import rank, rank_many, rank_valid_many
    rng = np.random.default_rng(5)
    for length in (1, 12, 16, 17, 31):
        seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, length)) for _ in range(300)]
        want = [rank(s, length) for s in seqs]
        assert [int(x) for x in rank_many(seqs, length)] == want
        assert [int(x) for x in rank_valid_many(seqs, length)] == want
    mixed = ["ACGTACGTACGTACGT", "ACGT", "", "NCGTACGTACGTACGT", "acgtacgtacgtacgt", "TTTTTTTTTTTTTTTT", "ACGTACGTACGTACGé", "ACGTACGTACGTACGTA"]
    assert [int(x) for x in rank_valid_many(mixed, 16)] == [rank(mixed[0], 16), rank(mixed[5], 16)]
    assert len(rank_valid_many([], 16)) == 0 and len(rank_valid_many(["AC"], 16)) == 0
    with pytest.raises(KeyError):
        rank_many(["ACGTACGTACGTACGN"], 16)
