"""badger_amd.stage2 (stage 2 on arrays) against the dictionary mirror of the reference's BarcodeGraph
(badger_amd.barcode_graph) and against the reference's own output fixture; edges come from the CPU oracle, no GPU."""
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from badger_amd import badger, synth
from badger_amd.barcode_graph import BarcodeGraph
from badger_amd.stage2 import Stage2, observed_from_strings, unrank_many
from oracle import pyoracle as orc


def _edges_from_oracle(st2, thr):
    e = orc.graph_edges(st2.uniq, thr, threads=4)
    st2.ea = np.searchsorted(st2.uniq, e["a"]).astype(np.intp)
    st2.eb = np.searchsorted(st2.uniq, e["b"]).astype(np.intp)
    return e


def _dict_graph(barcodes, e, thr):
    g = BarcodeGraph(thr)
    g.index_barcodes(barcodes, 16)
    g._take_edges(e["a"].tolist(), e["b"].tolist(), e["dist"].tolist())
    return g


def _reads(n_cells, n_reads, seed, extra=()):
    rng = np.random.default_rng(seed)
    cells = rng.integers(0, 1 << 32, n_cells, dtype=np.uint64)
    # every other cell is one or two substitutions off its predecessor: neighbouring clusters that fight over barcodes
    for k in range(1, n_cells, 2):
        c = cells[k - 1] ^ (np.uint64(1 + k % 3) << np.uint64(2 * (k % 16)))
        cells[k] = c ^ (np.uint64(2) << np.uint64(2 * ((k + 5) % 16))) if k % 4 == 1 else c
    # skewed cell sizes, a substitution or two, sometimes a deletion: a graph with conflicts on both levels
    w = np.exp(rng.standard_normal(n_cells))
    pick = cells[rng.choice(n_cells, n_reads, p=w / w.sum())]
    for _ in range(2):
        hit = rng.random(n_reads) < 0.3
        pick = np.where(hit, pick ^ (rng.integers(1, 4, n_reads).astype(np.uint64) << (2 * rng.integers(0, 16, n_reads).astype(np.uint64))), pick)
    bcs = unrank_many(pick.astype(np.uint32))
    out = []
    for i, s in enumerate(bcs):
        u = rng.random()
        out.append("*" if u < 0.03 else (s + "A" if u < 0.06 else (s[:9] if u < 0.07 else s)))
    return ["r%d" % i for i in range(n_reads)], out + list(extra)


def test_unrank_many_and_observed():
    rng = np.random.default_rng(0)
    r = rng.integers(0, 1 << 32, 50, dtype=np.uint64).astype(np.uint32)
    assert unrank_many(r) == [synth.rank_to_str(x) for x in r] and unrank_many(np.zeros(0, np.uint32)) == []
    ranks, usable = observed_from_strings(["*", "ACGTACGTACGTACGT", "ACGTACGTACGTACGTA", "ACG", ""])
    assert usable.tolist() == [False, True, True, False, False] and ranks[1] == ranks[2] == synth.str_to_rank("ACGTACGTACGTACGT")
    with pytest.raises(KeyError):
        observed_from_strings(["ACGTACGTACGTACGN"])


def test_centre_selection_walks_the_dictionary_mirrors_order():
    """get_cluster_centers sorts only the barcodes above the cutoff and produces the rest of the order when a loop walks into
    it: against the dictionary mirror of the reference's loops (barcode_graph.py:252-277) on random count tables - flat, wide
    and heavy-tailed counts, with and without a barcode list, cell numbers around and beyond the number of barcodes."""
    from badger_amd.barcode_graph import BarcodeGraph
    from badger_amd.common import BarcodeRanks
    rng = np.random.default_rng(11)
    tails = 0
    for it in range(400):
        nu = int(rng.integers(1, 300))
        st2 = Stage2(1)
        st2.uniq = np.sort(rng.choice(1 << 20, nu, replace=False)).astype(np.uint32)
        st2.first = rng.permutation(nu * 3)[:nu].astype(np.int64)
        st2.count = [rng.integers(1, 4, nu), rng.integers(1, 60, nu), (rng.pareto(1.0, nu) * 3 + 1).astype(np.int64)][it % 3].astype(np.int64)
        g = BarcodeGraph(1)
        for i in np.argsort(st2.first):
            g.counts[int(st2.uniq[i])] = int(st2.count[i])
        n_cells, interval = int(rng.integers(1, nu + 30)), int(rng.integers(0, 60))
        listed = np.unique(rng.choice(st2.uniq, max(1, nu // 2))) if it % 2 else None
        wl_set = {synth.rank_to_str(int(r)) for r in listed} if listed is not None else None
        wl_arr = BarcodeRanks(listed) if listed is not None else None

        def run(f, wl):
            try:
                return f(None, 16, wl, n_cells, interval)
            except IndexError:
                return "IndexError"
        want, got = run(g.get_cluster_centers, wl_set), run(st2.get_cluster_centers, wl_arr)
        assert got == want, (it, nu, n_cells, interval)
        tails += isinstance(want, list) and len(want) > int((st2.count > 5).sum())
    assert tails > 20                                  # the walk past the cutoff happened


@pytest.mark.parametrize("seed,thr,n_cells,hs_wl", [(1, 1, 40, "list"), (2, 2, 25, "list"), (3, 1, 30, "none"), (4, 2, 60, "true"), (5, 1, 8, "list")])
def test_arrays_equal_dictionary_mirror(tmp_path, seed, thr, n_cells, hs_wl):
    ids, bcs = _reads(n_cells, 6000, seed)
    obs_rank, usable = observed_from_strings(bcs)
    st2 = Stage2(thr)
    st2.count_host(obs_rank, usable)
    e = _edges_from_oracle(st2, thr)
    read_assignment = [(i, b[:-1] if len(b) == 17 else b) for i, b in zip(ids, bcs)]
    g = _dict_graph([b for b in bcs if b != "*"], e, thr)
    assert [int(st2.uniq[i]) for i in np.argsort(st2.first, kind="stable")] == list(g.counts.keys())
    assert [int(st2.count[i]) for i in np.argsort(st2.first, kind="stable")] == list(g.counts.values())
    rng = np.random.default_rng(seed)
    top = [synth.rank_to_str(k) for k, _ in sorted(g.counts.items(), key=lambda kv: -kv[1])[:200]]
    wl = set(top[::2]) | {"", "ACGT"} if hs_wl == "list" else None
    true_bcs = set(top[:15]) | {synth.rank_to_str(int(rng.integers(0, 1 << 32)))} if hs_wl == "true" else None
    n_c = max(4, n_cells // 2)
    with redirect_stdout(io.StringIO()) as o1:
        st2.cluster(true_bcs, wl, n_c, 16, 25)
    with redirect_stdout(io.StringIO()) as o2:
        g.cluster(true_bcs, wl, n_c, 16, 25)
    assert o1.getvalue() == o2.getvalue() == "1\n2\n"
    assert st2.centers == g.get_cluster_centers(true_bcs, 16, wl, n_c, 25)
    # per-barcode assignment
    want = g.assign_by_cluster(16)
    got = st2.assigned_rank(False)
    for i, r in enumerate(st2.uniq):
        s = synth.rank_to_str(int(r))
        assert (synth.rank_to_str(int(got[i])) if got[i] != 0xFFFFFFFF else "") == want[s]
    assert (got != 0xFFFFFFFF).sum() > 50 and (st2.owner == -1).sum() > 0           # both outcomes occur, conflicts too
    # output file and the printed count
    p1, p2 = str(tmp_path / "a"), str(tmp_path / "b")
    st2.output_file(ids, obs_rank, usable, p1, False)
    g.output_file(read_assignment, p2, true_bcs, 16, False)
    assert open(p1 + "_output_file.tsv").read() == open(p2 + "_output_file.tsv").read()
    assert st2.disconnected() == len(g.counts) - len(g.edges)


def test_arrays_reproduce_the_references_stage2_fixture(tmp_path, golden_dir):
    """the reference's own run (tests/golden): stage-1 TSV in, its edge list, its output file and its printed count out"""
    g = json.load(open(os.path.join(golden_dir, "graph.json")))["c1_thr1"]
    ra, _ = badger.import_tsv(os.path.join(golden_dir, "c1_expected.tsv"), 16)
    obs_rank, usable = observed_from_strings([b for _, b in ra])
    st2 = Stage2(1)
    st2.count_host(obs_rank, usable)
    assert [[int(st2.uniq[i]), int(st2.count[i])] for i in np.argsort(st2.first, kind="stable")] == g["counts"]
    ed = np.array(g["edges"], dtype=np.int64)
    st2.ea = np.searchsorted(st2.uniq, ed[:, 0].astype(np.uint32)).astype(np.intp)
    st2.eb = np.searchsorted(st2.uniq, ed[:, 1].astype(np.uint32)).astype(np.intp)
    wl = set(open(os.path.join(golden_dir, "c1_whitelist.txt")).read().split("\n"))
    with redirect_stdout(io.StringIO()):
        st2.cluster(None, wl, 50, 16, 25)
    prefix = str(tmp_path / "o")
    st2.output_file([r for r, _ in ra], obs_rank, usable, prefix, False)
    assert open(prefix + "_output_file.tsv").read() == open(os.path.join(golden_dir, "c1_stage2_output_file.tsv")).read()
    assert str(st2.disconnected()) == open(os.path.join(golden_dir, "c1_stage2_stdout_tail.txt")).read().strip()


def test_idstore_and_native_output_writer(tmp_path):
    from badger_amd import _native
    ids = ["r%d/x" % i for i in range(1000)] + ["", "a b"]
    st = _native.IdStore(ids[:400])
    st.extend(ids[400:])
    assert len(st) == len(ids) and st.to_list() == ids and st[401] == ids[401]
    rng = np.random.default_rng(3)
    rank = rng.integers(0, 1 << 32, len(ids), dtype=np.uint64).astype(np.uint32)
    rank[5] = 0xFFFFFFFF                                   # sixteen T: a barcode like any other
    has = (rng.random(len(ids)) < 0.7).astype(np.uint8)
    has[5] = 1
    p = str(tmp_path / "w.tsv")
    _native.write_assignments(st, rank, has, p)
    want = "readID\tbarcode\n" + "".join("%s\t%s\n" % (i, synth.rank_to_str(r) if h else "*") for i, r, h in zip(ids, rank, has))
    assert open(p).read() == want and "\tTTTTTTTTTTTTTTTT\n" in want
    with pytest.raises(_native.BadgerHipError):
        _native.write_assignments(st, rank[:-1], has[:-1], p)             # one id per read, or it is an error


@pytest.mark.gpu
@pytest.mark.parametrize("seed,thr,n_cells", [(1, 1, 40), (2, 2, 25), (4, 2, 60), (5, 1, 8), (6, 2, 300)])
def test_device_clustering_and_assignment_equal_the_array_code(seed, thr, n_cells):
    """bdg_cluster_dev (both levels with min / max atomics over the edge array) and bdg_assign_reads_dev against the numpy
    code, which the CPU tier checks against the dictionary mirror of the reference: graphs with conflicts on both levels"""
    from badger_amd import _native
    ids, bcs = _reads(n_cells, 20000 if n_cells > 100 else 6000, seed)
    obs_rank, usable = observed_from_strings(bcs)
    ref = Stage2(thr)
    ref.count_host(obs_rank, usable)
    _edges_from_oracle(ref, thr)
    with redirect_stdout(io.StringIO()):
        ref.cluster(None, None, max(4, n_cells // 2), 16, 25)
    ctx = _native.default_context(0)
    dev = Stage2(thr)
    dev.count_host(obs_rank, usable)
    dev.ea = dev.eb = None                                    # (the edges exist on the device only, as after build_edges)
    m = len(ref.ea)
    d_rows = _native.DeviceArray.from_host(ctx, np.stack([ref.ea, ref.eb]).astype(np.uint32) if m else np.zeros((2, 1), np.uint32))
    dev._dev = {"ctx": ctx, "rows": d_rows, "m": m, "uniq": _native.DeviceArray.from_host(ctx, dev.uniq)}
    with redirect_stdout(io.StringIO()) as o:
        dev.cluster(None, None, max(4, n_cells // 2), 16, 25)
    assert o.getvalue() == "1\n2\n"
    assert (dev.owner == ref.owner).all() and (ref.owner == -1).sum() > 0 and (ref.owner >= 0).sum() > n_cells // 2
    # the count badger.py prints, taken where the edges are (bdg_touched_count_dev) = the numpy form; then the host copy on demand
    assert dev._ea is None and dev.disconnected() == ref.disconnected() and dev._ea is None
    assert (dev.ea == ref.ea).all() and (dev.eb == ref.eb).all()
    dev.ea = dev.eb = None
    # per read, from extraction records on the device
    recs = np.zeros(len(obs_rank), dtype=_native.REC_DTYPE)
    recs["valid"] = 1
    recs["bc_rank"] = obs_rank
    recs["flags"] = np.where(usable, _native.FLAG_RANK_OK | _native.FLAG_BC16, 0)
    recs["valid"][::17] = 0                                   # reads without a barcode
    usable2 = usable & (recs["valid"] == 1)
    want_rank, want_has = ref.per_read(obs_rank, usable2, False)
    assigned, has = dev.assigned(False)
    d_recs = _native.DeviceArray.from_host(ctx, recs.view(np.uint8).reshape(-1, 32))
    d_a, d_h = _native.DeviceArray.from_host(ctx, assigned), _native.DeviceArray.from_host(ctx, has.astype(np.uint8))
    d_r, d_g = _native.DeviceArray(ctx, len(recs), np.uint32), _native.DeviceArray(ctx, len(recs), np.uint8)
    ctx.assign_reads_dev(d_recs, len(recs), dev._dev["uniq"], len(dev.uniq), d_a, d_h, d_r, d_g)
    got_rank, got_has = d_r.to_host(), d_g.to_host()
    assert (got_has == want_has).all() and (got_rank[want_has == 1] == want_rank[want_has == 1]).all() and want_has.sum() > 1000
    dev.release_device()


def test_barcode_list_file_as_ranks(tmp_path):
    """BarcodeRanks.from_file = the ranks of exactly those lines of the file that the reference's set of lines could match
    (16 letters of ACGT), whatever else the file holds; truthy even when empty, like the reference's set"""
    from badger_amd.common import BarcodeRanks, rank_valid_many
    rng = np.random.default_rng(4)
    good = unrank_many(rng.integers(0, 1 << 32, 500, dtype=np.uint64).astype(np.uint32))
    lines = good + ["", "ACGT", "ACGTACGTACGTACGTA", "ACGTACGTACGTACGN", "acgtacgtacgtacgt", good[3], "AAAAAAAAAAAAAAAA-1"]
    rng.shuffle(lines)
    for text in ("\n".join(lines) + "\n", "\n".join(lines), "\r\n".join(lines) + "\r\n", ""):
        p = tmp_path / "wl.txt"
        with open(p, "w", newline="") as f:
            f.write(text)
        with open(p) as f:
            ref = set(f.read().split("\n"))
        got = BarcodeRanks.from_file(str(p))
        assert bool(got) and sorted(got.ranks.tolist()) == sorted(set(rank_valid_many(ref).astype(np.uint32).tolist()))
    assert len(BarcodeRanks.from_file(str(p))) == 0


def test_native_tsv_import_equals_the_python_one(tmp_path, golden_dir):
    """bdg_import_stage1_tsv against badger.import_tsv + observed_from_strings (the restatement of badger.py:91-111): the
    reference's own stage-1 TSV, and a file with repeated headers, short rows, NA barcodes, 17-letter barcodes, CRLF"""
    from badger_amd import _native
    cases = [os.path.join(golden_dir, "c1_expected.tsv")]
    p = tmp_path / "odd.tsv"
    head = "#read_id\tbarcode\tUMI\tBC_score\tvalid_UMI\tstrand\tpolyT_start\tR1_end"
    rows = [head, "r1\tACGTACGTACGTACGT\tAAAA\t0\tFalse\t+\t5\t3", "r2\t*\t*\t-1\tFalse\t.\t-1\t-1", head, "r3\tACGTACGTACGTACGTA\tAAAA\t0\tFalse\t+\t5\t3",
            "r4\tACG\tAAAA\t0\tFalse\t+\t5\t3", "r5", "r6\t\tx", "r7\tNA\tx", "r8\tTTTTTTTTTTTTTTTT\tx\t0"]
    p.write_text("\n".join(rows) + "\n")
    cases.append(str(p))
    q = tmp_path / "crlf.tsv"
    q.write_bytes(("\r\n".join(rows[:5]) + "\r\n").encode())
    cases.append(str(q))
    w = tmp_path / "swapped.tsv"
    w.write_text("barcode\tx\t#read_id\nACGTACGTACGTACGT\t1\tq1\n*\t2\tq2\nbarcode\t3\t#read_id\n")
    cases.append(str(w))
    for path in cases:
        ra, _ = badger.import_tsv(path, 16)
        want_rank, want_usable = observed_from_strings([b for _, b in ra])
        ids, rank, usable = _native.import_stage1_tsv(path, 16)
        assert ids.to_list() == [r for r, _ in ra] and (usable == want_usable).all() and (rank[usable] == want_rank[usable]).all(), path
    assert len(ids) == 2
    bad = tmp_path / "bad.tsv"
    bad.write_text(head + "\nr1\tACGTACGTACGTACGN\tAAAA\t0\tFalse\t+\t5\t3\n")
    with pytest.raises(KeyError):
        _native.import_stage1_tsv(str(bad), 16)
    with pytest.raises(KeyError):
        observed_from_strings([b for _, b in badger.import_tsv(str(bad), 16)[0]])
    nohead = tmp_path / "nohead.tsv"
    nohead.write_text("a\tb\nr1\tACGTACGTACGTACGT\n")
    with pytest.raises(ValueError):
        _native.import_stage1_tsv(str(nohead), 16)
    with pytest.raises(ValueError):
        badger.import_tsv(str(nohead), 16)


def test_native_tsv_import_in_ranges(tmp_path, monkeypatch):
    """The importer parses the file in line-aligned byte ranges, one thread each, and joins them in file order: 1, 3 and 5
    threads give the same ids / ranks / flags on a 5 MB file with odd rows sprinkled in (repeated headers, short rows, '*',
    17-letter barcodes, no final newline), and a bad letter far into the file is reported with its line number."""
    from badger_amd import _native
    rng = np.random.default_rng(3)
    head = "#read_id\tbarcode\tUMI\tBC_score\tvalid_UMI\tstrand\tpolyT_start\tR1_end"
    ranks = rng.integers(0, 1 << 32, 110000, dtype=np.uint64).astype(np.uint32)
    lines = [head]
    for i, r in enumerate(ranks):
        bc = synth.rank_to_str(int(r))
        if i % 997 == 0:
            lines.append(head)
        if i % 13 == 0:
            lines.append("read_%d\t*\t*\t-1\tFalse\t.\t-1\t-1" % i)
        elif i % 101 == 0:
            lines.append("read_%d" % i)
        elif i % 37 == 0:
            lines.append("read_%d\t%sA\tACGTACGTACGT\t0\tTrue\t+\t120\t60" % (i, bc))
        else:
            lines.append("read_%d\t%s\tACGTACGTACGT\t0\tTrue\t+\t120\t60" % (i, bc))
    p = tmp_path / "big.tsv"
    p.write_text("\n".join(lines))                          # (no newline behind the last row)
    assert p.stat().st_size > 5 << 20
    ra, _ = badger.import_tsv(str(p), 16)
    want_rank, want_usable = observed_from_strings([b for _, b in ra])
    for threads in ("1", "3", "5"):
        monkeypatch.setenv("BADGER_AMD_IMPORT_THREADS", threads)
        ids, rank, usable = _native.import_stage1_tsv(str(p), 16)
        assert len(ids) == len(ra) and ids.to_list() == [r for r, _ in ra], threads
        assert (usable == want_usable).all() and (rank[usable] == want_rank[usable]).all(), threads
    bad_at = len(lines) - 5000
    lines[bad_at] = "read_x\tACGTACGTACGTACGN\tAAAA\t0\tFalse\t+\t5\t3"
    p.write_text("\n".join(lines) + "\n")
    for threads in ("1", "4"):
        monkeypatch.setenv("BADGER_AMD_IMPORT_THREADS", threads)
        with pytest.raises(KeyError, match="line %d " % (bad_at + 1)):
            _native.import_stage1_tsv(str(p), 16)


def test_output_writer_in_ranges(tmp_path, monkeypatch):
    """bdg_write_assignments cuts the rows into ranges whose place in the file is known beforehand; every range is written
    by a thread of its own: 1, 3 and 7 threads give the file Python writes, for ids of uneven length and a mix of rows
    with and without a barcode"""
    from badger_amd import _native
    rng = np.random.default_rng(8)
    n = 300000
    names = ["read_%d%s" % (i, "x" * int(rng.integers(0, 9))) for i in range(n)]
    rank = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    has = (rng.random(n) < 0.7).astype(np.uint8)
    want = "readID\tbarcode\n" + "".join("%s\t%s\n" % (nm, synth.rank_to_str(int(r)) if h else "*") for nm, r, h in zip(names, rank, has))
    ids = _native.IdStore(names)
    for threads in ("1", "3", "7"):
        monkeypatch.setenv("BADGER_AMD_WRITE_THREADS", threads)
        out = str(tmp_path / ("o%s.tsv" % threads))
        _native.write_assignments(ids, rank, has, out)
        assert open(out).read() == want, threads


def test_edge_build_gpus_from_the_flags(monkeypatch):
    """badger.py: --gpus N as given; otherwise -tr N (the reference's compare_in_parallel process count) mapped onto the
    devices the node has, at least one"""
    from badger_amd import badger, _native
    monkeypatch.setattr(_native, "device_count", lambda: 8)
    monkeypatch.delenv("BADGER_AMD_CONTEXTS_ON_ONE_DEVICE", raising=False)
    base = ["-r", "x.tsv", "-d", "tenX_v3"]
    assert badger.edge_build_gpus(badger.parse_args(base)) == 1
    assert badger.edge_build_gpus(badger.parse_args(base + ["-tr", "4"])) == 4
    assert badger.edge_build_gpus(badger.parse_args(base + ["-tr", "32"])) == 8
    assert badger.edge_build_gpus(badger.parse_args(base + ["-tr", "32", "--gpus", "2"])) == 2
    monkeypatch.setattr(_native, "device_count", lambda: 1)
    assert badger.edge_build_gpus(badger.parse_args(base + ["-tr", "16"])) == 1
    monkeypatch.setenv("BADGER_AMD_CONTEXTS_ON_ONE_DEVICE", "1")
    assert badger.edge_build_gpus(badger.parse_args(base + ["-tr", "3"])) == 3


def test_native_tsv_import_on_rows_pandas_reads_its_own_way(tmp_path):
    """badger.py:91-111 reads the stage-1 TSV with pandas.read_csv: a row that ends before the barcode column stays a read
    without a barcode, a blank line is skipped, a quoted field loses its quotes, an id spelled like a missing value is
    written as an empty field, repeated headers are dropped; an empty file is an error (pandas: EmptyDataError), a header
    without rows is no reads."""
    import pandas as pd
    from badger_amd import _native
    p = tmp_path / "edge.tsv"
    p.write_text("#read_id\tbarcode\tUMI\n"
                 "r1\tACGTACGTACGTACGT\tx\n"
                 "r2\n"
                 "NA\tCCCCCCCCCCCCCCCC\ty\n"
                 "\"q1\"\tGGGGGGGGGGGGGGGG\tz\n"
                 "\n"
                 "r5\t\tw\n"
                 "#read_id\tbarcode\tUMI\n"
                 "r6\tNA\tw\n"
                 "r7\tTTTTTTTTTTTTTTTTA\tw\n")
    ids, rank, usable = _native.import_stage1_tsv(str(p), 16)
    # what the reference's own statements make of the file
    reads = pd.read_csv(str(p), sep="\t")
    rid = reads["#read_id"].tolist()
    observed = reads["barcode"].fillna("*").tolist()
    want = []
    for i in range(len(rid)):
        if rid[i] != "#read_id" and observed[i] != "barcode":
            o = observed[i]
            want.append((rid[i], o[:-1] if len(o) == 17 else o))
    assert len(ids) == len(want) == 7
    got_ids = [ids[i] for i in range(len(ids))]
    assert got_ids == ["" if isinstance(r, float) else r for r, _ in want]          # (NaN -> to_csv writes an empty field)
    assert usable.tolist() == [o != "*" for _, o in want]
    from badger_amd import synth
    assert [synth.rank_to_str(int(r)) for r, u in zip(rank, usable) if u] == [o for _, o in want if o != "*"]
    empty = tmp_path / "empty.tsv"
    empty.write_text("")
    with pytest.raises(ValueError):
        _native.import_stage1_tsv(str(empty), 16)
    header_only = tmp_path / "header_only.tsv"
    header_only.write_text("#read_id\tbarcode\tUMI\n")
    ids, rank, usable = _native.import_stage1_tsv(str(header_only), 16)
    assert len(ids) == 0 and len(rank) == 0 and len(usable) == 0
