"""End-to-end CLI runs on the GPU: stage 1 TSV/.stats and stage 2 output byte-identical to the
fixtures the reference produced (tests/golden/)."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from badger_amd import badger, extract_raw_barcodes as erb

pytestmark = pytest.mark.gpu


def test_stage1_single_thread_matches_reference(tmp_path, golden_dir):
    out = str(tmp_path / "out.tsv")
    erb.main(["--mode", "tenX_v3", "-i", os.path.join(golden_dir, "c1_reads.fa.gz"), "-o", out, "-t", "1"])
    assert open(out).read() == open(os.path.join(golden_dir, "c1_expected.tsv")).read()
    assert open(out + ".stats").read() == open(os.path.join(golden_dir, "c1_expected.tsv.stats")).read()


def test_stage1_parallel_shape(tmp_path, golden_dir, monkeypatch):
    monkeypatch.setattr(erb, "READ_CHUNK_SIZE", 300)
    monkeypatch.setattr(erb.read_chunks, "__defaults__", (300,))
    out = str(tmp_path / "outp.tsv")
    erb.main(["--mode", "tenX_v3", "-i", os.path.join(golden_dir, "c1_reads.fa.gz"), "-o", out, "-t", "4"])
    want = open(os.path.join(golden_dir, "c1_expected.tsv")).read().split("\n")
    got = open(out).read().split("\n")
    assert [l for l in got if not l.startswith("#")] == [l for l in want if not l.startswith("#")]
    assert sum(l.startswith("#read_id") for l in got) == 4          # 300+300+300+100: one header per chunk
    stats = dict(l.rsplit(" ", 1) for l in open(out + ".stats").read().strip().split("\n"))
    assert stats["Total reads:"] == "1000" and stats["Barcode detected:"] == "993"


def _fastq_of(tmp_path, n, seed):
    """n synthetic reads as a FASTQ file + the TSV rows the oracle's records give for them"""
    import numpy as np
    from badger_amd import synth
    from badger_amd.barcode_extraction.barcode_callers import record_to_row
    from oracle import pyoracle as orc
    wl = synth.make_whitelist(2000)
    bases, off = synth.make_reads(n, wl, seed=seed)
    seqs = synth.reads_to_list(bases, off)
    path = str(tmp_path / "reads.fastq")
    with open(path, "w") as f:
        f.write("".join("@read_%d len=%d\n%s\n+\n%s\n" % (i, len(s), s, "I" * len(s)) for i, s in enumerate(seqs)))
    recs = orc.extract_batch(bases.numpy(), off.numpy().astype(np.uint64), 12, threads=16)
    rows = [record_to_row("read_%d" % i, s, r) for i, (s, r) in enumerate(zip(seqs, recs))]
    return path, rows, recs


def test_stage1_fastq_through_the_pipeline(tmp_path):
    """250,000 reads = three chunks (100K, 100K, 50K) through parser thread -> pinned chunks -> submit / collect ->
    native formatter: the TSV and .stats the CLI writes must be what the oracle's records give, in both file shapes."""
    path, rows, recs = _fastq_of(tmp_path, 250000, 21)
    header = "#read_id\tbarcode\tUMI\tBC_score\tvalid_UMI\tstrand\tpolyT_start\tR1_end"
    out = str(tmp_path / "o1.tsv")
    erb.main(["--mode", "tenX_v3", "-i", path, "-o", out, "-t", "1"])
    assert open(out).read() == "\n".join([header] + rows) + "\n"
    st = dict(l.split("\t") for l in open(out + ".stats").read().strip().split("\n"))
    assert st["Total reads:"] == "250000" and st["Barcode detected:"] == str(int(recs["valid"].sum()))
    assert st["PolyT detected:"] == str(int((recs["polyT"] != -1).sum())) and st["R1 detected:"] == str(int((recs["r1_end"] != -1).sum()))
    out = str(tmp_path / "o4.tsv")
    erb.main(["--mode", "tenX_v3", "-i", path, "-o", out, "-t", "4"])
    want = [header] + rows[:100000] + [header] + rows[100000:200000] + [header] + rows[200000:]
    assert open(out).read() == "\n".join(want) + "\n"


def test_pipeline_over_several_contexts_keeps_chunk_order(tmp_path):
    """--gpus N: chunk k goes to detector k mod N, two chunks in flight per detector, rows written in chunk order.
    Rehearsed on one GPU with three independent contexts (own streams and workspaces) standing in for three devices,
    with chunk sizes that do not divide the input; plus an input that ends exactly on a chunk boundary (the trailing
    empty chunk of the reference's generator)."""
    from badger_amd.barcode_extraction.barcode_callers import TenXBarcodeExtractorV3
    path, rows, recs = _fastq_of(tmp_path, 9000, 22)
    for ndet, size in ((3, 700), (2, 1000), (1, 4500), (3, 9000)):
        dets = [TenXBarcodeExtractorV3(device=0, instance=i) for i in range(ndet)]
        got, sizes = [], []

        def on_chunk(text, r):
            got.append(text)
            sizes.append(len(r))
        assert erb.run_fastx_pipeline(path, dets, on_chunk, chunk_size=size) == 9000
        assert b"".join(got).decode() == "\n".join(rows) + "\n"
        assert sum(sizes) == 9000 and max(sizes) <= size and len(sizes) <= 9000 // size + 2       # (the 18 MB file is two parse segments)
    # the same through the native file-to-file pipeline: contexts x parse segments x both file shapes; the header goes in
    # front of every `every` reads and once more when the input ends on a multiple (the reference's trailing empty chunk)
    from badger_amd import _native
    header = "#read_id\tbarcode\tUMI\tBC_score\tvalid_UMI\tstrand\tpolyT_start\tR1_end"
    out = str(tmp_path / "n.tsv")
    for ndet, seg, every, chunk in ((3, 0, 0, 700), (3, 1 << 20, 1000, 0), (2, 300000, 4500, 1300), (1, 50000, 9000, 0), (3, 2000, 7, 50)):
        ctxs = [_native.default_context(0, i) for i in range(ndet)]
        res = _native.stage1_run(ctxs, path, out, header, 12, threads=3, header_every=every, chunk_reads=chunk, segment_bytes=seg)
        want = []
        for i, r in enumerate(rows):
            if (every and i % every == 0) or (not every and i == 0):
                want.append(header)
            want.append(r)
        if every and len(rows) % every == 0:
            want.append(header)
        assert open(out).read() == "\n".join(want) + "\n", (ndet, seg, every)
        assert (res.reads, res.barcodes, res.polyt, res.r1) == (9000, int(recs["valid"].sum()), int((recs["polyT"] != -1).sum()), int((recs["r1_end"] != -1).sum()))
        assert res.first_polyt == int(np.argmax(recs["polyT"] != -1)) and res.first_r1 == int(np.argmax(recs["r1_end"] != -1))


def test_stage1_empty_and_tiny_inputs(tmp_path):
    """no reads at all, one read, reads with N and of length 0: the pipeline must write what the chunk loop of the
    reference writes (a header, the rows, a .stats with the right totals)"""
    from badger_amd import synth
    from badger_amd.barcode_extraction.barcode_callers import record_to_row
    from oracle import pyoracle as orc
    header = "#read_id\tbarcode\tUMI\tBC_score\tvalid_UMI\tstrand\tpolyT_start\tR1_end\n"
    empty = tmp_path / "e.fastq"
    empty.write_text("")
    for t in ("1", "3"):
        out = str(tmp_path / ("e%s.tsv" % t))
        erb.main(["--mode", "tenX_v3", "-i", str(empty), "-o", out, "-t", t])
        assert open(out).read() == header
        assert "Total reads:" in open(out + ".stats").read() and " 0" in open(out + ".stats").read().replace("\t", " ")
    seqs = ["", "ACGTNACGT", "N" * 50,
            "GATTACA" * 3 + "CTACACGACGCTCTTCCGATCT" + "ACGTACGTACGTACGT" + "AACCGGTTAACC" + "T" * 25 + "GATTACAGATTACA" * 4,
            "GATTACA" * 3 + "CTACACGACGCTCTTCCGATCT" + "ACGTACGTNCGTACGT" + "AACCGGTTAACC" + "T" * 25 + "GATTACAGATTACA" * 4]
    fa = tmp_path / "t.fa"
    fa.write_text("".join(">s%d\n%s\n" % (i, s) for i, s in enumerate(seqs)))
    out = str(tmp_path / "t.tsv")
    erb.main(["--mode", "tenX_v3", "-i", str(fa), "-o", out, "-t", "1"])
    b, o = synth.list_to_reads(seqs)
    recs = orc.extract_batch(b, o, 12, threads=2)
    assert open(out).read() == header + "".join(record_to_row("s%d" % i, s, r) + "\n" for i, (s, r) in enumerate(zip(seqs, recs)))
    assert recs["valid"].tolist() == [0, 0, 0, 1, 1]


def test_stage1_bad_base_raises_keyerror(tmp_path):
    p = tmp_path / "bad.fq"
    p.write_text("@a\nACGTACGTACGTACGTACGTAC\n+\n" + "I" * 22 + "\n@b\nACGTXCGT\n+\nIIIIIIII\n")
    with pytest.raises(KeyError):
        erb.main(["--mode", "tenX_v3", "-i", str(p), "-o", str(tmp_path / "o.tsv"), "-t", "1"])


def test_stage2_matches_reference(tmp_path, golden_dir):
    for hs in (False, True):
        prefix = str(tmp_path / ("s2hs" if hs else "s2"))
        argv = ["-r", os.path.join(golden_dir, "c1_expected.tsv"), "-d", "tenX_v3",
                "-l", os.path.join(golden_dir, "c1_whitelist.txt"), "-c", "50", "-o", prefix]
        if hs:
            argv.append("-hs")
        buf = io.StringIO()
        with redirect_stdout(buf):
            badger.main(argv)
        if hs:
            # the --high_sens pass matched against ~50 centres: the exhaustive kernel, no probe index was built for it
            from badger_amd import _native
            assert _native.default_context(0).nearest16_index_bytes() == 0
        name = "c1_stage2%s" % ("_hs" if hs else "")
        assert buf.getvalue().strip().split("\n")[-1] == open(os.path.join(golden_dir, name + "_stdout_tail.txt")).read().strip()
        got = open(prefix + "_output_file.tsv").read().split("\n")
        want = open(os.path.join(golden_dir, name + "_output_file.tsv")).read().split("\n")
        if not hs:
            assert got == want
        else:
            # High-sensitivity mode sends every unassigned barcode to its nearest centre (barcode_graph.py:370-385).  The
            # reference walks a set of strings, so among equally near centres its choice follows the hash seed; every
            # row that differs from the fixture must therefore be exactly such a tie: both centres at the same
            # Levenshtein distance (< 3) from the observed barcode, and the same centre for every read of that barcode.
            from oracle import pyoracle as orc
            assert len(got) == len(want)
            observed = {}
            for line in open(os.path.join(golden_dir, "c1_expected.tsv")).read().split("\n")[1:]:
                if line:
                    f = line.split("\t")
                    observed[f[0]] = f[1]
            ndiff = 0
            for g, w in zip(got, want):
                if g == w:
                    continue
                rid, gb = g.split("\t")
                rid2, wb = w.split("\t")
                assert rid == rid2 and gb != "*" and wb != "*"
                obs = observed[rid][:16]
                dg, dw = orc.levenshtein(obs, gb), orc.levenshtein(obs, wb)
                assert dg == dw and dg < 3, (rid, obs, gb, wb, dg, dw)
                ndiff += 1
            assert ndiff <= 0.02 * len(got)


def test_distinct_dev_against_the_references_counts(golden_dir):
    """bdg_distinct_dev (sort + run-length on the device) against BarcodeGraph.counts as the reference built it
    (index_bc_single_thread, barcode_graph.py:192-204; fixture from the reference's own run): same distinct ranks, same
    multiplicities, and - ordered by first occurrence - the same dict order."""
    import json
    import numpy as np
    import torch
    from badger_amd import _native, synth
    g = json.load(open(os.path.join(golden_dir, "graph.json")))
    ctx = _native.Context(0)
    for key in ("c1_thr1", "cells60_thr1"):
        case = g[key]
        recs = np.zeros(len(case["barcodes"]) + 3, dtype=_native.REC_DTYPE)
        for i, s in enumerate(case["barcodes"]):
            if len(s) == 17:
                s = s[:-1]
            if len(s) == 16:                     # anything else is dropped by the reference (:199-200)
                recs[i]["valid"], recs[i]["flags"], recs[i]["bc_rank"] = 1, _native.FLAG_RANK_OK | _native.FLAG_BC16, synth.str_to_rank(s)
            else:
                recs[i]["valid"], recs[i]["flags"] = 1, 0
        # three reads without a barcode at the end
        d_recs = torch.from_numpy(recs.view(np.int32).reshape(-1, 8).copy()).cuda()
        n = len(recs)
        uq = torch.zeros(n, dtype=torch.int32, device="cuda")
        ct = torch.zeros(n, dtype=torch.int32, device="cuda")
        fi = torch.zeros(n, dtype=torch.int32, device="cuda")
        dn = torch.zeros(2, dtype=torch.int32, device="cuda")
        ctx.distinct_dev(d_recs, n, uq, ct, fi, dn)
        ctx.synchronize()
        nu = int(dn[0])
        order = np.argsort(fi[:nu].cpu().numpy(), kind="stable")
        got = [[int(uq[:nu].cpu().numpy().view(np.uint32)[i]), int(ct[:nu].cpu().numpy()[i])] for i in order]
        assert got == case["counts"], key
    ctx.close()


def test_stage2_handoff_on_device_equals_the_tsv_route(tmp_path):
    """badger.py on a FASTQ (extraction records stay on the device -> bdg_distinct_dev -> bdg_graph_edges_dev) must write
    what it writes from the stage-1 TSV of the same reads (native import -> bdg_keep_observed -> the same device code), at
    both thresholds; both must really have taken the device route, and both must equal what the host array code makes of
    the TSV (numpy counting, searchsorted per read)."""
    from badger_amd import stage2, _native
    from badger_amd.common import BarcodeRanks
    path, rows, recs = _fastq_of(tmp_path, 30000, 23)
    tsv = str(tmp_path / "s1.tsv")
    erb.main(["--mode", "tenX_v3", "-i", path, "-o", tsv, "-t", "1"])
    wl = str(tmp_path / "wl.txt")
    from badger_amd import synth
    with open(wl, "w") as f:
        f.write("\n".join(synth.rank_to_str(r) for r in synth.make_whitelist(2000)) + "\n")
    for thr in ("1", "2"):
        outs = []
        for k, reads in enumerate((tsv, path)):
            prefix = str(tmp_path / ("o%s_%d" % (thr, k)))
            calls, observed = [], []
            orig, orig_keep = stage2.Stage2.count_device, _native.Context.keep_observed

            def spy(self, ctx, _orig=orig, _calls=calls):
                _calls.append(1)
                return _orig(self, ctx)

            def spy_keep(self, rank, usable, _orig=orig_keep, _calls=observed):
                _calls.append(len(rank))
                return _orig(self, rank, usable)
            stage2.Stage2.count_device, _native.Context.keep_observed = spy, spy_keep
            try:
                with redirect_stdout(io.StringIO()):
                    badger.main(["-r", reads, "-d", "tenX_v3", "-l", wl, "-c", "300", "-t", thr, "-o", prefix] + (["-hs"] if thr == "2" else []))
            finally:
                stage2.Stage2.count_device, _native.Context.keep_observed = orig, orig_keep
            assert len(calls) == 1 and observed == ([30000] if k == 0 else [])      # both count on the device; the TSV's barcodes were sent there
            outs.append(open(prefix + "_output_file.tsv").read())
        assert outs[0] == outs[1] and outs[0].count("\n") == 30001
        assert sum(1 for l in outs[0].split("\n")[1:] if l and not l.endswith("*")) > 10000
        # the host array code on the same TSV
        ids, obs, usable = _native.import_stage1_tsv(tsv, 16)
        st = stage2.Stage2(int(thr))
        st.count_host(obs, usable)
        st.build_edges()
        with redirect_stdout(io.StringIO()):
            st.cluster(None, BarcodeRanks.from_file(wl, 16), 300, 16, 25)
        prefix = str(tmp_path / ("h%s" % thr))
        st.output_file(ids, obs, usable, prefix, thr == "2")
        st.release_device()
        assert open(prefix + "_output_file.tsv").read() == outs[0]


def test_stage2_from_fastx_input(tmp_path, golden_dir):
    prefix = str(tmp_path / "s2fx")
    buf = io.StringIO()
    with redirect_stdout(buf):
        badger.main(["-r", os.path.join(golden_dir, "c1_reads.fa.gz"), "-d", "tenX_v3",
                     "-l", os.path.join(golden_dir, "c1_whitelist.txt"), "-c", "50", "-o", prefix])
    assert open(prefix + "_output_file.tsv").read() == open(os.path.join(golden_dir, "c1_stage2_output_file.tsv")).read()


def test_command_lines_run_without_torch(tmp_path, golden_dir):
    """Both command lines as a user starts them (python -m ...): same files as the in-process runs above, and torch is
    never imported on the way (device buffers come from bdg_mem_alloc, the device count from bdg_device_count)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    reads = os.path.join(golden_dir, "c1_reads.fa.gz")
    tsv = str(tmp_path / "s1.tsv")
    r = subprocess.run([sys.executable, "-X", "importtime", "-m", "badger_amd.extract_raw_barcodes", "--mode", "tenX_v3", "-i", reads,
                        "-o", tsv, "-t", "1"], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "| torch" not in r.stderr and "badger_amd._native" in r.stderr
    assert open(tsv).read() == open(os.path.join(golden_dir, "c1_expected.tsv")).read()
    prefix = str(tmp_path / "s2")
    r = subprocess.run([sys.executable, "-X", "importtime", "-m", "badger_amd.badger", "-r", reads, "-d", "tenX_v3",
                        "-l", os.path.join(golden_dir, "c1_whitelist.txt"), "-c", "50", "-o", prefix], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "| torch" not in r.stderr and "badger_amd.stage2" in r.stderr
    assert open(prefix + "_output_file.tsv").read() == open(os.path.join(golden_dir, "c1_stage2_output_file.tsv")).read()


def test_stage1_from_bgzf_and_gzip_inputs(tmp_path):
    """The same 120,000 reads as plain FASTQ, as BGZF (blocks inflated by a pool of threads) and as plain gzip (one zlib
    stream): the three TSVs are identical, rows as the oracle's records give them; -t sets the inflate threads."""
    import gzip
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_ingest import _bgzf
    path, rows, recs = _fastq_of(tmp_path, 120000, 29)
    raw = open(path, "rb").read()
    header = "#read_id\tbarcode\tUMI\tBC_score\tvalid_UMI\tstrand\tpolyT_start\tR1_end"
    want = "\n".join([header] + rows) + "\n"
    bg = str(tmp_path / "reads_bgzf.fastq.gz")
    open(bg, "wb").write(_bgzf(raw, level=1))
    gz = str(tmp_path / "reads_plain.fastq.gz")
    with gzip.open(gz, "wb", compresslevel=1) as f:
        f.write(raw)
    for k, src in enumerate((path, bg, gz)):
        out = str(tmp_path / ("o%d.tsv" % k))
        erb.main(["--mode", "tenX_v3", "-i", src, "-o", out, "-t", "1"])
        assert open(out).read() == want, src
    out = str(tmp_path / "o_t3.tsv")
    erb.main(["--mode", "tenX_v3", "-i", bg, "-o", out, "-t", "3"])           # three inflate threads, one header per chunk
    assert open(out).read() == "\n".join([header] + rows[:100000] + [header] + rows[100000:]) + "\n"


def _c1_reads(golden_dir):
    return list(erb.open_reads(os.path.join(golden_dir, "c1_reads.fa.gz")))


@pytest.mark.parametrize("container", ["bam", "sam", "sam.gz"])
def test_stage1_from_sam_and_bam(tmp_path, golden_dir, container):
    """The config-1 reads as BAM / SAM written by tests/bamio.py, with secondary and supplementary copies sprinkled in:
    -t 1 (reference :110-118) uses every record, -t N (:144-145) only the primary ones, which gives exactly the reference's
    TSV for the FASTA form of the same reads."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bamio
    reads = _c1_reads(golden_dir)
    recs, every = [], []
    for i, (rid, seq) in enumerate(reads):
        flag = 4 if i % 3 else (16 if i % 2 else 0)
        recs.append((rid, flag, seq, [] if flag & 4 else [(len(seq), "M")], b"" if i % 2 else b"NMC\x00"))
        every.append(i)
        if i % 10 == 0:                                          # a secondary / supplementary copy of the read, clipped
            recs.append((rid, 256 if i % 20 else 2048, seq[:len(seq) // 2], [(len(seq) // 2, "M")]))
            every.append(-1)
    p = str(tmp_path / ("r." + container))
    if container == "bam":
        open(p, "wb").write(bamio.bgzf(bamio.bam_raw(recs), block=20000))
    else:
        import gzip
        data = bamio.sam_text(recs).encode()
        open(p, "wb").write(gzip.compress(data) if container.endswith("gz") else data)
    golden = open(os.path.join(golden_dir, "c1_expected.tsv")).read().split("\n")
    out = str(tmp_path / "o.tsv")
    erb.main(["--mode", "tenX_v3", "-i", p, "-o", out, "-t", "4"])
    got = open(out).read().split("\n")
    assert [l for l in got if not l.startswith("#")] == [l for l in golden if not l.startswith("#")]
    erb.main(["--mode", "tenX_v3", "-i", p, "-o", out, "-t", "1"])
    got = open(out).read().split("\n")
    assert got[0] == golden[0] and len(got) == len(recs) + 2
    assert [l for l, k in zip(got[1:], every) if k >= 0] == golden[1:-1]            # the primary records' rows are the reference's
    assert all(l.split("\t")[0] == r[0] for l, r in zip(got[1:], recs))
    # stage 2 straight from the BAM / SAM (records stay on the device), both thread settings of the reference
    prefix = str(tmp_path / "s2")
    with redirect_stdout(io.StringIO()):
        badger.main(["-r", p, "-d", "tenX_v3", "-l", os.path.join(golden_dir, "c1_whitelist.txt"), "-c", "50", "-o", prefix, "-tr", "2"])
    assert open(prefix + "_output_file.tsv").read() == open(os.path.join(golden_dir, "c1_stage2_output_file.tsv")).read()


def test_stage1_iupac_codes_in_bam_raise_keyerror(tmp_path):
    """a BAM sequence may hold any of =ACMGRSVTWYHKDBN; anything outside ACGTN raises KeyError in the reference's
    reverese_complement"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bamio
    p = str(tmp_path / "i.bam")
    open(p, "wb").write(bamio.bgzf(bamio.bam_raw([("a", 4, "ACGTNACGTACGTACGTTTTTT"), ("b", 4, "ACGTRYACGT")])))
    with pytest.raises(KeyError):
        erb.main(["--mode", "tenX_v3", "-i", p, "-o", str(tmp_path / "o.tsv"), "-t", "1"])
    open(p, "wb").write(bamio.bgzf(bamio.bam_raw([("a", 4, "ACGTNACGTACGTACGTTTTTT"), ("b", 4, "")])))
    with pytest.raises(TypeError):
        erb.main(["--mode", "tenX_v3", "-i", p, "-o", str(tmp_path / "o.tsv"), "-t", "1"])


def test_stage1_gpus_flag_over_contexts_of_one_device(tmp_path, monkeypatch):
    """--gpus 3 as the command line drives it (chunk k on context k mod 3, two in flight each, rows in input order), rehearsed
    on one device with three independent contexts: the TSV and .stats are those of the one-context run, in both file shapes"""
    path, rows, recs = _fastq_of(tmp_path, 40000, 31)
    header = "#read_id\tbarcode\tUMI\tBC_score\tvalid_UMI\tstrand\tpolyT_start\tR1_end"
    monkeypatch.setenv("BADGER_AMD_CONTEXTS_ON_ONE_DEVICE", "1")
    monkeypatch.setenv("BADGER_AMD_SEGMENT_MB", "1")                      # 80 parse segments: many chunks per context
    outs = {}
    for gpus in ("1", "3"):
        for t in ("1", "5"):
            out = str(tmp_path / ("g%s_t%s.tsv" % (gpus, t)))
            erb.main(["--mode", "tenX_v3", "-i", path, "-o", out, "-t", t, "--gpus", gpus])
            outs[(gpus, t)] = (open(out).read(), open(out + ".stats").read())
    assert outs[("1", "1")] == outs[("3", "1")] and outs[("1", "5")] == outs[("3", "5")]
    assert outs[("3", "1")][0] == "\n".join([header] + rows) + "\n"
    assert outs[("3", "5")][0] == "\n".join([header] + rows) + "\n"       # 40,000 reads: one chunk of the reference's 100,000, one header


@pytest.mark.gpu
def test_stage2_gpus_flag_shares_the_edge_build(tmp_path, golden_dir, monkeypatch):
    """badger.py --gpus N (and -tr N mapped onto it): every context builds its share of the edge list
    (bdg_graph_edges_part_dev, the reference's compare_in_parallel fan-out, barcode_graph.py:164-189), the shares are put
    side by side before the clustering.  Rehearsed on one device with N independent contexts: the output file is that of
    one context and the reference's own, at both thresholds, from the TSV and from the reads; the shares are all non-empty
    on an input large enough to have edges in every part."""
    from badger_amd import stage2
    monkeypatch.setenv("BADGER_AMD_CONTEXTS_ON_ONE_DEVICE", "1")
    want = open(os.path.join(golden_dir, "c1_stage2_output_file.tsv")).read()
    for flags in (["--gpus", "3"], ["-tr", "2"]):
        for reads in ("c1_expected.tsv", "c1_reads.fa.gz"):
            prefix = str(tmp_path / ("mg_%s_%s" % (flags[1], reads[:5])))
            buf = io.StringIO()
            with redirect_stdout(buf):
                badger.main(["-r", os.path.join(golden_dir, reads), "-d", "tenX_v3", "-l", os.path.join(golden_dir, "c1_whitelist.txt"),
                             "-c", "50", "-o", prefix] + flags)
            assert open(prefix + "_output_file.tsv").read() == want, (flags, reads)
            assert buf.getvalue().strip().split("\n")[-1] == open(os.path.join(golden_dir, "c1_stage2_stdout_tail.txt")).read().strip()
    # a larger input, both thresholds: N = 1 against N = 4, and the cut itself
    path, rows, recs = _fastq_of(tmp_path, 30000, 29)
    from badger_amd import synth
    wl = str(tmp_path / "wl.txt")
    with open(wl, "w") as f:
        f.write("\n".join(synth.rank_to_str(r) for r in synth.make_whitelist(2000)) + "\n")
    for thr in ("1", "2"):
        outs, shares = [], []
        orig = stage2.Stage2._build_edges_parts

        def spy(self, *a, _orig=orig, _shares=shares, **k):
            r = _orig(self, *a, **k)
            _shares.append(list(self.edge_shares))
            return r
        stage2.Stage2._build_edges_parts = spy
        try:
            for gpus in ("1", "4"):
                prefix = str(tmp_path / ("mg%s_%s" % (thr, gpus)))
                with redirect_stdout(io.StringIO()):
                    badger.main(["-r", path, "-d", "tenX_v3", "-l", wl, "-c", "300", "-t", thr, "-o", prefix, "--gpus", gpus])
                outs.append(open(prefix + "_output_file.tsv").read())
        finally:
            stage2.Stage2._build_edges_parts = orig
        assert outs[0] == outs[1] and outs[0].count("\n") == 30001
        assert len(shares) == 1 and len(shares[0]) == 4 and min(shares[0]) > 0, shares
