"""End-to-end CLI runs on the GPU: stage 1 TSV/.stats and stage 2 output byte-identical to the
fixtures the reference produced (tests/golden/)."""
import io
import os
from contextlib import redirect_stdout

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
        erb.run_fastx_pipeline(path, dets, on_chunk, chunk_size=size)
        assert b"".join(got).decode() == "\n".join(rows) + "\n"
        full, rest = divmod(9000, size)
        assert sizes == [size] * full + ([rest] if rest else [0])


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
        name = "c1_stage2%s" % ("_hs" if hs else "")
        assert buf.getvalue().strip().split("\n")[-1] == open(os.path.join(golden_dir, name + "_stdout_tail.txt")).read().strip()
        got = open(prefix + "_output_file.tsv").read().split("\n")
        want = open(os.path.join(golden_dir, name + "_output_file.tsv")).read().split("\n")
        if not hs:
            assert got == want
        else:
            # high-sensitivity ties are hash-seed dependent in the reference (set iteration order);
            # everything else must agree, and an unassigned read stays unassigned in both
            assert len(got) == len(want)
            diff = [(g, w) for g, w in zip(got, want) if g != w]
            assert all(g.split("\t")[1] != "*" and w.split("\t")[1] != "*" for g, w in diff)
            assert len(diff) <= 0.02 * len(got)


def test_stage2_from_fastx_input(tmp_path, golden_dir):
    prefix = str(tmp_path / "s2fx")
    buf = io.StringIO()
    with redirect_stdout(buf):
        badger.main(["-r", os.path.join(golden_dir, "c1_reads.fa.gz"), "-d", "tenX_v3",
                     "-l", os.path.join(golden_dir, "c1_whitelist.txt"), "-c", "50", "-o", prefix])
    assert open(prefix + "_output_file.tsv").read() == open(os.path.join(golden_dir, "c1_stage2_output_file.tsv")).read()
