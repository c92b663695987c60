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
