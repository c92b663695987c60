"""The reader's parallel path (csrc/ingest.cpp): segments parsed by several threads must give exactly the records - and,
on malformed input, exactly the error - of the one-thread parse, whatever the segment size and wherever a boundary
falls; SAM / BAM decoding against files written by tests/bamio.py.  No GPU needed (pageable buffers)."""
import gzip
import os

import numpy as np
import pytest

from badger_amd import _native
from badger_amd import extract_raw_barcodes as erb

import bamio


def _read_all(path, size=100000, **kw):
    ing = _native.Ingest(str(path), size, 4, pinned=False, **kw)
    out, sizes = [], []
    try:
        while True:
            ch = ing.next()
            if ch.n == 0:
                break
            out += _native.chunk_reads(ch)
            sizes.append(ch.n)
            ing.release(ch)
        assert ing.reads() == len(out)
    finally:
        ing.close()
    return out, sizes


def _fastq_text(n, seed, blank_every=0, crlf=False, at_quals=True):
    rng = np.random.default_rng(seed)
    parts = []
    for i in range(n):
        L = int(rng.integers(0, 900))
        s = "".join("ACGTN"[k] for k in rng.integers(0, 5, L))
        q = ("@" * L) if (at_quals and i % 4 == 0) else ("+" * L if i % 7 == 0 else "I" * L)     # quality lines that look like headers
        nl = "\r\n" if crlf else "\n"
        parts.append("@r%d some words%s%s%s+%s%s%s%s" % (i, nl, s, nl, ("r%d some words" % i) if i % 3 == 0 else "", nl, q, nl))
        if blank_every and i % blank_every == 0:
            parts.append(nl)
    return "".join(parts)


@pytest.mark.parametrize("kind", ["plain", "gz", "bgzf"])
def test_fastq_segments_equal_the_sequential_parse(tmp_path, kind):
    text = _fastq_text(4000, 5, blank_every=37)
    name = {"plain": "r.fastq", "gz": "r.fastq.gz", "bgzf": "rb.fastq.gz"}[kind]
    p = tmp_path / name
    raw = text.encode()
    p.write_bytes(raw if kind == "plain" else gzip.compress(raw, 1) if kind == "gz" else bamio.bgzf(raw, block=4000, level=1))
    want = list(erb.open_reads(str(p)))
    assert len(want) == 4000
    for seg in (0, 1 << 20, 300000, 65536, 20011, 1500, 64):
        for threads in (0, 1, 4):
            got, sizes = _read_all(p, 512, inflate_threads=threads, segment_bytes=seg)
            assert got == want, (seg, threads)
            assert max(sizes) <= 512
    # one segment: the reference's exact chunking
    assert _read_all(p, 512)[1] == [512] * 7 + [416]


@pytest.mark.parametrize("crlf", [False, True])
def test_fasta_and_crlf_segments(tmp_path, crlf):
    rng = np.random.default_rng(6)
    nl = "\r\n" if crlf else "\n"
    text = "junk in front" + nl
    for i in range(1500):
        s = "".join("ACGT"[k] for k in rng.integers(0, 4, int(rng.integers(0, 700))))
        text += ">q%d desc%s" % (i, nl)
        w = int(rng.integers(5, 120))
        text += "".join(s[a:a + w] + nl for a in range(0, len(s), w))
    p = tmp_path / "r.fa"
    with open(p, "w", newline="") as f:
        f.write(text)
    want = list(erb.open_reads(str(p)))
    assert len(want) == 1500
    for seg in (0, 100000, 7001, 900, 100):
        assert _read_all(p, 1000, segment_bytes=seg, inflate_threads=3)[0] == want, seg
    q = tmp_path / "r.fastq"
    with open(q, "w", newline="") as f:
        f.write(_fastq_text(800, 7, crlf=crlf))
    want = list(erb.open_reads(str(q)))
    for seg in (0, 50000, 3000):
        assert _read_all(q, 1000, segment_bytes=seg, inflate_threads=3)[0] == want, seg


def test_malformed_input_fails_like_the_sequential_parse(tmp_path):
    """wherever the bad record lies relative to the segment boundaries: same message (with the line number), and the same
    reads delivered in front of it, as the one-thread, one-segment parse"""
    text = _fastq_text(1200, 8, at_quals=False)
    cut = text.index("@r900 ")
    for bad in ("r900\nACGT\n+\nIIII\n", "@x\nACGT\nIIII\n@y\nAC\n+\nII\n", "@x\nACGT\n+\nIII\n", "@x\nACGT\n"):
        p = tmp_path / "bad.fastq"
        p.write_text(text[:cut] + bad + text[cut:cut + 5000] if not bad.endswith("ACGT\n") else text[:cut] + bad)
        outcomes = set()
        for seg, threads in ((0, 1), (0, 4), (100000, 4), (30000, 3), (2048, 2)):
            ing = _native.Ingest(str(p), 100, 4, pinned=False, inflate_threads=threads, segment_bytes=seg)
            n = 0
            with pytest.raises(ValueError) as e:
                while True:
                    ch = ing.next()
                    assert ch.n
                    n += ch.n
                    ing.release(ch)
            ing.close()
            outcomes.add((str(e.value), n))
        assert len(outcomes) == 1 and list(outcomes)[0][1] == 900, outcomes
    # the reference's readers (Bio / gzip) fail on these as well
    with pytest.raises(ValueError):
        list(erb.open_reads(str(p)))


def test_truncated_gzip_is_an_error(tmp_path):
    """gzip.open in the reference raises EOFError on a stream that ends early; a file cut at a record boundary must not
    give a silently short TSV"""
    text = _fastq_text(500, 9)
    raw = gzip.compress(text.encode(), 6)
    p = tmp_path / "t.fastq.gz"
    for cut in (6, 20, len(raw) // 2):
        p.write_bytes(raw[:len(raw) - cut])
        for threads in (0, 1):
            with pytest.raises((ValueError, _native.BadgerHipError)) as e:
                _read_all(p, 100, inflate_threads=threads)
            assert "unexpected end of file" in str(e.value)
        with pytest.raises(EOFError):
            list(erb.open_reads(str(p)))
    p.write_bytes(raw)
    assert len(_read_all(p)[0]) == 500


def _records(n, seed):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        L = int(rng.integers(1, 600))
        alphabet = "ACGT" if i % 11 else "ACGTNMRWSYKVHDB="          # IUPAC codes survive the decoder (the GPU rejects them later)
        seq = "".join(alphabet[k] for k in rng.integers(0, len(alphabet), L))
        flag = [4, 0, 16, 256, 2048, 272, 4, 4][i % 8]               # unmapped, forward, reverse, secondary, supplementary, ...
        cigar = [] if flag & 4 else [(L, "M")] if i % 3 else [(3, "S"), (L - 3, "M")] if L > 3 else [(L, "M")]
        tags = b"" if i % 2 else b"NMC\x00RGZgrp1\x00"
        recs.append(("read/%d" % i, flag, seq, cigar, tags))
    return recs


@pytest.mark.parametrize("container", ["bam", "bam_tiny_blocks", "bam_uncompressed", "sam", "sam.gz"])
def test_sam_and_bam_records(tmp_path, container):
    recs = _records(700, 10)
    if container.startswith("bam"):
        raw = bamio.bam_raw(recs)
        p = tmp_path / "r.bam"
        p.write_bytes(raw if container == "bam_uncompressed" else bamio.bgzf(raw, block=911 if "tiny" in container else 65280))
    else:
        p = tmp_path / ("r." + container)
        data = bamio.sam_text(recs).encode()
        p.write_bytes(gzip.compress(data) if container.endswith("gz") else data)
    everything = [(r[0], r[2]) for r in recs]
    primary = [(r[0], r[2]) for r in recs if not r[1] & 0x900]
    assert len(primary) < len(everything)
    for seg in (0, 20000, 777):
        for threads in (0, 1, 3):
            assert _read_all(p, 64, inflate_threads=threads, segment_bytes=seg)[0] == everything        # reference :110-118 keeps every record
            assert _read_all(p, 64, inflate_threads=threads, segment_bytes=seg, skip_secondary=True)[0] == primary   # :144-145
    assert list(erb.open_reads(str(p), skip_secondary=True)) == primary


def test_sam_text_is_normalised_like_htslib(tmp_path):
    """pysam hands out what htslib stored: SEQ through the 4-bit code table (lower case -> upper, unknown letters -> N)"""
    p = tmp_path / "n.sam"
    p.write_text("@HD\tVN:1.6\nq1\t4\t*\t0\t0\t*\t*\t0\t0\tacgtnACGTNxXuU.=\t*\n")
    assert _read_all(p)[0] == [("q1", "ACGTNACGTNNNNNN=")]


def test_records_without_a_sequence_and_damaged_bam(tmp_path):
    recs = _records(50, 11)
    p = tmp_path / "r.bam"
    p.write_bytes(bamio.bgzf(bamio.bam_raw(recs[:20] + [("empty", 4, "")] + recs[20:])))
    with pytest.raises(TypeError):                       # the reference: find_barcode_umi(read_id, None) -> len(None)
        _read_all(p)
    s = tmp_path / "r.sam"
    s.write_text(bamio.sam_text(recs[:5] + [("empty", 4, "")]))
    with pytest.raises(TypeError):
        _read_all(s)
    raw = bamio.bam_raw(recs)
    p.write_bytes(bamio.bgzf(raw[:len(raw) - 7]))        # cut inside the last record
    with pytest.raises(ValueError):
        _read_all(p)
    p.write_bytes(bamio.bgzf(b"BAX\x01" + raw[4:]))
    with pytest.raises(ValueError):
        _read_all(p)
    s.write_text("q1\t4\t*\t0\t0\t*\n")                  # fewer than 11 fields
    with pytest.raises(ValueError):
        _read_all(s)
