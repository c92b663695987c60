"""Native read ingest and row formatter (include/badger_hip.h bdg_ingest_*, bdg_format_rows) against the Python
readers that restate Bio.SeqIO's record semantics and against the Python row formatter; no GPU needed (pageable
buffers, records from the CPU oracle)."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from badger_amd import _native, synth
from badger_amd import extract_raw_barcodes as erb
from badger_amd.barcode_extraction.barcode_callers import record_to_row
from oracle import pyoracle as orc


def _chunks(path, size, ring=3, **kw):
    ing = _native.Ingest(path, size, ring, pinned=False, **kw)
    out = []
    try:
        while True:
            ch = ing.next()
            if ch.n == 0:
                break
            out.append(_native.chunk_reads(ch))
            ing.release(ch)
    finally:
        ing.close()
    return out


def _write(path, text):
    if str(path).endswith((".gz", ".gzip")):
        with gzip.open(path, "wt", newline="") as f:
            f.write(text)
    else:
        with open(path, "w", newline="") as f:
            f.write(text)


@pytest.mark.parametrize("name", ["r.fa", "r.fasta.gz", "r.FA", "r.fa.gzip"])
def test_fasta_records_and_chunks(tmp_path, name):
    rng = np.random.default_rng(1)
    rnd = lambda k: "".join("ACGT"[i] for i in rng.integers(0, 4, k))
    text = "; junk before the first header is ignored\nACGT\n"
    for i in range(257):
        s = rnd(int(rng.integers(0, 400)))
        text += ">read_%d some description\tmore\n" % i
        w = int(rng.integers(1, 90))
        for a in range(0, len(s), w):                       # multi-line sequences, some with trailing blanks or CRLF
            text += s[a:a + w] + ("  \r\n" if (i + a) % 7 == 0 else "\n")
        if i % 50 == 0:
            text += "\n"
    text += ">\n\n>last_no_newline\nACGTN"
    p = tmp_path / name
    _write(p, text)
    want = list(erb.open_reads(str(p)))
    assert len(want) == 259 and want[-2] == ("", "") and want[-1] == ("last_no_newline", "ACGTN")
    for size in (1, 7, 100, 259, 1000):
        got = _chunks(str(p), size)
        assert [len(c) for c in got] == [len(c) for c in erb.read_chunks(iter(want), size) if c]
        assert [r for c in got for r in c] == want


@pytest.mark.parametrize("name", ["r.fq", "r.fastq.gz"])
def test_fastq_records_and_errors(tmp_path, name):
    rng = np.random.default_rng(2)
    rnd = lambda k: "".join("ACGTN"[i] for i in rng.integers(0, 5, k))
    text = ""
    for i in range(300):
        s = rnd(int(rng.integers(0, 3000)))
        text += "@r%d extra words\n%s\n+%s\n%s\n" % (i, s, "r%d" % i if i % 3 == 0 else "", "I" * len(s))
        if i % 40 == 0:
            text += "\n"                                     # blank line between records
    p = tmp_path / name
    _write(p, text)
    want = list(erb.open_reads(str(p)))
    assert len(want) == 300
    got = _chunks(str(p), 64)
    assert [len(c) for c in got] == [64, 64, 64, 64, 44] and [r for c in got for r in c] == want
    # a long line crossing the reader's 4 MB blocks
    big = "@big\n%s\n+\n%s\n" % ("A" * 9000001, "I" * 9000001)
    _write(p, big + text)
    got = [r for c in _chunks(str(p), 1000) for r in c]              # (18 MB: the record spans two parse segments)
    assert got[0][0] == "big" and len(got[0][1]) == 9000001 and got[1:] == want
    for bad in ("r1\nACGT\n+\nIIII\n", "@r1\nACGT\nIIII\n@r2\n", "@r1\nACGT\n+\nIII\n", "@r1\nACGT\n"):
        _write(p, text[:text.index("@r7 ")] + bad)
        with pytest.raises(ValueError):
            _chunks(str(p), 4)
        with pytest.raises(ValueError):
            list(erb.open_reads(str(p)))
    # the reads in front of the malformed record are still delivered
    _write(p, text[:text.index("@r9 ")] + "oops\n")
    ing = _native.Ingest(str(p), 4, 3, pinned=False)
    assert ing.next().n == 4 and ing.next().n == 4 and ing.next().n == 1
    with pytest.raises(ValueError):
        ing.next()
    ing.close()


def _bgzf(data, block=65280, eof_marker=True, level=6):
    """data as a BGZF file (SAM specification 4.1): gzip members of at most 64 KiB, each with the 'BC' extra field that
    states the member's size; what bgzip writes."""
    import struct
    import zlib
    out = bytearray()
    pieces = [data[a:a + block] for a in range(0, len(data), block)] + ([b""] if eof_marker else [])
    for piece in pieces:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(piece) + co.flush()
        total = 18 + len(body) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, total - 1)
        out += body + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece))
    return bytes(out)


def _flat(chunks):
    return [r for c in chunks for r in c]


@pytest.mark.parametrize("threads", [0, 1, 2, 5])
def test_bgzf_input_is_inflated_in_parallel_and_in_order(tmp_path, threads):
    rng = np.random.default_rng(3)
    rnd = lambda k: "".join("ACGTN"[i] for i in rng.integers(0, 5, k))
    text = "".join("@r%d w\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in ((i, rnd(int(rng.integers(0, 4000)))) for i in range(1500)))
    raw = text.encode()
    plain = tmp_path / "plain.fastq"
    plain.write_bytes(raw)
    want = _flat(_chunks(str(plain), 256))
    assert len(want) == 1500

    def chunks_of(path, size=256):
        return _flat(_chunks(str(path), size, inflate_threads=threads))

    p = tmp_path / "r.fastq.gz"
    for block, marker in ((65280, True), (1000, True), (65280, False)):
        p.write_bytes(_bgzf(raw, block, marker))
        assert chunks_of(p) == want
    p.write_bytes(_bgzf(raw[:raw.index(b"@r100 ")], 17))                    # 12,000 tiny blocks
    assert chunks_of(p) == want[:100]
    assert list(erb.open_reads(str(p)))[:20] == want[:20]                 # Python's gzip reads the same file as multi-member gzip
    # a plain gzip member behind BGZF blocks, BGZF behind plain gzip, and trailing bytes that are no gzip member
    half = raw.index(b"@r700 ")
    p.write_bytes(_bgzf(raw[:half], eof_marker=False) + gzip.compress(raw[half:]))
    assert chunks_of(p) == want
    p.write_bytes(_bgzf(raw[:half], eof_marker=False) + gzip.compress(raw[half:half + 100000]) + gzip.compress(raw[half + 100000:]))
    assert chunks_of(p) == want
    p.write_bytes(gzip.compress(raw[:half]) + _bgzf(raw[half:]))
    assert chunks_of(p) == want
    p.write_bytes(_bgzf(raw) + b"\0" * 37)
    assert chunks_of(p) == want
    # an empty BGZF file (the end marker alone) and the same name without the extension's help
    p.write_bytes(_bgzf(b""))
    assert chunks_of(p) == []
    q = tmp_path / "looks_plain.fastq"
    q.write_bytes(_bgzf(raw))
    assert chunks_of(q) == want
    # damage: a flipped payload byte (checksum), a truncated last block, a block that claims an impossible size
    good = bytearray(_bgzf(raw))
    bad = bytearray(good); bad[len(bad) // 2] ^= 0x55
    p.write_bytes(bytes(bad))
    with pytest.raises((ValueError, _native.BadgerHipError)):
        chunks_of(p)
    p.write_bytes(bytes(good[:len(good) - 40]))
    with pytest.raises((ValueError, _native.BadgerHipError)):
        chunks_of(p)
    bad = bytearray(good); bad[16:18] = b"\x05\x00"
    p.write_bytes(bytes(bad))
    if threads == 1:
        assert chunks_of(p) == want              # zlib's sequential reader never looks at the size field
    else:
        with pytest.raises((ValueError, _native.BadgerHipError)):
            chunks_of(p)


def test_unknown_extension_is_refused(tmp_path):
    p = tmp_path / "reads.txt"
    p.write_bytes(b"@r\nACGT\n+\nIIII\n")
    with pytest.raises(_native.BadgerHipError):
        _native.Ingest(str(p), 10, 2, pinned=False)
    assert not erb.is_fastx(str(p)) and erb.is_fastx("x.FASTQ.gz") and erb.is_fastx("a/b.fa")


def test_format_rows_equals_python_formatter(tmp_path, golden_dir):
    wl = synth.make_whitelist(500)
    bases, off = synth.make_reads(700, wl, seed=5)
    seqs = synth.reads_to_list(bases, off)
    seqs += ["", "ACGT", "T" * 40, "CTACACGACGCTCTTCCGATCT" + "ACGTACGTACGTACGT" + "AC"]        # short tails: slices clip like Python's
    p = tmp_path / "x.fastq"
    _write(p, "".join("@id%d/x y\n%s\n+\n%s\n" % (i, s, "#" * len(s)) for i, s in enumerate(seqs)))
    ing = _native.Ingest(str(p), 5000, 2, pinned=False)
    ch = ing.next()
    assert ch.n == len(seqs)
    b, o = synth.list_to_reads(seqs)
    recs = orc.extract_batch(b, o, 12, threads=4).view(_native.REC_DTYPE)
    assert (recs["flags"] & 1).sum() > 100 and (recs["valid"] == 0).sum() > 0
    rows, counts = _native.format_rows(ch, recs)
    want = "".join(record_to_row("id%d/x" % i, s, r) + "\n" for i, (s, r) in enumerate(zip(seqs, recs)))
    assert rows.decode() == want
    assert counts == (len(seqs), int(recs["valid"].sum()), int((recs["polyT"] != -1).sum()), int((recs["r1_end"] != -1).sum()))
    ing.release(ch)
    assert ing.next().n == 0
    ing.close()
