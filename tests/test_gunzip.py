"""The parallel gzip inflate behind the readers (csrc/pgunzip.cpp): chunks of the compressed stream are inflated from
guessed block starts into 16-bit symbols and accepted only where they continue the accepted data bit for bit.  Against
zlib / Python's gzip on FASTQ text at several levels, chunk sizes and thread counts, on the block kinds a guess cannot start
at (stored, fixed codes), on binary data (no guess succeeds: everything is inflated in sequence), on multi-member files and on
damaged ones (same verdict as zlib); then through the reader itself.  CPU tier."""
import ctypes as C
import gzip
import io
import os
import random
import zlib

import numpy as np
import pytest

from badger_amd import _native


def _lib():
    L = _native.load()
    L.bdg_test_gunzip.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.bdg_test_gunzip.restype = C.c_int
    return L


def gunzip(data, threads=4, chunk=0):
    L = _lib()
    out, n, err = C.c_void_p(), C.c_size_t(), C.create_string_buffer(200)
    rc = L.bdg_test_gunzip(data, len(data), threads, chunk, C.byref(out), C.byref(n), err, 200)
    b = C.string_at(out, n.value) if n.value else b""
    L.bdg_host_free(out)
    return rc, b, err.value.decode()


def _fastq(n, seed):
    rng = np.random.default_rng(seed)
    parts = []
    for i in range(n):
        m = int(rng.integers(50, 400))
        parts.append("@read_%d some text\n%s\n+\n%s\n" % (i, "".join("ACGT"[x] for x in rng.integers(0, 4, m)),
                                                         "".join(chr(33 + x) for x in rng.integers(0, 40, m))))
    return "".join(parts).encode()


@pytest.fixture(scope="module")
def text():
    return _fastq(12000, 1)


def test_levels_chunks_threads(text):
    for level in (1, 6, 9):
        z = gzip.compress(text, level)
        for chunk in (1 << 10, 16 << 10, 256 << 10, 0):
            for th in (1, 3, 8):
                rc, out, err = gunzip(z, th, chunk)
                assert rc == 0 and out == text, (level, chunk, th, err)


def test_block_kinds_members_and_headers(text):
    co = zlib.compressobj(0, zlib.DEFLATED, 31)
    stored = co.compress(text[:300000]) + co.flush()
    rnd = os.urandom(300000)
    buf = io.BytesIO()
    with gzip.GzipFile(filename="name.fq", mode="wb", fileobj=buf, mtime=5) as f:
        f.write(text[:50000])
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for k in range(0, 400000, 7000):                       # sync / full flushes: short blocks with empty stored blocks between
        parts.append(co.compress(text[k:k + 7000]))
        parts.append(co.flush(zlib.Z_SYNC_FLUSH if (k // 7000) % 3 else zlib.Z_FULL_FLUSH))
    parts.append(co.flush())
    run = b"A" * 500000 + text[:200000] + b"\n" * 100000
    cases = {"stored": (stored, text[:300000]), "tiny": (gzip.compress(b"ACGT\n"), b"ACGT\n"), "empty": (gzip.compress(b""), b""),
             "binary": (gzip.compress(rnd, 6), rnd), "zeros": (gzip.compress(bytes(1000000), 6), bytes(1000000)),
             "members": (gzip.compress(text[:100000], 6) + gzip.compress(text[100000:250000], 1) + gzip.compress(b"") +
                         gzip.compress(text[250000:400000], 9), text[:400000]),
             "runs": (gzip.compress(run, 6), run), "name field": (buf.getvalue(), text[:50000]),
             "flushes": (b"".join(parts), text[:400000 + 6000][:len(b"".join(text[k:k + 7000] for k in range(0, 400000, 7000)))])}
    for name, (z, want) in cases.items():
        assert zlib.decompress(z, 31) == want[:len(zlib.decompress(z, 31))]      # (what one member holds)
        for chunk in (1 << 10, 8 << 10, 0):
            for th in (1, 4):
                rc, out, err = gunzip(z, th, chunk)
                assert rc == 0 and out == want, (name, chunk, th, err, len(out), len(want))


def test_damaged_streams_get_zlibs_verdict(text):
    z = gzip.compress(text, 6)
    third = len(z) // 3
    cases = {"cut in the data": (z[:len(z) // 2], "unexpected end of file"), "cut in the trailer": (z[:-3], "unexpected end of file"),
             "cut in the header": (z[:5], "unexpected end of file"), "checksum": (z[:-8] + bytes([z[-8] ^ 1]) + z[-7:], "incorrect data check"),
             "length": (z[:-1] + bytes([z[-1] ^ 1]), "incorrect length check"),
             "a flipped bit": (z[:third] + bytes([z[third] ^ 0x10]) + z[third + 1:], None)}
    for name, (d, msg) in cases.items():
        with pytest.raises(Exception):
            zlib.decompress(d, 31)
        for chunk in (4 << 10, 0):
            rc, out, err = gunzip(d, 4, chunk)
            assert rc != 0 and err.startswith("gzip: ") and (msg is None or msg in err), (name, chunk, rc, err)
            if name.startswith("cut"):
                assert text.startswith(out)                 # what came out before the end is the text


def test_random_slices(text):
    random.seed(3)
    for it in range(40):
        a = random.randrange(0, len(text) - 1)
        b = min(len(text), a + random.randrange(1, 600000))
        z = gzip.compress(text[a:b], random.choice((1, 4, 6, 9)))
        rc, out, err = gunzip(z, random.choice((1, 2, 5)), random.choice((1 << 10, 3 << 10, 64 << 10, 1 << 20)))
        assert rc == 0 and out == text[a:b], (it, a, b, err)


def test_reader_takes_the_parallel_path(tmp_path, monkeypatch, text):
    """the FASTQ reader on a gzip file through PGunzip (forced onto a small file, 4 KiB chunks) gives the records of the
    plain file, whatever the thread count; one thread (the reference's gzip.open) stays on zlib; a cut file is an error"""
    plain = tmp_path / "r.fastq"
    plain.write_bytes(text)
    gz = tmp_path / "r.fastq.gz"
    gz.write_bytes(gzip.compress(text, 6))
    monkeypatch.setenv("BADGER_AMD_GUNZIP_MIN_KB", "0")
    monkeypatch.setenv("BADGER_AMD_GUNZIP_CHUNK_KB", "4")

    def records(path, threads):
        out = []
        ing = _native.Ingest(str(path), chunk_reads=5000, pinned=False, inflate_threads=threads)
        try:
            while True:
                ch = ing.next()
                if ch.n == 0:
                    break
                out += _native.chunk_reads(ch)
                ing.release(ch)
        finally:
            ing.close()
        return out
    ref = records(plain, 1)
    assert len(ref) == 12000
    for threads in (1, 2, 6):
        assert records(gz, threads) == ref, threads
    cut = tmp_path / "cut.fastq.gz"
    cut.write_bytes(gz.read_bytes()[:len(gz.read_bytes()) // 2])
    with pytest.raises(Exception, match="unexpected end of file"):
        records(cut, 4)


def test_reader_on_mixed_members_through_the_parallel_path(tmp_path, monkeypatch, text):
    """several plain gzip members, BGZF blocks in front of and behind them, an empty member and trailing bytes that are no
    member: every plain member goes through PGunzip (forced, 2 KiB chunks) and the records are those of the plain text"""
    from test_ingest import _bgzf
    monkeypatch.setenv("BADGER_AMD_GUNZIP_MIN_KB", "0")
    monkeypatch.setenv("BADGER_AMD_GUNZIP_CHUNK_KB", "2")
    cut = [text.index(b"@read_%d " % k) for k in (2000, 5000, 9000)]
    plain = tmp_path / "m.fastq"
    plain.write_bytes(text)
    mixed = tmp_path / "m.fastq.gz"
    mixed.write_bytes(_bgzf(text[:cut[0]], eof_marker=False) + gzip.compress(text[cut[0]:cut[1]], 6) + gzip.compress(b"") +
                      gzip.compress(text[cut[1]:cut[2]], 1) + _bgzf(text[cut[2]:]) + b"\0\0trailing")

    def records(path, threads):
        out = []
        ing = _native.Ingest(str(path), chunk_reads=3000, pinned=False, inflate_threads=threads, segment_bytes=300000)
        try:
            while True:
                ch = ing.next()
                if ch.n == 0:
                    break
                out += _native.chunk_reads(ch)
                ing.release(ch)
        finally:
            ing.close()
        return out
    want = records(plain, 1)
    assert len(want) == 12000
    for threads in (0, 3):
        assert records(mixed, threads) == want, threads


def test_streams_of_another_compressor(text):
    """libdeflate (when the system has it) cuts blocks differently from zlib, uses static-code blocks for short stretches
    and, at its high levels, near-optimal parsing with long distances: levels 1, 6 and 12 through the parallel inflate"""
    import ctypes.util
    name = ctypes.util.find_library("deflate") or "libdeflate.so.0"
    try:
        ld = C.CDLL(name)
    except OSError:
        pytest.skip("no libdeflate on this system")
    ld.libdeflate_alloc_compressor.restype = C.c_void_p
    ld.libdeflate_alloc_compressor.argtypes = [C.c_int]
    ld.libdeflate_gzip_compress.restype = C.c_size_t
    ld.libdeflate_gzip_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    ld.libdeflate_gzip_compress_bound.restype = C.c_size_t
    ld.libdeflate_gzip_compress_bound.argtypes = [C.c_void_p, C.c_size_t]
    ld.libdeflate_free_compressor.argtypes = [C.c_void_p]
    for level in (1, 6, 12):
        comp = ld.libdeflate_alloc_compressor(level)
        assert comp
        for data in (text, text[:70000], b"ACGT" * 50000 + text[:30000]):
            bound = ld.libdeflate_gzip_compress_bound(comp, len(data))
            buf = C.create_string_buffer(bound)
            n = ld.libdeflate_gzip_compress(comp, data, len(data), buf, bound)
            assert n > 0
            z = buf.raw[:n]
            assert zlib.decompress(z, 31) == data
            for chunk in (2 << 10, 64 << 10, 0):
                for th in (1, 4):
                    rc, out, err = gunzip(z, th, chunk)
                    assert rc == 0 and out == data, (level, len(data), chunk, th, err)
        ld.libdeflate_free_compressor(comp)


def test_a_chunks_output_is_bounded(text):
    """A speculative chunk stops when it has produced more symbols than its bound (a very compressible stream would
    otherwise hold gigabytes in flight); the chain takes what it has and goes on block by block.  Forced here with a bound
    of 64 K symbols on text that inflates 1000 : 1 and on FASTQ: the bytes, the CRC and the length stay right.  (The bound
    is read once per process: this runs in a child.)"""
    import subprocess
    import sys
    code = r"""
import gzip, sys, zlib
sys.path.insert(0, %r)
from tests.test_gunzip import gunzip, _fastq
rep = (b"ACGTACGTAAAACCCC" * 64 + b"\n") * 40000            # 41 MB, compresses ~1000 : 1
for data in (rep, _fastq(12000, 1)):
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    z = b"".join(co.compress(data[i:i + (1 << 20)]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(data), 1 << 20)) + co.flush()
    for chunk in (4 << 10, 64 << 10):
        rc, out, err = gunzip(z, 4, chunk)
        assert rc == 0 and out == data, (len(data), chunk, err)
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BADGER_AMD_GUNZIP_MAX_CHUNK_KSYM="64")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
