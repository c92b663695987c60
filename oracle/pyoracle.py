"""ctypes front-end of the CPU oracle (oracle/badger_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under badger_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

REC_DTYPE = np.dtype([
    ("polyT", "<i4"), ("r1_end", "<i4"), ("bc_start", "<i4"), ("umi_start", "<i4"),
    ("umi_end", "<i4"), ("bc_rank", "<u4"), ("r1_score", "i1"), ("strand", "i1"),
    ("valid", "u1"), ("flags", "u1"), ("reserved", "<u4")])
EDGE_DTYPE = np.dtype([("a", "<u4"), ("b", "<u4"), ("dist", "<u4")])
assert REC_DTYPE.itemsize == 32 and EDGE_DTYPE.itemsize == 12

FLAG_REV = 1
FLAG_RANK_OK = 2
R1 = "CTACACGACGCTCTTCCGATCT"


def build():
    """Compile liboracle with the committed Makefile (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libbadger_oracle.so"])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libbadger_oracle.so")
    src = os.path.join(_HERE, "badger_oracle.c")
    if os.environ.get("BADGER_ORACLE_LIB"):          # e.g. the ASan/UBSan build (make -C oracle asan)
        path = os.environ["BADGER_ORACLE_LIB"]
    elif not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)):
        build()
    L = C.CDLL(path)
    i32p = C.POINTER(C.c_int32)
    L.orc_find_polyt_start.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.orc_find_polyt_start.restype = C.c_int
    L.orc_revcomp.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
    L.orc_revcomp.restype = C.c_int
    L.orc_kmer_hits.argtypes = [C.c_char_p, C.c_int, i32p, C.c_int]
    L.orc_kmer_hits.restype = C.c_int
    L.orc_sw_align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, i32p]
    L.orc_sw_align.restype = None
    L.orc_detect_exact_positions.argtypes = [C.c_char_p, C.c_int, C.c_int, i32p, C.c_int,
                                             C.c_int, C.c_int, C.c_int, i32p]
    L.orc_detect_exact_positions.restype = C.c_int
    L.orc_extract_read.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    L.orc_extract_read.restype = C.c_int
    L.orc_extract_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
    L.orc_extract_batch.restype = C.c_int64
    L.orc_extract_read_rule.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_extract_read_rule.restype = C.c_int
    L.orc_extract_batch_rule.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_int]
    L.orc_extract_batch_rule.restype = C.c_int64
    L.orc_rank16.argtypes = [C.c_char_p]
    L.orc_rank16.restype = C.c_uint32
    L.orc_unrank16.argtypes = [C.c_uint32, C.c_char_p]
    L.orc_unrank16.restype = None
    L.orc_levenshtein.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.orc_levenshtein.restype = C.c_int
    L.orc_lev16_packed.argtypes = [C.c_uint32, C.c_int, C.c_uint32, C.c_int]
    L.orc_lev16_packed.restype = C.c_int
    L.orc_dmin3.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_dmin3.restype = C.c_int
    L.orc_qgram_S.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_qgram_S.restype = C.c_int
    L.orc_qgram_threshold.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_qgram_threshold.restype = C.c_int
    L.orc_graph_edges.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint64, C.c_int]
    L.orc_graph_edges.restype = C.c_uint64
    L.orc_graph_edges_sampled.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.c_void_p, C.c_uint64,
                                          C.c_int, C.c_void_p]
    L.orc_graph_edges_sampled.restype = C.c_uint64
    L.orc_graph_edges_brute.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint64]
    L.orc_graph_edges_brute.restype = C.c_uint64
    L.orc_nearest16.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_nearest16.restype = None
    L.orc_nearest16_probe.argtypes = L.orc_nearest16.argtypes
    L.orc_nearest16_probe.restype = None
    _LIB = L
    return L


def _b(s):
    return s.encode("ascii") if isinstance(s, str) else bytes(s)


def find_polyt_start(seq, window_size=16, polya_fraction=0.75):
    s = _b(seq)
    return lib().orc_find_polyt_start(s, len(s), window_size, int(window_size * polya_fraction))


def revcomp(seq):
    s = _b(seq)
    out = C.create_string_buffer(len(s) + 1)
    if lib().orc_revcomp(s, len(s), out) != 0:
        raise KeyError("base outside 'ACGTN '")
    return out.raw[:len(s)].decode("ascii")


def kmer_hits(seq):
    s = _b(seq)
    buf = (C.c_int32 * (len(s) + 1))()
    n = lib().orc_kmer_hits(s, len(s), buf, len(s) + 1)
    return list(buf[:n])


def sw_align(pattern, ref):
    """-> (reference_start, reference_end, read_start, read_end, optimal_score)"""
    p, r = _b(pattern), _b(ref)
    out = (C.c_int32 * 5)()
    lib().orc_sw_align(p, len(p), r, len(r), out)
    return tuple(out)


def detect_exact_positions(seq, start, end, hits, min_score=0, start_delta=-1, end_delta=-1):
    s = _b(seq)
    h = (C.c_int32 * max(1, len(hits)))(*hits)
    out = (C.c_int32 * 3)()
    ok = lib().orc_detect_exact_positions(s, start, end, h, len(hits), min_score, start_delta, end_delta, out)
    if not ok:
        return None, None, 0
    return out[0], out[1], out[2]


RULE_DEFAULT, RULE_NO_POLYA = 0, 1          # find_barcode_umi / find_barcode_umi_no_polya


def extract_read(seq, umi_len=12, rule=RULE_DEFAULT):
    s = _b(seq)
    rec = np.zeros(1, dtype=REC_DTYPE)
    if lib().orc_extract_read_rule(s, len(s), umi_len, rule, rec.ctypes.data) != 0:
        raise KeyError("base outside 'ACGTN '")
    return rec[0]


def extract_batch(bases, off, umi_len=12, threads=1, rule=RULE_DEFAULT):
    """bases: uint8 array of concatenated ASCII reads; off: uint64[n+1]."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    out = np.zeros(n, dtype=REC_DTYPE)
    rc = lib().orc_extract_batch_rule(bases.ctypes.data, off.ctypes.data, n, umi_len, rule, out.ctypes.data, threads)
    if rc != 0:
        raise KeyError("read %d holds a base outside 'ACGTN '" % (-rc - 1))
    return out


def rank16(seq):
    return lib().orc_rank16(_b(seq))


def unrank16(rk):
    out = C.create_string_buffer(17)
    lib().orc_unrank16(int(rk), out)
    return out.raw[:16].decode("ascii")


def levenshtein(a, b):
    a, b = _b(a), _b(b)
    return lib().orc_levenshtein(a, len(a), b, len(b))


def lev16_packed(a, la, b, lb):
    return lib().orc_lev16_packed(int(a), la, int(b), lb)


def dmin3(a, b):
    return lib().orc_dmin3(int(a), int(b))


def qgram_S(a, b):
    return lib().orc_qgram_S(int(a), int(b))


def qgram_threshold(threshold, bc_len=16, q=6):
    return lib().orc_qgram_threshold(threshold, bc_len, q)


def graph_edges(ranks, thr, qgram_T=None, threads=1, brute=False):
    ranks = np.ascontiguousarray(ranks, dtype=np.uint32)
    if qgram_T is None:
        qgram_T = qgram_threshold(thr)
    cap = 1 << 16
    while True:
        out = np.zeros(cap, dtype=EDGE_DTYPE)
        if brute:
            tot = lib().orc_graph_edges_brute(ranks.ctypes.data, len(ranks), thr, qgram_T, out.ctypes.data, cap)
        else:
            tot = lib().orc_graph_edges(ranks.ctypes.data, len(ranks), thr, qgram_T, out.ctypes.data, cap, threads)
        if tot <= cap:
            return out[:tot]
        cap = int(tot)


def graph_edges_sampled(ranks, thr, row_stride, qgram_T=None, threads=1, cap=1 << 16):
    """edges of every row_stride-th row of the sorted array against the whole array; returns (edges, seconds for the index,
    seconds in the row loop)"""
    ranks = np.ascontiguousarray(ranks, dtype=np.uint32)
    if qgram_T is None:
        qgram_T = qgram_threshold(thr)
    t = np.zeros(2, dtype=np.float64)
    while True:
        out = np.zeros(cap, dtype=EDGE_DTYPE)
        tot = lib().orc_graph_edges_sampled(ranks.ctypes.data, len(ranks), thr, qgram_T, row_stride, out.ctypes.data, cap,
                                            threads, t.ctypes.data)
        if tot <= cap:
            return out[:tot], float(t[0]), float(t[1])
        cap = int(tot)


def nearest16(q, wl, max_ed=2, threads=1, probe=False):
    """probe=True: neighbourhood enumeration against a hash set (max_ed <= 2), same answers as the exhaustive scan"""
    q = np.ascontiguousarray(q, dtype=np.uint32)
    wl = np.ascontiguousarray(wl, dtype=np.uint32)
    idx = np.zeros(len(q), dtype=np.uint32)
    ed = np.zeros(len(q), dtype=np.uint8)
    ties = np.zeros(len(q), dtype=np.uint16)
    fn = lib().orc_nearest16_probe if probe else lib().orc_nearest16
    fn(q.ctypes.data, len(q), wl.ctypes.data, len(wl), max_ed, idx.ctypes.data, ed.ctypes.data, ties.ctypes.data, threads)
    return idx, ed, ties
