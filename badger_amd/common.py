"""rank / unrank of barcodes (reference common.py:11-38): little-endian base-4, A0 C1 G2 T3,
first base in the two least-significant bits.  Vectorised forms for whole lists."""
import numpy as np

RANK = {"A": 0, "C": 1, "G": 2, "T": 3}
UNRANK = "ACGT"
_LUT = np.full(256, 255, dtype=np.uint8)
for _c, _v in RANK.items():
    _LUT[ord(_c)] = _v


def _pack(codes, length):
    """rows of 2-bit codes -> ranks (uint64); an OR-reduction of shifted columns (numpy's row sums of many short rows
    are an order of magnitude slower)"""
    dt = np.uint32 if length <= 16 else np.uint64
    sh = (2 * np.arange(length)).astype(dt)
    return np.bitwise_or.reduce(codes.astype(dt) << sh, axis=1).astype(np.uint64)


def rank(seq, length):
    rk = 0
    for i in range(length):
        rk += RANK[seq[i]] << (2 * i)
    return rk


def unrank(rk, length):
    return "".join(UNRANK[(rk >> (2 * i)) & 3] for i in range(length))


def rank_many(seqs, length=16):
    """list[str] (all of len `length`) -> uint64 array of ranks; KeyError on a non-ACGT base."""
    if not seqs:
        return np.zeros(0, dtype=np.uint64)
    raw = np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8).reshape(len(seqs), length)
    codes = _LUT[raw]
    if (codes == 255).any():
        bad = seqs[int(np.nonzero((codes == 255).any(axis=1))[0][0])]
        raise KeyError("barcode %r holds a base outside ACGT" % bad)
    return _pack(codes, length)


def rank_valid_many(seqs, length=16):
    """Ranks of those strings of an iterable that are `length` bases of ACGT (the only ones an unrank() output can equal,
    reference barcode_graph.py:264 `unrank(...) in barcode_list`); uint64 array, order of the input."""
    seqs = [s for s in seqs if len(s) == length]
    if not seqs:
        return np.zeros(0, dtype=np.uint64)
    try:
        raw = np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8)
    except UnicodeEncodeError:
        seqs = [s for s in seqs if s.isascii()]
        raw = np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8)
    codes = _LUT[raw.reshape(-1, length)]
    codes = codes[(codes != 255).all(axis=1)]
    return _pack(codes, length)


class BarcodeRanks:
    """The --barcode_list file (reference badger.py:82-88: a set of the file's lines) as the ranks of those lines that are
    `length` letters of ACGT - the only lines an unrank() output can equal (barcode_graph.py:264).  Truthy whatever it
    holds, like the reference's set, which always contains at least the empty string."""

    def __init__(self, ranks):
        self.ranks = np.asarray(ranks, dtype=np.uint32)

    def __bool__(self):
        return True

    def __len__(self):
        return len(self.ranks)

    @classmethod
    def from_file(cls, path, length=16):
        buf = np.fromfile(path, dtype=np.uint8)
        if (buf == 13).any():                                     # '\r': Python's text mode translates it; take the plain route
            with open(path) as f:
                return cls(rank_valid_many(set(f.read().split("\n")), length))
        nl = np.flatnonzero(buf == 10)
        starts = np.concatenate([np.zeros(1, dtype=np.int64), nl + 1])
        ends = np.concatenate([nl, np.array([len(buf)], dtype=np.int64)])
        starts = starts[ends - starts == length]
        if not len(starts):
            return cls(np.zeros(0, dtype=np.uint32))
        codes = _LUT[buf[starts[:, None] + np.arange(length)]]
        return cls(np.unique(_pack(codes[(codes != 255).all(axis=1)], length)))
