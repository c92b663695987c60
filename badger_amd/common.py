"""rank / unrank of barcodes (reference common.py:11-38): little-endian base-4, A0 C1 G2 T3,
first base in the two least-significant bits.  Vectorised forms for whole lists."""
import numpy as np

RANK = {"A": 0, "C": 1, "G": 2, "T": 3}
UNRANK = "ACGT"
_LUT = np.full(256, 255, dtype=np.uint8)
for _c, _v in RANK.items():
    _LUT[ord(_c)] = _v


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
    w = (np.uint64(1) << (2 * np.arange(length, dtype=np.uint64)))
    return (codes.astype(np.uint64) * w).sum(axis=1, dtype=np.uint64)
