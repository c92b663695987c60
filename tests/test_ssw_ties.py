"""Smith-Waterman coordinate tie rules (SSW: first column whose maximum strictly exceeds the running one, smallest read
index in it; begin = nearest start, found by the reverse pass).  The hand-derived known answers of
tests/golden/ssw_tie_kats.json are checked against (1) the literal definition on the full DP matrix, (2) the C oracle's
streaming implementation, (3) the pure-Python SSW stand-in the golden fixtures were generated with.  The GPU's packed
16-bit implementation meets the same cases through reads built around them (tests/test_hip_parity.py)."""
import importlib.util
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "ssw_tie_kats.json")))["kats"]


def _full_matrix(read, ref):
    m, n = len(read), len(ref)
    H = [[0] * (n + 1) for _ in range(m + 1)]
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            s = 0 if "N" in (read[i - 1], ref[j - 1]) else (1 if read[i - 1] == ref[j - 1] else -1)
            H[i][j] = max(0, H[i - 1][j - 1] + s, H[i - 1][j] - 1, H[i][j - 1] - 1)
    return H


def by_definition(read, ref):
    """the rule applied to the whole matrix: no running maximum, no early termination"""
    H = _full_matrix(read, ref)
    mx = max(max(r) for r in H)
    if mx == 0:
        return [None, None, None, None, 0]
    j = min(j for j in range(1, len(ref) + 1) if max(H[i][j] for i in range(1, len(read) + 1)) == mx)
    i = min(i for i in range(1, len(read) + 1) if H[i][j] == mx)
    re_, rr = j - 1, i - 1
    H2 = _full_matrix(read[:rr + 1][::-1], ref[:re_ + 1][::-1])
    j2 = min(j for j in range(1, re_ + 2) if max(H2[i][j] for i in range(1, rr + 2)) >= mx)
    i2 = min(i for i in range(1, rr + 2) if H2[i][j2] == mx)
    return [re_ - (j2 - 1), re_, rr - (i2 - 1), rr, mx]


def _shim():
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ROOT, "tools", "gen_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)            # defines the stand-ins; touches the reference only in main()
    return mod.AlignmentMgr


@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_hand_derived_answers(kat):
    want = kat["expect"]
    assert by_definition(kat["pattern"], kat["window"]) == want
    got = list(orc.sw_align(kat["pattern"], kat["window"]))
    assert got[4] == want[4]
    if want[4] > 0:
        assert got == want
    mgr = _shim()(match_score=1, mismatch_penalty=1)
    mgr.set_read(kat["pattern"])
    mgr.set_reference(kat["window"])
    a = mgr.align(gap_open=1, gap_extension=1)
    assert a.optimal_score == want[4]
    if want[4] > 0:
        assert [a.reference_start, a.reference_end, a.read_start, a.read_end] == want[:4]


def test_random_windows_oracle_and_shim_follow_the_definition():
    rng = np.random.default_rng(17)
    P = "CTACACGACGCTCTTCCGATCT"
    mgr = _shim()(match_score=1, mismatch_penalty=1)
    for k in range(400):
        # windows made of adapter pieces, repeats and noise: ties are the rule, not the exception
        parts = []
        for _ in range(int(rng.integers(1, 5))):
            a = int(rng.integers(0, 22))
            parts.append(P[a:a + int(rng.integers(1, 12))] if rng.random() < 0.7 else "".join("ACGTN"[i] for i in rng.integers(0, 5, int(rng.integers(1, 6)))))
        w = "".join(parts)[:39]
        want = by_definition(P, w)
        got = list(orc.sw_align(P, w))
        mgr.set_read(P)
        mgr.set_reference(w)
        a = mgr.align(gap_open=1, gap_extension=1)
        assert got[4] == want[4] == a.optimal_score, w
        if want[4] > 0:
            assert got == want, w
            assert [a.reference_start, a.reference_end, a.read_start, a.read_end] == want[:4], w
