#!/usr/bin/env python3
"""reads/s/core of the ACTUAL reference Python path (BarcodeCaller.process_chunk -> TenXBarcodeExtractorV3
.find_barcode_umi per read, extract_raw_barcodes.py:120-128), run in the build container with `ssw` / `editdistance`
replaced by the pure-Python stand-ins of tools/gen_golden.py and, as a second figure, with `ssw` backed by the C
oracle's Smith-Waterman (closer to what ssw-py's C code would cost).  SURVEY 8d asks for this number once, labelled
SW-substituted; it goes to BASELINE.md.  Needs /root/reference (never runs on the GPU box)."""
import importlib.util
import os
import sys
import time
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ROOT, "tools", "gen_golden.py"))
gg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gg)

from badger_amd import synth  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402


class _A:
    pass


class OracleAlignmentMgr:
    """ssw.AlignmentMgr surface over oracle/badger_oracle.c's orc_sw_align (C speed)"""

    def __init__(self, match_score=2, mismatch_penalty=2):
        pass

    def set_read(self, r):
        self.read = r

    def set_reference(self, r):
        self.ref = r

    def align(self, gap_open=3, gap_extension=1):
        rs, re_, qs, qe, sc = orc.sw_align(self.read, self.ref)
        a = _A()
        a.optimal_score, a.reference_start, a.reference_end, a.read_start, a.read_end = sc, rs, re_, qs, qe
        return a


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    wl = synth.make_whitelist(1000)
    bases, off = synth.make_reads(n, wl, seed=1)
    chunk = [("r%d" % i, s) for i, s in enumerate(synth.reads_to_list(bases, off))]
    gg.install_shims()
    from barcode_extraction.barcode_callers import TenXBarcodeExtractorV3        # the reference's own module
    for label, mgr in (("pure-Python SW stand-in", gg.AlignmentMgr), ("C oracle SW behind the ssw interface", OracleAlignmentMgr)):
        sys.modules["ssw"].AlignmentMgr = mgr
        import barcode_extraction.common as refcommon
        refcommon.AlignmentMgr = mgr
        det = TenXBarcodeExtractorV3()
        t0 = time.perf_counter()
        valid = 0
        for rid, seq in chunk:
            r = det.find_barcode_umi(rid, seq)
            valid += r.is_valid()
        dt = time.perf_counter() - t0
        print("%-40s %d reads (mean %d bp) in %.2f s = %.0f reads/s/core, %d barcodes detected" % (label, n, int(off[-1]) // n, dt, n / dt, valid))


if __name__ == "__main__":
    main()
