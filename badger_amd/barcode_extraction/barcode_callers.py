"""Host-side mirror of the reference's extractor interface (barcode_callers.py), backed by
the HIP library through the C ABI.  Same class and method names, same row text:

    TenXBarcodeExtractorV3().find_barcode_umi(read_id, sequence) -> TenXBarcodeDetectionResult

plus a batch form, find_barcode_umi_batch(), which is what the chunk loop uses (one device
call per chunk of reads instead of one Python call per read).  All arithmetic happens on
the GPU; this module only slices strings and formats rows.
"""
from collections import defaultdict
from dataclasses import dataclass
from enum import Enum, unique

import numpy as np

from .. import _native

NOSEQ = "*"
_COMP = str.maketrans("ACGTN ", "TGCAN ")
_STRAND = {1: "+", -1: "-", 0: "."}


def reverese_complement(seq):
    """Name kept from the reference (barcode_extraction/common.py:37)."""
    for ch in seq:
        if ch not in "ACGTN ":
            raise KeyError(ch)
    return seq.translate(_COMP)[::-1]


@dataclass
class BarcodeDetectionResult:
    read_id: str
    barcode: str = NOSEQ
    UMI: str = NOSEQ
    BC_score: int = -1
    UMI_good: bool = False
    strand: str = "."
    NOSEQ = NOSEQ

    def is_valid(self):
        return self.barcode != NOSEQ

    def set_strand(self, strand):
        self.strand = strand

    def get_additional_attributes(self):
        raise NotImplementedError()

    def _base_fields(self):
        return "%s\t%s\t%s\t%d\t%s\t%s" % (self.read_id, self.barcode, self.UMI, self.BC_score, self.UMI_good, self.strand)

    def __str__(self):
        return self._base_fields()

    @staticmethod
    def header():
        return "#read_id\tbarcode\tUMI\tBC_score\tvalid_UMI\tstrand"


@dataclass
class TenXBarcodeDetectionResult(BarcodeDetectionResult):
    polyT: int = -1
    r1: int = -1
    r1_score: int = 0

    def more_informative_than(self, that):
        return self.r1_score > that.r1_score

    def get_additional_attributes(self):
        attr = []
        if self.polyT != -1:
            attr.append("PolyT detected")
        if self.r1 != -1:
            attr.append("R1 detected")
        return attr

    def __str__(self):
        return self._base_fields() + "\t%d\t%d" % (self.polyT, self.r1)

    @staticmethod
    def header():
        return BarcodeDetectionResult.header() + "\tpolyT_start\tR1_end"


class ReadStats:
    """Counters of barcode_callers.py:122-143; also accumulates straight from device records."""

    def __init__(self):
        self.read_count = 0
        self.bc_count = 0
        self.umi_count = 0
        self.additional_attributes_counts = defaultdict(int)

    def add_read(self, res):
        self.read_count += 1
        for a in res.get_additional_attributes():
            self.additional_attributes_counts[a] += 1
        if res.barcode != NOSEQ:
            self.bc_count += 1
        if res.UMI_good:
            self.umi_count += 1

    def add_records(self, recs):
        self.read_count += len(recs)
        self.bc_count += int(recs["valid"].sum())
        # insertion order of the attribute keys follows the first read that shows them
        for rec_polyt, rec_r1 in zip(recs["polyT"], recs["r1_end"]):
            if len(self.additional_attributes_counts) == 2:
                break
            if rec_polyt != -1:
                self.additional_attributes_counts["PolyT detected"] += 0
            if rec_r1 != -1:
                self.additional_attributes_counts["R1 detected"] += 0
        npt, nr1 = int((recs["polyT"] != -1).sum()), int((recs["r1_end"] != -1).sum())
        if npt:
            self.additional_attributes_counts["PolyT detected"] += npt
        if nr1:
            self.additional_attributes_counts["R1 detected"] += nr1

    def __str__(self):
        s = "Total reads:\t%d\nBarcode detected:\t%d\nReliable UMI:\t%d\n" % (self.read_count, self.bc_count, self.umi_count)
        for a, v in self.additional_attributes_counts.items():
            s += "%s:\t%d\n" % (a, v)
        return s


@unique
class TenXVersions(Enum):
    v2 = 2
    v3 = 3


def record_to_result(read_id, sequence, rec):
    """One device record (include/badger_hip.h bdg_extract_rec) -> result object."""
    if not rec["valid"]:
        return TenXBarcodeDetectionResult(read_id, strand=_STRAND[int(rec["strand"])], polyT=int(rec["polyT"]))
    s = sequence.translate(_COMP)[::-1] if rec["flags"] & _native.FLAG_REV else sequence
    b0 = int(rec["bc_start"])
    return TenXBarcodeDetectionResult(read_id, s[b0:b0 + 16], s[int(rec["umi_start"]):int(rec["umi_end"])], BC_score=0,
                                      strand=_STRAND[int(rec["strand"])], polyT=int(rec["polyT"]),
                                      r1=int(rec["r1_end"]), r1_score=int(rec["r1_score"]))


def record_to_row(read_id, sequence, rec):
    """Row text without building a result object (hot loop of the TSV writer)."""
    if not rec["valid"]:
        return "%s\t*\t*\t-1\tFalse\t%s\t%d\t-1" % (read_id, _STRAND[int(rec["strand"])], rec["polyT"])
    s = sequence.translate(_COMP)[::-1] if rec["flags"] & _native.FLAG_REV else sequence
    b0 = int(rec["bc_start"])
    return "%s\t%s\t%s\t0\tFalse\t%s\t%d\t%d" % (read_id, s[b0:b0 + 16], s[int(rec["umi_start"]):int(rec["umi_end"])],
                                                  _STRAND[int(rec["strand"])], rec["polyT"], rec["r1_end"])


class TenXBarcodeExtractor:
    TSO = "CCCATGTACTCTGCGTTGATACCACTGCTT"
    R1 = "CTACACGACGCTCTTCCGATCT"
    BARCODE_LEN_10X = 16
    UMI_LENGTHS = {TenXVersions.v2: 10, TenXVersions.v3: 12}
    TERMINAL_MATCH_DELTA = 4
    STRICT_TERMINAL_MATCH_DELTA = 1

    def __init__(self, protocol_version=TenXVersions.v3, device=0, instance=0):
        self.UMI_LEN_10X = self.UMI_LENGTHS[protocol_version]
        self.device = device
        self.instance = instance          # > 0: a further independent context on the same device

    # the detector is pickled to workers in the reference; keep it stateless and cheap
    def __getstate__(self):
        return {"UMI_LEN_10X": self.UMI_LEN_10X, "device": self.device, "instance": self.instance}

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.__dict__.setdefault("instance", 0)

    def _ctx(self):
        return _native.default_context(self.device, self.instance)

    def extract_records(self, sequences, strand_rule=_native.STRAND_RULE_DEFAULT):
        """list[str] -> structured array of bdg_extract_rec (one device call)."""
        n = len(sequences)
        off = np.zeros(n + 1, dtype=np.uint64)
        if n:
            off[1:] = np.cumsum([len(s) for s in sequences], dtype=np.uint64)
        try:
            bases = np.frombuffer("".join(sequences).encode("ascii"), dtype=np.uint8)
        except UnicodeEncodeError as e:
            raise KeyError(str(e))
        ctx = self._ctx()
        try:
            ctx.extract_set_strand_rule(strand_rule)
            return ctx.extract_batch(bases, off, self.UMI_LEN_10X)
        except _native.BadgerHipError as e:
            if e.code == _native.E_BADBASE:
                raise KeyError(str(e))      # the reference raises KeyError in reverese_complement
            raise
        finally:
            ctx.extract_set_strand_rule(_native.STRAND_RULE_DEFAULT)     # the context is shared: leave it as found

    def find_barcode_umi_batch(self, read_chunk):
        """read_chunk: list[(read_id, seq)] -> list[TenXBarcodeDetectionResult], input order."""
        recs = self.extract_records([s for _, s in read_chunk])
        return [record_to_result(rid, s, r) for (rid, s), r in zip(read_chunk, recs)]

    def find_barcode_umi(self, read_id, sequence):
        return self.find_barcode_umi_batch([(read_id, sequence)])[0]

    def find_barcode_umi_no_polya_batch(self, read_chunk):
        """The reference's second strand rule (barcode_callers.py:231-248) over a chunk: the forward result if it is
        valid, else the reverse one if valid, else the more informative of the two."""
        recs = self.extract_records([s for _, s in read_chunk], _native.STRAND_RULE_NO_POLYA)
        return [record_to_result(rid, s, r) for (rid, s), r in zip(read_chunk, recs)]

    def find_barcode_umi_no_polya(self, read_id, sequence):
        return self.find_barcode_umi_no_polya_batch([(read_id, sequence)])[0]

    @staticmethod
    def result_type():
        return TenXBarcodeDetectionResult


class TenXBarcodeExtractorV2(TenXBarcodeExtractor):
    def __init__(self, device=0, instance=0):
        TenXBarcodeExtractor.__init__(self, TenXVersions.v2, device, instance)


class TenXBarcodeExtractorV3(TenXBarcodeExtractor):
    def __init__(self, device=0, instance=0):
        TenXBarcodeExtractor.__init__(self, TenXVersions.v3, device, instance)
