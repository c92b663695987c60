"""Minimal BAM / SAM writers for the tests (SAM specification 4.2): just enough to hand the native reader the files pysam
would read - unmapped and mapped records, secondary / supplementary flags, CIGAR and tag bytes to step over."""
import struct
import zlib

NT16 = "=ACMGRSVTWYHKDBN"


def bgzf(data, block=65280, eof_marker=True, level=6):
    """data as a BGZF file (SAM specification 4.1): gzip members of at most 64 KiB that state their own size"""
    out = bytearray()
    pieces = [data[a:a + block] for a in range(0, len(data), block)] + ([b""] if eof_marker else [])
    for piece in pieces:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(piece) + co.flush()
        total = 18 + len(body) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, total - 1)
        out += body + struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece))
    return bytes(out)


def bam_raw(records, refs=(("chr1", 100000), ("chrM", 16569)), text="@HD\tVN:1.6\tSO:unsorted\n"):
    """records: (name, flag, seq[, cigar ops list of (len, op), tags bytes]) -> uncompressed BAM stream"""
    out = bytearray(b"BAM\x01")
    t = text.encode()
    out += struct.pack("<i", len(t)) + t + struct.pack("<i", len(refs))
    for name, length in refs:
        nb = name.encode() + b"\0"
        out += struct.pack("<i", len(nb)) + nb + struct.pack("<i", length)
    for rec in records:
        name, flag, seq = rec[0], rec[1], rec[2]
        cigar = rec[3] if len(rec) > 3 else []
        tags = rec[4] if len(rec) > 4 else b""
        nb = name.encode() + b"\0"
        packed = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            code = NT16.index(ch)
            packed[i // 2] |= code << 4 if i % 2 == 0 else code
        cig = b"".join(struct.pack("<I", (n << 4) | "MIDNSHP=X".index(op)) for n, op in cigar)
        mapped = not (flag & 4)
        body = struct.pack("<iiBBHHHiiii", 0 if mapped else -1, 99 if mapped else -1, len(nb), 60 if mapped else 0, 4680,
                           len(cigar), flag, len(seq), -1, -1, 0)
        body += nb + cig + bytes(packed) + b"\xff" * len(seq) + tags
        out += struct.pack("<i", len(body)) + body
    return bytes(out)


def sam_text(records, with_header=True):
    """the same records as SAM text"""
    lines = ["@HD\tVN:1.6\tSO:unsorted", "@SQ\tSN:chr1\tLN:100000"] if with_header else []
    for rec in records:
        name, flag, seq = rec[0], rec[1], rec[2]
        cigar = "".join("%d%s" % (n, op) for n, op in rec[3]) if len(rec) > 3 and rec[3] else "*"
        mapped = not (flag & 4)
        lines.append("\t".join([name, str(flag), "chr1" if mapped else "*", "100" if mapped else "0", "60" if mapped else "0",
                                cigar, "*", "0", "0", seq if seq else "*", "*", "NM:i:0"]))
    return "\n".join(lines) + "\n"
