"""CPU restatement of the feed -- TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may use it.

In the reference BGZF inflation and BAM parsing belong to a third-party dependency, htslib 1.7-6-g6d2bfb7 (redist/; behind
L/htsapi/bam_streamer.cpp:268 sam_itr_next), which calls zlib.  The published formats are restated here with Python's zlib
binding and struct: BGZF (SAM specification v1 section 4.1: gzip members with a 'BC' extra subfield, CRC-32 + ISIZE trailer) and
BAM records (section 4.2).  Pinned to `samtools view` (samtools 1.6 from the reference's redist/, built by oracle/Makefile): the
committed tests/golden/feed_tiny.sam.txt.gz and, where oracle/_ref exists, the demo and synthetic BAMs live."""
import struct
import zlib

import numpy as np


def bgzf_blocks(data):
    """[(offset, size, isize)] of the BGZF blocks of a file image"""
    out, at, n = [], 0, len(data)
    while at < n:
        if data[at:at + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF block at %d" % at)
        xlen = struct.unpack_from("<H", data, at + 10)[0]
        x, bsize = 0, None
        while x + 4 <= xlen:
            si1, si2, slen = struct.unpack_from("<BBH", data, at + 12 + x)
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", data, at + 12 + x + 4)[0] + 1
            x += 4 + slen
        if bsize is None:
            raise ValueError("no BC subfield")
        isize = struct.unpack_from("<I", data, at + bsize - 4)[0]
        out.append((at, bsize, isize))
        at += bsize
    return out


def bgzf_inflate(data):
    """the concatenated inflated stream; CRC-32 and ISIZE of every block checked (htslib bgzf.c:472-493)"""
    parts = []
    for at, bsize, isize in bgzf_blocks(data):
        xlen = struct.unpack_from("<H", data, at + 10)[0]
        raw = zlib.decompress(data[at + 12 + xlen: at + bsize - 8], -15)
        crc = struct.unpack_from("<I", data, at + bsize - 8)[0]
        if len(raw) != isize or (zlib.crc32(raw) & 0xffffffff) != crc:
            raise ValueError("BGZF block at %d: bad ISIZE / CRC-32" % at)
        parts.append(raw)
    return b"".join(parts)


def bam_header_end(stream):
    if stream[:4] != b"BAM\x01":
        raise ValueError("not BAM")
    at = 8 + struct.unpack_from("<i", stream, 4)[0]
    n_ref = struct.unpack_from("<i", stream, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 4 + struct.unpack_from("<i", stream, at)[0] + 4
    return at


CIGAR_OPS = "MIDNSHP=X"
SEQ_CODES = "=ACMGRSVTWYHKDBN"


def bam_records(stream, first=None):
    """list of dicts: the fields bam_record.hh exposes, the sequence as BAM 4-bit codes, qualities, cigar as (op, len)"""
    at = bam_header_end(stream) if first is None else first
    out = []
    while at + 4 <= len(stream):
        block_size = struct.unpack_from("<i", stream, at)[0]
        if at + 4 + block_size > len(stream):
            break
        ref_id, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, mref, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", stream, at + 4)
        p = at + 36 + l_read_name
        cig = struct.unpack_from("<%dI" % n_cigar, stream, p)
        p += 4 * n_cigar
        packed = np.frombuffer(stream, np.uint8, (l_seq + 1) // 2, p)
        code = np.empty(2 * len(packed), np.uint8)
        code[0::2] = packed >> 4
        code[1::2] = packed & 15
        p += (l_seq + 1) // 2
        qual = np.frombuffer(stream, np.uint8, l_seq, p)
        out.append(dict(offset=at, ref_id=ref_id, pos=pos, mapq=mapq, flag=flag, l_seq=l_seq, mate_ref_id=mref, mate_pos=mpos, template_size=tlen,
                        cigar=[(c & 15, c >> 4) for c in cig], code=code[:l_seq].copy(), qual=qual.copy()))
        at += 4 + block_size
    return out


def sam_fields(rec):
    """(flag, refid-independent) text fields as `samtools view` prints them: FLAG, POS, MAPQ, CIGAR, SEQ, QUAL"""
    cigar = "".join("%d%s" % (l, CIGAR_OPS[op]) for op, l in rec["cigar"]) or "*"
    seq = "".join(SEQ_CODES[c] for c in rec["code"]) or "*"
    qual = "".join(chr(int(q) + 33) for q in rec["qual"]) if len(rec["qual"]) and rec["qual"][0] != 255 else "*"
    return str(rec["flag"]), str(rec["pos"] + 1), str(rec["mapq"]), cigar, seq, qual
