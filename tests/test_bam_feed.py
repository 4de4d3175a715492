"""SURVEY.md 8f rank 4, the feed: BGZF inflation + BAM record decoding.

  * the Python restatement (oracle/bam_oracle.py: zlib + struct) is pinned to `samtools view` -- the committed text of the tiny
    fixture, and live on the reference's demo BAMs where oracle/_ref holds samtools;
  * -m "not gpu": the host chain walks (sk_bgzf_scan, sk_bam_scan_records, sk_bam_header_end) against the restatement;
  * -m gpu: the kernels -- every inflated byte equal to zlib's, every decoded field / base / quality / path segment equal to the
    restatement's, on the fixture, the demo BAMs and the synthetic ones; malformed blocks are refused, not mis-decoded."""
import glob
import gzip
import os
import subprocess

import numpy as np
import pytest

from oracle import bam_oracle, pyoracle
from strelka_amd import capi
from tests import e2e_util as E

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY = os.path.join(GOLD, "feed_tiny.bam")
SAMTOOLS = os.path.join(E.REF_DIR, "bin", "samtools")


def _bytes(path):
    with open(path, "rb") as f:
        return f.read()


def _more_bams():
    out = [TINY]
    for d in (os.path.join(E.REF_DIR, "demo"), os.path.join(E.REF_DIR, "synth"), os.path.join(E.REF_DIR, "synth", "long_reads")):
        out += sorted(glob.glob(os.path.join(d, "*.bam")))
    return out


def test_restatement_reproduces_samtools_view_of_the_fixture():
    recs = bam_oracle.bam_records(bam_oracle.bgzf_inflate(_bytes(TINY)))
    with gzip.open(os.path.join(GOLD, "feed_tiny.sam.txt.gz"), "rt") as f:
        want = [l.rstrip("\n").split("\t") for l in f]
    assert len(recs) == len(want) == 687
    for r, w in zip(recs, want):
        flag, pos, mapq, cigar, seq, qual = bam_oracle.sam_fields(r)
        assert [flag, w[1], pos, mapq, cigar, seq, qual] == [w[0], w[1], w[2], w[3], w[4], w[5], w[6]]


@pytest.mark.skipif(not os.path.exists(SAMTOOLS), reason="oracle/_ref samtools not built")
def test_restatement_reproduces_samtools_view_live():
    n = 0
    for bam in _more_bams()[1:4]:
        recs = bam_oracle.bam_records(bam_oracle.bgzf_inflate(_bytes(bam)))
        text = subprocess.run([SAMTOOLS, "view", bam], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
        assert len(recs) == len(text) > 100
        for r, line in zip(recs, text):
            w = line.split("\t")
            flag, pos, mapq, cigar, seq, qual = bam_oracle.sam_fields(r)
            assert (flag, pos, mapq, cigar, seq, qual) == (w[1], w[3], w[4], w[5], w[9], w[10])
        n += len(recs)
    assert n > 1000


@pytest.mark.parametrize("bam", _more_bams()[:3])
def test_host_chain_walks(bam):
    data = np.frombuffer(_bytes(bam), np.uint8)
    blocks = bam_oracle.bgzf_blocks(data.tobytes())
    block_off, out_off = capi.bgzf_scan(data)
    assert list(block_off[:-1]) == [b[0] for b in blocks] and block_off[-1] == len(data)
    assert list(np.diff(out_off)) == [b[2] for b in blocks]
    stream = bam_oracle.bgzf_inflate(data.tobytes())
    s = np.frombuffer(stream, np.uint8)
    assert capi.lib().sk_bam_header_end(capi._p(s), len(s)) == bam_oracle.bam_header_end(stream)
    recs = bam_oracle.bam_records(stream)
    rec_off, read_off, path_off = capi.bam_scan_records(s)
    assert list(rec_off) == [r["offset"] for r in recs]
    assert list(np.diff(read_off)) == [r["l_seq"] for r in recs]
    assert list(np.diff(path_off)) == [len(r["cigar"]) for r in recs]
    # a stream cut in the middle of a record: the whole records before the cut, nothing else
    cut = int(rec_off[len(rec_off) // 2]) + 17
    ro, _, _ = capi.bam_scan_records(s[:cut])
    assert len(ro) == len(rec_off) // 2
    # malformed input is refused
    bad = data.copy()
    bad[1] = 0
    assert capi.lib().sk_bgzf_scan(capi._p(bad), len(bad), None, None, 0) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("bam", _more_bams())
def test_inflate_and_decode_equal_the_restatement(bam):
    capi.init(0)
    data = np.frombuffer(_bytes(bam), np.uint8)
    want_stream = bam_oracle.bgzf_inflate(data.tobytes())
    got = capi.bgzf_inflate(data)
    assert got.tobytes() == want_stream
    recs = bam_oracle.bam_records(want_stream)
    d = capi.bam_decode(got)
    assert len(d["rec"]) == len(recs) > 100
    for k in ("ref_id", "pos", "mapq", "flag", "l_seq", "mate_ref_id", "mate_pos", "template_size"):
        assert np.array_equal(d["rec"][k], np.array([r[k] for r in recs], d["rec"][k].dtype)), k
    assert np.array_equal(d["rec"]["is_fwd_strand"], np.array([0 if r["flag"] & 16 else 1 for r in recs], np.uint8))
    assert np.array_equal(d["read_code"], np.concatenate([r["code"] for r in recs]))
    assert np.array_equal(d["read_qual"], np.concatenate([r["qual"] for r in recs]))
    want_path = [(op + 1, l) for r in recs for op, l in r["cigar"]]
    assert [(int(t), int(l)) for t, l in d["path"]] == want_path
    assert set(np.unique(d["read_code"])) <= {1, 2, 4, 8, 15}  # what sk_read_input.read_code takes


def _python_deflate_blocks(payloads, level, mem_level=8, strategy=0, flush_every=0):
    """a BGZF image made here (zlib raw deflate at `level`; level 0 = stored blocks): [(block bytes)] joined.  `mem_level` 1 makes
    zlib end a deflate block every few hundred symbols, `strategy` picks fixed codes / Huffman only / run lengths, `flush_every` > 0
    puts a full flush (an empty stored block, byte alignment) after every so many input bytes"""
    import struct
    import zlib
    out = bytearray()
    for p in payloads:
        c = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
        if flush_every:
            body = b"".join(c.compress(p[i:i + flush_every]) + c.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(p), flush_every)) + c.flush()
        else:
            body = c.compress(p) + c.flush()
        if len(body) + 26 > 65536:
            continue  # (does not fit a BGZF block)
        bsize = 12 + 6 + len(body) + 8
        out += bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", bsize - 1) + body
        out += struct.pack("<II", zlib.crc32(p) & 0xffffffff, len(p))
    return bytes(out)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["thread", "wave"])
def test_both_inflate_kernels_on_every_block_shape(kernel, monkeypatch):
    """the thread-per-block and the wave-per-block kernels ($SK_INFLATE_KERNEL) on the fixture BAMs and on blocks made here: stored
    blocks, fixed codes (tiny payloads), dynamic codes with long matches at short distances (runs), 64 KiB blocks, empty blocks, a
    destination that is not 4-byte aligned, codes longer than the first-level table (skewed alphabets)"""
    capi.init(0)
    monkeypatch.setenv("SK_INFLATE_KERNEL", kernel)
    rng = np.random.default_rng(12)
    for bam in _more_bams()[:4]:
        data = np.frombuffer(_bytes(bam), np.uint8)
        assert capi.bgzf_inflate(data).tobytes() == bam_oracle.bgzf_inflate(data.tobytes())
    skew = np.minimum(255, rng.geometric(0.02, 60000)).astype(np.uint8).tobytes()  # a long-tailed byte histogram: 14/15-bit codes
    payloads = [b"", b"a", b"abc" * 5, bytes(65280), bytes([7]) * 3 + bytes(range(256)) * 200, rng.integers(0, 256, 65280, dtype=np.uint8).tobytes(),
                rng.integers(0, 4, 65281, dtype=np.uint8).tobytes(), b"ACGT" * 16000 + b"N", skew, skew[:33333], b"xyz" * 333]
    for level in (0, 1, 6, 9):
        image = np.frombuffer(_python_deflate_blocks(payloads, level), np.uint8)
        assert capi.bgzf_inflate(image).tobytes() == b"".join(payloads), level
    # every way of breaking a block is refused by this kernel too
    data = np.frombuffer(_bytes(TINY), np.uint8).copy()
    block_off, out_off = capi.bgzf_scan(data)
    out = np.zeros(int(out_off[-1]), np.uint8)
    b = int(np.argmax(np.diff(block_off)))  # the largest block: deflate data well past its header
    for at in (int(block_off[b]) + 40, int(block_off[b]) + 400, int(block_off[b + 1]) - 7, int(block_off[b + 1]) - 2):
        bad = data.copy()
        bad[at] ^= 0x5a
        rc = capi.lib().sk_bgzf_inflate(capi._p(bad), capi._p(block_off), capi._p(out_off), len(block_off) - 1, capi._p(out))
        assert rc != 0 and ("block %d:" % b) in capi.last_error(), (at, capi.last_error())


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["thread", "wave"])
def test_inflate_kernels_on_many_deflate_streams(kernel, monkeypatch):
    """streams zlib writes under other settings than bgzip's: many deflate blocks per BGZF block (memLevel 1: tables rebuilt every
    few hundred symbols), fixed codes throughout, Huffman only (no matches), run lengths (distance 1 only), stored pieces and empty
    stored blocks between compressed ones (full flushes), on text-like, BAM-like, run-heavy and random payloads of every size class"""
    import zlib
    capi.init(0)
    monkeypatch.setenv("SK_INFLATE_KERNEL", kernel)
    rng = np.random.default_rng(13)
    bam_like = _bytes(TINY)
    raw = bam_oracle.bgzf_inflate(bam_like)
    payloads = []
    for n in (1, 2, 63, 64, 65, 257, 258, 259, 1000, 4095, 32768, 32769, 50000, 65280):
        payloads.append(raw[7:7 + n] if len(raw) >= 7 + n else (raw * (1 + (7 + n) // max(1, len(raw))))[7:7 + n])
        payloads.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        payloads.append(bytes(rng.choice(np.frombuffer(b"ACGTN=\x00\xff", np.uint8), n, p=[0.24, 0.24, 0.24, 0.24, 0.01, 0.01, 0.01, 0.01]).astype(np.uint8)))
        payloads.append((bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 300)) + rng.integers(0, 256, 5, dtype=np.uint8).tobytes()) * (1 + n // 150))
    payloads = [p[:65280] for p in payloads]
    want = b"".join
    settings = [(6, 1, zlib.Z_DEFAULT_STRATEGY, 0), (9, 1, zlib.Z_DEFAULT_STRATEGY, 0), (6, 8, zlib.Z_FIXED, 0), (6, 8, zlib.Z_HUFFMAN_ONLY, 0),
                (6, 8, zlib.Z_RLE, 0), (1, 9, zlib.Z_FILTERED, 0), (6, 8, zlib.Z_DEFAULT_STRATEGY, 777), (0, 8, zlib.Z_DEFAULT_STRATEGY, 1000),
                (6, 1, zlib.Z_FIXED, 333)]
    for level, mem_level, strategy, flush_every in settings:
        kept = [p for p in payloads if len(_python_deflate_blocks([p], level, mem_level, strategy, flush_every)) > 0]
        image = np.frombuffer(_python_deflate_blocks(kept, level, mem_level, strategy, flush_every), np.uint8)
        assert len(kept) > len(payloads) // 2
        assert capi.bgzf_inflate(image).tobytes() == want(kept), (level, mem_level, strategy, flush_every)


@pytest.mark.gpu
def test_malformed_blocks_are_refused():
    capi.init(0)
    data = np.frombuffer(_bytes(TINY), np.uint8).copy()
    block_off, out_off = capi.bgzf_scan(data)
    out = np.zeros(int(out_off[-1]), np.uint8)
    for what, at in (("deflate data", int(block_off[0]) + 40), ("CRC", int(block_off[1]) - 7)):
        bad = data.copy()
        bad[at] ^= 0x5a
        rc = capi.lib().sk_bgzf_inflate(capi._p(bad), capi._p(block_off), capi._p(out_off), len(block_off) - 1, capi._p(out))
        assert rc != 0 and "block 0" in capi.last_error(), what
    # and a good run after a bad one is clean
    assert capi.bgzf_inflate(data).tobytes() == bam_oracle.bgzf_inflate(data.tobytes())


# ---- the index: hts_itr_query + hts_itr_next's record test (sk_bai_query, sk_bam_region_filter) -------------------------------------------

import json


def _zlib_inflate(data, block_off, out_off):
    return np.frombuffer(bam_oracle.bgzf_inflate(np.asarray(data, np.uint8).tobytes()), np.uint8)


def _restatement_decode(stream, first):
    """capi.bam_decode's dict from the Python restatement (the -m "not gpu" runs have no kernels to decode with)"""
    recs = bam_oracle.bam_records(np.asarray(stream, np.uint8).tobytes(), first)
    rec = np.zeros(len(recs), capi.BAM_RECORD_DTYPE)
    for k in ("ref_id", "pos", "mapq", "flag", "l_seq", "mate_ref_id", "mate_pos", "template_size"):
        rec[k] = [r[k] for r in recs]
    rec["n_cigar"] = [len(r["cigar"]) for r in recs]
    rec["is_fwd_strand"] = [0 if r["flag"] & 16 else 1 for r in recs]
    path = np.array([(op + 1, l) for r in recs for op, l in r["cigar"]], capi.PATH_SEG_DTYPE) if recs else np.zeros(0, capi.PATH_SEG_DTYPE)
    off = lambda xs: np.concatenate([[0], np.cumsum(xs)]).astype(np.int64)
    return dict(rec=rec, rec_off=np.array([r["offset"] for r in recs], np.int64), read_off=off([r["l_seq"] for r in recs]),
                read_code=np.concatenate([r["code"] for r in recs]) if recs else np.zeros(0, np.uint8),
                read_qual=np.concatenate([r["qual"] for r in recs]) if recs else np.zeros(0, np.uint8),
                path_off=off([len(r["cigar"]) for r in recs]), path=path)


def _check_regions(bam_path, regions, on_gpu):
    """regions: [(ref_id, begin, end, ordinals of the records samtools view returns)]"""
    bam = np.frombuffer(_bytes(bam_path), np.uint8)
    bai = np.frombuffer(_bytes(bam_path + ".bai"), np.uint8)
    blocks = capi.bgzf_scan(bam)
    stream = bam_oracle.bgzf_inflate(bam.tobytes())
    all_recs = bam_oracle.bam_records(stream)
    ordinal = {r["offset"]: i for i, r in enumerate(all_recs)}
    kw = {} if on_gpu else dict(inflate=_zlib_inflate, decode=_restatement_decode)
    n = 0
    for ref_id, begin, end, want in regions:
        got = capi.bam_fetch_region(bam, bai, ref_id, begin, end, blocks=blocks, **kw)
        assert [ordinal[int(o)] for o in got["stream_offset"]] == list(want), (bam_path, ref_id, begin, end)
        for j, i in enumerate(want[:50]):  # and they are those records
            r = all_recs[i]
            assert int(got["rec"]["pos"][j]) == r["pos"] and int(got["rec"]["flag"][j]) == r["flag"]
            assert np.array_equal(got["read_code"][j], r["code"]) and np.array_equal(got["read_qual"][j], r["qual"])
            assert [(int(t), int(l)) for t, l in got["path"][j]] == [(op + 1, l) for op, l in r["cigar"]]
        n += len(want)
    return n


def _golden_regions():
    with gzip.open(os.path.join(GOLD, "feed_regions.json.gz"), "rt") as f:
        g = json.load(f)
    return [tuple(r) for r in g["regions"]]


def test_region_fetch_reproduces_samtools_view_on_the_fixture():
    regions = _golden_regions()
    assert len(regions) > 100 and sum(1 for r in regions if not r[3]) > 10
    assert _check_regions(os.path.join(GOLD, "feed_regions.bam"), regions, on_gpu=False) > 10000


def test_index_query_edge_cases():
    bai = np.frombuffer(_bytes(os.path.join(GOLD, "feed_regions.bam.bai")), np.uint8)
    assert len(capi.bai_query(bai, 2, 0, 20000)) == 0          # a reference without records
    assert len(capi.bai_query(bai, 0, 500, 500)) == 0          # an empty interval has no bins (reg2bins)
    assert len(capi.bai_query(bai, 0, 700, 600)) == 0          # end < begin
    assert np.array_equal(capi.bai_query(bai, 0, -5, 1000), capi.bai_query(bai, 0, 0, 1000))  # hts_itr_query clamps begin
    with pytest.raises(capi.StrelkaAmdError):
        capi.bai_query(bai, 4, 0, 10)                          # four references in the index
    with pytest.raises(capi.StrelkaAmdError):
        capi.bai_query(bai[:40], 0, 0, 10)                     # cut short
    ch = capi.bai_query(bai, 0, 0, 1 << 29)
    assert len(ch) >= 1 and all(int(c["begin"]) < int(c["end"]) for c in ch)
    assert all(int(a["end"]) <= int(b["begin"]) for a, b in zip(ch[:-1], ch[1:]))  # sorted, disjoint


def _samtools_regions(bam, rng, n_regions):
    """random regions of an indexed BAM with what samtools view returns for them"""
    head = subprocess.run([SAMTOOLS, "view", "-H", bam], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    contigs = [(l.split("\t")[1][3:], int(l.split("\t")[2][3:])) for l in head if l.startswith("@SQ")]
    names = [l.split("\t")[0] + "/" + l.split("\t")[1] for l in subprocess.run([SAMTOOLS, "view", bam], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()]
    first = {}
    for i, q in enumerate(names):
        first.setdefault(q, []).append(i)
    regions = []
    for _ in range(n_regions):
        ci = int(rng.integers(0, len(contigs)))
        length = contigs[ci][1]
        b = int(rng.integers(0, length))
        e = min(b + int(rng.choice([1, 150, 2000, 16384, 40000, 200000])), 1 << 29)
        out = subprocess.run([SAMTOOLS, "view", bam, "%s:%d-%d" % (contigs[ci][0], b + 1, e)], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
        used, want = {}, []
        for l in out:  # (name/flag pairs may repeat: take their occurrences in order)
            q = l.split("\t")[0] + "/" + l.split("\t")[1]
            k = used.get(q, 0)
            used[q] = k + 1
            want.append(first[q][k])
        regions.append((ci, b, e, sorted(want)))
    return regions


@pytest.mark.skipif(not os.path.exists(SAMTOOLS), reason="oracle/_ref samtools not built")
def test_region_fetch_reproduces_samtools_view_live():
    """larger BAMs, whose indexes use the fine bins (htslib merges bins with less than 64 KB of records into their parents)"""
    rng = np.random.default_rng(555)
    bams = [b for b in _more_bams()[1:] if os.path.exists(b + ".bai")]
    assert bams
    n = 0
    for bam in bams[:3]:
        n += _check_regions(bam, _samtools_regions(bam, rng, 25), on_gpu=False)
    assert n > 5000


@pytest.mark.skipif(not os.path.exists(SAMTOOLS), reason="oracle/_ref samtools not built")
def test_region_fetch_on_a_fine_grained_index(tmp_path):
    """a BAM made on the spot that is large enough for bins on every level: 200 000 short reads on 3 x 3 Mb, with deep piles (a 16 kb
    bin of its own needs 64 KB of records) and reads spanning tens of kilobases (the higher bins)"""
    rng = np.random.default_rng(556)
    contigs = [("c%d" % i, 3_000_000) for i in range(3)]
    lines = ["@HD\tVN:1.5\tSO:unsorted"] + ["@SQ\tSN:%s\tLN:%d" % c for c in contigs]
    seq, qual = "ACGTTGCAAC" * 3, "I" * 30
    k = 0
    for name, length in contigs:
        pos = np.concatenate([rng.integers(1, length - 100, 50000), rng.integers(700000, 700000 + 3000, 12000),
                              rng.integers(2000000, 2000000 + 40000, 5000)])
        for p in pos:
            lines.append("r%d\t0\t%s\t%d\t30\t30M\t*\t0\t0\t%s\t%s" % (k, name, p, seq, qual))
            k += 1
        for p in rng.integers(1, length - 600000, 300):
            gap = int(rng.choice([20000, 150000, 500000]))
            lines.append("r%d\t0\t%s\t%d\t30\t15M%dN15M\t*\t0\t0\t%s\t%s" % (k, name, p, gap, seq, qual))
            k += 1
    sam = tmp_path / "big.sam"
    sam.write_text("\n".join(lines) + "\n")
    bam = str(tmp_path / "big.bam")
    subprocess.run([SAMTOOLS, "sort", "-o", bam, str(sam)], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([SAMTOOLS, "index", bam], check=True)
    bai = np.frombuffer(_bytes(bam + ".bai"), np.uint8)
    assert len(bai) > 3000                                                          # many bins
    assert max(len(capi.bai_query(bai, 0, b, b + 600000)) for b in (0, 650000, 1900000)) >= 3  # several chunks for one region
    regions = _samtools_regions(bam, rng, 60)
    assert _check_regions(bam, regions, on_gpu=False) > 20000
    # and chunk for chunk what htslib's own iterator holds (oracle/ref/ref_driver_bai.cpp: hts_idx_load + sam_itr_queryi)
    n_chunks = 0
    for _ in range(400):
        tid = int(rng.integers(0, 3))
        b = int(rng.integers(-10, 3_000_000))
        e = b + int(rng.choice([0, 1, 100, 16384, 16385, 131072, 500000, 4_000_000]))
        want = pyoracle.ref_bai_query(bam, tid, b, e)
        got = [(int(c["begin"]), int(c["end"])) for c in capi.bai_query(bai, tid, b, e)]
        assert got == want, (tid, b, e)
        n_chunks += len(want)
    assert n_chunks > 500


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")
def test_index_query_equals_htslib_on_the_fixture_and_the_demo_bams():
    rng = np.random.default_rng(557)
    n = 0
    for bam in [os.path.join(GOLD, "feed_regions.bam")] + [b for b in _more_bams()[1:] if os.path.exists(b + ".bai")][:4]:
        bai = np.frombuffer(_bytes(bam + ".bai"), np.uint8)
        n_ref = int(np.frombuffer(bai[4:8].tobytes(), "<i4")[0])
        for _ in range(150):
            tid = int(rng.integers(0, n_ref))
            b = int(rng.integers(0, 250000))
            e = b + int(rng.choice([0, 1, 50, 5000, 16384, 70000, 1 << 29]))
            want = pyoracle.ref_bai_query(bam, tid, b, e)
            assert [(int(c["begin"]), int(c["end"])) for c in capi.bai_query(bai, tid, b, e)] == want, (bam, tid, b, e)
            n += len(want)
    assert n > 150  # (these indexes are coarse: one or two chunks per region; the fine-grained one is above)


@pytest.mark.gpu
def test_prefix_only_slice_keeps_the_stream():
    """a slice that is only the carried record (no blocks): out receives the prefix and sk_bam_decode_kept decodes from the
    kept stream, as the header promises"""
    capi.init(0)
    stream = np.frombuffer(bam_oracle.bgzf_inflate(_bytes(TINY)), np.uint8)
    rec_off, read_off, path_off = capi.bam_scan_records(stream)
    a, b = int(rec_off[3]), int(rec_off[5])  # two whole records
    prefix = np.ascontiguousarray(stream[a:b])
    out = np.zeros(len(prefix), np.uint8)
    none = np.zeros(1, np.int64)
    rc = capi.lib().sk_bgzf_inflate_prefixed(capi._p(none.view(np.uint8)), capi._p(none), capi._p(none), 0, capi._p(prefix), len(prefix), capi._p(out))
    assert rc == 0, capi.last_error()
    assert out.tobytes() == prefix.tobytes()
    want = capi.bam_decode(prefix, first=0)
    ro, rd, po = capi.bam_scan_records(prefix, 0)
    n = len(ro)
    assert n == 2
    rec = np.zeros(n, capi.BAM_RECORD_DTYPE)
    code = np.zeros(int(rd[-1]), np.uint8)
    qual = np.zeros(int(rd[-1]), np.uint8)
    path = np.zeros(int(po[-1]), capi.PATH_SEG_DTYPE)
    rc = capi.lib().sk_bam_decode_kept(capi._p(prefix), len(prefix), capi._p(ro), n, capi._p(rd), capi._p(po), capi._p(rec), capi._p(code), capi._p(qual), capi._p(path))
    assert rc == 0, capi.last_error()
    assert rec.tobytes() == want["rec"].tobytes() and code.tobytes() == want["read_code"].tobytes() and qual.tobytes() == want["read_qual"].tobytes()


@pytest.mark.gpu
def test_region_fetch_through_the_kernels():
    capi.init(0)
    assert _check_regions(os.path.join(GOLD, "feed_regions.bam"), _golden_regions(), on_gpu=True) > 10000


# ---- normalizeAlignment (L/starling_common/normalizeAlignment.cpp:647-703): csrc/normalize_core.h on the host and as a kernel ---------------

import ctypes as C
import pickle

from oracle import pyoracle
from strelka_amd import synth


def _normalize_golden():
    with open(os.path.join(GOLD, "normalize_reference.pkl"), "rb") as f:
        g = pickle.load(f)
    code = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}
    cases = [dict(c, code=np.array([code[x] for x in c["read"]], np.uint8)) for c in g["cases"]]
    return cases, [tuple(e) for e in g["expect"]]


def _run_batched(cases, library=None):
    out = []
    i = 0
    while i < len(cases):  # one call per run of alignments that share a reference segment
        j = i
        while j < len(cases) and cases[j]["ref_seq"] is cases[i]["ref_seq"] and cases[j]["ref_offset"] == cases[i]["ref_offset"]:
            j += 1
        out += capi.normalize_alignments(cases[i]["ref_seq"], cases[i]["ref_offset"], cases[i:j], library=library)
        i = j
    return out


def _double():
    path = os.path.join(os.path.dirname(GOLD), "..", "oracle", "libstrelka_amd_double.so")
    if not os.path.exists(path):
        pytest.skip("oracle/libstrelka_amd_double.so not built")
    L = C.CDLL(os.path.abspath(path))
    L.sk_last_error.restype = C.c_char_p
    L.sk_normalize_alignments.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 7
    return L


def test_normalize_core_reproduces_the_reference_golden():
    cases, expect = _normalize_golden()
    got = _run_batched(cases, library=_double())
    assert sum(e[0] for e in expect) > 400
    assert got == expect


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")
def test_normalize_core_reproduces_the_live_reference():
    cases = synth.normalize_cases(2500, np.random.default_rng(99001))
    want = [pyoracle.ref_normalize_alignment(c["ref_seq"], c["ref_offset"], c["read"], c["pos"], c["path"]) for c in cases]
    assert _run_batched(cases, library=_double()) == want


@pytest.mark.gpu
def test_normalize_kernel_reproduces_the_reference_golden():
    capi.init(0)
    cases, expect = _normalize_golden()
    assert _run_batched(cases) == expect
    # many alignments against one segment in one launch
    big = [c for c in cases if c["ref_seq"] is cases[0]["ref_seq"]] * 5000
    got = capi.normalize_alignments(cases[0]["ref_seq"], cases[0]["ref_offset"], big)
    assert got == [expect[0]] * len(big)
