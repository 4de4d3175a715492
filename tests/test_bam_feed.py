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

from oracle import bam_oracle
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
