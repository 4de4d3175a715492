"""GlobalAligner<int>::align (L/alignment/GlobalAlignerImpl.hh:35-228), the haplotype-to-reference aligner of the
active-region code.

  * known-answer tests: the 22 cases of the reference's own unit test (L/alignment/test/GlobalAlignerTest.cpp:30-347);
  * the C restatement (oracle/strelka_oracle.c sko_global_align) against the reference itself on random pairs when
    oracle/_ref is present;
  * the HIP kernel against the KATs and the restatement (GPU tests), including queries longer than one 64-lane strip and
    problems whose back-pointer matrix does not fit LDS.
Integer DP: score, begin position and CIGAR are compared exactly."""
import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import capi, synth


def _S(off_edge=-4, insert_delete=0, allow=0, require=0):
    return dict(match=2, mismatch=-4, open=-5, extend=-1, off_edge=off_edge, insert_delete=insert_delete,
                allow=allow, require=require)


# (query, ref, scores, cigar, begin_pos, score or None) -- GlobalAlignerTest.cpp, in file order
KATS = [
    ("D", "ABCDEF", _S(), "1=", 3, None),
    ("BCDEFHIKLM", "ABCDEFGHIKLMN", _S(), "5=1D5=", 1, None),
    ("BCDEFGXHIKLM", "ABCDEFGHIKLMN", _S(), "6=1I5=", 1, None),
    ("BBBBBBCDXYZHIKLMMMM", "ABBBBBBCDEFGHIKLMMMMN", _S(), "8=3I3D8=", 1, None),
    ("BBBBBBCDEXYHIKLMMMM", "ABBBBBBCDEFGHIKLMMMMN", _S(), "9=2X8=", 1, None),
    ("ABCD", "BCD", _S(), "1S3=", 0, 2),
    ("ABCD", "ABC", _S(), "3=1S", 0, 2),
    ("ABCD", "B", _S(), "1S1=2S", 0, -10),
    ("ABCDEFFFFFGHIJKL", "ABCDEFFFFFFGHIJKL", _S(), "5=1D11=", 0, None),     # left-aligned deletion in a repeat
    ("ABCDEFFFFFFFGHIJKL", "ABCDEFFFFFFGHIJKL", _S(), "5=1I12=", 0, None),    # left-aligned insertion
    ("AABCC", "ZZBYY", _S(), "2X1=2X", 0, None),                              # global over the query
    ("12ABCDEFGHIJ12", "ABCDEFGHIJ", _S(-100), "1X2I8=2I1X", 0, None),
    ("12ABCDEFGHIJ12", "ABCDEFGHIJ", _S(-100, 0, 1), "2I10=2I", 0, 6),
    ("AB", "A", _S(-100, 0, 1), "1=1I", 0, -4),
    ("AB", "B", _S(-100, 0, 1), "1I1=", 0, -4),
    ("CDEFFFGHIJ", "ABCDEFFFGHIJKL", _S(-100, 0, 1), "10=", 2, 20),
    ("CDEFFFGHIJ", "ABCDEFFFGHIJKL", _S(-100, 0, 1, 1), "2D10=2D", 0, 6),
    ("A", "AC", _S(-100, 0, 1, 1), "1=1D", 0, -4),
    ("C", "AC", _S(-100, 0, 1, 1), "1D1=", 0, -4),
    ("ATTT", "AT", _S(-100, 0, 1, 1), "1=2I1=", 0, -3),
    ("AT", "ATTT", _S(-100, 0, 1, 1), "1=2D1=", 0, None),
    ("GCG", "GCCC", _S(-100, -10, 1, 1), "1=1D1=1X", 0, None),
]


def _osc(d):
    return pyoracle.align_scores(d["match"], d["mismatch"], d["open"], d["extend"], d["off_edge"], d["insert_delete"],
                                 d["allow"], d["require"])


def _gsc(d):
    return capi.align_scores(match=d["match"], mismatch=d["mismatch"], open=d["open"], extend=d["extend"],
                             off_edge=d["off_edge"], insert_delete=d["insert_delete"], is_allow_edge_insertion=d["allow"],
                             is_require_edge_deletion=d["require"])


@pytest.mark.parametrize("query,ref,sc,cigar,begin,score", KATS)
def test_restatement_reference_kats(built, query, ref, sc, cigar, begin, score):
    s, b, c = pyoracle.global_align(query, ref, _osc(sc))
    assert (c, b) == (cigar, begin)
    if score is not None:
        assert s == score


def _random_pairs(rng, n, max_ref=60, long_every=0):
    out = []
    for t in range(n):
        R = int(rng.integers(1, max_ref))
        if long_every and t % long_every == 0:
            R = int(rng.integers(150, 400))
        ref = "".join("ACGTN"[int(x)] for x in rng.choice(5, R, p=[.24, .24, .24, .24, .04]))
        a = int(rng.integers(0, R))
        b = int(rng.integers(a, R)) + 1
        q = list(ref[a:b])
        for _ in range(int(rng.integers(0, 5))):
            k = int(rng.integers(0, len(q) + 1))
            op = rng.random()
            if op < 0.4 and q:
                q[min(k, len(q) - 1)] = "ACGT"[int(rng.integers(4))]
            elif op < 0.7:
                q[k:k] = list("ACGT"[int(rng.integers(4))] * int(rng.integers(1, 8)))
            elif len(q) > 2:
                del q[k:k + int(rng.integers(1, 6))]
        out.append(("".join(q) or "A", ref))
    return out


def _random_scores(rng):
    return dict(match=int(rng.choice([1, 2])), mismatch=-4, open=-5, extend=int(rng.choice([-1, 0])),
                off_edge=int(rng.choice([-4, -100])), insert_delete=int(rng.choice([0, -5, -10])),
                allow=int(rng.integers(2)), require=int(rng.integers(2)))


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_restatement_matches_reference_live(built):
    rng = np.random.default_rng(12)
    for q, r in _random_pairs(rng, 1500):
        sc = _osc(_random_scores(rng))
        assert pyoracle.global_align(q, r, sc) == pyoracle.ref_global_align(q, r, sc)


@pytest.mark.gpu
def test_gpu_reference_kats(gpu):
    for query, ref, sc, cigar, begin, score in KATS:
        (s, b, c), = capi.global_align([(query, ref)], _gsc(sc))
        assert (c, b) == (cigar, begin), (query, ref)
        if score is not None:
            assert s == score


@pytest.mark.gpu
def test_gpu_matches_restatement(gpu):
    rng = np.random.default_rng(21)
    for rep in range(6):
        sc = _random_scores(rng) if rep else dict(match=1, mismatch=-4, open=-5, extend=-1, off_edge=-100, insert_delete=-5,
                                                  allow=1, require=1)  # the active-region detector's scores first
        pairs = _random_pairs(rng, 300, max_ref=90, long_every=25)
        got = capi.global_align(pairs, _gsc(sc))
        for (q, r), g in zip(pairs, got):
            assert g == pyoracle.global_align(q, r, _osc(sc)), (q, r, sc)


@pytest.mark.gpu
def test_gpu_large_problem_uses_global_back_pointers(gpu):
    rng = np.random.default_rng(5)
    ref = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 900))
    q = list(ref[40:860])
    del q[300:317]
    q[500:500] = list("ACGTACGTAC")
    q = "".join(q)
    sc = dict(match=1, mismatch=-4, open=-5, extend=-1, off_edge=-100, insert_delete=-5, allow=1, require=1)
    (got,) = capi.global_align([(q, ref)], _gsc(sc))
    assert got == pyoracle.global_align(q, ref, _osc(sc))
    with pytest.raises(capi.StrelkaAmdError, match="1..1024"):
        capi.global_align([("A" * 1025, "ACGT")])
    with pytest.raises(capi.StrelkaAmdError, match="1..1024"):
        capi.global_align([("", "ACGT")])


@pytest.mark.gpu
def test_gpu_device_resident_entry_equals_host_entry(gpu):
    from strelka_amd import device
    rng = np.random.default_rng(412)
    pairs = synth.align_pairs(300, rng, ref_len=(20, 330), max_edits=4)
    d = device.DeviceGlobalAlignBatch(pairs, "cuda:0")
    d.align()
    assert d.results() == capi.global_align(pairs)
