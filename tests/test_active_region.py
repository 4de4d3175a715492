"""Next row f2, post-processing: sk_discover_indels_and_mismatches (haplotype alignment -> left-shifted primitive alleles)
against the REFERENCE's ActiveRegionProcessor::discoverIndelsAndMismatches (L/starling_common/ActiveRegionProcessor.cpp:572-697).

  * tests/golden/active_region_reference.pkl: the reference's outputs (oracle/_ref) on seeded scenarios, with the CIGAR its own
    GlobalAligner produced; the CPU test feeds that CIGAR to the host stage, the GPU test runs sk_global_align first;
  * when oracle/_ref is present the same comparison also runs live on fresh scenarios.
Integer / byte work: everything is exact."""
import os
import pickle

import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import capi, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "active_region_reference.pkl")


@pytest.fixture(scope="module")
def gold(built):
    with open(GOLD, "rb") as f:
        return pickle.load(f)


def reference_outputs(scenarios):
    """(used by tests/make_golden.py too)"""
    from types import SimpleNamespace
    sc = SimpleNamespace(match=1, mismatch=-4, open=-5, extend=-1, offEdge=-100, insertDelete=-5, isAllowEdgeInsertion=1,
                         isRequireEdgeDeletion=1)
    out = []
    for s in scenarios:
        b, e = s["ar_begin"] - s["ref_offset"], s["ar_end"] - s["ref_offset"]
        score, beg, cigar = pyoracle.ref_global_align(s["haplotype"], s["ref_seq"][b:e], sc)
        keys, n_indels = pyoracle.ref_discover_indels_and_mismatches(s["ref_seq"], s["ref_offset"], s["ar_begin"], s["ar_end"],
                                                                     s["prev_ar_end"], s["max_indel_size"], s["haplotype"])
        out.append(dict(begin_pos=beg, cigar=cigar, keys=keys, n_indels=n_indels))
    return out


def _run(s, beg, cigar):
    return capi.discover_indels_and_mismatches(s["ref_seq"], s["ref_offset"], s["ar_begin"], s["ar_end"], s["prev_ar_end"],
                                               s["max_indel_size"], s["haplotype"], beg, cigar)


def _check(scenarios, expect, cigars=None):
    n_shifted = 0
    for i, (s, w) in enumerate(zip(scenarios, expect)):
        got, ni = _run(s, w["begin_pos"], w["cigar"] if cigars is None else cigars[i])
        assert got == [tuple(k) for k in w["keys"]], i
        assert ni == w["n_indels"], i
        n_shifted += ni
    return n_shifted


def test_fixture_is_interesting(gold):
    keys = [k for w in gold["expect"] for k in w["keys"]]
    assert len(gold["scenarios"]) >= 300
    assert sum(1 for k in keys if k[1] == capi.INDEL["MISMATCH"]) > 100
    assert sum(1 for k in keys if k[1] == capi.INDEL["INDEL"] and k[2] == 0) > 100   # insertions
    assert sum(1 for k in keys if k[1] == capi.INDEL["INDEL"] and k[2] > 0) > 100    # deletions
    assert any(w["n_indels"] == 0 for w in gold["expect"])


def test_host_stage_matches_reference_golden(gold):
    assert _check(gold["scenarios"], gold["expect"]) > 200


def test_host_stage_matches_reference_live(built):
    if pyoracle.ref() is None:
        pytest.skip("oracle/_ref not built")
    scenarios = synth.active_region_scenarios(150, np.random.default_rng(77))
    _check(scenarios, reference_outputs(scenarios))


def test_rejects_bad_input(built):
    s = synth.active_region_scenarios(1, np.random.default_rng(3))[0]
    with pytest.raises(capi.StrelkaAmdError):
        _run(s, 1, "10=")          # alignment must begin at the region's first base
    with pytest.raises(capi.StrelkaAmdError):
        _run(s, 0, "5=3N5=")       # no skips in a haplotype alignment


@pytest.mark.gpu
def test_gpu_aligner_then_host_stage_matches_reference_golden(gpu, gold):
    sc = gold["scenarios"]
    pairs = [(s["haplotype"], s["ref_seq"][s["ar_begin"] - s["ref_offset"]:s["ar_end"] - s["ref_offset"]]) for s in sc]
    res = capi.global_align(pairs)
    for (score, beg, cigar), w in zip(res, gold["expect"]):
        assert (beg, cigar) == (w["begin_pos"], w["cigar"])
    _check(sc, gold["expect"], cigars=[r[2] for r in res])
