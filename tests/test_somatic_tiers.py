"""Rows a12/a13 complete: the whole of somatic_snv_caller_strand_grid::position_somatic_snv_call
(L/applications/strelka/position_somatic_snv_strand_grid.cpp:230-363) -- tier1/tier2 loop, the qphred==0 short cut, tier
selection, NTYPE conflict, per-site forced output, isComputeNonSomatic -- as one record per locus.

  reference (oracle/_ref, the reference's own translation units)  ==  C restatement (oracle)   [CPU, live + golden]
  C restatement                                                   ==  sk_somatic_snv_call_tiers on the GPU, bytes   [-m gpu]
"""
import os

import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import capi, synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "somatic_tiers_reference.npz")
FIELDS = ["ref_gt", "snv_tier", "snv_from_ntype_tier", "is_forced_output", "ntype", "max_gt", "qphred", "from_ntype_qphred",
          "nonsomatic_qphred", "normal_alt_id", "tumor_alt_id", "strand_bias"]  # is_computed is not a reference field


def scenario(seed, n=1200):
    rng = np.random.Generator(np.random.PCG64(seed))  # (= default_rng(seed), but not shifted by SK_TEST_SEED_OFFSET: golden inputs)
    n1, t1, n2, t2 = synth.somatic_tier_pileups(n, rng, normal_depth=30.0, tumor_depth=60.0, somatic_rate=0.08,
                                                het_rate=0.05, somatic_frac=float(rng.choice([0.1, 0.2, 0.35])))
    n1.ref_base[rng.random(n) < 0.02] = 4  # 'N'
    for b in (t1, n2, t2):
        b.ref_base[:] = n1.ref_base
    forced = (rng.random(n) < 0.3).astype(np.uint8)
    return n1, t1, n2, t2, forced


CASES = [dict(tier2=True, forced=True, nonsom=False), dict(tier2=True, forced=False, nonsom=True),
         dict(tier2=False, forced=True, nonsom=False), dict(tier2=True, forced=False, nonsom=False)]


def run(fn, sc, case, **kw):
    n1, t1, n2, t2, forced = sc
    return fn(n1, t1, n2 if case["tier2"] else None, t2 if case["tier2"] else None,
              is_forced_output=forced if case["forced"] else None, is_compute_nonsomatic=case["nonsom"], **kw)


def same(a, b, fields=FIELDS):
    for f in fields:
        x, y = a[f], b[f]
        if x.dtype.kind == "f":
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64)), f
        else:
            assert np.array_equal(x, y), (f, np.flatnonzero(x != y)[:5])


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_restatement_equals_golden(ci):
    g = np.load(GOLDEN)
    sc = scenario(1000 + ci)
    got = run(pyoracle.somatic_snv_call_tiers, sc, CASES[ci])
    want = g["case%d" % ci]
    same(got, want)
    assert (want["qphred"] > 0).sum() > 10
    if CASES[ci]["tier2"]:
        assert (want["snv_tier"] == 1).sum() > 0
    if CASES[ci]["tier2"] and CASES[ci]["forced"]:
        assert (want["ntype"] == 3).sum() > 0


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_restatement_equals_live_reference(ci):
    sc = scenario(2000 + ci, n=400)
    same(run(pyoracle.somatic_snv_call_tiers, sc, CASES[ci]),
         run(pyoracle.somatic_snv_call_tiers, sc, CASES[ci], use_reference=True))


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_gpu_equals_restatement_bytes(ci):
    capi.init(0)
    assert capi.lib().sk_libm_restated() == 1
    for seed in (3000 + ci, 3100 + ci):
        sc = scenario(seed, n=3000)
        got = run(capi.somatic_snv_call_tiers, sc, CASES[ci])
        want = run(pyoracle.somatic_snv_call_tiers, sc, CASES[ci])
        assert got.tobytes() == want.tobytes()


@pytest.mark.gpu
def test_gpu_equals_golden_reference():
    capi.init(0)
    g = np.load(GOLDEN)
    for ci, case in enumerate(CASES):
        same(run(capi.somatic_snv_call_tiers, scenario(1000 + ci), case), g["case%d" % ci])


# ---- a14 complete: the whole of somatic_indel_caller_grid::get_somatic_indel (somatic_indel_grid.cpp:181-361) -----------------

INDEL_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "somatic_indel_tiers_reference.npz")


def indel_cases(seed, n=500):
    """seeded cases; the tumour sample's indelToRef error rate of every case is the one the reference's own error model
    attached to the key (recorded in the golden file), so restatement and kernels get the reference's value"""
    cases = synth.somatic_indel_cases(n, np.random.Generator(np.random.PCG64(seed)))  # (golden inputs: never shifted)
    g = np.load(INDEL_GOLDEN)
    if "err%d" % seed in g:
        for c, e in zip(cases, g["err%d" % seed]):
            c["indel_to_ref_error_prob"] = float(e)
    return cases


INDEL_SEEDS = (41, 42)


@pytest.mark.parametrize("seed", INDEL_SEEDS)
def test_indel_restatement_equals_golden(seed):
    g = np.load(INDEL_GOLDEN)
    got, _ = pyoracle.get_somatic_indel(indel_cases(seed))
    want = g["rec%d" % seed]
    assert got.tobytes() == want.tobytes()
    assert (want["qphred"] > 0).sum() > 100 and (want["sindel_tier"] == 1).sum() > 5 and (want["is_forced_output"] == 1).sum() > 20
    assert (want["is_overlap"] == 1).sum() + (g["rec%d" % INDEL_SEEDS[0]]["is_overlap"] == 1).sum() > 0


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")
def test_indel_restatement_equals_live_reference():
    cases = synth.somatic_indel_cases(300, np.random.default_rng(77))
    want, used = pyoracle.get_somatic_indel(cases, use_reference=True)
    for c, e in zip(cases, used):
        c["indel_to_ref_error_prob"] = float(e)
    got, _ = pyoracle.get_somatic_indel(cases)
    assert got.tobytes() == want.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", INDEL_SEEDS)
def test_indel_gpu_equals_golden_reference_bytes(seed):
    capi.init(0)
    assert capi.lib().sk_libm_restated() == 1
    g = np.load(INDEL_GOLDEN)
    got = capi.somatic_indel_call_tiers(indel_cases(seed))
    assert got.tobytes() == g["rec%d" % seed].tobytes()


@pytest.mark.gpu
def test_indel_gpu_equals_restatement_fresh_seeds():
    capi.init(0)
    for seed in (901, 902, 903):
        cases = synth.somatic_indel_cases(2000, np.random.default_rng(seed))
        want, _ = pyoracle.get_somatic_indel(cases)
        got = capi.somatic_indel_call_tiers(cases)
        assert got.tobytes() == want.tobytes(), seed
