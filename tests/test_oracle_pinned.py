"""CPU suite: the oracle restatement (oracle/strelka_oracle.c) is pinned to the REFERENCE ITSELF:
  * tests/golden/*.npz|*.pkl hold outputs of the reference's own translation units (oracle/_ref, built by
    oracle/Makefile from /root/reference) on seeded inputs -- see tests/golden/make_golden.py;
  * when oracle/_ref/libstrelka_ref.so is present (build container) the same comparisons also run live on fresh inputs.
Everything is BIT-EXACT: the restatement and the reference run the same libm on the same host.
The reference's own unit-test values for the scalar helpers (L/blt_util/test/qscore_test.cpp:29-42,
logSumUtilTest.cpp:39-66) are checked as known-answer tests."""
import ctypes as C
import math
import os
import pickle

import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import capi, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
vp = C.c_void_p


def _p(a):
    return a.ctypes.data_as(vp)


@pytest.fixture(scope="module")
def gold(built):
    return np.load(os.path.join(GOLD, "pathb_reference.npz"), allow_pickle=True)


@pytest.fixture(scope="module")
def O(built):
    return pyoracle.oracle()


def test_reference_kats_scalar_helpers(O):
    # L/blt_util/test/qscore_test.cpp:29-42
    assert O.sko_error_prob_to_qphred(C.c_double(0.1)) == 10
    assert O.sko_error_prob_to_qphred(C.c_double(0.01)) == 20
    assert O.sko_error_prob_to_qphred(C.c_double(0.001)) == 30
    assert O.sko_ln_error_prob_to_qphred_f(C.c_float(math.log(0.1))) == 10
    assert O.sko_ln_error_prob_to_qphred_f(C.c_float(math.log(0.001))) == 30
    # L/blt_util/test/logSumUtilTest.cpp:39-66: getLogSum vs log(x1+x2) to 1e-5 %
    for x1, x2 in ((0.5, 0.2), (0.00001, 0.00000001), (1.0, 1.0), (0.999, 1e-10)):
        got = O.sko_log_sum2(C.c_double(math.log(x1)), C.c_double(math.log(x2)))
        assert got == pytest.approx(math.log(x1 + x2), rel=1e-7)


def test_tables_and_scalars_match_reference(gold, O):
    tabs = [np.zeros(71) for _ in range(3)]
    O.sko_get_qscore_tables(*[_p(t) for t in tabs])
    for name, t in zip(("q2p", "q2lncompe", "q2lne"), tabs):
        assert np.array_equal(t.view(np.uint64), gold[name].view(np.uint64)), name
    # the product builds the same tables with the same expressions
    for name, t in zip(("q2p", "q2lncompe", "q2lne"), capi.qscore_tables()):
        assert np.array_equal(t.view(np.uint64), gold[name].view(np.uint64)), name
    got = np.array([O.sko_error_prob_to_qphred(C.c_double(p)) for p in gold["qphred_in"]], np.int32)
    assert np.array_equal(got, gold["qphred_out"])
    got = np.array([O.sko_ln_error_prob_to_qphred_f(C.c_float(x)) for x in gold["lnqphred_in"]], np.int32)
    assert np.array_equal(got, gold["lnqphred_out"])
    got = np.array([O.sko_log_sum2(C.c_double(a), C.c_double(b)) for a, b in gold["logsum_in"]])
    assert np.array_equal(got.view(np.uint64), gold["logsum_out"].view(np.uint64))
    got = np.array([O.sko_log_sum2f(C.c_float(a), C.c_float(b)) for a, b in gold["logsum_in"].astype(np.float32)], np.float32)
    assert np.array_equal(got.view(np.uint32), gold["logsumf_out"].view(np.uint32))
    pri = np.zeros(200, np.float32)
    O.sko_germline_lnpriors(C.c_double(0.001), _p(pri))
    assert np.array_equal(pri.view(np.uint32), gold["germline_lnpriors_theta0.001"].view(np.uint32))


def test_std_sort_tie_order_matches_reference(gold, O):
    for key, perm in zip(gold["sort_keys"], gold["sort_perms"]):
        n = len(perm)
        idx = np.arange(n, dtype=np.uint32)
        k = np.ascontiguousarray(key if n else np.zeros(1), np.uint16)
        O.sko_sort_idx_by_key_desc(_p(idx), n, _p(k))
        assert np.array_equal(idx, perm), n


def _golden_pileups(gold):
    pb = capi.HostPileupBatch(gold["g_call_off"], gold["g_calls"], gold["g_ref_base"])
    return pb, gold["g_ploidy"]


def test_germline_matches_reference(gold):
    pb, ploidy = _golden_pileups(gold)
    de = pyoracle.adjust_joint_eprob(pb)
    assert np.array_equal(de.view(np.uint32), gold["g_de"].view(np.uint32))
    # per locus on the cleaned pileup, as the fixture was generated
    want = gold["g_digt"].copy().view(pyoracle.DIGT_CALL_DTYPE).reshape(-1)
    opt = pyoracle.germline_options()
    for l in range(pb.n_loci):
        s, e = int(pb.call_off[l]), int(pb.call_off[l + 1])
        c, d = pb.calls[s:e], de[s:e]
        keep = ((c >> 12) & 1) == 0
        one = capi.HostPileupBatch(np.array([0, int(keep.sum())]), c[keep], pb.ref_base[l:l + 1], ploidy=ploidy[l:l + 1])
        got = pyoracle.site_digt_call(one, d[keep], opt)
        assert got.tobytes() == want[l:l + 1].tobytes(), l


def test_somatic_snv_matches_reference(gold, O):
    n = capi.HostPileupBatch(gold["s_n_off"], gold["s_n_calls"], gold["s_ref_base"])
    t = capi.HostPileupBatch(gold["s_t_off"], gold["s_t_calls"], gold["s_ref_base"])
    lnp3 = np.ascontiguousarray(gold["s_lnprior3"])
    for l in range(n.n_loci):
        for b, want, strand in ((n, gold["s_normal_lhood"], 0), (t, gold["s_tumor_lhood"], 1)):
            s, e = int(b.call_off[l]), int(b.call_off[l + 1])
            c = np.ascontiguousarray(b.calls[s:e])
            row = np.zeros(30, np.float32)
            O.sko_somatic_sample_lhood(_p(c), e - s, int(n.ref_base[l]), strand, _p(row))
            assert np.array_equal(row.view(np.uint32), want[l].view(np.uint32)), l
        mg, q, fq, nt = C.c_uint32(), C.c_int32(), C.c_int32(), C.c_uint32()
        nl, tl = np.ascontiguousarray(gold["s_normal_lhood"][l]), np.ascontiguousarray(gold["s_tumor_lhood"][l])
        O.sko_calculate_result_set_grid(C.c_float(0.15), C.c_float(math.log(5e-10)), C.c_float(math.log1p(-5e-10)), _p(nl),
                                        _p(tl), _p(lnp3), C.c_float(math.log1p(-1e-4)), C.c_float(math.log(1e-4)),
                                        C.byref(mg), C.byref(q), C.byref(fq), C.byref(nt))
        assert (mg.value, q.value, fq.value, nt.value) == tuple(gold["s_result"][l]), l
    assert (gold["s_result"][:, 1] > 0).sum() > 5


def test_indel_likelihoods_match_reference(gold):
    rb = capi.HostReadScoreBatch(gold["i_read_off"], gold["i_ref"], gold["i_indel"], gold["i_alt"], gold["i_na"], gold["i_rl"],
                                 gold["i_flags"], gold["i_del"], gold["i_ins"])
    for t2 in (0, 1):
        got = pyoracle.indel_grid_lhood(rb, 5, 0.25 if t2 else 0.5, bool(t2))
        assert np.array_equal(got.view(np.uint64), gold["i_grid"][t2].view(np.uint64))
    ab = capi.HostAlleleGroupBatch(gold["a_read_off"], gold["a_n_alt"], gold["a_ploidy"], gold["a_del"], gold["a_ins"],
                                   gold["a_ref"], gold["a_allele"], gold["a_na"], gold["a_rl"], gold["a_flags"])
    lh, counts, _ = pyoracle.allele_group_genotype_lhoods(ab)
    assert np.array_equal(lh.view(np.uint64), gold["a_lhood"].view(np.uint64))
    assert np.array_equal(counts, gold["a_counts"])


def test_wide_allele_groups_match_reference(built):
    """allele groups of 4..8 alternate alleles (multi-sample runs): the restatement against the reference's own
    getVariantAlleleGroupGenotypeLhoodsForSample -- the committed fixture, and live on fresh groups where oracle/_ref is built"""
    g = np.load(os.path.join(GOLD, "allele_group_wide_reference.npz"))
    ab = capi.HostAlleleGroupBatch(g["a_read_off"], g["a_n_alt"], g["a_ploidy"], g["a_del"], g["a_ins"], g["a_ref"], g["a_allele"],
                                   g["a_na"], g["a_rl"], g["a_flags"], width=capi.MAX_ALT_WIDE)
    assert ab.n_alt.min() == 4 and ab.n_alt.max() == 8
    lh, counts, ng = pyoracle.allele_group_genotype_lhoods(ab)
    assert ng.max() == 45 and np.array_equal(lh.view(np.uint64), g["a_lhood"].view(np.uint64)) and np.array_equal(counts, g["a_counts"])
    assert int(counts.sum()) > 2000  # reads were used
    if pyoracle.ref_available():
        from tests.golden import make_golden_wide_groups as W
        fresh = synth.allele_group_batch(40, np.random.default_rng(99), depth_mean=30.0, min_alt=4, max_alt=capi.MAX_ALT_WIDE, missing_rate=0.02)
        want_lh, want_counts = W.reference_lhoods(fresh)
        lh, counts, _ = pyoracle.allele_group_genotype_lhoods(fresh)
        assert np.array_equal(lh.view(np.uint64), want_lh.view(np.uint64)) and np.array_equal(counts, want_counts)


def test_xwide_allele_groups_match_reference(built):
    """... and of 9..16 alternate alleles (runs of five to eight samples, up to 153 genotypes): the restatement against the reference's own
    function -- the committed fixture (tests/golden/make_golden_wide_groups.py xwide), and live on fresh groups where oracle/_ref is built"""
    g = np.load(os.path.join(GOLD, "allele_group_xwide_reference.npz"))
    ab = capi.HostAlleleGroupBatch(g["a_read_off"], g["a_n_alt"], g["a_ploidy"], g["a_del"], g["a_ins"], g["a_ref"], g["a_allele"],
                                   g["a_na"], g["a_rl"], g["a_flags"], width=capi.MAX_ALT_XWIDE)
    assert ab.n_alt.min() == 9 and ab.n_alt.max() == 16
    lh, counts, ng = pyoracle.allele_group_genotype_lhoods(ab)
    assert ng.max() == 153 and np.array_equal(lh.view(np.uint64), g["a_lhood"].view(np.uint64)) and np.array_equal(counts, g["a_counts"])
    assert int(counts.sum()) > 500  # reads were used
    if pyoracle.ref_available():
        from tests.golden import make_golden_wide_groups as W
        fresh = synth.allele_group_batch(12, np.random.default_rng(98), depth_mean=30.0, min_alt=9, max_alt=capi.MAX_ALT_XWIDE, missing_rate=0.02)
        want_lh, want_counts = W.reference_lhoods(fresh)
        lh, counts, _ = pyoracle.allele_group_genotype_lhoods(fresh)
        assert np.array_equal(lh.view(np.uint64), want_lh.view(np.uint64)) and np.array_equal(counts, want_counts)


def test_alignment_scores_match_reference(built):
    with open(os.path.join(GOLD, "patha_scores_reference.pkl"), "rb") as f:
        g = pickle.load(f)
    got = pyoracle.score_cases(g["cases"])
    assert np.array_equal(got.view(np.uint64), g["scores"].view(np.uint64))
    # and the product's host adapter + a test-only op interpreter reach the same doubles from the same cases
    from tests.flat_interp import score_flat
    batch = synth.build_align_batch(g["cases"][:16])
    _, lnc, lne = capi.qscore_tables()
    n = batch.n_cals
    assert np.array_equal(score_flat(batch, lnc, lne).view(np.uint64), g["scores"][:n].view(np.uint64))


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built (no /root/reference here)")
def test_live_against_reference_build(built):
    """Fresh seeds every run of the build container: restatement vs the reference's own code."""
    R = pyoracle.ref()
    rng = np.random.default_rng()
    cases = synth.align_cases(30, rng)
    assert np.array_equal(pyoracle.score_cases(cases).view(np.uint64), pyoracle.ref_score_cases(cases).view(np.uint64))
    pb = synth.pileups(200, rng, het_rate=0.1, noise=0.1, nmm_rate=0.1, filter_rate=0.02)
    de = pyoracle.adjust_joint_eprob(pb)
    for l in range(pb.n_loci):
        s, e = int(pb.call_off[l]), int(pb.call_off[l + 1])
        c = np.ascontiguousarray(pb.calls[s:e])
        d = np.zeros(e - s, np.float32)
        R.ref_adjust_joint_eprob(_p(c), e - s, C.c_double(.35), C.c_double(.6), 1, C.c_double(.25), _p(d))
        assert np.array_equal(d.view(np.uint32), de[s:e].view(np.uint32))
    assert R.ref_sizeof_base_call() == 2
    for _ in range(200):
        f = [int(rng.integers(0, m)) for m in (64, 4, 2, 2, 2, 2)]
        assert R.ref_pack_base_call(*f) == int(capi.make_call(*f))
