"""GPU suite (-m gpu): the HIP path, called through the C-ABI, against the oracle on the same seeded inputs.

Bars (BASELINE.json north_star asks for bit-exact integer work and 1e-5 on log-likelihoods; the suite demands more):
  * hot path A scores: BIT-EXACT doubles (every term comes from host-built tables; adds are sequential, no FMA)
  * hot path B, SNVs: germline and somatic result records BYTE for BYTE -- table-driven float32 sums in call order, and
    the reference's powf / logf / expf / log1pf / exp / log / log10 calls evaluated with restatements of the host libm's
    routines (csrc/libm_flt32.h, libm_dbl64.h; sk_libm_restated() must be 1 on this image)
  * hot path B, indels: likelihood doubles BIT-EXACT (reference operation order + restated exp / log / log1p), supporting
    read counts and Q-scores identical
  * LL_TOL remains only for one edge-case comparison below
"""
import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import synth

pytestmark = pytest.mark.gpu

LL_TOL = 1e-5  # float likelihoods that end in a device transcendental (somatic strand states, strand bias)


def close_ll(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    scale = np.maximum(1.0, np.abs(b))
    with np.errstate(invalid="ignore"):
        ok = np.abs(a - b) <= LL_TOL * scale
    return bool(np.all(ok | both_inf))


# ---------------------------------------------------------------------------------------------------------- hot path A

def test_scores_bit_exact_random_cases(gpu):
    rng = np.random.default_rng(101)
    cases = synth.align_cases(300, rng)
    got = gpu.score_alignments(synth.build_align_batch(cases))
    want = pyoracle.score_cases(cases)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_scores_bit_exact_h64_workload(gpu):
    rng = np.random.default_rng(102)
    cases = synth.align_cases_h64(200, rng)
    got = gpu.score_alignments(synth.build_align_batch(cases))
    want = pyoracle.score_cases(cases)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_more_than_64_candidates_and_long_reads(gpu):
    rng = np.random.default_rng(103)
    cases = synth.align_cases(6, rng, L=150, K=7, max_cals=128)          # > one wave of candidates per read
    cases += synth.align_cases(4, rng, L=2000, K=6, win=5000, max_cals=20)  # too long for the LDS kernel -> generic
    got = gpu.score_alignments(synth.build_align_batch(cases))
    want = pyoracle.score_cases(cases)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_fast_and_generic_kernels_agree_at_scale(gpu):
    """Size-independent property at bench scale: the LDS wave-per-read kernel and the thread-per-candidate kernel are two
    independent implementations; their outputs over 2^21 pairs must be identical bit for bit, and tiling a batch must
    reproduce the untiled scores."""
    import torch
    from strelka_amd import device
    rng = np.random.default_rng(104)
    hb = synth.align_batch_flat(1 << 13, rng)
    d1 = device.DeviceAlignBatch(hb, "cuda:0", tile=4)
    fast = d1.score().clone()
    gen = d1.score(generic=True).clone()
    torch.cuda.synchronize()
    assert torch.equal(fast.view(torch.int64), gen.view(torch.int64))
    n = hb.n_cals
    assert torch.equal(fast[:n].view(torch.int64), fast[3 * n:].view(torch.int64))
    small = gpu.score_alignments(hb)
    assert np.array_equal(small.view(np.uint64), fast[:n].cpu().numpy().view(np.uint64))


def test_column_and_entry_kernels_agree(gpu):
    """the streaming kernel (column form of the batch) and the transition-entry kernel are two implementations of the same sum:
    identical bits on the h64 workload (dense transitions, non-candidate penalties) and on soft-clipped / clipped random cases"""
    import torch
    from strelka_amd import device
    rng = np.random.default_rng(106)
    for cases in (synth.align_cases_h64(300, rng), synth.align_cases(400, rng), synth.align_cases(12, rng, L=150, K=7, max_cals=128)):
        hb = synth.build_align_batch(cases)
        want = pyoracle.score_cases(cases)
        for columns in (True, False):
            hb.colmat = hb.colmat_off = hb.addmask = None
            d = device.DeviceAlignBatch(hb, "cuda:0", columns=columns)
            got = d.score().cpu().numpy()
            torch.cuda.synchronize()
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), columns


def test_flat_workload_matches_interpreter(gpu):
    from tests.flat_interp import score_flat
    rng = np.random.default_rng(105)
    hb = synth.align_batch_flat(64, rng)
    _, lnc, lne = gpu.qscore_tables()
    want = score_flat(hb, lnc, lne)
    got = gpu.score_alignments(hb)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_empty_batches(gpu):
    from strelka_amd import capi
    b = capi.AlignBuilder()
    assert len(gpu.score_alignments(b.finish())) == 0
    b.add_read(np.zeros(0, np.uint8), np.zeros(0, np.uint8), "ACGT", 0, [])
    assert len(gpu.score_alignments(b.finish())) == 0


def test_invalid_quality_rejected(gpu):
    from strelka_amd import capi
    hb = synth.align_batch_flat(4, np.random.default_rng(1))
    hb.read_qual[3] = 71
    with pytest.raises(capi.StrelkaAmdError):
        gpu.score_alignments(hb)


# ---------------------------------------------------------------------------------------------------------- hot path B

def _varied_pileups(rng, n=4000):
    parts = [synth.pileups(n, rng, het_rate=0.05, hom_rate=0.03, filter_rate=0.03),
             synth.pileups(n // 4, rng, depth_mean=3.0, het_rate=0.2),
             synth.pileups(n // 8, rng, depth_mean=400.0, het_rate=0.1, nmm_rate=0.2),
             synth.pileups(n // 2, rng, noise=0.6, nmm_rate=0.1),
             synth.pileups(50, rng, depth_mean=0.2)]
    off = [np.zeros(1, np.int64)]
    calls, ref = [], []
    base = 0
    for p in parts:
        off.append(p.call_off[1:] + base)
        base += p.call_off[-1]
        calls.append(p.calls)
        ref.append(p.ref_base)
    from strelka_amd import capi
    return capi.HostPileupBatch(np.concatenate(off), np.concatenate(calls), np.concatenate(ref))


def test_dependent_eprob(gpu):
    rng = np.random.default_rng(201)
    pb = _varied_pileups(rng)
    got = gpu.dependent_eprob(pb)
    want = pyoracle.adjust_joint_eprob(pb)
    # powf is evaluated with the restatement of the host libm's routine (csrc/libm_flt32.h; sk_init checks the host libm is
    # that implementation), and the sort emulation fixes WHICH call gets which exponent: bit for bit
    assert gpu.lib().sk_libm_restated() == 1
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_site_digt_call(gpu):
    rng = np.random.default_rng(202)
    pb = _varied_pileups(rng)
    pb.de = pyoracle.adjust_joint_eprob(pb)
    pb.ploidy = rng.choice(np.array([1, 2, 2, 2], np.uint8), pb.n_loci)
    pb.ref_base[::97] = 4  # some 'N' reference bases
    got = gpu.site_digt_call(pb)
    want = pyoracle.site_digt_call(pb, pb.de)
    assert np.array_equal(got["is_called"], want["is_called"])
    assert np.array_equal(got["ref_gt"], want["ref_gt"])
    # every term of the float32 sums is a host-built table value or logf_glibc(de) + ln(1/3): bit for bit
    assert np.array_equal(got["lhood"].view(np.uint32), want["lhood"].view(np.uint32))
    assert np.array_equal(got["phredLoghood"], want["phredLoghood"])
    assert np.array_equal(got["strand_bias"].view(np.uint64), want["strand_bias"].view(np.uint64))
    # the double-precision posterior uses exp / log10 restated from the host libm (csrc/libm_dbl64.h): the whole record,
    # doubles included, is the oracle's byte for byte
    for rs in ("genome", "poly"):
        for f in ("max_gt", "snp_qphred", "max_gt_qphred"):
            assert np.array_equal(got[rs][f], want[rs][f]), (rs, f)
        assert np.array_equal(got[rs]["ref_pprob"].view(np.uint64), want[rs]["ref_pprob"].view(np.uint64)), rs
    assert got.tobytes() == want.tobytes()


def test_site_digt_call_exact_when_de_is_tabulated(gpu):
    """With de[i] = 0.5 (logf exact-ish input shared by both sides is not enough) -- use de values whose logf is the same
    on both sides: powers of two.  Then every term is table- or exactly-computed and the float32 sums must be bit-exact."""
    rng = np.random.default_rng(203)
    pb = synth.pileups(3000, rng, het_rate=0.1, hom_rate=0.05)
    pb.de = np.full(len(pb.calls), 0.25, np.float32)
    got = gpu.site_digt_call(pb)
    want = pyoracle.site_digt_call(pb, pb.de)
    assert np.array_equal(got["lhood"].view(np.uint32), want["lhood"].view(np.uint32))
    assert np.array_equal(got["phredLoghood"], want["phredLoghood"])


def test_site_digt_call_fused_equals_two_step(gpu):
    """The fused LDS kernel (a9+a10 in one pass, de never materialised) must reproduce the two separate kernels bit for
    bit: same arithmetic, different data movement.  Includes loci deeper than the LDS path allows (>1023 calls) and a
    span larger than one LDS sub-batch."""
    rng = np.random.default_rng(207)
    parts = [_varied_pileups(rng), synth.pileups(40, rng, depth_mean=1500.0, het_rate=0.2),
             synth.pileups(300, rng, depth_mean=150.0, het_rate=0.1), synth.pileups(1, rng, depth_mean=9000.0)]
    off = [np.zeros(1, np.int64)]
    calls, ref = [], []
    base = 0
    for p in parts:
        off.append(p.call_off[1:] + base)
        base += p.call_off[-1]
        calls.append(p.calls)
        ref.append(p.ref_base)
    pb = gpu.HostPileupBatch(np.concatenate(off), np.concatenate(calls), np.concatenate(ref))
    pb.ploidy = rng.choice(np.array([1, 2, 2], np.uint8), pb.n_loci)
    pb.ref_base[::53] = 4
    de2 = gpu.dependent_eprob(pb)
    pb.de = de2
    two = gpu.site_digt_call(pb)
    fused, de1 = gpu.site_digt_call_fused(pb, want_de=True)
    assert np.array_equal(de1.view(np.uint32), de2.view(np.uint32))
    assert fused.tobytes() == two.tobytes()
    fused_no_de, none = gpu.site_digt_call_fused(pb)
    assert none is None and fused_no_de.tobytes() == two.tobytes()
    # and against the oracle
    want_de = pyoracle.adjust_joint_eprob(pb)
    assert np.array_equal(de1.view(np.uint32), want_de.view(np.uint32))
    want = pyoracle.site_digt_call(pb, want_de)
    assert np.array_equal(fused["lhood"].view(np.uint32), want["lhood"].view(np.uint32))
    assert fused.tobytes() == want.tobytes()


def test_site_digt_call_fused_nondefault_options(gpu):
    rng = np.random.default_rng(208)
    pb = synth.pileups(3000, rng, het_rate=0.1, nmm_rate=0.3)
    for kw in (dict(bsnp_ssd_no_mismatch=0.05, bsnp_ssd_one_mismatch=0.1), dict(is_min_vexp=0),
               dict(bsnp_ssd_no_mismatch=0.0, bsnp_ssd_one_mismatch=0.0), dict(min_vexp=0.6, bsnp_diploid_theta=0.01)):
        opt = gpu.germline_options()
        oopt = pyoracle.germline_options()
        for k, v in kw.items():
            setattr(opt, k, v)
            setattr(oopt, k, v)
        de2 = gpu.dependent_eprob(pb, opt)
        pb.de = de2
        two = gpu.site_digt_call(pb, opt)
        fused, de1 = gpu.site_digt_call_fused(pb, opt, want_de=True)
        assert np.array_equal(de1.view(np.uint32), de2.view(np.uint32)), kw
        assert fused.tobytes() == two.tobytes(), kw
        want_de = pyoracle.adjust_joint_eprob(pb, oopt)
        assert np.array_equal(de1.view(np.uint32), want_de.view(np.uint32)), kw
        want = pyoracle.site_digt_call(pb, want_de, oopt)
        assert np.array_equal(fused["lhood"].view(np.uint32), want["lhood"].view(np.uint32)), kw


def test_gvcf_site_summaries(gpu):
    """gvcf_site_summary_kernel (csrc/gvcf_site_core.h: would process_pos_snp_digt build a homozygous-reference locus with no alternate
    allele here?  its GQX, its reference AD counts) against the numpy statement of the reference's getSiteAltAlleles / setGqx / AD loop
    (oracle/pyoracle.py), on ordinary 40x loci, het and hom-alt sites, noisy and shallow columns, haploid and N-reference loci"""
    rng = np.random.default_rng(2290)
    pb = _varied_pileups(rng)
    pb.ploidy = rng.choice(np.array([1, 2, 2, 2], np.uint8), pb.n_loci)
    pb.ref_base[::41] = 4
    calls, _ = gpu.site_digt_call_fused(pb)
    got = gpu.gvcf_site_summaries(pb, calls)
    want = pyoracle.gvcf_site_summaries(pb, calls)
    assert got.tobytes() == want.tobytes()
    plain = (got["flags"] & 1) != 0
    assert 0.3 < plain.mean() < 0.95 and (~plain).sum() > 500  # (the batch holds plenty of both kinds)
    # a plain site: diploid, reference known, covered, both most likely genotypes the homozygous-reference one
    assert (pb.ploidy[plain] == 2).all() and (pb.ref_base[plain] < 4).all()
    assert (calls["poly"]["max_gt"][plain] == pb.ref_base[plain]).all() and (calls["genome"]["max_gt"][plain] == pb.ref_base[plain]).all()
    # ... and on a 40x batch with human variant density nearly every position is one
    pw = synth.pileups(20000, rng)
    cw, _ = gpu.site_digt_call_fused(pw)
    sw = gpu.gvcf_site_summaries(pw, cw)
    assert sw.tobytes() == pyoracle.gvcf_site_summaries(pw, cw).tobytes()
    assert ((sw["flags"] & 1) != 0).mean() > 0.97


@pytest.mark.parametrize("variant", [0, 1])
def test_site_digt_call_fused_every_kernel_variant(gpu, variant):
    """csrc/germline_fused.hip holds round 1's kernel (0) and the second statement (1: tabled + pending ranked terms, calls read
    ahead): both against the oracle, byte for byte -- ordinary 40x loci with and
    without neighbouring mismatches, dense mismatch flags (every group pending: the list overflows into the global-memory pass), low
    quality calls, deep loci, loci beyond the LDS path, haploid and N-reference loci, non-default options"""
    from strelka_amd import capi
    rng = np.random.default_rng(2090 + variant)
    capi.lib().sk_debug_set_g3_variant(variant)
    try:
        parts = [synth.pileups(3000, rng, het_rate=0.05, hom_rate=0.02), synth.pileups(1500, rng, het_rate=0.1, nmm_rate=0.6),
                 synth.pileups(700, rng, nmm_rate=0.0), _varied_pileups(rng), synth.pileups(30, rng, depth_mean=1500.0, het_rate=0.2),
                 synth.pileups(300, rng, depth_mean=150.0, het_rate=0.1), synth.pileups(600, rng, depth_mean=3.0),
                 synth.pileups(1, rng, depth_mean=9000.0), synth.pileups(513, rng, depth_mean=70.0, nmm_rate=0.1)]
        off = [np.zeros(1, np.int64)]
        calls, ref = [], []
        base = 0
        for p in parts:
            off.append(p.call_off[1:] + base)
            base += p.call_off[-1]
            calls.append(p.calls)
            ref.append(p.ref_base)
        pb = gpu.HostPileupBatch(np.concatenate(off), np.concatenate(calls), np.concatenate(ref))
        pb.ploidy = rng.choice(np.array([1, 2, 2], np.uint8), pb.n_loci)
        pb.ref_base[::53] = 4
        for kw in ({}, dict(bsnp_ssd_no_mismatch=0.05, bsnp_ssd_one_mismatch=0.1), dict(is_min_vexp=0), dict(min_vexp=0.6, bsnp_diploid_theta=0.01)):
            opt = gpu.germline_options()
            oopt = pyoracle.germline_options()
            for k, v in kw.items():
                setattr(opt, k, v)
                setattr(oopt, k, v)
            fused, de1 = gpu.site_digt_call_fused(pb, opt, want_de=True)
            want_de = pyoracle.adjust_joint_eprob(pb, oopt)
            assert np.array_equal(de1.view(np.uint32), want_de.view(np.uint32)), (variant, kw)
            want = pyoracle.site_digt_call(pb, want_de, oopt)
            assert np.array_equal(fused["lhood"].view(np.uint32), want["lhood"].view(np.uint32)), (variant, kw)
            assert fused.tobytes() == want.tobytes(), (variant, kw)
            again, none = gpu.site_digt_call_fused(pb, opt)
            assert none is None and again.tobytes() == want.tobytes(), (variant, kw)
    finally:
        capi.lib().sk_debug_set_g3_variant(-1)


def test_somatic_snv(gpu):
    rng = np.random.default_rng(204)
    n, t = synth.somatic_pileups(6000, rng, somatic_rate=0.03, het_rate=0.03)
    for forced in (False, True):
        got = gpu.somatic_snv_call(n, t, is_forced_output=forced)
        want = pyoracle.somatic_snv_call(n, t, is_forced_output=forced)
        assert np.array_equal(got["is_called"], want["is_called"])
        # table-driven float sums: bit exact (21 prestrand states of both samples)
        assert np.array_equal(got["normal_lhood"][:, :21].view(np.uint32), want["normal_lhood"][:, :21].view(np.uint32))
        assert np.array_equal(got["tumor_lhood"][:, :21].view(np.uint32), want["tumor_lhood"][:, :21].view(np.uint32))
        # 9 strand states end in a float logsum: expf / log1pf / logf restated from the host libm (csrc/libm_flt32.h)
        assert np.array_equal(got["tumor_lhood"][:, 21:].view(np.uint32), want["tumor_lhood"][:, 21:].view(np.uint32))
        assert np.array_equal(got["normal_alt_id"], want["normal_alt_id"])
        assert np.array_equal(got["tumor_alt_id"], want["tumor_alt_id"])
        assert np.array_equal(got["max_gt"], want["max_gt"])
        assert np.array_equal(got["ntype"], want["ntype"])
        assert np.array_equal(got["qphred"], want["qphred"])
        assert np.array_equal(got["from_ntype_qphred"], want["from_ntype_qphred"])
        assert np.array_equal(got["strand_bias"].view(np.uint32), want["strand_bias"].view(np.uint32))
        assert got.tobytes() == want.tobytes()  # every field of every record


def test_somatic_snv_skipped_deep_and_empty_loci(gpu):
    """the queue of non-skipped loci: all-reference loci (the reference's early return), 'N' reference, empty pileups,
    one locus far deeper than the staging chunk, a locus count that is not a multiple of the block size"""
    from strelka_amd import capi
    rng = np.random.default_rng(205)
    n_loci = 777
    ref = rng.integers(0, 4, n_loci).astype(np.uint8)
    ref[rng.random(n_loci) < 0.05] = 4
    kind = rng.integers(0, 4, n_loci)  # 0: all-ref both, 1: noisy, 2: empty normal, 3: all-ref normal / variant tumor

    def sample(depth_mean, is_tumor):
        depth = rng.poisson(depth_mean, n_loci)
        depth[kind == 2] = 0 if not is_tumor else depth[kind == 2]
        depth[5] = 3000 if is_tumor else 1500  # >> CHUNK
        off = np.zeros(n_loci + 1, np.int64)
        np.cumsum(depth, out=off[1:])
        total = int(off[-1])
        locus = np.repeat(np.arange(n_loci), depth)
        r = np.minimum(ref[locus], 3)
        alt = (r + 1 + rng.integers(0, 3, total)) % 4
        p_alt = np.where(kind[locus] == 1, 0.05, np.where((kind[locus] == 3) & is_tumor, 0.3, 0.0))
        p_alt = np.where(locus == 5, 0.2, p_alt)
        base = np.where(rng.random(total) < p_alt, alt, r).astype(np.uint8)
        q = rng.choice(synth.QUAL_VALUES, total, p=synth.QUAL_PROBS)
        return capi.HostPileupBatch(off, capi.make_call(q, base, rng.integers(0, 2, total), 0, 0, 0), ref)

    n, t = sample(30, False), sample(60, True)
    for forced in (False, True):
        got = gpu.somatic_snv_call(n, t, is_forced_output=forced)
        want = pyoracle.somatic_snv_call(n, t, is_forced_output=forced)
        assert np.array_equal(got["is_called"], want["is_called"])
        assert 0.2 < want["is_called"].mean() < 1.0
        for f in ("normal_lhood", "tumor_lhood"):
            assert np.array_equal(got[f][:, :21].view(np.uint32), want[f][:, :21].view(np.uint32)), f
        assert np.array_equal(got["tumor_lhood"][:, 21:].view(np.uint32), want["tumor_lhood"][:, 21:].view(np.uint32))
        for f in ("normal_alt_id", "tumor_alt_id", "max_gt", "ntype"):
            assert np.array_equal(got[f], want[f]), f
        assert np.array_equal(got["qphred"], want["qphred"])
        assert np.array_equal(got["from_ntype_qphred"], want["from_ntype_qphred"])
        skipped = want["is_called"] == 0
        assert not got["normal_lhood"][skipped].any() and not got["qphred"][skipped].any()


def test_pileup_edge_cases(gpu):
    from strelka_amd import capi
    # empty loci, single-call loci, all-filtered loci, N reference
    calls = capi.make_call([30, 40, 2, 63, 20], [0, 1, 2, 3, 0], [1, 0, 1, 0, 1], 0, [0, 0, 0, 0, 1], 0)
    off = np.array([0, 0, 1, 3, 5, 5], np.int64)
    pb = capi.HostPileupBatch(off, calls, np.array([0, 0, 1, 4, 2], np.uint8))
    de = gpu.dependent_eprob(pb)
    want_de = pyoracle.adjust_joint_eprob(pb)
    assert np.array_equal(de.view(np.uint32), want_de.view(np.uint32))
    pb.de = want_de
    got = gpu.site_digt_call(pb)
    want = pyoracle.site_digt_call(pb, want_de)
    assert got.tobytes() == want.tobytes()


# ---------------------------------------------------------------------------------------------------- hot path B, indels



def test_indel_grid_lhood(gpu):
    rng = np.random.default_rng(301)
    rb = synth.readscore_batch(400, rng, depth_mean=110.0, breakpoint_rate=0.05)
    for tier2 in (False, True):
        opt = gpu.indel_options(True)
        got = gpu.indel_grid_lhood(rb, opt, tier2)
        want = pyoracle.indel_grid_lhood(rb, opt.min_read_bp_flank, 0.25 if tier2 else 0.5, tier2)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))  # reference order + restated exp/log/log1p
    # empty and single-read rows
    rb2 = synth.readscore_batch(50, rng, depth_mean=0.7)
    opt = gpu.indel_options(True)
    opt.min_read_bp_flank = 1
    got = gpu.indel_grid_lhood(rb2, opt, False)
    want = pyoracle.indel_grid_lhood(rb2, 1, 0.5, False)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_somatic_indel_call(gpu):
    rng = np.random.default_rng(302)
    n = 500
    normal = synth.readscore_batch(n, rng, depth_mean=40.0)
    tumor = synth.readscore_batch(n, rng, depth_mean=110.0)
    tumor.del_len, tumor.ins_len = normal.del_len, normal.ins_len
    # make a fraction of them look somatic / germline
    k = len(tumor.indel_lnp)
    boost = np.repeat(rng.random(n) < 0.4, np.diff(tumor.read_off))
    tumor.indel_lnp = np.where(boost & (rng.random(k) < 0.3), 0.0, tumor.indel_lnp).astype(np.float32)
    tumor.ref_lnp = np.where(boost, np.minimum(tumor.ref_lnp, -1.0), tumor.ref_lnp).astype(np.float32)
    err = rng.choice([5e-5, 1e-4, 3e-3, 2e-2], n)
    got = gpu.somatic_indel_call(normal, tumor, err)
    nl = pyoracle.indel_grid_lhood(normal, 1, 0.5, False)
    tl = pyoracle.indel_grid_lhood(tumor, 5, 0.5, False)
    assert np.array_equal(got["normal_lhood"].view(np.uint64), nl.view(np.uint64))
    assert np.array_equal(got["tumor_lhood"].view(np.uint64), tl.view(np.uint64))
    want = pyoracle.somatic_indel_result(nl, tl, err)
    assert np.array_equal(got["max_gt"], want["max_gt"])
    assert np.array_equal(got["ntype"], want["ntype"])
    assert np.array_equal(got["qphred"], want["qphred"])
    assert np.array_equal(got["from_ntype_qphred"], want["from_ntype_qphred"])
    assert (got["qphred"] > 0).sum() >= 1  # the test data does contain calls


def test_allele_group_genotype_lhoods(gpu):
    rng = np.random.default_rng(303)
    ab = synth.allele_group_batch(600, rng, depth_mean=45.0)
    got = gpu.allele_group_genotype_lhoods(ab)
    lh, counts, ng = pyoracle.allele_group_genotype_lhoods(ab)
    assert np.array_equal(got["n_genotypes"], ng)
    assert np.array_equal(got["lhood"].view(np.uint64), lh.view(np.uint64))
    assert np.array_equal(got["counts"], counts)
    # deep group (> one 64-read chunk) and an empty one
    ab2 = synth.allele_group_batch(8, rng, depth_mean=300.0)
    got2 = gpu.allele_group_genotype_lhoods(ab2)
    lh2, counts2, _ = pyoracle.allele_group_genotype_lhoods(ab2)
    assert np.array_equal(got2["lhood"].view(np.uint64), lh2.view(np.uint64))
    assert np.array_equal(got2["counts"], counts2)


def test_indel_kernels_with_one_read_length(gpu):
    """every read of an indel / allele group as long as the others (a WGS sample): the kernels evaluate the allele-ratio priors once per
    indel / genotype then, not per read -- the same bits; groups where a few reads differ take the per-read form"""
    from strelka_amd import capi
    rng = np.random.default_rng(305)
    rb = synth.readscore_batch(300, rng, depth_mean=110.0, breakpoint_rate=0.05)
    for length in (150, 36, 251):
        rb.read_length[:] = length
        rb.read_length[rb.read_off[7]:rb.read_off[9]] = rng.integers(8, 200, int(rb.read_off[9] - rb.read_off[7]))  # two mixed indels among them
        opt = gpu.indel_options(True)
        got = gpu.indel_grid_lhood(rb, opt, False)
        want = pyoracle.indel_grid_lhood(rb, opt.min_read_bp_flank, 0.5, False)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), length
    for max_alt in (capi.MAX_ALT, capi.MAX_ALT_WIDE):
        ab = synth.allele_group_batch(250, rng, depth_mean=45.0, min_alt=1, max_alt=max_alt)
        for length in (150, 40):
            ab.read_length[:] = length
            ab.read_length[ab.read_off[3]:ab.read_off[5]] = rng.integers(8, 200, int(ab.read_off[5] - ab.read_off[3]))
            got = gpu.allele_group_genotype_lhoods(ab)
            lh, counts, ng = pyoracle.allele_group_genotype_lhoods(ab)
            assert np.array_equal(got["n_genotypes"], ng)
            assert np.array_equal(got["lhood"].view(np.uint64), lh.view(np.uint64)), (max_alt, length)
            assert np.array_equal(got["counts"], counts)


def test_allele_group_genotype_lhoods_wide(gpu):
    """groups of 4..8 alternate alleles (a multi-sample run's): sk_allele_group_genotype_lhoods_wide against the oracle on fresh
    groups and against the REFERENCE's own function on the committed fixture (tests/golden/make_golden_wide_groups.py)"""
    import os
    from strelka_amd import capi
    rng = np.random.default_rng(304)
    ab = synth.allele_group_batch(300, rng, depth_mean=45.0, min_alt=1, max_alt=capi.MAX_ALT_WIDE, missing_rate=0.01)
    assert ab.width == capi.MAX_ALT_WIDE
    got = gpu.allele_group_genotype_lhoods(ab)
    lh, counts, ng = pyoracle.allele_group_genotype_lhoods(ab)
    assert got["lhood"].shape[1] == 45 and np.array_equal(got["n_genotypes"], ng) and ng.max() == 45
    assert np.array_equal(got["lhood"].view(np.uint64), lh.view(np.uint64))
    assert np.array_equal(got["counts"], counts)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "allele_group_wide_reference.npz"))
    gb = capi.HostAlleleGroupBatch(g["a_read_off"], g["a_n_alt"], g["a_ploidy"], g["a_del"], g["a_ins"], g["a_ref"], g["a_allele"],
                                   g["a_na"], g["a_rl"], g["a_flags"], width=capi.MAX_ALT_WIDE)
    got = gpu.allele_group_genotype_lhoods(gb)
    assert np.array_equal(got["lhood"].view(np.uint64), g["a_lhood"].view(np.uint64))
    assert np.array_equal(got["counts"], g["a_counts"])
    # deep groups (more than one 64-read chunk)
    ab2 = synth.allele_group_batch(6, rng, depth_mean=200.0, min_alt=5, max_alt=capi.MAX_ALT_WIDE, missing_rate=0.0)
    got2 = gpu.allele_group_genotype_lhoods(ab2)
    lh2, counts2, _ = pyoracle.allele_group_genotype_lhoods(ab2)
    assert np.array_equal(got2["lhood"].view(np.uint64), lh2.view(np.uint64)) and np.array_equal(got2["counts"], counts2)


def test_allele_group_genotype_lhoods_xwide(gpu):
    """groups of up to 16 alternate alleles (runs of up to eight samples; 153 genotypes, three to a lane): sk_allele_group_genotype_lhoods_xwide
    against the oracle on fresh groups of 1..16 and against the REFERENCE's own function on the committed fixture"""
    import os
    from strelka_amd import capi
    rng = np.random.default_rng(305)
    ab = synth.allele_group_batch(120, rng, depth_mean=40.0, min_alt=1, max_alt=capi.MAX_ALT_XWIDE, missing_rate=0.01)
    assert ab.width == capi.MAX_ALT_XWIDE
    got = gpu.allele_group_genotype_lhoods(ab)
    lh, counts, ng = pyoracle.allele_group_genotype_lhoods(ab)
    assert got["lhood"].shape[1] == 153 and np.array_equal(got["n_genotypes"], ng) and ng.max() == 153
    assert np.array_equal(got["lhood"].view(np.uint64), lh.view(np.uint64))
    assert np.array_equal(got["counts"], counts)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "allele_group_xwide_reference.npz"))
    gb = capi.HostAlleleGroupBatch(g["a_read_off"], g["a_n_alt"], g["a_ploidy"], g["a_del"], g["a_ins"], g["a_ref"], g["a_allele"],
                                   g["a_na"], g["a_rl"], g["a_flags"], width=capi.MAX_ALT_XWIDE)
    got = gpu.allele_group_genotype_lhoods(gb)
    assert np.array_equal(got["lhood"].view(np.uint64), g["a_lhood"].view(np.uint64))
    assert np.array_equal(got["counts"], g["a_counts"])
    # deep groups (more than one 64-read chunk), reads of several lengths and of one length (the priors once per group)
    for one_length in (False, True):
        ab2 = synth.allele_group_batch(5, rng, depth_mean=180.0, min_alt=10, max_alt=capi.MAX_ALT_XWIDE, missing_rate=0.0)
        if one_length:
            ab2.read_length[:] = 150
            ab2.non_ambig[:] = np.minimum(ab2.non_ambig, 150)
        got2 = gpu.allele_group_genotype_lhoods(ab2)
        lh2, counts2, _ = pyoracle.allele_group_genotype_lhoods(ab2)
        assert np.array_equal(got2["lhood"].view(np.uint64), lh2.view(np.uint64)) and np.array_equal(got2["counts"], counts2)


def test_the_librarys_default_is_the_fast_form_and_the_helpers_ask_for_the_exact_one():
    """sk_indel_options_default sets fast_form = 1 (what the drop-in runs; $STRELKA_AMD_INDEL_EXACT=1 in the adapter
    for the other); the test helpers (capi.indel_options) ask for the exact form, because they compare doubles bit for bit"""
    from strelka_amd import capi
    assert capi.indel_options(True, exact=False).fast_form == 1 and capi.indel_options(False, exact=False).fast_form == 1
    assert capi.indel_options(True).fast_form == 0


def test_indel_fast_form_keeps_every_integer_output(gpu):
    """sk_indel_options.fast_form (the library's default): two exp per read shared by its 21 states instead of the reference's operation order.
    Likelihoods then agree to ~1e-13 absolute (1e-15 relative to the terms' magnitude) instead of bit for bit; over 10^6 candidate
    indels the somatic calls derived from them -- QSI, QSI_NT, NTYPE, max_gt of either tier -- are the same."""
    from strelka_amd import capi
    rng = np.random.default_rng(811)
    total = same = 0
    worst = 0.0
    for rep in range(8):
        n = 1 << 17
        normal = synth.readscore_batch(n, rng, depth_mean=40.0)
        tumor = synth.readscore_batch(n, rng, depth_mean=110.0)
        tumor.del_len, tumor.ins_len = normal.del_len, normal.ins_len
        k = len(tumor.indel_lnp)
        boost = np.repeat(rng.random(n) < 0.5, np.diff(tumor.read_off))
        tumor.indel_lnp = np.where(boost & (rng.random(k) < 0.3), 0.0, tumor.indel_lnp).astype(np.float32)
        tumor.ref_lnp = np.where(boost, np.minimum(tumor.ref_lnp, -1.0), tumor.ref_lnp).astype(np.float32)
        err = rng.choice([5e-5, 1e-4, 3e-3, 2e-2], n)
        res = {}
        for fast in (0, 1):
            nopt, topt = capi.indel_options(True), capi.indel_options(True)
            nopt.min_read_bp_flank = 1
            nopt.fast_form = topt.fast_form = fast
            res[fast] = [capi.somatic_indel_call(normal, tumor, err, nopt, topt, is_include_tier2=t2) for t2 in (False, True)]
        for t in range(2):
            a, b = res[0][t], res[1][t]
            for f in ("max_gt", "qphred", "from_ntype_qphred", "ntype"):
                assert np.array_equal(a[f], b[f]), (rep, t, f, int((a[f] != b[f]).sum()))
            d = np.abs(a["tumor_lhood"] - b["tumor_lhood"]) / np.maximum(1.0, np.abs(a["tumor_lhood"]))
            worst = max(worst, float(d.max()))
        total += n
    assert total >= 1000000
    assert worst < 1e-12, worst


def test_process_waits_blocking_on_its_device(gpu):
    """the blocking-wait flag is per device and must be set after the device is chosen; sk_sync_mode reports what the
    runtime holds for this process's device"""
    from strelka_amd import capi
    assert capi.lib().sk_sync_mode() == 1
