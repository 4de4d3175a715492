"""The stated limits of the library fail loudly (never a silently different answer), and the one documented inexact mode
keeps its bar.  DESIGN.md section 1 lists the limits: DNA reads only (no spliced read segments), <= SK_MAX_ALT alternate
alleles per germline allele group, <= SK_MAX_SAMPLES samples per realignment job, basecall qualities <= 70, and -- when
the host C library is not the one the kernels restate -- agreement with the reference to 1e-5 instead of bit for bit."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import capi, synth


def test_more_samples_than_supported_is_refused():
    o = capi.RealignOptions()
    capi.lib().sk_realign_options_default(C.byref(o))
    o.sample_count = 9  # (SK_MAX_SAMPLES = 8)
    capi.lib().sk_realign_job_create.restype = C.c_void_p
    assert not capi.lib().sk_realign_job_create(C.byref(o))


def test_spliced_read_is_refused():
    """a read whose path holds a SKIP (N) segment is an RNA read: exon pins are not built, the job must say so"""
    job = capi.RealignJob()
    job.set_reference("ACGT" * 100, 0)
    job.set_indels([])
    code = np.full(50, 1, np.uint8)
    qual = np.full(50, 30, np.uint8)
    path = [(1, 25), (4, 100), (1, 25)]  # SK_SEG_MATCH, SK_SEG_SKIP
    with pytest.raises(RuntimeError, match="(?i)spliced"):
        job.add_read(code, qual, 10, path, realign_range=(0, 400))


@pytest.mark.gpu
def test_more_alt_alleles_than_supported_is_refused():
    capi.init(0)
    rng = np.random.default_rng(1)
    b = synth.allele_group_batch(4, rng)
    b.n_alt[:] = 4
    with pytest.raises(RuntimeError, match="n_alt"):
        capi.allele_group_genotype_lhoods(b)
    # ... and the wide entry (a multi-sample run's groups) holds ploidy x 4 samples = 8, no more
    w = synth.allele_group_batch(4, rng, max_alt=capi.MAX_ALT_WIDE)
    capi.allele_group_genotype_lhoods(w)
    w.n_alt[:] = 9
    with pytest.raises(RuntimeError, match="n_alt"):
        capi.allele_group_genotype_lhoods(w)
    # ... and the widest one ploidy x 8 samples = 16
    x = synth.allele_group_batch(3, rng, max_alt=capi.MAX_ALT_XWIDE)
    capi.allele_group_genotype_lhoods(x)
    x.n_alt[:] = 17
    with pytest.raises(RuntimeError, match="n_alt"):
        capi.allele_group_genotype_lhoods(x)


@pytest.mark.gpu
def test_qscore_above_70_is_refused_on_host_and_flagged_on_device():
    capi.init(0)
    rng = np.random.default_rng(2)
    cases = synth.align_cases(8, rng)
    hb = synth.build_align_batch(cases)
    hb.read_qual[3] = 71
    with pytest.raises(RuntimeError, match="exceeds the maximum cached score of 70"):
        capi.score_alignments(hb)
    # device-resident twin: cannot validate, must not stay silent
    from strelka_amd import device
    db = device.DeviceAlignBatch(hb)
    db.score()
    assert capi.lib().sk_check_device_errors() != 0
    assert b"exceeds the maximum cached score of 70" in capi.lib().sk_last_error()
    assert capi.lib().sk_check_device_errors() == 0  # cleared


@pytest.mark.gpu
def test_device_libm_fallback_keeps_the_documented_bar():
    """sk_libm_restated() == 0 path: the device math library stands in for the host libm's routines; log-likelihoods agree
    with the reference to 1e-5 relative (north_star's bar), integer outputs almost everywhere"""
    capi.init(0)
    L = capi.lib()
    rng = np.random.default_rng(3)
    pb = synth.pileups(20000, rng, het_rate=0.05, hom_rate=0.02)
    want_de = pyoracle.adjust_joint_eprob(pb)
    want = pyoracle.site_digt_call(pb, want_de)
    n, t = synth.somatic_pileups(20000, rng, somatic_rate=0.02, het_rate=0.02)
    want_s = pyoracle.somatic_snv_call(n, t)
    assert L.sk_debug_force_device_libm(1) == 0
    try:
        assert L.sk_libm_restated() == 0
        got, de = capi.site_digt_call_fused(pb, want_de=True)
        got_s = capi.somatic_snv_call(n, t)
    finally:
        L.sk_debug_force_device_libm(0)
    assert L.sk_libm_restated() == 1
    assert np.allclose(de, want_de, rtol=1e-5, atol=0)
    assert np.allclose(got["lhood"], want["lhood"], rtol=1e-5, atol=1e-5)
    assert np.mean(got["genome"]["max_gt"] == want["genome"]["max_gt"]) > 0.9999
    assert np.abs(got["genome"]["snp_qphred"].astype(int) - want["genome"]["snp_qphred"]).max() <= 1
    assert np.allclose(got_s["tumor_lhood"], want_s["tumor_lhood"], rtol=1e-5, atol=1e-5)
    assert np.abs(got_s["qphred"].astype(int) - want_s["qphred"]).max() <= 1
    # and the exact path is back
    again, _ = capi.site_digt_call_fused(pb)
    assert again.tobytes() == want.tobytes()
