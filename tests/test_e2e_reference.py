"""The reference's own caller binaries, built here from its own translation units against oracle/boost_shim
(oracle/Makefile, `make ref`), reproduce the results the reference ships: src/demo/expectedResults/somatic.{snvs,indels}.vcf.gz
(every record, including QSS/QSS_NT/QSI/QSI_NT, the tier fields and the SomaticEVS random-forest score).  This is what
validates the stand-in Boost headers (log1p, binomial/hypergeometric distributions, program_options, ...) against outputs
held by the reference itself, and the baseline the adapter binaries are compared with (tests/test_e2e_adapter.py).
"""
import pytest

from tests import e2e_util as E

pytestmark = pytest.mark.skipif(not E.have("strelka2_ref", "starling2_ref"),
                                reason="oracle/_ref binaries not built (needs /root/reference: make -C oracle ref)")


def test_strelka2_ref_reproduces_expected_somatic_results(tmp_path):
    out = str(tmp_path) + "/"
    # src/demo/runStrelkaSomaticWorkflowDemo.bash: tumor NA12891, normal NA12892
    E.run(E.somatic_argv("strelka2_ref", out, E.demo("NA12892_demo20.bam"), E.demo("NA12891_demo20.bam")))
    for kind in ("snvs", "indels"):
        got = E.vcf_body(out + "somatic.%s.vcf" % kind)
        want = E.vcf_body(E.demo("somatic.%s.vcf.gz" % kind))
        assert len(want) > 2
        assert got == want, kind


def test_starling2_ref_runs_germline_demo(tmp_path):
    """The germline demo has no expected file in the reference (its diff step is commented out,
    src/demo/runStrelkaGermlineWorkflowDemo.bash:119-152): check the run completes and calls the demo's variants."""
    out = str(tmp_path) + "/"
    E.run(E.germline_argv("starling2_ref", out, [E.demo("NA12891_demo20.bam"), E.demo("NA12892_demo20.bam")]))
    body = E.vcf_body(out + "variants.vcf")
    assert body[0].startswith("#CHROM") and body[0].endswith("NA12891\tNA12892")
    pos = [int(l.split("\t")[1]) for l in body[1:]]
    assert 1706 in pos and 3664 in pos and len(pos) >= 15
    assert len(E.vcf_body(out + "genome.S1.vcf")) > 50
