"""Command lines of the per-segment caller processes, as the reference's pyflow workflow builds them.

TEST INFRASTRUCTURE.  `germline_argv` follows PY/strelkaGermlineWorkflow.py:81-147 + PY/strelkaSharedWorkflow.py:164-200
(`PY/` = /root/reference/src/python/lib), `somatic_argv` PY/strelkaSomaticWorkflow.py:74-146 with the defaults of
src/python/bin/configureStrelkaSomaticWorkflow.py.ini, both for the `--exome` demo configuration of
src/demo/run*WorkflowDemo.bash (no depth filter).  The binaries are built by oracle/Makefile (`*_ref`: the reference's own
translation units and main()) and adapter/Makefile (`*_amd`: the same with the hot-path call sites re-routed through
libstrelka_amd.so); they and the demo inputs live under oracle/_ref/, which travels to the GPU box.
"""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(REPO, "oracle", "_ref")
BIN_DIR = os.path.join(REF_DIR, "bin")
DEMO_DIR = os.path.join(REF_DIR, "demo")


def have(*names):
    return all(os.path.exists(os.path.join(BIN_DIR, n)) for n in names) and os.path.exists(
        os.path.join(DEMO_DIR, "demo20.fa"))


def demo(name):
    return os.path.join(DEMO_DIR, name)


def germline_argv(binary, out_prefix, bams, region="demo20:1-5000", ref=None, extra=()):
    cmd = [os.path.join(BIN_DIR, binary), "--region", region, "--ref", ref or demo("demo20.fa"),
           "--max-indel-size", "49", "--min-mapping-quality", "20",
           "--gvcf-output-prefix", out_prefix, "--gvcf-min-gqx", "15", "--gvcf-min-homref-gqx", "15",
           "--gvcf-max-snv-strand-bias", "10", "--enable-read-backed-phasing",
           "--stats-file", out_prefix + "runStats.xml"]
    for b in bams:
        cmd += ["--align-file", b]
    cmd += ["--indel-error-models-file", demo("indelErrorModel.json"), "--theta-file", demo("theta.json")]
    return cmd + list(extra)


def germline_wgs_argv(binary, out_prefix, bams, regions, ref, chrom_depth, ploidy_vcf=None, nocompress_bed=None, skip_header=False,
                      extra=()):
    """the segment command of a WGS run, as the farm builds it (strelka_amd/farm.py cites the workflow lines)"""
    from strelka_amd import farm
    return farm.germline_segment_argv(binary, out_prefix, bams, regions, ref, chrom_depth=chrom_depth, ploidy_vcf=ploidy_vcf,
                                      nocompress_bed=nocompress_bed, skip_header=skip_header, extra=extra)


def wgs_dataset(length=1000000, depth=40.0, seed=20260926):
    from strelka_amd import farm
    return farm.wgs_dataset(length, depth, seed)


def somatic_argv(binary, out_prefix, normal_bam, tumor_bam, region="demo20:1-5000", ref=None, extra=()):
    cmd = [os.path.join(BIN_DIR, binary), "--region", region, "--ref", ref or demo("demo20.fa"),
           "--max-indel-size", "49", "--min-mapping-quality", "20",
           "--somatic-snv-rate", "0.0001", "--shared-site-error-rate", "0.0000000005",
           "--shared-site-error-strand-bias-fraction", "0.0", "--somatic-indel-rate", "0.000001",
           "--shared-indel-error-factor", "2.2", "--tier2-min-mapping-quality", "0",
           "--strelka-snv-max-filtered-basecall-frac", "0.4", "--strelka-snv-max-spanning-deletion-frac", "0.75",
           "--strelka-snv-min-qss-ref", "15", "--strelka-indel-max-window-filtered-basecall-frac", "0.3",
           "--strelka-indel-min-qsi-ref", "40", "--ssnv-contam-tolerance", "0.15", "--indel-contam-tolerance", "0.15",
           "--somatic-snv-scoring-model-file", demo("somaticSNVScoringModels.json"),
           "--somatic-indel-scoring-model-file", demo("somaticIndelScoringModels.json"),
           "--normal-align-file", normal_bam, "--tumor-align-file", tumor_bam,
           "--somatic-snv-file", out_prefix + "somatic.snvs.vcf", "--somatic-indel-file", out_prefix + "somatic.indels.vcf",
           "--stats-file", out_prefix + "runStats.xml"]
    return cmd + list(extra)


def run(cmd, env=None, timeout=600):
    e = dict(os.environ)
    if env:
        e.update(env)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError("%s failed (%d):\n%s" % (" ".join(cmd), p.returncode, p.stderr.decode(errors="replace")[-4000:]))
    return p


def vcf_body(path, keep_header=False):
    """Lines of a VCF; header lines carrying the command line / start time / paths are dropped unless asked for."""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        lines = f.read().splitlines()
    if keep_header:
        return [l for l in lines if not (l.startswith("##cmdline=") or l.startswith("##startTime=") or
                                         l.startswith("##fileDate="))]
    return [l for l in lines if not l.startswith("##")]
