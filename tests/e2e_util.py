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
    """The segment command of a WGS run (isHighDepthFilter on, no --exome): PY/strelkaGermlineWorkflow.py:81-147 with several
    --region per process (gsegGroup, PY/strelkaSharedWorkflow.py:174-175), --chrom-depth-file (:125-126), --ploidy-region-vcf
    (:131-132), --nocompress-bed (:128-129) and --gvcf-skip-header for every segment but the first (:120-121)."""
    cmd = [os.path.join(BIN_DIR, binary)]
    for r in regions:
        cmd += ["--region", r]
    cmd += ["--ref", ref, "--max-indel-size", "49", "--min-mapping-quality", "20",
            "--gvcf-output-prefix", out_prefix, "--gvcf-min-gqx", "15", "--gvcf-min-homref-gqx", "15",
            "--gvcf-max-snv-strand-bias", "10", "--enable-read-backed-phasing",
            "--stats-file", out_prefix + "runStats.xml"]
    for b in bams:
        cmd += ["--align-file", b]
    if skip_header:
        cmd.append("--gvcf-skip-header")
    cmd += ["--chrom-depth-file", chrom_depth]
    if nocompress_bed:
        cmd += ["--nocompress-bed", nocompress_bed]
    if ploidy_vcf:
        cmd += ["--ploidy-region-vcf", ploidy_vcf]
    cmd += ["--indel-error-models-file", demo("indelErrorModel.json"), "--theta-file", demo("theta.json")]
    return cmd + list(extra)


def wgs_dataset(length=1000000, depth=40.0, seed=20260926):
    """A WGS-like sample (tools/make_wgs_bam.py), made on the spot under oracle/_ref/synth/ (git-ignored; too large to travel with
    a snapshot: ~18 MB per Mb) and kept for the next caller.  -> directory with wgs.bam(.bai), wgs.fa(.fai), chrom_depth.txt"""
    import sys
    d = os.path.join(REF_DIR, "synth", "wgs_%d_%g_%d" % (length, depth, seed))
    if not os.path.exists(os.path.join(d, "chrom_depth.txt")):
        os.makedirs(d, exist_ok=True)
        subprocess.run([sys.executable, os.path.join(REPO, "tools", "make_wgs_bam.py"), d, os.path.join(BIN_DIR, "samtools"),
                        "--length", str(length), "--depth", str(depth), "--seed", str(seed)], check=True, stdout=subprocess.DEVNULL)
        with open(os.path.join(d, "chrom_depth.txt"), "w") as f:  # GetChromDepth's output format: chrom <tab> mean depth
            f.write("chrW\t%.3f\n" % depth)
    return d


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
