"""The segment farm (strelka_amd/farm.py) and the workflow's WGS flags, end to end.

A WGS-like sample (tools/make_wgs_bam.py) is cut into segments the way the reference's workflow cuts a genome
(getChromIntervals / getGenomeSegmentGroups), one caller process per group -- some groups with several --region --, with the
flags a WGS run passes (--chrom-depth-file, --ploidy-region-vcf, --nocompress-bed, --gvcf-skip-header): the farm's joined
outputs through the adapter must be the unmodified reference's, byte for byte, whatever the number of concurrent processes
and devices.  CPU tier: the adapter on the test double of the C-ABI; GPU tier: the product library, several processes
sharing the device."""
import os
import subprocess

import pytest

from strelka_amd import farm
from tests import e2e_util as E

LENGTH = 600000
OUTPUTS = ("variants.vcf", "genome.S1.vcf")


def _have(*names):
    return E.have(*names) and os.path.exists(os.path.join(E.BIN_DIR, "tabix"))


def _dataset(tmp_path_factory):
    d = E.wgs_dataset(LENGTH)
    extra = tmp_path_factory.mktemp("wgs_flags")
    # a haploid stretch and a ploidy-0 stretch (e.g. chrX / chrY of a male sample): ##FORMAT CN per sample, END in INFO
    # (parsePloidyFromVcf, L/starling_common/ploidy_util.cpp)
    vcf = extra / "ploidy.vcf"
    vcf.write_text("##fileformat=VCFv4.1\n##INFO=<ID=END,Number=1,Type=Integer,Description=\"end\">\n"
                   "##FORMAT=<ID=CN,Number=1,Type=Integer,Description=\"copy number\">\n"
                   "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tNA_SYNTH\n"
                   "chrW\t150000\t.\tN\t<CNV>\t.\tPASS\tEND=260000\tCN\t1\n"
                   "chrW\t400001\t.\tN\t<CNV>\t.\tPASS\tEND=430000\tCN\t0\n")
    subprocess.run([os.path.join(E.BIN_DIR, "bgzip"), "-f", str(vcf)], check=True)
    subprocess.run([os.path.join(E.BIN_DIR, "tabix"), "-p", "vcf", str(vcf) + ".gz"], check=True)
    bed = extra / "nocompress.bed"
    bed.write_text("chrW\t99000\t101000\nchrW\t505000\t505100\n")
    subprocess.run([os.path.join(E.BIN_DIR, "bgzip"), "-f", str(bed)], check=True)
    subprocess.run([os.path.join(E.BIN_DIR, "tabix"), "-p", "bed", str(bed) + ".gz"], check=True)
    return d, str(vcf) + ".gz", str(bed) + ".gz"


def _groups():
    # 600 kb in 70 kb pieces -> 9 segments; the workflow groups consecutive pieces up to 200 kb: three or so --region per process
    segs = list(farm.chrom_intervals(["chrW"], {"chrW": LENGTH}, 70000))
    groups = list(farm.segment_groups(segs, min_group_size=200000))
    assert len(segs) == 9 and 3 <= len(groups) <= 5 and max(len(g) for g in groups) >= 2
    assert segs[0][2] == 1 and segs[-1][3] == LENGTH and all(a[3] + 1 == b[2] for a, b in zip(segs, segs[1:]))
    return groups


def _run(binary, tmp, data, jobs, n_gpus=1, env=None, evs_models=None):
    d, ploidy, bed = data

    def argv(index, regions, prefix, skip_header):
        return farm.germline_segment_argv(binary, prefix, [os.path.join(d, "wgs.bam")], regions, os.path.join(d, "wgs.fa"),
                                          chrom_depth=os.path.join(d, "chrom_depth.txt"), ploidy_vcf=ploidy, nocompress_bed=bed,
                                          skip_header=skip_header, evs_models=evs_models, report_evs_features=bool(evs_models))
    return farm.run_farm(_groups(), argv, str(tmp), OUTPUTS, n_gpus=n_gpus, jobs=jobs, env=env)


def _body(path):
    with open(path) as f:
        return [l for l in f.read().splitlines() if not (l.startswith("##cmdline=") or l.startswith("##startTime=") or l.startswith("##fileDate="))]


def test_chrom_intervals_follow_the_workflow():
    # PY/workflowUtil.py:182-218 on a 30 Mb + 5 Mb genome at 12 Mb: 3 equal pieces, then 1
    segs = list(farm.chrom_intervals(["a", "b"], {"a": 30000000, "b": 5000000}, 12000000))
    assert [(s[1], s[2], s[3], s[4]) for s in segs] == [("a", 1, 10000000, 0), ("a", 10000001, 20000000, 1), ("a", 20000001, 30000000, 2),
                                                        ("b", 1, 5000000, 0)]
    segs = list(farm.chrom_intervals(["a"], {"a": 25}, 10))
    assert [(s[2], s[3]) for s in segs] == [(1, 9), (10, 17), (18, 25)]  # 25 = 9 + 8 + 8: the first `size % n` pieces get the extra base


@pytest.mark.skipif(not _have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
def test_farm_with_wgs_flags_identical_through_adapter_cpu_double(tmp_path, tmp_path_factory):
    data = _dataset(tmp_path_factory)
    ref = _run("starling2_ref", tmp_path / "ref", data, jobs=4)
    want = {n: _body(ref.outputs[n]) for n in OUTPUTS}
    assert sum(1 for l in want["variants.vcf"] if not l.startswith("#")) > 400
    # the ploidy regions reached the records: haploid genotypes inside 150000-260000
    hap = [l for l in want["variants.vcf"] if not l.startswith("#") and 150000 <= int(l.split("\t")[1]) <= 260000]
    assert hap and any(l.split("\t")[9].split(":")[0] in ("0", "1") for l in hap)
    for jobs in (1, 4):
        got = _run("starling2_dbl", tmp_path / ("dbl%d" % jobs), data, jobs=jobs, env={"STRELKA_AMD_VERBOSE": "1"})
        for n in OUTPUTS:
            assert _body(got.outputs[n]) == want[n], (jobs, n)
        assert len(got.process_s) == len(_groups())
        assert all("pileup: pushes=" in t and "genotyping=1" in t for t in got.stderr_tails)


@pytest.mark.skipif(not _have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
def test_farm_with_wgs_flags_and_evs_models_identical_cpu_double(tmp_path, tmp_path_factory):
    """the same farm with the workflow's default scoring (EVS models on the command line; stand-in models, every feature printed):
    several regions per process, haploid and ploidy-0 stretches, the rank sums rebuilt from the stream per region"""
    import subprocess
    import sys
    data = _dataset(tmp_path_factory)
    md = tmp_path / "models"
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "make_dummy_germline_models.py"),
                    str(md)], check=True)
    models = (str(md / "germlineSNVScoringModels.json"), str(md / "germlineIndelScoringModels.json"))
    ref = _run("starling2_ref", tmp_path / "ref", data, jobs=4, evs_models=models)
    want = {n: _body(ref.outputs[n]) for n in OUTPUTS}
    assert sum(1 for l in want["variants.vcf"] if "EVSF=" in l) > 300
    got = _run("starling2_dbl", tmp_path / "dbl", data, jobs=4, env={"STRELKA_AMD_VERBOSE": "1"}, evs_models=models)
    for n in OUTPUTS:
        assert _body(got.outputs[n]) == want[n], n
    assert all("pileup: pushes=" in t and "genotyping=1" in t for t in got.stderr_tails)


@pytest.mark.skipif(not _have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
def test_reference_with_n_runs_identical_cpu_double(tmp_path):
    """A genome has runs of N (gaps, masked repeats) and contigs start at position 1; the synthetic reference has neither.  The same
    reads against a copy of the reference with N runs written over it -- short ones inside homopolymers and tandem repeats, a long
    one, one that begins the contig -- from position 1: every base comparison of the host code beside the routed sites
    (active-region bookkeeping, repeat finder, valid alignment range) and of the routed sites themselves now meets N on the
    reference side, and the reads over the runs are all mismatch there."""
    import subprocess
    import sys
    import numpy as np
    d = E.wgs_dataset(LENGTH)
    lines = open(os.path.join(d, "wgs.fa")).read().split("\n")
    seq = bytearray("".join(lines[1:]).encode())
    rng = np.random.default_rng(77)
    seq[0:180] = b"N" * 180                      # the contig begins with a gap
    seq[60000:63000] = b"N" * 3000               # longer than any read and than the detector's ring of 1000 positions
    for p in rng.integers(1000, 250000, 120):    # short runs, some of them inside the planted repeats
        n = int(rng.integers(1, 60))
        seq[int(p):int(p) + n] = b"N" * n
    fa = tmp_path / "masked.fa"
    with open(fa, "w") as f:
        f.write(lines[0] + "\n")
        for i in range(0, len(seq), 60):
            f.write(seq[i:i + 60].decode() + "\n")
    subprocess.run([os.path.join(E.BIN_DIR, "samtools"), "faidx", str(fa)], check=True)
    md = tmp_path / "models"
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "make_dummy_germline_models.py"),
                    str(md)], check=True)
    models = (str(md / "germlineSNVScoringModels.json"), str(md / "germlineIndelScoringModels.json"))
    out = {}
    for variant in ("ref", "dbl"):
        o = tmp_path / variant
        o.mkdir()
        argv = farm.germline_segment_argv("starling2_" + variant, str(o) + "/", [os.path.join(d, "wgs.bam")], ["chrW:1-100000", "chrW:100001-260000"],
                                          str(fa), chrom_depth=os.path.join(d, "chrom_depth.txt"), evs_models=models, report_evs_features=True)
        E.run(argv, env={"STRELKA_AMD_VERBOSE": "1"} if variant == "dbl" else None, timeout=1800)
        out[variant] = {n: _body(str(o / n)) for n in OUTPUTS}
    assert sum(1 for l in out["ref"]["variants.vcf"] if l and l[0] != "#") > 100
    assert sum(1 for l in out["ref"]["genome.S1.vcf"] if "\tN\t" in l) > 50  # (records whose reference base is N)
    for n in OUTPUTS:
        assert out["dbl"][n] == out["ref"][n], n


@pytest.mark.gpu
@pytest.mark.skipif(not _have("starling2_ref", "starling2_amd"), reason="oracle/_ref binaries not built")
def test_farm_with_wgs_flags_identical_through_adapter_gpu(tmp_path, tmp_path_factory):
    data = _dataset(tmp_path_factory)
    ref = _run("starling2_ref", tmp_path / "ref", data, jobs=4)
    want = {n: _body(ref.outputs[n]) for n in OUTPUTS}
    for jobs, n_gpus in ((1, 1), (4, 1), (4, 4)):  # (4 devices named on a 1-GPU box: the adapter takes the index modulo the devices present)
        env = {"STRELKA_AMD_VERBOSE": "1"}
        got = _run("starling2_amd", tmp_path / ("amd%d_%d" % (jobs, n_gpus)), data, jobs=jobs, n_gpus=n_gpus, env=env)
        for n in OUTPUTS:
            assert _body(got.outputs[n]) == want[n], (jobs, n_gpus, n)


# ---- the somatic caller through the farm: several --region per process (the stream's region reset), EVS models, callable regions ----

SOMATIC_LENGTH = 200000
SOMATIC_OUTPUTS = ("somatic.snvs.vcf", "somatic.indels.vcf", "somatic.callable.regions.bed")


def _somatic_groups():
    segs = list(farm.chrom_intervals(["chrW"], {"chrW": SOMATIC_LENGTH}, 30000))
    groups = list(farm.segment_groups(segs, min_group_size=70000))
    assert len(segs) == 7 and 2 <= len(groups) <= 4 and max(len(g) for g in groups) >= 2
    return groups


def _run_somatic(binary, tmp, jobs, env=None):
    d = farm.wgs_somatic_dataset(SOMATIC_LENGTH)

    def argv(index, regions, prefix, skip_header):
        return farm.somatic_segment_argv(binary, prefix, os.path.join(d, "normal.bam"), os.path.join(d, "tumor.bam"), regions,
                                         os.path.join(d, "normal.fa"), chrom_depth=os.path.join(d, "chrom_depth.txt"), callable_regions=True,
                                         skip_header=skip_header)
    return farm.run_farm(_somatic_groups(), argv, str(tmp), SOMATIC_OUTPUTS, jobs=jobs, env=env)


@pytest.mark.skipif(not _have("strelka2_ref", "strelka2_dbl"), reason="oracle/_ref binaries not built")
def test_somatic_farm_identical_through_adapter_cpu_double(tmp_path):
    ref = _run_somatic("strelka2_ref", tmp_path / "ref", jobs=4)
    want = {n: _body(ref.outputs[n]) for n in SOMATIC_OUTPUTS}
    assert sum(1 for l in want["somatic.snvs.vcf"] if not l.startswith("#")) >= 5
    got = _run_somatic("strelka2_dbl", tmp_path / "dbl", jobs=4, env={"STRELKA_AMD_VERBOSE": "1"})
    for n in SOMATIC_OUTPUTS:
        assert _body(got.outputs[n]) == want[n], n
    assert all("pileup: pushes=" in t and "genotyping=1" in t for t in got.stderr_tails)


@pytest.mark.skipif(not _have("strelka2_ref", "strelka2_dbl"), reason="oracle/_ref binaries not built")
def test_somatic_reference_with_n_runs_identical_cpu_double(tmp_path):
    """the somatic caller on the tumour / normal pair against the reference with N runs written over it, from position 1 (see
    test_reference_with_n_runs_identical_cpu_double)"""
    import subprocess
    import numpy as np
    d = farm.wgs_somatic_dataset(SOMATIC_LENGTH)
    lines = open(os.path.join(d, "normal.fa")).read().split("\n")
    seq = bytearray("".join(lines[1:]).encode())
    rng = np.random.default_rng(78)
    seq[0:150] = b"N" * 150
    seq[40000:41500] = b"N" * 1500
    for p in rng.integers(1000, 110000, 60):
        n = int(rng.integers(1, 50))
        seq[int(p):int(p) + n] = b"N" * n
    fa = tmp_path / "masked.fa"
    with open(fa, "w") as f:
        f.write(lines[0] + "\n")
        for i in range(0, len(seq), 60):
            f.write(seq[i:i + 60].decode() + "\n")
    subprocess.run([os.path.join(E.BIN_DIR, "samtools"), "faidx", str(fa)], check=True)
    names = ("somatic.snvs.vcf", "somatic.indels.vcf", "somatic.callable.regions.bed")
    out = {}
    for variant in ("ref", "dbl"):
        o = tmp_path / variant
        o.mkdir()
        argv = farm.somatic_segment_argv("strelka2_" + variant, str(o) + "/", os.path.join(d, "normal.bam"), os.path.join(d, "tumor.bam"),
                                         ["chrW:1-120000"], str(fa), chrom_depth=os.path.join(d, "chrom_depth.txt"), callable_regions=True)
        E.run(argv, env={"STRELKA_AMD_VERBOSE": "1"} if variant == "dbl" else None, timeout=1800)
        out[variant] = {n: _body(str(o / n)) for n in names}
    assert sum(1 for l in out["ref"]["somatic.snvs.vcf"] if l and l[0] != "#") > 5
    for n in names:
        assert out["dbl"][n] == out["ref"][n], n


@pytest.mark.gpu
@pytest.mark.skipif(not _have("strelka2_ref", "strelka2_amd"), reason="oracle/_ref binaries not built")
def test_somatic_farm_identical_through_adapter_gpu(tmp_path):
    ref = _run_somatic("strelka2_ref", tmp_path / "ref", jobs=4)
    want = {n: _body(ref.outputs[n]) for n in SOMATIC_OUTPUTS}
    for jobs in (1, 4):
        got = _run_somatic("strelka2_amd", tmp_path / ("amd%d" % jobs), jobs=jobs, env={"STRELKA_AMD_VERBOSE": "1"})
        for n in SOMATIC_OUTPUTS:
            assert _body(got.outputs[n]) == want[n], (jobs, n)
