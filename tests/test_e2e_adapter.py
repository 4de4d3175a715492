"""End to end through the drop-in boundary: the reference's own caller programs with the hot-path call sites re-routed
through the C-ABI (adapter/, built by adapter/Makefile) against the unmodified reference programs (oracle/Makefile),
same command line, same inputs -- every output file must be byte-identical (only the two header lines that echo the
command line and the start time are ignored).

  *_ref : the reference's translation units, untouched                                   (baseline)
  *_amd : the same with adapter/ + strelka_amd/lib/libstrelka_amd.so (gfx950)            -> `-m gpu`
  *_dbl : the same with adapter/ + oracle/libstrelka_amd_double.so (CPU double of the C-ABI, test infrastructure)
          -> CPU suite: checks the adapter's host logic (marshalling, stage-window batching, geometry shadow)

Each run also reports what went through the C-ABI (STRELKA_AMD_VERBOSE=1), so an identical VCF cannot come from an adapter
that routed nothing.
"""
import re

import pytest

from tests import e2e_util as E

GERMLINE_BAMS = ["NA12891_demo20.bam", "NA12892_demo20.bam"]


def _counters(stderr):
    m = re.search(r"strelka_amd adapter: (.*)", stderr)
    assert m, "adapter did not report:\n" + stderr[-2000:]
    c = {k: int(v) for k, v in (kv.split("=") for kv in m.group(1).split())}
    f = re.search(r"strelka_amd adapter feed: (.*)", stderr)  # site 8: the region's reads came through the feed entry points
    assert f, "adapter did not report its feed:\n" + stderr[-2000:]
    c.update({"feed_" + k: (float(v) if "seconds" in k else int(v)) for k, v in (kv.split("=") for kv in f.group(1).split())})
    g = re.search(r"strelka_amd adapter pileup: (.*)", stderr)  # site 9: the pileup stream (germline)
    assert g, "adapter did not report its pileup stream:\n" + stderr[-2000:]
    c.update({"pileup_" + k: int(v) for k, v in (kv.split("=") for kv in g.group(1).split())})
    return c


def _germline(variant, tmp_path, windows=None, extra_env=None):
    ref_out, out = str(tmp_path / "ref") + "/", str(tmp_path / variant) + "/"
    (tmp_path / "ref").mkdir(exist_ok=True)
    (tmp_path / variant).mkdir(exist_ok=True)
    bams = [E.demo(b) for b in GERMLINE_BAMS]
    E.run(E.germline_argv("starling2_ref", ref_out, bams))
    env = {"STRELKA_AMD_VERBOSE": "1"}
    env.update(extra_env or {})
    if windows:
        env["STRELKA_AMD_READ_WINDOW"], env["STRELKA_AMD_SITE_WINDOW"] = str(windows[0]), str(windows[1])
    p = E.run(E.germline_argv("starling2_" + variant, out, bams), env=env)
    c = _counters(p.stderr.decode())
    assert c["realign_reads"] > 1000 and c["realign_jobs"] >= 1 and c["site_loci"] > 5000
    if (extra_env or {}).get("STRELKA_AMD_FEED") == "0":
        assert c["feed_regions"] == 0
    elif (extra_env or {}).get("STRELKA_AMD_FEED_NORMALIZE") == "0":
        assert c["feed_regions"] == 2 and c["feed_normalize_batches"] == 0
    else:  # both BAMs' regions, every read the realigner saw and more (the reference filters some after the stream)
        assert c["feed_regions"] == 2 and c["feed_records"] >= c["realign_reads"] and c["feed_bgzf_blocks"] >= 2
        # and their alignments were normalised in one batch per slice of a region (kernel B4 behind sk_normalize_alignments)
        assert c["feed_normalize_batches"] >= 2 and c["feed_normalized"] >= c["realign_reads"] and c["feed_normalize_declined"] == 0
    assert c["indel_groups"] >= 1 and c["haplotypes"] >= 1
    if (extra_env or {}).get("STRELKA_AMD_PILEUP") == "0":
        assert c["pileup_pushes"] == 0
    else:  # site 9: every read that piles up went through the stream, and (unless switched off) the genotypes came with the columns
        assert c["pileup_pushes"] >= 2 and c["pileup_reads"] > 1000 and c["pileup_loci"] > 5000
        assert c["pileup_genotyping"] == (0 if (extra_env or {}).get("STRELKA_AMD_PILEUP_GENOTYPE") == "0" else 1)
    for f in ("variants.vcf", "genome.S1.vcf", "genome.S2.vcf"):
        want, got = E.vcf_body(ref_out + f, keep_header=True), E.vcf_body(out + f, keep_header=True)
        assert len(want) > 50
        assert got == want, f
    return c


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("windows", [None, (0, 0), (1, 1), (7, 13), (256, 512), (500, 512), (1000, 3000), (1500, 100), (5000, 8000)])
def test_germline_demo_identical_through_adapter_cpu_double(tmp_path, windows):
    c = _germline("dbl", tmp_path, windows)
    if windows == (1000, 3000):
        assert c["realign_jobs"] <= 10 and c["pileup_pushes"] <= 12
    if windows is None:  # (the site window that hides a pileup window's device time behind the stage machine: the push in two halves)
        assert c["read_window"] == 8192 and c["site_window"] == 1024


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("windows", [None, (300, 7)])
def test_germline_demo_identical_with_pushes_finished_at_once(tmp_path, windows):
    """STRELKA_AMD_PUSH_ASYNC=0: every pileup window begun and finished in one call (no site window by default)"""
    c = _germline("dbl", tmp_path, windows, extra_env={"STRELKA_AMD_PUSH_ASYNC": "0"})
    if windows is None:
        assert c["site_window"] == 0


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("blocks", ["1", "3"])
def test_germline_demo_identical_with_small_feed_slices(tmp_path, blocks):
    """the feed holds one slice of a region at a time (a record the slice end cuts is carried into the next): slices of one and of
    three BGZF blocks on the demo BAMs, where the default slice holds a whole region"""
    c = _germline("dbl", tmp_path, extra_env={"STRELKA_AMD_FEED_SLICE_BLOCKS": blocks})
    assert c["feed_normalize_batches"] > (2 if blocks == "1" else 1)


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
def test_germline_demo_identical_with_the_reference_stream(tmp_path):
    """STRELKA_AMD_FEED=0: the reads come through the reference's own htslib iterator instead of site 8"""
    _germline("dbl", tmp_path, extra_env={"STRELKA_AMD_FEED": "0"})


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
def test_germline_demo_identical_with_the_reference_normalisation(tmp_path):
    """STRELKA_AMD_FEED_NORMALIZE=0: the reads come through the feed, normalizeAlignment stays the reference's, record by record"""
    _germline("dbl", tmp_path, extra_env={"STRELKA_AMD_FEED_NORMALIZE": "0"})


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("windows", [None, (700, 900)])
def test_germline_demo_identical_with_the_reference_pileup(tmp_path, windows):
    """STRELKA_AMD_PILEUP=0: pileup_read_segment stays the reference's; the window's columns go up for the genotypes (sites 2+3)"""
    c = _germline("dbl", tmp_path, windows, extra_env={"STRELKA_AMD_PILEUP": "0"})
    assert c["site_batches"] >= 1


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
def test_germline_demo_identical_with_columns_only(tmp_path):
    """STRELKA_AMD_PILEUP_GENOTYPE=0: the stream builds the columns, the genotypes are computed per site window from the host's copy"""
    _germline("dbl", tmp_path, extra_env={"STRELKA_AMD_PILEUP_GENOTYPE": "0"})


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("starling2_ref", "starling2_amd"), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("windows", [None, (7, 13)])
def test_germline_demo_identical_through_adapter_gpu(tmp_path, windows):
    _germline("amd", tmp_path, windows)


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("starling2_ref", "starling2_amd"), reason="oracle/_ref binaries not built")
def test_germline_demo_identical_with_the_reference_pileup_gpu(tmp_path):
    _germline("amd", tmp_path, extra_env={"STRELKA_AMD_PILEUP": "0"})


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("starling2_ref", "starling2_amd"), reason="oracle/_ref binaries not built")
def test_germline_demo_identical_with_device_enumeration(tmp_path):
    """candidate alignments listed, flattened and scored on the device (sk_realign_options.enumeration = 2)"""
    c = _germline("amd", tmp_path, extra_env={"SK_ENUMERATION": "2"})
    assert c["enum_device_reads"] > 50 and c["enum_host_instead"] <= c["enum_device_reads"] // 20


def _somatic(variant, tmp_path, windows=None, callable_regions=False, extra_env=None):
    ref_out, out = str(tmp_path / "ref") + "/", str(tmp_path / variant) + "/"
    (tmp_path / "ref").mkdir(exist_ok=True)
    (tmp_path / variant).mkdir(exist_ok=True)
    normal, tumor = E.demo("NA12892_demo20.bam"), E.demo("NA12891_demo20.bam")
    extra = lambda o: (["--somatic-callable-regions-file", o + "somatic.callable.regions.bed"] if callable_regions else [])
    E.run(E.somatic_argv("strelka2_ref", ref_out, normal, tumor, extra=extra(ref_out)))
    env = {"STRELKA_AMD_VERBOSE": "1"}
    if windows:
        env["STRELKA_AMD_READ_WINDOW"], env["STRELKA_AMD_SITE_WINDOW"] = str(windows[0]), str(windows[1])
    env.update(extra_env or {})
    p = E.run(E.somatic_argv("strelka2_" + variant, out, normal, tumor, extra=extra(out)), env=env)
    c = _counters(p.stderr.decode())
    assert c["realign_reads"] > 1000 and c["site_loci"] > 3000 and c["indel_groups"] >= 1
    # site 9: both samples through the one somatic stream, its records serving site 5 (the somatic EVS models are loaded:
    # the tumor sample's read-position accumulators are rebuilt from the stream's window when a record is written)
    if (extra_env or {}).get("STRELKA_AMD_PILEUP") == "0":
        assert c["pileup_pushes"] == 0
    else:
        assert c["pileup_pushes"] >= 1 and c["pileup_reads"] > 1000 and c["pileup_loci"] > 3000
        assert c["pileup_genotyping"] == (0 if (extra_env or {}).get("STRELKA_AMD_PILEUP_GENOTYPE") == "0" else 1)
    files = ["somatic.snvs.vcf", "somatic.indels.vcf"] + (["somatic.callable.regions.bed"] if callable_regions else [])
    for f in files:
        want, got = E.vcf_body(ref_out + f, keep_header=True), E.vcf_body(out + f, keep_header=True)
        assert len(want) > 10
        assert got == want, f
    # ... and both equal what the reference ships (src/demo/expectedResults)
    if not callable_regions:
        for kind in ("snvs", "indels"):
            assert E.vcf_body(out + "somatic.%s.vcf" % kind) == E.vcf_body(E.demo("somatic.%s.vcf.gz" % kind))
    return c


@pytest.mark.skipif(not E.have("strelka2_ref", "strelka2_dbl"), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("windows,callable_regions", [(None, False), ((0, 0), False), ((5, 9), True), ((256, 512), False), ((500, 700), True), ((1000, 3000), True), ((3000, 5000), True)])
def test_somatic_demo_identical_through_adapter_cpu_double(tmp_path, windows, callable_regions):
    _somatic("dbl", tmp_path, windows, callable_regions)


@pytest.mark.skipif(not E.have("strelka2_ref", "strelka2_dbl"), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("env", [{"STRELKA_AMD_PILEUP": "0"}, {"STRELKA_AMD_PILEUP_GENOTYPE": "0"}, {"STRELKA_AMD_LAZY_CLEAN": "0"},
                                 {"STRELKA_AMD_PUSH_ASYNC": "0", "STRELKA_AMD_SITE_WINDOW": "900"}])
def test_somatic_demo_identical_with_reference_pileup_or_columns_only(tmp_path, env):
    """STRELKA_AMD_PUSH_ASYNC=0: the stream's pushes begun and finished in one call;
    STRELKA_AMD_PILEUP=0: the reference's pileup_read_segment, site 5 per site window from the host's copy of the columns;
    STRELKA_AMD_PILEUP_GENOTYPE=0: the stream builds the columns (and the EVS read positions), site 5 as before;
    STRELKA_AMD_LAZY_CLEAN=0: process_pos_snp_somatic builds its four cleaned pileups for every position, as the reference does"""
    _somatic("dbl", tmp_path, callable_regions=True, extra_env=env)


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("strelka2_ref", "strelka2_amd"), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("windows,callable_regions", [(None, False), ((5, 9), True)])
def test_somatic_demo_identical_through_adapter_gpu(tmp_path, windows, callable_regions):
    _somatic("amd", tmp_path, windows, callable_regions)


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("strelka2_ref", "strelka2_amd"), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("env", [{"STRELKA_AMD_PILEUP": "0"}, {"STRELKA_AMD_PILEUP_GENOTYPE": "0"}])
def test_somatic_demo_identical_with_reference_pileup_or_columns_only_gpu(tmp_path, env):
    _somatic("amd", tmp_path, callable_regions=True, extra_env=env)


# ---- larger synthetic inputs (tools/make_synth_bam.py, built by `make -C oracle ref` into oracle/_ref/synth) ------------------
# dense indels and SNV/indel clusters, reads with end-of-read indels soft-clipped or run through by the "mapper", MAPQ tiers,
# a coverage gap and a pile-up; the second set has 250 bp reads, so the position processor grows its stage geometry
# while running.  (The first version of the geometry shadow failed exactly here: the coverage gap.)

import os

SYNTH = os.path.join(E.REF_DIR, "synth")
SYNTH_SETS = {"short_reads": (SYNTH, 60000), "long_reads": (os.path.join(SYNTH, "long_reads"), 36000)}


def _have_synth():
    return all(os.path.exists(os.path.join(d, "somatic_tumor.bam")) for d, _ in SYNTH_SETS.values())


def _synth(variant, tmp_path, which, windows=None, extra_env=None):
    d, length = SYNTH_SETS[which]
    region, fa = "chrS:1-%d" % length, os.path.join(d, "synth.fa")
    env = {"STRELKA_AMD_VERBOSE": "1"}
    env.update(extra_env or {})
    if windows:
        env["STRELKA_AMD_READ_WINDOW"], env["STRELKA_AMD_SITE_WINDOW"] = str(windows[0]), str(windows[1])
    outs = {}
    for v in ("ref", variant):
        o = str(tmp_path / v) + "/"
        os.makedirs(o, exist_ok=True)
        outs[v] = o
        e = env if v != "ref" else None
        pg = E.run(E.germline_argv("starling2_" + v, o, [os.path.join(d, "germline_S1.bam"), os.path.join(d, "germline_S2.bam")],
                                   region=region, ref=fa), env=e)
        ps = E.run(E.somatic_argv("strelka2_" + v, o, os.path.join(d, "somatic_normal.bam"), os.path.join(d, "somatic_tumor.bam"),
                                  region=region, ref=fa, extra=["--somatic-callable-regions-file", o + "callable.bed"]), env=e)
        if v != "ref":
            cg, cs = _counters(pg.stderr.decode()), _counters(ps.stderr.decode())
            assert cg["realign_reads"] > 10000 and cg["indel_groups"] > 100 and cg["haplotypes"] > 100 and cg["site_recomputed"] > 100
            # site 7: a region's alternate haplotypes go through sk_global_align in one batch
            assert 0 < cg["haplotype_batches"] < cg["haplotypes"]
            assert cs["realign_reads"] > 15000 and cs["indel_groups"] > 30
            assert cg["pileup_pushes"] >= 2 and cg["pileup_reads"] > 10000 and cg["pileup_genotyping"] == 1
            assert cs["pileup_pushes"] >= 2 and cs["pileup_reads"] > 10000 and cs["pileup_genotyping"] == 1
            assert cg["feed_regions"] == 2 and cs["feed_regions"] == 2 and cg["feed_records"] > 10000 and cs["feed_records"] > 15000
            for c in (cg, cs):  # normalizeAlignment in batches, with alignments that it changes, none handed back to the reference
                assert c["feed_normalize_batches"] >= 2 and c["feed_normalized"] > 10000 and c["feed_normalize_changed"] > 100
                assert c["feed_normalize_declined"] == 0
            if (extra_env or {}).get("SK_ENUMERATION") == "2":
                for c in (cg, cs):
                    assert c["enum_device_reads"] > 500 and c["enum_host_instead"] <= c["enum_device_reads"] // 20
    n_records = 0
    for f in ("variants.vcf", "genome.S1.vcf", "genome.S2.vcf", "somatic.snvs.vcf", "somatic.indels.vcf", "callable.bed"):
        want, got = E.vcf_body(outs["ref"] + f, keep_header=True), E.vcf_body(outs[variant] + f, keep_header=True)
        assert got == want, (which, f)
        n_records += sum(1 for l in want if not l.startswith("#"))
    assert n_records > 2000


@pytest.mark.skipif(not (E.have("starling2_dbl", "strelka2_dbl") and _have_synth()), reason="oracle/_ref binaries / synthetic inputs not built")
@pytest.mark.parametrize("which", ["short_reads", "long_reads"])
def test_synthetic_identical_with_small_feed_slices(tmp_path, which):
    _synth("dbl", tmp_path, which, extra_env={"STRELKA_AMD_FEED_SLICE_BLOCKS": "2"})


@pytest.mark.skipif(not (E.have("starling2_dbl", "strelka2_dbl") and _have_synth()), reason="oracle/_ref binaries / synthetic inputs not built")
@pytest.mark.parametrize("which,windows", [("short_reads", None), ("short_reads", (3, 5)), ("short_reads", (256, 512)), ("short_reads", (500, 500)),
                                           ("short_reads", (1000, 3000)), ("short_reads", (1500, 2500)), ("short_reads", (6000, 9000)),
                                           ("long_reads", None), ("long_reads", (64, 64)), ("long_reads", (500, 900)), ("long_reads", (3000, 3000))])
def test_synthetic_identical_through_adapter_cpu_double(tmp_path, which, windows):
    _synth("dbl", tmp_path, which, windows)


@pytest.mark.gpu
@pytest.mark.skipif(not (E.have("starling2_amd", "strelka2_amd") and _have_synth()), reason="oracle/_ref binaries / synthetic inputs not built")
@pytest.mark.parametrize("which", ["short_reads", "long_reads"])
def test_synthetic_identical_through_adapter_gpu(tmp_path, which):
    _synth("amd", tmp_path, which)


@pytest.mark.gpu
@pytest.mark.skipif(not (E.have("starling2_amd", "strelka2_amd") and _have_synth()), reason="oracle/_ref binaries / synthetic inputs not built")
@pytest.mark.parametrize("which", ["short_reads", "long_reads"])
def test_synthetic_identical_with_device_enumeration(tmp_path, which):
    _synth("amd", tmp_path, which, extra_env={"SK_ENUMERATION": "2"})


# ---- the workflow's variant inputs: --candidate-indel-input-vcf (Manta's candidates in a somatic run, PY/strelkaSharedWorkflow.py:189) and
# --force-output-vcf (forced genotyping, :190).  Candidates arrive as observations that no read made (IndelData::status.notDiscoveredFromReads,
# the external-candidate flag of the realignment job's table), forced positions switch off the zero-coverage / all-reference exits of the
# site callers and make every window's cached result answer for isForcedOutput.
def _variant_inputs(tmp_path, d, length):
    import subprocess
    region, fa = "chrS:1-%d" % length, os.path.join(d, "synth.fa")
    o = str(tmp_path / "probe") + "/"
    os.makedirs(o, exist_ok=True)
    E.run(E.germline_argv("starling2_ref", o, [os.path.join(d, "germline_S1.bam"), os.path.join(d, "germline_S2.bam")], region=region, ref=fa))
    recs = [l.split("\t") for l in E.vcf_body(o + "variants.vcf") if not l.startswith("#")]
    indels = [r for r in recs if len(r[3]) != len(r[4].split(",")[0])]
    snvs = [r for r in recs if len(r[3]) == 1 and len(r[4].split(",")[0]) == 1]
    assert len(indels) > 60 and len(snvs) > 60
    head = "##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    line = lambda r: "%s\t%s\t.\t%s\t%s\t.\tPASS\t.\n" % (r[0], r[1], r[3], r[4].split(",")[0])
    with open(fa) as f:
        seq = "".join(l.strip() for l in f if not l.startswith(">"))
    cand = tmp_path / "candidates.vcf"
    cand.write_text(head + "".join(line(r) for r in indels[::3]))
    forced_recs = sorted([(int(r[1]), line(r)) for r in snvs[::5] + indels[1::7]] +
                         [(p, "chrS\t%d\t.\t%s\t%s\t.\tPASS\t.\n" % (p, seq[p - 1], "ACGT"[("ACGT".index(seq[p - 1]) + 1) % 4])) for p in range(997, length - 500, 4001)])
    forced = tmp_path / "forced.vcf"
    forced.write_text(head + "".join(l for _, l in forced_recs))
    for v in (cand, forced):
        subprocess.run([os.path.join(E.BIN_DIR, "bgzip"), "-f", str(v)], check=True)
        subprocess.run([os.path.join(E.BIN_DIR, "tabix"), "-p", "vcf", str(v) + ".gz"], check=True)
    return ["--candidate-indel-input-vcf", str(cand) + ".gz", "--force-output-vcf", str(forced) + ".gz"]


def _synth_with_variant_inputs(variant, tmp_path, env=None):
    d, length = SYNTH_SETS["short_reads"]
    region, fa = "chrS:1-%d" % length, os.path.join(d, "synth.fa")
    extra = _variant_inputs(tmp_path, d, length)
    outs = {}
    for v in ("ref", variant):
        o = str(tmp_path / v) + "/"
        os.makedirs(o, exist_ok=True)
        outs[v] = o
        e = dict({"STRELKA_AMD_VERBOSE": "1"}, **(env or {})) if v != "ref" else None
        E.run(E.germline_argv("starling2_" + v, o, [os.path.join(d, "germline_S1.bam"), os.path.join(d, "germline_S2.bam")], region=region, ref=fa,
                              extra=extra), env=e)
        E.run(E.somatic_argv("strelka2_" + v, o, os.path.join(d, "somatic_normal.bam"), os.path.join(d, "somatic_tumor.bam"), region=region, ref=fa,
                             extra=extra), env=e)
    n_forced = 0
    for f in ("variants.vcf", "genome.S1.vcf", "genome.S2.vcf", "somatic.snvs.vcf", "somatic.indels.vcf"):
        want, got = E.vcf_body(outs["ref"] + f, keep_header=True), E.vcf_body(outs[variant] + f, keep_header=True)
        assert got == want, f
        n_forced += sum(1 for l in want if not l.startswith("#"))
    # forced positions reached the outputs: the somatic SNV file reports every forced site, called or not
    assert sum(1 for l in E.vcf_body(outs["ref"] + "somatic.snvs.vcf") if not l.startswith("#")) > 20 and n_forced > 1000


@pytest.mark.skipif(not (E.have("starling2_dbl", "strelka2_dbl") and _have_synth() and os.path.exists(os.path.join(E.BIN_DIR, "tabix"))),
                    reason="oracle/_ref binaries / synthetic inputs / tabix not built")
@pytest.mark.parametrize("windows", [None, (700, 900)])
def test_candidate_indel_and_forced_output_vcfs_identical_cpu_double(tmp_path, windows):
    env = {} if windows is None else {"STRELKA_AMD_READ_WINDOW": str(windows[0]), "STRELKA_AMD_SITE_WINDOW": str(windows[1])}
    _synth_with_variant_inputs("dbl", tmp_path, env)


@pytest.mark.gpu
@pytest.mark.skipif(not (E.have("starling2_amd", "strelka2_amd") and _have_synth() and os.path.exists(os.path.join(E.BIN_DIR, "tabix"))),
                    reason="oracle/_ref binaries / synthetic inputs / tabix not built")
def test_candidate_indel_and_forced_output_vcfs_identical_gpu(tmp_path):
    _synth_with_variant_inputs("amd", tmp_path)


# ---- germline with EVS models (the workflow's default): the pileup's rank-sum accumulators rebuilt from the stream ------------------
# The reference tree does not carry its germline models; tools/make_dummy_germline_models.py writes small ones whose trees split on
# the read-position and mapping-quality rank sums.  With --report-evs-features every feature value is printed into the VCF.

def _evs_models(tmp_path):
    import subprocess
    import sys
    d = tmp_path / "models"
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "make_dummy_germline_models.py"), str(d)], check=True)
    return str(d / "germlineSNVScoringModels.json"), str(d / "germlineIndelScoringModels.json")


def _germline_evs(variant, tmp_path, extra_env=None, length=300000):
    from strelka_amd import farm
    d = E.wgs_dataset(length)
    models = _evs_models(tmp_path)
    outs = {}
    for v in ("ref", variant):
        o = tmp_path / v
        o.mkdir()
        argv = farm.germline_segment_argv("starling2_" + v, str(o) + "/", [os.path.join(d, "wgs.bam")], ["chrW:1-%d" % length],
                                          os.path.join(d, "wgs.fa"), chrom_depth=os.path.join(d, "chrom_depth.txt"), evs_models=models,
                                          report_evs_features=True)
        env = {"STRELKA_AMD_VERBOSE": "1"}
        env.update(extra_env or {})
        p = E.run(argv, env=env if v != "ref" else None, timeout=1800)
        outs[v] = ({f: E.vcf_body(str(o / f), keep_header=True) for f in ("variants.vcf", "genome.S1.vcf")}, p.stderr.decode())
    want, got = outs["ref"][0], outs[variant][0]
    records = [l for l in want["variants.vcf"] if not l.startswith("#")]
    assert len(records) > 200 and all("EVSF=" in l for l in records if "\tPASS\t" in l or "LowGQX" in l)
    # the rank-sum features are really there: not all zero
    evsf = [l.split("EVSF=")[1].split(";")[0].split("\t")[0].split(",") for l in records if "EVSF=" in l and len(l.split("\t")[3]) == 1 and len(l.split("\t")[4]) == 1]
    assert sum(1 for f in evsf if float(f[4]) != 0.0) > 50 and sum(1 for f in evsf if float(f[5]) != 0.0) > 50
    for f in want:
        assert got[f] == want[f], f
    return _counters(outs[variant][1])


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("env", [None, {"STRELKA_AMD_PILEUP": "0"}, {"STRELKA_AMD_PILEUP_GENOTYPE": "0"}])
def test_germline_with_evs_models_identical_through_adapter_cpu_double(tmp_path, env):
    c = _germline_evs("dbl", tmp_path, extra_env=env)
    if env and env.get("STRELKA_AMD_PILEUP") == "0":
        assert c["pileup_pushes"] == 0
    else:
        assert c["pileup_pushes"] >= 10 and c["pileup_loci"] > 250000  # the stream ran although the EVS accumulators are wanted


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("starling2_ref", "starling2_amd"), reason="oracle/_ref binaries not built")
def test_germline_with_evs_models_identical_through_adapter_gpu(tmp_path):
    c = _germline_evs("amd", tmp_path)
    assert c["pileup_pushes"] >= 10 and c["pileup_genotyping"] == 1


def _two_sample_evs(variant, tmp_path, which, windows=None):
    """two germline samples called jointly with EVS models (no feature report: the reference allows that for one sample only); the dense
    synthetic sets: indel clusters, MAPQ tiers, a coverage gap, long reads"""
    d, length = SYNTH_SETS[which]
    env = {"STRELKA_AMD_VERBOSE": "1"}
    if windows:
        env["STRELKA_AMD_READ_WINDOW"], env["STRELKA_AMD_SITE_WINDOW"] = str(windows[0]), str(windows[1])
    models = _evs_models(tmp_path)
    outs = {}
    for v in ("ref", variant):
        o = str(tmp_path / v) + "/"
        os.makedirs(o, exist_ok=True)
        p = E.run(E.germline_argv("starling2_" + v, o, [os.path.join(d, "germline_S1.bam"), os.path.join(d, "germline_S2.bam")],
                                  region="chrS:1-%d" % length, ref=os.path.join(d, "synth.fa"),
                                  extra=["--snv-scoring-model-file", models[0], "--indel-scoring-model-file", models[1]]),
                  env=env if v != "ref" else None)
        outs[v] = ({f: E.vcf_body(o + f, keep_header=True) for f in ("variants.vcf", "genome.S1.vcf", "genome.S2.vcf")}, p.stderr.decode())
    assert len(outs["ref"][0]["variants.vcf"]) > 100
    for f in outs["ref"][0]:
        assert outs[variant][0][f] == outs["ref"][0][f], f
    c = _counters(outs[variant][1])
    assert c["pileup_pushes"] >= 4 and c["pileup_genotyping"] == 1
    return c


@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_dbl") and _have_synth()), reason="oracle/_ref binaries / synthetic sets not built")
@pytest.mark.parametrize("which", ["short_reads", "long_reads"])
def test_two_sample_germline_with_evs_models_cpu_double(tmp_path, which):
    _two_sample_evs("dbl", tmp_path, which)


@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_dbl") and _have_synth()), reason="oracle/_ref binaries / synthetic sets not built")
@pytest.mark.parametrize("windows", [(300, 2500)])
def test_evs_words_outlive_their_output_block(tmp_path, windows):
    """POST_ALIGN several pileup windows behind READ_BUFFER: a window's EVS words are still unread when the stream's output block comes
    round again (three in rotation, strelka_amd.h SK_PILEUP_WINDOW_LIFETIME) -- the chunk takes its copy then, and the records are the
    reference's all the same"""
    c = _two_sample_evs("dbl", tmp_path, "short_reads", windows)
    assert c["pileup_evs_words_copied"] > 0


@pytest.mark.gpu
@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_amd") and _have_synth()), reason="oracle/_ref binaries / synthetic sets not built")
@pytest.mark.parametrize("which", ["short_reads", "long_reads"])
def test_two_sample_germline_with_evs_models_gpu(tmp_path, which):
    _two_sample_evs("amd", tmp_path, which)


@pytest.mark.gpu
@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_amd") and _have_synth()), reason="oracle/_ref binaries / synthetic sets not built")
def test_evs_words_outlive_their_output_block_gpu(tmp_path):
    c = _two_sample_evs("amd", tmp_path, "short_reads", (300, 2500))
    assert c["pileup_evs_words_copied"] > 0


# ---- allele groups of a multi-sample run (tools/make_multiallelic_bam.py): four samples, each heterozygous for its own pair of
# overlapping indels at the same loci -> groups of up to ploidy x samples alternate alleles (selectTopOrthogonalAllelesInAllSamples).
# Round 3's adapter handed groups of more than SK_MAX_ALT alleles back to the reference's own function; now they go through
# sk_allele_group_genotype_lhoods_wide and nothing on this path runs the reference's likelihood code.
MULTI = os.path.join(SYNTH, "multi")


def _multi_sample_wide_groups(variant, tmp_path):
    outs = {}
    bams = [os.path.join(MULTI, "multi_M%d.bam" % k) for k in (1, 2, 3, 4)]
    for v in ("ref", variant):
        o = str(tmp_path / v) + "/"
        os.makedirs(o, exist_ok=True)
        p = E.run(E.germline_argv("starling2_" + v, o, bams, region="chrS:1-24000", ref=os.path.join(MULTI, "multi.fa")),
                  env={"STRELKA_AMD_VERBOSE": "1"} if v != "ref" else None)
        outs[v] = ({f: E.vcf_body(o + f, keep_header=True) for f in ["variants.vcf"] + ["genome.S%d.vcf" % k for k in (1, 2, 3, 4)]},
                   p.stderr.decode())
    records = [l for l in outs["ref"][0]["variants.vcf"] if not l.startswith("#")]
    assert max(len(l.split("\t")[4].split(",")) for l in records) >= 6  # records with six and more alternate alleles
    for f in outs["ref"][0]:
        assert outs[variant][0][f] == outs["ref"][0][f], f
    c = _counters(outs[variant][1])
    assert c["indel_groups_wide"] >= 40 and c["indel_groups"] > c["indel_groups_wide"]
    assert "indel_groups_reference" not in c  # (the fall-back to the reference's function is gone, and its counter with it)


@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_dbl") and os.path.exists(os.path.join(MULTI, "multi_M4.bam"))),
                    reason="oracle/_ref binaries / synthetic sets not built")
def test_multi_sample_wide_allele_groups_cpu_double(tmp_path):
    _multi_sample_wide_groups("dbl", tmp_path)


@pytest.mark.gpu
@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_amd") and os.path.exists(os.path.join(MULTI, "multi_M4.bam"))),
                    reason="oracle/_ref binaries / synthetic sets not built")
def test_multi_sample_wide_allele_groups_gpu(tmp_path):
    _multi_sample_wide_groups("amd", tmp_path)


# ---- six samples (SK_MAX_SAMPLES = 8 since round 6): groups of more than eight alternate alleles go through
# sk_allele_group_genotype_lhoods_xwide (153 genotypes); every sample's gVCF and the variants VCF byte for byte
MULTI6 = os.path.join(SYNTH, "multi6")


def _six_samples(variant, tmp_path):
    if not os.path.exists(os.path.join(MULTI6, "multi_M6.bam")):  # (a tree whose synthetic sets were made before round 6)
        import subprocess
        import sys
        subprocess.run([sys.executable, os.path.join(E.REPO, "tools", "make_multiallelic_bam.py"), MULTI6, os.path.join(E.BIN_DIR, "samtools"),
                        "--samples", "6"], check=True, stdout=subprocess.DEVNULL)
    outs = {}
    bams = [os.path.join(MULTI6, "multi_M%d.bam" % k) for k in range(1, 7)]
    files = ["variants.vcf"] + ["genome.S%d.vcf" % k for k in range(1, 7)]
    for v in ("ref", variant):
        o = str(tmp_path / v) + "/"
        os.makedirs(o, exist_ok=True)
        p = E.run(E.germline_argv("starling2_" + v, o, bams, region="chrS:1-24000", ref=os.path.join(MULTI6, "multi.fa")),
                  env={"STRELKA_AMD_VERBOSE": "1"} if v != "ref" else None)
        outs[v] = ({f: E.vcf_body(o + f, keep_header=True) for f in files}, p.stderr.decode())
    records = [l for l in outs["ref"][0]["variants.vcf"] if not l.startswith("#")]
    assert max(len(l.split("\t")[4].split(",")) for l in records) >= 10  # records with ten and more alternate alleles
    for f in files:
        assert outs[variant][0][f] == outs["ref"][0][f], f
    c = _counters(outs[variant][1])
    assert c["indel_groups_xwide"] >= 10 and c["indel_groups_wide"] > c["indel_groups_xwide"]


@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_dbl") and os.path.exists(os.path.join(E.BIN_DIR, "samtools"))),
                    reason="oracle/_ref binaries not built")
def test_six_samples_and_their_allele_groups_cpu_double(tmp_path):
    _six_samples("dbl", tmp_path)


@pytest.mark.gpu
@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_amd") and os.path.exists(os.path.join(MULTI6, "multi_M6.bam"))),
                    reason="oracle/_ref binaries / synthetic sets not built")
def test_six_samples_and_their_allele_groups_gpu(tmp_path):
    _six_samples("amd", tmp_path)


# ---- site 10 with several samples: a position that is a plain site of EVERY sample's window goes from the windows into each sample's
# open block (the writer's own loop over the samples); the variants VCF and every sample's gVCF byte for byte
def _two_sample_gvcf(variant, tmp_path):
    d, length = SYNTH_SETS["short_reads"]
    bams = [os.path.join(d, "germline_S1.bam"), os.path.join(d, "germline_S2.bam")]
    outs, err = {}, None
    for v in ("ref", variant):
        o = str(tmp_path / v) + "/"
        os.makedirs(o, exist_ok=True)
        p = E.run(E.germline_argv("starling2_" + v, o, bams, region="chrS:1-%d" % length, ref=os.path.join(d, "synth.fa")),
                  env={"STRELKA_AMD_VERBOSE": "1"} if v != "ref" else None)
        outs[v] = {f: E.vcf_body(o + f, keep_header=True) for f in ("variants.vcf", "genome.S1.vcf", "genome.S2.vcf")}
        if v != "ref":
            err = p.stderr.decode()
    for f in outs["ref"]:
        assert len(outs["ref"][f]) > 100 and outs[variant][f] == outs["ref"][f], f
    g = _gvcf_counters(err)
    assert g["gvcf_plain_sites"] > 0.9 * length and g["gvcf_reference_sites"] < 0.05 * length and g["gvcf_blocks_installed"] == 0, g


@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_dbl") and _have_synth()), reason="oracle/_ref binaries / synthetic sets not built")
def test_two_sample_gvcf_sites_come_from_the_streams_cpu_double(tmp_path):
    _two_sample_gvcf("dbl", tmp_path)


@pytest.mark.gpu
@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_amd") and _have_synth()), reason="oracle/_ref binaries / synthetic sets not built")
def test_two_sample_gvcf_sites_come_from_the_streams_gpu(tmp_path):
    _two_sample_gvcf("amd", tmp_path)


# ---- the ninth caller process of a device says so (the driver runs eight compute processes side by side and time-slices the rest:
# INTEGRATION.md "How many caller processes per GPU"): every process holds one of eight advisory file locks for its lifetime
@pytest.mark.skipif(not E.have("starling2_dbl"), reason="oracle/_ref binaries not built")
def test_a_ninth_caller_process_on_a_device_is_told(tmp_path):
    import fcntl
    slot_dir = tmp_path / "slots"
    slot_dir.mkdir()
    env = {"STRELKA_AMD_SLOT_DIR": str(slot_dir)}

    def run():
        o = str(tmp_path / "o") + "/"
        os.makedirs(o, exist_ok=True)
        return E.run(E.germline_argv("starling2_dbl", o, [E.demo("NA12891_demo20.bam")]), env=env).stderr.decode()
    assert "caller processes on device" not in run()
    held = []
    for i in range(8):  # (eight other processes' slots)
        f = open(str(slot_dir / ("strelka_amd_device0_slot%d.lock" % i)), "a+")
        fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)
        held.append(f)
    assert "more than 8 caller processes on device 0" in run()
    held[3].close()  # (one of them exits)
    assert "caller processes on device" not in run()
    for f in held:
        f.close()


# ---- reads longer than the device pileup takes: refused when they arrive, with the way out in the message
@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
def test_reads_over_the_pileup_limit_are_refused_with_the_way_out(tmp_path):
    import subprocess
    import sys
    d = str(tmp_path / "long")
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "make_synth_bam.py"), d,
                    os.path.join(E.BIN_DIR, "samtools"), "--seed", "5", "--length", "9000", "--read-length", "1200", "--spacing", "600"],
                   check=True, stdout=subprocess.DEVNULL)
    outs = {}
    for v in ("ref", "dbl"):
        o = str(tmp_path / v) + "/"
        os.makedirs(o, exist_ok=True)
        argv = E.germline_argv("starling2_" + v, o, [os.path.join(d, "germline_S1.bam")], region="chrS:1-9000", ref=os.path.join(d, "synth.fa"))
        if v == "dbl":
            with pytest.raises(RuntimeError, match="at most 1024 bases; rerun with STRELKA_AMD_PILEUP=0"):
                E.run(argv)
            E.run(argv, env={"STRELKA_AMD_PILEUP": "0"})
        else:
            E.run(argv)
        outs[v] = E.vcf_body(o + "variants.vcf", keep_header=True)
    assert sum(1 for l in outs["ref"] if not l.startswith("#")) >= 5
    assert outs["dbl"] == outs["ref"]


# ---- site 10: the gVCF writer's non-variant blocks fed from the stream's window (adapter/sk_adapter_gvcf.cpp) ---------------------
# The fast path is taken for ONE sample (several samples keep the reference's path), so the two-sample runs above never reach it.  One
# sample of each dense synthetic set -- indel clusters (positions inside a called deletion have their ploidy lowered: declined by
# state; variant indels buffered in the overlap resolver; the phaser's open active regions), a coverage gap (the writer's skip_to_pos
# fills it between two injected sites), soft clips, MAPQ tiers, 250 bp reads -- alone and with what changes the writer's decisions
# from outside: forced-output positions, external candidate indels, a ploidy VCF with haploid and zero-ploidy stretches, a
# no-compress BED.  Everything the reference writes must come out the same; the counters say the path was the routed one.
def _gvcf_counters(stderr):
    m = re.search(r"strelka_amd adapter gvcf: (.*)", stderr)
    return {k: int(v) for k, v in (kv.split("=") for kv in m.group(1).split())} if m else {}


def _single_sample(variant, tmp_path, which, sample="germline_S1.bam", extra=(), env=None, min_plain=0.7):
    d, length = SYNTH_SETS[which]
    region, fa = "chrS:1-%d" % length, os.path.join(d, "synth.fa")
    outs, err = {}, None
    for v in ("ref", variant):
        o = str(tmp_path / v) + "/"
        os.makedirs(o, exist_ok=True)
        outs[v] = o
        e = dict({"STRELKA_AMD_VERBOSE": "1"}, **(env or {})) if v != "ref" else None
        p = E.run(E.germline_argv("starling2_" + v, o, [os.path.join(d, sample)], region=region, ref=fa, extra=list(extra)), env=e)
        if v != "ref":
            err = p.stderr.decode()
    for f in ("variants.vcf", "genome.S1.vcf"):
        want, got = E.vcf_body(outs["ref"] + f, keep_header=True), E.vcf_body(outs[variant] + f, keep_header=True)
        assert len(want) > 100
        assert got == want, (which, sample, f)
    g = _gvcf_counters(err)
    if (env or {}).get("STRELKA_AMD_GVCF_FAST") == "0":
        assert g.get("gvcf_plain_sites", 0) == 0
    else:
        routed = g["gvcf_plain_sites"] + g["gvcf_block_sites"]
        covered = routed + g["gvcf_reference_sites"]
        assert covered > 0.5 * length and routed >= min_plain * covered and g["gvcf_filter_key_mismatches"] == 0, g
        if (env or {}).get("STRELKA_AMD_GVCF_BLOCKS") == "0":
            assert g["gvcf_blocks_installed"] == 0
        else:
            assert g["gvcf_blocks_installed"] > 100 and g["gvcf_block_sites"] > g["gvcf_blocks_installed"], g
    return g, outs


def _ploidy_and_nocompress(tmp_path, length):
    """a ploidy VCF (haploid and zero-ploidy stretches, PY/strelkaGermlineWorkflow.py:131-132) and a no-compress BED (:128-129)"""
    import subprocess
    pv = tmp_path / "ploidy.vcf"
    rows = [(3001, 5000, 1), (9001, 9500, 0), (20001, 26000, 1), (length - 4000, length - 3000, 0)]
    pv.write_text("##fileformat=VCFv4.1\n##FORMAT=<ID=CN,Number=1,Type=Integer,Description=\"copy number\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n" +
                  "".join("chrS\t%d\t.\tN\t<CNV>\t.\tPASS\tEND=%d\tCN\t%d\n" % (b, e, cn) for b, e, cn in rows))
    bed = tmp_path / "nocompress.bed"
    bed.write_text("chrS\t1500\t1700\nchrS\t12000\t12040\nchrS\t30000\t30900\n")
    for v, preset in ((pv, "vcf"), (bed, "bed")):
        subprocess.run([os.path.join(E.BIN_DIR, "bgzip"), "-f", str(v)], check=True)
        subprocess.run([os.path.join(E.BIN_DIR, "tabix"), "-p", preset, str(v) + ".gz"], check=True)
    return ["--ploidy-region-vcf", str(pv) + ".gz", "--nocompress-bed", str(bed) + ".gz"]


@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_dbl") and _have_synth()), reason="oracle/_ref binaries / synthetic sets not built")
@pytest.mark.parametrize("which,sample,windows", [("short_reads", "germline_S1.bam", None), ("short_reads", "germline_S2.bam", (500, 0)),
                                                  ("short_reads", "germline_S1.bam", (1500, 100)), ("long_reads", "germline_S1.bam", None),
                                                  ("long_reads", "germline_S2.bam", (64, 0))])
def test_single_sample_gvcf_blocks_from_the_window_cpu_double(tmp_path, which, sample, windows):
    env = {} if windows is None else {"STRELKA_AMD_READ_WINDOW": str(windows[0]), "STRELKA_AMD_SITE_WINDOW": str(windows[1])}
    g, _ = _single_sample("dbl", tmp_path, which, sample, env=env)
    assert g["gvcf_plain_declined_by_state"] > 50  # positions under called deletions / beside buffered variant indels took the reference's path


@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_dbl") and _have_synth()), reason="oracle/_ref binaries / synthetic sets not built")
@pytest.mark.parametrize("env", [{"STRELKA_AMD_GVCF_FAST": "0"}, {"STRELKA_AMD_GVCF_BLOCKS": "0"}])
def test_single_sample_gvcf_fast_path_switched_off(tmp_path, env):
    """without site 10 altogether, and with its plain sites going into the writer one by one (no whole blocks)"""
    _single_sample("dbl", tmp_path, "short_reads", env=env)


@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_dbl") and _have_synth() and os.path.exists(os.path.join(E.BIN_DIR, "tabix"))),
                    reason="oracle/_ref binaries / synthetic inputs / tabix not built")
@pytest.mark.parametrize("inputs", ["forced_and_candidates", "ploidy_and_nocompress", "all"])
def test_single_sample_gvcf_blocks_with_outside_inputs_cpu_double(tmp_path, inputs):
    d, length = SYNTH_SETS["short_reads"]
    extra = []
    if inputs in ("forced_and_candidates", "all"):
        extra += _variant_inputs(tmp_path, d, length)
    if inputs in ("ploidy_and_nocompress", "all"):
        extra += _ploidy_and_nocompress(tmp_path, length)
    g, outs = _single_sample("dbl", tmp_path, "short_reads", extra=extra, min_plain=0.5)
    if inputs != "forced_and_candidates":
        body = "\n".join(E.vcf_body(outs["ref"] + "genome.S1.vcf"))
        assert "PloidyConflict" in body or "\t.:" in body or "\t0:" in body  # (the haploid / zero-ploidy stretches reached the output)


@pytest.mark.gpu
@pytest.mark.skipif(not (E.have("starling2_ref", "starling2_amd") and _have_synth() and os.path.exists(os.path.join(E.BIN_DIR, "tabix"))),
                    reason="oracle/_ref binaries / synthetic inputs / tabix not built")
@pytest.mark.parametrize("which,inputs", [("short_reads", None), ("long_reads", None), ("short_reads", "all")])
def test_single_sample_gvcf_blocks_from_the_window_gpu(tmp_path, which, inputs):
    d, length = SYNTH_SETS[which]
    extra = (_variant_inputs(tmp_path, d, length) + _ploidy_and_nocompress(tmp_path, length)) if inputs else []
    _single_sample("amd", tmp_path, which, extra=extra, min_plain=0.5)


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
def test_single_sample_gvcf_blocks_with_regions_out_of_order(tmp_path):
    """several --region of one process, the later one LOWER on the chromosome than the earlier one (as a second chromosome's would be):
    nothing of a block installed in one region may carry into the next"""
    from strelka_amd import farm
    d = E.wgs_dataset(400000)
    outs, err = {}, None
    for v in ("ref", "dbl"):
        o = str(tmp_path / v) + "/"
        os.makedirs(o, exist_ok=True)
        outs[v] = o
        argv = farm.germline_segment_argv("starling2_" + v, o, [os.path.join(d, "wgs.bam")], ["chrW:250001-400000", "chrW:1-120000"], os.path.join(d, "wgs.fa"),
                                          chrom_depth=os.path.join(d, "chrom_depth.txt"))
        p = E.run(argv, env={"STRELKA_AMD_VERBOSE": "1"} if v != "ref" else None, timeout=1800)
        if v != "ref":
            err = p.stderr.decode()
    for f in ("variants.vcf", "genome.S1.vcf"):
        want, got = E.vcf_body(outs["ref"] + f, keep_header=True), E.vcf_body(outs["dbl"] + f, keep_header=True)
        assert len(want) > 100 and got == want, f
    g = _gvcf_counters(err)
    assert g["gvcf_blocks_installed"] > 1000 and g["gvcf_block_sites"] > 150000, g
