"""The drop-in against the reference on fresh seeded WGS-like samples (build container: the CPU double of the C-ABI, `starling2_dbl`; on the
GPU box pass variant=amd): depth, variant density, the way the run is cut into regions and their order vary with the seed; both VCFs must
be the reference's byte for byte.  The routed gVCF path (site 10: plain sites and whole blocks from the device's window) sees shallow and
deep samples, regions that start inside blocks, regions called out of order.

usage: python tools/fuzz/e2e_seeds.py [n_seeds=16] [first_seed=1] [variant=dbl] [workers=8] [somatic|multi|adversarial]

`adversarial` aims at site 10 (whole gVCF blocks installed from the device's runs, adapter/sk_adapter_gvcf.cpp): quiet samples -- long runs
of plain sites, many installed blocks -- with everything that must break a block placed INSIDE such runs: candidate indels from a VCF that
no read supports (--candidate-indel-input-vcf), forced-output SNV and indel records (--force-output-vcf), ploidy regions and no-compress
regions that begin and end mid-run, a read buffer small enough to drop reads (--max-sample-read-buffer), regions cut inside runs and called
out of order.  A run counts only if blocks were installed (the drop-in's own counter, STRELKA_AMD_VERBOSE)."""
import os
import random
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, ".")
from tests import e2e_util as E


def one(seed, variant, models):
    rng = random.Random(9000 + seed)
    length = rng.choice([120000, 200000, 300000])
    depth = rng.choice([4.0, 12.0, 25.0, 40.0, 70.0])
    snv_every = rng.choice([150, 1000, 4000])
    indel_every = rng.choice([400, 3000, 20000])
    read_length = 150
    if os.environ.get("SK_FUZZ_HARD"):  # other read lengths, indels every hundred-odd bases, deeper samples
        hard = random.Random(39000 + seed)
        read_length = hard.choice([100, 150, 250])
        indel_every = hard.choice([120, 250, indel_every])
        depth = hard.choice([depth, 100.0])
        length = min(length, 160000)
    d = os.path.join(E.REPO, "oracle", "_ref", "synth", "fuzz_%d" % seed)
    if not os.path.exists(os.path.join(d, "chrom_depth.txt")):
        os.makedirs(d, exist_ok=True)
        subprocess.run([sys.executable, "tools/make_wgs_bam.py", d, os.path.join(E.BIN_DIR, "samtools"), "--length", str(length),
                        "--depth", str(depth), "--seed", str(seed), "--snv-every", str(snv_every), "--indel-every", str(indel_every),
                        "--read-length", str(read_length), "--procs", "1"], check=True, stdout=subprocess.DEVNULL)
        with open(os.path.join(d, "chrom_depth.txt"), "w") as f:
            f.write("chrW\t%.3f\n" % depth)
    # the process's regions: one to four pieces of the sample, sometimes with a gap, sometimes out of order
    cuts = sorted(rng.sample(range(1000, length - 1000), rng.choice([0, 1, 2, 3])))
    edges = [1] + cuts + [length + 1]
    regions = []
    for a, b in zip(edges[:-1], edges[1:]):
        gap = rng.choice([0, 0, 37, 1500])
        if b - gap > a:
            regions.append("chrW:%d-%d" % (a, b - 1 - gap))
    if rng.random() < 0.3:
        rng.shuffle(regions)
    extra = list(models) if rng.random() < 0.7 else []
    # inputs from outside, at random places: a ploidy VCF with haploid and zero-ploidy stretches, a no-compress BED, forced-output positions
    if rng.random() < 0.5:
        rows, at = [], 2000
        while at < length - 5000 and len(rows) < 6:
            at += rng.randrange(3000, 60000)
            n = rng.choice([1, 40, 700, 5000])
            if at + n < length:
                rows.append((at, at + n, rng.choice([0, 1, 1])))
            at += n
        pv = os.path.join(d, "ploidy_%d.vcf" % seed)
        with open(pv, "w") as f:
            f.write("##fileformat=VCFv4.1\n##FORMAT=<ID=CN,Number=1,Type=Integer,Description=\"copy number\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tNA_SYNTH\n" +
                    "".join("chrW\t%d\t.\tN\t<CNV>\t.\tPASS\tEND=%d\tCN\t%d\n" % r for r in rows))
        bed = os.path.join(d, "nocompress_%d.bed" % seed)
        with open(bed, "w") as f:
            at = 500
            for _ in range(rng.choice([1, 3, 8])):
                at += rng.randrange(1000, 40000)
                n = rng.choice([1, 30, 400, 3000])
                if at + n < length:
                    f.write("chrW\t%d\t%d\n" % (at, at + n))
                at += n
        for v, preset in ((pv, "vcf"), (bed, "bed")):
            subprocess.run([os.path.join(E.BIN_DIR, "bgzip"), "-f", v], check=True)
            subprocess.run([os.path.join(E.BIN_DIR, "tabix"), "-f", "-p", preset, v + ".gz"], check=True)
        extra += ["--ploidy-region-vcf", pv + ".gz", "--nocompress-bed", bed + ".gz"]
    if rng.random() < 0.4:
        fa = open(os.path.join(d, "wgs.fa")).read().split("\n", 1)[1].replace("\n", "")
        fv = os.path.join(d, "forced_%d.vcf" % seed)
        with open(fv, "w") as f:
            f.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
            for pos in sorted(rng.sample(range(100, length - 100), rng.choice([3, 40, 400]))):
                ref = fa[pos - 1].upper()
                if ref in "ACGT":
                    f.write("chrW\t%d\t.\t%s\t%s\t.\t.\t.\n" % (pos, ref, rng.choice([b for b in "ACGT" if b != ref])))
        subprocess.run([os.path.join(E.BIN_DIR, "bgzip"), "-f", fv], check=True)
        subprocess.run([os.path.join(E.BIN_DIR, "tabix"), "-f", "-p", "vcf", fv + ".gz"], check=True)
        extra += ["--force-output-vcf", fv + ".gz"]
    out = {}
    for binary in ("starling2_ref", "starling2_" + variant):
        with tempfile.TemporaryDirectory() as o:
            E.run(E.germline_wgs_argv(os.path.basename(binary), o + "/", [os.path.join(d, "wgs.bam")], regions, os.path.join(d, "wgs.fa"),
                                      os.path.join(d, "chrom_depth.txt"), extra=extra), timeout=3600)
            out[binary] = {f: E.vcf_body(os.path.join(o, f), keep_header=True) for f in ("variants.vcf", "genome.S1.vcf")}
    want, got = out["starling2_ref"], out["starling2_" + variant]
    if not os.environ.get("SK_FUZZ_KEEP"):
        shutil.rmtree(d, ignore_errors=True)  # (a sample is 5-20 MB; oracle/_ref travels to the GPU box)
    what = "seed %d: %d bp at %gx, %d bp reads, snv/%d indel/%d, regions %s%s" % (seed, length, depth, read_length, snv_every, indel_every, ",".join(regions),
                                                                     "".join(" " + x for x in extra if x.startswith("--")))
    for f in want:
        if want[f] != got[f]:
            k = next((i for i, (x, y) in enumerate(zip(want[f], got[f])) if x != y), min(len(want[f]), len(got[f])))
            return False, "%s: %s differs at line %d\n  reference: %s\n  drop-in:   %s" % (
                what, f, k + 1, want[f][k] if k < len(want[f]) else "<end>", got[f][k] if k < len(got[f]) else "<end>")
    return True, "%s: identical (%d variant records, %d gVCF lines)" % (
        what, sum(1 for l in want["variants.vcf"] if l[0] != "#"), len(want["genome.S1.vcf"]))


def one_adversarial(seed, variant, models):
    import re
    rng = random.Random(49000 + seed)
    length = rng.choice([100000, 160000, 240000])
    depth = rng.choice([12.0, 25.0, 40.0])
    snv_every, indel_every = rng.choice([2000, 8000]), rng.choice([8000, 40000])
    d = os.path.join(E.REPO, "oracle", "_ref", "synth", "fuzz_adv_%d" % seed)
    if not os.path.exists(os.path.join(d, "chrom_depth.txt")):
        os.makedirs(d, exist_ok=True)
        subprocess.run([sys.executable, "tools/make_wgs_bam.py", d, os.path.join(E.BIN_DIR, "samtools"), "--length", str(length), "--depth", str(depth),
                        "--seed", str(70000 + seed), "--snv-every", str(snv_every), "--indel-every", str(indel_every), "--procs", "1"],
                       check=True, stdout=subprocess.DEVNULL)
        with open(os.path.join(d, "chrom_depth.txt"), "w") as f:
            f.write("chrW\t%.3f\n" % depth)
    fa = open(os.path.join(d, "wgs.fa")).read().split("\n", 1)[1].replace("\n", "").upper()

    def vcf(path, rows, header_extra=""):
        with open(path, "w") as f:
            f.write("##fileformat=VCFv4.1\n" + header_extra + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
            for pos, ref, alt in sorted(rows):
                f.write("chrW\t%d\t.\t%s\t%s\t.\t.\t.\n" % (pos, ref, alt))
        subprocess.run([os.path.join(E.BIN_DIR, "bgzip"), "-f", path], check=True)
        subprocess.run([os.path.join(E.BIN_DIR, "tabix"), "-f", "-p", "vcf", path + ".gz"], check=True)
        return path + ".gz"

    def indel_row(pos):
        # a left-anchored deletion or insertion of 1-12 bases at `pos` (1-based anchor), or None where the reference has an N
        n = rng.choice([1, 1, 2, 3, 5, 8, 12])
        seg = fa[pos - 1:pos + n]
        if len(seg) < n + 1 or any(c not in "ACGT" for c in seg):
            return None
        # (normalised records only -- the reference rejects a forced record that could be shifted left: the last base of the deleted /
        # inserted sequence differs from the anchor base)
        if rng.random() < 0.5:
            return (pos, seg, seg[0]) if seg[-1] != seg[0] else None
        ins = "".join(rng.choice("ACGT") for _ in range(n - 1)) + rng.choice([b for b in "ACGT" if b != seg[0]])
        return (pos, seg[0], seg[0] + ins)
    extra = list(models) if rng.random() < 0.7 else []
    what = []
    if rng.random() < 0.8:
        rows = [r for r in (indel_row(p) for p in rng.sample(range(200, length - 200), rng.choice([20, 120, 400]))) if r]
        extra += ["--candidate-indel-input-vcf", vcf(os.path.join(d, "cand_%d.vcf" % seed), rows)]
        what.append("%d candidate indels" % len(rows))
    if rng.random() < 0.8:
        rows = []
        for p in rng.sample(range(200, length - 200), rng.choice([30, 200, 800])):
            if rng.random() < 0.3:
                r = indel_row(p)
                if r:
                    rows.append(r)
            elif fa[p - 1] in "ACGT":
                rows.append((p, fa[p - 1], rng.choice([b for b in "ACGT" if b != fa[p - 1]])))
        extra += ["--force-output-vcf", vcf(os.path.join(d, "forced_%d.vcf" % seed), rows)]
        what.append("%d forced records" % len(rows))
    if rng.random() < 0.7:
        rows, at = [], 500
        while at < length - 3000:
            at += rng.randrange(400, 12000)
            n = rng.choice([1, 2, 17, 120, 900])
            if at + n < length:
                rows.append((at, at + n, rng.choice([0, 1, 1])))
            at += n
        pv = os.path.join(d, "ploidy_%d.vcf" % seed)
        with open(pv, "w") as f:
            f.write("##fileformat=VCFv4.1\n##FORMAT=<ID=CN,Number=1,Type=Integer,Description=\"copy number\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tNA_SYNTH\n" +
                    "".join("chrW\t%d\t.\tN\t<CNV>\t.\tPASS\tEND=%d\tCN\t%d\n" % r for r in rows))
        subprocess.run([os.path.join(E.BIN_DIR, "bgzip"), "-f", pv], check=True)
        subprocess.run([os.path.join(E.BIN_DIR, "tabix"), "-f", "-p", "vcf", pv + ".gz"], check=True)
        extra += ["--ploidy-region-vcf", pv + ".gz"]
        what.append("%d ploidy regions" % len(rows))
    if rng.random() < 0.7:
        bed, n_bed = os.path.join(d, "nocompress_%d.bed" % seed), 0
        with open(bed, "w") as f:
            at = 300
            while at < length - 2000:
                at += rng.randrange(300, 9000)
                n = rng.choice([1, 3, 40, 500])
                if at + n < length:
                    f.write("chrW\t%d\t%d\n" % (at, at + n))
                    n_bed += 1
                at += n
        subprocess.run([os.path.join(E.BIN_DIR, "bgzip"), "-f", bed], check=True)
        subprocess.run([os.path.join(E.BIN_DIR, "tabix"), "-f", "-p", "bed", bed + ".gz"], check=True)
        extra += ["--nocompress-bed", bed + ".gz"]
        what.append("%d no-compress regions" % n_bed)
    if rng.random() < 0.4:
        cap = rng.choice([150, 400, 1500])
        extra += ["--max-sample-read-buffer", str(cap)]
        what.append("read buffer %d" % cap)
    cuts = sorted(rng.sample(range(1000, length - 1000), rng.choice([0, 1, 2, 4])))
    edges = [1] + cuts + [length + 1]
    regions = []
    for a, b in zip(edges[:-1], edges[1:]):
        gap = rng.choice([0, 0, 1, 37])
        if b - gap > a:
            regions.append("chrW:%d-%d" % (a, b - 1 - gap))
    if rng.random() < 0.4:
        rng.shuffle(regions)
    out, installed = {}, -1
    for binary in ("starling2_ref", "starling2_" + variant):
        with tempfile.TemporaryDirectory() as o:
            p = subprocess.run(E.germline_wgs_argv(os.path.basename(binary), o + "/", [os.path.join(d, "wgs.bam")], regions, os.path.join(d, "wgs.fa"),
                                                   os.path.join(d, "chrom_depth.txt"), extra=extra), env=dict(os.environ, STRELKA_AMD_VERBOSE="1"),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=3600)
            if p.returncode != 0:
                return False, "adversarial seed %d: %s failed (%d): %s" % (seed, binary, p.returncode, p.stderr.decode(errors="replace")[-600:])
            m = re.search(r"gvcf_blocks_installed=(\d+)", p.stderr.decode(errors="replace"))
            if m and not binary.endswith("_ref"):
                installed = int(m.group(1))
            out[binary] = {f: E.vcf_body(os.path.join(o, f), keep_header=True) for f in ("variants.vcf", "genome.S1.vcf")}
    want, got = out["starling2_ref"], out["starling2_" + variant]
    if not os.environ.get("SK_FUZZ_KEEP"):
        shutil.rmtree(d, ignore_errors=True)
    desc = "adversarial seed %d: %d bp at %gx, %s, regions %s, %d blocks installed" % (seed, length, depth, ", ".join(what) or "nothing from outside", ",".join(regions), installed)
    for f in want:
        if want[f] != got[f]:
            k = next((i for i, (x, y) in enumerate(zip(want[f], got[f])) if x != y), min(len(want[f]), len(got[f])))
            return False, "%s: %s differs at line %d\n  reference: %s\n  drop-in:   %s" % (
                desc, f, k + 1, want[f][k] if k < len(want[f]) else "<end>", got[f][k] if k < len(got[f]) else "<end>")
    if installed <= 0:
        return False, desc + ": identical, but NO block was installed (the run does not count)"
    return True, desc + ": identical (%d variant records, %d gVCF lines)" % (sum(1 for l in want["variants.vcf"] if l[0] != "#"), len(want["genome.S1.vcf"]))


def one_somatic(seed, variant):
    """a tumour / normal pair: depths, variant densities, the tumour clones' share and the region cuts from the seed; the somatic workflow's
    command line (EVS models, callable regions, the chromosome depth filter on or off)"""
    from strelka_amd import farm
    rng = random.Random(19000 + seed)
    length = rng.choice([100000, 160000, 240000])
    nd, td = rng.choice([(20.0, 40.0), (40.0, 110.0), (30.0, 60.0), (60.0, 60.0), (8.0, 25.0)])
    snv_every, indel_every, som_every = rng.choice([150, 1000]), rng.choice([400, 3000]), rng.choice([2000, 20000])
    clone = rng.choice([0.1, 0.3, 0.6])
    d = os.path.join(E.REPO, "oracle", "_ref", "synth", "fuzz_som_%d" % seed)
    if not os.path.exists(os.path.join(d, "chrom_depth.txt")):
        os.makedirs(d, exist_ok=True)
        for role, depth, name in (("normal", nd, "normal"), ("tumor", td, "tumor")):
            subprocess.run([sys.executable, "tools/make_wgs_bam.py", d, os.path.join(E.BIN_DIR, "samtools"), "--length", str(length), "--depth", str(depth),
                            "--seed", str(seed), "--procs", "1", "--role", role, "--name", name, "--sample", name.upper(), "--snv-every", str(snv_every),
                            "--indel-every", str(indel_every), "--somatic-every", str(som_every), "--clone-fraction", str(clone)],
                           check=True, stdout=subprocess.DEVNULL)
        with open(os.path.join(d, "chrom_depth.txt"), "w") as f:
            f.write("chrW\t%.3f\n" % nd)
    cuts = sorted(rng.sample(range(1000, length - 1000), rng.choice([0, 1, 2])))
    edges = [1] + cuts + [length + 1]
    regions = []
    for a, b in zip(edges[:-1], edges[1:]):
        gap = rng.choice([0, 0, 37, 1500])
        if b - gap > a:
            regions.append("chrW:%d-%d" % (a, b - 1 - gap))
    if rng.random() < 0.3:
        rng.shuffle(regions)
    callable_regions, depth_filter = rng.random() < 0.5, rng.random() < 0.7
    outputs = ["somatic.snvs.vcf", "somatic.indels.vcf"] + (["somatic.callable.regions.bed"] if callable_regions else [])
    out = {}
    for binary in ("strelka2_ref", "strelka2_" + variant):
        with tempfile.TemporaryDirectory() as o:
            E.run(farm.somatic_segment_argv(binary, o + "/", os.path.join(d, "normal.bam"), os.path.join(d, "tumor.bam"), regions, os.path.join(d, "normal.fa"),
                                            chrom_depth=os.path.join(d, "chrom_depth.txt") if depth_filter else None, callable_regions=callable_regions),
                  timeout=3600)
            out[binary] = {f: E.vcf_body(os.path.join(o, f), keep_header=True) for f in outputs}
    want, got = out["strelka2_ref"], out["strelka2_" + variant]
    if not os.environ.get("SK_FUZZ_KEEP"):
        shutil.rmtree(d, ignore_errors=True)
    what = "somatic seed %d: %d bp at %gx / %gx, snv/%d indel/%d somatic/%d clones %g, regions %s%s%s" % (
        seed, length, nd, td, snv_every, indel_every, som_every, clone, ",".join(regions), " callable-regions" if callable_regions else "",
        " depth-filter" if depth_filter else "")
    for f in want:
        if want[f] != got[f]:
            k = next((i for i, (x, y) in enumerate(zip(want[f], got[f])) if x != y), min(len(want[f]), len(got[f])))
            return False, "%s: %s differs at line %d\n  reference: %s\n  drop-in:   %s" % (
                what, f, k + 1, want[f][k] if k < len(want[f]) else "<end>", got[f][k] if k < len(got[f]) else "<end>")
    return True, "%s: identical (%d + %d records)" % (what, sum(1 for l in want["somatic.snvs.vcf"] if l[0] != "#"),
                                                     sum(1 for l in want["somatic.indels.vcf"] if l[0] != "#"))


def one_multi(seed, variant, models):
    """a joint germline run over two samples that share the reference and the germline variants (the pair generator's normal and tumour:
    the second sample carries extra variants of its own at a clone's share of its reads), depths and region cuts from the seed: the
    variants VCF and both samples' gVCFs"""
    rng = random.Random(29000 + seed)
    length = rng.choice([100000, 160000, 240000])
    d1, d2 = rng.choice([(30.0, 30.0), (40.0, 15.0), (8.0, 50.0), (20.0, 70.0)])
    snv_every, indel_every, som_every = rng.choice([150, 1000]), rng.choice([400, 3000]), rng.choice([500, 5000])
    clone = rng.choice([0.3, 0.5, 0.9])
    d = os.path.join(E.REPO, "oracle", "_ref", "synth", "fuzz_multi_%d" % seed)
    if not os.path.exists(os.path.join(d, "chrom_depth.txt")):
        os.makedirs(d, exist_ok=True)
        for role, depth, name in (("normal", d1, "normal"), ("tumor", d2, "tumor")):
            subprocess.run([sys.executable, "tools/make_wgs_bam.py", d, os.path.join(E.BIN_DIR, "samtools"), "--length", str(length), "--depth", str(depth),
                            "--seed", str(seed), "--procs", "1", "--role", role, "--name", name, "--sample", name.upper(), "--snv-every", str(snv_every),
                            "--indel-every", str(indel_every), "--somatic-every", str(som_every), "--clone-fraction", str(clone)],
                           check=True, stdout=subprocess.DEVNULL)
        with open(os.path.join(d, "chrom_depth.txt"), "w") as f:
            f.write("chrW\t%.3f\n" % max(d1, d2))
    cuts = sorted(rng.sample(range(1000, length - 1000), rng.choice([0, 1, 2])))
    edges = [1] + cuts + [length + 1]
    regions = []
    for a, b in zip(edges[:-1], edges[1:]):
        gap = rng.choice([0, 0, 37, 1500])
        if b - gap > a:
            regions.append("chrW:%d-%d" % (a, b - 1 - gap))
    if rng.random() < 0.3:
        rng.shuffle(regions)
    extra = list(models) if rng.random() < 0.7 else []
    outputs = ("variants.vcf", "genome.S1.vcf", "genome.S2.vcf")
    out = {}
    for binary in ("starling2_ref", "starling2_" + variant):
        with tempfile.TemporaryDirectory() as o:
            E.run(E.germline_wgs_argv(binary, o + "/", [os.path.join(d, "normal.bam"), os.path.join(d, "tumor.bam")], regions, os.path.join(d, "normal.fa"),
                                      os.path.join(d, "chrom_depth.txt"), extra=extra), timeout=3600)
            out[binary] = {f: E.vcf_body(os.path.join(o, f), keep_header=True) for f in outputs}
    want, got = out["starling2_ref"], out["starling2_" + variant]
    if not os.environ.get("SK_FUZZ_KEEP"):
        shutil.rmtree(d, ignore_errors=True)
    what = "two-sample seed %d: %d bp at %gx + %gx, snv/%d indel/%d second sample's own/%d at %g, regions %s%s" % (
        seed, length, d1, d2, snv_every, indel_every, som_every, clone, ",".join(regions), " EVS" if extra else "")
    for f in want:
        if want[f] != got[f]:
            k = next((i for i, (x, y) in enumerate(zip(want[f], got[f])) if x != y), min(len(want[f]), len(got[f])))
            return False, "%s: %s differs at line %d\n  reference: %s\n  drop-in:   %s" % (
                what, f, k + 1, want[f][k] if k < len(want[f]) else "<end>", got[f][k] if k < len(got[f]) else "<end>")
    return True, "%s: identical (%d variant records, %d + %d gVCF lines)" % (
        what, sum(1 for l in want["variants.vcf"] if l[0] != "#"), len(want["genome.S1.vcf"]), len(want["genome.S2.vcf"]))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    variant = sys.argv[3] if len(sys.argv) > 3 else "dbl"
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    md = tempfile.mkdtemp(prefix="sk_models_")
    subprocess.run([sys.executable, "tools/make_dummy_germline_models.py", md], check=True)
    models = ("--snv-scoring-model-file", md + "/germlineSNVScoringModels.json", "--indel-scoring-model-file", md + "/germlineIndelScoringModels.json")
    bad = 0
    mode = sys.argv[5] if len(sys.argv) > 5 else "germline"
    fn = {"somatic": lambda s: one_somatic(s, variant), "multi": lambda s: one_multi(s, variant, models),
          "adversarial": lambda s: one_adversarial(s, variant, models)}.get(mode, lambda s: one(s, variant, models))
    with ThreadPoolExecutor(workers) as ex:
        for ok, msg in ex.map(fn, range(first, first + n)):
            print(msg, flush=True)
            bad += 0 if ok else 1
    print("%d of %d seeds identical" % (n - bad, n))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
