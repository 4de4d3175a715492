"""tests/golden/gvcf_site_reference.npz: what the REFERENCE itself says of every position of three small WGS-like samples -- the
unmodified starling2 (oracle/_ref/bin/starling2_ref, built from the reference's own translation units) run with a no-compress BED
over the whole sample, so that its gVCF has one record per position instead of blocks.  Per position: is it a homozygous-reference
site without an alternate allele (ALT '.', GT 0/0), GQX, DP (the used basecalls: the cleaned pileup's size), DPF (the unused ones),
the FILTER / FT keys.  That is what sk_gvcf_site_summary (kernel V2, strelka_amd/csrc/gvcf_block.hip, over gvcf_site_core.h) claims
per position: "plain site?", GQX, and -- with the window's counts -- DP and DPF.

The samples are regenerated from their seeds by the test (tools/make_wgs_bam.py is deterministic); only the reference's answers are
stored.   usage: python tests/golden/make_gvcf_site_golden.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import e2e_util as E  # noqa: E402

SAMPLES = [dict(length=40000, depth=30.0, seed=7101, snv_every=300, indel_every=2500),
           dict(length=30000, depth=8.0, seed=7102, snv_every=1500, indel_every=6000),
           dict(length=30000, depth=70.0, seed=7103, snv_every=120, indel_every=900)]
FILTERS = ("LowGQX", "LowDepth", "HighDepth", "HighBaseFilt")  # (the keys a hom-ref site can carry; bit i of `ft`)


def make_sample(d, s):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_wgs_bam.py"), d, os.path.join(E.BIN_DIR, "samtools"), "--length", str(s["length"]),
                    "--depth", str(s["depth"]), "--seed", str(s["seed"]), "--snv-every", str(s["snv_every"]), "--indel-every", str(s["indel_every"]),
                    "--procs", "1"], check=True, stdout=subprocess.DEVNULL)
    with open(os.path.join(d, "chrom_depth.txt"), "w") as f:
        f.write("chrW\t%.3f\n" % s["depth"])


def reference_sites(d, s, out_dir):
    bed = os.path.join(out_dir, "all.bed")
    with open(bed, "w") as f:
        f.write("chrW\t0\t%d\n" % s["length"])
    subprocess.run([os.path.join(E.BIN_DIR, "bgzip"), "-f", bed], check=True)
    subprocess.run([os.path.join(E.BIN_DIR, "tabix"), "-f", "-p", "bed", bed + ".gz"], check=True)
    E.run(E.germline_wgs_argv("starling2_ref", out_dir + "/", [os.path.join(d, "wgs.bam")], ["chrW:1-%d" % s["length"]], os.path.join(d, "wgs.fa"),
                              os.path.join(d, "chrom_depth.txt"), nocompress_bed=bed + ".gz"))
    rows = []
    for line in open(os.path.join(out_dir, "genome.S1.vcf")):
        if line.startswith("#"):
            continue
        c = line.rstrip("\n").split("\t")
        fmt = dict(zip(c[8].split(":"), c[9].split(":")))
        if "DP" not in fmt:  # (an indel record: DPI)
            continue
        keys = set(c[6].split(";")) | set(fmt.get("FT", "PASS").split(";"))
        ft = sum(1 << i for i, k in enumerate(FILTERS) if k in keys)
        # (a position under a called deletion is genotyped with a lowered ploidy -- GT "0", or "." under a homozygous one: the window's
        # summary is for ploidy 2, the adapter checks spanningIndelPloidyModification before it uses one; `diploid` marks the others)
        diploid = int("/" in fmt["GT"] or "|" in fmt["GT"])
        rows.append((int(c[1]) - 1, int(c[4] == "." and fmt["GT"] == "0/0"), int(fmt["GQX"]) if fmt.get("GQX", ".") != "." else -1, int(fmt["DP"]), int(fmt["DPF"]), ft,
                     diploid, int("END=" in c[7])))
    a = np.array(rows, dtype=np.int64)
    assert a[:, 7].sum() == 0, "the no-compress region left a block record"
    # (a position can have two records when a site record follows an overlapping indel's: the site records only, one per position)
    _, first = np.unique(a[:, 0], return_index=True)
    return a[first, :7]


def main():
    out = {}
    for i, s in enumerate(SAMPLES):
        with tempfile.TemporaryDirectory() as t:
            d = os.path.join(t, "sample")
            make_sample(d, s)
            a = reference_sites(d, s, t)
            out["sites_%d" % i] = a
            print("sample %d: %d positions with a site record, %d of them hom-ref without an alternate allele" % (i, len(a), int(a[:, 1].sum())))
    np.savez_compressed(os.path.join(HERE, "gvcf_site_reference.npz"), columns=np.array(["pos0", "is_homref_no_alt", "gqx", "dp", "dpf", "filters", "diploid"]), **out)


if __name__ == "__main__":
    main()
