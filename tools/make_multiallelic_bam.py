#!/usr/bin/env python3
"""Four germline samples whose indels differ at the same loci: every sample is heterozygous for its OWN pair of overlapping indels at each
planted locus, so the allele group the caller forms there (selectTopOrthogonalAllelesInAllSamples: the union of every sample's top
alleles) holds up to ploidy x samples = 8 alternate alleles -- the case sk_allele_group_genotype_lhoods_wide exists for.  Built from
the pieces of tools/make_synth_bam.py.  TEST INFRASTRUCTURE; deterministic from the seed.

usage: make_multiallelic_bam.py <out dir> <samtools> [--seed N] [--length BP] [--samples K]"""
import argparse
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_synth_bam as M  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("samtools")
    ap.add_argument("--seed", type=int, default=20260927)
    ap.add_argument("--length", type=int, default=24000)
    ap.add_argument("--samples", type=int, default=4)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    os.makedirs(a.out, exist_ok=True)
    ref = "".join(M.BASES[b] for b in rng.integers(0, 4, a.length))
    with open(os.path.join(a.out, "multi.fa"), "w") as f:
        f.write(">chrS\n")
        for i in range(0, len(ref), 60):
            f.write(ref[i:i + 60] + "\n")
    subprocess.run([a.samtools, "faidx", os.path.join(a.out, "multi.fa")], check=True)
    loci = list(range(600, a.length - 600, 800))
    for k in range(a.samples):
        h1, h2 = [], []
        for j, p in enumerate(loci):
            # sample k, locus j: one haplotype deletes 2k+1+(j%2) bases at p, the other inserts a sample-specific sequence there; every
            # third locus two samples share an allele, every fifth one sample is homozygous
            dl = 2 * k + 1 + (j % 2)
            ins = "".join(M.BASES[(k + i + j) % 4] for i in range(2 * k + 2))
            if j % 3 == 0 and k >= 2:
                dl = 1 + (j % 2)
            h1.append((p, dl, ""))
            if not (j % 5 == 0 and k == 1):
                h2.append((p, 0, ins))
            else:
                h2.append((p, dl, ""))
            # a private SNV nearby
            q = p + 40 + 3 * k
            h1.append((q, 1, M.BASES[(M.BASES.index(ref[q]) + 1) % 4]))
        haps = [M.build_haplotype(ref, sorted(h1)), M.build_haplotype(ref, sorted(h2))]
        name = "M%d" % (k + 1)
        M.write_bam(os.path.join(a.out, "multi_%s.bam" % name), name, len(ref),
                    M.sample_reads(name, haps, [0.5, 0.5], lambda p: 45.0, len(ref), rng, sloppy_rate=0.3), a.samtools)
    print("multi-allelic inputs in %s: %d samples, %d loci" % (a.out, a.samples, len(loci)))


if __name__ == "__main__":
    main()
