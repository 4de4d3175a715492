#!/usr/bin/env python3
"""Synthetic end-to-end inputs for the adapter tests: a random reference contig with planted variation and reads drawn
from the sample's two haplotypes, written as SAM and turned into indexed BAMs with the samtools of the reference's own
redist/ tarball (built by oracle/Makefile).  TEST INFRASTRUCTURE; deterministic from the seed, nothing is committed.

What the data is built to exercise (the demo BAMs of the reference are 5 kb with a handful of variants):
  * dense candidate indels: 1-15 bp insertions/deletions, some 30-49 bp deletions, homopolymer / short-tandem-repeat
    contexts, indels and SNVs a few bases apart (active regions, orthogonal alleles, multi-allelic loci);
  * reads the mapper got "wrong": an indel close to a read end reported as soft clip or as a gapless stretch of
    mismatches, indels not left-shifted -- the reads realignment exists for;
  * MAPQ tiers: most reads MAPQ 60, some below the tier1 threshold (tier2 evidence in somatic mode), a few MAPQ 0;
  * somatic pairs: tumour-only SNVs / indels at 10-40 % allele fraction on top of shared germline variation;
  * depth spikes and zero-coverage gaps.

usage: make_synth_bam.py <out dir> <samtools> [--seed N] [--length BP]
"""
import argparse
import os
import subprocess

import numpy as np

BASES = "ACGT"


def random_reference(length, rng):
    seq = rng.integers(0, 4, length)
    # homopolymers and short tandem repeats every ~700 bp
    p = 300
    while p < length - 200:
        kind = rng.integers(0, 3)
        if kind == 0:
            n = int(rng.integers(6, 16))
            seq[p:p + n] = rng.integers(0, 4)
        elif kind == 1:
            unit = rng.integers(0, 4, int(rng.integers(2, 5)))
            n = int(rng.integers(4, 12))
            rep = np.tile(unit, n)
            seq[p:p + len(rep)] = rep
        p += int(rng.integers(400, 1000))
    return "".join(BASES[b] for b in seq)


def plant_variants(ref, rng, spacing, somatic=False):
    """-> sorted list of (pos, ref_len, alt_seq): pos 0-based, replaces ref[pos:pos+ref_len] by alt_seq"""
    out = []
    p = 150
    L = len(ref)
    while p < L - 300:
        r = rng.random()
        if r < 0.45:      # SNV
            alt = BASES[(BASES.index(ref[p]) + int(rng.integers(1, 4))) % 4]
            out.append((p, 1, alt))
        elif r < 0.70:    # deletion
            n = int(rng.integers(1, 16)) if rng.random() < 0.85 else int(rng.integers(30, 50))
            out.append((p, n, ""))
        elif r < 0.92:    # insertion
            n = int(rng.integers(1, 16))
            ins = "".join(BASES[b] for b in rng.integers(0, 4, n)) if rng.random() < 0.6 else ref[p:p + n] or "A"
            out.append((p, 0, ins))
        else:             # a cluster: SNV + indel + SNV within a dozen bases
            alt = BASES[(BASES.index(ref[p]) + 1) % 4]
            out.append((p, 1, alt))
            q = p + int(rng.integers(3, 9))
            if rng.random() < 0.5:
                out.append((q, int(rng.integers(1, 6)), ""))
                q += 7
            else:
                out.append((q, 0, "".join(BASES[b] for b in rng.integers(0, 4, int(rng.integers(1, 6))))))
                q += 2
            out.append((q + int(rng.integers(2, 6)), 1, BASES[(BASES.index(ref[q + 5]) + 2) % 4] if q + 5 < L else "A"))
        p = out[-1][0] + out[-1][1] + int(rng.integers(spacing // 3, spacing * 2))
    return sorted(out)


def build_haplotype(ref, variants):
    """-> (hap sequence, blocks): blocks = [(op, ref_pos, hap_pos, length)] with op in M, I, D"""
    seq, blocks = [], []
    rp, hp = 0, 0
    for pos, ref_len, alt in variants:
        if pos < rp:
            continue
        if pos > rp:
            seq.append(ref[rp:pos])
            blocks.append(("M", rp, hp, pos - rp))
            hp += pos - rp
            rp = pos
        if ref_len == 1 and len(alt) == 1:
            seq.append(alt)
            blocks.append(("M", rp, hp, 1))
            hp += 1
            rp += 1
        else:
            if ref_len:
                blocks.append(("D", rp, hp, ref_len))
                rp += ref_len
            if alt:
                seq.append(alt)
                blocks.append(("I", rp, hp, len(alt)))
                hp += len(alt)
    seq.append(ref[rp:])
    blocks.append(("M", rp, hp, len(ref) - rp))
    return "".join(seq), blocks


def read_alignment(blocks, start, length, rng, sloppy_rate):
    """CIGAR ops [(op, len)] and reference start of hap[start:start+length]"""
    ops = []
    ref_start = None
    end = start + length
    for op, rp, hp, n in blocks:
        if op == "D":
            if hp > start and hp < end:
                ops.append(["D", n, rp])
            continue
        lo, hi = max(hp, start), min(hp + n, end)
        if lo >= hi:
            continue
        if op == "M":
            if ref_start is None:
                ref_start = rp + (lo - hp)
            ops.append(["M", hi - lo, rp + (lo - hp)])
        else:
            ops.append(["I", hi - lo, rp])
    # an alignment cannot start or end with an insertion / deletion
    while ops and ops[0][0] == "D":
        ops.pop(0)
    while ops and ops[-1][0] == "D":
        ops.pop()
    if ops and ops[0][0] == "I":
        ops[0][0] = "S"
    if ops and ops[-1][0] == "I":
        ops[-1][0] = "S"
    if ref_start is None:
        return None, None
    # the mapper's habit near read ends: an indel within a few bases of the end becomes a soft clip or is run through
    if rng.random() < sloppy_rate and len(ops) >= 3:
        tail = ops[-1]
        if tail[0] == "M" and tail[1] <= 10 and ops[-2][0] in "ID":
            if ops[-2][0] == "I":
                n = ops[-2][1] + tail[1]
                ops = ops[:-2] + [["S", n, 0]]
            else:  # deletion skipped: the tail is laid down without the gap
                ops = ops[:-2]
                ops[-1][1] += tail[1]
        head = ops[0]
        if len(ops) >= 3 and head[0] == "M" and head[1] <= 10 and ops[1][0] == "I":
            n = head[1] + ops[1][1]
            ops = [["S", n, 0]] + ops[2:]
            ref_start = ops[1][2]
    # merge neighbours of the same kind
    merged = []
    for o in ops:
        if merged and merged[-1][0] == o[0]:
            merged[-1][1] += o[1]
        else:
            merged.append([o[0], o[1]])
    return merged, ref_start


QUALS = np.array([12, 20, 25, 30, 33, 37, 40])
QPROB = np.array([0.03, 0.05, 0.07, 0.15, 0.2, 0.3, 0.2])


def sample_reads(name, haps, hap_fracs, depth_fn, ref_len, rng, read_len=150, sloppy_rate=0.6):
    """SAM lines for one sample.  haps: [(seq, blocks)], hap_fracs: probability of each haplotype"""
    lines = []
    n_reads = int(sum(depth_fn(p) for p in range(0, ref_len, read_len)))
    starts = []
    for p in range(0, ref_len, 25):
        lam = depth_fn(p) * 25.0 / read_len
        for _ in range(rng.poisson(lam)):
            starts.append(p + int(rng.integers(0, 25)))
    starts.sort()
    rid = 0
    for s in starts:
        h = int(rng.choice(len(haps), p=hap_fracs))
        seq, blocks = haps[h]
        L = read_len if rng.random() < 0.9 else int(rng.integers(60, read_len + 1))
        # s is a reference coordinate; use it as the haplotype coordinate (offsets are small) clamped to the haplotype
        hs = min(max(0, s), max(0, len(seq) - L - 1))
        ops, ref_start = read_alignment(blocks, hs, L, rng, sloppy_rate)
        if not ops or ref_start is None:
            continue
        bases = list(seq[hs:hs + L])
        q = rng.choice(QUALS, L, p=QPROB)
        if rng.random() < 0.05:
            q[-int(rng.integers(5, 40)):] = 2   # a '#' tail
        err = rng.random(L) < np.power(10.0, -q / 10.0)
        for i in np.flatnonzero(err):
            bases[i] = BASES[(BASES.index(bases[i]) + int(rng.integers(1, 4))) % 4]
        if rng.random() < 0.002:
            bases[int(rng.integers(0, L))] = "N"
        r = rng.random()
        mapq = 60 if r < 0.88 else (int(rng.integers(1, 20)) if r < 0.97 else 0)
        flag = 16 if rng.random() < 0.5 else 0
        cigar = "".join("%d%s" % (n, o) for o, n in ops)
        qual = "".join(chr(33 + int(x)) for x in q)
        lines.append((ref_start, "%s_%06d\t%d\tchrS\t%d\t%d\t%s\t*\t0\t0\t%s\t%s" % (
            name, rid, flag, ref_start + 1, mapq, cigar, "".join(bases), qual)))
        rid += 1
    lines.sort(key=lambda x: x[0])
    return [l for _, l in lines]


def write_bam(path, sample, ref_len, lines, samtools):
    sam = path[:-4] + ".sam"
    with open(sam, "w") as f:
        f.write("@HD\tVN:1.5\tSO:coordinate\n@SQ\tSN:chrS\tLN:%d\n@RG\tID:%s\tSM:%s\n" % (ref_len, sample, sample))
        for l in lines:
            f.write(l + "\tRG:Z:%s\n" % sample)
    subprocess.run([samtools, "view", "-b", "-o", path, sam], check=True)
    subprocess.run([samtools, "index", path], check=True)
    os.remove(sam)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("samtools")
    ap.add_argument("--seed", type=int, default=20250925)
    ap.add_argument("--length", type=int, default=60000)
    ap.add_argument("--read-length", type=int, default=150)
    ap.add_argument("--spacing", type=int, default=350, help="mean distance between planted germline variants")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    os.makedirs(a.out, exist_ok=True)
    ref = random_reference(a.length, rng)
    with open(os.path.join(a.out, "synth.fa"), "w") as f:
        f.write(">chrS\n")
        for i in range(0, len(ref), 60):
            f.write(ref[i:i + 60] + "\n")
    subprocess.run([a.samtools, "faidx", os.path.join(a.out, "synth.fa")], check=True)

    def depth(mean):
        def fn(p):
            if a.length // 3 <= p < a.length // 3 + 400:
                return 0.0            # a coverage gap
            if 2 * a.length // 3 <= p < 2 * a.length // 3 + 600:
                return mean * 6.0     # a pile-up
            return mean
        return fn

    # germline trio-like pair: shared and private variants, het and hom
    germ = plant_variants(ref, rng, a.spacing)
    for name, drop in (("S1", 0.0), ("S2", 0.5)):
        v = [x for x in germ if rng.random() >= drop]
        h1 = [x for x in v if rng.random() < 0.75]
        h2 = [x for x in v if (x in h1 and rng.random() < 0.35) or (x not in h1)]
        haps = [build_haplotype(ref, h1), build_haplotype(ref, h2)]
        write_bam(os.path.join(a.out, "germline_%s.bam" % name), name, len(ref),
                  sample_reads(name, haps, [0.5, 0.5], depth(40.0), len(ref), rng, read_len=a.read_length), a.samtools)

    # tumour / normal pair
    shared = plant_variants(ref, rng, 900)
    h1 = [x for x in shared if rng.random() < 0.7]
    h2 = [x for x in shared if x not in h1 or rng.random() < 0.3]
    normal_haps = [build_haplotype(ref, h1), build_haplotype(ref, h2)]
    som = [x for x in plant_variants(ref, rng, 700) if all(abs(x[0] - y[0]) > 60 for y in shared)]
    tumor_h1 = sorted(h1 + som[::2])
    tumor_h2 = sorted(h2 + som[1::2])
    tumor_haps = normal_haps + [build_haplotype(ref, tumor_h1), build_haplotype(ref, tumor_h2)]
    write_bam(os.path.join(a.out, "somatic_normal.bam"), "NORMAL", len(ref),
              sample_reads("N", normal_haps, [0.5, 0.5], depth(40.0), len(ref), rng, read_len=a.read_length), a.samtools)
    write_bam(os.path.join(a.out, "somatic_tumor.bam"), "TUMOR", len(ref),
              sample_reads("T", tumor_haps, [0.2, 0.2, 0.3, 0.3], depth(80.0), len(ref), rng, read_len=a.read_length), a.samtools)
    print("synthetic inputs in %s: %d germline variants, %d shared + %d somatic variants, %d bp" % (
        a.out, len(germ), len(shared), len(som), len(ref)))


if __name__ == "__main__":
    main()
