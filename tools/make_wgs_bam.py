#!/usr/bin/env python3
"""A WGS-like synthetic germline sample for the end-to-end throughput leg (BASELINE.json configs[1]: 40x, 150 bp reads).

TEST / BENCH INFRASTRUCTURE; deterministic from the seed; nothing it writes is committed (oracle/_ref/synth/ is git-ignored,
and the data set is made on the spot where it is missing -- it is too large to travel with a repository snapshot).

Unlike tools/make_synth_bam.py (which packs every awkward case the realigner exists for into 60 kb), the variation here has
the density of a human genome, so that the time of a caller process is spent where it is spent on real data: most reads never
meet a candidate indel, every locus is piled up and genotyped, almost all of the gVCF is non-variant blocks.

  * reference: uniform ACGT with a homopolymer or short tandem repeat every ~1.5 kb (indel error-model contexts);
  * variants: an SNV every ~1 kb, an indel (1-12 bp, 55 % deletions) every ~8 kb, 2/3 het, 1/3 hom; 35 % of the indels sit
    in the repeats (where real indels are);
  * reads: uniform starts at the requested depth, 150 bp, qualities in runs over {12, 23, 37} plus a '#' tail on 3 % of the
    reads, substitution errors at the rate the quality states, 0.5 per mille of the reads carry a sequencing-error indel,
    MAPQ 60 for 93 %, 1-19 for 5 %, 0 for 2 %;
  * the mapper's habits: an indel within 8 bases of a read end comes out as a soft clip (insertions) or is run through as
    mismatches (deletions) for 70 % of such reads.

usage: make_wgs_bam.py <out dir> <samtools> [--length BP] [--depth X] [--seed N] [--procs P] [--contig NAME] [--sample NAME]
writes <out>/wgs.fa (+ .fai), <out>/wgs.bam (+ .bai), <out>/truth.tsv
"""
import argparse
import os
import subprocess
import sys

import numpy as np

BASES = np.frombuffer(b"ACGT", np.uint8)


def random_reference(length, rng):
    seq = rng.integers(0, 4, length, dtype=np.uint8)
    repeats = []
    p = 500
    while p < length - 300:
        if rng.random() < 0.5:
            n = int(rng.integers(6, 18))
            seq[p:p + n] = rng.integers(0, 4)
        else:
            unit = rng.integers(0, 4, int(rng.integers(2, 5)), dtype=np.uint8)
            n = int(rng.integers(4, 12)) * len(unit)
            seq[p:p + n] = np.tile(unit, n // len(unit))
        repeats.append((p, n))
        p += int(rng.integers(800, 2200))
    return seq, repeats


def plant(ref, repeats, rng, snv_every, indel_every):
    """-> sorted, non-overlapping [(pos, ref_len, alt bytes, zygosity)]; zygosity 0/1 = het on that haplotype, 2 = hom"""
    L = len(ref)
    out = {}
    n_snv = int(L / snv_every)
    for p in rng.integers(200, L - 200, n_snv):
        p = int(p)
        alt = BASES[(int(ref[p]) + int(rng.integers(1, 4))) % 4]
        out[p] = (p, 1, bytes([alt]), int(rng.choice(3, p=[1 / 3, 1 / 3, 1 / 3])))
    n_indel = int(L / indel_every)
    for _ in range(n_indel):
        if repeats and rng.random() < 0.35:
            rp, rn = repeats[int(rng.integers(0, len(repeats)))]
            p = rp + int(rng.integers(0, max(1, rn - 2)))
        else:
            p = int(rng.integers(200, L - 200))
        n = int(rng.integers(1, 13)) if rng.random() < 0.9 else int(rng.integers(13, 40))
        if rng.random() < 0.55:
            out[p] = (p, n, b"", int(rng.choice(3, p=[1 / 3, 1 / 3, 1 / 3])))
        else:
            if rng.random() < 0.5:
                ins = bytes(BASES[rng.integers(0, 4, n)])
            else:  # duplication of the sequence that follows
                ins = bytes(BASES[ref[p:p + n]])
            out[p] = (p, 0, ins, int(rng.choice(3, p=[1 / 3, 1 / 3, 1 / 3])))
    v = sorted(out.values())
    keep, end = [], 0
    for x in v:
        if x[0] < end + 12:
            continue
        keep.append(x)
        end = x[0] + x[1]
    return keep


def haplotype(ref, variants, which):
    """-> (hap codes u8, hap->ref map arrays): seg_hap_start[], seg_ref_start[], seg_kind[] (0 M, 1 I, 2 D-after marker)"""
    pieces, blocks = [], []
    rp = hp = 0
    for pos, ref_len, alt, zyg in variants:
        if not (zyg == 2 or zyg == which):
            continue
        if pos > rp:
            pieces.append(ref[rp:pos])
            blocks.append((0, rp, hp, pos - rp))
            hp += pos - rp
            rp = pos
        if ref_len == 1 and len(alt) == 1:
            pieces.append(np.array([b"ACGT".index(alt)], np.uint8))
            blocks.append((0, rp, hp, 1))
            hp += 1
            rp += 1
        else:
            if ref_len:
                blocks.append((2, rp, hp, ref_len))
                rp += ref_len
            if alt:
                pieces.append(np.array([b"ACGT".index(bytes([c])) for c in alt], np.uint8))
                blocks.append((1, rp, hp, len(alt)))
                hp += len(alt)
    pieces.append(ref[rp:])
    blocks.append((0, rp, hp, len(ref) - rp))
    return np.concatenate(pieces), blocks


def merge_m(ops):
    out = []
    for o, n in ops:
        if n <= 0:
            continue
        if out and out[-1][0] == o:
            out[-1][1] += n
        else:
            out.append([o, n])
    return out


def cigar_for(blocks, bstart, hs, L, rng):
    """ops [(op, len)] + reference start of hap[hs:hs+L]; bstart[i] = hap start of block i (sorted)"""
    end = hs + L
    i = max(0, int(np.searchsorted(bstart, hs, side="right")) - 1)
    ops, ref_start = [], None
    while i < len(blocks):
        kind, rp, hp, n = blocks[i]
        if hp >= end:
            break
        if kind == 2:
            if hs < hp < end and ops:
                ops.append(["D", n])
        else:
            lo, hi = max(hp, hs), min(hp + n, end)
            if lo < hi:
                if kind == 0:
                    if ref_start is None:
                        ref_start = rp + (lo - hp)
                    ops.append(["M", hi - lo])
                else:
                    ops.append(["I", hi - lo])
        i += 1
    while ops and ops[-1][0] == "D":
        ops.pop()
    if ops and ops[0][0] == "I":
        ops[0][0] = "S"
    if ops and ops[-1][0] == "I":
        ops[-1][0] = "S"
    if ref_start is None:
        return None, None
    # the mapper near read ends
    if len(ops) >= 3 and rng.random() < 0.7:
        if ops[-1][0] == "M" and ops[-1][1] <= 8 and ops[-2][0] in "ID":
            if ops[-2][0] == "I":
                ops = ops[:-2] + [["S", ops[-2][1] + ops[-1][1]]]
            else:
                tail = ops[-1][1]
                ops = ops[:-2]
                ops[-1][1] += tail
        if len(ops) >= 3 and ops[0][0] == "M" and ops[0][1] <= 8 and ops[1][0] == "I":
            n0 = ops[0][1]
            ops = [["S", n0 + ops[1][1]]] + ops[2:]
            ref_start += n0
    return merge_m(ops), ref_start


def qualities(n, L, rng):
    """(n, L) u8: runs over three levels (as binned instrument qualities), a few '#' tails"""
    levels = np.array([37, 37, 37, 37, 37, 23, 23, 12], np.uint8)
    run = 10
    k = (L + run - 1) // run
    q = np.repeat(levels[rng.integers(0, len(levels), (n, k))], run, axis=1)[:, :L].copy()
    tails = np.flatnonzero(rng.random(n) < 0.03)
    for i in tails:
        q[i, L - int(rng.integers(5, 40)):] = 2
    return q


_WORKER = None


def _run_worker(k):
    return _WORKER(k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("samtools")
    ap.add_argument("--length", type=int, default=4000000)
    ap.add_argument("--depth", type=float, default=40.0)
    ap.add_argument("--seed", type=int, default=20260926)
    ap.add_argument("--read-length", type=int, default=150)
    ap.add_argument("--contig", default="chrW")
    ap.add_argument("--sample", default="NA_SYNTH")
    ap.add_argument("--name", default="wgs")
    ap.add_argument("--snv-every", type=float, default=1000.0)
    ap.add_argument("--indel-every", type=float, default=8000.0)
    ap.add_argument("--role", choices=["germline", "normal", "tumor"], default="germline",
                    help="tumor: reads also come from two clone haplotypes carrying somatic variants (--clone-fraction of the reads); normal / "
                         "tumor of one seed share the reference and the germline variants")
    ap.add_argument("--somatic-every", type=float, default=40000.0, help="mean distance between somatic variants (tumor)")
    ap.add_argument("--clone-fraction", type=float, default=0.6)
    ap.add_argument("--procs", type=int, default=max(1, min(16, len(os.sched_getaffinity(0)))),
                    help="worker processes (the data set depends on the seed AND this number)")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    os.makedirs(a.out, exist_ok=True)
    L, RL = a.length, a.read_length
    ref, repeats = random_reference(L, rng)
    fa = os.path.join(a.out, a.name + ".fa")
    with open(fa, "wb") as f:
        f.write(b">" + a.contig.encode() + b"\n")
        text = BASES[ref].tobytes()
        for i in range(0, L, 60):
            f.write(text[i:i + 60] + b"\n")
    subprocess.run([a.samtools, "faidx", fa], check=True)
    variants = plant(ref, repeats, rng, a.snv_every, a.indel_every)
    with open(os.path.join(a.out, a.name + ".truth.tsv"), "w") as f:
        for pos, ref_len, alt, zyg in variants:
            f.write("%d\t%d\t%s\t%d\n" % (pos + 1, ref_len, alt.decode(), zyg))
    haps = [haplotype(ref, variants, w) for w in (0, 1)]
    hap_probs = [0.5, 0.5]
    if a.role == "tumor":
        # somatic variants: their own generator stream (the germline draw above is the pair's common part), kept clear of the germline ones
        srng = np.random.default_rng(a.seed + 7)
        taken = np.array([v[0] for v in variants], np.int64)
        som = []
        for v in plant(ref, repeats, srng, a.somatic_every * 1.25, a.somatic_every * 5.0):
            j = int(np.searchsorted(taken, v[0]))
            near = min(abs(int(taken[k]) - v[0]) for k in (j - 1, j) if 0 <= k < len(taken)) if len(taken) else 1 << 30
            if near > 60:
                som.append((v[0], v[1], v[2], int(srng.integers(0, 2))))  # on one clone haplotype
        with open(os.path.join(a.out, a.name + ".somatic_truth.tsv"), "w") as f:
            for pos, ref_len, alt, zyg in som:
                f.write("%d\t%d\t%s\t%d\n" % (pos + 1, ref_len, alt.decode(), zyg))
        merged = sorted(variants + som)
        haps += [haplotype(ref, [v for v in merged], w) for w in (0, 1)]
        cf = a.clone_fraction
        hap_probs = [(1 - cf) / 2, (1 - cf) / 2, cf / 2, cf / 2]
    hap_cum = np.cumsum(hap_probs)
    hap_text = [BASES[h[0]] for h in haps]
    bstarts = [np.array([b[2] for b in h[1]], np.int64) for h in haps]
    # hap position of every "structural" block (I or D) for the fast path test
    indel_hp = [np.array(sorted(b[2] for b in h[1] if b[0] != 0), np.int64) for h in haps]
    # a map hap position -> reference position for reads that touch no indel block: piecewise offsets
    off_pos = []
    ref_to_hap = []   # read starts are drawn in reference coordinates (the BAM is sorted by them) and mapped to the haplotype
    for h in haps:
        hp_list, off_list, rp_list, roff_list = [0], [0], [0], [0]
        for kind, rp, hp, n in h[1]:
            if kind == 0:
                hp_list.append(hp)
                off_list.append(rp - hp)
                rp_list.append(rp)
                roff_list.append(hp - rp)
        off_pos.append((np.array(hp_list, np.int64), np.array(off_list, np.int64)))
        ref_to_hap.append((np.array(rp_list, np.int64), np.array(roff_list, np.int64)))

    n_reads = int(L * a.depth / RL)
    starts = np.sort((rng if a.role != "tumor" else np.random.default_rng(a.seed + 11)).integers(0, L - RL - 50, n_reads))
    bam = os.path.join(a.out, a.name + ".bam")
    header = ("@HD\tVN:1.5\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n@RG\tID:%s\tSM:%s\n" % (a.contig, L, a.sample, a.sample)).encode()
    tail = ("\tRG:Z:%s\n" % a.sample).encode()
    contig = a.contig.encode()
    CH = 1 << 16

    # The reads are made in `procs` stretches of the (sorted) start list, one worker process each writing its own BAM; a worker
    # keeps the reads whose reference start lies in its stretch [lo, hi) (a soft clip can move a start by a few bases), so the
    # pieces joined with `samtools cat` are coordinate-sorted.  Worker k draws from its own generator (seed + 1000 + k): the
    # data set depends on the seed AND the number of stretches.
    procs = max(1, min(a.procs, n_reads // 1000 or 1))
    cuts = [int(round(k * n_reads / procs)) for k in range(procs + 1)]

    def worker(k):
        wrng = np.random.default_rng(a.seed + 1000 + k + {"germline": 0, "normal": 0, "tumor": 500000}[a.role])
        lo = int(starts[cuts[k]]) if k > 0 else -(1 << 62)
        hi = int(starts[cuts[k + 1]]) if k + 1 < procs else 1 << 62
        part = os.path.join(a.out, "%s.part%03d.bam" % (a.name, k))
        proc = subprocess.Popen([a.samtools, "view", "-b", "-o", part, "-"], stdin=subprocess.PIPE)
        w = proc.stdin
        w.write(header)
        rid = 0
        carry = []  # (ref_start, line): flushed in order once no later read can start before them
        for c0 in range(cuts[k], cuts[k + 1], CH):
            st = starts[c0:min(c0 + CH, cuts[k + 1])]
            n = len(st)
            which = np.minimum(np.searchsorted(hap_cum, wrng.random(n), side="right"), len(haps) - 1)
            lens = np.where(wrng.random(n) < 0.93, RL, wrng.integers(70, RL + 1, n))
            q = qualities(n, RL, wrng)
            err = wrng.random((n, RL)) < np.power(10.0, -q.astype(np.float64) / 10.0)
            shift = wrng.integers(1, 4, (n, RL), dtype=np.uint8)
            r = wrng.random(n)
            mapq = np.where(r < 0.93, 60, np.where(r < 0.98, wrng.integers(1, 20, n), 0))
            flag = np.where(wrng.random(n) < 0.5, 16, 0)
            seq_err_indel = wrng.random(n) < 0.0005
            lines = []
            for i in range(n):
                h = int(which[i])
                kk = int(np.searchsorted(ref_to_hap[h][0], st[i], side="right")) - 1
                hs = int(min(max(0, st[i] + ref_to_hap[h][1][kk]), len(haps[h][0]) - RL - 1))
                Lr = int(lens[i])
                codes = haps[h][0][hs:hs + Lr]
                e = err[i, :Lr]
                if e.any():
                    codes = codes.copy()
                    codes[e] = (codes[e] + shift[i, :Lr][e]) & 3
                j = int(np.searchsorted(indel_hp[h], hs, side="right"))
                if j < len(indel_hp[h]) and indel_hp[h][j] < hs + Lr or (j > 0 and indel_hp[h][j - 1] >= hs - 64):
                    ops, ref_start = cigar_for(haps[h][1], bstarts[h], hs, Lr, wrng)
                    if ops is None:
                        continue
                else:
                    kq = int(np.searchsorted(off_pos[h][0], hs, side="right")) - 1
                    ref_start = hs + int(off_pos[h][1][kq])
                    ops = [["M", Lr]]
                if not (lo <= ref_start < hi):
                    continue
                seq = BASES[codes].tobytes()
                if seq_err_indel[i] and len(ops) == 1 and Lr > 60:
                    p = int(wrng.integers(20, Lr - 20))
                    if wrng.random() < 0.5:      # a base dropped by the instrument
                        seq = seq[:p] + seq[p + 1:]
                        ops = [["M", p], ["D", 1], ["M", Lr - p - 1]]
                        Lr -= 1
                    else:
                        seq = seq[:p] + b"ACGT"[int(wrng.integers(0, 4)):][:1] + seq[p:]
                        ops = [["M", p], ["I", 1], ["M", Lr - p]]
                        Lr += 1
                qs = (q[i, :Lr] + 33).tobytes() if Lr <= RL else (q[i, :RL] + 33).tobytes() + b"F" * (Lr - RL)
                cigar = "".join("%d%s" % (m, o) for o, m in ops).encode()
                lines.append((ref_start, b"r%02d_%07d\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s%s" % (
                    k, rid, int(flag[i]), contig, ref_start + 1, int(mapq[i]), cigar, seq, qs, tail)))
                rid += 1
            carry.extend(lines)
            carry.sort(key=lambda x: x[0])
            limit = int(st[-1]) - 2000 if c0 + CH < cuts[k + 1] else 1 << 62
            kc = 0
            while kc < len(carry) and carry[kc][0] <= limit:
                kc += 1
            w.write(b"".join(l for _, l in carry[:kc]))
            carry = carry[kc:]
        w.close()
        if proc.wait() != 0:
            sys.exit("samtools view failed")
        return part, rid

    if procs == 1:
        results = [worker(0)]
    else:
        import multiprocessing as mp
        global _WORKER
        _WORKER = worker
        with mp.get_context("fork").Pool(procs) as pool:   # (fork: the workers read the haplotypes where they are)
            results = pool.map(_run_worker, range(procs))
    parts = [r[0] for r in results]
    if len(parts) == 1:
        os.replace(parts[0], bam)
    else:
        subprocess.run([a.samtools, "cat", "-o", bam] + parts, check=True)
        for pth in parts:
            os.remove(pth)
    subprocess.run([a.samtools, "index", bam], check=True)
    print("%s: %d bp, %d reads, %d variants (%d indels)" % (bam, L, sum(r[1] for r in results), len(variants),
                                                           sum(1 for v in variants if not (v[1] == 1 and len(v[2]) == 1))))


if __name__ == "__main__":
    main()
