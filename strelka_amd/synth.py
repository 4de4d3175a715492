"""Seeded synthetic workloads of SURVEY.md section 8d (inputs A, B, C) for tests and bench.py.

Two flavours of input A:
  * `align_cases(...)`      reference-shaped (CIGAR paths + indel keys): small, for parity tests through the host adapter
  * `align_batch_flat(...)` already flattened sk_align_batch arrays, vectorised: bench-scale

No model of the caller lives here -- only input generation.
"""
import numpy as np

from . import capi

BAM_CODE = np.array([1, 2, 4, 8], np.uint8)  # A C G T
BASES = "ACGT"
QUAL_VALUES = np.array([2, 11, 25, 32, 37, 40], np.uint8)
QUAL_PROBS = np.array([0.01, 0.04, 0.10, 0.20, 0.35, 0.30])

SEG = dict(NONE=0, MATCH=1, INSERT=2, DELETE=3, SKIP=4, SOFT_CLIP=5, HARD_CLIP=6, PAD=7, SEQ_MATCH=8, SEQ_MISMATCH=9)
INDEL = dict(NONE=0, INDEL=1, MISMATCH=2, BP_LEFT=3, BP_RIGHT=4)


# ----------------------------------------------------------------------------------------------------------------------
# input B / C: pileups

def pileups(n_loci, rng, depth_mean=40.0, het_rate=0.00067, hom_rate=0.00033, nmm_rate=0.02, filter_rate=0.0,
            alt_frac=None, with_n_ref=False, noise=0.0):
    """Germline-style pileup batch: depth ~ Poisson(depth_mean), hom-ref except het/hom-alt loci, strand Bernoulli(.5),
    quals from QUAL_VALUES, a 0.1 %-per-base sequencing error.  Returns capi.HostPileupBatch (de=None)."""
    depth = rng.poisson(depth_mean, n_loci).astype(np.int64)
    off = np.zeros(n_loci + 1, np.int64)
    np.cumsum(depth, out=off[1:])
    total = int(off[-1])
    locus = np.repeat(np.arange(n_loci), depth)
    ref = rng.integers(0, 4, n_loci).astype(np.uint8)
    alt = ((ref + rng.integers(1, 4, n_loci)) % 4).astype(np.uint8)
    u = rng.random(n_loci)
    if alt_frac is None:
        frac = np.where(u < het_rate, 0.5, np.where(u < het_rate + hom_rate, 1.0, 0.0))
    else:
        frac = np.broadcast_to(np.asarray(alt_frac, float), (n_loci,))
    base = np.where(rng.random(total) < frac[locus], alt[locus], ref[locus]).astype(np.uint8)
    q = rng.choice(QUAL_VALUES, total, p=QUAL_PROBS)
    err = rng.random(total) < np.power(10.0, -q.astype(np.float64) / 10.0)
    base = np.where(err, (base + rng.integers(1, 4, total)) % 4, base).astype(np.uint8)
    if noise > 0:  # stress case for tests: many bases present at one locus (all eight strand/base groups populated)
        base = np.where(rng.random(total) < noise, rng.integers(0, 4, total), base).astype(np.uint8)
    fwd = rng.integers(0, 2, total)
    nmm = rng.random(total) < nmm_rate
    filt = rng.random(total) < filter_rate
    calls = capi.make_call(q, base, fwd, nmm, filt, 0)
    if with_n_ref:
        ref = np.where(rng.random(n_loci) < 0.01, 4, ref).astype(np.uint8)
    return capi.HostPileupBatch(off, calls, ref)


def somatic_pileups(n_loci, rng, normal_depth=40.0, tumor_depth=110.0, somatic_rate=1e-4, het_rate=1e-3,
                    somatic_frac=0.2):
    """Normal + tumor pileups sharing ref bases (input C)."""
    ref = rng.integers(0, 4, n_loci).astype(np.uint8)
    alt = ((ref + rng.integers(1, 4, n_loci)) % 4).astype(np.uint8)
    u = rng.random(n_loci)
    nfrac = np.where(u < het_rate, 0.5, 0.0)
    tfrac = np.where(u < het_rate, 0.5, np.where(u < het_rate + somatic_rate, somatic_frac, 0.0))

    def one(depth_mean, frac):
        depth = rng.poisson(depth_mean, n_loci).astype(np.int64)
        off = np.zeros(n_loci + 1, np.int64)
        np.cumsum(depth, out=off[1:])
        total = int(off[-1])
        locus = np.repeat(np.arange(n_loci), depth)
        base = np.where(rng.random(total) < frac[locus], alt[locus], ref[locus]).astype(np.uint8)
        q = rng.choice(QUAL_VALUES, total, p=QUAL_PROBS)
        err = rng.random(total) < np.power(10.0, -q.astype(np.float64) / 10.0)
        base = np.where(err, (base + rng.integers(1, 4, total)) % 4, base).astype(np.uint8)
        calls = capi.make_call(q, base, rng.integers(0, 2, total), 0, 0, 0)
        return capi.HostPileupBatch(off, calls, ref)

    return one(normal_depth, nfrac), one(tumor_depth, tfrac)


def somatic_tier_pileups(n_loci, rng, **kw):
    """Input C with tier2 evidence: (normal_t1, tumor_t1, normal_t2, tumor_t2).  The tier2 columns are what
    CleanPileupFilter(pi, is_include_tier2=true) gives: the tier1 calls, a few calls that only the tier-specific filter
    had removed, then the tier2 reads' calls (noisier: lower qualities, more non-reference bases)."""
    n1, t1 = somatic_pileups(n_loci, rng, **kw)

    def widen(b, extra_mean):
        ref = b.ref_base
        depth = np.diff(b.call_off)
        extra = rng.poisson(extra_mean, n_loci).astype(np.int64)
        extra[rng.random(n_loci) < 0.5] = 0
        extra = np.where(rng.random(n_loci) < 0.25, extra + 12, extra)
        off = np.zeros(n_loci + 1, np.int64)
        np.cumsum(depth + extra, out=off[1:])
        calls = np.zeros(int(off[-1]), np.uint16)
        tot = int(extra.sum())
        locus = np.repeat(np.arange(n_loci), extra)
        q = rng.integers(2, 35, tot)
        base = np.where(rng.random(tot) < 0.85, ref[locus], rng.integers(0, 4, tot)).astype(np.uint8)
        # at a few loci the tier2 reads carry one alternate allele at ~50 %: the two tiers then disagree on the normal
        # genotype (NTYPE conflict, position_somatic_snv_strand_grid.cpp:341-346)
        alt_locus = rng.random(n_loci) < 0.25
        alt_base = ((ref.astype(np.int64) + 1 + (np.arange(n_loci) % 3)) % 4).astype(np.uint8)
        flip = alt_locus[locus] & (rng.random(tot) < 0.5)
        base = np.where(flip, alt_base[locus], base).astype(np.uint8)
        q = np.where(alt_locus[locus], 30, q)
        ecalls = capi.make_call(q, base, rng.integers(0, 2, tot), 0, 0, 0)
        epos = 0
        for l in range(n_loci):
            s, e = int(b.call_off[l]), int(b.call_off[l + 1])
            o = int(off[l])
            calls[o:o + (e - s)] = b.calls[s:e]
            k = int(extra[l])
            calls[o + (e - s):o + (e - s) + k] = ecalls[epos:epos + k]
            epos += k
        return capi.HostPileupBatch(off, calls, ref)

    return n1, t1, widen(n1, 4.0), widen(t1, 8.0)


def somatic_indel_cases(n_cases, rng, normal_depth=40.0, tumor_depth=110.0):
    """Input D for the whole of get_somatic_indel: per candidate indel the two samples' ReadPathScores rows (ref / indel
    scores, up to two alternate-indel scores keyed into a small per-indel table of overlapping alternate alleles, tier flag)
    for germline-absent / somatic / germline-het / noisy-multi-allele loci."""
    cases = []
    for _ in range(n_cases):
        kind = rng.choice(["somatic", "absent", "het", "multi", "compound"], p=[0.35, 0.15, 0.2, 0.15, 0.15])
        is_del = rng.random() < 0.5
        length = int(rng.integers(1, 12))
        del_len, ins_len = (length, 0) if is_del else (0, length)
        n_keys = int(rng.integers(0, 4))
        alt_keys = []
        for _k in range(n_keys):
            b = 1000 + int(rng.integers(-6, 8))
            is_mm = int(rng.random() < 0.2)
            e = b + (1 if is_mm else int(rng.integers(0, 6)))
            if (b, e, is_mm) not in alt_keys:  # distinct alleles only: the table is keyed by IndelKey
                alt_keys.append((b, e, is_mm))
        n_keys = len(alt_keys)

        def sample(depth, frac, alt_rate):
            n = int(rng.poisson(depth))
            has = rng.random(n) < frac
            ref = np.where(has, rng.normal(-22, 6, n), rng.normal(-3, 2, n)).clip(max=0).astype(np.float32)
            ind = np.where(has, rng.normal(-3, 2, n), rng.normal(-22, 6, n)).clip(max=0).astype(np.float32)
            alt_key = np.full((n, 2), -1, np.int32)
            alt_lnp = np.zeros((n, 2), np.float32)
            if n_keys:
                for r in range(n):
                    k = 0
                    if rng.random() < alt_rate:
                        ids = rng.permutation(n_keys)[:int(rng.integers(1, min(2, n_keys) + 1))]
                        for a in ids:
                            alt_key[r, k] = a
                            alt_lnp[r, k] = np.float32(min(0.0, rng.normal(-4, 3) if rng.random() < 0.5 else rng.normal(-25, 5)))
                            k += 1
            if kind == "compound" and n_keys:
                # the reads without the indel carry alternate allele 0 instead of the reference
                for r in range(n):
                    if not has[r] and rng.random() < 0.9:
                        ref[r] = np.float32(min(0.0, rng.normal(-22, 5)))
                        alt_key[r] = (0, -1)
                        alt_lnp[r] = (np.float32(min(0.0, rng.normal(-3, 2))), 0)
            return dict(ref_lnp=ref, indel_lnp=ind, alt_key=alt_key, alt_lnp=alt_lnp,
                        non_ambig=rng.integers(90, 151, n).astype(np.uint16), read_length=np.full(n, 150, np.uint16),
                        is_tier1=(rng.random(n) < 0.85).astype(np.uint8))

        nf, tf = dict(somatic=(0.0, 0.25), absent=(0.0, 0.0), het=(0.5, 0.5), multi=(0.05, 0.3),
                      compound=(0.0 if rng.random() < 0.5 else 0.5, 0.5))[kind]
        alt_rate = 0.7 if kind == "multi" else 0.15
        cases.append(dict(normal=sample(normal_depth, nf, alt_rate), tumor=sample(tumor_depth, tf, alt_rate),
                          alt_keys=alt_keys, del_len=del_len, ins_len=ins_len, forced=int(rng.random() < 0.15),
                          indel_to_ref_error_prob=float(rng.choice([5e-5, 3e-4, 2e-3]))))
    return cases


# ----------------------------------------------------------------------------------------------------------------------
# input A, flattened and vectorised (bench scale)

def align_batch_flat(n_reads, rng, H=64, L=150, K=6, win=400, noncand_rate=0.1):
    """R reads x H=2^K candidate alignments: every subset of K candidate indels (1-10 bp deletions / insertions with
    random insert sequence) toggled on a start-pinned alignment of an L-bp read against a `win`-bp reference window.
    Returns a capi.HostAlignBatch with exactly H candidates per read."""
    assert H == (1 << K)
    ref = BAM_CODE[rng.integers(0, 4, (n_reads, win))]
    start = rng.integers(20, 60, n_reads)
    # K indels, sorted ascending, >= 12 apart, inside the first ~110 bases of the read span
    gaps = rng.integers(12, 18, (n_reads, K))
    ipos = start[:, None] + 8 + np.cumsum(gaps, axis=1)            # reference position (window coords) of indel k
    is_ins = rng.random((n_reads, K)) < 0.5
    ilen = rng.integers(1, 11, (n_reads, K))
    is_cand = rng.random((n_reads, K)) >= noncand_rate
    ins_seq = BAM_CODE[rng.integers(0, 4, (n_reads, K, 10))]        # pool slot k of read r: 10 bytes
    pool_len = win + K * 10
    hap = np.concatenate([ref, ins_seq.reshape(n_reads, K * 10)], axis=1)

    # the read: sampled along the window from `start` with substitution errors at rate 10^(-q/10)
    qual = rng.choice(QUAL_VALUES, (n_reads, L), p=QUAL_PROBS)
    idx = start[:, None] + np.arange(L)[None, :]
    read = np.take_along_axis(ref, idx, axis=1)
    err = rng.random((n_reads, L)) < np.power(10.0, -qual.astype(np.float64) / 10.0)
    sub = BAM_CODE[rng.integers(0, 4, (n_reads, L))]
    read = np.where(err, sub, read).astype(np.uint8)
    read = np.where(rng.random((n_reads, L)) < 0.002, 15, read).astype(np.uint8)  # a few N base calls

    MAXOPS = 3 * K + 1
    ops = np.zeros((n_reads, H, MAXOPS), capi.SCORE_OP_DTYPE)
    nops = np.zeros((n_reads, H), np.int64)
    rows = np.arange(n_reads)

    def push(h, mask, length, kind, flags, src):
        slot = nops[:, h]
        sel = mask & (slot < MAXOPS)
        r = rows[sel]
        s = slot[sel]
        ops["length"][r, h, s] = length[sel] if np.ndim(length) else length
        ops["kind"][r, h, s] = kind
        ops["flags"][r, h, s] = flags[sel] if np.ndim(flags) else flags
        ops["src"][r, h, s] = src[sel] if np.ndim(src) else src
        nops[:, h] += sel

    for h in range(H):
        ref_pos = start.copy()
        read_pos = np.zeros(n_reads, np.int64)
        for k in range(K):
            if not (h >> k) & 1:
                continue
            p = ipos[:, k]
            mlen = p - ref_pos
            # the indel is used when it lies ahead on the reference and strictly inside the read
            use = (mlen >= 0) & (read_pos + mlen < L) & (read_pos + mlen > 0)
            push(h, use & (mlen > 0), mlen, capi.OP_BASES, 0, ref_pos)
            read_pos = np.where(use, read_pos + mlen, read_pos)
            ref_pos = np.where(use, p, ref_pos)
            pen = np.where(is_cand[:, k], 0, capi.OPFLAG_PENALTY).astype(np.uint8)
            ins = use & is_ins[:, k]
            il = np.minimum(ilen[:, k], L - read_pos)
            push(h, ins, il, capi.OP_BASES, pen, np.full(n_reads, win + 10 * k))
            read_pos = np.where(ins, read_pos + il, read_pos)
            dele = use & ~is_ins[:, k]
            push(h, dele & (pen > 0), np.zeros(n_reads, np.int64), capi.OP_NOBASE, pen, np.zeros(n_reads, np.int64))
            ref_pos = np.where(dele, ref_pos + ilen[:, k], ref_pos)
        rest = L - read_pos
        push(h, rest > 0, rest, capi.OP_BASES, 0, ref_pos)

    keep = np.arange(MAXOPS)[None, None, :] < nops[:, :, None]
    flat_ops = ops[keep]
    op_off = np.zeros(n_reads * H + 1, np.int64)
    np.cumsum(nops.reshape(-1), out=op_off[1:])
    read_off = np.arange(n_reads + 1, dtype=np.int64) * L
    hap_off = np.arange(n_reads + 1, dtype=np.int64) * pool_len
    cal_off = (np.arange(n_reads + 1, dtype=np.int64) * H).astype(np.int32)
    return capi.HostAlignBatch(read_off, read.reshape(-1), qual.reshape(-1).astype(np.uint8), hap_off, hap.reshape(-1),
                               cal_off, op_off, flat_ops, L, pool_len)


# ----------------------------------------------------------------------------------------------------------------------
# input A, reference-shaped (parity tests through the host adapter)

def _path_from_indels(start_pos, read_len, indels, lead_clip=0, trail_clip=0):
    """Start-pinned CIGAR of a read against the reference with `indels` (sorted dicts pos,del_len,ins_seq) applied.
    Returns (path, used_indels)."""
    path = []
    used = []
    if lead_clip:
        path.append((SEG["SOFT_CLIP"], lead_clip))
    ref_pos = start_pos
    remaining = read_len - lead_clip - trail_clip
    for ind in indels:
        if remaining <= 0:
            break
        mlen = ind["pos"] - ref_pos
        if mlen <= 0 or mlen >= remaining:
            continue
        path.append((SEG["MATCH"], mlen))
        remaining -= mlen
        ref_pos = ind["pos"]
        ins = len(ind.get("ins_seq", ""))
        dele = ind.get("del_len", 0)
        if ins >= remaining:
            # would run off the read: stop before it
            path.pop()
            remaining += mlen
            ref_pos -= mlen
            break
        if dele:
            path.append((SEG["DELETE"], dele))
            ref_pos += dele
        if ins:
            path.append((SEG["INSERT"], ins))
            remaining -= ins
        used.append(ind)
    if remaining > 0:
        path.append((SEG["MATCH"], remaining))
    if trail_clip:
        path.append((SEG["SOFT_CLIP"], trail_clip))
    return path, used


def align_cases(n_reads, rng, L=150, K=4, win=400, ref_offset=1000, max_cals=16):
    """List of dicts(read_code, read_qual, ref_seq, ref_offset, cals) with reference-shaped candidate alignments,
    including soft clips, swaps (deletion+insertion at one position), leading/trailing edge insertions,
    non-candidate indels, 'N' and '=' read bases and Q0/Q70 qualities."""
    out = []
    for r in range(n_reads):
        ref_seq = "".join(BASES[i] for i in rng.integers(0, 4, win))
        if rng.random() < 0.2:  # some N in the reference
            j = int(rng.integers(0, win))
            ref_seq = ref_seq[:j] + "N" + ref_seq[j + 1:]
        Lr = int(L if rng.random() < 0.7 else rng.integers(30, 2 * L))
        start = int(rng.integers(20, 60))
        # candidate indels
        indels = []
        p = start + int(rng.integers(5, 20))
        for k in range(K):
            kind = rng.random()
            ind = dict(pos=ref_offset + p, type=INDEL["INDEL"], del_len=0, ins_seq="", is_candidate=int(rng.random() > 0.15))
            if kind < 0.4:
                ind["del_len"] = int(rng.integers(1, 11))
            elif kind < 0.8:
                ind["ins_seq"] = "".join(BASES[i] for i in rng.integers(0, 4, int(rng.integers(1, 11))))
            else:  # swap
                ind["del_len"] = int(rng.integers(1, 6))
                ind["ins_seq"] = "".join(BASES[i] for i in rng.integers(0, 4, int(rng.integers(1, 6))))
            indels.append(ind)
            p += ind["del_len"] + int(rng.integers(3, 25))
        q = rng.choice(np.array([0, 2, 11, 25, 32, 37, 40, 70], np.uint8), Lr,
                       p=[0.01, 0.01, 0.04, 0.10, 0.20, 0.34, 0.29, 0.01])
        span = (ref_seq * 3)[start:start + Lr]
        read = np.array([BAM_CODE[BASES.index(c)] if c in BASES else 15 for c in span], np.uint8)
        err = rng.random(Lr) < 0.02
        read = np.where(err, BAM_CODE[rng.integers(0, 4, Lr)], read).astype(np.uint8)
        read = np.where(rng.random(Lr) < 0.01, 15, read).astype(np.uint8)
        read = np.where(rng.random(Lr) < 0.01, 0, read).astype(np.uint8)

        cals = []
        n_sub = min(1 << K, max_cals)
        masks = rng.permutation(1 << K)[:n_sub]
        for m in masks:
            subset = [indels[k] for k in range(K) if (int(m) >> k) & 1]
            lead = int(rng.integers(1, 8)) if rng.random() < 0.15 else 0
            trail = int(rng.integers(1, 8)) if rng.random() < 0.15 else 0
            if lead + trail >= Lr - 2:
                lead = trail = 0
            path, used = _path_from_indels(ref_offset + start, Lr, subset, lead, trail)
            cal = dict(pos=ref_offset + start, path=path, indels=used, leading=None, trailing=None)
            u = rng.random()
            if u < 0.12 and lead == 0 and path and path[0][0] == SEG["MATCH"] and path[0][1] > 12:
                # leading edge insertion: the read starts inside an insertion; only its last n bases are read
                n = int(rng.integers(1, 6))
                full = "".join(BASES[i] for i in rng.integers(0, 4, n + int(rng.integers(0, 4))))
                path[0] = (SEG["MATCH"], path[0][1] - n)
                path.insert(0, (SEG["INSERT"], n))
                cal["pos"] += n  # the first aligned base moves right by the bases now read from the insertion
                cal["leading"] = dict(pos=ref_offset + start, type=INDEL["INDEL"], del_len=0, ins_seq=full,
                                      is_candidate=int(rng.random() > 0.3))
            elif u < 0.24 and trail == 0 and path and path[-1][0] == SEG["MATCH"] and path[-1][1] > 12:
                n = int(rng.integers(1, 6))
                full = "".join(BASES[i] for i in rng.integers(0, 4, n + int(rng.integers(0, 4))))
                path[-1] = (SEG["MATCH"], path[-1][1] - n)
                path.append((SEG["INSERT"], n))
                ref_end = cal["pos"] + sum(l for t, l in path if t in (SEG["MATCH"], SEG["DELETE"]))
                cal["trailing"] = dict(pos=ref_end, type=INDEL["INDEL"], del_len=0, ins_seq=full,
                                       is_candidate=int(rng.random() > 0.3))
            elif u < 0.30 and path and path[0][0] == SEG["MATCH"]:
                path.insert(0, (SEG["HARD_CLIP"], int(rng.integers(1, 30))))
            cals.append(cal)
        out.append(dict(read_code=read, read_qual=q.astype(np.uint8), ref_seq=ref_seq, ref_offset=ref_offset, cals=cals))
    return out


def build_align_batch(cases):
    b = capi.AlignBuilder()
    for c in cases:
        b.add_read(c["read_code"], c["read_qual"], c["ref_seq"], c["ref_offset"], c["cals"])
    return b.finish()


def align_cases_h64(n_reads, rng, L=150, K=6, win=400, ref_offset=0, noncand_rate=0.1):
    """Reference-shaped twin of align_batch_flat: every subset of K=6 candidate indels, start-pinned, 64 candidates."""
    out = []
    for r in range(n_reads):
        ref_seq = "".join(BASES[i] for i in rng.integers(0, 4, win))
        start = int(rng.integers(20, 60))
        indels = []
        p = start + 8
        for k in range(K):
            p += int(rng.integers(12, 18))
            ind = dict(pos=ref_offset + p, type=INDEL["INDEL"], del_len=0, ins_seq="",
                       is_candidate=int(rng.random() >= noncand_rate))
            if rng.random() < 0.5:
                ind["ins_seq"] = "".join(BASES[i] for i in rng.integers(0, 4, int(rng.integers(1, 11))))
            else:
                ind["del_len"] = int(rng.integers(1, 11))
            indels.append(ind)
        q = rng.choice(QUAL_VALUES, L, p=QUAL_PROBS)
        read = np.array([BAM_CODE[BASES.index(c)] for c in ref_seq[start:start + L]], np.uint8)
        err = rng.random(L) < np.power(10.0, -q.astype(np.float64) / 10.0)
        read = np.where(err, BAM_CODE[rng.integers(0, 4, L)], read).astype(np.uint8)
        cals = []
        for m in range(1 << K):
            subset = [indels[k] for k in range(K) if (m >> k) & 1]
            path, used = _path_from_indels(ref_offset + start, L, subset)
            cals.append(dict(pos=ref_offset + start, path=path, indels=used, leading=None, trailing=None))
        out.append(dict(read_code=read, read_qual=q.astype(np.uint8), ref_seq=ref_seq, ref_offset=ref_offset, cals=cals))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# input D: per-(indel, read) ReadPathScores

def _read_fields(total, rng, tier1_rate=0.9):
    non_ambig = rng.integers(60, 151, total).astype(np.uint16)
    read_length = np.where(rng.random(total) < 0.85, 150, rng.integers(8, 200, total)).astype(np.uint16)
    flags = ((rng.random(total) < tier1_rate).astype(np.uint8) * 1) | (rng.integers(0, 2, total).astype(np.uint8) * 2)
    return non_ambig, read_length, flags.astype(np.uint8)


def readscore_batch(n_indels, rng, depth_mean=40.0, alt_rate=0.2, breakpoint_rate=0.0):
    """ref, indel ~ N(-5,3) clipped at 0 (SURVEY.md 8d input D), some reads with an alternate-indel score."""
    depth = rng.poisson(depth_mean, n_indels).astype(np.int64)
    off = np.zeros(n_indels + 1, np.int64)
    np.cumsum(depth, out=off[1:])
    total = int(off[-1])
    ref = np.minimum(0, rng.normal(-5, 3, total)).astype(np.float32)
    ind = np.minimum(0, rng.normal(-5, 3, total)).astype(np.float32)
    alt = np.where(rng.random(total) < alt_rate, np.minimum(0, rng.normal(-4, 3, total)), np.nan).astype(np.float32)
    na, rl, fl = _read_fields(total, rng)
    is_del = rng.random(n_indels) < 0.5
    ln = rng.integers(1, 30, n_indels)
    del_len = np.where(is_del, ln, 0).astype(np.uint32)
    ins_len = np.where(is_del, np.where(rng.random(n_indels) < 0.1, rng.integers(1, 5, n_indels), 0), ln).astype(np.uint32)
    bp = (rng.random(n_indels) < breakpoint_rate).astype(np.uint8) if breakpoint_rate > 0 else None
    return capi.HostReadScoreBatch(off, ref, ind, alt, na, rl, fl, del_len, ins_len, bp)


def allele_group_batch(n_groups, rng, depth_mean=40.0, missing_rate=0.05, min_alt=1, max_alt=None):
    """max_alt > capi.MAX_ALT: a batch for the wide entry points (multi-sample allele groups, rows of capi.MAX_ALT_WIDE, or of
    capi.MAX_ALT_XWIDE when max_alt > capi.MAX_ALT_WIDE)"""
    max_alt = max_alt or capi.MAX_ALT
    width = capi.MAX_ALT if max_alt <= capi.MAX_ALT else (capi.MAX_ALT_WIDE if max_alt <= capi.MAX_ALT_WIDE else capi.MAX_ALT_XWIDE)
    depth = rng.poisson(depth_mean, n_groups).astype(np.int64)
    off = np.zeros(n_groups + 1, np.int64)
    np.cumsum(depth, out=off[1:])
    total = int(off[-1])
    n_alt = rng.integers(min_alt, max_alt + 1, n_groups).astype(np.uint8)
    ploidy = rng.choice(np.array([1, 2, 2, 2], np.uint8), n_groups)
    is_del = rng.random((n_groups, width)) < 0.5
    ln = rng.integers(1, 30, (n_groups, width))
    del_len = np.where(is_del, ln, 0).astype(np.uint32)
    ins_len = np.where(is_del, 0, ln).astype(np.uint32)
    refl = np.minimum(0, rng.normal(-5, 3, (total, width))).astype(np.float32)
    al = np.minimum(0, rng.normal(-5, 3, (total, width))).astype(np.float32)
    al = np.where(rng.random((total, width)) < missing_rate, np.nan, al).astype(np.float32)
    na, rl, fl = _read_fields(total, rng)
    return capi.HostAlleleGroupBatch(off, n_alt, ploidy, del_len, ins_len, refl, al, na, rl, fl, width=width)


# ----------------------------------------------------------------------------------------------------------------------
# whole-read realignment scenarios (realignAndScoreRead inputs): a reference window, an indel table, reads with input
# alignments of the kinds a mapper produces (true gapped, gapless anchored either end, soft-clipped, edge inserts)

_BASES = "ACGT"
_CODE = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}


def _random_ref(L, rng):
    """random sequence with homopolymer and short-tandem-repeat stretches (so that equivalent indel placements exist)"""
    out = []
    while len(out) < L:
        r = rng.random()
        if r < 0.08:
            out += [_BASES[int(rng.integers(4))]] * int(rng.integers(4, 12))
        elif r < 0.14:
            unit = [_BASES[int(x)] for x in rng.integers(0, 4, int(rng.integers(2, 4)))]
            out += unit * int(rng.integers(3, 7))
        else:
            out += [_BASES[int(x)] for x in rng.integers(0, 4, int(rng.integers(5, 30)))]
    return "".join(out[:L])


def _conflict(a, b):
    """is_indel_conflict for plain indels: closed ranges [pos, pos+del] intersect"""
    return a["pos"] <= b["pos"] + b["del_len"] and b["pos"] <= a["pos"] + a["del_len"]


def realign_scenarios(n, rng, reads_per=6, haplotyping_rate=0.25, max_indels=6, read_len=(40, 101), window=(160, 360), min_indels=1):
    """read_len / window: half-open ranges of the read length and of the reference window; min_indels..max_indels candidate
    indels per scenario (bench.py's a5 leg asks for 150 bp reads over 6 indels: ~64 candidate alignments per read)"""
    out = []
    for _ in range(n):
        L = int(rng.integers(window[0], window[1]))
        off = int(rng.choice([0, 0, 1000, 25000]))
        ref = _random_ref(L, rng)
        n_ind = int(rng.integers(min_indels, max_indels + 1))
        indels = []
        for _k in range(n_ind * 3):
            if len(indels) >= n_ind:
                break
            p = off + int(rng.integers(25, L - 25))
            r = rng.random()
            if indels and rng.random() < 0.3:  # a shifted copy of an existing indel (same type/size, nearby)
                src = indels[int(rng.integers(len(indels)))]
                d = dict(src)
                d["pos"] = src["pos"] + int(rng.integers(-3, 4))
                if d["ins_seq"]:
                    i0 = d["pos"] - off
                    if rng.random() < 0.7 and 0 <= i0 and i0 + len(d["ins_seq"]) <= L:
                        d["ins_seq"] = ref[i0:i0 + len(d["ins_seq"])]
            elif r < 0.45:
                d = dict(pos=p, type=INDEL["INDEL"], del_len=int(rng.choice([1, 1, 2, 3, 5, 8, 15, 30])), ins_seq="")
            elif r < 0.9:
                ln = int(rng.choice([1, 1, 2, 3, 4, 6, 10]))
                if rng.random() < 0.5:
                    seq = ref[p - off:p - off + ln]
                else:
                    seq = "".join(_BASES[int(x)] for x in rng.integers(0, 4, ln))
                d = dict(pos=p, type=INDEL["INDEL"], del_len=0, ins_seq=seq)
            else:
                ln = int(rng.integers(1, 5))
                d = dict(pos=p, type=INDEL["INDEL"], del_len=int(rng.integers(1, 6)),
                         ins_seq="".join(_BASES[int(x)] for x in rng.integers(0, 4, ln)))
            if d["pos"] < off + 10 or d["pos"] + d["del_len"] > off + L - 10:
                continue
            if any((d["pos"], d["del_len"], d["ins_seq"]) == (e["pos"], e["del_len"], e["ins_seq"]) for e in indels):
                continue
            d["is_candidate"] = int(rng.random() < 0.8)
            indels.append(d)
        is_hap = rng.random() < haplotyping_rate
        for d in indels:
            if is_hap and rng.random() < 0.8:
                d["arid"] = int(rng.integers(0, 2))
                d["hap"] = int(rng.integers(0, 4))
                d["bypass"] = int(rng.random() < 0.2)
                d["forced"] = int(rng.random() < 0.1)
                d["ndfr"] = int(rng.random() < 0.15)
        reads = []
        for _r in range(reads_per):
            # haplotype = a non-conflicting subset of the indels
            hap = []
            for i in rng.permutation(len(indels)):
                if rng.random() < 0.5 and not any(_conflict(indels[i], indels[j]) for j in hap):
                    hap.append(int(i))
            hap.sort(key=lambda i: (indels[i]["pos"], indels[i]["del_len"]))
            rl = int(rng.integers(read_len[0], read_len[1]))
            start = off + int(rng.integers(0, max(1, L - rl - 35)))
            # walk the haplotype from `start`
            seq, path, used, p, hi = [], [], [], start, 0
            while hi < len(hap) and indels[hap[hi]]["pos"] <= start:
                hi += 1

            def push(t, ln):
                if ln <= 0:
                    return
                if path and path[-1][0] == t:
                    path[-1] = (t, path[-1][1] + ln)
                else:
                    path.append((t, ln))
            fixed = set()  # read positions that belong to inserted sequence (kept free of mismatches)
            while len(seq) < rl:
                nxt = indels[hap[hi]]["pos"] if hi < len(hap) else off + L
                m = min(nxt - p, rl - len(seq), off + L - p)
                if m > 0:
                    seq += list(ref[p - off:p - off + m])
                    push(SEG["MATCH"], m)
                    p += m
                if len(seq) >= rl or p >= off + L or hi >= len(hap):
                    if p >= off + L:
                        break
                    if hi >= len(hap) and len(seq) < rl:
                        continue
                    break
                d = indels[hap[hi]]
                hi += 1
                used.append(hap[hi - 1])
                if d["del_len"]:
                    push(SEG["DELETE"], d["del_len"])
                    p += d["del_len"]
                if d["ins_seq"]:
                    k = min(len(d["ins_seq"]), rl - len(seq))
                    fixed.update(range(len(seq), len(seq) + k))
                    seq += list(d["ins_seq"][:k])
                    push(SEG["INSERT"], k)
            rl = len(seq)
            if rl < 20:
                continue
            while path and path[-1][0] == SEG["DELETE"]:
                path.pop()
            for i in range(rl):
                if i not in fixed and rng.random() < 0.015:
                    seq[i] = _BASES[(_BASES.index(seq[i]) + int(rng.integers(1, 4))) % 4]
                if i not in fixed and rng.random() < 0.004:
                    seq[i] = "N"
            code = np.array([_CODE[c] for c in seq], np.uint8)
            qual = rng.integers(2, 41, rl).astype(np.uint8)
            ref_len = sum(l for t, l in path if t in (SEG["MATCH"], SEG["DELETE"]))
            kind = rng.random()
            observed = list(used)
            if kind < 0.45:
                in_pos, in_path = start, list(path)
            elif kind < 0.6:   # gapless, anchored at the read start
                in_pos, in_path, observed = start, [(SEG["MATCH"], rl)], []
            elif kind < 0.75:  # gapless, anchored at the read end
                in_pos, in_path, observed = start + ref_len - rl, [(SEG["MATCH"], rl)], []
            else:              # true alignment up to the first indel, the rest soft-clipped (either side)
                observed = []
                if len(path) == 1:
                    c = int(rng.integers(1, 12))
                    in_pos, in_path = start, [(SEG["MATCH"], rl - c), (SEG["SOFT_CLIP"], c)]
                elif rng.random() < 0.5:
                    m0 = path[0][1]
                    in_pos, in_path = start, [(SEG["MATCH"], m0), (SEG["SOFT_CLIP"], rl - m0)]
                else:
                    m1 = path[-1][1] if path[-1][0] == SEG["MATCH"] else 0
                    if m1 == 0:
                        in_pos, in_path = start, list(path)
                        observed = list(used)
                    else:
                        in_pos, in_path = start + ref_len - m1, [(SEG["SOFT_CLIP"], rl - m1), (SEG["MATCH"], m1)]
            if in_pos < 0:
                continue
            lvl = rng.random()
            rr = (max(0, off - 60), off + L + 60)
            if rng.random() < 0.08:
                rr = (max(0, start - int(rng.integers(0, 12))), start + ref_len + int(rng.integers(0, 12)))
            reads.append(dict(code=code, qual=qual, pos=int(in_pos), path=in_path, is_fwd=bool(rng.random() < 0.5),
                              map_level=1 if lvl < 0.8 else (2 if lvl < 0.93 else 3), observed=observed,
                              realign_range=rr))
        out.append(dict(ref_seq=ref, ref_offset=off, indels=indels, reads=reads, is_haplotyping_enabled=int(is_hap),
                        min_read_bp_flank=int(rng.choice([5, 5, 5, 1]))))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# reads with final alignments for the pileup row (a8)

class ReadBatch:
    """SoA of reads + best alignments in pileup (read-buffer) order; mirrors sk_read_batch"""

    def __init__(self, read_off, read_code, read_qual, path_off, path, pos, is_fwd, mapq, map_level, ref_seq, ref_offset,
                 cand_snv_mask=None):
        self.read_off = np.ascontiguousarray(read_off, np.int64)
        self.read_code = np.ascontiguousarray(read_code, np.uint8)
        self.read_qual = np.ascontiguousarray(read_qual, np.uint8)
        self.path_off = np.ascontiguousarray(path_off, np.int64)
        self.path = np.ascontiguousarray(path, np.uint32).reshape(-1, 2)
        self.pos = np.ascontiguousarray(pos, np.int32)
        self.is_fwd = np.ascontiguousarray(is_fwd, np.uint8)
        self.mapq = np.ascontiguousarray(mapq, np.uint8)
        self.map_level = np.ascontiguousarray(map_level, np.uint8)
        self.ref_seq = ref_seq
        self.ref_offset = int(ref_offset)
        self.cand_snv_mask = None if cand_snv_mask is None else np.ascontiguousarray(cand_snv_mask, np.uint8)
        self.n_reads = len(self.pos)
        self.n_bases = int(self.read_off[-1]) if self.n_reads else 0

    @staticmethod
    def from_reads(reads, ref_seq, ref_offset, cand_snv_mask=None):
        """reads: dicts(code, qual, pos, path=[(type,len)], is_fwd, mapq, map_level)"""
        ro, po = [0], [0]
        for r in reads:
            ro.append(ro[-1] + len(r["code"]))
            po.append(po[-1] + len(r["path"]))
        cat = lambda k, dt: np.concatenate([np.asarray(r[k], dt) for r in reads]) if reads else np.zeros(0, dt)
        path = np.array([s for r in reads for s in r["path"]], np.uint32).reshape(-1, 2)
        return ReadBatch(ro, cat("code", np.uint8), cat("qual", np.uint8), po, path, [r["pos"] for r in reads],
                         [int(r["is_fwd"]) for r in reads], [r["mapq"] for r in reads], [r["map_level"] for r in reads],
                         ref_seq, ref_offset, cand_snv_mask)


def pileup_reads(n_reads, rng, ref_len=600, ref_offset=5000, read_len=(36, 151), indel_rate=0.25, clip_rate=0.15,
                 submapped_rate=0.08, tier2_rate=0.1, burst_rate=0.15, sorted_by_pos=True):
    """reads over a random reference window: mismatches (with bursts that trip the mismatch-density filter), N runs at
    either end, soft clips, insertions/deletions (also at the edges), assorted MAPQ and mapping tiers"""
    ref = _random_ref(ref_len, rng)
    reads = []
    for _ in range(n_reads):
        L = int(rng.integers(read_len[0], read_len[1]))
        start = ref_offset + int(rng.integers(-10, max(1, ref_len - L // 2)))
        path, seq, p = [], [], start
        lead_clip = int(rng.integers(1, 12)) if rng.random() < clip_rate else 0
        trail_clip = int(rng.integers(1, 12)) if rng.random() < clip_rate else 0

        def rbase(pp):
            i = pp - ref_offset
            return ref[i] if 0 <= i < ref_len else "N"
        if lead_clip:
            path.append((SEG["SOFT_CLIP"], lead_clip))
            seq += [_BASES[int(x)] for x in rng.integers(0, 4, lead_clip)]
        body = L - lead_clip - trail_clip
        n_ind = int(rng.integers(1, 4)) if rng.random() < indel_rate else 0
        cuts = sorted(int(x) for x in rng.integers(1, max(2, body - 1), n_ind)) if body > 8 else []
        prev = 0
        edge_lead_del = rng.random() < 0.03
        if edge_lead_del:
            path.append((SEG["DELETE"], int(rng.integers(1, 4))))
            p += path[-1][1]
        for c in cuts + [body]:
            m = c - prev
            if m > 0:
                path.append((SEG["MATCH"], m))
                seq += [rbase(p + j) for j in range(m)]
                p += m
            prev = c
            if c < body:
                if rng.random() < 0.5:
                    d = int(rng.choice([1, 2, 3, 8, 20]))
                    path.append((SEG["DELETE"], d))
                    p += d
                else:
                    k = min(int(rng.choice([1, 2, 5])), body - c)
                    if k > 0:
                        path.append((SEG["INSERT"], k))
                        seq += [_BASES[int(x)] for x in rng.integers(0, 4, k)]
                        prev = c + k
        if trail_clip:
            path.append((SEG["SOFT_CLIP"], trail_clip))
            seq += [_BASES[int(x)] for x in rng.integers(0, 4, trail_clip)]
        # merge adjacent equal segment types, fix the length
        merged = []
        for t, l in path:
            if merged and merged[-1][0] == t:
                merged[-1] = (t, merged[-1][1] + l)
            else:
                merged.append((t, l))
        path = merged
        seq = seq[:sum(l for t, l in path if t in (SEG["MATCH"], SEG["INSERT"], SEG["SOFT_CLIP"]))]
        L = len(seq)
        if L < 10:
            continue
        # mismatches: background + an occasional burst
        for i in range(L):
            if seq[i] != "N" and rng.random() < 0.01:
                seq[i] = _BASES[(_BASES.index(seq[i]) + int(rng.integers(1, 4))) % 4]
        if rng.random() < burst_rate:
            c0 = int(rng.integers(0, L))
            for i in range(c0, min(L, c0 + int(rng.integers(3, 25)))):
                if seq[i] != "N" and rng.random() < 0.4:
                    seq[i] = _BASES[(_BASES.index(seq[i]) + 1) % 4]
        if rng.random() < 0.1:
            for i in range(int(rng.integers(1, 6))):
                seq[L - 1 - i] = "N"
        if rng.random() < 0.1:
            for i in range(int(rng.integers(1, 6))):
                seq[i] = "N"
        if rng.random() < 0.05:
            seq[int(rng.integers(0, L))] = "N"
        u = rng.random()
        level = 3 if u < submapped_rate else (2 if u < submapped_rate + tier2_rate else 1)
        reads.append(dict(code=np.array([_CODE[c] for c in seq], np.uint8), qual=rng.integers(2, 42, L).astype(np.uint8),
                          pos=int(p - sum(l for t, l in path if t in (SEG["MATCH"], SEG["DELETE"]))), path=path,
                          is_fwd=bool(rng.random() < 0.5), mapq=int(rng.choice([0, 3, 10, 20, 40, 60, 60, 60, 85, 255])),
                          map_level=level))
    if sorted_by_pos:
        reads.sort(key=lambda r: r["pos"])
    return reads, ref, ref_offset


def pileup_reads_flat(n_reads, rng, L=150, depth=40.0, del_rate=0.05, mismatch_rate=0.01):
    """bench-sized ReadBatch, vectorised: position-sorted 150 bp reads at the given depth over a random reference;
    5% carry one internal deletion; 1% substitutions; Q from a 2..40 mix; MAPQ 60 tier1"""
    n_loci = int(n_reads * L / depth)
    ref_codes = rng.integers(0, 4, n_loci + 2 * L + 64).astype(np.uint8)
    pos = np.sort(rng.integers(0, n_loci, n_reads)).astype(np.int32)
    has_del = rng.random(n_reads) < del_rate
    cut = rng.integers(20, L - 20, n_reads)
    dlen = rng.integers(1, 9, n_reads)
    idx = np.arange(L)[None, :]
    shift = np.where(has_del[:, None] & (idx >= cut[:, None]), dlen[:, None], 0)
    base = ref_codes[pos[:, None] + idx + shift]
    mm = rng.random((n_reads, L)) < mismatch_rate
    base = np.where(mm, (base + rng.integers(1, 4, (n_reads, L))) % 4, base).astype(np.uint8)
    code = (1 << base).astype(np.uint8)
    qual = rng.choice(np.array([2, 12, 22, 27, 32, 37, 37, 40], np.uint8), (n_reads, L))
    nseg = np.where(has_del, 3, 1)
    path_off = np.concatenate([[0], np.cumsum(nseg)]).astype(np.int64)
    path = np.zeros((int(path_off[-1]), 2), np.uint32)
    first = path_off[:-1]
    path[first, 0] = SEG["MATCH"]
    path[first, 1] = np.where(has_del, cut, L)
    d = first[has_del]
    path[d + 1, 0] = SEG["DELETE"]
    path[d + 1, 1] = dlen[has_del]
    path[d + 2, 0] = SEG["MATCH"]
    path[d + 2, 1] = L - cut[has_del]
    ref = "".join(np.array(list("ACGT"))[ref_codes])
    return ReadBatch(np.arange(n_reads + 1, dtype=np.int64) * L, code.reshape(-1), qual.reshape(-1), path_off, path, pos,
                     rng.integers(0, 2, n_reads).astype(np.uint8), np.full(n_reads, 60, np.uint8), np.ones(n_reads, np.uint8),
                     ref, 0), n_loci


def align_pairs(n, rng, ref_len=(100, 270), max_edits=3):
    """(haplotype, reference segment) string pairs for GlobalAligner: the reference segment with a few indels applied."""
    bases = np.array(list("ACGT"))

    def seq(k):
        return "".join(bases[rng.integers(0, 4, k)])

    pairs = []
    for _ in range(n):
        r = seq(int(rng.integers(ref_len[0], ref_len[1])))
        q = list(r)
        for _k in range(int(rng.integers(0, max_edits + 1))):
            if len(q) < 12:
                break
            p = int(rng.integers(5, len(q) - 5))
            if rng.random() < 0.5:
                del q[p:p + int(rng.integers(1, 12))]
            else:
                q[p:p] = list(seq(int(rng.integers(1, 12))))
        pairs.append(("".join(q), r))
    return pairs


def active_region_scenarios(n, rng):
    """inputs of ActiveRegionProcessor::discoverIndelsAndMismatches: a repeat-rich reference segment, an active region
    in it, and a haplotype = the region's sequence with a few SNVs / insertions / deletions (often of repeat units, so
    that left-shifting has something to do, sometimes up to the region's or the segment's edge)"""
    out = []
    while len(out) < n:
        L = int(rng.integers(120, 400))
        parts = []
        while sum(map(len, parts)) < L:
            r = rng.random()
            if r < 0.25:
                parts.append(_BASES[int(rng.integers(0, 4))] * int(rng.integers(3, 15)))
            elif r < 0.45:
                unit = "".join(_BASES[int(x)] for x in rng.integers(0, 4, int(rng.integers(2, 5))))
                parts.append(unit * int(rng.integers(2, 8)))
            elif r < 0.5:
                parts.append("N" * int(rng.integers(1, 4)))
            else:
                parts.append("".join(_BASES[int(x)] for x in rng.integers(0, 4, int(rng.integers(3, 20)))))
        ref = "".join(parts)[:L]
        off = int(rng.choice([0, 0, 1000, 52000]))
        size = int(rng.integers(20, min(200, L - 4)))
        b = int(rng.integers(0, L - size + 1))
        e = b + size
        seg = ref[b:e]
        hap = list(seg)
        for _ in range(int(rng.integers(1, 5))):
            if len(hap) < 12:
                break
            p = int(rng.integers(0, len(hap)))
            r = rng.random()
            if r < 0.3:
                hap[p] = _BASES[(_BASES.find(hap[p]) + int(rng.integers(1, 4))) % 4] if hap[p] in _BASES else "A"
            elif r < 0.65:
                k = int(rng.choice([1, 1, 2, 3, 4, 8, 20, 60]))
                src = hap[max(0, p - k):p] if rng.random() < 0.6 and p >= k else [_BASES[int(x)] for x in rng.integers(0, 4, k)]
                hap[p:p] = src
            else:
                k = int(rng.choice([1, 1, 2, 3, 4, 8, 20, 60]))
                del hap[p:p + k]
        hap = "".join(hap)
        if hap == seg or len(hap) < 1:
            continue
        prev = int(rng.choice([off + b, off + b - 5, off + b + 1, off, off + b + 10]))
        out.append(dict(ref_seq=ref, ref_offset=off, ar_begin=off + b, ar_end=off + e, prev_ar_end=prev,
                        max_indel_size=int(rng.choice([49, 49, 49, 10])), haplotype=hap))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# the feed: alignments for normalizeAlignment -- indels inside repeats placed anywhere in their equivalence range, insert+delete
# pairs that partly cancel, indels at the alignment's edges, soft / hard clips

def normalize_cases(n, rng):
    """-> list of dict(ref_seq, ref_offset, read (ACGTN string), code, pos, path)"""
    out = []
    seg = dict(M=1, I=2, D=3, N=4, S=5, H=6, P=7, EQ=8, X=9)
    char2code = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}
    for _ in range(n):
        # a reference rich in homopolymers and short tandem repeats
        parts = []
        while sum(len(p) for p in parts) < 260:
            r = rng.random()
            if r < 0.35:
                parts.append(BASES[int(rng.integers(0, 4))] * int(rng.integers(3, 12)))
            elif r < 0.6:
                unit = "".join(BASES[i] for i in rng.integers(0, 4, int(rng.integers(2, 5))))
                parts.append(unit * int(rng.integers(2, 7)))
            else:
                parts.append("".join(BASES[i] for i in rng.integers(0, 4, int(rng.integers(4, 20)))))
        ref = "".join(parts)
        off = int(rng.choice([0, 0, 500, 100000]))
        start = int(rng.integers(0, 40))
        rp = start          # position in ref (0-based inside ref)
        read = []
        path = []
        if rng.random() < 0.2:
            path.append((seg["H"], int(rng.integers(1, 9))))
        if rng.random() < 0.25:
            k = int(rng.integers(1, 8))
            read += [BASES[i] for i in rng.integers(0, 4, k)]
            path.append((seg["S"], k))
        if rng.random() < 0.15:  # leading edge indel
            if rng.random() < 0.5:
                k = int(rng.integers(1, 6))
                if rng.random() < 0.5 and rp >= k:
                    read += list(ref[rp - k:rp])      # an edge insertion that is really a match one step to the left
                else:
                    read += [BASES[i] for i in rng.integers(0, 4, k)]
                path.append((seg["I"], k))
            else:
                k = int(rng.integers(1, 6))
                path.append((seg["D"], k))
                rp += k
        n_ev = int(rng.integers(0, 5))
        for e in range(n_ev + 1):
            m = int(rng.integers(4, 45))
            m = min(m, len(ref) - rp - 30)
            if m <= 0:
                break
            chunk = list(ref[rp:rp + m])
            for i in range(len(chunk)):
                if rng.random() < 0.03:
                    chunk[i] = BASES[int(rng.integers(0, 4))] if rng.random() < 0.8 else "N"
            read += chunk
            t = seg["M"] if rng.random() < 0.85 else (seg["EQ"] if rng.random() < 0.5 else seg["X"])
            path.append((t, m))
            rp += m
            if e == n_ev:
                break
            r = rng.random()
            if r < 0.35:      # deletion (often of a repeat unit -> shiftable)
                k = int(rng.integers(1, 9))
                path.append((seg["D"], k))
                rp += k
            elif r < 0.7:     # insertion: a copy of the following / preceding reference bases (shiftable) or random
                k = int(rng.integers(1, 9))
                q = rng.random()
                if q < 0.4:
                    ins = list(ref[rp:rp + k])
                elif q < 0.7 and rp >= k:
                    ins = list(ref[rp - k:rp])
                else:
                    ins = [BASES[i] for i in rng.integers(0, 4, k)]
                read += ins
                path.append((seg["I"], len(ins)))
            elif r < 0.9:     # insertion and deletion side by side, partly or wholly cancelling
                kd, ki = int(rng.integers(1, 7)), int(rng.integers(1, 7))
                ins = list(ref[rp:rp + ki]) if rng.random() < 0.6 else [BASES[i] for i in rng.integers(0, 4, ki)]
                if rng.random() < 0.5:
                    path += [(seg["I"], len(ins)), (seg["D"], kd)]
                else:
                    path += [(seg["D"], kd), (seg["I"], len(ins))]
                read += ins
                rp += kd
            else:             # two runs of the same kind split in two segments (the cleaner merges them)
                k1, k2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
                path += [(seg["D"], k1), (seg["D"], k2)]
                rp += k1 + k2
        if rng.random() < 0.12:  # trailing edge indel
            if rng.random() < 0.5:
                k = int(rng.integers(1, 6))
                ins = list(ref[rp:rp + k]) if rng.random() < 0.5 else [BASES[i] for i in rng.integers(0, 4, k)]
                read += ins
                path.append((seg["I"], len(ins)))
            else:
                path.append((seg["D"], int(rng.integers(1, 6))))
        if rng.random() < 0.25:
            k = int(rng.integers(1, 8))
            read += [BASES[i] for i in rng.integers(0, 4, k)]
            path.append((seg["S"], k))
        if rng.random() < 0.2:
            path.append((seg["H"], int(rng.integers(1, 9))))
        if not any(t in (1, 8, 9) for t, _ in path):
            continue
        read = "".join(read)
        out.append(dict(ref_seq=ref, ref_offset=off, read=read, code=np.array([char2code[c] for c in read], np.uint8), pos=off + start, path=path))
    return out


def gvcf_sites(n, rng):
    """a run of germline sites of one sample as the gVCF writer's block logic sees them (capi.GVCF_SITE_DTYPE): mostly hom-ref
    sites whose depth and GQX drift, with variant / filtered / uncovered / non-compressible sites, gaps and flushes in between"""
    s = np.zeros(n, [("pos", "<i4"), ("is_compressible", "u1"), ("is_gqx", "u1"), ("ploidy", "u1"), ("flush_before", "u1"), ("gt", "<u4"),
                     ("locus_filters", "<u4"), ("sample_filters", "<u4"), ("gqx", "<i4"), ("used_basecalls", "<u4"), ("unused_basecalls", "<u4"),
                     ("is_ref_unknown", "u1")])
    pos = int(rng.integers(0, 1000))
    depth, gqx, unused = float(rng.integers(0, 60)), float(rng.integers(0, 90)), float(rng.integers(0, 6))
    ploidy, lf, sf, noise = 2, 0, 0, 1.0
    for i in range(n):
        pos += 1 if rng.random() > 0.01 else int(rng.integers(2, 50))
        if rng.random() < 0.03:  # a new regime
            depth, gqx, unused = float(rng.integers(0, 120)), float(rng.integers(0, 120)), float(rng.integers(0, 12))
            noise = float(rng.choice([0.05, 0.3, 1.0]))  # quiet regimes give long blocks
        if rng.random() < 0.01:
            ploidy = int(rng.choice([1, 2]))
        if rng.random() < 0.02:
            lf = int(rng.choice([0, 0, 1 << 3, 1 << 6, (1 << 3) | (1 << 10)]))
        if rng.random() < 0.02:
            sf = int(rng.choice([0, 0, 1 << 3, 1 << 2]))
        depth = max(0.0, depth + rng.normal(0, 1.5 * noise))
        gqx = max(0.0, gqx + rng.normal(0, 3 * noise))
        unused = max(0.0, unused + rng.normal(0, 0.5 * noise))
        used = int(round(depth)) if rng.random() > 0.02 else 0
        r = rng.random()
        if ploidy == 2:
            a0, a1 = (0, 0) if r > 0.04 else ((0, 1) if r > 0.015 else (1, 1))
            gt = (2 << 24) | (a0 << 8) | a1
        else:
            a0 = 0 if r > 0.03 else 1
            gt = (1 << 24) | (a0 << 8)
        ref_unknown = rng.random() < 0.01
        s[i] = (pos, 0 if rng.random() < 0.03 else 1, 0 if (ref_unknown or used == 0) else 1, ploidy, 1 if rng.random() < 0.01 else 0, gt, lf, sf,
                int(round(gqx)), used, int(round(unused)) if rng.random() > 0.05 else 0, 1 if ref_unknown else 0)
    return s
