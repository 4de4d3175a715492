#!/usr/bin/env python
"""Generate tests/golden/*.npz|*.pkl: outputs of the REFERENCE ITSELF (oracle/_ref/libstrelka_ref.so, i.e. the
reference's own translation units compiled by oracle/Makefile) on seeded inputs.

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py
The fixtures travel with the repo, so the oracle restatement stays pinned to reference outputs where /root/reference
does not exist (the GPU box)."""
import ctypes as C
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from strelka_amd import capi, synth  # noqa: E402

vp = C.c_void_p


def _p(a):
    return a.ctypes.data_as(vp)


def pathb_vectors(rng):
    """inputs + the REFERENCE's outputs for the scalar helpers, std::sort tie order and every path-B function"""
    R = pyoracle.ref()
    assert R is not None, "oracle/_ref/libstrelka_ref.so missing"
    out = {}

    # ---- scalar helpers / tables
    tabs = [np.zeros(71) for _ in range(3)]
    R.ref_get_qscore_tables(*[_p(t) for t in tabs])
    out["q2p"], out["q2lncompe"], out["q2lne"] = tabs
    out["mapped_q"] = np.array([[R.ref_mapped_qscore(q, m) for q in range(71)] for m in range(0, 91, 5)], np.int32)
    probs = np.concatenate([10.0 ** -rng.uniform(0, 320, 200), rng.random(100)])
    out["qphred_in"] = probs
    out["qphred_out"] = np.array([R.ref_error_prob_to_qphred(C.c_double(p)) for p in probs], np.int32)
    lnp = (-rng.uniform(0, 120, 300)).astype(np.float32)
    out["lnqphred_in"] = lnp
    out["lnqphred_out"] = np.array([R.ref_ln_error_prob_to_qphred_f(C.c_float(x)) for x in lnp], np.int32)
    pairs = (-rng.uniform(0, 60, (300, 2)))
    out["logsum_in"] = pairs
    out["logsum_out"] = np.array([R.ref_log_sum2(C.c_double(a), C.c_double(b)) for a, b in pairs])
    out["logsumf_out"] = np.array([R.ref_log_sum2f(C.c_float(a), C.c_float(b)) for a, b in pairs.astype(np.float32)], np.float32)
    pri = np.zeros(200, np.float32)
    R.ref_germline_lnpriors(C.c_double(0.001), _p(pri))
    out["germline_lnpriors_theta0.001"] = pri

    # ---- std::sort tie order
    keys, perms = [], []
    for n in list(range(0, 36)) + [50, 64, 100, 257, 1000]:
        for rep in range(3):
            key = rng.integers(0, int(rng.integers(1, 64)), size=max(n, 1)).astype(np.uint16)
            idx = np.arange(n, dtype=np.uint32)
            R.ref_sort_idx_by_key_desc(_p(idx), n, _p(key))
            keys.append(key[:n].copy())
            perms.append(idx)
    out["sort_keys"] = np.array(keys, dtype=object)
    out["sort_perms"] = np.array(perms, dtype=object)

    # ---- germline: adjust_joint_eprob + position_snp_call_pprob_digt
    parts = [synth.pileups(150, rng, het_rate=0.1, hom_rate=0.05, filter_rate=0.03),
             synth.pileups(60, rng, depth_mean=3.0, het_rate=0.2), synth.pileups(30, rng, depth_mean=120.0, nmm_rate=0.3),
             synth.pileups(60, rng, noise=0.6, nmm_rate=0.1)]
    off = [np.zeros(1, np.int64)]
    calls, refb = [], []
    base = 0
    for p in parts:
        off.append(p.call_off[1:] + base)
        base += p.call_off[-1]
        calls.append(p.calls)
        refb.append(p.ref_base)
    pb = capi.HostPileupBatch(np.concatenate(off), np.concatenate(calls), np.concatenate(refb))
    pb.ref_base[::37] = 4
    ploidy = rng.choice(np.array([1, 2, 2], np.uint8), pb.n_loci)
    de = np.zeros(len(pb.calls), np.float32)
    digt = np.zeros(pb.n_loci, pyoracle.DIGT_CALL_DTYPE)
    for l in range(pb.n_loci):
        s, e = int(pb.call_off[l]), int(pb.call_off[l + 1])
        c = np.ascontiguousarray(pb.calls[s:e])
        d = np.zeros(e - s, np.float32)
        R.ref_adjust_joint_eprob(_p(c), e - s, C.c_double(.35), C.c_double(.6), 1, C.c_double(.25), _p(d))
        de[s:e] = d
        # the caller consumes the CLEANED pileup (filtered calls removed, PileupCleaner.cpp:28-66)
        keep = ((c >> 12) & 1) == 0
        cc, dd = np.ascontiguousarray(c[keep]), np.ascontiguousarray(d[keep])
        row = np.zeros(1, pyoracle.DIGT_CALL_DTYPE)
        R.ref_position_snp_call_pprob_digt(_p(cc), _p(dd), len(cc), int(pb.ref_base[l]), int(ploidy[l]), C.c_double(0.001), _p(row))
        digt[l] = row[0]
    out.update(g_call_off=pb.call_off, g_calls=pb.calls, g_ref_base=pb.ref_base, g_ploidy=ploidy, g_de=de,
               g_digt=digt.view(np.uint8).reshape(pb.n_loci, -1))

    # ---- somatic SNV: sample likelihoods + posterior
    n, t = synth.somatic_pileups(250, rng, somatic_rate=0.1, het_rate=0.1)
    nl = np.zeros((n.n_loci, 30), np.float32)
    tl = np.zeros((n.n_loci, 30), np.float32)
    res = np.zeros((n.n_loci, 4), np.int64)
    lnp3 = np.zeros(3, np.float32)
    R.ref_germline_genotype_log_prior(C.c_double(0.001), _p(lnp3))
    import math
    for l in range(n.n_loci):
        for b, dst, strand in ((n, nl, 0), (t, tl, 1)):
            s, e = int(b.call_off[l]), int(b.call_off[l + 1])
            c = np.ascontiguousarray(b.calls[s:e])
            row = np.zeros(30, np.float32)
            R.ref_somatic_sample_lhood(_p(c), e - s, int(n.ref_base[l]), strand, _p(row))
            dst[l] = row
        mg, q, fq, nt = C.c_uint32(), C.c_int32(), C.c_int32(), C.c_uint32()
        R.ref_calculate_result_set_grid(C.c_float(0.15), C.c_float(math.log(5e-10)), C.c_float(math.log1p(-5e-10)),
                                        _p(nl[l]), _p(tl[l]), _p(lnp3), C.c_float(math.log1p(-1e-4)),
                                        C.c_float(math.log(1e-4)), C.byref(mg), C.byref(q), C.byref(fq), C.byref(nt))
        res[l] = (mg.value, q.value, fq.value, nt.value)
    out.update(s_n_off=n.call_off, s_n_calls=n.calls, s_t_off=t.call_off, s_t_calls=t.calls, s_ref_base=n.ref_base,
               s_normal_lhood=nl, s_tumor_lhood=tl, s_result=res, s_lnprior3=lnp3)

    # ---- indels: 21-state grid likelihoods + allele-group genotype likelihoods
    rb = synth.readscore_batch(80, rng, depth_mean=60.0)
    bases = "ACGT"
    grid = np.zeros((2, rb.n_indels, 21))
    for i in range(rb.n_indels):
        s, e = int(rb.read_off[i]), int(rb.read_off[i + 1])
        ins = ("A" * int(rb.ins_len[i])).encode()
        for t2 in (0, 1):
            row = np.zeros(21)
            args = [np.ascontiguousarray(x[s:e]) for x in (rb.ref_lnp, rb.indel_lnp, rb.alt_lnp, rb.non_ambig, rb.read_length)]
            t1 = np.ascontiguousarray(rb.read_flags[s:e] & 1)
            R.ref_indel_grid_lhood(e - s, *[_p(x) for x in args], _p(t1), int(rb.del_len[i]), ins, 5, C.c_double(0.5), t2, 1, _p(row))
            grid[t2, i] = row
    out.update(i_read_off=rb.read_off, i_ref=rb.ref_lnp, i_indel=rb.indel_lnp, i_alt=rb.alt_lnp, i_na=rb.non_ambig,
               i_rl=rb.read_length, i_flags=rb.read_flags, i_del=rb.del_len, i_ins=rb.ins_len, i_grid=grid)
    ab = synth.allele_group_batch(120, rng, depth_mean=45.0)
    glh = np.zeros((ab.n_groups, 10))
    gcnt = np.zeros((ab.n_groups, 2, 5), np.uint32)
    for g in range(ab.n_groups):
        s, e = int(ab.read_off[g]), int(ab.read_off[g + 1])
        A, pl = int(ab.n_alt[g]), int(ab.ploidy[g])
        G = A + 1 if pl == 1 else (A + 1) * (A + 2) // 2
        refl = np.ascontiguousarray(ab.ref_lnp[s:e, :A])
        al = np.ascontiguousarray(ab.allele_lnp[s:e, :A])
        t1 = np.ascontiguousarray(ab.read_flags[s:e] & 1)
        fw = np.ascontiguousarray((ab.read_flags[s:e] >> 1) & 1)
        inss = (C.c_char_p * A)(*[("C" * int(ab.ins_len[g, k])).encode() for k in range(A)])
        dl = np.ascontiguousarray(ab.del_len[g, :A])
        ol, oc = np.zeros(G), np.zeros(2 * (A + 2), np.uint32)
        R.ref_allele_group_genotype_lhoods(e - s, A, _p(refl), _p(al), _p(np.ascontiguousarray(ab.non_ambig[s:e])),
                                           _p(np.ascontiguousarray(ab.read_length[s:e])), _p(t1), _p(fw), _p(dl), inss, pl,
                                           5, C.c_double(0.25), _p(ol), _p(oc))
        glh[g, :G] = ol
        oc = oc.reshape(2, A + 2)
        gcnt[g, :, :A + 1] = oc[:, :A + 1]
        gcnt[g, :, A + 1] = oc[:, A + 1]
    out.update(a_read_off=ab.read_off, a_n_alt=ab.n_alt, a_ploidy=ab.ploidy, a_del=ab.del_len, a_ins=ab.ins_len,
               a_ref=ab.ref_lnp, a_allele=ab.allele_lnp, a_na=ab.non_ambig, a_rl=ab.read_length, a_flags=ab.read_flags,
               a_lhood=glh, a_counts=gcnt)

    return out


def main():
    pyoracle.build(ref=True)
    rng = np.random.default_rng(20240925)
    out = pathb_vectors(rng)
    np.savez_compressed(os.path.join(HERE, "pathb_reference.npz"), **out)

    # ---- hot path A: scoreCandidateAlignment of the reference on reference-shaped candidate alignments
    cases = synth.align_cases(48, rng) + synth.align_cases_h64(4, rng)
    scores = pyoracle.ref_score_cases(cases)
    with open(os.path.join(HERE, "patha_scores_reference.pkl"), "wb") as f:
        pickle.dump(dict(cases=cases, scores=scores), f, protocol=4)
    print("wrote", sorted(os.listdir(HERE)))


def realign_golden():
    """hot path A, whole read: realignAndScoreRead of the reference on seeded scenarios (reference window + indel table +
    reads); the fixture keeps inputs (with the error rates the reference's IndelBuffer attached) and per-read outputs"""
    pyoracle.build(ref=True)
    rng = np.random.default_rng(20240926)
    scenarios = synth.realign_scenarios(70, rng) + synth.realign_scenarios(12, rng, max_indels=14)
    expect = pyoracle.ref_realign_scenarios(scenarios)
    with open(os.path.join(HERE, "patha_realign_reference.pkl"), "wb") as f:
        pickle.dump(dict(scenarios=scenarios, expect=expect), f, protocol=4)
    n = sum(len(e) for e in expect)
    print("realign golden: %d scenarios, %d reads, %d realigned, %d indel scores" % (
        len(scenarios), n, sum(r.get("is_realigned", False) for e in expect for r in e),
        sum(len(r.get("scores", [])) for e in expect for r in e)))


def pileup_golden():
    """row a8 through the reference's own position processor (oracle/ref/ref_driver_pileup.cpp): reads -> read buffer ->
    realignment -> pileup_read_segment; the fixture keeps the reads with the alignments the reference piled up and the
    per-position columns it built"""
    pyoracle.build(ref=True)
    rng = np.random.default_rng(20240927)
    trials = []
    for t in range(8):
        reads, ref, off = synth.pileup_reads(110, rng)
        trim = (t % 3 == 0)
        kw = dict(report_begin=off + (40 if trim else 0), report_end=off + len(ref) - (55 if trim else 0))
        if t % 2:
            kw.update(min_basecall_qscore=0, mismatch_density_max_count=3, use_tier2_evidence=1, tier2_mismatch_density_max_count=10)
        if t == 6:
            kw.update(is_mapq_adjust=0, min_distance_from_read_edge=3)
        if t == 7:
            kw.update(mismatch_density_flank_size=0)
        opt = pyoracle.pileup_options(**kw)
        # externally supplied candidate indels (deletions some reads carry) make the reference realign reads around them
        cands = []
        if t in (1, 2, 4, 5):
            for r in reads:
                p = r["pos"]
                for i, (ty, ln) in enumerate(r["path"]):
                    if ty == synth.SEG["DELETE"] and 0 < i < len(r["path"]) - 1 and len(cands) < 12 and ln <= 20:
                        if p not in [c["pos"] for c in cands]:
                            cands.append(dict(pos=p, del_len=ln))
                    if ty in (synth.SEG["MATCH"], synth.SEG["DELETE"]):
                        p += ln
        finals, cols = pyoracle.ref_pileup_pipeline(reads, ref, off, opt, candidate_indels=cands)
        piled = []
        for f in finals:
            if f["skipped"]:
                continue
            r = dict(reads[f["read_id"]])
            r.update(pos=f["pos"], path=capi.cigar_to_path(f["cigar"]), is_fwd=f["is_fwd"], is_realigned=f["is_realigned"])
            piled.append(r)
        # columns as CSR over the report range
        n_loci = opt.report_end - opt.report_begin
        empty = dict(calls=np.zeros(0, np.uint16), tier2_calls=np.zeros(0, np.uint16), spandel=0, submapped=0)
        col = [cols.get(opt.report_begin + l, empty) for l in range(n_loci)]
        csr = lambda k: (np.concatenate([[0], np.cumsum([len(c[k]) for c in col])]).astype(np.int64),
                         np.concatenate([c[k] for c in col] + [np.zeros(0, np.uint16)]).astype(np.uint16))
        t1_off, t1 = csr("calls")
        t2_off, t2 = csr("tier2_calls")
        trials.append(dict(reads=piled, ref_seq=ref, ref_offset=off, opt=kw, t1_off=t1_off, t1=t1, t2_off=t2_off, t2=t2,
                           spandel=np.array([c["spandel"] for c in col], np.uint32),
                           submapped=np.array([c["submapped"] for c in col], np.uint32)))
        assert all(p in range(opt.report_begin, opt.report_end) for p in cols)
    import gzip
    with gzip.open(os.path.join(HERE, "pileup_reference.pkl.gz"), "wb") as f:
        pickle.dump(trials, f, protocol=4)
    print("pileup golden: %d trials, %d reads, %d tier1 calls, %d tier2 calls" % (
        len(trials), sum(len(t["reads"]) for t in trials), sum(len(t["t1"]) for t in trials), sum(len(t["t2"]) for t in trials)))


def _candidates_from(reads, k=12):
    cands = []
    for r in reads:
        p = r["pos"]
        for i, (ty, ln) in enumerate(r["path"]):
            if ty == synth.SEG["DELETE"] and 0 < i < len(r["path"]) - 1 and len(cands) < k and ln <= 20:
                if p not in [c["pos"] for c in cands]:
                    cands.append(dict(pos=p, del_len=ln))
            if ty in (synth.SEG["MATCH"], synth.SEG["DELETE"]):
                p += ln
    return cands


def pipeline_golden():
    """rows a1-a8 end to end through the reference's own position processor: reads -> read buffer -> realignAndScoreRead
    -> pileup_read_segment.  The fixture keeps, per trial, the reads, the IndelBuffer as the realigner saw it, what
    realignAndScoreRead was given per read (normalised input alignment, realignment range), the alignment the reference
    piled up, and the columns it built."""
    pyoracle.build(ref=True)
    rng = np.random.default_rng(20240928)
    trials = []
    for t in range(6):
        reads, ref, off = synth.pileup_reads(110, rng, read_len=(36, 120))
        # total indel reference span per read <= maxIndelSize keeps the processor's stage sizes (hence realignment ranges) fixed
        reads = [r for r in reads if sum(l for ty, l in r["path"] if ty in (synth.SEG["INSERT"], synth.SEG["DELETE"])) <= 49]
        kw = dict(report_begin=off, report_end=off + len(ref))
        opt = pyoracle.pileup_options(**kw)
        finals, cols, indels = pyoracle.ref_pileup_pipeline(reads, ref, off, opt, candidate_indels=_candidates_from(reads),
                                                            return_indels=True)
        n_loci = opt.report_end - opt.report_begin
        empty = dict(calls=np.zeros(0, np.uint16), tier2_calls=np.zeros(0, np.uint16), spandel=0, submapped=0)
        col = [cols.get(opt.report_begin + l, empty) for l in range(n_loci)]
        csr = lambda k: (np.concatenate([[0], np.cumsum([len(c[k]) for c in col])]).astype(np.int64),
                         np.concatenate([c[k] for c in col] + [np.zeros(0, np.uint16)]).astype(np.uint16))
        t1_off, t1 = csr("calls")
        t2_off, t2 = csr("tier2_calls")
        trials.append(dict(reads=reads, ref_seq=ref, ref_offset=off, opt=kw, finals=finals, indels=indels, t1_off=t1_off, t1=t1,
                           t2_off=t2_off, t2=t2, spandel=np.array([c["spandel"] for c in col], np.uint32),
                           submapped=np.array([c["submapped"] for c in col], np.uint32)))
    import gzip
    with gzip.open(os.path.join(HERE, "pipeline_reference.pkl.gz"), "wb") as f:
        pickle.dump(trials, f, protocol=4)
    print("pipeline golden: %d trials, %d reads (%d realigned), %d indels" % (
        len(trials), sum(len(t["finals"]) for t in trials), sum(f["is_realigned"] for t in trials for f in t["finals"]),
        sum(len(t["indels"]) for t in trials)))


def active_region_golden():
    """next row f2 post-processing: the reference's ActiveRegionProcessor::discoverIndelsAndMismatches (with the CIGAR of
    its own GlobalAligner) on seeded (reference segment, active region, haplotype) scenarios"""
    pyoracle.build(ref=True)
    from tests.test_active_region import reference_outputs
    scenarios = synth.active_region_scenarios(400, np.random.default_rng(4242))
    expect = reference_outputs(scenarios)
    with open(os.path.join(HERE, "active_region_reference.pkl"), "wb") as f:
        pickle.dump(dict(scenarios=scenarios, expect=expect), f, protocol=4)
    print("active-region golden: %d scenarios, %d keys, %d indels" % (
        len(scenarios), sum(len(w["keys"]) for w in expect), sum(w["n_indels"] for w in expect)))


def somatic_tiers_golden():
    """rows a12/a13 complete: the reference's own position_somatic_snv_call (both tiers, forced output, non-somatic
    quality) on the seeded scenarios of tests/test_somatic_tiers.py"""
    pyoracle.build(ref=True)
    from tests import test_somatic_tiers as T
    out = {}
    for ci, case in enumerate(T.CASES):
        out["case%d" % ci] = T.run(pyoracle.somatic_snv_call_tiers, T.scenario(1000 + ci), case, use_reference=True)
    np.savez_compressed(T.GOLDEN, **out)
    print("somatic tiers golden: %s" % ", ".join("%d calls / %d tier2-chosen / %d conflicts" % (
        (v["qphred"] > 0).sum(), (v["snv_tier"] == 1).sum(), (v["ntype"] == 3).sum()) for v in out.values()))


def somatic_indel_tiers_golden():
    """row a14 complete: the reference's own get_somatic_indel (multi-indel-allele filter, both tiers, tier combination)
    on seeded cases; the error rate the reference attached to each key is recorded with its record"""
    pyoracle.build(ref=True)
    from tests import test_somatic_tiers as T
    out = {}
    for seed in T.INDEL_SEEDS:
        cases = synth.somatic_indel_cases(500, np.random.default_rng(seed))
        rec, used = pyoracle.get_somatic_indel(cases, use_reference=True)
        out["rec%d" % seed] = rec
        out["err%d" % seed] = used
        print("somatic indel golden seed %d: %d calls, %d tier2-chosen, %d overlap, %d conflicts, %d forced" % (
            seed, (rec["qphred"] > 0).sum(), (rec["sindel_tier"] == 1).sum(), (rec["is_overlap"] == 1).sum(),
            (rec["ntype"] == 3).sum(), (rec["is_forced_output"] == 1).sum()))
    np.savez_compressed(T.INDEL_GOLDEN, **out)


def normalize_golden():
    """normalizeAlignment (L/starling_common/normalizeAlignment.cpp) of the reference itself on synth.normalize_cases"""
    import pickle
    rng = np.random.default_rng(424242)
    cases = synth.normalize_cases(700, rng)
    expect = [pyoracle.ref_normalize_alignment(c["ref_seq"], c["ref_offset"], c["read"], c["pos"], c["path"]) for c in cases]
    assert sum(e[0] for e in expect) > 400
    # (the inputs travel with the answers: the test suite can be run on shifted seeds, tools/fuzz/gpu_seeds.sh)
    slim = [dict(ref_seq=c["ref_seq"], ref_offset=c["ref_offset"], read=c["read"], pos=c["pos"], path=c["path"]) for c in cases]
    with open(os.path.join(HERE, "normalize_reference.pkl"), "wb") as f:
        pickle.dump(dict(cases=slim, expect=expect), f, protocol=4)
    print("normalize_reference.pkl:", len(cases), "alignments,", sum(e[0] for e in expect), "changed")


def feed_regions_golden():
    """the index: a small coordinate-sorted BAM with its .bai (samtools 1.6 of the reference's redist/) and, for a list of regions,
    WHICH records `samtools view BAM region` returns (their ordinals in file order) -- what sam_itr_queryi + sam_itr_next give
    the reference's bam_streamer.  (htslib merges bins whose records take less than 64 KB of the file into their parents, so a file
    this small has a coarse index; the fine-grained indexes are covered live, tests/test_bam_feed.py, where oracle/_ref exists.)"""
    import gzip
    import json
    import subprocess
    import tempfile
    samtools = os.path.join(os.path.dirname(HERE), "..", "oracle", "_ref", "bin", "samtools")
    rng = np.random.default_rng(777001)
    contigs = [("chrA", 200000), ("chrB", 50000), ("chrC", 20000), ("chrD", 140000)]  # chrC stays empty
    lines = ["@HD\tVN:1.5\tSO:unsorted"] + ["@SQ\tSN:%s\tLN:%d" % c for c in contigs]
    n = 0

    def add(contig, pos1, cigar, seq_len, flag=0):
        nonlocal n
        seq = "".join(rng.choice(list("ACGT"), seq_len))
        lines.append("\t".join(["r%d" % n, str(flag), contig, str(pos1), "40", cigar, "*", "0", "0", seq, "I" * seq_len]))
        n += 1
    for contig, length in (contigs[0], contigs[1], contigs[3]):
        for _ in range(int(length / 130)):
            add(contig, int(rng.integers(1, length - 60)), "40M", 40, flag=int(rng.choice([0, 16])))
        for _ in range(25):  # reads that reach far: deletions and skipped regions put them into the higher bins
            p = int(rng.integers(1, length - 40000))
            gap = int(rng.choice([300, 5000, 17000, 33000]))
            add(contig, p, "20M%d%s20M" % (gap, rng.choice(["D", "N"])), 40)
        for _ in range(15):  # unmapped reads placed at their mate's position: one base long for the index
            add(contig, int(rng.integers(1, length - 60)), "*", 40, flag=4)
        for k in range(8):   # reads that straddle the 16 kb / 128 kb bin boundaries, and some with clips and insertions
            b = 16384 * int(rng.integers(1, length // 16384))
            add(contig, b - int(rng.integers(1, 39)), "5S30M2I3M", 40)
        p0 = int(rng.integers(1000, length - 3000))
        for _ in range(300):  # a deep pile
            add(contig, p0 + int(rng.integers(0, 200)), "40M", 40)
    for _ in range(40):  # unplaced
        seq = "".join(rng.choice(list("ACGT"), 40))
        lines.append("\t".join(["r%d" % n, "4", "*", "0", "0", "*", "*", "0", "0", seq, "I" * 40]))
        n += 1
    with tempfile.TemporaryDirectory() as d:
        sam = os.path.join(d, "x.sam")
        with open(sam, "w") as f:
            f.write("\n".join(lines) + "\n")
        bam = os.path.join(HERE, "feed_regions.bam")
        subprocess.run([samtools, "sort", "-o", bam, sam], check=True)
        subprocess.run([samtools, "index", bam], check=True)
        names = [l.split("\t")[0] for l in subprocess.run([samtools, "view", bam], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()]
        ordinal = {q: i for i, q in enumerate(names)}
        regions = []
        for ci, (contig, length) in enumerate(contigs):
            fixed = [(0, length), (0, 1), (length - 1, length), (16383, 16385), (16384, 16384 + 1), (0, 16384), (16384, 32768),
                     (131071, 131073), (length // 2, length // 2 + 500), (length - 5000, length + 100000)]
            rnd = []
            for _ in range(30):
                b = int(rng.integers(0, length))
                rnd.append((b, b + int(rng.choice([1, 40, 300, 5000, 20000, 70000]))))
            for b, e in fixed + rnd:
                if b >= e or b >= length:
                    continue
                out = subprocess.run([samtools, "view", bam, "%s:%d-%d" % (contig, b + 1, e)], stdout=subprocess.PIPE, check=True).stdout.decode()
                regions.append([ci, b, e, [ordinal[l.split("\t")[0]] for l in out.splitlines()]])
    with gzip.open(os.path.join(HERE, "feed_regions.json.gz"), "wt") as f:
        json.dump(dict(n_records=len(names), regions=regions), f)
    print("feed_regions.bam:", len(names), "records,", os.path.getsize(bam), "bytes;", len(regions), "regions,",
          sum(len(r[3]) for r in regions), "records in them,", sum(1 for r in regions if not r[3]), "empty")


def gvcf_block_golden():
    """the gVCF writer's non-variant block logic: the reference's own gvcf_block_site_record (oracle/ref/ref_driver_gvcf_block.cpp) on
    synth.gvcf_sites; the fixture carries the sites (the suite can run on shifted seeds) and what the reference made of them"""
    pyoracle.build(ref=True)
    runs = []
    for seed, n, tol in ((1, 6000, (30, 3)), (2, 6000, (10, 1)), (3, 4000, (50, 5)), (4, 3000, (0, 0)), (5, 1, (30, 3)), (6, 2, (30, 3))):
        sites = synth.gvcf_sites(n, np.random.default_rng(880000 + seed))
        kind, blocks = pyoracle.ref_gvcf_block_sites(sites, *tol)
        runs.append(dict(sites=sites, tol=tol, kind=kind, blocks=blocks))
    np.savez_compressed(os.path.join(HERE, "gvcf_block_reference.npz"), **{"%s_%d" % (k, i): np.asarray(r[k]) for i, r in enumerate(runs) for k in r})
    print("gvcf_block_reference.npz:", sum(len(r["sites"]) for r in runs), "sites,", sum(len(r["blocks"]) for r in runs), "blocks")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "somatic_indel_tiers"):
        somatic_indel_tiers_golden()
    if what in ("all", "somatic_tiers"):
        somatic_tiers_golden()
    if what in ("all", "main"):
        main()
    if what in ("all", "realign"):
        realign_golden()
    if what in ("all", "pileup"):
        pileup_golden()
    if what in ("all", "pipeline"):
        pipeline_golden()
    if what in ("all", "active_region"):
        active_region_golden()
    if what in ("all", "normalize"):
        normalize_golden()
    if what in ("all", "feed_regions"):
        feed_regions_golden()
    if what in ("all", "gvcf_block"):
        gvcf_block_golden()
