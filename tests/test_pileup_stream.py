"""Row a8 as a stream (sk_pileup_stream_*): a sample's pileup pushed one stage window at a time, chained into a9+a10.

What is pinned: whatever way the reads of a region are cut into pushes, the finalised ranges put together are the columns
the one-shot restatement (oracle sko_pileup_reads, itself pinned to the reference's own position processor in
tests/test_pileup.py) builds from all the reads at once -- raw tier1 / tier2 columns in read order, spanning-deletion and
submapped counters, the MapqTracker sums -- and the genotypes are the oracle caller's on the CleanPileupFilter'ed columns,
record for record.  The CPU tier drives the test double of the C-ABI (the adapter's `*_dbl` binaries use it), the GPU tier
the product library."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import capi, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _double():
    path = os.path.join(HERE, "..", "oracle", "libstrelka_amd_double.so")
    if not os.path.exists(path):
        pytest.skip("oracle/libstrelka_amd_double.so not built")
    L = C.CDLL(os.path.abspath(path))
    L.sk_init(0)
    return L


def _sub_batch(reads, lo, hi):
    return synth.ReadBatch.from_reads(reads[lo:hi], "", 0)


def _read_end(r):
    return r["pos"] + sum(l for t, l in r["path"] if t in (capi.SEG["MATCH"], capi.SEG["DELETE"]))


# the gVCF block options of the streams under test: the reference's defaults with a depth ceiling low enough to set HighDepth somewhere
GVCF_OPT = capi.gvcf_block_options(is_max_depth=1, max_chrom_depth=30.0, min_homref_gqx=15.0)


HALVES = {"on": False}  # the pushes as sk_pileup_stream_push_begin + _finish (set by the *_in_two_halves tests)


def _run_stream(library, reads, ref, off, kw, cuts, mask=None, genotype=True, ploidy=None, region=None, evs_words=False):
    """push reads[cuts[i]:cuts[i+1]] one after the other; final_to = the lowest start of any later read"""
    opt = capi.pileup_options(**kw)
    st = capi.PileupStream(opt, capi.germline_options() if genotype else None, library=library, evs_words=evs_words,
                           gvcf_block_opt=GVCF_OPT if genotype else None)
    rb, re = region or (kw["report_begin"], kw["report_end"])
    st.begin_region(ref, off, rb, re)
    wins = []
    for i in range(len(cuts) - 1):
        later = [r["pos"] for r in reads[cuts[i + 1]:] if r["path"]]
        final_to = min(later) if later else 2**31 - 1
        sub = _sub_batch(reads, cuts[i], cuts[i + 1])
        between = None
        if HALVES["on"]:
            def between(sub=sub):
                # with the window in flight: the caller's arrays are its own again, and the stream takes no other call
                sub.read_code[:] = 0
                sub.read_qual[:] = 0
                with pytest.raises(RuntimeError, match="has not been finished"):
                    st.begin_region(ref, off, rb, re)
                with pytest.raises(RuntimeError, match="has not been finished"):
                    st.push(sub, final_to)
        wins.append(st.push(sub, final_to, mask=mask, mask_begin=off, ploidy=ploidy, ploidy_begin=rb, halves=HALVES["on"], between=between))
    st.close()
    return wins


def _expect(reads, ref, off, kw, mask):
    rb = synth.ReadBatch.from_reads(reads, ref, off, cand_snv_mask=mask)
    o = pyoracle.pileup_options(**kw)
    o1, c1, sd, sm, mn, mz, sq = pyoracle.pileup_reads_mapq(rb, o)
    o2, c2, _, _ = pyoracle.pileup_reads(rb, o, capi.PILEUP_RAW_TIER2)
    oc, cc, _, _ = pyoracle.pileup_reads(rb, o, capi.PILEUP_CLEAN_TIER1)
    return dict(o1=o1, c1=c1, o2=o2, c2=c2, oc=oc, cc=cc, sd=sd, sm=sm, mn=mn, mz=mz, sq=sq)


def _compare(wins, want, ref, off, kw, ploidy=None, genotype=True):
    rb, re = kw["report_begin"], kw["report_end"]
    covered = np.zeros(re - rb, bool)
    prev_end = None
    for w in wins:
        b, e = w["begin"], w["end"]
        assert rb <= b <= e <= re
        if prev_end is not None:
            assert b >= prev_end
        prev_end = e
        n = e - b
        if n == 0:
            continue
        covered[b - rb:e - rb] = True
        sl = slice(b - rb, e - rb)
        for off_key, call_key, wo, wc in (("tier1_off", "tier1_calls", "o1", "c1"), ("tier2_off", "tier2_calls", "o2", "c2")):
            assert np.array_equal(np.diff(w[off_key]), np.diff(want[wo][b - rb:e - rb + 1]))
            assert np.array_equal(w[call_key], want[wc][want[wo][b - rb]:want[wo][e - rb]])
        assert np.array_equal(w["spandel"], want["sd"][sl]) and np.array_equal(w["submapped"], want["sm"][sl])
        assert np.array_equal(w["mapq_count"], want["mn"][sl]) and np.array_equal(w["mapq_zero"], want["mz"][sl])
        assert np.array_equal(w["mapq_sumsq"], want["sq"][sl])
        assert np.array_equal(w["clean_count"], np.diff(want["oc"][b - rb:e - rb + 1]))
        if genotype:
            co = want["oc"][b - rb:e - rb + 1] - want["oc"][b - rb]
            cc = want["cc"][want["oc"][b - rb]:want["oc"][e - rb]]
            ref_base = np.array(["ACGTN".index(ref[p - off]) if 0 <= p - off < len(ref) and ref[p - off] in "ACGT" else 4
                                 for p in range(b, e)], np.uint8)
            pl = None if ploidy is None else np.ascontiguousarray(ploidy[b - rb:e - rb], np.uint8)
            pb = capi.HostPileupBatch(co, cc, ref_base, ploidy=pl)
            de = pyoracle.adjust_joint_eprob(pb)
            wg = pyoracle.site_digt_call(pb, de)
            assert w["genotype"].tobytes() == wg.tobytes()
            # ... and what the gVCF writer's block logic reads of each position (site 10): plain site?, GQX, reference AD counts
            ws = pyoracle.gvcf_site_summaries(pb, wg)
            assert w["site_summary"].tobytes() == ws.tobytes()
            if w.get("gvcf_runs") is not None:
                # ... and, from every plain site, the non-variant block the writer would start there (gvcf_plain_run_kernel)
                wr = pyoracle.gvcf_plain_runs(ws, np.diff(co), np.diff(w["tier1_off"]), w["mapq_count"], GVCF_OPT)
                assert w["gvcf_runs"].tobytes() == wr.tobytes()
    # positions no window reported have nothing in them
    quiet = ~covered
    assert not np.diff(want["o1"])[quiet].any() and not np.diff(want["o2"])[quiet].any()
    assert not want["sd"][quiet].any() and not want["sm"][quiet].any() and not want["mn"][quiet].any()


def _scenarios():
    rng = np.random.default_rng(77)
    out = []
    for trial in range(6):
        n = (300, 900, 60, 500, 1200, 5)[trial]
        reads, ref, off = synth.pileup_reads(n, rng, ref_len=(900, 2500, 400, 1500, 3000, 300)[trial],
                                             read_len=(36, 151) if trial != 4 else (120, 300))
        kw = dict(report_begin=off + 7 * trial, report_end=off + len(ref) - 5 * trial)
        if trial % 2:
            kw.update(min_basecall_qscore=0, mismatch_density_max_count=3, use_tier2_evidence=1)
        mask = rng.integers(0, 16, len(ref)).astype(np.uint8) if trial in (1, 3) else None
        k = len(reads)
        if trial == 0:
            cuts = [0, k]                                   # everything in one push
        elif trial == 2:
            cuts = list(range(0, k + 1))                    # one read per push
        else:
            cuts = sorted(set([0, k] + [int(x) for x in rng.integers(0, k, (3, 12, 0, 25, 40, 2)[trial])]))
        if trial == 5:
            cuts = [0, 0, k, k]                             # empty pushes before and after
        ploidy = None
        if trial == 3:
            ploidy = np.where(rng.random(kw["report_end"] - kw["report_begin"]) < 0.3, 1, 2).astype(np.uint8)
        out.append((reads, ref, off, kw, cuts, mask, ploidy))
    return out


def _run_all(library, evs_words=False):
    n_windows = n_loci = n_words = 0
    for reads, ref, off, kw, cuts, mask, ploidy in _scenarios():
        want = _expect(reads, ref, off, kw, mask)
        wins = _run_stream(library, reads, ref, off, kw, cuts, mask=mask, ploidy=ploidy, evs_words=evs_words)
        _compare(wins, want, ref, off, kw, ploidy=ploidy)
        if evs_words:
            # the germline EVS words: the one-shot restatement's, window by window; a position has mapq_count of them
            rb = synth.ReadBatch.from_reads(reads, ref, off, cand_snv_mask=mask)
            eo, ew = pyoracle.pileup_reads_evs(rb, pyoracle.pileup_options(**kw))
            assert np.array_equal(np.diff(eo), want["mn"])
            b0 = kw["report_begin"]
            for w in wins:
                b, e = w["begin"], w["end"]
                assert np.array_equal(np.diff(w["evs_off"]), np.diff(eo[b - b0:e - b0 + 1])) and int(w["evs_off"][0]) == 0 if e > b else True
                assert np.array_equal(w["evs_words"], ew[eo[b - b0]:eo[e - b0]])
                n_words += len(w["evs_words"])
        n_windows += len(wins)
        n_loci += sum(w["end"] - w["begin"] for w in wins)
    assert n_windows > 60 and n_loci > 5000 and (n_words > 100000 or not evs_words)


def test_double_stream_equals_one_shot_restatement(built):
    _run_all(_double())


def test_double_stream_with_evs_words(built):
    _run_all(_double(), evs_words=True)


def test_double_stream_in_two_halves(built):
    """sk_pileup_stream_push_begin + _finish: the same windows; between the two the reads' arrays are the caller's again and the stream
    refuses other calls"""
    HALVES["on"] = True
    try:
        _run_all(_double(), evs_words=True)
    finally:
        HALVES["on"] = False


def test_double_stream_rejects_a_read_behind_the_final_position(built):
    L = _double()
    ref = "ACGTTGCA" * 40
    mk = lambda pos: dict(code=np.full(30, 1, np.uint8), qual=np.full(30, 30, np.uint8), pos=pos, path=[(capi.SEG["MATCH"], 30)],
                          is_fwd=True, mapq=60, map_level=1)
    st = capi.PileupStream(capi.pileup_options(), None, library=L)
    st.begin_region(ref, 1000, 1000, 1320)
    st.push(synth.ReadBatch.from_reads([mk(1010)], "", 0), 1100)
    with pytest.raises(RuntimeError, match="declared final"):
        st.push(synth.ReadBatch.from_reads([mk(1090)], "", 0), 1200)
    st.close()


@pytest.mark.gpu
def test_gpu_stream_equals_one_shot_restatement(gpu):
    _run_all(None)


@pytest.mark.gpu
def test_gpu_stream_with_evs_words(gpu):
    _run_all(None, evs_words=True)


@pytest.mark.gpu
def test_gpu_stream_in_two_halves(gpu):
    HALVES["on"] = True
    try:
        _run_all(None, evs_words=True)
    finally:
        HALVES["on"] = False


@pytest.mark.gpu
def test_gpu_stream_rejects_a_read_behind_the_final_position(gpu):
    ref = "ACGTTGCA" * 40
    mk = lambda pos: dict(code=np.full(30, 1, np.uint8), qual=np.full(30, 30, np.uint8), pos=pos, path=[(capi.SEG["MATCH"], 30)],
                          is_fwd=True, mapq=60, map_level=1)
    st = capi.PileupStream(capi.pileup_options(), None)
    st.begin_region(ref, 1000, 1000, 1320)
    w = st.push(synth.ReadBatch.from_reads([mk(1010)], "", 0), 1100)
    assert (w["begin"], w["end"]) == (1010, 1040) and int(w["tier1_off"][-1]) == 30
    with pytest.raises(RuntimeError, match="declared final"):
        st.push(synth.ReadBatch.from_reads([mk(1090)], "", 0), 1200)
    st.close()


@pytest.mark.gpu
def test_gpu_stream_at_window_scale(gpu):
    """a 40x sample pushed in windows of ~4000 reads (what the adapter sends): equal to the one-shot kernels' columns"""
    rng = np.random.default_rng(5)
    rb, n_loci = synth.pileup_reads_flat(40000, rng)
    kw = dict(report_begin=0, report_end=n_loci + 200)
    o1, c1, sd, sm = capi.pileup_reads(rb, capi.pileup_options(**kw), capi.PILEUP_RAW_TIER1)
    st = capi.PileupStream(capi.pileup_options(**kw), capi.germline_options())
    st.begin_region(rb.ref_seq, 0, 0, n_loci + 200)
    got_off, got_calls = [0], []
    L = 150
    for lo in range(0, rb.n_reads, 4000):
        hi = min(rb.n_reads, lo + 4000)
        sub = synth.ReadBatch(rb.read_off[lo:hi + 1] - rb.read_off[lo], rb.read_code[rb.read_off[lo]:rb.read_off[hi]],
                              rb.read_qual[rb.read_off[lo]:rb.read_off[hi]], rb.path_off[lo:hi + 1] - rb.path_off[lo],
                              rb.path[rb.path_off[lo]:rb.path_off[hi]], rb.pos[lo:hi], rb.is_fwd[lo:hi], rb.mapq[lo:hi],
                              rb.map_level[lo:hi], "", 0)
        final_to = int(rb.pos[hi]) if hi < rb.n_reads else 2**31 - 1
        w = st.push(sub, final_to)
        if w["end"] > w["begin"]:
            assert w["begin"] == len(got_off) - 1 or not np.diff(o1[len(got_off) - 1:w["begin"] + 1]).any()
            while len(got_off) - 1 < w["begin"]:
                got_off.append(got_off[-1])
            got_off += list(got_off[-1] + w["tier1_off"][1:])
            got_calls.append(w["tier1_calls"])
            assert np.array_equal(w["spandel"], sd[w["begin"]:w["end"]])
    got_calls = np.concatenate(got_calls)
    assert np.array_equal(np.array(got_off), o1[:len(got_off)]) and np.array_equal(got_calls, c1[:len(got_calls)])
    assert len(got_calls) == len(c1)
    st.close()


# ---- the two samples of a somatic run through one stream (sk_somatic_pileup_stream_*), chained into a12+a13 --------------------

def _expect_somatic_sample(reads, ref, off, kw, mask):
    want = _expect(reads, ref, off, kw, mask)
    rb = synth.ReadBatch.from_reads(reads, ref, off, cand_snv_mask=mask)
    o = pyoracle.pileup_options(**kw)
    want["o4"], want["c4"], _, _ = pyoracle.pileup_reads(rb, o, capi.PILEUP_CLEAN_TIER2)
    orp, crp, rp = pyoracle.pileup_reads_readpos(rb, o)
    assert np.array_equal(orp, want["o1"]) and np.array_equal(crp, want["c1"])
    want["rp"] = rp
    return want


def _somatic_scenarios():
    rng = np.random.default_rng(1234)
    out = []
    for trial in range(4):
        ref_len = (700, 1800, 300, 1200)[trial]
        off = 5000
        ref = synth._random_ref(ref_len, rng)
        samples = []
        for depth_reads in ((120, 300), (400, 900), (20, 70), (250, 0))[trial]:
            reads, _, _ = synth.pileup_reads(depth_reads, rng, ref_len=ref_len, ref_offset=off, read_len=(36, 151)) if depth_reads else ([], None, None)
            # pileup_reads draws its own reference: re-draw the bases of the reads against the shared one, keeping geometry
            for r in reads:
                p, rp = r["pos"], 0
                code = r["code"].copy()
                for t, l in r["path"]:
                    if t == capi.SEG["MATCH"]:
                        for j in range(l):
                            i = p + j - off
                            if 0 <= i < ref_len and rng.random() > 0.04 and code[rp + j] != 15:
                                code[rp + j] = 1 << "ACGT".index(ref[i])
                        p += l
                        rp += l
                    elif t == capi.SEG["DELETE"]:
                        p += l
                    elif t in (capi.SEG["INSERT"], capi.SEG["SOFT_CLIP"]):
                        rp += l
                r["code"] = code
            samples.append(reads)
        kw = dict(report_begin=off + 5 * trial, report_end=off + ref_len - 3 * trial, min_basecall_qscore=0, mismatch_density_max_count=3,
                  use_tier2_evidence=1)
        if trial == 2:
            kw["use_tier2_evidence"] = 0
        mask = rng.integers(0, 16, ref_len).astype(np.uint8) if trial == 1 else None
        n_cuts = (0, 9, 30, 4)[trial]
        cuts = sorted(set(int(x) for x in rng.integers(off - 10, off + ref_len, n_cuts)))
        forced = (rng.random(ref_len) < 0.05).astype(np.uint8) if trial in (1, 3) else None
        out.append((samples, ref, off, kw, cuts, mask, forced, trial % 2 == 1))
    return out


def _run_somatic(library, genotype=True):
    n_windows = n_loci = n_computed = 0
    for samples, ref, off, kw, cuts, mask, forced, nonsom in _somatic_scenarios():
        want = [_expect_somatic_sample(reads, ref, off, kw, mask) for reads in samples]
        opt = capi.pileup_options(**kw)
        sopt = capi.somatic_snv_options()
        st = capi.SomaticPileupStream(opt, sopt if genotype else None, with_read_pos=True, library=library)
        st.begin_region(ref, off, kw["report_begin"], kw["report_end"])
        bounds = [-(2**31)] + cuts + [2**31 - 1]
        rb, re = kw["report_begin"], kw["report_end"]
        covered = np.zeros(re - rb, bool)
        for i in range(len(bounds) - 1):
            lo, hi = bounds[i], bounds[i + 1]
            subs, later = [], []
            for reads in samples:
                subs.append(synth.ReadBatch.from_reads([r for r in reads if lo <= r["pos"] < hi], "", 0))
                later += [r["pos"] for r in reads if r["pos"] >= hi and r["path"]]
            final_to = min(later) if later else 2**31 - 1
            w = st.push(subs[0], subs[1], final_to, mask=mask, mask_begin=off, forced=forced, forced_begin=off, is_compute_nonsomatic=nonsom)
            b, e = w["begin"], w["end"]
            assert rb <= b <= e <= re and w["normal"]["begin"] == b and w["tumor"]["end"] == e
            n = e - b
            n_windows += 1
            if n == 0:
                continue
            assert not covered[b - rb:e - rb].any()
            covered[b - rb:e - rb] = True
            n_loci += n
            cols = []
            for si, name in enumerate(("normal", "tumor")):
                ws, wa = w[name], want[si]
                for off_key, call_key, wo, wc in (("tier1_off", "tier1_calls", "o1", "c1"), ("tier2_off", "tier2_calls", "o2", "c2")):
                    assert np.array_equal(np.diff(ws[off_key]), np.diff(wa[wo][b - rb:e - rb + 1]))
                    assert np.array_equal(ws[call_key], wa[wc][wa[wo][b - rb]:wa[wo][e - rb]])
                sl = slice(b - rb, e - rb)
                assert np.array_equal(ws["spandel"], wa["sd"][sl]) and np.array_equal(ws["submapped"], wa["sm"][sl])
                assert np.array_equal(ws["mapq_count"], wa["mn"][sl]) and np.array_equal(ws["mapq_sumsq"], wa["sq"][sl])
                assert np.array_equal(ws["clean_count"], np.diff(wa["oc"][b - rb:e - rb + 1]))
                assert np.array_equal(ws["clean2_count"], np.diff(wa["o4"][b - rb:e - rb + 1]))
                for o_key, c_key in (("oc", "cc"), ("o4", "c4")):
                    cols.append((wa[o_key][b - rb:e - rb + 1] - wa[o_key][b - rb], wa[c_key][wa[o_key][b - rb]:wa[o_key][e - rb]]))
            assert np.array_equal(w["read_pos"], want[1]["rp"][want[1]["o1"][b - rb]:want[1]["o1"][e - rb]])
            if genotype:
                ref_base = np.array(["ACGTN".index(ref[p - off]) if 0 <= p - off < len(ref) and ref[p - off] in "ACGT" else 4
                                     for p in range(b, e)], np.uint8)
                n1, n2, t1, t2 = (capi.HostPileupBatch(co, cc, ref_base) for co, cc in cols)
                fo = None if forced is None else np.ascontiguousarray(forced[b - off:e - off])
                tier2 = bool(kw["use_tier2_evidence"])
                wg = pyoracle.somatic_snv_call_tiers(n1, t1, n2 if tier2 else None, t2 if tier2 else None, sopt, is_forced_output=fo,
                                                     is_compute_nonsomatic=nonsom)
                assert w["genotype"].tobytes() == wg.tobytes()
                n_computed += int(w["genotype"]["is_computed"].sum())
            else:
                assert w["genotype"] is None
        st.close()
        quiet = ~covered
        for wa in want:
            assert not np.diff(wa["o1"])[quiet].any() and not np.diff(wa["o2"])[quiet].any() and not wa["mn"][quiet].any()
    assert n_windows > 40 and n_loci > 3000 and (n_computed > 1000 or not genotype)


def test_double_somatic_stream_equals_one_shot_restatement(built):
    _run_somatic(_double())


def test_double_somatic_stream_columns_only(built):
    _run_somatic(_double(), genotype=False)


@pytest.mark.gpu
def test_gpu_somatic_stream_equals_one_shot_restatement(gpu):
    _run_somatic(None)


@pytest.mark.gpu
def test_gpu_somatic_stream_columns_only(gpu):
    _run_somatic(None, genotype=False)
