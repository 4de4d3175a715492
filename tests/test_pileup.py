"""Row a8: pileup of aligned reads into per-locus basecall columns (pileup_read_segment + mismatch-density filter).

  * tests/golden/pileup_reference.pkl.gz: columns built by the REFERENCE's own position processor
    (starling_pos_processor_base, driven by oracle/ref/ref_driver_pileup.cpp: read buffer -> realignment -> pileup) for
    seeded read sets, with the alignments it piled up.  Integer/byte work: everything is compared exactly, including the
    order of the calls inside a column;
  * the C restatement (oracle/strelka_oracle.c sko_pileup_reads) is pinned to those fixtures on the CPU, and live against
    the reference when oracle/_ref is present;
  * the HIP kernels are compared with the fixtures and, on larger random batches, with the restatement (GPU tests)."""
import gzip
import os
import pickle

import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import capi, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold(built):
    with gzip.open(os.path.join(GOLD, "pileup_reference.pkl.gz"), "rb") as f:
        return pickle.load(f)


def _clean(t, include_tier2):
    """PileupCleaner::CleanPileupFilter (L/starling_common/PileupCleaner.cpp:28-66) applied to the fixture's raw columns"""
    off, out = [0], []
    for l in range(len(t["t1_off"]) - 1):
        c1 = t["t1"][t["t1_off"][l]:t["t1_off"][l + 1]]
        filt, tscf = (c1 >> 12) & 1, (c1 >> 13) & 1
        keep = (filt == 0) | ((tscf == 1) if include_tier2 else False)
        col = [c1[keep]]
        if include_tier2:
            c2 = t["t2"][t["t2_off"][l]:t["t2_off"][l + 1]]
            col.append(c2[((c2 >> 12) & 1) == 0])
        col = np.concatenate(col)
        out.append(col)
        off.append(off[-1] + len(col))
    return np.array(off, np.int64), np.concatenate(out + [np.zeros(0, np.uint16)]).astype(np.uint16)


def _expected(t, mode):
    if mode == capi.PILEUP_RAW_TIER1:
        return t["t1_off"], t["t1"]
    if mode == capi.PILEUP_RAW_TIER2:
        return t["t2_off"], t["t2"]
    return _clean(t, mode == capi.PILEUP_CLEAN_TIER2)


def _check_trial(t, fn, make_opt):
    rb = synth.ReadBatch.from_reads(t["reads"], t["ref_seq"], t["ref_offset"])
    opt = make_opt(**t["opt"])
    for mode in range(4):
        off, calls, sd, sm = fn(rb, opt, mode)
        woff, wcalls = _expected(t, mode)
        assert np.array_equal(off, woff), mode
        assert np.array_equal(calls, wcalls), mode
        assert np.array_equal(sd, t["spandel"]) and np.array_equal(sm, t["submapped"]), mode


def test_fixture_covers_the_interesting_paths(gold):
    assert sum(len(t["reads"]) for t in gold) > 800
    assert sum(len(t["t1"]) for t in gold) > 50000 and sum(len(t["t2"]) for t in gold) > 5000
    allc = np.concatenate([t["t1"] for t in gold])
    assert ((allc >> 12) & 1).sum() > 1000   # filtered calls (low quality, N, mismatch density)
    assert ((allc >> 11) & 1).sum() > 1000   # neighbour-mismatch flags
    assert ((allc >> 13) & 1).sum() > 10     # tier-specific filter (somatic settings)
    assert sum(int(t["spandel"].sum()) for t in gold) > 100 and sum(int(t["submapped"].sum()) for t in gold) > 1000
    assert sum(r.get("is_realigned", False) for t in gold for r in t["reads"]) > 5


def test_restatement_matches_reference_columns(gold):
    for t in gold:
        _check_trial(t, pyoracle.pileup_reads, pyoracle.pileup_options)


def test_mapped_qscore_table_matches_reference(built):
    g = np.load(os.path.join(GOLD, "pathb_reference.npz"), allow_pickle=True)
    tab = pyoracle.mapped_qscore_table()
    assert np.array_equal(tab[0:91:5, :], g["mapped_q"])  # fixture rows: MAPQ 0,5,...,90 from the reference's qphred_cache


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_restatement_matches_reference_live(built):
    rng = np.random.default_rng(99)
    for trial in range(4):
        reads, ref, off = synth.pileup_reads(90, rng)
        kw = dict(report_begin=off + 10 * trial, report_end=off + len(ref) - 7 * trial)
        if trial % 2:
            kw.update(min_basecall_qscore=0, mismatch_density_max_count=3, use_tier2_evidence=1)
        opt = pyoracle.pileup_options(**kw)
        finals, cols = pyoracle.ref_pileup_pipeline(reads, ref, off, opt)
        piled = []
        for f in finals:
            if f["skipped"]:
                continue
            r = dict(reads[f["read_id"]])
            r.update(pos=f["pos"], path=capi.cigar_to_path(f["cigar"]), is_fwd=f["is_fwd"])
            piled.append(r)
        rb = synth.ReadBatch.from_reads(piled, ref, off)
        for mode, key in ((0, "calls"), (1, "tier2_calls")):
            co, calls, sd, sm = pyoracle.pileup_reads(rb, opt, mode)
            for l in range(opt.report_end - opt.report_begin):
                want = cols.get(opt.report_begin + l)
                got = calls[co[l]:co[l + 1]]
                if want is None:
                    assert len(got) == 0 and sd[l] == 0 and sm[l] == 0
                else:
                    assert np.array_equal(got, want[key]) and sd[l] == want["spandel"] and sm[l] == want["submapped"]


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_evs_words_match_reference_live(built):
    """The germline EVS words (one per live match position: base id, mapq, qscore, cycle, capped edge distance, submapped) against the
    reference's position processor run with its germline scoring metrics on: per position the number of words is the MapqTracker's
    count, and the reference's own rank-sum / mean accumulators fed from the words give the six numbers its pileup holds."""
    rng = np.random.default_rng(1234)
    n_checked = n_informative = 0
    for trial in range(4):
        reads, ref, off = synth.pileup_reads(120, rng)
        kw = dict(report_begin=off + 5 * trial, report_end=off + len(ref) - 3 * trial)
        if trial % 2:
            kw.update(min_basecall_qscore=0, mismatch_density_max_count=3)
        if trial == 3:
            kw.update(is_mapq_adjust=0, min_distance_from_read_edge=3)
        opt = pyoracle.pileup_options(**kw)
        finals, cols = pyoracle.ref_pileup_pipeline(reads, ref, off, opt, germline_metrics=True)
        piled = []
        for f in finals:
            if f["skipped"]:
                continue
            r = dict(reads[f["read_id"]])
            r.update(pos=f["pos"], path=capi.cigar_to_path(f["cigar"]), is_fwd=f["is_fwd"])
            piled.append(r)
        rb = synth.ReadBatch.from_reads(piled, ref, off)
        eo, ew = pyoracle.pileup_reads_evs(rb, opt)
        for l in range(opt.report_end - opt.report_begin):
            pos = opt.report_begin + l
            words = ew[eo[l]:eo[l + 1]]
            want = cols.get(pos)
            if want is None:
                assert len(words) == 0
                continue
            assert len(words) == want["mapq_count"]
            got = pyoracle.ref_germline_metrics(words, "ACGTN".index(ref[pos - off]) if ref[pos - off] in "ACGT" else 4)
            assert np.array_equal(got, want["evs"], equal_nan=True), (pos, got, want["evs"])
            n_checked += 1
            n_informative += int(np.any(got[:3] != 0))
    assert n_checked > 800 and n_informative > 100


def test_candidate_snv_mask_and_edge_cases_restatement(built):
    # a read whose three mismatches sit within the flank: filtered; marking them candidate SNVs lifts the filter
    ref = "ACGT" * 30
    seq = list(ref[10:70])
    for i in (20, 24, 28):
        seq[i] = "A" if seq[i] != "A" else "C"
    code = np.array([{"A": 1, "C": 2, "G": 4, "T": 8}[c] for c in seq], np.uint8)
    rd = dict(code=code, qual=np.full(60, 30, np.uint8), pos=110, path=[(capi.SEG["MATCH"], 60)], is_fwd=True, mapq=60, map_level=1)
    opt = pyoracle.pileup_options(report_begin=100, report_end=220)
    rb = synth.ReadBatch.from_reads([rd], ref, 100)
    _, calls, _, _ = pyoracle.pileup_reads(rb, opt, 0)
    assert len(calls) == 60 and ((calls >> 12) & 1).sum() > 0
    mask = np.zeros(len(ref), np.uint8)
    for i in (20, 24, 28):
        mask[10 + i] = 1 << "ACGT".index(seq[i])
    rb2 = synth.ReadBatch.from_reads([rd], ref, 100, cand_snv_mask=mask)
    _, calls2, _, _ = pyoracle.pileup_reads(rb2, opt, 0)
    assert ((calls2 >> 12) & 1).sum() == 0 and ((calls2 >> 11) & 1).sum() == 0
    # no reads / empty report range
    co, c, sd, sm = pyoracle.pileup_reads(synth.ReadBatch.from_reads([], ref, 100), opt, 0)
    assert co[-1] == 0 and len(c) == 0


# ------------------------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_gpu_pileup_matches_reference_golden(gpu, gold):
    for t in gold:
        _check_trial(t, capi.pileup_reads, capi.pileup_options)


@pytest.mark.gpu
def test_gpu_pileup_matches_restatement_on_random_batches(gpu):
    rng = np.random.default_rng(2024)
    for trial in range(7):
        # trial 6: reads longer than 256 bases take the general per-read path of the kernel
        reads, ref, off = synth.pileup_reads(400 if trial < 5 else 3000 if trial == 5 else 200, rng,
                                             ref_len=900 if trial < 5 else 6000 if trial == 5 else 1500,
                                             sorted_by_pos=(trial != 3), read_len=(36, 151) if trial != 6 else (200, 420))
        kw = dict(report_begin=off + 13 * trial, report_end=off + len(ref) - 11 * trial)
        if trial % 2:
            kw.update(min_basecall_qscore=0, mismatch_density_max_count=3, use_tier2_evidence=1)
        if trial == 4:
            kw.update(mismatch_density_flank_size=0, is_mapq_adjust=0, min_distance_from_read_edge=2)
        mask = None
        if trial == 2:
            mask = rng.integers(0, 16, len(ref)).astype(np.uint8)
        rb = synth.ReadBatch.from_reads(reads, ref, off, cand_snv_mask=mask)
        for mode in range(4):
            got = capi.pileup_reads(rb, capi.pileup_options(**kw), mode)
            want = pyoracle.pileup_reads(rb, pyoracle.pileup_options(**kw), mode)
            for g, w in zip(got, want):
                assert np.array_equal(g, w), (trial, mode)


@pytest.mark.gpu
def test_gpu_pileup_edge_cases(gpu):
    ref = "ACGTTGCA" * 20
    opt = capi.pileup_options(report_begin=1000, report_end=1160)
    co, calls, sd, sm = capi.pileup_reads(synth.ReadBatch.from_reads([], ref, 1000), opt, 0)
    assert co[-1] == 0 and len(calls) == 0 and sd.sum() == 0
    # one read entirely outside the report range, one all-N read, one with an empty alignment
    n_read = dict(code=np.full(30, 15, np.uint8), qual=np.full(30, 30, np.uint8), pos=1010, path=[(capi.SEG["MATCH"], 30)],
                  is_fwd=True, mapq=60, map_level=1)
    out_read = dict(n_read, code=np.full(30, 1, np.uint8), pos=1200)
    empty = dict(code=np.full(5, 1, np.uint8), qual=np.full(5, 30, np.uint8), pos=1020, path=[], is_fwd=True, mapq=60, map_level=1)
    rb = synth.ReadBatch.from_reads([n_read, empty, out_read], ref, 1000)
    got = capi.pileup_reads(rb, opt, 0)
    want = pyoracle.pileup_reads(rb, pyoracle.pileup_options(report_begin=1000, report_end=1160), 0)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert got[0][-1] == 0  # the all-N forward read is trimmed away entirely
    with pytest.raises(capi.StrelkaAmdError, match="exceeds the maximum cached basecall quality"):
        capi.pileup_reads(synth.ReadBatch.from_reads([dict(out_read, qual=np.full(30, 71, np.uint8))], ref, 1000), opt, 0)
    with pytest.raises(capi.StrelkaAmdError, match="does not span"):
        capi.pileup_reads(synth.ReadBatch.from_reads([dict(out_read, path=[(capi.SEG["MATCH"], 29)])], ref, 1000), opt, 0)


@pytest.mark.gpu
def test_gpu_pileup_feeds_the_germline_caller(gpu):
    """a8 (cleaned columns) -> a9+a10 on the device equals the CPU chain restatement -> oracle caller"""
    rng = np.random.default_rng(8)
    reads, ref, off = synth.pileup_reads(1500, rng, ref_len=2500, indel_rate=0.1)
    kw = dict(report_begin=off, report_end=off + len(ref))
    rb = synth.ReadBatch.from_reads(reads, ref, off)
    co, calls, _, _ = capi.pileup_reads(rb, capi.pileup_options(**kw), capi.PILEUP_CLEAN_TIER1)
    wo, wc, _, _ = pyoracle.pileup_reads(rb, pyoracle.pileup_options(**kw), capi.PILEUP_CLEAN_TIER1)
    assert np.array_equal(co, wo) and np.array_equal(calls, wc)
    ref_base = np.array(["ACGT".index(c) for c in ref], np.uint8)
    pb = capi.HostPileupBatch(co, calls, ref_base)
    got, _ = capi.site_digt_call_fused(pb)
    de = pyoracle.adjust_joint_eprob(pb)
    want = pyoracle.site_digt_call(pb, de)
    assert np.array_equal(got["lhood"].view(np.uint32), want["lhood"].view(np.uint32))
    assert np.array_equal(got["phredLoghood"], want["phredLoghood"])
    assert int((np.diff(co) > 0).sum()) > 2000
