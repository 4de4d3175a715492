"""Hot path A, whole read (realignAndScoreRead): the host stages around the GPU scoring kernel.

  * known-answer tests: the reference's own unit-test vectors for make_start_pos_alignment / get_end_pin_start_pos
    (L/starling_common/test/starling_read_align_test.cpp:67-335);
  * tests/golden/patha_realign_reference.pkl: outputs of the REFERENCE's realignAndScoreRead (oracle/_ref) on seeded
    scenarios -- realigned CIGAR/position, per-indel ReadPathScores, alternate indels, suboverlap reads.
    CPU tests feed stage 3 with scores from the test-only op interpreter (tests/flat_interp.py); the GPU test runs the
    whole job through the C-ABI (sk_realign_job_run -> the HIP scoring kernel);
  * when oracle/_ref is present the same comparison also runs live on fresh scenarios.
Everything is exact: integers/CIGARs identical, float scores bit-identical after the reference's double->float store."""
import os
import pickle

import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import capi, synth
from tests.flat_interp import score_flat

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INS10 = "AAAAACCCCC"


def _ik(pos, del_len=0, ins=""):
    return dict(pos=pos, type=capi.INDEL["INDEL"], del_len=del_len, ins_seq=ins)


FIXED = _ik(1075, 1)


# ---------------------------------------------------------------------------------------------------- reference KATs

@pytest.mark.parametrize("key,read_start,cigar,lead,trail", [
    (_ik(1050, 10), 0, "50M10D15M1D35M", None, None),            # basic delete
    (_ik(1050, 0, INS10), 0, "50M10I25M1D15M", None, None),      # basic insert
    (_ik(1050, 5, INS10), 0, "50M5D10I20M1D20M", None, None),    # basic swap
    (_ik(1091, 0, INS10), 0, "75M1D15M10I", None, "key"),        # trailing edge insert
    (_ik(1096, 0, INS10), 0, "75M1D20M5I", None, "key"),
    (_ik(1101, 0, INS10), 0, "75M1D25M", None, "none"),          # trailing edge insert miss
    (_ik(1000, 0, INS10), 0, "75M1D25M", "none", None),          # leading edge insert miss
    (_ik(1000, 0, INS10), 5, "5I75M1D20M", "key", None),         # leading edge insert
    (_ik(1000, 5, INS10), 5, "5I5D70M1D25M", "key", None),       # leading edge swap
    (_ik(1000, 10), 0, "10D65M1D35M", "key", None),              # leading edge delete
    (_ik(1101, 10), 0, "75M1D25M10D", None, "key"),              # trailing edge delete
    (_ik(1102, 10), 0, "75M1D25M", "none", None),                # trailing off-edge delete
])
def test_make_start_pos_alignment_kat(built, key, read_start, cigar, lead, trail):
    indels = [FIXED, key]
    r = capi.make_start_pos_alignment(1000, read_start, True, 100, indels)
    assert r is not None
    assert capi.path_to_cigar(r["path"]) == cigar
    assert r["pos"] == 1000
    for got, want in ((r["leading"], lead), (r["trailing"], trail)):
        if want == "key":
            assert got == 1
        elif want == "none":
            assert got == -1


@pytest.mark.parametrize("key,read_end,want", [
    (_ik(1050, 10), 100, (989, 0)),
    (_ik(1050, 0, INS10), 100, (1009, 0)),
    (_ik(1050, 5, INS10), 100, (1004, 0)),
    (_ik(1005, 0, INS10), 100, (1005, 6)),     # leading edge insert
    (_ik(999, 0, INS10), 100, (999, 0)),
    (_ik(99, 0, INS10), 100, (999, 0)),
    (_ik(1000, 5, INS10), 100, (1000, 6)),     # leading edge swap
    (_ik(1100, 0, INS10), 95, (1004, 0)),      # trailing edge insert
    (_ik(1100, 0, INS10), 100, (999, 0)),
    (_ik(1110, 0, INS10), 100, (999, 0)),
    (_ik(2000, 0, INS10), 100, (999, 0)),
    (_ik(1094, 5, INS10), 100, (1004, 0)),
    (_ik(1095, 5, INS10), 100, (994, 0)),      # trailing edge swap
    (_ik(1095, 5, INS10), 95, (999, 0)),
    (_ik(1074, 1), 100, None),                 # interfering indel: the reference throws
])
def test_get_end_pin_start_pos_kat(built, key, read_end, want):
    assert capi.get_end_pin_start_pos([FIXED, key], 100, 1100, read_end) == want


def test_retained_soft_clip_case_leaves_gate(built):
    # starling_read_align_test.cpp:338-402 (its point, isRetainOptimalSoftClipping, is an RNA-only option; on the DNA
    # path the same read realigns its soft-clip to a match).  Here: the read overlaps the candidate deletion at 4.
    job = capi.RealignJob()
    job.set_reference("ACGTACGTACGTACGTACGT", 0)
    job.set_indels([dict(_ik(4, 1), is_candidate=1, r2i=-9.9, i2r=-9.9)])
    code = [{"A": 1, "C": 2, "G": 4, "T": 8}[c] for c in "GTACGG"]
    i = job.add_read(code, [40] * 6, 2, capi.cigar_to_path("5M1S"), True, capi.MAPLEVEL["UNKNOWN"], 0, (0, 20))
    b = job.batch()
    _, lnc, lne = capi.qscore_tables()
    job.finish(score_flat(b, lnc, lne))
    r = job.result(i)
    assert r["n_cals"] >= 2 and r["is_realigned"] and r["scores"] == []  # not tier1/2 mapped -> no indel scores
    assert capi.path_to_cigar(r["path"]) == "6M" and r["pos"] == 2


# ------------------------------------------------------------------------------------------------ reference outputs

def _key_of(sc, i):
    d = sc["indels"][i]
    return (d["pos"], d["type"], d["del_len"], d["ins_seq"])


def _add_reads(job, sc):
    idx = []
    for rd in sc["reads"]:
        try:
            idx.append(job.add_read(rd["code"], rd["qual"], rd["pos"], rd["path"], rd["is_fwd"], rd["map_level"], 0,
                                    rd["realign_range"], rd["observed"]))
        except capi.StrelkaAmdError:
            idx.append(None)
    return idx


def _make_job(sc):
    job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                               min_read_bp_flank=sc["min_read_bp_flank"]))
    job.set_reference(sc["ref_seq"], sc["ref_offset"])
    job.set_indels(sc["indels"])
    return job


def _f32(x):
    return np.float32(x).view(np.uint32)


def _check_read(sc, got, want, tag):
    if want["threw"]:
        assert got is None, tag
        return
    assert got is not None, tag
    assert got["is_realigned"] == want["is_realigned"], tag
    if want["is_realigned"]:
        assert (got["pos"], capi.path_to_cigar(got["path"])) == (want["pos"], want["cigar"]), tag
    gs = sorted(got["scores"], key=lambda s: _key_of(sc, s["indel"]))
    ws = sorted(want["scores"], key=lambda s: s["key"])
    assert [_key_of(sc, s["indel"]) for s in gs] == [s["key"] for s in ws], tag
    for a, b in zip(gs, ws):
        for f in ("non_ambig", "read_length", "is_tier1_read", "is_fwd_strand", "read_pos", "edge_dist"):
            assert a[f] == b[f], (tag, f)
        assert _f32(a["ref_lnp"]) == _f32(b["ref_lnp"]) and _f32(a["indel_lnp"]) == _f32(b["indel_lnp"]), tag
        assert [(_key_of(sc, k), _f32(l)) for k, l in a["alt"]] == [(k, _f32(l)) for k, l in b["alt"]], tag
    assert sorted(_key_of(sc, i) for i in got["suboverlap"]) == sorted(want["suboverlap"]), tag


def _run_scenarios(scenarios, expect, on_gpu):
    _, lnc, lne = capi.qscore_tables()
    n_reads = n_cals = 0
    for si, (sc, exp) in enumerate(zip(scenarios, expect)):
        job = _make_job(sc)
        idx = _add_reads(job, sc)
        if on_gpu:
            job.run()
        else:
            b = job.batch()
            job.finish(score_flat(b, lnc, lne))
            n_cals += b.n_cals
        for ri, (i, want) in enumerate(zip(idx, exp)):
            _check_read(sc, None if i is None else job.result(i), want, "scenario %d read %d" % (si, ri))
            n_reads += 1
    return n_reads, n_cals


@pytest.fixture(scope="module")
def gold(built):
    with open(os.path.join(GOLD, "patha_realign_reference.pkl"), "rb") as f:
        return pickle.load(f)


def test_golden_fixture_covers_the_interesting_paths(gold):
    exp = [r for e in gold["expect"] for r in e]
    assert len(exp) > 400
    assert sum(r.get("is_realigned", False) for r in exp) > 200
    assert sum(1 for r in exp if not r.get("is_realigned", True)) > 50        # reads that leave at the gate
    assert sum("S" in r.get("cigar", "") for r in exp) > 3                     # ambiguous pools get soft-clipped edges
    assert sum(len(s["alt"]) for r in exp for s in r.get("scores", [])) > 50   # orthogonal (alternate) indels
    assert sum(len(r.get("suboverlap", [])) for r in exp) > 5
    assert sum(sc["is_haplotyping_enabled"] for sc in gold["scenarios"]) > 5


def test_host_stages_match_reference_golden(gold):
    n_reads, n_cals = _run_scenarios(gold["scenarios"], gold["expect"], on_gpu=False)
    assert n_reads > 400 and n_cals > 2000


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_host_stages_match_reference_live(built):
    rng = np.random.default_rng(77)
    scenarios = synth.realign_scenarios(40, rng) + synth.realign_scenarios(6, rng, max_indels=12)
    expect = pyoracle.ref_realign_scenarios(scenarios)
    _run_scenarios(scenarios, expect, on_gpu=False)


def test_job_is_reusable_and_rejects_bad_input(built):
    job = capi.RealignJob()
    job.set_reference("ACGT" * 20, 100)
    job.set_indels([dict(_ik(130, 2), is_candidate=1, r2i=-9.0, i2r=-9.0)])
    with pytest.raises(capi.StrelkaAmdError, match="invalid alignment path"):
        job.add_read([1] * 10, [30] * 10, 120, [(capi.SEG["MATCH"], 9)])
    with pytest.raises(capi.StrelkaAmdError, match="duplicate"):
        capi.RealignJob().set_indels([dict(_ik(5, 1)), dict(_ik(5, 1))])
    i = job.add_read([1, 2, 4, 8] * 5, [30] * 20, 120, [(capi.SEG["MATCH"], 20)])
    assert job.n_reads() == 1 and i == 0
    with pytest.raises(capi.StrelkaAmdError, match="clear the job"):
        job.set_indels([])
    job.clear_reads()
    assert job.n_reads() == 0 and job.batch().n_cals == 0
    # a read far from every candidate indel leaves at the gate: no candidate alignments, no result
    i = job.add_read([1, 2, 4, 8] * 5, [30] * 20, 150, [(capi.SEG["MATCH"], 20)])
    job.finish(np.zeros(0))
    r = job.result(i)
    assert r["n_cals"] == 0 and not r["is_realigned"] and r["scores"] == []


# ------------------------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_gpu_whole_read_job_matches_reference_golden(gpu, gold):
    n_reads, _ = _run_scenarios(gold["scenarios"], gold["expect"], on_gpu=True)
    assert n_reads > 400


def test_threaded_batch_add_equals_read_by_read(built, gold):
    """sk_realign_job_add_reads (stage 1 on host threads) and a threaded sk_realign_job_finish give exactly what the
    read-by-read, single-threaded calls give: same batch bytes, same per-read results; a rejected read adds nothing"""
    _, lnc, lne = capi.qscore_tables()
    sc = max(gold["scenarios"], key=lambda s: len(s["reads"]))
    reads = [(rd["code"], rd["qual"], rd["pos"], rd["path"], rd["is_fwd"], rd["map_level"], 0, rd["realign_range"], rd["observed"])
             for rd in sc["reads"] if not rd.get("bad")]
    ok = []
    probe = _make_job(sc)
    for r in reads:  # keep the reads the job accepts
        try:
            probe.add_read(*r)
            ok.append(r)
        except capi.StrelkaAmdError:
            pass
    many = ok * (300 // max(len(ok), 1) + 1)  # > 64 reads per thread, so that several threads really run
    def run(threads, batched):
        opt = capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"], min_read_bp_flank=sc["min_read_bp_flank"])
        opt.host_threads = threads
        job = capi.RealignJob(opt)
        job.set_reference(sc["ref_seq"], sc["ref_offset"])
        job.set_indels(sc["indels"])
        if batched:
            assert job.add_reads(many) == 0
        else:
            for r in many:
                job.add_read(*r)
        b = job.batch()
        job.finish(score_flat(b, lnc, lne))
        return b, [job.result(i) for i in range(len(many))]
    b1, r1 = run(1, False)
    b4, r4 = run(4, True)
    for f in ("read_off", "read_code", "read_qual", "hap_off", "hap_code", "cal_off", "op_off"):
        assert np.array_equal(getattr(b1, f), getattr(b4, f)), f
    assert b1.ops.tobytes() == b4.ops.tobytes()
    assert r1 == r4
    # all or nothing
    job = _make_job(sc)
    bad = list(many[:70]) + [(many[0][0], many[0][1], many[0][2], [(capi.SEG["MATCH"], 3)]) + tuple(many[0][4:])]
    with pytest.raises(capi.StrelkaAmdError, match="read 70: invalid alignment path"):
        job.add_reads(bad)
    assert job.n_reads() == 0


@pytest.mark.gpu
def test_gpu_one_job_many_reads_matches_per_scenario_jobs(gpu, gold):
    """the batched job (all reads of a scenario scored in ONE launch) gives the same bits as the CPU interpreter"""
    _, lnc, lne = capi.qscore_tables()
    for sc in gold["scenarios"][:20]:
        job = _make_job(sc)
        _add_reads(job, sc)
        b = job.batch()
        if b.n_cals == 0:
            continue
        got = capi.score_alignments(b)
        want = score_flat(b, lnc, lne)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
