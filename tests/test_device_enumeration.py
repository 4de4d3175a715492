"""candidate_alignment_search (L/starling_common/starling_read_align.cpp:859-1277) as the container-free
core of strelka_amd/csrc/realign_core.h -- on the host (sk_realign_options.enumeration = 1) and on the device (= 2).

  * CPU: the core is run in place of the container-based search on seeded realignment scenarios and must reproduce the
    REFERENCE's realignAndScoreRead results (golden fixture + live reference where oracle/_ref is present) -- the
    std::set<CandidateAlignment> order included (the batch order feeds the tie rules of the selection);
  * the core must actually have run (enumeration counters), not the fallback;
  * GPU: enumeration = 2 gives the same per-read results as enumeration = 0, on the golden scenarios and on dense ones;
  * stage 3 (selection, clipping, score_indels) in its container-free form (csrc/stage3_core.h) is covered by the same runs:
    on the host with enumeration = 1, in stage3_kernel with enumeration = 2 (sk_realign_job_stage3_counts says where it ran).
"""
import os
import pickle

import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import capi, synth
from tests.flat_interp import score_flat
from tests import test_read_realign as T


def _run(scenarios, expect, mode, on_gpu):
    _, lnc, lne = capi.qscore_tables()
    core = dev = fb = reads = 0
    _run.stage3 = [0, 0]
    for si, (sc, exp) in enumerate(zip(scenarios, expect)):
        job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                   min_read_bp_flank=sc["min_read_bp_flank"], enumeration=mode))
        job.set_reference(sc["ref_seq"], sc["ref_offset"])
        job.set_indels(sc["indels"])
        idx = T._add_reads(job, sc)
        if on_gpu:
            job.run()
        else:
            b = job.batch()
            job.finish(score_flat(b, lnc, lne))
        for ri, (i, want) in enumerate(zip(idx, exp)):
            T._check_read(sc, None if i is None else job.result(i), want, "scenario %d read %d" % (si, ri))
            reads += 1
        a, b_, c = job.enumeration_counts()
        core, dev, fb = core + a, dev + b_, fb + c
        s3 = job.stage3_counts()
        _run.stage3 = [_run.stage3[0] + s3[0], _run.stage3[1] + s3[1]]
    return reads, core, dev, fb


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(T.GOLD, "patha_realign_reference.pkl"), "rb") as f:
        return pickle.load(f)


def test_host_core_reproduces_the_reference_golden(gold):
    reads, core, dev, fb = _run(gold["scenarios"], gold["expect"], mode=1, on_gpu=False)
    assert reads > 400 and core > 200 and dev == 0
    assert fb <= core // 50  # the fixed capacities are rarely exceeded
    assert _run.stage3[0] >= core * 0.98  # stage 3 ran in its container-free form as well (csrc/stage3_core.h)


def test_host_core_without_the_conflict_tables(gold, monkeypatch):
    """score_indels' per-read tables cover the span of table indices a read's alignments hold (<= 256); beyond that the same is
    computed from the table entries -- forced here for every read"""
    monkeypatch.setenv("SK_STAGE3_NO_TABLES", "1")
    reads, core, dev, fb = _run(gold["scenarios"], gold["expect"], mode=1, on_gpu=False)
    assert reads > 400 and _run.stage3[0] >= core * 0.98


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed,max_indels,hap", [(31, 6, 0.25), (32, 12, 0.6), (33, 9, 0.0)])
def test_host_core_reproduces_the_live_reference(seed, max_indels, hap):
    rng = np.random.default_rng(90000 + seed)
    scs = synth.realign_scenarios(60, rng, reads_per=10, max_indels=max_indels, haplotyping_rate=hap)
    exp = pyoracle.ref_realign_scenarios(scs)
    reads, core, dev, fb = _run(scs, exp, mode=1, on_gpu=False)
    assert reads >= 500 and core > 100
    assert _run.stage3[0] >= core * 0.98


@pytest.mark.gpu
@pytest.mark.parametrize("chain", ["fused", "fused_three_waits", "staged", "fused_too_few_levels"])
def test_device_enumeration_reproduces_the_reference_golden(gold, monkeypatch, chain):
    """chain: flattening + scoring as F5 (flatten_score_kernel: the records -> the scores in one launch, the default) or as the staged
    chain F1-F3 + A1c ($SK_A5_FUSED=0; also what a job with a read outside F5's form takes).  With F5 a job is ONE fixed sequence of
    launches and one host wait (the default; job_scan_kernel makes the prefix sums between the stages on the device) or the sequence with
    a wait after the search, the sets and stage 3 ($SK_ENUM_ONE_WAIT=0); with too few search levels launched the device reports it and
    the job runs again the other way -- the same records from all of them."""
    capi.init(0)
    monkeypatch.setenv("SK_A5_FUSED", "0" if chain == "staged" else "1")
    monkeypatch.setenv("SK_ENUM_ONE_WAIT", "0" if chain == "fused_three_waits" else "1")
    if chain == "fused_too_few_levels":
        monkeypatch.setenv("SK_ENUM_TEST_LEVELS", "2")
    before = capi.RealignJob.device_job_counts()
    reads, core, dev, fb = _run(gold["scenarios"], gold["expect"], mode=2, on_gpu=True)
    one_wait, redone, staged = (b - a for a, b in zip(before, capi.RealignJob.device_job_counts()))
    assert reads > 400 and dev > 200 and core == 0
    assert fb <= dev // 50
    assert _run.stage3[1] >= dev * 0.95  # and stage 3 of those reads ran on the device as well (stage3_kernel)
    if chain == "fused":
        assert one_wait > 0 and redone <= one_wait // 10 and staged == redone
    elif chain == "fused_too_few_levels":
        assert redone > 0 and staged == redone
    else:
        assert one_wait == 0 and redone == 0 and staged > 0


@pytest.mark.gpu
@pytest.mark.parametrize("grid,waves", [("1", "1"), ("3", "16"), ("7", "4"), ("100000", "2")])
def test_flatten_score_read_queue_under_any_grid(grid, waves):
    """F5's waves take their reads from a queue (one counter; the last wave out puts it back to zero): the same records from a grid of ONE
    wave (every read in turn), from grids that do not divide the job, and from one with far more blocks than reads -- $SK_F5_GRID /
    $SK_F5_WAVES are read once per process, so each shape runs the golden and the long-read tests in a process of its own"""
    import subprocess
    import sys
    env = dict(os.environ, SK_F5_GRID=grid, SK_F5_WAVES=waves)
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_device_enumeration.py"), "-m", "gpu", "-x", "-q", "-k",
                        "reproduces_the_reference_golden and fused or long_reads_and_mixed_jobs", "-p", "no:cacheprovider"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = p.stdout.decode(errors="replace")[-1500:]
    assert p.returncode == 0 and " passed" in tail, tail


@pytest.mark.gpu
@pytest.mark.parametrize("one_wait", ["1", "0"])
@pytest.mark.parametrize("caps", ["4,12", "16,64"])
def test_device_stage3_launch_shapes(gold, monkeypatch, caps, one_wait):
    """stage3_kernel keeps a read's per-alignment arrays in LDS when they fit (two launches: few / many candidate alignments) and
    in HBM otherwise; with tiny capacities the golden scenarios go through all three -- the lists of the two launches made by the host
    (three waits) and by the device (one sequence)"""
    capi.init(0)
    monkeypatch.setenv("SK_ENUM_ONE_WAIT", one_wait)
    monkeypatch.setenv("SK_STAGE3_TEST_LDS_CALS", caps)
    reads, core, dev, fb = _run(gold["scenarios"], gold["expect"], mode=2, on_gpu=True)
    assert reads > 400 and _run.stage3[1] >= dev * 0.95


@pytest.mark.gpu
def test_device_stage3_without_the_conflict_tables(gold, monkeypatch):
    capi.init(0)
    monkeypatch.setenv("SK_STAGE3_NO_TABLES", "1")
    reads, core, dev, fb = _run(gold["scenarios"], gold["expect"], mode=2, on_gpu=True)
    assert reads > 400 and _run.stage3[1] >= dev * 0.95


@pytest.mark.gpu
@pytest.mark.parametrize("seed,max_indels,hap,chain", [(41, 6, 0.25, "fused"), (42, 12, 0.6, "fused"), (43, 12, 0.0, "fused"), (42, 12, 0.6, "staged"),
                                                       (44, 12, 0.4, "fused_three_waits")])
def test_device_enumeration_equals_host(seed, max_indels, hap, chain, monkeypatch):
    capi.init(0)
    monkeypatch.setenv("SK_A5_FUSED", "0" if chain == "staged" else "1")
    monkeypatch.setenv("SK_ENUM_ONE_WAIT", "0" if chain == "fused_three_waits" else "1")
    rng = np.random.default_rng(91000 + seed)
    scs = synth.realign_scenarios(80, rng, reads_per=12, max_indels=max_indels, haplotyping_rate=hap)
    n_dev = n_s3 = 0
    for sc in scs:
        res, cons = {}, {}
        for mode in (0, 2):
            job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                       min_read_bp_flank=sc["min_read_bp_flank"], enumeration=mode))
            job.set_reference(sc["ref_seq"], sc["ref_offset"])
            job.set_indels(sc["indels"])
            idx = T._add_reads(job, sc)
            job.run()
            res[mode] = [None if i is None else job.result(i) for i in idx]
            cons[mode] = job.indels_consulted()
            if mode == 2:
                n_dev += job.enumeration_counts()[1]
                n_s3 += job.stage3_counts()[1]
        for a, b in zip(res[0], res[2]):
            assert (a is None) == (b is None)
            if a is None:
                continue
            assert repr(a) == repr(b)
        # the candidate status of exactly the same indels was looked up (the adapter commits its cache from this)
        assert np.array_equal(cons[0], cons[2])
    assert n_dev > 200 and n_s3 >= n_dev * 0.9  # the whole read path of these reads ran on the device


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["thread", "wave"])
def test_device_flatten_kernels_on_long_reads(kernel, monkeypatch):
    """F3 in both forms ($SK_F3_KERNEL: a thread per candidate alignment / a wave per read with the base comparisons made once per
    haplotype offset) on reads of 120-260 bases over 4-9 indels: more than 64 candidate alignments per read (several passes of the
    wave's tile), reads past 256 bases (the wave form hands them to the serial walk); results equal to the host path's"""
    capi.init(0)
    monkeypatch.setenv("SK_F3_KERNEL", kernel)
    monkeypatch.setenv("SK_A5_FUSED", "0")  # (F3 belongs to the staged chain)
    rng = np.random.default_rng(91333)
    scs = synth.realign_scenarios(40, rng, reads_per=10, max_indels=9, min_indels=4, read_len=(120, 261), window=(330, 520), haplotyping_rate=0.2)
    n_dev = n_cals = n_long = 0
    for sc in scs:
        res = {}
        for mode in (0, 2):
            job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                       min_read_bp_flank=sc["min_read_bp_flank"], enumeration=mode))
            job.set_reference(sc["ref_seq"], sc["ref_offset"])
            job.set_indels(sc["indels"])
            idx = T._add_reads(job, sc)
            job.run()
            res[mode] = [None if i is None else job.result(i) for i in idx]
            if mode == 2:
                n_dev += job.enumeration_counts()[1]
        for rd, a, b in zip(sc["reads"], res[0], res[2]):
            assert (a is None) == (b is None)
            if a is not None:
                assert repr(a) == repr(b)
                n_cals += a["n_cals"]
                n_long += len(rd["code"]) > 256
    assert n_dev > 150 and n_cals > 64 * n_dev // 4


@pytest.mark.gpu
def test_fused_flatten_score_on_long_reads_and_mixed_jobs():
    """F5 on reads of 120-260 bases over 4-9 indels (more than 64 candidate alignments per read: several rounds of the wave); a job
    holding a read past 256 bases takes the staged chain whole -- either way the host path's results"""
    capi.init(0)
    rng = np.random.default_rng(91444)
    scs = synth.realign_scenarios(40, rng, reads_per=10, max_indels=9, min_indels=4, read_len=(120, 261), window=(330, 520), haplotyping_rate=0.2)
    n_dev = n_cals = n_long_jobs = n_jobs_on_device = 0
    before = capi.RealignJob.device_job_counts()
    for sc in scs:
        res = {}
        for mode in (0, 2):
            job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                       min_read_bp_flank=sc["min_read_bp_flank"], enumeration=mode))
            job.set_reference(sc["ref_seq"], sc["ref_offset"])
            job.set_indels(sc["indels"])
            idx = T._add_reads(job, sc)
            job.run()
            res[mode] = [None if i is None else job.result(i) for i in idx]
            if mode == 2:
                counts = job.enumeration_counts()
                n_dev += counts[1]
                n_jobs_on_device += (counts[1] + counts[2]) > 0  # (a job all of whose reads leave at the gate never reaches the device)
        n_long_jobs += any(len(rd["code"]) > 256 for rd in sc["reads"])
        for a, b in zip(res[0], res[2]):
            assert (a is None) == (b is None)
            if a is not None:
                assert repr(a) == repr(b)
                n_cals += a["n_cals"]
    assert n_dev > 150 and n_cals > 64 * n_dev // 4 and 0 < n_long_jobs < len(scs)
    # the jobs without a long read among those that pass the gate ran as one sequence; those with one went the staged way at once (the host
    # knows the read lengths), and a pool over F5's 768 bytes is reported by the device: that job runs again (redone)
    one_wait, redone, staged = (b - a for a, b in zip(before, capi.RealignJob.device_job_counts()))
    # (one_wait counts the jobs that COMPLETED as one sequence: a job that is run again is counted as redone and as staged, not as one_wait)
    assert one_wait > 0 and 0 < staged <= n_long_jobs + redone and one_wait + staged == n_jobs_on_device
    assert n_jobs_on_device >= len(scs) - 2


@pytest.mark.gpu
def test_device_capacity_overflow_falls_back_to_the_host(monkeypatch):
    """with capacities far too small for the job the device turns reads down; the host code enumerates those and nothing changes"""
    capi.init(0)
    rng = np.random.default_rng(91077)
    scs = synth.realign_scenarios(30, rng, reads_per=12, max_indels=12, haplotyping_rate=0.3)
    fallbacks = 0
    for sc in scs:
        res = {}
        for mode, caps in ((0, None), (2, "1,8")):
            if caps:
                monkeypatch.setenv("SK_ENUM_TEST_CAPS", caps)
            else:
                monkeypatch.delenv("SK_ENUM_TEST_CAPS", raising=False)
            job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                       min_read_bp_flank=sc["min_read_bp_flank"], enumeration=mode))
            job.set_reference(sc["ref_seq"], sc["ref_offset"])
            job.set_indels(sc["indels"])
            idx = T._add_reads(job, sc)
            job.run()
            res[mode] = [None if i is None else job.result(i) for i in idx]
            if mode == 2:
                fallbacks += job.enumeration_counts()[2]
        assert [repr(x) for x in res[0]] == [repr(x) for x in res[2]]
    assert fallbacks > 20


@pytest.mark.gpu
def test_device_batch_scores_equal_host_batch_scores():
    """get_batch after a device run rebuilds the host batch from the device's candidate alignments: same alignments, same order,
    and the device's scores are the host batch's scores bit for bit.  The candidate alignments of reads finished on the device
    stay there: get_batch right after the run fetches them, get_batch after ANOTHER job has run lists them again on the host, and
    get_batch without a run has the device list them without scoring."""
    capi.init(0)
    rng = np.random.default_rng(91088)
    later = []
    for si, sc in enumerate(synth.realign_scenarios(24, rng, reads_per=12, max_indels=10)):
        def job_of(mode):
            job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                       min_read_bp_flank=sc["min_read_bp_flank"], enumeration=mode))
            job.set_reference(sc["ref_seq"], sc["ref_offset"])
            job.set_indels(sc["indels"])
            T._add_reads(job, sc)
            return job
        host = job_of(0)
        host.run()
        b = host.batch()
        want = (np.array(b.cal_off), capi.score_alignments(b))
        dev = job_of(2)
        if si % 3 == 2:
            later.append((dev, want))  # no run at all: enumeration only
            continue
        dev.run()
        if si % 3 == 1:
            later.append((dev, want))  # batch asked for after other jobs have used the device
            continue
        b = dev.batch()
        assert np.array_equal(want[0], np.array(b.cal_off))
        assert np.array_equal(want[1].view(np.uint64), capi.score_alignments(b).view(np.uint64))
    assert len(later) >= 12
    for dev, want in later:
        b = dev.batch()
        assert np.array_equal(want[0], np.array(b.cal_off))
        assert np.array_equal(want[1].view(np.uint64), capi.score_alignments(b).view(np.uint64))
