"""The reference's OWN translation units (oracle/_ref/libstrelka_ref.so, built from /root/reference by oracle/Makefile) timed on
the host cores: bench.py's `cpu_baseline` with kind "reference" (SURVEY.md 8d).

TEST / MEASUREMENT INFRASTRUCTURE.  The reference is single-threaded and its caches are not thread-safe, so -- as its own
workflow does (one process per genome segment) -- P independent OS processes are started, each pinned to one core, each
with its own synthetic sample of the workload materialised in the reference's own structs; only the reference's compute calls
are inside the clock:

  A        scoreCandidateAlignment                         150 bp reads x 64 candidate alignments      cells/s
  reads    realignAndScoreRead (whole read: a1-a7)         synthetic realignment scenarios             reads/s
  loci     adjust_joint_eprob + position_snp_call_pprob_digt   depth ~Poisson(40)                     loci/s
  somatic  position_somatic_snv_call                       normal 40x + tumor 110x                     loci/s

Per-core rate = mean over the processes; P-core rate = sum.
"""
import ctypes as C
import multiprocessing as mp
import os
import time

import numpy as np

vp = C.c_void_p


def _bind(L):
    from oracle import pyoracle
    L.ref_session_create.restype = vp
    L.ref_session_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.ref_session_destroy.argtypes = [vp]
    L.ref_session_add_indel.argtypes = [vp, C.POINTER(pyoracle.RefIndel)]
    L.ref_session_add_indel_observed.argtypes = [vp, C.POINTER(pyoracle.RefIndel), C.c_uint]
    L.ref_session_set_indel_haplotype.argtypes = [vp, C.POINTER(pyoracle.RefIndel)] + [C.c_int] * 5
    L.ref_session_prepare_score_read.argtypes = [vp, C.c_char_p, vp, C.c_int, vp, C.c_int]
    L.ref_session_time_score.restype = C.c_double
    L.ref_session_time_score.argtypes = [vp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ref_session_prepare_realign_read.argtypes = [vp, C.c_char_p, vp, C.c_int, C.c_int, C.POINTER(pyoracle.PathSeg), C.c_int,
                                                   C.c_int, C.c_int, C.c_int, C.c_uint]
    L.ref_session_time_realign.restype = C.c_double
    L.ref_session_time_realign.argtypes = [vp, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.ref_time_germline_sites.restype = C.c_double
    L.ref_time_germline_sites.argtypes = [vp, vp, vp, C.c_int, C.c_double, C.POINTER(C.c_double)]
    L.ref_time_somatic_sites.restype = C.c_double
    L.ref_time_somatic_sites.argtypes = [vp] * 5 + [C.c_int, vp, C.POINTER(C.c_double)]


def _time_scoring(L, seconds, rng, n_reads):
    from oracle import pyoracle
    from strelka_amd import synth
    cases = synth.align_cases_h64(n_reads, rng)
    sessions, keep = [], []
    for c in cases:
        ref_b = c["ref_seq"].encode() if isinstance(c["ref_seq"], str) else bytes(c["ref_seq"])
        s = L.ref_session_create(ref_b, int(c["ref_offset"]), 0)
        seen = set()
        for cal in c["cals"]:
            for k in list(cal["indels"]) + [cal.get("leading"), cal.get("trailing")]:
                if k is None:
                    continue
                ident = (k["pos"], k["type"], k.get("del_len", 0), k.get("ins_seq", ""))
                if ident in seen:
                    continue
                seen.add(ident)
                ri = pyoracle._ref_indel(k)
                if L.ref_session_add_indel(s, C.byref(ri)) != 0:
                    raise RuntimeError("reference rejected an indel")
        read = "".join(pyoracle._CODE2CHAR.get(int(x), "N") for x in c["read_code"]).encode()
        qual = np.ascontiguousarray(c["read_qual"], np.uint8)
        rcs = (pyoracle.RefCal * len(c["cals"]))()
        for i, cal in enumerate(c["cals"]):
            path = (pyoracle.PathSeg * max(len(cal["path"]), 1))(*[pyoracle.PathSeg(t, l) for t, l in cal["path"]])
            ind = (pyoracle.RefIndel * max(len(cal["indels"]), 1))(*[pyoracle._ref_indel(k) for k in cal["indels"]])
            keep.append((path, ind))
            rcs[i] = pyoracle.RefCal(cal["pos"], len(cal["path"]), path, len(cal["indels"]), ind,
                                     pyoracle._ref_indel(cal.get("leading")), pyoracle._ref_indel(cal.get("trailing")))
        if L.ref_session_prepare_score_read(s, read, qual.ctypes.data, len(qual), C.cast(rcs, vp), len(c["cals"])) != 0:
            raise RuntimeError("reference rejected a read")
        sessions.append(s)
    cells = secs = 0.0
    per = seconds / len(sessions)
    t_end = time.perf_counter() + seconds
    while True:
        for s in sessions:
            n, chk = C.c_double(), C.c_double()
            secs += L.ref_session_time_score(s, per * 0.25, C.byref(n), C.byref(chk))
            cells += n.value
        if time.perf_counter() >= t_end:
            break
    for s in sessions:
        L.ref_session_destroy(s)
    return cells, secs


def _time_realign(L, seconds, rng, n_scenarios, max_indels=6):
    from oracle import pyoracle
    from strelka_amd import synth
    scenarios = synth.realign_scenarios(n_scenarios, rng, reads_per=12, max_indels=max_indels)
    sessions = []
    for sc in scenarios:
        s = L.ref_session_create(sc["ref_seq"].encode(), int(sc["ref_offset"]), 0)
        for d in sc["indels"]:
            ri = pyoracle._ref_indel(d)
            L.ref_session_add_indel(s, C.byref(ri))
        for rid, rd in enumerate(sc["reads"]):
            for o in rd["observed"]:
                ri = pyoracle._ref_indel(sc["indels"][o])
                L.ref_session_add_indel_observed(s, C.byref(ri), rid + 1)
        for d in sc["indels"]:
            if "arid" in d:
                ri = pyoracle._ref_indel(d)
                L.ref_session_set_indel_haplotype(s, C.byref(ri), d["arid"], d["hap"], d["bypass"], d["forced"], d["ndfr"])
        for rid, rd in enumerate(sc["reads"]):
            read = "".join(pyoracle._CODE2CHAR.get(int(x), "N") for x in rd["code"]).encode()
            qual = np.ascontiguousarray(rd["qual"], np.uint8)
            path = (pyoracle.PathSeg * len(rd["path"]))(*[pyoracle.PathSeg(t, l) for t, l in rd["path"]])
            L.ref_session_prepare_realign_read(s, read, qual.ctypes.data, rd["pos"], len(rd["path"]), path, int(rd["is_fwd"]),
                                               rd["map_level"], rd["realign_range"][0], rd["realign_range"][1], rid + 1)
        sessions.append((s, int(sc.get("is_haplotyping_enabled", 0)), int(sc.get("min_read_bp_flank", 5))))
    reads = secs = 0.0
    t_end = time.perf_counter() + seconds
    while True:
        for s, hap, flank in sessions:
            n = C.c_double()
            secs += L.ref_session_time_realign(s, 0.0, hap, flank, C.byref(n))
            reads += n.value
        if time.perf_counter() >= t_end:
            break
    for s, _, _ in sessions:
        L.ref_session_destroy(s)
    return reads, secs


def _time_loci(L, seconds, rng, n_loci):
    from strelka_amd import synth
    pb = synth.pileups(n_loci, rng)
    loci = secs = 0.0
    t_end = time.perf_counter() + seconds
    while True:
        chk = C.c_double()
        secs += L.ref_time_germline_sites(pb.call_off.ctypes.data, pb.calls.ctypes.data, pb.ref_base.ctypes.data, pb.n_loci, 0.001,
                                          C.byref(chk))
        loci += pb.n_loci
        if time.perf_counter() >= t_end:
            break
    return loci, secs


def _time_somatic(L, seconds, rng, n_loci):
    from oracle import pyoracle
    from strelka_amd import synth
    n, t = synth.somatic_pileups(n_loci, rng)
    opt = pyoracle.somatic_snv_options()
    loci = secs = 0.0
    t_end = time.perf_counter() + seconds
    while True:
        chk = C.c_double()
        secs += L.ref_time_somatic_sites(n.call_off.ctypes.data, n.calls.ctypes.data, t.call_off.ctypes.data, t.calls.ctypes.data,
                                         n.ref_base.ctypes.data, n.n_loci, C.addressof(opt), C.byref(chk))
        loci += n.n_loci
        if time.perf_counter() >= t_end:
            break
    return loci, secs


def _worker(job):
    core, seconds, seed = job
    try:
        os.sched_setaffinity(0, {core})
    except (AttributeError, OSError):
        pass
    # the reference logs realignment warnings on stderr; they are not part of the measurement
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 2)
    from oracle import pyoracle
    L = pyoracle.ref()
    _bind(L)
    rng = np.random.default_rng(seed)
    out = {}
    out["cells"] = _time_scoring(L, seconds, rng, 48)
    # the scenarios of bench.py's whole-read legs themselves (whole_read_leg: default_rng(4242), 24 scenarios x 12 reads, up to 6
    # candidate indels around a read; "dense": up to 14), the same in every process
    out["reads"] = _time_realign(L, seconds, np.random.default_rng(4242), 24, max_indels=6)
    out["reads_dense"] = _time_realign(L, seconds, np.random.default_rng(4242), 24, max_indels=14)
    out["loci"] = _time_loci(L, seconds, rng, 20000)
    out["somatic_loci"] = _time_somatic(L, seconds, rng, 20000)
    return out


def usable_cores():
    """cores this process may use: the affinity mask, cut down to the cgroup CPU quota where one is set"""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = list(range(os.cpu_count() or 1))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        cores = cores[:max(1, int(quota + 0.5))]
    return cores


def _run(cores, seconds):
    ctx = mp.get_context("spawn")
    with ctx.Pool(len(cores)) as pool:
        return pool.map(_worker, [(c, seconds, 7000 + i) for i, c in enumerate(cores)])


LEGS = ("cells", "reads", "reads_dense", "loci", "somatic_loci")


def reference_baseline(seconds_per_leg=5.0, processes=None):
    """-> dict: `cores` = P processes run at once (one per usable core); per leg `*_per_s` = sum over the P processes,
    `*_per_s_per_core` = that sum / P, `*_per_s_one_process` = one process running alone (what a core does when the box is
    otherwise idle: hosts that present more logical cores than they can run at full speed show per-core rates well below it)"""
    cores = usable_cores()
    if processes:
        cores = cores[:processes]
    solo = _run(cores[:1], max(1.0, seconds_per_leg / 3.0))[0]
    res = _run(cores, seconds_per_leg)
    out = {"cores": len(cores), "seconds_per_leg": seconds_per_leg}
    for leg in LEGS:
        rates = [r[leg][0] / r[leg][1] for r in res if r[leg][1] > 0]
        out[leg + "_per_s"] = float(np.sum(rates))
        out[leg + "_per_s_per_core"] = float(np.sum(rates) / len(cores))
        out[leg + "_per_s_one_process"] = float(solo[leg][0] / solo[leg][1])
    return out


if __name__ == "__main__":
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    print(json.dumps(reference_baseline(float(sys.argv[1]) if len(sys.argv) > 1 else 3.0), indent=1))
