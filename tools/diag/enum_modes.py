"""whole-read leg (add_reads + run) per enumeration mode on sparse and dense scenario sets: reads/s, candidate alignments per read,
and how many reads the device kept"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from strelka_amd import capi, synth


def build(scenarios, rep, mode, host_threads=1):
    jobs = []
    for sc in scenarios:
        keep, inputs = [], []
        for rd in sc["reads"]:
            code = np.ascontiguousarray(rd["code"], np.uint8)
            qual = np.ascontiguousarray(rd["qual"], np.uint8)
            segs = (capi.PathSeg * max(len(rd["path"]), 1))(*[capi.PathSeg(t, l) for t, l in rd["path"]])
            obs = (C.c_int32 * max(len(rd["observed"]), 1))(*rd["observed"])
            keep.append((code, qual, segs, obs))
            inputs.append(capi.ReadInput(capi._p(code), capi._p(qual), len(code), rd["pos"], len(rd["path"]), segs, int(rd["is_fwd"]),
                                         rd["map_level"], 0, rd["realign_range"][0], rd["realign_range"][1], len(rd["observed"]), obs))
        job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                   min_read_bp_flank=sc["min_read_bp_flank"], enumeration=0, host_threads=host_threads))
        job.set_reference(sc["ref_seq"], sc["ref_offset"])
        job.set_indels(sc["indels"])
        ok = [r for r in inputs if capi.lib().sk_realign_job_add_read(job._j, C.byref(r)) >= 0]
        job.clear_reads()
        if not ok:
            continue
        job2 = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                    min_read_bp_flank=sc["min_read_bp_flank"], enumeration=mode, host_threads=host_threads))
        job2.set_reference(sc["ref_seq"], sc["ref_offset"])
        job2.set_indels(sc["indels"])
        n = len(ok) * rep
        arr = (capi.ReadInput * n)(*[ok[i % len(ok)] for i in range(n)])
        jobs.append((job2, arr, n, keep))
    return jobs


def step(jobs):
    t_add = t_run = 0.0
    for job, arr, n, _ in jobs:
        job.clear_reads()
        t0 = time.perf_counter()
        if capi.lib().sk_realign_job_add_reads(job._j, arr, n) < 0:
            raise RuntimeError("add_reads: " + job.error())
        t1 = time.perf_counter()
        job.run()
        t2 = time.perf_counter()
        t_add += t1 - t0
        t_run += t2 - t1
    return t_add, t_run


def main():
    capi.init(0)
    for name, kw, rep in (("sparse", dict(max_indels=6), 200), ("dense", dict(max_indels=14), 40)):
        rng = np.random.default_rng(5)
        scs = synth.realign_scenarios(24, rng, reads_per=12, **kw)
        for mode, s3 in ((0, None), (1, None), (2, "0"), (2, "1")):  # (2, "0"): device enumeration + scoring, stage 3 on the host
            import os
            if s3 is None:
                os.environ.pop("SK_STAGE3_DEVICE", None)
            else:
                os.environ["SK_STAGE3_DEVICE"] = s3
            jobs = build(scs, rep, mode)
            step(jobs)
            ta, tr = step(jobs)
            reads = sum(j[2] for j in jobs)
            cnt = np.sum([j[0].enumeration_counts() for j in jobs], axis=0)
            cnt3 = np.sum([j[0].stage3_counts() for j in jobs], axis=0)
            cals = sum(j[0].batch().n_cals for j in jobs)
            print("%s mode %d%s: %d reads %.1f cals/read  add %.1f ms run %.1f ms  -> %.3g reads/s  (core,dev,fallback)=%s stage3(core,dev)=%s" %
                  (name, mode, "" if s3 is None else " stage3-dev=" + s3, reads, cals / reads, ta * 1e3, tr * 1e3, reads / (ta + tr),
                   cnt.tolist(), cnt3.tolist()), flush=True)


if __name__ == "__main__":
    main()
