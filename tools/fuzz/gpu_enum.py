"""Device enumeration against the host code on fresh seeded scenarios (run on the GPU box): every read's result through
sk_realign_job with enumeration = 2 (search, ordering, flattening, scoring and stage 3 in kernels) must equal enumeration = 0 (the
container-based host statement, which tools/fuzz/fuzz_realign.py pins to the reference itself) -- the full per-read record, the
candidate status lookups reported to the adapter, and the batch rebuilt from the device's candidate alignments.

usage: python tools/fuzz/gpu_enum.py [n_rounds=40] [first_seed=1]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from strelka_amd import capi, synth
from tests import test_read_realign as T


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    capi.init(0)
    reads = cals = dev = host_instead = s3_dev = 0
    t0 = time.time()
    for k in range(rounds):
        rng = np.random.default_rng(7_000_000 + first + k)
        max_indels = int(rng.choice([4, 6, 9, 12, 14]))
        hap = float(rng.choice([0.0, 0.25, 0.6]))
        for sc in synth.realign_scenarios(60, rng, reads_per=int(rng.choice([6, 12, 24])), max_indels=max_indels, haplotyping_rate=hap):
            res, cons, off = {}, {}, {}
            for mode in (0, 2):
                job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                           min_read_bp_flank=sc["min_read_bp_flank"], enumeration=mode))
                job.set_reference(sc["ref_seq"], sc["ref_offset"])
                job.set_indels(sc["indels"])
                idx = T._add_reads(job, sc)
                job.run()
                res[mode] = [None if i is None else repr(job.result(i)) for i in idx]
                cons[mode] = job.indels_consulted()
                off[mode] = np.array(job.batch().cal_off)
                if mode == 2:
                    c = job.enumeration_counts()
                    dev += c[1]
                    host_instead += c[2]
                    s3_dev += job.stage3_counts()[1]
            if res[0] != res[2] or not np.array_equal(cons[0], cons[2]) or not np.array_equal(off[0], off[2]):
                print("MISMATCH: round", k, "seed", 7_000_000 + first + k)
                sys.exit(1)
            reads += sum(r is not None for r in res[0])
            cals += int(off[0][-1])
    one_wait, redone, staged = capi.RealignJob.device_job_counts()
    print("device enumeration == host enumeration: %d reads, %d candidate alignments, %d enumerated on the device (stage 3 on the device: %d), "
          "%d handed to the host; device jobs: %d as one sequence with one wait, %d of them run again the staged way, %d staged in all; %.0f s"
          % (reads, cals, dev, s3_dev, host_instead, one_wait, redone, staged, time.time() - t0))


if __name__ == "__main__":
    main()
