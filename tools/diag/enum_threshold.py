"""at which job size the device read path (enumeration = 2: search, F5, stage 3 on the device) overtakes the host one: seconds per job
of `rep x 12` reads (rep = 1 .. 64), both modes, sparse WGS-like scenarios (up to 3 candidate indels around a read) and denser ones.
What the adapter's $STRELKA_AMD_DEVICE_ENUM_MIN_READS is set from (adapter/sk_adapter_realign.cpp)."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from strelka_amd import capi, synth
from tools.diag.enum_modes import build, step


def main():
    capi.init(0)
    for name, kw in (("wgs-like (<= 3 indels)", dict(max_indels=3)), ("denser (<= 6 indels)", dict(max_indels=6))):
        scs = synth.realign_scenarios(12, np.random.default_rng(4242), reads_per=12, **kw)
        print(name)
        for rep in (1, 2, 4, 8, 16, 32, 64, 128):
            row = []
            for mode in (0, 2):
                jobs = build(scs, rep, mode)
                step(jobs)
                t0 = time.perf_counter()
                n_it = 5
                for _ in range(n_it):
                    step(jobs)
                dt = (time.perf_counter() - t0) / n_it / len(jobs)
                reads = sum(j[2] for j in jobs) / len(jobs)
                row.append((reads, dt))
            print("  reads/job %6.0f   host %8.1f us   device %8.1f us   device/host %.2f" % (row[0][0], row[0][1] * 1e6, row[1][1] * 1e6, row[1][1] / row[0][1]))


if __name__ == "__main__":
    main()
