import sys
import numpy as np
sys.path.insert(0, ".")
from strelka_amd import capi, synth
from tests import test_read_realign as T

capi.init(0)
seed, max_indels, hap = 43, 12, 0.0
rng = np.random.default_rng(91000 + seed)
scs = synth.realign_scenarios(80, rng, reads_per=12, max_indels=max_indels, haplotyping_rate=hap)
shown = 0
for si, sc in enumerate(scs):
    res = {}
    bat = {}
    for mode in (0, 2):
        job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                   min_read_bp_flank=sc["min_read_bp_flank"], enumeration=mode))
        job.set_reference(sc["ref_seq"], sc["ref_offset"])
        job.set_indels(sc["indels"])
        idx = T._add_reads(job, sc)
        job.run()
        res[mode] = [None if i is None else job.result(i) for i in idx]
        bat[mode] = (job, idx)
    for ri, (a, b) in enumerate(zip(res[0], res[2])):
        if a is None or b is None:
            if (a is None) != (b is None):
                print("scenario", si, "read", ri, "None mismatch", a is None, b is None)
            continue
        if repr(a) != repr(b):
            print("scenario", si, "read", ri, "counts", bat[2][0].enumeration_counts())
            for k in a:
                if repr(a[k]) != repr(b[k]):
                    if isinstance(a[k], list):
                        for x, y in zip(a[k], b[k]):
                            if repr(x) != repr(y):
                                print("  ", k, "\n    host", x, "\n    dev ", y)
                        if len(a[k]) != len(b[k]):
                            print("  ", k, "len", len(a[k]), len(b[k]))
                    else:
                        print("  ", k, a[k], b[k])
            shown += 1
            if shown > 3:
                sys.exit(0)
