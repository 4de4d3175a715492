"""kernel times (read from rocprofv3 --stats) of the indel and aligner kernels on sizeable batches through the host entry points"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from strelka_amd import capi, synth
capi.init(0)
rng = np.random.default_rng(5)
rb = synth.readscore_batch(1 << 17, rng, depth_mean=110.0)
t0 = time.time(); capi.indel_grid_lhood(rb); print("indel_grid_lhood host call: %d indels, %d reads, %.1f ms" % (rb.n_indels, len(rb.ref_lnp), (time.time() - t0) * 1e3))
ab = synth.allele_group_batch(1 << 17, rng, depth_mean=40.0)
t0 = time.time(); capi.allele_group_genotype_lhoods(ab); print("allele_group host call: %d groups, %.1f ms" % (ab.n_groups, (time.time() - t0) * 1e3))
pairs = []
for _ in range(4096):
    R = int(rng.integers(150, 300)); ref = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, R))
    q = list(ref[5:R - 5]); k = int(rng.integers(10, len(q) - 10)); del q[k:k + int(rng.integers(1, 10))]; q[k:k] = list("ACG"[: int(rng.integers(0, 4))])
    pairs.append(("".join(q), ref))
capi.global_align(pairs[:8])
t0 = time.time(); capi.global_align(pairs); print("global_align host call: %d pairs (~220 x 225), %.1f ms" % (len(pairs), (time.time() - t0) * 1e3))
