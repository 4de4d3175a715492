"""Launch the indel-likelihood and GlobalAligner kernels on bench-sized inputs (for rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from strelka_amd import capi, synth
torch.cuda.set_device(0); capi.init(0)
rng = np.random.default_rng(7)
N = 1 << 18
ag = synth.allele_group_batch(N, rng)
for _ in range(3): capi.allele_group_genotype_lhoods(ag)
nb = synth.readscore_batch(N, rng, depth_mean=40.0)
tb = synth.readscore_batch(N, rng, depth_mean=110.0)
tb.del_len, tb.ins_len = nb.del_len, nb.ins_len
err = np.full(N, 5e-5)
for _ in range(3): capi.somatic_indel_call(nb, tb, err)
for _ in range(3): capi.indel_grid_lhood(tb)
bases = np.array(list("ACGT"))
def seq(n): return "".join(bases[rng.integers(0, 4, n)])
pairs = []
for _ in range(4096):
    r = seq(int(rng.integers(100, 270)))
    q = list(r)
    for _k in range(int(rng.integers(0, 4))):
        p = int(rng.integers(5, len(q) - 5))
        if rng.random() < 0.5: del q[p:p + int(rng.integers(1, 12))]
        else: q[p:p] = list(seq(int(rng.integers(1, 12))))
    pairs.append(("".join(q), r))
for _ in range(3): capi.global_align(pairs)
print("cells per global_align launch:", sum(len(q) * len(r) for q, r in pairs))
print("reads per allele-group launch:", int(ag.read_off[-1]), "somatic indel reads:", int(nb.read_off[-1] + tb.read_off[-1]))
