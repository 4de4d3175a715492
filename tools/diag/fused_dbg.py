import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from strelka_amd import capi as gpu, synth
from tests.test_gpu_parity import _varied_pileups
gpu.init(0)
rng = np.random.default_rng(207)
parts = [_varied_pileups(rng), synth.pileups(40, rng, depth_mean=1500.0, het_rate=0.2),
         synth.pileups(300, rng, depth_mean=150.0, het_rate=0.1), synth.pileups(1, rng, depth_mean=9000.0)]
off = [np.zeros(1, np.int64)]; calls, ref = [], []; base = 0
for p in parts:
    off.append(p.call_off[1:] + base); base += p.call_off[-1]; calls.append(p.calls); ref.append(p.ref_base)
pb = gpu.HostPileupBatch(np.concatenate(off), np.concatenate(calls), np.concatenate(ref))
pb.ploidy = rng.choice(np.array([1, 2, 2], np.uint8), pb.n_loci)
pb.ref_base[::53] = 4
de2 = gpu.dependent_eprob(pb); pb.de = de2
two = gpu.site_digt_call(pb)
fused, de1 = gpu.site_digt_call_fused(pb, want_de=True)
bad = [i for i in range(pb.n_loci) if fused[i].tobytes() != two[i].tobytes()]
print("n_loci", pb.n_loci, "bad", len(bad), bad[:20])
d = np.diff(pb.call_off)
for i in bad[:5]:
    print(i, "depth", d[i], "block", i // 256, "in-block", i % 256)
    for f in fused.dtype.names:
        if fused[i][f].tobytes() != two[i][f].tobytes(): print("   field", f, fused[i][f], two[i][f])
for i in bad:
    c = pb.calls[pb.call_off[i]:pb.call_off[i + 1]]
    q = c & 63; valid = (((c >> 12) & 1) == 0) & (q >= 3)
    g = ((c >> 10) & 1) + 2 * ((c >> 6) & 15)
    cnt = np.bincount(g[valid], minlength=8)
    need = sum(min(x - 1, 3) for x in cnt if x > 1)
    print(i, "counts", cnt.tolist(), "need", need)
