import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from strelka_amd import capi, device, synth
torch.cuda.set_device(0); capi.init(0)
rng = np.random.default_rng(5)
n, t = synth.somatic_pileups(1 << 18, rng)
dn = device.DevicePileupBatch(n, "cuda:0", tile=4); dt = device.DevicePileupBatch(t, "cuda:0", tile=4)
def run(): device.somatic_snv_call_dev(dn, dt)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
print("somatic SNV: %d loci, normal calls %d, tumor calls %d: %.2f ms  %.3e loci/s" % (dn.n_loci, dn.n_calls, dt.n_calls, ms, dn.n_loci / ms * 1e3))
print("queued loci:", int(dn.som_scratch.view(torch.int32)[0].item()), "of", dn.n_loci)
