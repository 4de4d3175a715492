import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from strelka_amd import capi, device, synth
torch.cuda.set_device(0); capi.init(0)
rng = np.random.default_rng(1000)
ha = synth.align_batch_flat(1 << 14, rng)
da = device.DeviceAlignBatch(ha, "cuda:0", tile=16)
def t(n=5):
    da.score(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): da.score()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("%d reads: %.3f ms" % (da.n_reads, t()), flush=True)
