"""the headline leg alone (kernel A1 on the bench's workload), for counter passes"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from strelka_amd import capi, device, synth

capi.init(0)
rng = np.random.default_rng(1234)
uniq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 128
columns = (sys.argv[3] != "entries") if len(sys.argv) > 3 else True
ha = synth.build_align_batch(synth.align_cases_h64(uniq, rng))
da = device.DeviceAlignBatch(ha, "cuda:0", tile=tile, columns=columns)
for _ in range(3):
    da.score()
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(5):
    da.score()
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 5
print("reads %d cals %d: %.3f ms  %.3g cells/s" % (da.n_reads, da.n_cals, ms, da.n_cals * 150 / ms * 1e3))
