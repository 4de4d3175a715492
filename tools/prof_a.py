"""Profile driver: a few launches of the alignment-scoring kernel only (used under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from strelka_amd import capi, device, synth
which = sys.argv[1] if len(sys.argv) > 1 else "a"
torch.cuda.set_device(0); capi.init(0)
rng = np.random.default_rng(1000)
if which == "a":
    ha = synth.align_batch_flat(1 << 14, rng)
    da = device.DeviceAlignBatch(ha, "cuda:0", tile=16)
    for _ in range(3): da.score()
else:
    hb = synth.pileups(1 << 20, rng)
    db = device.DevicePileupBatch(hb, "cuda:0", tile=4)
    g = capi.germline_options()
    for _ in range(3):
        db.site_digt_call_fused(g)
torch.cuda.synchronize()
