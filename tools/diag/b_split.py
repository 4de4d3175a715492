"""Where does the fused germline kernel spend its time?  Times it with the dependent-eprob phase on/off and at several depths."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from strelka_amd import capi, device, synth
torch.cuda.set_device(0); capi.init(0)
def t(db, g, n=5):
    db.site_digt_call_fused(g); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): db.site_digt_call_fused(g)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for depth in (10.0, 20.0, 40.0, 80.0):
    rng = np.random.default_rng(5)
    hb = synth.pileups(1 << 19, rng, depth_mean=depth)
    db = device.DevicePileupBatch(hb, "cuda:0", tile=8)
    g = capi.germline_options()
    g0 = capi.germline_options(); g0.bsnp_ssd_no_mismatch = 0.0; g0.bsnp_ssd_one_mismatch = 0.0
    g1 = capi.germline_options(); g1.is_min_vexp = 1; g1.min_vexp = 1.0
    print("depth %5.1f  loci %d  calls %d  fused %.2f ms   sort-but-no-pow/log (min_vexp=1) %.2f ms   no-dependent-eprob %.2f ms" % (depth, db.n_loci, db.n_calls, t(db, g), t(db, g1), t(db, g0)), flush=True)
