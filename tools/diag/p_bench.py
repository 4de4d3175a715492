"""a8 throughput: host-buffer entry point timing split is not meaningful; times the device entry point on resident data."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from strelka_amd import capi, synth
from oracle import pyoracle
torch.cuda.set_device(0); capi.init(0)
rng = np.random.default_rng(5)
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
t0 = time.time(); rb, n_loci = synth.pileup_reads_flat(n_reads, rng); print("gen %.1fs reads %d loci %d" % (time.time() - t0, n_reads, n_loci), flush=True)
dev = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
t = dict(read_off=T(rb.read_off), read_code=T(rb.read_code), read_qual=T(rb.read_qual), path_off=T(rb.path_off), path=T(rb.path.view(np.int32)),
         pos=T(rb.pos), is_fwd=T(rb.is_fwd), mapq=T(rb.mapq), map_level=T(rb.map_level), ref=T(np.frombuffer(rb.ref_seq.encode(), np.uint8).copy()))
s = capi.ReadBatchStruct(rb.n_reads, *[t[k].data_ptr() for k in ("read_off", "read_code", "read_qual", "path_off", "path", "pos", "is_fwd", "mapq", "map_level", "ref")], 0, len(rb.ref_seq), None)
opt = capi.pileup_options(report_begin=0, report_end=n_loci)
cap = rb.n_bases + 16
call_off = torch.empty(n_loci + 1, dtype=torch.int64, device=dev); calls = torch.empty(cap, dtype=torch.int16, device=dev)
sd = torch.empty(n_loci, dtype=torch.int32, device=dev); sm = torch.empty(n_loci, dtype=torch.int32, device=dev)
scratch = torch.empty(capi.lib().sk_pileup_scratch_bytes(rb.n_reads, rb.n_bases, n_loci), dtype=torch.uint8, device=dev)
out = capi.PileupColumns(n_loci, cap, call_off.data_ptr(), calls.data_ptr(), sd.data_ptr(), sm.data_ptr())
def run():
    capi._check(capi.lib().sk_pileup_reads_dev(C.byref(s), rb.n_bases, C.byref(opt), capi.PILEUP_CLEAN_TIER1, C.byref(out), scratch.data_ptr(), torch.cuda.current_stream().cuda_stream))
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("pileup: %.3f ms  %.3e read bases/s  %.3e loci/s  calls %d" % (dt * 1e3, rb.n_bases / dt, n_loci / dt, int(call_off[-1].item())), flush=True)
if n_reads <= 1 << 16:
    co, c, _, _ = pyoracle.pileup_reads(rb, pyoracle.pileup_options(report_begin=0, report_end=n_loci), 2)
    print("matches oracle:", np.array_equal(co, call_off.cpu().numpy()), np.array_equal(c, calls[:co[-1]].cpu().numpy().view(np.uint16)))
