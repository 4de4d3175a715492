"""throughput of the feed kernels: a BGZF file image tiled on the device to ~10^5 blocks, inflated and decoded"""
import sys
import time
import ctypes as C
import numpy as np
import torch
sys.path.insert(0, ".")
from strelka_amd import capi
from oracle import bam_oracle

capi.init(0)
path = sys.argv[1] if len(sys.argv) > 1 else "oracle/_ref/synth/somatic_tumor.bam"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 64
data = np.frombuffer(open(path, "rb").read(), np.uint8)
block_off, out_off = capi.bgzf_scan(data)
nb = len(block_off) - 1
dev = "cuda:0"
d_data = torch.from_numpy(data.copy()).to(dev).repeat(tile)
k = torch.arange(tile, dtype=torch.int64)[:, None]
boff = torch.cat([(torch.from_numpy(block_off[:-1])[None, :] + k * len(data)).reshape(-1), torch.tensor([len(data) * tile])]).to(dev)
ooff = torch.cat([(torch.from_numpy(out_off[:-1])[None, :] + k * int(out_off[-1])).reshape(-1), torch.tensor([int(out_off[-1]) * tile])]).to(dev)
out = torch.empty(int(out_off[-1]) * tile, dtype=torch.uint8, device=dev)
status = torch.empty(nb * tile, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    capi._check(capi.lib().sk_bgzf_inflate_dev(C.c_void_p(d_data.data_ptr()), C.c_void_p(boff.data_ptr()), C.c_void_p(ooff.data_ptr()), nb * tile,
                                               C.c_void_p(out.data_ptr()), C.c_void_p(status.data_ptr()), C.c_void_p(st)))


run()
torch.cuda.synchronize()
assert int(status.abs().sum().item()) == 0
want = bam_oracle.bgzf_inflate(data.tobytes())
assert out[:len(want)].cpu().numpy().tobytes() == want and out[-len(want):].cpu().numpy().tobytes() == want
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
t0 = time.perf_counter()
bam_oracle.bgzf_inflate(data.tobytes())
cpu = time.perf_counter() - t0
print("%d blocks, %.1f MB in, %.1f MB out: %.3f ms -> %.1f GB/s inflated (%.1f GB/s compressed); zlib on one host core %.2f GB/s" %
      (nb * tile, len(data) * tile / 1e6, out.numel() / 1e6, ms, out.numel() / ms / 1e6, len(data) * tile / ms / 1e6, len(want) / cpu / 1e9))
