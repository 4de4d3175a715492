"""the BGZF inflation kernels against each other at the bench's full launch size and at one slice: thread per block with the code
tables in LDS (default for large launches), the same with private (scratch) tables (round 1), wave per block.
usage: python tools/diag/inflate_kernels.py [blocks=131072]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from strelka_amd import capi, device


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    capi.init(0)
    dev = torch.device("cuda:0")
    with open("tests/golden/feed_tiny.bam", "rb") as f:
        image = np.frombuffer(f.read(), np.uint8)
    nfix = len(capi.bgzf_scan(image)[0]) - 1
    for blocks in (n, 512):
        for kern in ("thread", "wave"):
            if kern == "wave" and blocks > 20000:
                continue
            os.environ["SK_INFLATE_KERNEL"] = kern
            b = device.DeviceBgzfBatch(image, dev, tile=max(1, blocks // nfix))
            for _ in range(2):
                b.inflate()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            reps = 5
            for _ in range(reps):
                b.inflate()
            ev[1].record()
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / reps
            assert int(b.status.abs().sum().item()) == 0
            print("%-15s %7d blocks: %8.2f ms  %6.2f GB/s inflated" % (kern, b.n_blocks, ms, b.out_bytes / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
