"""Multi-GPU plumbing: the path shards by independent genome segments (the reference already runs one process per
12 Mb segment, src/python/lib/workflowUtil.py:315-332), so there is NO data-path collective.  One process per GPU;
torch.distributed is used only to bracket a timed region (barrier) and to combine per-rank elapsed time (max) and unit
counts (sum).  Backend: "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests."""
import os
import time


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def segments_for_rank(n_segments, rank, world):
    """round-robin: segment i -> rank i mod world; disjoint and complete by construction"""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return list(range(rank, n_segments, world))


class Region:
    """times K calls of `fn` between two (barrier + device sync) brackets; combines ranks as the bench contract asks"""

    def __init__(self, dist=None, world=1, device_sync=None, make_tensor=None):
        self.dist, self.world = dist, world
        self.device_sync = device_sync or (lambda: None)
        self.make_tensor = make_tensor

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.device_sync()

    def timed(self, fn, steps, warmup, units_per_step=0, on_step=None):
        """-> (elapsed seconds: max over ranks, units: sum over ranks of units_per_step*steps)"""
        for _ in range(warmup):
            fn()
        self.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            if on_step:
                on_step(i, True)
            fn()
            if on_step:
                on_step(i, False)
        self.barrier()
        dt = time.perf_counter() - t0
        units = float(units_per_step) * steps
        if self.world > 1:
            t = self.make_tensor([dt])
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t[0].item())
            u = self.make_tensor([units])
            self.dist.all_reduce(u, op=self.dist.ReduceOp.SUM)
            units = float(u[0].item())
        return dt, units
