"""N>1 path on CPU: two processes, gloo backend.  The data path has no collective (independent genome segments per
rank); what is covered here is the sharding plan and the timed-region protocol bench.py uses (barrier both sides,
elapsed = max over ranks, units = sum over ranks)."""
import os
import socket
import subprocess
import sys
import textwrap

from strelka_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_segment_plan_is_a_partition():
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 260):
            got = sorted(s for r in range(world) for s in shard.segments_for_rank(n, r, world))
            assert got == list(range(n))
            sizes = [len(shard.segments_for_rank(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from strelka_amd import shard
    rank, local_rank, world = shard.env_rank()
    dist.init_process_group(backend="gloo")
    region = shard.Region(dist, world, None, lambda v: torch.tensor(v, dtype=torch.float64))
    segs = shard.segments_for_rank(7, rank, world)
    calls = []
    # rank 1 is the slow one: the reported time must be ITS time on both ranks
    dt, units = region.timed(lambda: (calls.append(1), time.sleep(0.05 * (1 + 2 * rank))), steps=4, warmup=1,
                             units_per_step=100 * len(segs))
    print(json.dumps(dict(rank=rank, world=world, segs=segs, dt=dt, units=units, calls=len(calls))), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_timed_region(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [d["segs"] for d in outs] == [[0, 2, 4, 6], [1, 3, 5]]
    assert all(d["calls"] == 5 for d in outs)                     # 1 warm-up + exactly 4 timed steps
    assert outs[0]["dt"] == outs[1]["dt"] and outs[0]["dt"] >= 4 * 0.15 * 0.95   # max over ranks (slow rank: 0.15 s/step)
    assert outs[0]["units"] == outs[1]["units"] == 4 * 100 * 7    # whole-job units: sum over ranks


# ---- the end-to-end legs with N > 1 (bench.py under torch.distributed.run): every rank farms the whole sample on ITS device with
# cores / N caller processes, rank 0 makes the data set, barriers bracket the clock, the wall time is the max over ranks.  Here: two
# ranks over gloo, the caller processes on the CPU double of the C-ABI.

E2E_WORKER = textwrap.dedent("""
    import argparse, json, os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import bench
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    os.environ["SK_E2E_VARIANT"] = "dbl"
    args = argparse.Namespace(e2e_bp=300000, e2e_segment_bp=100000, e2e_somatic_bp=120000, e2e_somatic_segment_bp=40000, e2e_max_procs_per_gpu=2)
    out = {}
    for mode in ("germline", "somatic"):
        o = bench.e2e_leg(args, rank, world, local_rank, dist.barrier, max_over_ranks, with_reference=False, mode=mode)
        out[mode] = {k: o[k] for k in ("bp", "segments", "amd_wall_s", "amd_procs", "amd_procs_per_gpu")}
        out[mode]["pileup_abi"] = o["hook_seconds"].get("pileup_abi", 0.0)
    print(json.dumps(dict(rank=rank, out=out)), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_end_to_end_legs(tmp_path):
    import json
    import pytest
    from tests import e2e_util as E
    if not E.have("starling2_dbl", "strelka2_dbl"):
        pytest.skip("oracle/_ref binaries not built")
    script = tmp_path / "e2e_worker.py"
    script.write_text(E2E_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    for mode, bp in (("germline", 300000), ("somatic", 120000)):
        a, b = outs[0]["out"][mode], outs[1]["out"][mode]
        assert a["bp"] == b["bp"] == 2 * bp and a["segments"] == 6            # whole-job units: both ranks' samples
        assert a["amd_wall_s"] == b["amd_wall_s"] > 0                         # max over ranks, the same on both
        assert a["amd_procs"] == 2 * a["amd_procs_per_gpu"] and a["pileup_abi"] > 0 and b["pileup_abi"] > 0
