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
