"""Many processes on one GPU (SURVEY.md 8b "Threading": pyflow -j N >> number of GPUs; every segment process owns its own
context and stream, nothing is shared or locked across processes).

  * four caller processes (two germline, two somatic: the adapter binaries) started together on device 0, every output
    compared with the reference binary's;
  * four library processes started together, each running a realignment job and a germline site batch in a loop: results
    identical in every process and to a process that ran alone; the aggregate rate is printed (-s).
"""
import hashlib
import json
import os
import subprocess
import sys
import time

import pytest

from tests import e2e_util as E

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_amd", "strelka2_ref", "strelka2_amd"), reason="oracle/_ref binaries not built")
def test_four_caller_processes_share_one_gpu(tmp_path):
    bams = [E.demo("NA12891_demo20.bam"), E.demo("NA12892_demo20.bam")]
    gref, sref = str(tmp_path / "gref") + "/", str(tmp_path / "sref") + "/"
    os.makedirs(gref)
    os.makedirs(sref)
    E.run(E.germline_argv("starling2_ref", gref, bams))
    E.run(E.somatic_argv("strelka2_ref", sref, bams[1], bams[0]))
    procs = []
    # (a context per process, which is this file's subject; the adapter's default -- clients of the device's broker -- is tests/test_broker.py's)
    env = dict(os.environ, STRELKA_AMD_DEVICE="0", STRELKA_AMD_VERBOSE="1", STRELKA_AMD_BROKER="0")
    for i in range(4):
        out = str(tmp_path / ("p%d" % i)) + "/"
        os.makedirs(out)
        cmd = (E.germline_argv("starling2_amd", out, bams) if i % 2 == 0 else E.somatic_argv("strelka2_amd", out, bams[1], bams[0]))
        procs.append((i, out, subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
    for i, out, p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err.decode()[-2000:]
        assert b"strelka_amd adapter:" in err
        files = ("variants.vcf", "genome.S1.vcf", "genome.S2.vcf") if i % 2 == 0 else ("somatic.snvs.vcf", "somatic.indels.vcf")
        for f in files:
            assert E.vcf_body(out + f, True) == E.vcf_body((gref if i % 2 == 0 else sref) + f, True), (i, f)


WORKER = r'''
import hashlib, json, sys, time
import numpy as np
sys.path.insert(0, %r)
from strelka_amd import capi, synth
capi.init_strict(0)
rng = np.random.default_rng(99)
scen = synth.realign_scenarios(60, rng)
pb = synth.pileups(1 << 17, np.random.default_rng(98), het_rate=0.01)
h = hashlib.sha256()
reps = int(sys.argv[1])
t0 = time.time()
for rep in range(reps):
    for sc in scen:
        job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                   min_read_bp_flank=sc["min_read_bp_flank"]))
        job.set_reference(sc["ref_seq"], sc["ref_offset"])
        job.set_indels(sc["indels"])
        idx = []
        for rd in sc["reads"]:
            try:
                idx.append(job.add_read(rd["code"], rd["qual"], rd["pos"], rd["path"], rd["is_fwd"], rd["map_level"], 0,
                                        rd["realign_range"], rd["observed"]))
            except capi.StrelkaAmdError:
                pass
        job.run()
        if rep == 0:
            for i in idx:
                r = job.result(i)
                h.update(repr((r["is_realigned"], r["pos"], r["path"], float(r["max_score"]).hex(),
                               [(s["indel"], float(s["ref_lnp"]).hex(), float(s["indel_lnp"]).hex()) for s in r["scores"]])).encode())
    out, _ = capi.site_digt_call_fused(pb)
    if rep == 0:
        h.update(out.tobytes())
dt = time.time() - t0
print(json.dumps(dict(digest=h.hexdigest(), seconds=dt, loci=reps * pb.n_loci, reads=reps * sum(len(s["reads"]) for s in scen))))
'''


def _spawn(reps):
    return subprocess.Popen([sys.executable, "-c", WORKER % REPO, str(reps)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def test_four_library_processes_share_one_gpu():
    alone = _spawn(2)
    out, err = alone.communicate(timeout=600)
    assert alone.returncode == 0, err.decode()[-3000:]
    ref = json.loads(out.decode().strip().splitlines()[-1])
    t0 = time.time()
    procs = [_spawn(6) for _ in range(4)]
    res = []
    for p in procs:
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, err.decode()[-3000:]
        res.append(json.loads(out.decode().strip().splitlines()[-1]))
    wall = time.time() - t0
    assert all(r["digest"] == ref["digest"] for r in res)
    print("\n4 processes on one GPU: %.3g loci/s and %.3g reads/s aggregate (wall %.1f s incl. start-up; alone: %.3g loci/s)" % (
        sum(r["loci"] for r in res) / max(r["seconds"] for r in res), sum(r["reads"] for r in res) / max(r["seconds"] for r in res),
        wall, ref["loci"] / ref["seconds"]))
