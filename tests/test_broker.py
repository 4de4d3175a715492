"""The per-GPU broker (strelka_amd/csrc/sk_rt.h, sk_rt.hip, `sk_broker`): many caller processes, ONE GPU context.

SURVEY.md section 8(b) "Threading": the C-ABI is called from one thread per process and by MANY processes per GPU (the workflow runs
one caller process per core, PY/strelkaSharedOptions.py:153-161).  The device gives compute work eight process slots; with
$STRELKA_AMD_BROKER=1 a caller process has no GPU context of its own and its launches, copies and waits are executed by the device's
one server process.

CPU tier (here): the whole client / server path -- rendezvous and start on demand, the ring, page-locked segments at one address in both
processes, staged copies from and to pageable memory, waits, many clients at once, a client that dies, a client of another build -- over
the broker's no-GPU backend (device memory = the server's heap, a launch = a call of a host function).
GPU tier: the same self-test through kernels, the library's own entry points from 16 client processes at once with results identical
to a process that holds its own context, and the caller programs (the adapter) as clients, byte-identical to the reference."""
import ctypes
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time
import uuid

import pytest

from strelka_amd import build as sk_build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLIENT = r'''
import ctypes, os, sys, time
L = ctypes.CDLL(sys.argv[1])
L.sk_last_error.restype = ctypes.c_char_p
if L.sk_broker_client() != 1: sys.exit(10)
n_dev = L.sk_device_count()
if L.sk_init(0) != 0:
    sys.stderr.write(L.sk_last_error().decode()); sys.exit(11)
mode = sys.argv[2]
if mode == "die":  # (a client that is killed in the middle of its work)
    L.sk_broker_selftest(100000, 3)
    os._exit(0)
rc = 0
for n, mul in ((1, 3), (1000, 5), (2500000, 7), (17, 11), (40000, 13)):
    for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
        r = L.sk_broker_selftest(n, mul)
        if r: rc = 20 + r
print("devices", n_dev, "rc", rc)
sys.exit(rc)
'''


def _env(tmp_path, **extra):
    e = dict(os.environ, STRELKA_AMD_BROKER="1", STRELKA_AMD_BROKER_SOCKET="sktest_" + uuid.uuid4().hex[:12],
             STRELKA_AMD_BROKER_LOG=str(tmp_path / "broker.log"), STRELKA_AMD_BROKER_IDLE_S="2", STRELKA_AMD_BROKER_VERBOSE="1")
    e.update(extra)
    return e


def _client(env, mode="run", reps=1, lib=None):
    return subprocess.Popen([sys.executable, "-c", CLIENT, lib or sk_build.LIB_PATH, mode, str(reps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def _wait_for_exit_line(log, timeout=30):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if os.path.exists(log) and "leaving" in open(log).read():
            return open(log).read()
        time.sleep(0.2)
    return open(log).read() if os.path.exists(log) else ""


def test_library_exports_the_broker_and_the_server_is_built(built):
    L = ctypes.CDLL(sk_build.LIB_PATH)
    for s in ("sk_broker_serve", "sk_broker_selftest", "sk_broker_client", "sk_broker_enable"):
        assert hasattr(L, s), s
    assert os.access(sk_build.BROKER_PATH, os.X_OK)
    assert L.sk_broker_client() == (1 if os.environ.get("STRELKA_AMD_BROKER", "0") not in ("", "0") else 0)


def test_a_client_starts_the_server_and_every_kind_of_call_round_trips(built, tmp_path):
    env = _env(tmp_path, STRELKA_AMD_BROKER_BACKEND="host", STRELKA_AMD_BROKER_HOST_DEVICES="3")
    p = _client(env)
    out, err = p.communicate(timeout=120)
    assert p.returncode == 0, (out, err)
    assert b"devices 3 rc 0" in out  # (sk_device_count of a client is the broker's answer)
    log = _wait_for_exit_line(str(tmp_path / "broker.log"))
    assert log.count("serving device 0") == 1 and "host backend" in log
    assert "no client for 2 s after 1 served, leaving" in log  # the server does not outlive its clients


def test_many_clients_share_one_server_and_a_dead_client_is_cleaned_up(built, tmp_path):
    env = _env(tmp_path, STRELKA_AMD_BROKER_BACKEND="host")
    procs = [_client(env, "die" if i == 2 else "run", reps=3) for i in range(6)]
    for i, p in enumerate(procs):
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, (i, out, err)
    late = _client(env)  # (after the crowd, the same server still serves)
    out, err = late.communicate(timeout=120)
    assert late.returncode == 0, (out, err)
    log = _wait_for_exit_line(str(tmp_path / "broker.log"))
    assert log.count("serving device 0") == 1, log
    assert log.count(" left: ") == 7, log
    assert "after 7 served" in log


def test_a_client_of_another_build_is_refused(built, tmp_path):
    env = _env(tmp_path, STRELKA_AMD_BROKER_BACKEND="host", STRELKA_AMD_BROKER_IDLE_S="6")
    first = _client(env)
    out, err = first.communicate(timeout=120)
    assert first.returncode == 0, (out, err)
    other = tmp_path / "otherbuild"
    other.mkdir()
    shutil.copy(sk_build.LIB_PATH, other / "libstrelka_amd.so")  # (another file: another modification time)
    p = _client(dict(env, STRELKA_AMD_BROKER_NO_SPAWN="1"), lib=str(other / "libstrelka_amd.so"))
    out, err = p.communicate(timeout=60)
    assert p.returncode == 11 and b"another build of libstrelka_amd.so" in err, (out, err)


def test_without_a_server_binary_or_with_spawning_off_the_client_fails_loudly(built, tmp_path):
    env = _env(tmp_path, STRELKA_AMD_BROKER_BACKEND="host", STRELKA_AMD_BROKER_NO_SPAWN="1")
    p = _client(env)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 11 and b"no broker is listening" in err, (out, err)


# ---------------------------------------------------------------------------------------------------------------------
# GPU tier

WORKER = r'''
import hashlib, json, sys, time
import numpy as np
sys.path.insert(0, %r)
from strelka_amd import capi, synth
capi.init_strict(0)
rng = np.random.default_rng(99)
scen = synth.realign_scenarios(30, rng)
pb = synth.pileups(1 << 16, np.random.default_rng(98), het_rate=0.01)
n, t = synth.somatic_pileups(1 << 13, np.random.default_rng(97), somatic_rate=0.02, het_rate=0.02)
h = hashlib.sha256()
reps = int(sys.argv[1])
abi = 0.0
for rep in range(reps):
    for sc in scen:
        job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"], min_read_bp_flank=sc["min_read_bp_flank"]))
        job.set_reference(sc["ref_seq"], sc["ref_offset"])
        job.set_indels(sc["indels"])
        idx = []
        for rd in sc["reads"]:
            try:
                idx.append(job.add_read(rd["code"], rd["qual"], rd["pos"], rd["path"], rd["is_fwd"], rd["map_level"], 0, rd["realign_range"], rd["observed"]))
            except capi.StrelkaAmdError:
                pass
        t0 = time.perf_counter()
        job.run()
        abi += time.perf_counter() - t0
        if rep == 0:
            for i in idx:
                r = job.result(i)
                h.update(repr((r["is_realigned"], r["pos"], r["path"], float(r["max_score"]).hex(),
                               [(s["indel"], float(s["ref_lnp"]).hex(), float(s["indel_lnp"]).hex()) for s in r["scores"]])).encode())
    t0 = time.perf_counter()
    out, _ = capi.site_digt_call_fused(pb)
    som = capi.somatic_snv_call(n, t)
    abi += time.perf_counter() - t0
    if rep == 0:
        h.update(out.tobytes())
        h.update(som.tobytes())
import os
def _kfd_open():  # has this process opened the GPU driver's compute device?  (a broker client must not: no context, no process slot)
    for f in os.listdir("/proc/self/fd"):
        try:
            if "kfd" in os.readlink("/proc/self/fd/" + f):
                return True
        except OSError:
            pass
    return False
print(json.dumps(dict(digest=h.hexdigest(), abi_seconds=abi, client=capi.lib().sk_broker_client(), kfd_open=_kfd_open())))
'''


def _worker(reps, env):
    return subprocess.Popen([sys.executable, "-c", WORKER % REPO, str(reps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


@pytest.mark.gpu
def test_gpu_selftest_through_the_broker(built, tmp_path):
    p = _client(_env(tmp_path), reps=2)
    out, err = p.communicate(timeout=300)
    assert p.returncode == 0, (out, err)
    log = _wait_for_exit_line(str(tmp_path / "broker.log"))
    assert "hip backend" in log


@pytest.mark.gpu
def test_thirty_two_clients_of_one_broker(built, tmp_path):
    """four times the device's eight process slots, all at once, through ONE GPU context"""
    env = _env(tmp_path, STRELKA_AMD_BROKER_IDLE_S="5")
    procs = [_client(env, reps=2) for _ in range(32)]
    for i, p in enumerate(procs):
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, (i, out, err)
    log = _wait_for_exit_line(str(tmp_path / "broker.log"))
    assert log.count("serving device 0") == 1 and log.count(" left: ") == 32, log


@pytest.mark.gpu
def test_sixteen_client_processes_equal_a_process_with_its_own_context(built, tmp_path):
    """realignment jobs (the one-sequence device job: ~20 launches, page-locked mirrors read and written by kernels), the fused germline
    site call and the somatic SNV call (host buffers: staged copies) from 16 clients of one broker at once"""
    own = _worker(1, dict(os.environ, STRELKA_AMD_BROKER="0"))
    out, err = own.communicate(timeout=600)
    assert own.returncode == 0, err.decode()[-3000:]
    ref = json.loads(out.decode().strip().splitlines()[-1])
    assert ref["client"] == 0
    env = _env(tmp_path, STRELKA_AMD_BROKER_IDLE_S="5")
    procs = [_worker(3, env) for _ in range(16)]
    res = []
    for p in procs:
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, err.decode()[-3000:]
        res.append(json.loads(out.decode().strip().splitlines()[-1]))
    assert all(r["client"] == 1 for r in res)
    assert all(r["digest"] == ref["digest"] for r in res)
    # a client never touches the GPU driver: no runtime, no context, none of the device's eight process slots (a device-library size
    # query on the pileup push path once did, profiles/r06_v22)
    assert ref["kfd_open"] is True and not any(r["kfd_open"] for r in res)
    log = _wait_for_exit_line(str(tmp_path / "broker.log"))
    assert log.count("serving device 0") == 1 and log.count(" left: ") == 16, log
    print("\n16 broker clients: ABI seconds per process %.2f .. %.2f (3 repetitions); a process with its own context, alone: %.2f (1 repetition)" % (
        min(r["abi_seconds"] for r in res), max(r["abi_seconds"] for r in res), ref["abi_seconds"]))


@pytest.mark.gpu
def test_caller_programs_as_broker_clients_are_byte_identical(built, tmp_path):
    from tests import e2e_util as E
    if not E.have("starling2_ref", "starling2_amd", "strelka2_ref", "strelka2_amd"):
        pytest.skip("oracle/_ref binaries not built")
    bams = [E.demo("NA12891_demo20.bam"), E.demo("NA12892_demo20.bam")]
    gref, sref = str(tmp_path / "gref") + "/", str(tmp_path / "sref") + "/"
    os.makedirs(gref)
    os.makedirs(sref)
    E.run(E.germline_argv("starling2_ref", gref, bams))
    E.run(E.somatic_argv("strelka2_ref", sref, bams[1], bams[0]))
    env = _env(tmp_path, STRELKA_AMD_DEVICE="0", STRELKA_AMD_VERBOSE="1", STRELKA_AMD_BROKER_IDLE_S="5")
    procs = []
    for i in range(12):
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
    log = _wait_for_exit_line(str(tmp_path / "broker.log"))
    assert log.count("serving device 0") == 1 and log.count(" left: ") == 12, log
