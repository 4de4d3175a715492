"""which threads of the drop-in's caller processes use the CPU when 16 of them share the GPU: runs the bench's segment farm and, for
every process, keeps the last /proc/<pid>/task/*/stat snapshot before it exits (thread name, user and system seconds).

usage: python tools/diag/thread_times.py [LENGTH=16000000] [SEGMENT=1000000] [JOBS=cores]"""
import os
import subprocess
import sys
import tempfile
import time
import shutil

sys.path.insert(0, ".")
from strelka_amd import farm

TICK = os.sysconf("SC_CLK_TCK")


def snapshot(pid):
    out = {}
    try:
        for tid in os.listdir("/proc/%d/task" % pid):
            with open("/proc/%d/task/%s/stat" % (pid, tid)) as f:
                s = f.read()
            name = s[s.index("(") + 1:s.rindex(")")]
            rest = s[s.rindex(")") + 2:].split()
            out[tid] = (name, int(rest[11]) / TICK, int(rest[12]) / TICK)
    except (OSError, ValueError):
        pass
    return out


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 16000000
    seg = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    jobs = int(sys.argv[3]) if len(sys.argv) > 3 else len(farm.usable_cores())
    d = farm.wgs_dataset(L)
    groups = [[s] for s in farm.chrom_intervals(["chrW"], {"chrW": L}, seg)]
    for binary, env in (("starling2_ref", {}), ("starling2_amd", {}), ("starling2_amd", {"STRELKA_AMD_PILEUP": "0", "STRELKA_AMD_FEED": "0"})):
        root = tempfile.mkdtemp(prefix="sk_threads_")
        e = dict(os.environ)
        e.update(env)
        procs, last = [], {}
        t0 = time.perf_counter()
        pending = list(enumerate(groups))
        running = []
        while pending or running:
            while pending and len(running) < jobs:
                i, g = pending.pop(0)
                prefix = os.path.join(root, "seg%04d." % i)
                argv = farm.germline_segment_argv(binary, prefix, [os.path.join(d, "wgs.bam")], [farm.region_arg(s) for s in g], os.path.join(d, "wgs.fa"),
                                                  chrom_depth=os.path.join(d, "chrom_depth.txt"), skip_header=(i != 0))
                p = subprocess.Popen(argv, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=e)
                running.append(p)
            for p in list(running):
                snap = snapshot(p.pid)
                if snap:
                    last[p.pid] = snap
                if p.poll() is not None:
                    running.remove(p)
            time.sleep(0.02)
        wall = time.perf_counter() - t0
        shutil.rmtree(root, ignore_errors=True)
        agg = {}
        for pid, snap in last.items():
            main_tid = str(pid)
            for tid, (name, u, s) in snap.items():
                key = "main" if tid == main_tid else name
                a = agg.setdefault(key, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += u
                a[2] += s
        print("%s %s jobs %d: wall %.2f s" % (binary, env, jobs, wall))
        for k, (n, u, s) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            print("    %-24s threads %3d  user %7.2f  sys %7.2f" % (k, n, u, s))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
