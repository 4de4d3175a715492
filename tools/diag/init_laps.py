"""where a caller process's sk_init goes when N of them start together (SK_INIT_TIMING laps), own contexts against broker clients
usage: python tools/diag/init_laps.py [N=16]"""
import os
import re
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, ".")
from strelka_amd import farm

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = 4000000
d = farm.wgs_dataset(L)
for label, env in (("own contexts", {"STRELKA_AMD_BROKER": "0"}), ("broker clients (cold server)", {"STRELKA_AMD_BROKER": "1", "STRELKA_AMD_BROKER_SOCKET": "sk_init_laps"}),
                   ("broker clients (warm server)", {"STRELKA_AMD_BROKER": "1", "STRELKA_AMD_BROKER_SOCKET": "sk_init_laps"})):
    root = tempfile.mkdtemp(prefix="sk_init_laps_")
    procs = []
    t0 = time.time()
    for i in range(N):
        out = os.path.join(root, "p%d_" % i)
        argv = farm.germline_segment_argv("starling2_amd", out, [os.path.join(d, "wgs.bam")], ["chrW:%d-%d" % (1 + i * 200000, i * 200000 + 20000)], os.path.join(d, "wgs.fa"),
                                          chrom_depth=os.path.join(d, "chrom_depth.txt"))
        procs.append(subprocess.Popen(argv, env=dict(os.environ, SK_INIT_TIMING="1", STRELKA_AMD_VERBOSE="1", STRELKA_AMD_BROKER_TIMING="1", **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    laps = {}
    inits = []
    for p in procs:
        _, err = p.communicate(timeout=600)
        text = err.decode(errors="replace")
        prev = 0.0
        for m in re.finditer(r"\[sk_init\] (.*?)\s+t=([0-9.]+) ms", text):
            laps.setdefault(m.group(1), []).append(float(m.group(2)) - prev)
            prev = float(m.group(2))
        m = re.search(r" init=([0-9.e+-]+)", text)
        if m:
            inits.append(float(m.group(1)))
    print("%s, %d processes at once (wall %.2f s): adapter init wait mean %.3f s max %.3f s" % (label, N, time.time() - t0, sum(inits) / max(1, len(inits)), max(inits or [0])))
    for k, v in laps.items():
        print("    %-36s mean %7.1f ms   max %7.1f ms" % (k, sum(v) / len(v), max(v)))
