"""per job of the dense scenario set: reads, candidate alignments, device pipeline and host times (SK_ENUM_TIMING lines, condensed)"""
import os
import re
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ".")
    sys.path.insert(0, "tools/diag")
    import numpy as np
    from strelka_amd import capi, synth
    import enum_modes as M
    capi.init(0)
    rng = np.random.default_rng(5)
    scs = synth.realign_scenarios(24, rng, reads_per=12, max_indels=14)
    jobs = M.build(scs, 40, 2)
    M.step(jobs)
    os.environ["SK_ENUM_TIMING"] = "1"
    print("STEP", M.step(jobs), file=sys.stderr)
else:
    out = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True).stderr
    out = out[out.rfind("----") if "----" in out else 0:]
    tot = {}
    for line in out.splitlines():
        m = re.match(r"\[enum-dev\] (.+?)\s+t=([\d.]+) ms", line)
        if m:
            last = (m.group(1), float(m.group(2)))
            tot.setdefault("_cur", []).append(last)
        m = re.match(r"\[enum\] (\d+) reads (\d+) cals: device pipeline ([\d.]+) ms, results -> host structures ([\d.]+) ms", line)
        if m:
            cur = tot.pop("_cur", [])
            prev = 0.0
            parts = []
            for name, t in cur:
                parts.append("%s %.2f" % (name.split()[0], t - prev))
                prev = t
            print("%5s reads %7s cals  pipeline %6s ms  host %5s ms | %s" % (m.group(1), m.group(2), m.group(3), m.group(4), "  ".join(parts)))
        if line.startswith("STEP"):
            print(line)
