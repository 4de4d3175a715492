"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-launch HBM traffic per kernel.

Units and the gfx950 correction follow /opt/skills/guides (MI355X_MICROARCH.md, HBM section): the counters are in KiB and
FETCH_SIZE reports half the bytes of a wide coalesced read stream, so  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def full_size(values):
    """the launches of a kernel's full-size leg: bench.py also launches several kernels at window size (the pileup stream, the
    slice-sized feed, the whole-read jobs); a kernel's traffic per launch is quoted for its largest launches only"""
    top = max(values)
    return [v for v in values if v >= 0.5 * top] if top > 0 else values


def collect(prefix):
    acc = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        per = defaultdict(list)
        for f in glob.glob(os.path.join(root, prefix + counter, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if row.get("Counter_Name") != counter:
                        continue
                    name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                    if name.startswith("void "):
                        name = name[5:]
                    per[name.split("(")[0].split("<")[0]].append(float(row["Counter_Value"]))
        acc[counter] = per
    res = {}
    for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
        f = acc["FETCH_SIZE"].get(k, [])
        w = acc["WRITE_SIZE"].get(k, [])
        if not f or not w:
            continue
        f, w = full_size(f), full_size(w)
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        res[k] = dict(launches=len(f), fetch_kib=fk, write_kib=wk, hbm_bytes_per_launch=(2 * fk + wk) * 1024,
                      hbm_bytes_uncorrected=(fk + wk) * 1024)
    return res


acc = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    per = defaultdict(list)
    for f in glob.glob(os.path.join(root, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                if name.startswith("void "):
                    name = name[5:]
                per[name.split("(")[0].split("<")[0]].append(float(row["Counter_Value"]))
    acc[counter] = per
out = {}
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    f = acc["FETCH_SIZE"].get(k, [])
    w = acc["WRITE_SIZE"].get(k, [])
    if not f or not w:
        continue
    n_all = len(f)
    f, w = full_size(f), full_size(w)
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    out[k] = dict(launches=len(f), launches_of_any_size=n_all, fetch_kib=fk, write_kib=wk, hbm_bytes_per_launch=(2 * fk + wk) * 1024,
                  hbm_bytes_uncorrected=(fk + wk) * 1024)
# the per-launch workload these counters belong to: bench.py's defaults (tools/gpu_round.sh runs it without size flags);
# bench.py only quotes `traffic` from this file when its own run has the same workload
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
argv, sys.argv = sys.argv, [sys.argv[0]]
a = bench.parse()
sys.argv = argv
workload = dict(reads_per_step_per_gpu=a.reads, loci_per_step_per_gpu=a.loci, unique_reads=a.unique_reads,
                unique_loci=a.unique_loci, pileup_reads=a.pileup_reads, somatic_loci=a.somatic_loci)
# the headline leg alone (bench.py --only a5): a step launches each of these kernels once, all launches of a kernel have one size
a5 = collect("pmc_a5_")
a5_only = None
step_kernels = ("pool_fill_kernel", "flatten_kernel", "entries_wave_kernel", "score_wave_per_read_cols_hostbuf", "score_wave_per_read_cols")
if a5:
    if "flatten_score_kernel" in a5:  # F5: a step is one launch of the fused kernel
        step_kernels = ("flatten_score_kernel",)
    per_step = sum(a5[k]["hbm_bytes_per_launch"] for k in step_kernels if k in a5)
    a5_only = dict(workload=dict(a5_scenarios=a.a5_scenarios, a5_reads=a.a5_reads), hbm_bytes_per_step=per_step,
                   kernels={k: a5[k] for k in step_kernels if k in a5})
# the commit the counter passes ran at: tools/gpu_visit.sh writes `git rev-parse HEAD` of the pushing container into the visit's
# directory before the snapshot leaves (the GPU box has no .git)
commit = None
for cand in (os.path.join(root, "commit.txt"), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_commit.txt")):
    if os.path.exists(cand):
        commit = open(cand).read().strip() or None
        break
json.dump(dict(workload=workload, kernels=out, a5_only=a5_only, commit=commit), sys.stdout, indent=1)
