#!/usr/bin/env python
"""Where a device realignment job's time goes in the drop-in: the germline leg with every job on the device, (a) eight caller processes
sharing the GPU and (b) one alone -- the library's own split of a job's wall time (set-up, submissions, the wait: $SK_ENUM_JOB_SECONDS) --
and (c) one process under rocprofv3 --kernel-trace: per job (root_kernel .. the second stage3_kernel) the sum of the kernels' durations
against the span from the first kernel's start to the last one's end.

usage: enum_job_profile.py <out dir> [bp] [segment bp]"""
import csv
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from strelka_amd import farm  # noqa: E402

OUTPUTS = ("variants.vcf", "genome.S1.vcf")


def main():
    out_dir = sys.argv[1]
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 32000000
    seg_bp = int(sys.argv[3]) if len(sys.argv) > 3 else 4000000
    os.makedirs(out_dir, exist_ok=True)
    d = farm.wgs_dataset(L)
    root = tempfile.mkdtemp(prefix="sk_enumprof_")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_dummy_germline_models.py"), os.path.join(root, "models")], check=True)
    evs = (os.path.join(root, "models", "germlineSNVScoringModels.json"), os.path.join(root, "models", "germlineIndelScoringModels.json"))
    groups = [[s] for s in farm.chrom_intervals(["chrW"], {"chrW": L}, seg_bp)]

    def argv_fn(index, regions, prefix, skip_header):
        return farm.germline_segment_argv("starling2_amd", prefix, [os.path.join(d, "wgs.bam")], regions, os.path.join(d, "wgs.fa"),
                                          chrom_depth=os.path.join(d, "chrom_depth.txt"), skip_header=skip_header, evs_models=evs)
    farm.run_farm([[(0, "chrW", 1, 50000, 0)]], argv_fn, os.path.join(root, "warm"), OUTPUTS, jobs=1)
    report = {}
    for name, grp, jobs in (("eight_processes", groups, 8), ("one_process", groups[:1], 1)):
        for mode, extra in (("device_one_wait", {"STRELKA_AMD_DEVICE_ENUM_MIN_READS": "1"}), ("host", {"STRELKA_AMD_DEVICE_ENUM_MIN_READS": "1000000"}),
                            ("device_one_wait_copy_calls", {"STRELKA_AMD_DEVICE_ENUM_MIN_READS": "1", "SK_ENUM_STAGE_COPIES": "1"}),
                            ("device_three_waits", {"STRELKA_AMD_DEVICE_ENUM_MIN_READS": "1", "SK_ENUM_ONE_WAIT": "0"})):
            env = dict(extra, STRELKA_AMD_VERBOSE="1", SK_ENUM_JOB_SECONDS="1")
            res = farm.run_farm(grp, argv_fn, os.path.join(root, name + mode), OUTPUTS, jobs=jobs, env=env)
            acc = {}
            for tail in res.stderr_tails:
                for pat in (r"strelka_amd enum job seconds: (.*)", r"strelka_amd adapter seconds: (.*)"):
                    m = re.search(pat, tail)
                    if m:
                        for kv in m.group(1).split():
                            k, v = kv.split("=")
                            acc[k] = round(acc.get(k, 0.0) + float(v), 4)
            acc["wall_s"] = res.wall_s
            report[name + "/" + mode] = acc
            print(name, mode, json.dumps(acc), flush=True)
    # (c) kernel trace of one process
    prefix = os.path.join(root, "trace.")
    argv = argv_fn(0, [farm.region_arg(s) for s in groups[0]], prefix, False)
    tdir = os.path.join(out_dir, "ktrace")
    env = dict(os.environ, STRELKA_AMD_DEVICE_ENUM_MIN_READS="1", TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", tdir, "-o", "kt", "--"] + argv, env=env, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, check=False)
    rows = []
    for f in glob.glob(os.path.join(tdir, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    per = {}
    jobs, cur = [], None
    for s, e, k in rows:
        name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = per.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
        if name.startswith("root_kernel"):
            cur = {"start": s, "busy": 0, "end": e, "n": 0}
            jobs.append(cur)
        if cur is not None:
            cur["busy"] += e - s
            cur["end"] = e
            cur["n"] += 1
            if name.startswith("stage3_kernel<4>"):
                cur = None
    spans = [j["end"] - j["start"] for j in jobs]
    busys = [j["busy"] for j in jobs]
    report["kernel_trace_one_process"] = {
        "jobs": len(jobs), "kernels_per_job": sum(j["n"] for j in jobs) / max(1, len(jobs)),
        "span_us_mean": sum(spans) / max(1, len(spans)) / 1e3, "busy_us_mean": sum(busys) / max(1, len(busys)) / 1e3,
        "kernels": {k: {"launches": v[0], "mean_us": v[1] / v[0] / 1e3} for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]}}
    print(json.dumps(report["kernel_trace_one_process"], indent=1))
    with open(os.path.join(out_dir, "enum_job_profile.json"), "w") as f:
        json.dump(report, f, indent=1)
    subprocess.run(["rm", "-rf", tdir])


if __name__ == "__main__":
    main()
