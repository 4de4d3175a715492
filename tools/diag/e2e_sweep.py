#!/usr/bin/env python
"""The germline end-to-end leg under several adapter settings: wall seconds of the drop-in's farm, the share of the realignment jobs'
reads whose candidate alignments the device listed, the hook timers -- each run compared byte for byte with the reference's output.

usage: e2e_sweep.py <out.json> [bp] [segment bp] [procs] -- settings are the ENV_SETS below; $SK_SWEEP_DEPTH / $SK_SWEEP_SEED: the sample"""
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
ENV_SETS = [
    ("default", {}),
    ("min_reads_512", {"STRELKA_AMD_DEVICE_ENUM_MIN_READS": "512"}),
    ("min_reads_96", {"STRELKA_AMD_DEVICE_ENUM_MIN_READS": "96"}),
    ("window_16384", {"STRELKA_AMD_READ_WINDOW": "16384"}),
    ("window_32768", {"STRELKA_AMD_READ_WINDOW": "32768"}),
    ("staged_chain", {"SK_A5_FUSED": "0"}),
    ("min_reads_1", {"STRELKA_AMD_DEVICE_ENUM_MIN_READS": "1"}),
    ("min_reads_32", {"STRELKA_AMD_DEVICE_ENUM_MIN_READS": "32"}),
    ("gvcf_fast_off", {"STRELKA_AMD_GVCF_FAST": "0"}),
    ("gvcf_blocks_off", {"STRELKA_AMD_GVCF_BLOCKS": "0"}),
    ("early_init_off", {"STRELKA_AMD_EARLY_INIT": "0"}),
    ("broker", {"STRELKA_AMD_BROKER": "1"}),
    ("min_reads_1_three_waits", {"STRELKA_AMD_DEVICE_ENUM_MIN_READS": "1", "SK_ENUM_ONE_WAIT": "0"}),
]
if os.environ.get("SK_SWEEP_ONLY"):  # a comma-separated choice of the settings above
    ENV_SETS = [s for s in ENV_SETS if s[0] in os.environ["SK_SWEEP_ONLY"].split(",")]


def body(path):
    with open(path, "rb") as f:
        return [l for l in f.read().split(b"\n") if not (l.startswith(b"##cmdline=") or l.startswith(b"##startTime=") or l.startswith(b"##fileDate="))]


def main():
    out_json = sys.argv[1]
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 16000000
    seg_bp = int(sys.argv[3]) if len(sys.argv) > 3 else 2000000
    procs = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    depth, seed = float(os.environ.get("SK_SWEEP_DEPTH", "40")), int(os.environ.get("SK_SWEEP_SEED", "20260926"))  # (another sample than the bench's)
    d = farm.wgs_dataset(L, depth, seed)
    root = tempfile.mkdtemp(prefix="sk_sweep_")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_dummy_germline_models.py"), os.path.join(root, "models")], check=True)
    evs = (os.path.join(root, "models", "germlineSNVScoringModels.json"), os.path.join(root, "models", "germlineIndelScoringModels.json"))
    groups = [[s] for s in farm.chrom_intervals(["chrW"], {"chrW": L}, seg_bp)]
    drop_in = "starling2_" + os.environ.get("SK_E2E_VARIANT", "amd")

    def argv_fn(binary):
        def fn(index, regions, prefix, skip_header):
            return farm.germline_segment_argv(binary, prefix, [os.path.join(d, "wgs.bam")], regions, os.path.join(d, "wgs.fa"),
                                              chrom_depth=os.path.join(d, "chrom_depth.txt"), skip_header=skip_header, evs_models=evs)
        return fn
    ref = farm.run_farm(groups, argv_fn("starling2_ref"), os.path.join(root, "ref"), OUTPUTS, jobs=procs)
    report = {"bp": L, "segment_bp": seg_bp, "procs": procs, "depth": depth, "seed": seed, "ref_wall_s": ref.wall_s, "ref_process_seconds_sum": sum(ref.process_s), "runs": []}
    farm.run_farm([[(0, "chrW", 1, min(L, 50000), 0)]], argv_fn(drop_in), os.path.join(root, "warm"), OUTPUTS, jobs=1)
    for name, env in ENV_SETS:
        res = farm.run_farm(groups, argv_fn(drop_in), os.path.join(root, name), OUTPUTS, jobs=procs, env=dict(env, STRELKA_AMD_VERBOSE="1"))
        counters, hooks, gvcf = {}, {}, {}
        for tail in res.stderr_tails:
            m = re.search(r"strelka_amd adapter: (.*)", tail)
            if m:
                for kv in m.group(1).split():
                    k, v = kv.split("=")
                    counters[k] = counters.get(k, 0) + int(v)
            m = re.search(r"strelka_amd adapter gvcf: (.*)", tail)
            if m:
                for kv in m.group(1).split():
                    k, v = kv.split("=")
                    gvcf[k] = gvcf.get(k, 0) + int(v)
            m = re.search(r"strelka_amd adapter seconds: (.*)", tail)
            if m:
                for kv in m.group(1).split():
                    k, v = kv.split("=")
                    hooks[k] = round(hooks.get(k, 0.0) + float(v), 3)
        same = all(body(res.outputs[n]) == body(ref.outputs[n]) for n in OUTPUTS)
        row = {"name": name, "env": env, "wall_s": res.wall_s, "process_seconds_sum": sum(res.process_s), "identical": same,
               "speedup": ref.wall_s / res.wall_s, "realign_jobs": counters.get("realign_jobs"), "realign_job_reads": counters.get("realign_job_reads"),
               "enum_device_reads": counters.get("enum_device_reads"), "enum_host_instead": counters.get("enum_host_instead"),
               "enum_jobs": {k[len("enum_jobs_"):]: v for k, v in counters.items() if k.startswith("enum_jobs_")},
               "gvcf": gvcf,
               "device_share": (counters.get("enum_device_reads", 0) / max(1, counters.get("realign_job_reads", 1))), "hook_seconds": hooks}
        report["runs"].append(row)
        print(json.dumps(row), flush=True)
        with open(out_json, "w") as f:
            json.dump(report, f, indent=1)
    return 0 if all(r["identical"] for r in report["runs"]) else 1


if __name__ == "__main__":
    sys.exit(main())
