#!/usr/bin/env python
"""Which segment of the end-to-end germline leg differs from the reference's output, and which switch makes the difference go away.

usage: e2e_parity_hunt.py <out dir> [bp] [segment bp] [procs]
Runs the unmodified reference once over bench.py's germline configuration (kept), the drop-in with the default settings twice (is the
difference reproducible?), then -- on the differing segments only -- the drop-in with one switch changed at a time.  Per run: the
segments whose variants.vcf / genome.S1.vcf differ and the first differing line of each; everything lands in <out dir>/hunt.json."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from strelka_amd import farm  # noqa: E402

OUTPUTS = ("variants.vcf", "genome.S1.vcf")


def body(path):
    with open(path, "rb") as f:
        return [l for l in f.read().split(b"\n") if not (l.startswith(b"##cmdline=") or l.startswith(b"##startTime=") or l.startswith(b"##fileDate="))]


def main():
    out_dir = sys.argv[1]
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 16000000
    seg_bp = int(sys.argv[3]) if len(sys.argv) > 3 else 2000000
    procs = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    os.makedirs(out_dir, exist_ok=True)
    d = farm.wgs_dataset(L)
    models = os.path.join(out_dir, "models")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_dummy_germline_models.py"), models], check=True)
    evs = (os.path.join(models, "germlineSNVScoringModels.json"), os.path.join(models, "germlineIndelScoringModels.json"))
    groups = [[s] for s in farm.chrom_intervals(["chrW"], {"chrW": L}, seg_bp)]
    drop_in = "starling2_" + os.environ.get("SK_E2E_VARIANT", "amd")

    def argv_fn(binary):
        def fn(index, regions, prefix, skip_header):
            return farm.germline_segment_argv(binary, prefix, [os.path.join(d, "wgs.bam")], regions, os.path.join(d, "wgs.fa"),
                                              chrom_depth=os.path.join(d, "chrom_depth.txt"), skip_header=skip_header, evs_models=evs)
        return fn

    report = {"bp": L, "segment_bp": seg_bp, "procs": procs, "runs": []}
    ref_dir = os.path.join(out_dir, "ref")
    farm.run_farm(groups, argv_fn("starling2_ref"), ref_dir, OUTPUTS, jobs=procs, join=False)

    def compare(run_dir, indices):
        bad = {}
        for i in indices:
            for n in OUTPUTS:
                got, want = body(os.path.join(run_dir, "seg%04d.%s" % (i, n))), body(os.path.join(ref_dir, "seg%04d.%s" % (i, n)))
                if got != want:
                    k = next((j for j, (x, y) in enumerate(zip(got, want)) if x != y), min(len(got), len(want)))
                    n_diff = sum(1 for x, y in zip(got, want) if x != y) + abs(len(got) - len(want))
                    bad.setdefault(i, {})[n] = {"line": k + 1, "lines_differing": n_diff,
                                                "drop_in": got[k].decode(errors="replace")[:500] if k < len(got) else None,
                                                "reference": want[k].decode(errors="replace")[:500] if k < len(want) else None}
        return bad

    def run(name, env, indices, jobs):
        run_dir = os.path.join(out_dir, name)
        sub = [groups[i] for i in indices]
        # (run_farm numbers the groups it is given from 0: map back)
        res = farm.run_farm(sub, lambda idx, regions, prefix, skip: argv_fn(drop_in)(indices[idx], regions, prefix, indices[idx] != 0),
                            run_dir + "_tmp", OUTPUTS, jobs=jobs, env=dict(env, STRELKA_AMD_VERBOSE="1"), join=False)
        os.makedirs(run_dir, exist_ok=True)
        for k, i in enumerate(indices):
            for n in OUTPUTS + ("stderr.txt",):
                os.replace(os.path.join(run_dir + "_tmp", "seg%04d.%s" % (k, n)), os.path.join(run_dir, "seg%04d.%s" % (i, n)))
        bad = compare(run_dir, indices)
        counters = [t.strip().split("\n")[-4:] for t in res.stderr_tails]
        report["runs"].append({"name": name, "env": env, "segments": list(indices), "jobs": jobs, "wall_s": res.wall_s,
                               "differing": {str(i): v for i, v in bad.items()}, "counters": counters if bad else None})
        with open(os.path.join(out_dir, "hunt.json"), "w") as f:
            json.dump(report, f, indent=1)
        print(name, "segments", list(indices), "->", "differing: %s" % sorted(bad) if bad else "identical", flush=True)
        return bad

    everything = list(range(len(groups)))
    bad = run("default_a", {}, everything, procs)
    bad2 = run("default_b", {}, everything, procs)
    suspects = sorted(set(bad) | set(bad2))
    if not suspects:
        print("no difference in two default runs")
        return 0
    for name, env, jobs in (("alone", {}, 1),
                            ("enumeration_host", {"SK_ENUMERATION": "0"}, procs),
                            ("haplotype_unbatched", {"STRELKA_AMD_HAPLOTYPE_BATCH": "0"}, procs),
                            ("columns_only", {"STRELKA_AMD_PILEUP_GENOTYPE": "0"}, procs),
                            ("reference_pileup", {"STRELKA_AMD_PILEUP": "0"}, procs),
                            ("feed_off", {"STRELKA_AMD_FEED": "0"}, procs)):
        run(name, env, suspects, min(jobs, len(suspects)))
    return 1


if __name__ == "__main__":
    sys.exit(main())
