"""Adapter settings against each other on one box, INTERLEAVED: the germline (or somatic) sample of the bench's end-to-end leg cut into one
segment per usable core, one caller process per segment through the device's broker, the settings taken in turn for `rounds` rounds --
a box's speed drifts by 10-15 % over minutes (profiles/r06_v40: the same build 13.5 s and 16.0 s a quarter of an hour apart), so runs
made one after the other say nothing about a 3 % effect; alternating runs do.  Prints wall seconds, the processes' wall, user and system
seconds per setting and round, then the medians.
    python tools/diag/e2e_ab.py [germline|somatic] [rounds=4] NAME:VAR=VAL,VAR=VAL NAME:...      (NAME: alone = the defaults)"""
import json
import os
import shutil
import statistics
import sys
import tempfile

sys.path.insert(0, ".")
from strelka_amd import farm


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "germline"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    settings = []
    for a in sys.argv[3:] or ["async:", "sync:STRELKA_AMD_PUSH_ASYNC=0"]:
        name, _, kv = a.partition(":")
        settings.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
    somatic = mode == "somatic"
    L = 16000000 if somatic else 64000000
    d = (farm.wgs_somatic_dataset if somatic else farm.wgs_dataset)(L, *((40.0, 110.0) if somatic else (40.0,)))
    cores = farm.usable_cores()
    P = int(os.environ.get("E2E_AB_PROCS", len(cores)))  # ($E2E_AB_PROCS=8 $E2E_AB_BROKER=0: eight callers with a GPU context each)
    cores = cores[:P]
    groups = [[s] for s in farm.chrom_intervals(["chrW"], {"chrW": L}, L // P)]
    drop_in = "strelka2_amd" if somatic else "starling2_amd"
    outputs = ["somatic.snvs.vcf", "somatic.indels.vcf"] if somatic else ["variants.vcf", "genome.S1.vcf"]

    evs_models = None
    if not somatic:  # (EVS on, as the workflow runs and as bench.py's leg does: stand-in models)
        import subprocess
        md = tempfile.mkdtemp(prefix="sk_models_")
        subprocess.run([sys.executable, "tools/make_dummy_germline_models.py", md], check=True)
        evs_models = (md + "/germlineSNVScoringModels.json", md + "/germlineIndelScoringModels.json")

    def argv_fn(index, regions, prefix, skip_header):
        if somatic:
            return farm.somatic_segment_argv(drop_in, prefix, os.path.join(d, "normal.bam"), os.path.join(d, "tumor.bam"), regions,
                                             os.path.join(d, "normal.fa"), chrom_depth=os.path.join(d, "chrom_depth.txt"),
                                             callable_regions=True, skip_header=skip_header)
        return farm.germline_segment_argv(drop_in, prefix, [os.path.join(d, "wgs.bam")], regions, os.path.join(d, "wgs.fa"),
                                          chrom_depth=os.path.join(d, "chrom_depth.txt"), skip_header=skip_header, evs_models=evs_models)

    root = tempfile.mkdtemp(prefix="e2e_ab_")
    base = {"STRELKA_AMD_BROKER": os.environ.get("E2E_AB_BROKER", "1")}
    farm.run_farm(groups, argv_fn, os.path.join(root, "warm"), outputs, n_gpus=1, jobs=P, env=base, pin_cores=cores)  # page cache, broker up
    shutil.rmtree(os.path.join(root, "warm"))
    res = {name: [] for name, _ in settings}
    digest = {}
    for r in range(rounds):
        order = settings if r % 2 == 0 else settings[::-1]
        for name, env in order:
            out = os.path.join(root, "%s_%d" % (name, r))
            fr = farm.run_farm(groups, argv_fn, out, outputs, n_gpus=1, jobs=P, env=dict(base, **env), pin_cores=cores)
            # (without the header lines that name the run: command line, start time, paths)
            body = b"".join(b"\n".join(l for l in open(os.path.join(out, o), "rb").read().split(b"\n") if not l.startswith(b"##")) for o in outputs)
            digest.setdefault(name, set()).add(hash(body))
            rec = {"wall_s": fr.wall_s, "process_s": sum(fr.process_s), "user_s": sum(fr.user_s), "sys_s": sum(getattr(fr, "sys_s", []) or [0.0])}
            res[name].append(rec)
            print("round %d %-12s wall %.2f s, processes %.1f s (user %.1f)" % (r, name, rec["wall_s"], rec["process_s"], rec["user_s"]), flush=True)
            shutil.rmtree(out)
    summary = {name: {k: statistics.median(x[k] for x in v) for k in ("wall_s", "process_s", "user_s")} for name, v in res.items()}
    same = len({frozenset(v) for v in digest.values()}) == 1 and all(len(v) == 1 for v in digest.values())
    print(json.dumps({"mode": mode, "callers": P, "rounds": rounds, "median": summary, "outputs_identical_across_settings": same}))
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
