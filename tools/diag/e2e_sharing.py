"""how the drop-in's segment processes share one GPU: the farm of bench.py's end-to-end leg with different numbers of concurrent
processes, stage-window sizes and HIP queue settings; wall seconds and the adapter's hook timers summed over the processes.

usage: python tools/diag/e2e_sharing.py [LENGTH=4000000] [SEGMENT=250000]"""
import os
import re
import shutil
import sys
import tempfile

sys.path.insert(0, ".")
from strelka_amd import farm

OUTPUTS = ("variants.vcf", "genome.S1.vcf")


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
    seg = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
    d = farm.wgs_dataset(L)
    groups = [[s] for s in farm.chrom_intervals(["chrW"], {"chrW": L}, seg)]
    cores = len(farm.usable_cores())

    def argv_fn(binary):
        def fn(index, regions, prefix, skip_header):
            return farm.germline_segment_argv(binary, prefix, [os.path.join(d, "wgs.bam")], regions, os.path.join(d, "wgs.fa"),
                                              chrom_depth=os.path.join(d, "chrom_depth.txt"), skip_header=skip_header)
        return fn

    def run(binary, jobs, env):
        root = tempfile.mkdtemp(prefix="sk_share_")
        try:
            e = {"STRELKA_AMD_VERBOSE": "1"}
            e.update(env)
            r = farm.run_farm(groups, argv_fn(binary), root, OUTPUTS, jobs=jobs, env=e)
            hooks = {}
            for tail in r.stderr_tails:
                m = re.search(r"strelka_amd adapter seconds: (.*)", tail)
                if m:
                    for kv in m.group(1).split():
                        k, v = kv.split("=")
                        hooks[k] = hooks.get(k, 0.0) + float(v)
            return r.wall_s, sum(r.process_s), hooks
        finally:
            shutil.rmtree(root, ignore_errors=True)

    print("%d bp, %d segments of %d bp, %d usable cores" % (L, len(groups), seg, cores), flush=True)
    run("starling2_amd", 1, {})  # warm
    for jobs in (cores,):
        w, ps, _ = run("starling2_ref", jobs, {})
        print("reference            jobs %2d: wall %.2f s, process seconds %.1f" % (jobs, w, ps), flush=True)
    configs = [("default (32k/64k, 1 queue)", {}), ("4 HW queues", {"GPU_MAX_HW_QUEUES": "4"}),
               ("windows 2k/4k", {"STRELKA_AMD_READ_WINDOW": "2048", "STRELKA_AMD_SITE_WINDOW": "4096"}),
               ("windows 8k/16k", {"STRELKA_AMD_READ_WINDOW": "8192", "STRELKA_AMD_SITE_WINDOW": "16384"}),
               ("device enumeration always", {"SK_ENUMERATION": "2"}), ("host enumeration always", {"SK_ENUMERATION": "0"}),
               ("reference pileup", {"STRELKA_AMD_PILEUP": "0"}), ("reference feed", {"STRELKA_AMD_FEED": "0"})]
    for jobs in (cores, 1):
        for label, env in configs:
            if jobs != cores and not label.startswith("default"):
                continue
            w, ps, hooks = run("starling2_amd", jobs, env)
            print("adapter %-28s jobs %2d: wall %.2f s, process seconds %.1f, init %.2f, abi seconds realign %.2f pileup %.2f (hooks %.2f / %.2f)" %
                  (label, jobs, w, ps, hooks.get("init", 0), hooks.get("realign_abi", 0), hooks.get("pileup_abi", 0), hooks.get("realign_hook", 0),
                   hooks.get("pileup_hook", 0)), flush=True)


if __name__ == "__main__":
    main()
