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
            e = {"STRELKA_AMD_VERBOSE": "1", "STRELKA_AMD_BROKER_TIMING": "1", "SK_PILEUP_PUSH_SECONDS": "1"}
            e.update(env)
            r = farm.run_farm(groups, argv_fn(binary), root, OUTPUTS, jobs=jobs, env=e)
            hooks = {}
            for tail in r.stderr_tails:
                m = re.search(r"strelka_amd adapter seconds: (.*)", tail)
                if m:
                    for kv in m.group(1).split():
                        k, v = kv.split("=")
                        hooks[k] = hooks.get(k, 0.0) + float(v)
                m = re.search(r"strelka_amd pileup push laps: (.*)", tail)
                if m:
                    for i, v in enumerate(m.group(1).split()):
                        hooks["push_lap%d" % i] = hooks.get("push_lap%d" % i, 0.0) + float(v)
                m = re.search(r"strelka_amd pileup push seconds: (.*)", tail)
                if m:
                    for kv in m.group(1).split():
                        k, v = kv.split("=")
                        hooks["push_" + k] = hooks.get("push_" + k, 0.0) + float(v)
                m = re.search(r"strelka_amd broker client: (.*)", tail)
                if m:
                    for kv in m.group(1).split():
                        k, v = kv.split("=")
                        hooks["broker_" + k] = max(hooks.get("broker_" + k, 0.0), float(v)) if k == "wait_max" else hooks.get("broker_" + k, 0.0) + float(v)
                m = re.search(r"strelka_amd adapter feed: .* seconds=(\S+) abi_seconds=(\S+)", tail)
                if m:
                    hooks["feed"] = hooks.get("feed", 0.0) + float(m.group(1))
                    hooks["feed_abi"] = hooks.get("feed_abi", 0.0) + float(m.group(2))
            return r.wall_s, sum(r.process_s), hooks, sum(r.user_s), sum(r.sys_s)
        finally:
            shutil.rmtree(root, ignore_errors=True)

    print("%d bp, %d segments of %d bp, %d usable cores" % (L, len(groups), seg, cores), flush=True)
    run("starling2_amd", cores, {"STRELKA_AMD_BROKER": "0"})  # warm
    for jobs in [int(x) for x in os.environ.get("SK_SHARING_REF_JOBS", str(cores)).split(",")]:
        w, ps, _, us, ss = run("starling2_ref", jobs, {})
        print("reference            jobs %2d: wall %.2f s, process seconds %.1f (user %.1f, sys %.1f)" % (jobs, w, ps, us, ss), flush=True)
    malloc_env = {"MALLOC_TRIM_THRESHOLD_": "2147483647", "MALLOC_TOP_PAD_": "268435456", "MALLOC_MMAP_THRESHOLD_": "1073741824"}
    extra = []
    if os.environ.get("SK_SHARING_SPIN"):
        extra = [("spin wait", {"STRELKA_AMD_SPIN_WAIT": "1"}), ("spin wait, no SDMA", {"STRELKA_AMD_SPIN_WAIT": "1", "HSA_ENABLE_SDMA": "0"})]
    for w in [x for x in os.environ.get("SK_SHARING_WINDOWS", "").split(",") if x]:
        extra.append(("read window %s" % w, {"STRELKA_AMD_READ_WINDOW": w}))
    configs = [("own contexts", {"STRELKA_AMD_BROKER": "0"})] + extra + ([("no SDMA", {"HSA_ENABLE_SDMA": "0"}), ("no SDMA, 2 HW queues", {"HSA_ENABLE_SDMA": "0", "GPU_MAX_HW_QUEUES": "2"})] if os.environ.get("SK_SHARING_SDMA") else [])
    if os.environ.get("SK_SHARING_BROKER"):  # the per-GPU broker (csrc/sk_rt.h) with 4 / 8 / 2 hardware queues
        configs += [("broker", {"STRELKA_AMD_BROKER": "1"})]
        if os.environ.get("SK_SHARING_BROKER_DMA"):
            configs.append(("broker, dma copies", {"STRELKA_AMD_BROKER": "1", "STRELKA_AMD_BROKER_COPY": "dma", "STRELKA_AMD_BROKER_SOCKET": "sk_share_dma"}))
        for q in [x for x in os.environ.get("SK_SHARING_BROKER_QUEUES", "").split(",") if x]:
            configs.append(("broker, %s queues" % q, {"STRELKA_AMD_BROKER": "1", "STRELKA_AMD_BROKER_QUEUES": q, "STRELKA_AMD_BROKER_SOCKET": "sk_share_q" + q}))
    # "name:K=V,K=V;name:..." -- broker variants, each with a server of its own (server-side knobs are read when the server starts)
    for i, spec in enumerate([x for x in os.environ.get("SK_SHARING_BROKER_VARIANTS", "").split(";") if x]):
        name, _, kvs = spec.partition(":")
        env = {"STRELKA_AMD_BROKER": "1", "STRELKA_AMD_BROKER_SOCKET": "sk_share_v%d" % i}
        env.update(dict(kv.split("=", 1) for kv in kvs.split(",") if kv))
        configs.append(("broker " + name, env))
    job_list = [int(x) for x in os.environ.get("SK_SHARING_JOBS", "%d,%d,%d" % (cores, cores * 3 // 2, cores * 2)).split(",")]
    for jobs in job_list:
        for label, env in configs:
            w, ps, hooks, us, ss = run("starling2_amd", jobs, env)
            print("adapter %-34s jobs %2d: wall %.2f s, process seconds %.1f (user %.1f, sys %.1f), init %.2f, abi seconds realign %.2f pileup %.2f feed %.2f indel %.2f haplotype %.2f (hooks %.2f / %.2f / %.2f)" %
                  (label, jobs, w, ps, us, ss, hooks.get("init", 0), hooks.get("realign_abi", 0), hooks.get("pileup_abi", 0), hooks.get("feed_abi", 0),
                   hooks.get("indel_abi", 0), hooks.get("haplotype_abi", 0), hooks.get("realign_hook", 0), hooks.get("pileup_hook", 0), hooks.get("feed", 0)), flush=True)
            if "push_pushes" in hooks:
                print("        pileup pushes: " + " ".join("%s=%.6g" % (k[5:], v) for k, v in sorted(hooks.items()) if k.startswith("push_")), flush=True)
            if "broker_waits" in hooks:
                print("        broker clients: " + " ".join("%s=%.6g" % (k[7:], v) for k, v in sorted(hooks.items()) if k.startswith("broker_")), flush=True)


if __name__ == "__main__":
    main()
