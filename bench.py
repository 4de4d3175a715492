#!/usr/bin/env python
"""bench.py -- throughput of the MI355X-native Strelka2 hot path on the germline chr20-style synthetic workload
(BASELINE.json configs[1]): 150 bp reads x 64 candidate alignments (hot path A) and 40x pileup loci (hot path B).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One process per GPU.  The path shards by independent genome segments, so every rank scores its own,
equally sized batch ("weak" scaling) and there is NO data-path collective; torch.distributed is used only for the
barrier that brackets the timed region and the max-over-ranks of the elapsed time.

A "step" = one pass of hot path A over one resident batch of R reads x 64 candidate alignments (the headline metric,
cells/s = read bases x candidate alignments scored per second).  Hot path B (dependent error probabilities + diploid
genotype likelihoods per locus) is timed the same way in a second K-step region and reported as loci_per_s beside it.
Inputs are resident in HBM before any timed region starts.
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# algorithmic bytes per unit, SURVEY.md 8(d)
A_BYTES_PER_READ = 2 * 150 + 64 * (160 + 8)  # 11 052 B per 9 600 cells
B_BYTES_PER_CALL_DE = 6                      # sk_dependent_eprob: 2 B call in + 4 B de out
B_BYTES_PER_LOCUS_FIXED = 1 + 120            # ref base + results


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=1 << 20, help="reads per step per GPU (x64 candidates x150 bp)")
    ap.add_argument("--loci", type=int, default=1 << 26, help="germline loci per step per GPU (depth ~Poisson(40)); 2^26 ~ chr20")
    ap.add_argument("--feed-blocks", type=int, default=1 << 17, help="BGZF blocks per step of the feed leg (8f rank 4)")
    ap.add_argument("--realign-reads", type=int, default=1 << 16, help="reads per step of the whole-read leg (a1-a7)")
    ap.add_argument("--cpu-seconds-per-leg", type=float, default=5.0, help="seconds each CPU-baseline leg runs (all cores in parallel)")
    ap.add_argument("--unique-reads", type=int, default=1 << 14, help="distinct synthetic reads (tiled on device)")
    ap.add_argument("--unique-loci", type=int, default=1 << 20)
    ap.add_argument("--pileup-reads", type=int, default=1 << 20, help="reads per step per GPU for the pileup leg (row a8)")
    ap.add_argument("--somatic-loci", type=int, default=1 << 22, help="somatic loci per step per GPU (40x normal + 110x tumor)")
    ap.add_argument("--indels", type=int, default=1 << 18, help="indel loci per step per GPU for the indel legs (a11, a14)")
    ap.add_argument("--align-problems", type=int, default=4096, help="GlobalAligner problems per step (next row f2)")
    ap.add_argument("--a5-scenarios", type=int, default=12, help="scenarios (jobs) of the flatten + score leg")
    ap.add_argument("--a5-reads", type=int, default=1 << 16,
                    help="reads per job of the flatten + score leg (2^16 reads x ~88 candidate alignments = 5.8e6 alignments in one job; "
                         "SURVEY 8d asks for 2^14 .. 2^20 reads)")
    ap.add_argument("--a5-reps", type=int, default=3)
    ap.add_argument("--e2e-bp", type=int, default=64000000,
                    help="length of the WGS-like 40x sample of the end-to-end leg per GPU (0: skip the leg); the default is chr20's size "
                         "(BASELINE.json configs[1])")
    ap.add_argument("--e2e-segment-bp", type=int, default=12000000,
                    help="segment size of the end-to-end leg (one caller process per segment); the default is the workflow's: a genome is cut "
                         "into equal pieces no longer than scanSizeMb = 12 Mb (PY/strelkaSharedOptions.py:161), chr20 -> 6 segments")
    ap.add_argument("--only", default="", help="'a5' / 'feed' / 'feed_slice' / 'loci' / 'pileup' / 'somatic': run one kernel leg alone (the counter passes of tools/gpu_round.sh use it: "
                                               "per-kernel averages then belong to that leg's launches) and print a short line; 'e2e', "
                                               "'e2e_germline', 'e2e_somatic': the end-to-end legs alone (exit code 1 when the drop-in's outputs "
                                               "differ from the reference's)")
    ap.add_argument("--e2e-somatic-bp", type=int, default=16000000,
                    help="length of the WGS-like 110x / 40x tumour-normal pair of the somatic end-to-end leg per GPU (0: skip the leg); "
                         "a quarter of chr20 (BASELINE.json configs[2]) at 150x in all")
    ap.add_argument("--e2e-somatic-segment-bp", type=int, default=2000000, help="segment size of the somatic end-to-end leg")
    ap.add_argument("--e2e-max-procs-per-gpu", type=int, default=8,
                    help="caller processes that share one GPU in the end-to-end leg (beyond ~8 the device's scheduler time-slices them: "
                         "profiles/r03_v10_processes_per_gpu.txt, r03_v11_gpu_sharing_sdma.txt); the reference gets the same number of cores")
    ap.add_argument("--no-e2e-box", action="store_true", help="skip the e2e_box legs (the reference on all cores against the drop-in's best process count)")
    ap.add_argument("--realign-processes", type=int, default=8, help="caller processes of the multi-process whole-read leg (0: skip it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reads", type=int, default=1500, help="reads in the CPU-baseline sample (x64 candidates)")
    ap.add_argument("--cpu-loci", type=int, default=2000000, help="loci in the CPU-baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU time spent on the alignment-scoring sample")
    return ap.parse_args()


def cpu_baseline(args):
    """The CPU number the GPU line is put beside.  Where oracle/_ref travelled with the repo: the REFERENCE's
    own translation units, one single-threaded process per host core, each pinned, all running at once (oracle/ref_timing.py):
    kind "reference".  Otherwise the plain-C restatement on one core: kind "port".  This is the only place bench.py touches
    oracle/: as the thing the GPU number is put beside, never as part of the measured product path."""
    from oracle import pyoracle
    if pyoracle.ref_available():
        from oracle import ref_timing
        r = ref_timing.reference_baseline(args.cpu_seconds_per_leg)
        return {"value": r["cells_per_s"], "unit": "cells/s", "cores": r["cores"], "kind": "reference",
                "value_per_core": r["cells_per_s_per_core"],
                "loci_per_s": r["loci_per_s"], "loci_per_s_per_core": r["loci_per_s_per_core"],
                "somatic_loci_per_s": r["somatic_loci_per_s"], "somatic_loci_per_s_per_core": r["somatic_loci_per_s_per_core"],
                "realign_reads_per_s": r["reads_per_s"], "realign_reads_per_s_per_core": r["reads_per_s_per_core"],
                "realign_dense_reads_per_s": r["reads_dense_per_s"], "realign_dense_reads_per_s_per_core": r["reads_dense_per_s_per_core"],
                "one_process_alone": {"cells_per_s": r["cells_per_s_one_process"], "loci_per_s": r["loci_per_s_one_process"],
                                      "somatic_loci_per_s": r["somatic_loci_per_s_one_process"],
                                      "realign_reads_per_s": r["reads_per_s_one_process"],
                                      "realign_dense_reads_per_s": r["reads_dense_per_s_one_process"]},
                "sample": "the reference's own translation units (oracle/_ref), %d processes pinned to %d cores, %.0f s per leg, "
                          "inputs pre-materialised in the reference's structs, only its compute calls inside the clock: "
                          "scoreCandidateAlignment on 48 reads x 64 candidate alignments x 150 bp per process (cells/s); "
                          "realignAndScoreRead on the scenarios of the whole-read legs themselves (default_rng(4242), 24 scenarios x 12 "
                          "reads, up to 6 candidate indels around a read; realign_dense_*: up to 14), every process the same ones "
                          "(reads/s); adjust_joint_eprob + "
                          "position_snp_call_pprob_digt on 20000 loci depth~Poisson(40) per process (loci/s); "
                          "position_somatic_snv_call on 20000 loci 40x+110x per process (somatic loci/s); rates summed over the "
                          "processes, *_per_core = that sum / cores; one_process_alone = the same legs with a single process on the otherwise idle "
                          "host" % (r["cores"], r["cores"], r["seconds_per_leg"])}
    from strelka_amd import synth
    pyoracle.build(ref=False)
    rng = np.random.default_rng(1)
    cases = synth.align_cases_h64(args.cpu_reads, rng)
    m = pyoracle.MarshalledCases(cases)
    pyoracle.score_cases(m)  # warm
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < args.cpu_seconds:  # repeated passes over the (cache-resident) sample
        pyoracle.score_cases(m)
        reps += 1
    ta = time.perf_counter() - t0
    cells = reps * sum(len(c["read_code"]) * len(c["cals"]) for c in cases)
    pb = synth.pileups(args.cpu_loci, rng)
    t0 = time.perf_counter()
    de = pyoracle.adjust_joint_eprob(pb)
    pyoracle.site_digt_call(pb, de)
    tb = time.perf_counter() - t0
    return {"value": cells / ta, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "oracle/_ref absent: the C restatement on one core; %d reads x 64 candidate alignments x 150 bp, %d passes "
                      "through sko_score_cases (%.1f s); loci: %d loci depth~Poisson(40) through sko_adjust_joint_eprob+"
                      "sko_position_snp_call_pprob_digt (%.1f s)" % (len(cases), reps, ta, args.cpu_loci, tb),
            "loci_per_s": args.cpu_loci / tb}


def whole_read_leg(args, capi, synth, rng, enumeration=2, max_indels=6, reads=None, host_threads=1):
    """Rows a1-a7 end to end: reads -> sk_realign_job_add_reads (gate, normalisation) -> run (candidate-alignment enumeration,
    flattening, scoring; then selection and score_indels on the host) on the same scenario distribution the reference's
    realignAndScoreRead is timed on (oracle/ref_timing.py).  Host buffers in and out, PCIe included: this is the path the adapter
    calls.  `enumeration`: 2 = search + flattening on the device (csrc/read_enumerate.hip), 0 = on the host (round 1's path)."""
    scenarios = synth.realign_scenarios(24, rng, reads_per=12, max_indels=max_indels)
    jobs = []
    total = 0
    rep = max(1, (reads or args.realign_reads) // (24 * 12))
    for sc in scenarios:
        keep, inputs = [], []
        for rd in sc["reads"]:
            code = np.ascontiguousarray(rd["code"], np.uint8)
            qual = np.ascontiguousarray(rd["qual"], np.uint8)
            segs = (capi.PathSeg * max(len(rd["path"]), 1))(*[capi.PathSeg(t, l) for t, l in rd["path"]])
            obs = (C.c_int32 * max(len(rd["observed"]), 1))(*rd["observed"])
            keep.append((code, qual, segs, obs))
            inputs.append(capi.ReadInput(capi._p(code), capi._p(qual), len(code), rd["pos"], len(rd["path"]), segs, int(rd["is_fwd"]),
                                         rd["map_level"], 0, rd["realign_range"][0], rd["realign_range"][1], len(rd["observed"]), obs))
        probe = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                     min_read_bp_flank=sc["min_read_bp_flank"], enumeration=0))
        probe.set_reference(sc["ref_seq"], sc["ref_offset"])
        probe.set_indels(sc["indels"])
        ok = [r for r in inputs if capi.lib().sk_realign_job_add_read(probe._j, C.byref(r)) >= 0]
        del probe
        if not ok:
            continue
        job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"],
                                                   min_read_bp_flank=sc["min_read_bp_flank"], enumeration=enumeration, host_threads=host_threads))
        job.set_reference(sc["ref_seq"], sc["ref_offset"])
        job.set_indels(sc["indels"])
        n = len(ok) * rep
        arr = (capi.ReadInput * n)(*[ok[i % len(ok)] for i in range(n)])
        jobs.append((job, arr, n, keep))
        total += n

    def step():
        for job, arr, n, _ in jobs:
            job.clear_reads()
            if capi.lib().sk_realign_job_add_reads(job._j, arr, n) < 0:
                raise RuntimeError("sk_realign_job_add_reads failed")
            job.run()
    cals = 0
    step()
    for job, _, _, _ in jobs:
        cals += job.batch().n_cals
    return step, total, cals


E2E_OUTPUTS = ("variants.vcf", "genome.S1.vcf")


E2E_SOMATIC_OUTPUTS = ("somatic.snvs.vcf", "somatic.indels.vcf", "somatic.callable.regions.bed")


def e2e_leg(args, rank, world, local_rank, barrier, max_over_ranks, with_reference, mode="germline"):
    """BASELINE.json's second metric, "40x WGS germline wall-clock" (mode "germline", configs[1]) and its somatic counterpart (mode
    "somatic", configs[2]: tumour 110x / normal 40x): a WGS-like sample (tools/make_wgs_bam.py, made before the clock starts) cut
    into segments as the reference's workflow cuts a genome, one caller process per segment with the command line the workflow builds
    (strelka_amd/farm.py), as many processes at a time as this rank has host cores (at most --e2e-max-procs-per-gpu): the drop-in
    (`starling2_amd` / `strelka2_amd`: the reference's own program with its hot-path call sites routed through libstrelka_amd.so) on
    this rank's GPU, and -- rank 0 of a 1-GPU run -- the unmodified reference (`*_ref`, kind "reference") on the same cores, outputs
    compared byte for byte.  Weak scaling: every rank calls the whole sample on its own GPU with cores/N processes."""
    import re
    import shutil
    import tempfile
    from strelka_amd import farm
    somatic = (mode == "somatic")
    L = args.e2e_somatic_bp if somatic else args.e2e_bp
    seg_bp = args.e2e_somatic_segment_bp if somatic else args.e2e_segment_bp
    outputs = E2E_SOMATIC_OUTPUTS if somatic else E2E_OUTPUTS
    program = "strelka2" if somatic else "starling2"
    # (tests/test_bench_e2e.py runs the leg's plumbing on the CPU double)
    drop_in = program + "_" + os.environ.get("SK_E2E_VARIANT", "amd")
    if not (os.path.exists(os.path.join(farm.BIN_DIR, drop_in)) and os.path.exists(os.path.join(farm.BIN_DIR, "samtools"))):
        return {"skipped": "oracle/_ref/bin/%s (adapter/Makefile, needs the reference tree at build time) did not travel" % drop_in}
    dataset = farm.wgs_somatic_dataset if somatic else farm.wgs_dataset
    if rank == 0:
        dataset(L)
    barrier()
    d = dataset(L)
    cores = farm.usable_cores()
    jobs = max(1, min(len(cores) // world, args.e2e_max_procs_per_gpu))
    groups = [[s] for s in farm.chrom_intervals(["chrW"], {"chrW": L}, seg_bp)]
    root = tempfile.mkdtemp(prefix="sk_e2e_r%d_" % rank)

    def argv_fn(binary):
        def fn(index, regions, prefix, skip_header):
            if somatic:
                return farm.somatic_segment_argv(binary, prefix, os.path.join(d, "normal.bam"), os.path.join(d, "tumor.bam"), regions,
                                                 os.path.join(d, "normal.fa"), chrom_depth=os.path.join(d, "chrom_depth.txt"),
                                                 callable_regions=True, skip_header=skip_header)
            return farm.germline_segment_argv(binary, prefix, [os.path.join(d, "wgs.bam")], regions, os.path.join(d, "wgs.fa"),
                                              chrom_depth=os.path.join(d, "chrom_depth.txt"), skip_header=skip_header, evs_models=evs_models)
        return fn
    evs_models = None
    if not somatic:
        # the germline workflow runs with EVS on by default (strelkaSharedOptions.py:173): scoring models on the command line, the EVS
        # accumulators in the pileup.  The reference tree does not carry its germline models; small stand-ins are written here.
        import subprocess
        import sys
        subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "make_dummy_germline_models.py"),
                        os.path.join(root, "models")], check=True)
        evs_models = (os.path.join(root, "models", "germlineSNVScoringModels.json"), os.path.join(root, "models", "germlineIndelScoringModels.json"))
    try:
        warm = [[(0, "chrW", 1, min(L, 50000), 0)]]
        farm.run_farm(warm, argv_fn(drop_in), os.path.join(root, "warm"), outputs, n_gpus=1, jobs=1, device_offset=local_rank,
                      env=({"STRELKA_AMD_BROKER": "1", "STRELKA_AMD_BROKER_SOCKET": "sk_bench_rank_%d" % os.getpid(), "STRELKA_AMD_BROKER_IDLE_S": "60"}
                           if world > 1 else {"STRELKA_AMD_BROKER": "0"}))  # page the binary and the GPU runtime in (N > 1: start the device's broker)
        barrier()
        t0 = time.perf_counter()
        # (one rank: every caller process holds a GPU context of its own -- at most eight, the leg's equal-process-count comparison; the
        # adapter's own default is the broker, measured by e2e_box)
        amd_env = {"STRELKA_AMD_VERBOSE": "1", "STRELKA_AMD_BROKER": "0"}
        if world > 1:
            # N > 1: this rank's own process holds a GPU context (torch, RCCL); its caller processes are clients of the device's broker
            # (one more context per device, however many callers: strelka_amd/csrc/sk_rt.h) instead of a context each
            amd_env.update({"STRELKA_AMD_BROKER": "1", "STRELKA_AMD_BROKER_SOCKET": "sk_bench_rank_%d" % os.getpid(), "STRELKA_AMD_BROKER_IDLE_S": "60"})
        amd = farm.run_farm(groups, argv_fn(drop_in), os.path.join(root, "amd"), outputs, n_gpus=1, jobs=jobs,
                            device_offset=local_rank, env=amd_env)
        barrier()
        amd_wall = max_over_ranks(time.perf_counter() - t0)
        hooks = {}
        counters = {}
        for tail in amd.stderr_tails:
            m = re.search(r"strelka_amd adapter seconds: (.*)", tail)
            if m:
                for kv in m.group(1).split():
                    k, v = kv.split("=")
                    hooks[k] = hooks.get(k, 0.0) + float(v)
            # what went through the C-ABI (adapter/sk_adapter_common.cpp, sk_adapter_feed.cpp: STRELKA_AMD_VERBOSE=1), summed over the
            # segment processes: the identical bytes below are the routed path's only if these say so
            for pat in (r"strelka_amd adapter: (.*)", r"strelka_amd adapter pileup: (.*)", r"strelka_amd adapter feed: (.*)", r"strelka_amd adapter gvcf: (.*)"):
                m = re.search(pat, tail)
                if m:
                    for kv in m.group(1).split():
                        if "=" in kv:
                            k, v = kv.split("=", 1)
                            if re.fullmatch(r"-?\d+", v) and k not in ("read_window", "site_window"):
                                counters[k] = counters.get(k, 0) + int(v)
        depth = 150.0 if somatic else 40.0
        what = ("tumour / normal pair, %d bp at 110x / 40x" if somatic else "germline sample, %d bp at 40x") % L
        flags = ("somatic workflow's command line (EVS scoring models, --somatic-callable-regions-file, --strelka-chrom-depth-file ...)"
                 if somatic else "workflow's WGS command line (--chrom-depth-file, --gvcf-skip-header, EVS on as by default: "
                                 "--snv-scoring-model-file / --indel-scoring-model-file with stand-in models, "
                                 "tools/make_dummy_germline_models.py ...)")
        out = {"workload": "WGS-like synthetic %s, 150 bp reads (tools/make_wgs_bam.py), %d segments of %d bp, one caller process per segment "
                           "with the %s" % (what, len(groups), seg_bp, flags),
               "bp": L * world, "reads": int(L * depth / 150) * world, "segments": len(groups) * world,
               "amd_wall_s": amd_wall, "amd_procs": jobs * world, "amd_procs_per_gpu": jobs, "host_cores": len(cores),
               "cores_used": jobs * world,
               "procs_note": "one caller process per segment and core, each with a GPU context of its own (at most --e2e-max-procs-per-gpu, 8: the "
                             "device runs eight processes side by side); the reference leg runs with the same number of processes.  More "
                             "callers than eight go through the device's broker: the e2e_box leg (and this leg when --gpus > 1).",
               "bp_per_s": L * world / amd_wall, "process_seconds_sum": sum(amd.process_s),
               "hook_seconds": {k: round(v, 4) for k, v in hooks.items()},
               "counters": counters,
               "counters_note": "summed over the segment processes: enum_device_reads / enum_host_instead = reads whose candidate alignments the "
                                "device search listed / reads the device turned down and the host statement listed instead; "
                                "normalize_declined = alignments handed back to the reference's own normalizeAlignment (must be 0)",
               "hook_seconds_note": "summed over this rank's segment processes: wall seconds inside the adapter's hooks (*_hook) of which inside "
                                    "the C-ABI (*_abi); the rest of process_seconds_sum is the reference's own host code (BAM records, read "
                                    "buffer, active regions, locus objects, VCF text)",
               "host_note": "the drop-in is the reference's own program with the routed call sites (adapter/apply_hooks.py) and nothing else "
                            "changed: the host code beside those sites is the reference's, built with the same release flags (-O3)"}
        if with_reference:
            t0 = time.perf_counter()
            ref = farm.run_farm(groups, argv_fn(program + "_ref"), os.path.join(root, "ref"), outputs, jobs=jobs)
            ref_wall = time.perf_counter() - t0

            def body(path):
                with open(path, "rb") as f:
                    return [l for l in f.read().split(b"\n") if not (l.startswith(b"##cmdline=") or l.startswith(b"##startTime=") or l.startswith(b"##fileDate="))]
            identical, first_difference = True, None
            for n in outputs:
                got, want = body(amd.outputs[n]), body(ref.outputs[n])
                if got != want:
                    identical = False
                    k = next((i for i, (x, y) in enumerate(zip(got, want)) if x != y), min(len(got), len(want)))
                    first_difference = {"file": n, "line": k + 1, "lines_drop_in": len(got), "lines_reference": len(want),
                                        "drop_in": got[k].decode(errors="replace")[:400] if k < len(got) else None,
                                        "reference": want[k].decode(errors="replace")[:400] if k < len(want) else None}
                    keep = os.environ.get("SK_E2E_KEEP_DIR")
                    if keep:  # (diagnosis: the two joined files, for a diff after the run)
                        os.makedirs(keep, exist_ok=True)
                        shutil.copy(amd.outputs[n], os.path.join(keep, mode + "_drop_in_" + n))
                        shutil.copy(ref.outputs[n], os.path.join(keep, mode + "_reference_" + n))
                    break
            out.update({"ref_wall_s": ref_wall, "ref_cores": jobs, "ref_process_seconds_sum": sum(ref.process_s), "speedup": ref_wall / amd_wall,
                        "identical": identical, "first_difference": first_difference,
                        "variant_records": sum(1 for n in outputs if n.endswith(".vcf") and not n.startswith("genome")
                                               for l in body(ref.outputs[n]) if l and not l.startswith(b"#"))})
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def _vcf_body(path):
    with open(path, "rb") as f:
        return [l for l in f.read().split(b"\n") if not (l.startswith(b"##cmdline=") or l.startswith(b"##startTime=") or l.startswith(b"##fileDate="))]


def e2e_box_leg(args, local_rank, mode="germline"):
    """SURVEY.md 8(d)'s end-to-end figure as it is defined there: the reference with `-j P`, P = the box's usable cores (the workflow runs
    one caller process per core, PY/strelkaSharedOptions.py:153-161), against the drop-in at the process count that suits IT -- the same
    sample as the `e2e` leg, cut into max(P, 16) segments so that every core has one.  The drop-in runs with 8, 12 and 16 (at most P)
    caller processes as clients of the GPU's broker ($STRELKA_AMD_BROKER=1: one GPU context however many callers, strelka_amd/csrc/sk_rt.h)
    and, for comparison, with 8 processes that each hold a context of their own (what `e2e` measures; the device runs eight such processes
    side by side).  Every run is compared byte for byte with the reference's output; `speedup` = reference at -j P / the fastest drop-in run."""
    import re
    import shutil
    import subprocess
    import tempfile
    from strelka_amd import farm
    somatic = (mode == "somatic")
    L = args.e2e_somatic_bp if somatic else args.e2e_bp
    outputs = E2E_SOMATIC_OUTPUTS if somatic else E2E_OUTPUTS
    program = "strelka2" if somatic else "starling2"
    drop_in = program + "_" + os.environ.get("SK_E2E_VARIANT", "amd")
    if not (os.path.exists(os.path.join(farm.BIN_DIR, drop_in)) and os.path.exists(os.path.join(farm.BIN_DIR, "samtools"))):
        return {"skipped": "oracle/_ref/bin/%s (adapter/Makefile, needs the reference tree at build time) did not travel" % drop_in}
    d = (farm.wgs_somatic_dataset if somatic else farm.wgs_dataset)(L)
    cores = farm.usable_cores()
    P = len(cores)
    n_seg = max(P, 16)
    seg_bp = -(-L // n_seg)
    groups = [[s] for s in farm.chrom_intervals(["chrW"], {"chrW": L}, seg_bp)]
    root = tempfile.mkdtemp(prefix="sk_e2e_box_")
    evs_models = None
    if not somatic:
        subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "make_dummy_germline_models.py"),
                        os.path.join(root, "models")], check=True)
        evs_models = (os.path.join(root, "models", "germlineSNVScoringModels.json"), os.path.join(root, "models", "germlineIndelScoringModels.json"))

    def argv_fn(binary):
        def fn(index, regions, prefix, skip_header):
            if somatic:
                return farm.somatic_segment_argv(binary, prefix, os.path.join(d, "normal.bam"), os.path.join(d, "tumor.bam"), regions,
                                                 os.path.join(d, "normal.fa"), chrom_depth=os.path.join(d, "chrom_depth.txt"),
                                                 callable_regions=True, skip_header=skip_header)
            return farm.germline_segment_argv(binary, prefix, [os.path.join(d, "wgs.bam")], regions, os.path.join(d, "wgs.fa"),
                                              chrom_depth=os.path.join(d, "chrom_depth.txt"), skip_header=skip_header, evs_models=evs_models)
        return fn
    broker_env = {"STRELKA_AMD_BROKER": "1", "STRELKA_AMD_BROKER_SOCKET": "sk_bench_%d" % os.getpid(), "STRELKA_AMD_BROKER_IDLE_S": "60"}
    try:
        ref = farm.run_farm(groups, argv_fn(program + "_ref"), os.path.join(root, "ref"), outputs, jobs=P)
        want = {n: _vcf_body(ref.outputs[n]) for n in outputs}
        # The broker is a service: in a whole-genome run (~260 segments) every caller process but the first wave's finds it running, with
        # the buffers of the callers that have left in its pool.  The clock starts in that state: one wave of P short callers first (50 kb
        # each; not timed).  What the very first wave pays instead is measured once and reported (`first_wave_wall_s`: the server's
        # start-up and ~0.7 GB of fresh device memory per caller, all callers at once).
        cold_env = dict(broker_env, STRELKA_AMD_BROKER_SOCKET=broker_env["STRELKA_AMD_BROKER_SOCKET"] + "_cold")
        first_wave = farm.run_farm(groups, argv_fn(drop_in), os.path.join(root, "first_wave"), outputs, n_gpus=1, jobs=min(16, P), device_offset=local_rank, env=cold_env)
        first_wave_same = all(_vcf_body(first_wave.outputs[n]) == want[n] for n in outputs)
        shutil.rmtree(os.path.join(root, "first_wave"), ignore_errors=True)
        warm = [[(0, "chrW", 1 + i * 50000, min(L, (i + 1) * 50000), 0)] for i in range(P) if i * 50000 < L]
        farm.run_farm(warm, argv_fn(drop_in), os.path.join(root, "warm"), outputs, n_gpus=1, jobs=P, device_offset=local_rank, env=broker_env)
        runs = []
        plan = [("broker", j, broker_env) for j in sorted({min(8, P), min(12, P), min(16, P)}, reverse=True)] + [("own_context", min(8, P), {"STRELKA_AMD_BROKER": "0"})]
        for kind, jobs, env in plan:
            res = farm.run_farm(groups, argv_fn(drop_in), os.path.join(root, "%s_%d" % (kind, jobs)), outputs, n_gpus=1, jobs=jobs,
                                device_offset=local_rank, env=dict(env, STRELKA_AMD_VERBOSE="1"))
            hooks = {}
            for tail in res.stderr_tails:
                m = re.search(r"strelka_amd adapter seconds: (.*)", tail)
                if m:
                    for kv in m.group(1).split():
                        k, v = kv.split("=")
                        hooks[k] = hooks.get(k, 0.0) + float(v)
            same = all(_vcf_body(res.outputs[n]) == want[n] for n in outputs)
            runs.append({"callers": kind, "procs": jobs, "wall_s": res.wall_s, "process_seconds_sum": sum(res.process_s), "identical": same,
                         "speedup_vs_reference_all_cores": ref.wall_s / res.wall_s,
                         "abi_seconds": round(sum(v for k, v in hooks.items() if k.endswith("_abi")), 3), "init_seconds": round(hooks.get("init", 0.0), 3)})
            shutil.rmtree(os.path.join(root, "%s_%d" % (kind, jobs)), ignore_errors=True)
        best = min(runs, key=lambda r: r["wall_s"])
        return {"workload": "the %s sample of the e2e leg (%d bp), cut into %d segments of %d bp: one per usable core" % (mode, L, len(groups), seg_bp),
                "bp": L, "segments": len(groups), "host_cores": P,
                "ref_cores": P, "ref_wall_s": ref.wall_s, "ref_process_seconds_sum": sum(ref.process_s),
                "runs": runs, "best": {"callers": best["callers"], "procs": best["procs"]}, "amd_wall_s": best["wall_s"],
                "first_wave_wall_s": first_wave.wall_s, "first_wave_procs": min(16, P), "first_wave_identical": first_wave_same,
                "first_wave_speedup": ref.wall_s / first_wave.wall_s,
                "speedup": ref.wall_s / best["wall_s"], "identical": all(r["identical"] for r in runs) and first_wave_same,
                "note": "reference: the unmodified program at -j P (P = usable cores), SURVEY.md 8(d); drop-in: the fastest of `runs`, each "
                        "byte-compared with the reference's output, with the GPU's broker already serving (as every wave of a whole-genome run but the "
                        "first finds it); first_wave_*: the same farm against a broker that has to be started and has nothing in its pool.  "
                        "`e2e.speedup` beside this is the equal-process-count ratio."}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def realign_processes_leg(args, n_procs):
    """The whole read path (rows a1-a7, device legs) driven the way the reference is driven for cpu_baseline: one single-threaded caller
    process per core, all at once -- here n_procs processes sharing the one GPU (at most 8: beyond that the driver time-slices them).
    Every process runs `bench.py --only realign` (the same scenarios, the same jobs as the one-process legs); the timed regions are
    started together through ready / go files; rate = all reads / (last end - first start).  Runs before this process has a GPU
    context of its own (see the end-to-end legs)."""
    import shutil
    import subprocess
    import tempfile
    sync = tempfile.mkdtemp(prefix="sk_bench_sync_")
    try:
        env = dict(os.environ, SK_BENCH_SYNC_DIR=sync)
        steps = max(3, args.steps // 4)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--only", "realign", "--steps", str(steps), "--realign-reads", str(args.realign_reads)],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for _ in range(n_procs)]
        for name in ("sparse", "dense"):
            deadline = time.time() + 600
            while len(glob.glob(os.path.join(sync, "ready_%s_*" % name))) < n_procs:
                if time.time() > deadline or any(p.poll() not in (None, 0) for p in procs):
                    for p in procs:
                        p.kill()
                    return {"skipped": "a worker of the multi-process whole-read leg failed: " + b" ".join(p.stderr.read()[-300:] for p in procs if p.poll()).decode(errors="replace")}
                time.sleep(0.01)
            open(os.path.join(sync, "go_" + name), "w").close()
        outs = []
        for p in procs:
            o, e = p.communicate(timeout=900)
            if p.returncode != 0:
                return {"skipped": "worker exit %d: %s" % (p.returncode, e.decode(errors="replace")[-300:])}
            outs.append(json.loads(o.decode().strip().splitlines()[-1])["legs"])
        res = {"processes": n_procs, "steps": steps}
        for name in ("sparse", "dense"):
            reads = sum(o[name]["reads"] for o in outs)
            span = max(o[name]["t1"] for o in outs) - min(o[name]["t0"] for o in outs)
            res["realign%s_reads_per_s" % ("" if name == "sparse" else "_dense")] = reads / span
            res["%s_overlap" % name] = min(o[name]["t1"] for o in outs) - max(o[name]["t0"] for o in outs) > 0
        return res
    finally:
        shutil.rmtree(sync, ignore_errors=True)


def a5_leg(args, capi, synth):
    """Row a5 measured as ONE function, like the reference's (scoreCandidateAlignment, starling_read_align_score.cpp:261-499):
    from the candidate alignments as the search left them on the device (position, path, indel indices: PCal) to one double each --
    the haplotype bytes an alignment faces, the walk of its path, every base comparison and the sums: kernel F5 (flatten_score_kernel;
    with $SK_A5_FUSED=0 the staged chain F1-F3 + A1c it replaced).  150 bp reads
    over a window with 5-7 candidate indels (the densest of a dozen seeded scenarios); the job is enumerated on the device once,
    outside the clock; a step = sk_realign_job_rescore(1): flattening + scoring of the resident records, timed by stream events.
    -> (step function, per-step meta, list the step appends its event milliseconds to)"""
    rng = np.random.default_rng(4242)
    scenarios = synth.realign_scenarios(args.a5_scenarios, rng, reads_per=16, max_indels=7, min_indels=5, read_len=(150, 151), window=(330, 420),
                                        haplotyping_rate=0.0)
    best = None
    for sc in scenarios:
        keep, inputs = [], []
        for rd in sc["reads"]:
            code = np.ascontiguousarray(rd["code"], np.uint8)
            qual = np.ascontiguousarray(rd["qual"], np.uint8)
            segs = (capi.PathSeg * max(len(rd["path"]), 1))(*[capi.PathSeg(t, l) for t, l in rd["path"]])
            obs = (C.c_int32 * max(len(rd["observed"]), 1))(*rd["observed"])
            keep.append((code, qual, segs, obs))
            inputs.append(capi.ReadInput(capi._p(code), capi._p(qual), len(code), rd["pos"], len(rd["path"]), segs, int(rd["is_fwd"]),
                                         rd["map_level"], 0, rd["realign_range"][0], rd["realign_range"][1], len(rd["observed"]), obs))
        probe = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=0, min_read_bp_flank=sc["min_read_bp_flank"], enumeration=0))
        probe.set_reference(sc["ref_seq"], sc["ref_offset"])
        probe.set_indels(sc["indels"])
        ok = [r for r in inputs if capi.lib().sk_realign_job_add_read(probe._j, C.byref(r)) >= 0]
        if ok:
            probe.run()
            cals = probe.batch().n_cals / len(ok)
            if best is None or cals > best[0]:
                best = (cals, sc, ok, keep)
        del probe
    if best is None:
        raise RuntimeError("a5 leg: no scenario passed the gate")
    _, sc, ok, keep = best
    job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=0, min_read_bp_flank=sc["min_read_bp_flank"], enumeration=2))
    job.set_reference(sc["ref_seq"], sc["ref_offset"])
    job.set_indels(sc["indels"])
    n = max(len(ok), args.a5_reads // len(ok) * len(ok))
    arr = (capi.ReadInput * n)(*[ok[i % len(ok)] for i in range(n)])
    if capi.lib().sk_realign_job_add_reads(job._j, arr, n) < 0:
        raise RuntimeError("sk_realign_job_add_reads failed")
    job.run()
    ms, nr, nc, cells = C.c_float(0), C.c_int32(0), C.c_int32(0), C.c_int64(0)
    event_ms = []

    def step():
        if capi.lib().sk_realign_job_rescore(1, C.byref(ms), C.byref(nr), C.byref(nc), C.byref(cells)) != 0:
            raise RuntimeError("sk_realign_job_rescore: " + capi.last_error())
        event_ms.append(ms.value)
    step()
    del event_ms[:]
    n_core, n_dev, n_fall = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    capi.lib().sk_realign_job_enumeration_counts(job._j, C.byref(n_core), C.byref(n_dev), C.byref(n_fall))
    meta = {"reads": nr.value, "candidate_alignments": nc.value, "candidate_alignments_per_read": nc.value / max(1, nr.value), "cells": cells.value,
            "reads_enumerated_on_device": n_dev.value, "reads_handed_back_to_host": n_fall.value,
            "algorithmic_bytes": 2 * 150 * nr.value + (160 + 8) * nc.value, "_keep": (job, arr, keep)}
    return step, meta, event_ms


def pmc_traffic(args):
    """HBM bytes per launch from the newest committed PMC summary (profiles/*_pmc_traffic.json: separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes over this same command, tools/gpu_round.sh) -- used only when that run had the same
    per-launch workload as this one; PMC counters cannot be collected from inside the timed run itself."""
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")))
    if not files:
        return {}
    with open(files[-1]) as f:
        d = json.load(f)
    w = d.get("workload", {})
    if (w.get("reads_per_step_per_gpu"), w.get("loci_per_step_per_gpu"), w.get("unique_reads"), w.get("unique_loci")) != \
            (args.reads, args.loci, args.unique_reads, args.unique_loci):
        return {}
    out = {k: v["hbm_bytes_per_launch"] for k, v in d.get("kernels", {}).items()}
    a5 = d.get("a5_only")
    if a5 and a5.get("workload", {}).get("a5_scenarios") == args.a5_scenarios and a5.get("workload", {}).get("a5_reads") == args.a5_reads:
        out["__a5_step__"] = a5["hbm_bytes_per_step"]
    # where the numbers come from: every roofline object names the file, and the commit the counter passes ran at (tools/pmc_traffic.py
    # records it; older files only carry their visit tags in `note`)
    out["__source__"] = {"file": "profiles/" + os.path.basename(files[-1]), "commit": d.get("commit"), "note": d.get("note")}
    out["__bound__"] = d.get("bound") or {}   # per kernel: which unit binds it (SQ counter passes of the same visit, tools/pmc_traffic.py)
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    # ---- end to end FIRST when this is the only rank: the caller processes of these legs share the GPU eight at a time, and eight is
    # all the device runs side by side -- a ninth process with a queue on it (this one, once it has initialised HIP) puts all of them
    # under the driver's time-slicing (1.46x instead of 1.6x on the germline leg, profiles/r03_v20_bench.json vs r03_v19).  So the
    # legs run before this process creates its context.
    e2e = e2e_somatic = e2e_box = e2e_somatic_box = None
    if world == 1 and (not args.only or args.only.startswith("e2e")):
        if args.e2e_bp > 0 and args.only in ("", "e2e", "e2e_germline"):
            e2e = e2e_leg(args, 0, 1, local_rank, lambda: None, lambda v: v, with_reference=not args.no_cpu_baseline)
        if args.e2e_somatic_bp > 0 and args.only in ("", "e2e", "e2e_somatic"):
            e2e_somatic = e2e_leg(args, 0, 1, local_rank, lambda: None, lambda v: v, with_reference=not args.no_cpu_baseline, mode="somatic")
        # the box ratio: the reference on ALL cores against the drop-in's best process count
        if not args.no_cpu_baseline and not args.no_e2e_box:
            if args.e2e_bp > 0 and args.only in ("", "e2e", "e2e_box", "e2e_germline_box"):
                e2e_box = e2e_box_leg(args, local_rank)
            if args.e2e_somatic_bp > 0 and args.only in ("", "e2e", "e2e_box", "e2e_somatic_box"):
                e2e_somatic_box = e2e_box_leg(args, local_rank, mode="somatic")
    realign_processes = None
    if world == 1 and not args.only and args.realign_processes > 0:
        from strelka_amd import farm as _farm
        realign_processes = realign_processes_leg(args, max(1, min(args.realign_processes, len(_farm.usable_cores()), args.e2e_max_procs_per_gpu)))
    e2e_legs = {"e2e": e2e, "e2e_somatic": e2e_somatic, "e2e_box": e2e_box, "e2e_somatic_box": e2e_somatic_box}
    e2e_failed = [name for name, leg in e2e_legs.items() if leg and leg.get("identical") is False]
    for name in e2e_failed:
        print("bench.py: the %s leg's outputs are NOT identical to the reference's: %s" % (name, json.dumps(e2e_legs[name].get("first_difference") or e2e_legs[name].get("runs"))),
              file=sys.stderr, flush=True)
    if args.only.startswith("e2e"):
        print(json.dumps(dict(e2e_legs, only=args.only)), flush=True)
        sys.exit(1 if e2e_failed else 0)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank

    from strelka_amd import capi, device, shard, synth
    capi.init(local_rank)

    if args.only == "a5":
        a5_step, a5_meta, a5_event_ms = a5_leg(args, capi, synth)
        for _ in range(args.warmup + args.steps):
            a5_step()
        torch.cuda.synchronize()
        a5_meta.pop("_keep")
        print(json.dumps({"only": "a5", "steps": args.steps, "warmup": args.warmup, "a5": a5_meta,
                          "kernel_ms": float(np.mean(a5_event_ms[-args.steps:]))}))
        return

    if args.only == "loci":
        # the germline site kernel alone (G3: adjust_joint_eprob + position_snp_call_pprob_digt fused), for counter passes
        hb = synth.pileups(min(args.unique_loci, args.loci), np.random.default_rng(1000 + rank))
        db = device.DevicePileupBatch(hb, dev, tile=max(1, args.loci // hb.n_loci))
        gopt = capi.germline_options()
        evs = []
        for i in range(args.warmup + args.steps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            db.site_digt_call_fused(gopt)
            e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
        ms = float(np.mean([s.elapsed_time(e) for s, e in evs[args.warmup:]]))
        alg = 6 * db.n_calls + B_BYTES_PER_LOCUS_FIXED * db.n_loci
        print(json.dumps({"only": "loci", "loci": db.n_loci, "calls": db.n_calls, "kernel_ms": ms, "loci_per_s": db.n_loci / (ms * 1e-3),
                          "algorithmic_bytes": alg, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}))
        return

    if args.only == "realign":
        # one caller process of the multi-process whole-read leg (realign_processes_leg): the sparse and the dense scenarios through
        # sk_realign_job_add_reads + _run on the device; ready / go files keep the processes' timed regions together
        res = {}
        legs = {}
        for name, kw in (("sparse", dict(enumeration=2)), ("dense", dict(enumeration=2, max_indels=14, reads=args.realign_reads // 6))):
            step, n_reads, _ = whole_read_leg(args, capi, synth, np.random.default_rng(4242), **kw)
            legs[name] = (step, n_reads)
        sync = os.environ.get("SK_BENCH_SYNC_DIR")
        for name, (step, n_reads) in legs.items():
            if sync:
                open(os.path.join(sync, "ready_%s_%d" % (name, os.getpid())), "w").close()
                while not os.path.exists(os.path.join(sync, "go_" + name)):
                    time.sleep(0.001)
            t0 = time.time()
            for _ in range(args.steps):
                step()
            t1 = time.time()
            res[name] = {"reads": n_reads * args.steps, "t0": t0, "t1": t1}
        print(json.dumps({"only": "realign", "legs": res}))
        return

    if args.only == "pileup":
        # the one-shot pileup leg alone (P1 + two P2 launches + the scan), for counter passes
        rbatch, rb_loci = synth.pileup_reads_flat(args.pileup_reads, np.random.default_rng(1000 + rank))
        dr = device.DeviceReadBatch(rbatch, rb_loci, dev)
        evs = []
        for i in range(args.warmup + args.steps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            dr.pileup()
            e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
        ms = float(np.mean([s.elapsed_time(e) for s, e in evs[args.warmup:]]))
        print(json.dumps({"only": "pileup", "reads": rbatch.n_reads, "bases": rbatch.n_bases, "kernel_ms": ms, "bases_per_s": rbatch.n_bases / (ms * 1e-3)}))
        return

    if args.only == "somatic":
        # the somatic SNV kernels alone (S0 classify, S1 likelihoods, S2 posterior), for counter passes and A/B runs
        ns, ts = synth.somatic_pileups(min(args.unique_loci, args.somatic_loci), np.random.default_rng(1000 + rank))
        tile_s = max(1, args.somatic_loci // ns.n_loci)
        dns = device.DevicePileupBatch(ns, dev, tile=tile_s)
        dts = device.DevicePileupBatch(ts, dev, tile=tile_s)
        evs = []
        for i in range(args.warmup + args.steps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            device.somatic_snv_call_dev(dns, dts)
            e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
        ms = float(np.mean([s.elapsed_time(e) for s, e in evs[args.warmup:]]))
        print(json.dumps({"only": "somatic", "loci": dns.n_loci, "calls": dns.n_calls + dts.n_calls, "kernel_ms": ms,
                          "loci_per_s": dns.n_loci / (ms * 1e-3)}))
        return

    if args.only == "feed":
        fixture = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "feed_tiny.bam")
        with open(fixture, "rb") as f:
            bgzf_image = np.frombuffer(f.read(), np.uint8)
        n_fix_blocks = len(capi.bgzf_scan(bgzf_image)[0]) - 1
        os.environ["SK_INFLATE_KERNEL"] = "thread"
        dfeed = device.DeviceBgzfBatch(bgzf_image, dev, tile=max(1, args.feed_blocks // n_fix_blocks))
        for _ in range(args.warmup + args.steps):
            dfeed.inflate()
        torch.cuda.synchronize()
        assert int(dfeed.status.abs().sum().item()) == 0
        print(json.dumps({"only": "feed", "blocks": dfeed.n_blocks, "algorithmic_bytes": dfeed.in_bytes + dfeed.out_bytes}))
        return

    if args.only == "feed_slice":  # the launch a caller process makes: one slice of a region's BGZF blocks, every inflate kernel
        fixture = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "feed_tiny.bam")
        with open(fixture, "rb") as f:
            bgzf_image = np.frombuffer(f.read(), np.uint8)
        n_fix_blocks = len(capi.bgzf_scan(bgzf_image)[0]) - 1
        res = {}
        for kern in ("wave", "thread"):
            os.environ["SK_INFLATE_KERNEL"] = kern
            dsm = device.DeviceBgzfBatch(bgzf_image, dev, tile=max(1, 512 // n_fix_blocks))
            for _ in range(args.warmup):
                dsm.inflate()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(args.steps):
                dsm.inflate()
            ev1.record()
            torch.cuda.synchronize()
            assert int(dsm.status.abs().sum().item()) == 0, "BGZF inflation reported a malformed block"
            ms = ev0.elapsed_time(ev1) / args.steps
            res[kern] = {"blocks": dsm.n_blocks, "inflated_bytes": dsm.out_bytes, "compressed_bytes": dsm.in_bytes, "ms": ms,
                         "inflated_bytes_per_s": dsm.out_bytes / (ms * 1e-3)}
            del dsm
        del os.environ["SK_INFLATE_KERNEL"]
        print(json.dumps({"only": "feed_slice", "kernels": res}))
        return

    # ---- resident inputs (per rank: an independent batch, seeded by rank = an independent genome segment) ----
    rng = np.random.default_rng(1000 + rank)
    ua = min(args.unique_reads, args.reads)
    tile_a = max(1, args.reads // ua)
    ha = synth.align_batch_flat(ua, rng)
    da = device.DeviceAlignBatch(ha, dev, tile=tile_a)
    ub = min(args.unique_loci, args.loci)
    tile_b = max(1, args.loci // ub)
    hb = synth.pileups(ub, rng)
    db = device.DevicePileupBatch(hb, dev, tile=tile_b)
    gopt = capi.germline_options()
    torch.cuda.synchronize()

    region = shard.Region(dist, world, torch.cuda.synchronize,
                          lambda v: torch.tensor(v, dtype=torch.float64, device=dev))

    def timed(fn, steps, warmup, units_per_step):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        dt, units = region.timed(fn, steps, warmup, units_per_step,
                                 on_step=lambda i, before: evs[i][0 if before else 1].record())
        kern_ms = float(np.mean([s.elapsed_time(e) for s, e in evs]))
        return dt, units, kern_ms

    # ---- hot path A ----
    cells_per_step = int(np.diff(ha.read_off).astype(np.int64).dot(np.diff(ha.cal_off).astype(np.int64))) * tile_a
    dt_a, cells, kms_a = timed(lambda: da.score(), args.steps, args.warmup, cells_per_step)
    value = cells / dt_a
    alg_bytes_a = A_BYTES_PER_READ * da.n_reads
    ach_a = alg_bytes_a / (kms_a * 1e-3) / 1e9

    # ---- a5 as one function: flattening + scoring from the resident candidate alignments (the headline) ----
    a5_step, a5_meta, a5_event_ms = a5_leg(args, capi, synth)
    dt_a5, a5_cells, _ = timed(a5_step, args.steps, args.warmup, a5_meta["cells"])
    kms_a5 = float(np.mean(a5_event_ms[-args.steps:]))
    a5_keep = a5_meta.pop("_keep")

    # ---- hot path B (germline): dependent eprob + site genotype call ----
    def step_b():
        db.site_digt_call_fused(gopt)
    dt_b, loci, kms_b = timed(step_b, args.steps, args.warmup, db.n_loci)
    loci_per_s = loci / dt_b
    alg_bytes_b = 6 * db.n_calls + B_BYTES_PER_LOCUS_FIXED * db.n_loci
    ach_b = alg_bytes_b / (kms_b * 1e-3) / 1e9

    # ---- row a8: reads -> pileup columns (feeds hot path B) ----
    rbatch, rb_loci = synth.pileup_reads_flat(args.pileup_reads, rng)
    dr = device.DeviceReadBatch(rbatch, rb_loci, dev)
    dt_p, pbases, kms_p = timed(lambda: dr.pileup(), args.steps, args.warmup, rbatch.n_bases)
    pileup_alg_bytes = 8 * rbatch.n_bases  # 4 B per read base (P1) + 4 B per call (P2)
    del dr

    # ---- row a8 as the adapter drives it: sk_pileup_stream_push, one stage window (2 200 reads at 40x) at a time, host buffers in,
    # columns + counters + genotypes out, one host thread (site 9; P1 + three-column P2 + G3 chained on the device)
    sb, sb_loci = synth.pileup_reads_flat(1 << 16, np.random.default_rng(77))
    stream = capi.PileupStream(capi.pileup_options(report_begin=0, report_end=sb_loci + 200), capi.germline_options())
    win_reads = 2200
    subs = []
    for lo in range(0, sb.n_reads, win_reads):
        hi = min(sb.n_reads, lo + win_reads)
        subs.append((synth.ReadBatch(sb.read_off[lo:hi + 1] - sb.read_off[lo], sb.read_code[sb.read_off[lo]:sb.read_off[hi]],
                                     sb.read_qual[sb.read_off[lo]:sb.read_off[hi]], sb.path_off[lo:hi + 1] - sb.path_off[lo],
                                     sb.path[sb.path_off[lo]:sb.path_off[hi]], sb.pos[lo:hi], sb.is_fwd[lo:hi], sb.mapq[lo:hi],
                                     sb.map_level[lo:hi], "", 0), int(sb.pos[hi]) if hi < sb.n_reads else 2**31 - 1))

    def stream_step():
        stream.begin_region(sb.ref_seq, 0, 0, sb_loci + 200)
        for sub, final_to in subs:
            stream.push_raw(sub, final_to)
    dt_ps, ps_bases, _ = timed(stream_step, max(2, args.steps // 4), 1, sb.n_bases)
    stream_windows = len(subs)
    stream.close()

    # ---- hot path B (somatic SNV): 30-state grid likelihoods + posterior, normal 40x + tumor 110x ----
    ns, ts = synth.somatic_pileups(min(args.unique_loci, args.somatic_loci), rng)
    tile_s = max(1, args.somatic_loci // ns.n_loci)
    dns = device.DevicePileupBatch(ns, dev, tile=tile_s)
    dts = device.DevicePileupBatch(ts, dev, tile=tile_s)
    dt_s, sloci, kms_s = timed(lambda: device.somatic_snv_call_dev(dns, dts), args.steps, args.warmup, dns.n_loci)
    somatic_calls = dns.n_calls + dts.n_calls
    somatic_loci_n = dns.n_loci
    del dns, dts

    # ---- hot path B (indels): a14 21-state grid likelihoods of one sample at tumor depth; a11 allele-group genotypes ----
    hrs = synth.readscore_batch(args.indels, rng, depth_mean=110.0)
    drs = device.DeviceReadScoreBatch(hrs, dev)
    # (the library's default -- the fast form, what the drop-in runs -- is the leg's figure; the exact form beside it)
    fast_opt_i = capi.indel_options(True, exact=False)
    assert fast_opt_i.fast_form == 1
    dt_i, iloci, kms_i = timed(lambda: drs.grid_lhood(opt=fast_opt_i), args.steps, args.warmup, drs.n_indels)
    dt_if, iloci_f, kms_if = dt_i, iloci, kms_i
    dt_ix, iloci_x, kms_ix = timed(lambda: drs.grid_lhood(opt=capi.indel_options(True, exact=True)), args.steps, args.warmup, drs.n_indels)
    indel_alg_bytes = 16 * drs.n_reads + 8 * 21 * drs.n_indels  # 16 B per read + 8 B per state
    # ... and on the same reads with ONE read length, what a WGS sample has (the batch above mixes lengths 8..200 into 15 % of the reads, a
    # test input): the two logs of get_het_observed_allele_ratio are then 19 pairs per indel, not per read
    hrs.read_length[:] = 150
    drs1 = device.DeviceReadScoreBatch(hrs, dev)
    dt_i1, iloci_1, kms_i1 = timed(lambda: drs1.grid_lhood(opt=capi.indel_options(True, exact=True)), args.steps, args.warmup, drs1.n_indels)
    del drs1
    hag = synth.allele_group_batch(args.indels, rng)
    dag = device.DeviceAlleleGroupBatch(hag, dev)
    dt_g, gloci, kms_g = timed(lambda: dag.genotype_lhoods(), args.steps, args.warmup, dag.n_groups)
    fast_opt_g = capi.indel_options(False)
    fast_opt_g.fast_form = 1
    dt_gf, gloci_f, kms_gf = timed(lambda: dag.genotype_lhoods(opt=fast_opt_g), args.steps, args.warmup, dag.n_groups)
    group_alg_bytes = (8 * capi.MAX_ALT + 5) * dag.n_reads + 128 * dag.n_groups
    hag.read_length[:] = 150  # ... and with one read length, as for the grid kernel
    dag1 = device.DeviceAlleleGroupBatch(hag, dev)
    dt_g1, gloci_1, kms_g1 = timed(lambda: dag1.genotype_lhoods(), args.steps, args.warmup, dag1.n_groups)
    del dag1
    del drs, dag

    # ---- next row f2: GlobalAligner<int>, haplotype vs reference segment (device-resident entry point) ----
    pairs = synth.align_pairs(args.align_problems, rng)
    dga = device.DeviceGlobalAlignBatch(pairs, dev)
    dt_ga, ga_cells, kms_ga = timed(lambda: dga.align(), args.steps, args.warmup, dga.cells)
    n_ga = dga.n
    ga_alg_bytes = dga.nq + dga.nr + 72 * dga.n  # sequences in, score / begin / ~8 path segments out per problem
    del dga

    # ---- next row f4, the feed: BGZF inflation (the committed fixture BAM tiled on the device) ----
    fixture = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "feed_tiny.bam")
    with open(fixture, "rb") as f:
        bgzf_image = np.frombuffer(f.read(), np.uint8)
    n_fix_blocks = len(capi.bgzf_scan(bgzf_image)[0]) - 1
    dfeed = device.DeviceBgzfBatch(bgzf_image, dev, tile=max(1, args.feed_blocks // n_fix_blocks))
    dt_f, feed_bytes, kms_f = timed(lambda: dfeed.inflate(), max(2, args.steps // 4), 1, dfeed.out_bytes)
    assert int(dfeed.status.abs().sum().item()) == 0, "BGZF inflation reported a malformed block"
    feed_alg_bytes = dfeed.in_bytes + dfeed.out_bytes
    feed_blocks = dfeed.n_blocks
    del dfeed
    # ... and at the launch size a caller process has (one slice of a region: a few hundred blocks), where the latency of one block is
    # the whole cost: the wave-per-block kernel (the default up to 16 384 blocks) against the thread-per-block one
    feed_small = {}
    for kern in ("wave", "thread"):
        os.environ["SK_INFLATE_KERNEL"] = kern
        dsm = device.DeviceBgzfBatch(bgzf_image, dev, tile=max(1, 512 // n_fix_blocks))
        dt_fs, fs_bytes, kms_fs = timed(lambda: dsm.inflate(), max(2, args.steps // 4), 1, dsm.out_bytes)
        assert int(dsm.status.abs().sum().item()) == 0, "BGZF inflation reported a malformed block"
        feed_small[kern] = {"blocks": dsm.n_blocks, "inflated_bytes": dsm.out_bytes, "kernel_ms": kms_fs, "inflated_bytes_per_s": fs_bytes / dt_fs}
        del dsm
    del os.environ["SK_INFLATE_KERNEL"]

    # ---- rows a1-a7: the whole read path as the adapter drives it (host stages + kernel), one host thread ----
    wr = {}
    for name, kw in (("", dict(enumeration=2)), ("_host_enumeration", dict(enumeration=0)),
                     ("_dense", dict(enumeration=2, max_indels=14, reads=args.realign_reads // 6)),
                     ("_dense_host_enumeration", dict(enumeration=0, max_indels=14, reads=args.realign_reads // 6))):
        wr_step, wr_reads, wr_cals = whole_read_leg(args, capi, synth, np.random.default_rng(4242), **kw)
        n_wr = max(2, args.steps // 4)
        dt_wr, wr_done, _ = timed(wr_step, n_wr, 1, wr_reads)
        wr["realign%s_reads_per_s" % name] = wr_done / dt_wr
        wr["realign%s_ms_per_step" % name] = dt_wr / n_wr * 1e3
        wr["realign%s_reads_per_step_per_gpu" % name] = wr_reads
        wr["realign%s_candidate_alignments_per_read" % name] = wr_cals / max(1, wr_reads)

    # ---- end to end (N > 1; a single rank ran these legs before it touched the GPU, see the top of main) ----
    if world > 1:
        # (this process steps aside as far as it can: its cached device memory goes back and its stream is idle; the caller processes of
        # the legs are clients of the device's broker, so this process's own context costs them one of the device's eight process slots
        # and nothing else)
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        args.e2e_max_procs_per_gpu = max(args.e2e_max_procs_per_gpu, 16)  # (broker clients: no process-slot ceiling, see e2e_leg)

        def barrier():
            dist.barrier()

        def max_over_ranks(v):
            t = torch.tensor([v], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        if args.e2e_bp > 0:
            e2e = e2e_leg(args, rank, world, local_rank, barrier, max_over_ranks, with_reference=False)
        if args.e2e_somatic_bp > 0:
            e2e_somatic = e2e_leg(args, rank, world, local_rank, barrier, max_over_ranks, with_reference=False, mode="somatic")

    traffic = pmc_traffic(args)
    # the germline site kernel this run launched: variant 0 is round 1's statement, every other value the v2 kernel (the default)
    g3_kernel = "germline_site_fused_kernel" if os.environ.get("SK_G3_VARIANT", "") == "0" else "germline_site_fused_v2_kernel"
    som_kernels = ("somatic_classify_kernel", "somatic_lhood_kernel", "somatic_posterior_kernel")
    som_traffic = sum(traffic[k] for k in som_kernels) if all(k in traffic for k in som_kernels) else None
    pil_kernels = ("pileup_read_kernel", "pileup_column_kernel_t")
    pil_traffic = (traffic["pileup_read_kernel"] + 2 * traffic["pileup_column_kernel_t"]) if all(k in traffic for k in pil_kernels) else None

    def roof(kernel, alg_bytes, kernel_ms, hbm_traffic):
        """roofline object: `achieved` = algorithmic bytes / kernel time; `measured_hbm_gbs` = counter-measured HBM
        bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/) / the same kernel time, i.e. what the memory system
        actually delivered; both as fractions of the 8 TB/s peak"""
        ach = alg_bytes / (kernel_ms * 1e-3) / 1e9
        o = {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "traffic": hbm_traffic, "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kernel_ms}
        # the binding resource beside the HBM fraction: busy cycles of the busiest unit / kernel cycles, from the SQ counter passes of the
        # visit the traffic came from (VALU issue, LDS arrays, scalar unit, vector-memory issue; "hbm" when the measured HBM fraction is
        # the largest; "latency" when no unit is busy 30 % of the time)
        names = [n.split(" ")[0] for n in kernel.replace("2*", "").split("+")]
        bounds = [traffic.get("__bound__", {}).get(n) for n in names]
        bounds = [b for b in bounds if b and "bound_by" in b]
        if bounds:
            worst = max(bounds, key=lambda b: b.get("kernel_cycles", 0))  # (several kernels: the one that takes the longest)
            o["bound_by"], o["frac_bound"] = worst["bound_by"], worst["frac_bound"]
            o["bound_units"] = {u: worst.get(u) for u in ("valu_busy", "lds_busy", "salu_busy", "vmem_busy", "wave_issue", "wave_wait", "waves_per_cu", "hbm_frac_measured") if worst.get(u) is not None}
            o["bound_source"] = traffic.get("__source__")
        if hbm_traffic:
            o["traffic_source"] = traffic.get("__source__")
            o["measured_hbm_gbs"] = hbm_traffic / (kernel_ms * 1e-3) / 1e9
            o["frac_measured"] = o["measured_hbm_gbs"] / HBM_PEAK_GBS
            o["traffic_over_algorithmic"] = hbm_traffic / alg_bytes
            if hbm_traffic < alg_bytes:
                # a kernel that moves fewer bytes than the formula prices (its inputs were pre-digested elsewhere) is described by
                # what the memory system delivered, not by the formula
                o["frac_by_formula"] = o["frac"]
                o["achieved_by_formula"] = o["achieved"]
                o["achieved"] = o["measured_hbm_gbs"]
                o["frac"] = o["frac_measured"]
        return o
    out = {
        "metric": "candidate-alignment scoring cells/s (read bases x candidate alignments; Strelka2 has no pair-HMM, "
                  "SURVEY.md section 0) + germline loci/s + 40x WGS-like germline wall-clock (e2e) + 110x/40x somatic wall-clock (e2e_somatic)",
        "value": a5_cells / dt_a5, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_a5 / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "germline chr20-style synthetic (BASELINE.json configs[1]).  Headline: scoreCandidateAlignment as one "
                               "function -- per GPU per step %d reads of 150 bp x %.1f candidate alignments each (5-7 candidate indels "
                               "around them), from the candidate alignments the device search left (PCal) to one double each: "
                               "one kernel (F5: the walk of each alignment's path + the terms of its bases, penalties and clips).  sum_only: the table "
                               "sums alone over %d reads x 64 candidate alignments whose base comparisons were made before the clock "
                               "started (round 2's headline).  loci leg: %d loci, depth~Poisson(40).  e2e: see e2e.workload"
                               % (a5_meta["reads"], a5_meta["candidate_alignments_per_read"], da.n_reads, db.n_loci),
                   "reads_per_step_per_gpu": a5_meta["reads"], "candidates_per_read": a5_meta["candidate_alignments_per_read"], "read_len": 150,
                   "loci_per_step_per_gpu": db.n_loci, "sharding": "independent segments per GPU, no collective"},
        "a5": dict(a5_meta, kernel_ms=kms_a5),
        "sum_only_cells_per_s": value, "sum_only_ms_per_step": dt_a / args.steps * 1e3, "sum_only_reads_per_step_per_gpu": da.n_reads,
        "pileup_read_bases_per_s": pbases / dt_p, "pileup_ms_per_step": dt_p / args.steps * 1e3,
        "pileup_reads_per_step_per_gpu": rbatch.n_reads,
        "pileup_stream_read_bases_per_s": ps_bases / dt_ps, "pileup_stream_windows_per_step": stream_windows,
        "pileup_stream_ms_per_window": dt_ps / max(2, args.steps // 4) / stream_windows * 1e3,
        "pileup_stream_note": "sk_pileup_stream_push as the adapter calls it (site 9): windows of 2 200 reads, host buffers in, raw columns + "
                              "MAPQ tracker + genotypes of the finalised positions out, one host thread, one device round trip per window",
        "somatic_loci_per_s": sloci / dt_s, "somatic_ms_per_step": dt_s / args.steps * 1e3,
        "somatic_loci_per_step_per_gpu": somatic_loci_n,
        "roofline_somatic": roof("somatic_classify_kernel+somatic_lhood_kernel+somatic_posterior_kernel", 2 * somatic_calls + 273 * somatic_loci_n, kms_s, som_traffic),
        "indel_grid_loci_per_s": iloci / dt_i, "indel_grid_ms_per_step": dt_i / args.steps * 1e3,
        "roofline_indel_grid": roof("indel_grid_lhood_kernel", indel_alg_bytes, kms_i, traffic.get("indel_grid_lhood_kernel")),
        "indel_grid_fast_form_loci_per_s": iloci_f / dt_if, "indel_grid_fast_form_kernel_ms": kms_if,
        "indel_grid_exact_form_loci_per_s": iloci_x / dt_ix, "indel_grid_exact_form_kernel_ms": kms_ix,
        "indel_grid_one_read_length_loci_per_s": iloci_1 / dt_i1, "indel_grid_one_read_length_kernel_ms": kms_i1,
        "allele_group_one_read_length_loci_per_s": gloci_1 / dt_g1, "allele_group_one_read_length_kernel_ms": kms_g1,
        "allele_group_fast_form_loci_per_s": gloci_f / dt_gf, "allele_group_fast_form_kernel_ms": kms_gf,
        "fast_form_note": "sk_indel_options.fast_form = 1, the library's default and what indel_grid_* / roofline_indel_grid measure: the "
                          "algebraically equal form with two exp per read shared by its states and one log per state (agrees with the "
                          "reference's operation order to ~1e-13 relative, within north_star's 1e-5; the integer outputs are pinned by "
                          "tests/test_gpu_parity.py, the somatic end-to-end outputs stay byte-identical); *_exact_form_*: fast_form = 0, "
                          "bit-identical doubles; *_one_read_length_*: the exact form on reads of one length",
        "allele_group_loci_per_s": gloci / dt_g, "allele_group_ms_per_step": dt_g / args.steps * 1e3,
        "roofline_allele_group": roof("allele_group_kernel", group_alg_bytes, kms_g, traffic.get("allele_group_kernel")),
        "roofline_pileup": roof("pileup_read_kernel+2*pileup_column_kernel_t", pileup_alg_bytes, kms_p, pil_traffic),
        "roofline_global_align": roof("global_align_kernel", ga_alg_bytes, kms_ga, traffic.get("global_align_kernel")),
        "realign_host_threads": 1,
        "realign_note": "whole read path (rows a1-a7) through sk_realign_job_add_reads + _run, one host thread, host buffers in and out; "
                        "realign_* = candidate alignments listed, flattened, scored and selected / indel-scored (stage 3) on the device "
                        "(enumeration=2), *_host_enumeration = listed, flattened and finished on the host (round 1's path), *_dense = scenarios "
                        "with up to 14 indels around a read, realign_processes = the device legs driven by several caller processes at "
                        "once on the one GPU, as the reference is run for cpu_baseline (one process per core): rates summed",
        "feed_inflated_bytes_per_s": feed_bytes / dt_f, "feed_ms_per_step": dt_f / max(2, args.steps // 4) * 1e3, "feed_bgzf_blocks_per_step": feed_blocks,
        "roofline_feed": roof("bgzf_inflate_kernel+bgzf_crc32_kernel", feed_alg_bytes, kms_f,
                              (traffic["bgzf_inflate_kernel"] + traffic["bgzf_crc32_kernel"]) if all(k in traffic for k in ("bgzf_inflate_kernel", "bgzf_crc32_kernel")) else None),
        "feed_slice_sized_launch": feed_small,
        "global_align_cells_per_s": ga_cells / dt_ga, "global_align_ms_per_step": dt_ga / args.steps * 1e3,
        "global_align_problems_per_step": n_ga,
        "loci_per_s": loci_per_s, "loci_ms_per_step": dt_b / args.steps * 1e3, "loci_dtype": "f32",
        "roofline": roof("flatten_score_kernel (F5: flattening + scoring of a read's candidate alignments in one launch)",
                         a5_meta["algorithmic_bytes"], kms_a5, traffic.get("__a5_step__")),
        "roofline_sum_only": roof("score_wave_per_read_cols", alg_bytes_a, kms_a, traffic.get("score_wave_per_read_cols")),
        "roofline_loci": roof(g3_kernel, alg_bytes_b, kms_b, traffic.get(g3_kernel)),
    }
    out.update(wr)
    out["roofline_global_align"]["note"] = ("latency-bound integer DP; its traffic is the back-pointer matrix (one byte per cell in global "
                                            "scratch, written by the sweep and read back by the traceback), not the sequences the formula "
                                            "prices: frac says nothing here")
    out["roofline_feed"]["note"] = ("a serial bit stream per block: the traffic is the matches' sources (a lane's 32 KB window, 1.3e5 lanes) "
                                    "and its partial-line stores, DESIGN.md section 3 B1")
    out["roofline"]["note"] = ("scoreCandidateAlignment as one kernel, from the records the device search left to one double each; `traffic` from the "
                               "--only a5 counter passes (tools/gpu_visit.sh pmc_traffic); the table sums alone over a prepared batch are roofline_sum_only; "
                               "the staged chain it replaced (F1-F3 + A1c, $SK_A5_FUSED=0) was 2.8x slower than round 4's F5 on this job (profiles/r04_*) and is ~4x slower than this one (profiles/r05_f5_history.txt)")
    out["realign_processes"] = realign_processes
    out["e2e"] = e2e
    out["e2e_somatic"] = e2e_somatic
    out["e2e_box"] = e2e_box
    out["e2e_somatic_box"] = e2e_somatic_box
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if e2e_failed:
        sys.exit(1)  # a drop-in whose bytes differ is not a result: the line above carries first_difference, the exit code says so


if __name__ == "__main__":
    main()
