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
    ap.add_argument("--loci", type=int, default=1 << 24, help="germline loci per step per GPU (depth ~Poisson(40))")
    ap.add_argument("--unique-reads", type=int, default=1 << 14, help="distinct synthetic reads (tiled on device)")
    ap.add_argument("--unique-loci", type=int, default=1 << 20)
    ap.add_argument("--pileup-reads", type=int, default=1 << 20, help="reads per step per GPU for the pileup leg (row a8)")
    ap.add_argument("--somatic-loci", type=int, default=1 << 22, help="somatic loci per step per GPU (40x normal + 110x tumor)")
    ap.add_argument("--indels", type=int, default=1 << 18, help="indel loci per step per GPU for the indel legs (a11, a14)")
    ap.add_argument("--align-problems", type=int, default=4096, help="GlobalAligner problems per step (next row f2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reads", type=int, default=1500, help="reads in the CPU-baseline sample (x64 candidates)")
    ap.add_argument("--cpu-loci", type=int, default=2000000, help="loci in the CPU-baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU time spent on the alignment-scoring sample")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle (plain-C restatement of the reference, oracle/strelka_oracle.c) timed on ONE host core over a bounded
    sample of the same workload.  This is the only place bench.py touches oracle/: as the thing the GPU number is put
    beside, never as part of the measured product path."""
    from oracle import pyoracle
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
    out = {"value": cells / ta, "unit": "cells/s", "cores": 1, "kind": "port",
           "sample": "%d reads x 64 candidate alignments x 150 bp, %d passes through sko_score_cases (%.1f s); loci: %d loci "
                     "depth~Poisson(40) through sko_adjust_joint_eprob+sko_position_snp_call_pprob_digt (%.1f s)"
                     % (len(cases), reps, ta, args.cpu_loci, tb),
           "loci_per_s": args.cpu_loci / tb}
    # where the reference's own translation units travelled with the repo (oracle/_ref, built from /root/reference by
    # oracle/Makefile), time the REFERENCE's adjust_joint_eprob + position_snp_call_pprob_digt on part of the same loci
    try:
        if pyoracle.ref_available():
            R = pyoracle.ref()
            R.ref_time_germline_sites.restype = C.c_double
            R.ref_time_germline_sites.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_double, C.POINTER(C.c_double)]
            n = min(pb.n_loci, 300000)
            chk = C.c_double()
            secs = R.ref_time_germline_sites(pb.call_off.ctypes.data, pb.calls.ctypes.data, pb.ref_base.ctypes.data, n, 0.001,
                                             C.byref(chk))
            out["loci_per_s_reference"] = n / secs
            out["sample"] += "; reference TUs (oracle/_ref) on %d of those loci: %.1f s" % (n, secs)
    except Exception as e:  # the reference build is optional test infrastructure
        out["loci_reference_error"] = str(e)
    return out


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
    return {k: v["hbm_bytes_per_launch"] for k, v in d.get("kernels", {}).items()}


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
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank

    from strelka_amd import capi, device, shard, synth
    capi.init(local_rank)

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
    del dr

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
    dt_i, iloci, kms_i = timed(lambda: drs.grid_lhood(), args.steps, args.warmup, drs.n_indels)
    indel_alg_bytes = 16 * drs.n_reads + 8 * 21 * drs.n_indels  # 16 B per read + 8 B per state
    hag = synth.allele_group_batch(args.indels, rng)
    dag = device.DeviceAlleleGroupBatch(hag, dev)
    dt_g, gloci, kms_g = timed(lambda: dag.genotype_lhoods(), args.steps, args.warmup, dag.n_groups)
    group_alg_bytes = (8 * capi.MAX_ALT + 5) * dag.n_reads + 128 * dag.n_groups
    del drs, dag

    # ---- next row f2: GlobalAligner<int>, haplotype vs reference segment (device-resident entry point) ----
    pairs = synth.align_pairs(args.align_problems, rng)
    dga = device.DeviceGlobalAlignBatch(pairs, dev)
    dt_ga, ga_cells, kms_ga = timed(lambda: dga.align(), args.steps, args.warmup, dga.cells)
    n_ga = dga.n
    del dga

    traffic = pmc_traffic(args)
    som_kernels = ("somatic_classify_kernel", "somatic_lhood_kernel", "somatic_posterior_kernel")
    som_traffic = sum(traffic[k] for k in som_kernels) if all(k in traffic for k in som_kernels) else None
    out = {
        "metric": "candidate-alignment scoring cells/s (read bases x candidate alignments; Strelka2 has no pair-HMM, "
                  "SURVEY.md section 0) + germline loci/s",
        "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_a / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "germline chr20-style synthetic (BASELINE.json configs[1]): per GPU per step %d reads x 64 "
                               "candidate alignments x 150 bp (K=6 toggled candidate indels); loci leg: %d loci, "
                               "depth~Poisson(40)" % (da.n_reads, db.n_loci),
                   "reads_per_step_per_gpu": da.n_reads, "candidates_per_read": 64, "read_len": 150,
                   "loci_per_step_per_gpu": db.n_loci, "sharding": "independent segments per GPU, no collective"},
        "pileup_read_bases_per_s": pbases / dt_p, "pileup_ms_per_step": dt_p / args.steps * 1e3,
        "pileup_reads_per_step_per_gpu": rbatch.n_reads,
        "somatic_loci_per_s": sloci / dt_s, "somatic_ms_per_step": dt_s / args.steps * 1e3,
        "somatic_loci_per_step_per_gpu": somatic_loci_n,
        "roofline_somatic": {"kernel": "somatic_classify_kernel+somatic_lhood_kernel+somatic_posterior_kernel", "bound": "hbm", "achieved": (2 * somatic_calls + 273 * somatic_loci_n) / (kms_s * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (2 * somatic_calls + 273 * somatic_loci_n) / (kms_s * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": som_traffic, "algorithmic_bytes_per_launch": 2 * somatic_calls + 273 * somatic_loci_n,
                             "kernel_ms": kms_s},
        "indel_grid_loci_per_s": iloci / dt_i, "indel_grid_ms_per_step": dt_i / args.steps * 1e3,
        "roofline_indel_grid": {"kernel": "indel_grid_lhood_kernel", "bound": "hbm", "achieved": indel_alg_bytes / (kms_i * 1e-3) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": indel_alg_bytes / (kms_i * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "traffic": traffic.get("indel_grid_lhood_kernel"), "algorithmic_bytes_per_launch": indel_alg_bytes,
                                "kernel_ms": kms_i},
        "allele_group_loci_per_s": gloci / dt_g, "allele_group_ms_per_step": dt_g / args.steps * 1e3,
        "roofline_allele_group": {"kernel": "allele_group_kernel", "bound": "hbm", "achieved": group_alg_bytes / (kms_g * 1e-3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": group_alg_bytes / (kms_g * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "traffic": traffic.get("allele_group_kernel"), "algorithmic_bytes_per_launch": group_alg_bytes,
                                  "kernel_ms": kms_g},
        "global_align_cells_per_s": ga_cells / dt_ga, "global_align_ms_per_step": dt_ga / args.steps * 1e3,
        "global_align_problems_per_step": n_ga,
        "loci_per_s": loci_per_s, "loci_ms_per_step": dt_b / args.steps * 1e3, "loci_dtype": "f32",
        "roofline": {"kernel": "score_wave_per_read", "bound": "hbm", "achieved": ach_a, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ach_a / HBM_PEAK_GBS,
                     "traffic": traffic.get("score_wave_per_read"),
                     "algorithmic_bytes_per_launch": alg_bytes_a, "kernel_ms": kms_a},
        "roofline_loci": {"kernel": "germline_site_fused_kernel", "bound": "hbm", "achieved": ach_b,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_b / HBM_PEAK_GBS,
                          "traffic": traffic.get("germline_site_fused_kernel"),
                          "algorithmic_bytes_per_launch": alg_bytes_b, "kernel_ms": kms_b},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
