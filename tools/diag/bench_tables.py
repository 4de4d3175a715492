#!/usr/bin/env python
"""DESIGN.md section 6's tables from a bench.py line: bench_tables.py <bench.json> [<pmc_traffic.json>]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
pmc = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None


def tr(kernels, mult=None):
    if not pmc:
        return None
    k = pmc["kernels"]
    if not all(x in k for x in kernels):
        return None
    return sum(k[x]["hbm_bytes_per_launch"] * (mult[i] if mult else 1) for i, x in enumerate(kernels))


def row(name, rate, r, traffic=None):
    t = traffic if traffic is not None else r.get("traffic")
    alg = r["algorithmic_bytes_per_launch"]
    print("| %s | %s | %.3g | %.4f | %s | %s |" % (name, rate, r["kernel_ms"], alg / (r["kernel_ms"] * 1e-3) / 1e9 / 8000.0,
                                                 ("%.0f" % (t / (r["kernel_ms"] * 1e-3) / 1e9)) if t else "--", ("%.2f" % (t / alg)) if t else "--"))


print("| leg | rate | kernel ms | 8d frac | counter GB/s | traffic / algorithmic |\n|---|---|---|---|---|---|")
a5t = pmc["a5_only"]["hbm_bytes_per_step"] if pmc and pmc.get("a5_only") and pmc["a5_only"]["workload"].get("a5_reads") == d["a5"]["reads"] else None
row("a5 F5 `flatten_score_kernel`, %d reads x %.1f candidate alignments x 150 bp" % (d["a5"]["reads"], d["a5"]["candidate_alignments_per_read"]),
    "%.3g cells/s" % d["value"], d["roofline"], a5t)
row("A1c `score_wave_per_read_cols` alone (`sum_only`), %d reads x 64" % d["sum_only_reads_per_step_per_gpu"], "%.3g cells/s" % d["sum_only_cells_per_s"],
    d["roofline_sum_only"], tr(["score_wave_per_read_cols"]))
row("G3 v2 `germline_site_fused_v2_kernel`, 2^26 loci", "%.3g loci/s" % d["loci_per_s"], d["roofline_loci"], tr(["germline_site_fused_v2_kernel"]))
row("S0+S1+S2 somatic SNV, %d loci" % d["somatic_loci_per_step_per_gpu"], "%.3g loci/s" % d["somatic_loci_per_s"], d["roofline_somatic"],
    tr(["somatic_classify_kernel", "somatic_lhood_kernel", "somatic_posterior_kernel"]))
row("P1 + 2 P2 pileup, %d reads" % d["pileup_reads_per_step_per_gpu"], "%.3g bases/s" % d["pileup_read_bases_per_s"], d["roofline_pileup"],
    tr(["pileup_read_kernel", "pileup_column_kernel_t"], [1, 2]))
row("I1 `indel_grid_lhood_kernel` (exact)", "%.3g indels/s" % d["indel_grid_loci_per_s"], d["roofline_indel_grid"], tr(["indel_grid_lhood_kernel"]))
row("I3 `allele_group_kernel<3>`", "%.3g groups/s" % d["allele_group_loci_per_s"], d["roofline_allele_group"], tr(["allele_group_kernel"]))
row("B1+B2 feed, %d blocks" % d["feed_bgzf_blocks_per_step"], "%.3g inflated B/s" % d["feed_inflated_bytes_per_s"], d["roofline_feed"],
    tr(["bgzf_inflate_kernel", "bgzf_crc32_kernel"]))
print("| P-stream push (2 200 reads per window, one round trip) | %.3g bases/s | %.3f per window | | | |" % (d["pileup_stream_read_bases_per_s"], d["pileup_stream_ms_per_window"]))
print("| whole read, sparse: device / host path | %.3g / %.3g reads/s | | | | |" % (d["realign_reads_per_s"], d["realign_host_enumeration_reads_per_s"]))
print("| whole read, dense: device / host path | %.3g / %.3g reads/s | | | | |" % (d["realign_dense_reads_per_s"], d["realign_dense_host_enumeration_reads_per_s"]))
print("| D1 `global_align_kernel`, %d problems | %.3g cells/s | %.3g | | | |" % (d["global_align_problems_per_step"], d["global_align_cells_per_s"], d["roofline_global_align"]["kernel_ms"]))
print("| feed slice (510 blocks): wave / thread per block | %.3g / %.3g inflated B/s | %.3g / %.3g | | | |" % (
    d["feed_slice_sized_launch"]["wave"]["inflated_bytes_per_s"], d["feed_slice_sized_launch"]["thread"]["inflated_bytes_per_s"],
    d["feed_slice_sized_launch"]["wave"]["kernel_ms"], d["feed_slice_sized_launch"]["thread"]["kernel_ms"]))
c = d.get("cpu_baseline") or {}
if c:
    print("\ncpu_baseline (%s, %d cores): %.3g cells/s, %.3g loci/s, %.3g somatic loci/s, %.3g reads/s" % (c["kind"], c["cores"], c["value"], c["loci_per_s"],
                                                                                                   c["somatic_loci_per_s"], c["realign_reads_per_s"]))
for k in ("e2e", "e2e_somatic"):
    e = d.get(k)
    if e and "ref_wall_s" in e:
        print("%s: %d bp, %d segments, %d procs: reference %.2f s, drop-in %.2f s, %.2fx, process seconds %.1f -> %.1f, identical %s, hooks %s" % (
            k, e["bp"], e["segments"], e["amd_procs"], e["ref_wall_s"], e["amd_wall_s"], e["speedup"], e["ref_process_seconds_sum"], e["process_seconds_sum"],
            e["identical"], e["hook_seconds"]))
