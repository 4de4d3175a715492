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


# ---- which unit binds a kernel: SQ counter passes of the same commands (tools/gpu_visit.sh pmc_bound ->
# pmc_sq1 / pmc_sq2, pmc_a5_sq1 / pmc_a5_sq2).  Per kernel, over its full-size launches:
#   valu_busy  = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * kernel cycles)     the vector units' issue cycles (the counter is in quad-cycles,
#                                                                       MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units")
#   salu_busy  = SQ_ACTIVE_INST_SCA * 4 / (SIMDs * kernel cycles)
#   lds_busy   = SQ_LDS_IDX_ACTIVE / (CUs * kernel cycles)             cycles the LDS arrays were active
#   vmem_busy  = SQ_ACTIVE_INST_VMEM * 4 / (SIMDs * kernel cycles)     vector-memory instruction issue
#   wave_issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, wave_wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES   how a resident wave spends its life
#   waves_per_cu = SQ_WAVE_CYCLES * 4 / (CUs * kernel cycles)          waves in flight per CU, averaged over the kernel
# kernel cycles = GRBM_GUI_ACTIVE of the launch (one XCD's count; divided by the 8 XCDs when the tool reports their sum: decided by
# comparing with the launch's duration at the 2.4 GHz peak clock).  `bound_by`: the busiest of valu / lds / salu / vmem, "hbm" when the
# counter-measured HBM fraction (FETCH/WRITE passes) is larger than all of them, "latency" when nothing is busier than 30 % (the
# kernel's waves wait: launch- or dependency-bound).
N_SIMD, N_CU, PEAK_HZ = 1024.0, 256.0, 2.4e9


def collect_sq(prefixes):
    per = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values per launch (in dispatch order)
    dur = defaultdict(list)
    for prefix in prefixes:
        for f in glob.glob(os.path.join(root, prefix, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                    if name.startswith("void "):
                        name = name[5:]
                    k = name.split("(")[0].split("<")[0]
                    per[k][row["Counter_Name"]].append((row.get("Dispatch_Id"), float(row["Counter_Value"])))
                    try:
                        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                            dur[k].append((row.get("Dispatch_Id"), float(row["End_Timestamp"]) - float(row["Start_Timestamp"])))
                    except (KeyError, TypeError, ValueError):
                        pass
    res = {}
    for k, counters in per.items():
        if "GRBM_GUI_ACTIVE" not in counters or "SQ_WAVE_CYCLES" not in counters:
            continue
        gui = [v for _, v in counters["GRBM_GUI_ACTIVE"]]
        top = max(gui)
        if top <= 0:
            continue
        keep = {d for d, v in counters["GRBM_GUI_ACTIVE"] if v >= 0.5 * top}   # the full-size launches (as full_size above)
        mean = {}
        for c, vals in counters.items():
            # (the passes are separate runs of one deterministic command: the n-th launch of a kernel is the same work in each;
            # counters of the pass that also took GRBM_GUI_ACTIVE are matched by dispatch id, the others by size)
            sel = [v for d, v in vals if d in keep] or full_size([v for _, v in vals])
            mean[c] = sum(sel) / len(sel)
        cycles = mean["GRBM_GUI_ACTIVE"]
        durs = [v for d, v in dur.get(k, []) if d in keep]
        xcd_sum = None
        if durs:
            ratio = cycles / (sum(durs) / len(durs) * 1e-9 * PEAK_HZ)
            xcd_sum = ratio > 2.5   # (a sum over the 8 XCDs reads ~8x the launch's duration in cycles)
            if xcd_sum:
                cycles /= 8.0

        def g(name):
            return mean.get(name)
        o = {"kernel_cycles": cycles, "cycles_were_a_sum_over_xcds": xcd_sum, "launches": len(keep)}
        if g("SQ_ACTIVE_INST_VALU") is not None:
            o["valu_busy"] = g("SQ_ACTIVE_INST_VALU") * 4 / (N_SIMD * cycles)
        if g("SQ_ACTIVE_INST_SCA") is not None:
            o["salu_busy"] = g("SQ_ACTIVE_INST_SCA") * 4 / (N_SIMD * cycles)
        if g("SQ_ACTIVE_INST_VMEM") is not None:
            o["vmem_busy"] = g("SQ_ACTIVE_INST_VMEM") * 4 / (N_SIMD * cycles)
        if g("SQ_LDS_IDX_ACTIVE") is not None:
            o["lds_busy"] = g("SQ_LDS_IDX_ACTIVE") / (N_CU * cycles)
        wc = g("SQ_WAVE_CYCLES")
        if wc:
            if g("SQ_ACTIVE_INST_ANY") is not None:
                o["wave_issue"] = g("SQ_ACTIVE_INST_ANY") / wc
            if g("SQ_WAIT_ANY") is not None:
                o["wave_wait"] = g("SQ_WAIT_ANY") / wc
            o["waves_per_cu"] = wc * 4 / (N_CU * cycles)
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT"):
            if g(c) is not None:
                o[c.lower()] = g(c)
        units = {u: o[u + "_busy"] for u in ("valu", "lds", "salu", "vmem") if (u + "_busy") in o}
        if units:
            best = max(units, key=units.get)
            o["bound_by"] = best if units[best] >= 0.30 else "latency"
            o["frac_bound"] = units[best]
        res[k] = o
    return res


bound = collect_sq(("pmc_sq1", "pmc_sq2"))
bound_a5 = collect_sq(("pmc_a5_sq1", "pmc_a5_sq2"))
for k, o in bound_a5.items():
    if k == "flatten_score_kernel":
        bound[k] = o    # (the headline's kernel at the headline's size)
for k, o in bound.items():
    if k in out and "bound_by" in o:
        hbm = out[k]["hbm_bytes_per_launch"] / max(o["kernel_cycles"] / PEAK_HZ, 1e-12) / 8.0e12
        o["hbm_frac_measured"] = hbm
        if hbm > o["frac_bound"]:
            o["bound_by"], o["frac_bound"] = "hbm", hbm
json.dump(dict(workload=workload, kernels=out, a5_only=a5_only, bound=bound or None, commit=commit), sys.stdout, indent=1)
