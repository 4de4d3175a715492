"""end to end: reads -> (sk_realign_job + flat interpreter scores) -> pileup restatement  vs  the reference's position processor"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pyoracle
from strelka_amd import capi, synth
from tests.flat_interp import score_flat

def cands_from(reads, k=12):
    cands = []
    for r in reads:
        p = r["pos"]
        for i, (ty, ln) in enumerate(r["path"]):
            if ty == synth.SEG["DELETE"] and 0 < i < len(r["path"]) - 1 and len(cands) < k and ln <= 20:
                if p not in [c["pos"] for c in cands]: cands.append(dict(pos=p, del_len=ln))
            if ty in (synth.SEG["MATCH"], synth.SEG["DELETE"]): p += ln
    return cands

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
tot = bad_al = bad_col = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    reads, ref, off = synth.pileup_reads(110, rng, read_len=(36, 120))
    # keep the stage sizes fixed: total indel reference span per read <= maxIndelSize
    reads = [r for r in reads if sum(l for t, l in r["path"] if t in (2, 3)) <= 49]
    opt = pyoracle.pileup_options(report_begin=off, report_end=off + len(ref))
    finals, cols, indels = pyoracle.ref_pileup_pipeline(reads, ref, off, opt, candidate_indels=cands_from(reads), return_indels=True)
    job = capi.RealignJob(capi.realign_options(is_haplotyping_enabled=0, min_read_bp_flank=5))
    job.set_reference(ref, off)
    job.set_indels(indels)
    idx = []
    for f in finals:
        r = reads[f["read_id"]]
        obs = [k for k, d in enumerate(indels) if f["read_id"] in d["read_ids"]]
        if r["map_level"] not in (1, 2):   # align_pos skips reads that are not tier1/2 mapped (is_realign_submapped_reads off)
            idx.append(None); continue
        idx.append(job.add_read(r["code"], r["qual"], f["input_pos"], capi.cigar_to_path(f["input_cigar"]), r["is_fwd"], r["map_level"], 0, f["realign_range"], obs))
    b = job.batch(); _, lnc, lne = capi.qscore_tables()
    job.finish(score_flat(b, lnc, lne))
    piled = []
    for f, i in zip(finals, idx):
        r = dict(reads[f["read_id"]])
        res = job.result(i) if i is not None else dict(is_realigned=False)
        tot += 1
        if res["is_realigned"]:
            mine = (True, res["pos"], capi.path_to_cigar(res["path"]))
        else:
            mine = (False, f["input_pos"], f["input_cigar"])
        want = (f["is_realigned"], f["pos"], f["cigar"])
        # an unchanged realignment is still "realigned" in the reference; compare the alignment actually piled up
        if mine[1:] != want[1:] or (mine[0] != want[0]):
            bad_al += 1
            if bad_al < 6: print("ALIGN MISMATCH", f["read_id"], r["map_level"], "mine", mine, "want", want, "input", f["input_pos"], f["input_cigar"])
        if f["skipped"]: continue
        r.update(pos=mine[1], path=capi.cigar_to_path(mine[2]))
        piled.append(r)
    rb = synth.ReadBatch.from_reads(piled, ref, off)
    for mode, key in ((0, "calls"), (1, "tier2_calls")):
        co, calls, sd, sm = pyoracle.pileup_reads(rb, opt, mode)
        for l in range(opt.report_end - opt.report_begin):
            w = cols.get(opt.report_begin + l)
            g = calls[co[l]:co[l + 1]]
            ok = (len(g) == 0 and sd[l] == 0 and sm[l] == 0) if w is None else (np.array_equal(g, w[key]) and sd[l] == w["spandel"] and sm[l] == w["submapped"])
            bad_col += (not ok)
    print("trial", trial, "reads", len(finals), "realigned(ref)", sum(f["is_realigned"] for f in finals), "indels", len(indels), "cand", sum(d["is_candidate"] for d in indels))
print("reads", tot, "alignment mismatches", bad_al, "column mismatches", bad_col)
