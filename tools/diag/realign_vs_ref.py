import numpy as np, sys
from oracle import pyoracle
from strelka_amd import capi, synth
from tests.flat_interp import score_flat

def run_ours(sc):
    opt = capi.realign_options(is_haplotyping_enabled=sc["is_haplotyping_enabled"], min_read_bp_flank=sc["min_read_bp_flank"])
    job = capi.RealignJob(opt)
    job.set_reference(sc["ref_seq"], sc["ref_offset"])
    job.set_indels(sc["indels"])
    res = []
    idx = []
    for rd in sc["reads"]:
        try:
            idx.append(job.add_read(rd["code"], rd["qual"], rd["pos"], rd["path"], rd["is_fwd"], rd["map_level"], 0, rd["realign_range"], rd["observed"]))
        except capi.StrelkaAmdError as e:
            idx.append(("threw", str(e)))
    b = job.batch()
    _, lnc, lne = capi.qscore_tables()
    scores = score_flat(b, lnc, lne)
    job.finish(scores)
    for i in idx:
        if isinstance(i, tuple):
            res.append(dict(threw=True, msg=i[1])); continue
        r = job.result(i)
        r["threw"] = False
        r["cigar"] = capi.path_to_cigar(r["path"])
        res.append(r)
    return res

def key_of(sc, i):
    d = sc["indels"][i]
    return (d["pos"], d["type"], d["del_len"], d["ins_seq"])

def compare(sc, ours, want, verbose=True):
    bad = 0
    for ri, (o, w) in enumerate(zip(ours, want)):
        if o["threw"] != w["threw"]:
            print("threw mismatch", ri, o, w); bad += 1; continue
        if o["threw"]: continue
        ok = o["is_realigned"] == w["is_realigned"] and (not w["is_realigned"] or (o["pos"] == w["pos"] and o["cigar"] == w["cigar"]))
        os_ = [dict(s, key=key_of(sc, s["indel"]), alt=[(key_of(sc, a), np.float32(l)) for a, l in s["alt"]]) for s in o["scores"]]
        os_.sort(key=lambda s: s["key"][0:1] + s["key"][1:])
        ws = w["scores"]
        if len(os_) != len(ws): ok = False
        else:
            for a, b in zip(sorted(os_, key=lambda s: s["key"]), sorted(ws, key=lambda s: s["key"])):
                for f in ("key", "non_ambig", "read_length", "is_tier1_read", "is_fwd_strand", "read_pos", "edge_dist"):
                    if a[f] != b[f]: ok = False
                if np.float32(a["ref_lnp"]) != np.float32(b["ref_lnp"]) or np.float32(a["indel_lnp"]) != np.float32(b["indel_lnp"]): ok = False
                if [(k, np.float32(l)) for k, l in a["alt"]] != [(k, np.float32(l)) for k, l in b["alt"]]: ok = False
        if sorted(key_of(sc, i) for i in o["suboverlap"]) != sorted(w["suboverlap"]): ok = False
        if not ok:
            bad += 1
            if verbose:
                print("MISMATCH read", ri, sc["reads"][ri]["pos"], capi.path_to_cigar(sc["reads"][ri]["path"]))
                print("  ours:", {k: v for k, v in o.items() if k != "path"})
                print("  want:", w)
    return bad

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
scs = synth.realign_scenarios(n, np.random.default_rng(seed), max_indels=int(sys.argv[3]) if len(sys.argv) > 3 else 6)
want = pyoracle.ref_realign_scenarios(scs)
tot = bad = 0
for sc, w in zip(scs, want):
    o = run_ours(sc)
    b = compare(sc, o, w)
    if b:
        print("scenario indels:", sc["indels"], "hap", sc["is_haplotyping_enabled"], "flank", sc["min_read_bp_flank"])
    bad += b; tot += len(w)
print("S-clipped out", sum("S" in r.get("cigar", "") for w in want for r in w), "alts", sum(len(x["alt"]) for w in want for r in w for x in r.get("scores", [])))
print("reads", tot, "bad", bad, "realigned", sum(r.get("is_realigned", False) for w in want for r in w), "threw", sum(r["threw"] for w in want for r in w), "scores", sum(len(r.get("scores", [])) for w in want for r in w))
