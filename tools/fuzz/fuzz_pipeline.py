import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import pyoracle
from strelka_amd import capi, synth
from tests.test_pipeline import _run_trial
from tests.golden.make_golden import _candidates_from
tot = 0; t0 = time.time(); bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(500000 + seed)
    reads, ref, off = synth.pileup_reads(int(rng.integers(40, 160)), rng, read_len=(36, int(rng.integers(80, 151))))
    reads = [r for r in reads if sum(l for ty, l in r["path"] if ty in (synth.SEG["INSERT"], synth.SEG["DELETE"])) <= 49]
    end = off + len(ref) - 10
    reads = [r for r in reads if r["pos"] + sum(l for ty, l in r["path"] if ty in (synth.SEG["MATCH"], synth.SEG["DELETE"], synth.SEG["INSERT"], synth.SEG["SOFT_CLIP"])) <= end]
    kw = dict(report_begin=off, report_end=off + len(ref))
    if seed % 3 == 1:
        kw.update(min_basecall_qscore=0, mismatch_density_max_count=3, use_tier2_evidence=1)
    opt = pyoracle.pileup_options(**kw)
    try:
        finals, cols, indels = pyoracle.ref_pileup_pipeline(reads, ref, off, opt, candidate_indels=_candidates_from(reads), return_indels=True)
    except Exception as e:
        print("reference failed seed", seed, e); continue
    n_loci = opt.report_end - opt.report_begin
    empty = dict(calls=np.zeros(0, np.uint16), tier2_calls=np.zeros(0, np.uint16), spandel=0, submapped=0)
    col = [cols.get(opt.report_begin + l, empty) for l in range(n_loci)]
    csr = lambda k: (np.concatenate([[0], np.cumsum([len(c[k]) for c in col])]).astype(np.int64),
                     np.concatenate([c[k] for c in col] + [np.zeros(0, np.uint16)]).astype(np.uint16))
    t1_off, t1 = csr("calls"); t2_off, t2 = csr("tier2_calls")
    try:
        tot += _run_trial(dict(reads=reads, ref_seq=ref, ref_offset=off, opt=kw, finals=finals, indels=indels, t1_off=t1_off, t1=t1,
                        t2_off=t2_off, t2=t2, spandel=np.array([c["spandel"] for c in col], np.uint32),
                        submapped=np.array([c["submapped"] for c in col], np.uint32)), on_gpu=False)
    except Exception as e:
        bad += 1; print("MISMATCH seed", seed, type(e).__name__, str(e)[:300])
print("ok units", tot, "mismatching trials", bad, "in %.0fs" % (time.time() - t0))
