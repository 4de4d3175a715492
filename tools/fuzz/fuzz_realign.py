import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import pyoracle
from strelka_amd import capi, synth
from tests.test_read_realign import _run_scenarios
tot = 0
t0 = time.time()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(100000 + seed)
    scs = synth.realign_scenarios(150, rng, reads_per=10, max_indels=int(rng.integers(3, 13)),
                                  haplotyping_rate=float(rng.choice([0.0, 0.25, 0.6])))
    exp = pyoracle.ref_realign_scenarios(scs)
    try:
        n_reads, n_cals = _run_scenarios(scs, exp, on_gpu=False)
    except AssertionError as e:
        print("MISMATCH seed", seed, str(e)[:300]); continue
    tot += n_reads
print("ok reads", tot, "in %.0fs" % (time.time() - t0))
