import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from strelka_amd import synth
from tests.test_active_region import reference_outputs, _check
t0 = time.time(); tot = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    sc = synth.active_region_scenarios(200, np.random.default_rng(900000 + seed))
    try:
        _check(sc, reference_outputs(sc)); tot += len(sc)
    except Exception as e:
        print("MISMATCH seed", seed, type(e).__name__, str(e)[:200])
print("ok scenarios", tot, "in %.0fs" % (time.time() - t0))
