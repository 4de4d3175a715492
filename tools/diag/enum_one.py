import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tools/diag")
from strelka_amd import capi, synth
import enum_modes as M
capi.init(0)
rng = np.random.default_rng(5)
scs = synth.realign_scenarios(24, rng, reads_per=12, max_indels=14)
jobs = M.build(scs, 40, 2)
M.step(jobs[:3]); print("----", file=sys.stderr, flush=True)
import os
os.environ["SK_ENUM_TIMING"] = "1"
print(M.step(jobs[:3]))
