"""path B restatement (oracle/strelka_oracle.c) vs the reference's own translation units on fresh seeds: scalar helpers,
std::sort tie order, adjust_joint_eprob + position_snp_call_pprob_digt, somatic sample likelihoods + grid posterior,
indel grid / allele-group likelihoods.  Usage: fuzz_pathb.py SEED_BEGIN SEED_END"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pyoracle
from tests.golden.make_golden import pathb_vectors
from tests import test_oracle_pinned as T

pyoracle.build(ref=True, quiet=True)
O = pyoracle.oracle()
t0 = time.time(); n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    gold = pathb_vectors(np.random.default_rng(7000000 + seed))
    try:
        T.test_tables_and_scalars_match_reference.__wrapped__(gold, O) if hasattr(T.test_tables_and_scalars_match_reference, "__wrapped__") else T.test_tables_and_scalars_match_reference(gold, O)
        T.test_std_sort_tie_order_matches_reference(gold, O)
        T.test_germline_matches_reference(gold)
        T.test_somatic_snv_matches_reference(gold, O)
        T.test_indel_likelihoods_match_reference(gold)
        n += 1
    except AssertionError as e:
        print("MISMATCH seed", seed, str(e)[:300])
print("ok seeds", n, "in %.0fs" % (time.time() - t0))
