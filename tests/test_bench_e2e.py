"""bench.py's end-to-end leg (the "40x WGS germline wall-clock" half of BASELINE.json's metric): its plumbing -- data set, segment
farm, reference leg on the same cores, byte comparison, hook-timer sums -- run here on the CPU double of the C-ABI.  On the GPU box
bench.py runs the same function with `starling2_amd`."""
import argparse
import os

import pytest

from tests import e2e_util as E


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
def test_e2e_leg_runs_and_compares(monkeypatch):
    import bench
    monkeypatch.setenv("SK_E2E_VARIANT", "dbl")
    args = argparse.Namespace(e2e_bp=400000, e2e_segment_bp=100000, e2e_max_procs_per_gpu=2)
    out = bench.e2e_leg(args, 0, 1, 0, lambda: None, lambda v: v, with_reference=True)
    assert out["identical"] is True and out["segments"] == 4 and out["bp"] == 400000
    # the node-level pair of legs: a mixed farm (2 drop-in processes + the reference on the other cores) against the reference on all
    ac = out["all_cores"]
    assert ac["identical"] is True and ac["amd_gpu_procs"] == 2 and ac["amd_fill_procs_running_the_reference"] >= 1
    assert ac["fill_segments"] == ac["amd_fill_procs_running_the_reference"] and ac["ref_procs"] == ac["cores"]
    assert out["variant_records"] > 300 and out["ref_wall_s"] > 0 and out["amd_wall_s"] > 0
    assert out["hook_seconds"]["pileup_abi"] > 0 and out["hook_seconds"]["pileup_hook"] >= out["hook_seconds"]["pileup_abi"]


@pytest.mark.skipif(not E.have("strelka2_ref", "strelka2_dbl"), reason="oracle/_ref binaries not built")
def test_e2e_somatic_leg_runs_and_compares(monkeypatch):
    """the somatic leg (configs[2]): a 110x / 40x tumour-normal pair with the somatic workflow's flags, EVS models on"""
    import bench
    monkeypatch.setenv("SK_E2E_VARIANT", "dbl")
    args = argparse.Namespace(e2e_somatic_bp=200000, e2e_somatic_segment_bp=50000, e2e_max_procs_per_gpu=8)
    out = bench.e2e_leg(args, 0, 1, 0, lambda: None, lambda v: v, with_reference=True, mode="somatic")
    assert out["identical"] is True and out["segments"] == 4 and out["bp"] == 200000
    assert out["variant_records"] >= 10 and out["ref_wall_s"] > 0 and out["amd_wall_s"] > 0
    assert out["hook_seconds"]["pileup_abi"] > 0 and out["hook_seconds"]["site_abi"] == 0  # site 5 served by the stream's records
