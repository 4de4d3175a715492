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
    assert out["first_difference"] is None and "all_cores" not in out
    assert out["variant_records"] > 300 and out["ref_wall_s"] > 0 and out["amd_wall_s"] > 0
    assert out["hook_seconds"]["pileup_abi"] > 0 and out["hook_seconds"]["pileup_hook"] >= out["hook_seconds"]["pileup_abi"]
    c = out["counters"]
    assert c["normalize_declined"] == 0 and c["realign_jobs"] > 0 and c["pushes"] > 0 and c["loci"] > 300000, c
    # site 10: nearly every covered position went from the stream's window into the writer's block -- most of them as members of a
    # block installed whole (the device's walk from its first site), the rest one by one; the reference built a locus for the others
    routed = c["gvcf_plain_sites"] + c["gvcf_block_sites"]
    assert routed > 0.9 * 400000 and c["gvcf_reference_sites"] < 0.05 * routed and c["gvcf_block_sites"] > 0.5 * routed, c
    assert c["gvcf_filter_key_mismatches"] == 0, c


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


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
def test_e2e_leg_reports_the_first_difference(monkeypatch, tmp_path):
    """a drop-in whose bytes differ is reported, not averaged away: the leg names the first differing line (here: the drop-in's
    second segment writes its header again, through the argv hook the test installs)"""
    import bench
    from strelka_amd import farm
    monkeypatch.setenv("SK_E2E_VARIANT", "dbl")
    monkeypatch.setenv("SK_E2E_KEEP_DIR", str(tmp_path / "kept"))
    real = farm.germline_segment_argv

    def skewed(binary, *a, **kw):
        argv = real(binary, *a, **kw)
        if binary.endswith("_dbl") and "--gvcf-skip-header" in argv:
            argv.remove("--gvcf-skip-header")
        return argv
    monkeypatch.setattr(farm, "germline_segment_argv", skewed)
    args = argparse.Namespace(e2e_bp=300000, e2e_segment_bp=150000, e2e_max_procs_per_gpu=2)
    out = bench.e2e_leg(args, 0, 1, 0, lambda: None, lambda v: v, with_reference=True)
    assert out["identical"] is False
    fd = out["first_difference"]
    assert fd["file"] in ("variants.vcf", "genome.S1.vcf") and fd["line"] >= 1 and fd["drop_in"] != fd["reference"]
    assert os.path.exists(str(tmp_path / "kept" / ("germline_drop_in_" + fd["file"])))


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl"), reason="oracle/_ref binaries not built")
def test_e2e_box_leg_runs_and_compares(monkeypatch):
    """the box ratio's plumbing (the reference at -j P, P = usable cores, against the drop-in's best process count; every
    run byte-compared), on the CPU double -- which has no broker: the runs differ in process count only"""
    import bench
    from strelka_amd import farm
    monkeypatch.setenv("SK_E2E_VARIANT", "dbl")
    args = argparse.Namespace(e2e_bp=480000, e2e_somatic_bp=0)
    out = bench.e2e_box_leg(args, 0)
    P = len(farm.usable_cores())
    assert out["identical"] is True and out["segments"] == max(P, 16) and out["ref_cores"] == P and out["host_cores"] == P
    assert [r["callers"] for r in out["runs"]][-1] == "own_context" and {r["callers"] for r in out["runs"][:-1]} == {"broker"}
    assert out["first_wave_identical"] is True and out["first_wave_wall_s"] > 0
    assert all(r["identical"] and r["wall_s"] > 0 for r in out["runs"])
    assert out["amd_wall_s"] == min(r["wall_s"] for r in out["runs"]) and abs(out["speedup"] - out["ref_wall_s"] / out["amd_wall_s"]) < 1e-9


# ---- the legs at the configuration the metric is quoted on (what the driver's bench run does; bench.py's defaults): chr20's 64 Mb cut
# into 12 Mb segments as the workflow cuts it (configs[1]) / a 16 Mb tumour-normal pair in 2 Mb segments (configs[2]), caller processes
# sharing one GPU, the workflow's command line with the EVS models on.  BENCH_r03's germline leg failed at its bench configuration while
# every smaller test passed: the sizes here ARE the bench's (read from its argument parser, so they cannot drift apart).
def _bench_defaults():
    import sys
    import bench
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        return bench.parse()
    finally:
        sys.argv = argv


def _assert_routed(out, somatic=False):
    """identical bytes are the routed path's only if the counters say the work went through the C-ABI"""
    c = out["counters"]
    assert c["normalize_declined"] == 0, c          # no alignment went back to the reference's own normalizeAlignment
    assert c["enum_host_instead"] == 0, c           # no read of a device job was turned down and listed by the host statement
    assert c["realign_jobs"] > 0 and c["enum_device_reads"] > 0, c
    assert c["pushes"] > 0 and c["loci"] >= out["bp"] * 0.9, c
    if not somatic:
        # site 10: the reference built a site locus for at most 5 % of the covered positions; the rest went from the stream's window
        # straight into the writer's open block
        covered = c["gvcf_plain_sites"] + c["gvcf_block_sites"] + c["gvcf_reference_sites"]
        assert covered >= out["bp"] * 0.9 and c["gvcf_reference_sites"] <= 0.05 * covered, c
        assert c["gvcf_block_sites"] > 0.5 * covered and c["gvcf_filter_key_mismatches"] == 0, c


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("starling2_ref", "starling2_amd"), reason="oracle/_ref binaries not built")
def test_e2e_germline_at_bench_configuration_identical_gpu():
    import bench
    d = _bench_defaults()
    assert d.e2e_bp == 64000000 and d.e2e_segment_bp == 12000000
    args = argparse.Namespace(e2e_bp=d.e2e_bp, e2e_segment_bp=d.e2e_segment_bp, e2e_max_procs_per_gpu=d.e2e_max_procs_per_gpu)
    out = bench.e2e_leg(args, 0, 1, 0, lambda: None, lambda v: v, with_reference=True)
    assert out["identical"] is True, out["first_difference"]
    assert out["segments"] == 6 and out["bp"] == 64000000 and out["variant_records"] > 50000
    _assert_routed(out)


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("strelka2_ref", "strelka2_amd"), reason="oracle/_ref binaries not built")
def test_e2e_somatic_at_bench_configuration_identical_gpu():
    import bench
    d = _bench_defaults()
    assert d.e2e_somatic_bp >= 16000000 and d.e2e_somatic_segment_bp >= 2000000
    args = argparse.Namespace(e2e_somatic_bp=d.e2e_somatic_bp, e2e_somatic_segment_bp=d.e2e_somatic_segment_bp,
                              e2e_max_procs_per_gpu=d.e2e_max_procs_per_gpu)
    out = bench.e2e_leg(args, 0, 1, 0, lambda: None, lambda v: v, with_reference=True, mode="somatic")
    assert out["identical"] is True, out["first_difference"]
    assert out["segments"] == 8
    _assert_routed(out, somatic=True)


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("starling2_ref", "starling2_amd"), reason="oracle/_ref binaries not built")
def test_e2e_box_germline_at_bench_configuration_identical_gpu():
    """chr20's 64 Mb with a segment per core: the reference on all cores against 8 own-context callers and 8 / 12 / 16 broker clients on
    the one GPU -- every run identical to the reference, and more than eight callers FASTER than eight (the broker's point)"""
    import bench
    d = _bench_defaults()
    args = argparse.Namespace(e2e_bp=d.e2e_bp, e2e_somatic_bp=0)
    out = bench.e2e_box_leg(args, 0)
    assert out["identical"] is True, out["runs"]
    by = {(r["callers"], r["procs"]): r for r in out["runs"]}
    if out["host_cores"] >= 16:
        assert by[("broker", 16)]["wall_s"] < by[("own_context", 8)]["wall_s"], out["runs"]
    print("\ne2e_box germline: reference -j%d %.1f s; %s; speedup %.2fx" % (out["ref_cores"], out["ref_wall_s"],
          ", ".join("%s x%d %.1f s" % (r["callers"], r["procs"], r["wall_s"]) for r in out["runs"]), out["speedup"]))


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("strelka2_ref", "strelka2_amd"), reason="oracle/_ref binaries not built")
def test_e2e_box_somatic_at_bench_configuration_identical_gpu():
    import bench
    d = _bench_defaults()
    args = argparse.Namespace(e2e_bp=0, e2e_somatic_bp=d.e2e_somatic_bp)
    out = bench.e2e_box_leg(args, 0, mode="somatic")
    assert out["identical"] is True, out["runs"]
