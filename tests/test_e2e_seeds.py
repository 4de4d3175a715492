"""tools/fuzz/e2e_seeds.py as a test: the drop-in against the reference on fresh seeded WGS-like samples whose depth, variant density,
region cuts (in and out of order) and outside inputs (ploidy VCF, no-compress BED, forced-output positions) come from the seed; both VCFs
byte for byte.  A few seeds here (the campaigns are in profiles/r05_v24_fuzz_e2e_seeds.txt and r05_v28_fuzz_e2e_seeds_gpu.txt)."""
import os
import subprocess
import sys
import tempfile

import pytest

from tests import e2e_util as E

sys.path.insert(0, os.path.join(E.REPO, "tools", "fuzz"))


def _run(seeds, variant):
    import e2e_seeds
    md = tempfile.mkdtemp(prefix="sk_models_")
    subprocess.run([sys.executable, os.path.join(E.REPO, "tools", "make_dummy_germline_models.py"), md], check=True)
    models = ("--snv-scoring-model-file", md + "/germlineSNVScoringModels.json", "--indel-scoring-model-file", md + "/germlineIndelScoringModels.json")
    cwd = os.getcwd()
    os.chdir(E.REPO)  # (the tool names its helper scripts relative to the repository)
    try:
        for seed in seeds:
            ok, msg = e2e_seeds.one(seed, variant, models)
            assert ok, msg
    finally:
        os.chdir(cwd)


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl", "samtools", "bgzip", "tabix"), reason="oracle/_ref binaries not built")
def test_seeded_samples_identical_over_the_cpu_double():
    _run((119, 126), "dbl")  # (both with a ploidy VCF, a no-compress BED and forced-output positions; 70x and 4x)


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("starling2_ref", "starling2_amd", "samtools", "bgzip", "tabix"), reason="oracle/_ref binaries not built")
def test_seeded_samples_identical_gpu():
    _run((119, 126, 128, 207, 363), "amd")


def _run_somatic(seeds, variant):
    import e2e_seeds
    cwd = os.getcwd()
    os.chdir(E.REPO)
    try:
        for seed in seeds:
            ok, msg = e2e_seeds.one_somatic(seed, variant)
            assert ok, msg
    finally:
        os.chdir(cwd)


@pytest.mark.skipif(not E.have("strelka2_ref", "strelka2_dbl", "samtools"), reason="oracle/_ref binaries not built")
def test_seeded_tumour_normal_pairs_identical_over_the_cpu_double():
    _run_somatic((8,), "dbl")  # (8x / 25x, three regions out of order, callable regions and the depth filter on)


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("strelka2_ref", "strelka2_amd", "samtools"), reason="oracle/_ref binaries not built")
def test_seeded_tumour_normal_pairs_identical_gpu():
    _run_somatic((8, 14, 16), "amd")


def _run_multi(seeds, variant):
    import e2e_seeds
    md = tempfile.mkdtemp(prefix="sk_models_")
    subprocess.run([sys.executable, os.path.join(E.REPO, "tools", "make_dummy_germline_models.py"), md], check=True)
    models = ("--snv-scoring-model-file", md + "/germlineSNVScoringModels.json", "--indel-scoring-model-file", md + "/germlineIndelScoringModels.json")
    cwd = os.getcwd()
    os.chdir(E.REPO)
    try:
        for seed in seeds:
            ok, msg = e2e_seeds.one_multi(seed, variant, models)
            assert ok, msg
    finally:
        os.chdir(cwd)


@pytest.mark.skipif(not E.have("starling2_ref", "starling2_dbl", "samtools"), reason="oracle/_ref binaries not built")
def test_seeded_two_sample_runs_identical_over_the_cpu_double():
    _run_multi((4,), "dbl")  # (40x + 15x, two regions out of order, EVS on: the variants VCF and both samples' gVCFs)


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("starling2_ref", "starling2_amd", "samtools"), reason="oracle/_ref binaries not built")
def test_seeded_two_sample_runs_identical_gpu():
    _run_multi((4, 9, 15), "amd")
