"""SURVEY.md 8f rank 4, the output side: the gVCF writer's non-variant block logic (gvcf_block_site_record::testCanSiteJoinSampleBlock
/ joinSiteToSampleBlock, L/applications/starling/gvcf_block_site_record.cpp, driven as gvcf_writer::queue_site_record drives them).

  * -m "not gpu": the statement (csrc/gvcf_block_core.h, run on the host through the CPU double of the ABI) against what the
    REFERENCE's own class made of the same sites -- the committed fixture, and live where oracle/_ref exists: which sites start /
    continue a block or stand alone, and per block the position, length, GQX, MIN_DP and the two running means bit for bit;
  * -m gpu: gvcf_block_kernel against the fixture and, at 4M sites, against the host statement."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import capi, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = int(os.environ.get("SK_TEST_SEED_OFFSET", "0"))


def _double():
    path = os.path.join(os.path.dirname(GOLD), "..", "oracle", "libstrelka_amd_double.so")
    if not os.path.exists(path):
        pytest.skip("oracle/libstrelka_amd_double.so not built")
    L = C.CDLL(os.path.abspath(path))
    L.sk_last_error.restype = C.c_char_p
    return L


def _sites(rec):
    s = np.zeros(len(rec), capi.GVCF_SITE_DTYPE)
    for k in capi.GVCF_SITE_DTYPE.names:
        s[k] = rec[k]
    return s


def _check(kind, blocks, ref_kind, ref_blocks):
    assert np.array_equal(kind, ref_kind)
    starts = np.nonzero(kind == 1)[0]
    assert np.array_equal(starts, ref_blocks["first_site"])
    b = blocks[starts]
    for k in ("pos", "count", "is_gqx_defined"):
        assert np.array_equal(b[k], ref_blocks[k]), k
    gq = ref_blocks["is_gqx_defined"] == 1
    assert np.array_equal(b["gqx_min"][gq], ref_blocks["gqx_min"][gq].astype(np.int32))
    assert np.array_equal(b["dpu_min"], ref_blocks["dpu_min"].astype(np.int32))
    for k in ("dpu_mean", "dpf_mean"):  # stream_stat's running mean, bit for bit (DP / DPF are its rounding)
        assert np.array_equal(b[k].view(np.uint64), ref_blocks[k].view(np.uint64)), k
    return len(starts)


def _golden():
    g = np.load(os.path.join(GOLD, "gvcf_block_reference.npz"))
    n = len([k for k in g.files if k.startswith("sites_")])
    return [dict(sites=g["sites_%d" % i], tol=tuple(int(x) for x in g["tol_%d" % i]), kind=g["kind_%d" % i], blocks=g["blocks_%d" % i]) for i in range(n)]


def test_statement_reproduces_the_reference_golden():
    n_blocks = n_alone = 0
    for run in _golden():
        kind, blocks = capi.gvcf_block_sites(_sites(run["sites"]), *run["tol"], library=_double())
        n_blocks += _check(kind, blocks, run["kind"], run["blocks"])
        n_alone += int(np.sum(kind == 2))
    assert n_blocks > 3000 and n_alone > 300


@pytest.mark.skipif(not pyoracle.ref_available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed,tol", [(11, (30, 3)), (12, (10, 1)), (13, (50, 8)), (14, (100, 0))])
def test_statement_reproduces_the_live_reference(seed, tol):
    sites = synth.gvcf_sites(40000, np.random.default_rng(881000 + seed + SEED))
    ref_kind, ref_blocks = pyoracle.ref_gvcf_block_sites(sites, *tol)
    kind, blocks = capi.gvcf_block_sites(_sites(sites), *tol, library=_double())
    assert _check(kind, blocks, ref_kind, ref_blocks) > 2000
    assert max(ref_blocks["count"]) > 30  # long blocks as well as the many short ones


def test_edge_cases():
    D = _double()
    kind, blocks = capi.gvcf_block_sites(np.zeros(0, capi.GVCF_SITE_DTYPE), library=D)
    assert len(kind) == 0
    s = np.zeros(3, capi.GVCF_SITE_DTYPE)
    s["pos"] = [5, 6, 8]  # a gap before the third site
    s["is_compressible"] = 1
    s["gt"] = 2 << 24
    s["ploidy"] = 2
    s["used_basecalls"] = 20
    s["gqx"] = 40
    s["is_gqx"] = 1
    kind, blocks = capi.gvcf_block_sites(s, library=D)
    assert list(kind) == [1, 0, 1] and blocks["count"][0] == 2 and blocks["count"][2] == 1 and blocks["gqx_min"][0] == 40
    s["is_compressible"][1] = 0  # a site written on its own splits the block around it
    kind, blocks = capi.gvcf_block_sites(s, library=D)
    assert list(kind) == [1, 2, 1] and blocks["count"][0] == 1


@pytest.mark.gpu
def test_kernel_reproduces_the_reference_golden():
    capi.init(0)
    n = 0
    for run in _golden():
        kind, blocks = capi.gvcf_block_sites(_sites(run["sites"]), *run["tol"])
        n += _check(kind, blocks, run["kind"], run["blocks"])
    assert n > 3000


@pytest.mark.gpu
def test_kernel_equals_the_host_statement_at_size():
    capi.init(0)
    rng = np.random.default_rng(882000 + SEED)
    sites = _sites(np.concatenate([synth.gvcf_sites(1 << 18, rng) for _ in range(4)]))
    sites["pos"] = np.cumsum(np.maximum(np.diff(sites["pos"], prepend=sites["pos"][0] - 1), 1))  # (keep positions ascending across the pieces)
    kind, blocks = capi.gvcf_block_sites(sites, 30, 3)
    hk, hb = capi.gvcf_block_sites(sites, 30, 3, library=_double())
    assert np.array_equal(kind, hk)
    st = kind == 1
    assert np.array_equal(blocks[st].tobytes(), hb[st].tobytes()) and int(st.sum()) > 100000
    # every compressible site belongs to exactly one block, blocks tile their stretches
    assert int(blocks["count"][st].sum()) == int(np.sum(kind != 2))


# ---- the block that would start at every plain site (sk_gvcf_run: site 10's whole blocks) ------------------------------------------
def _run_inputs(rng, n, depth=40.0, wobble=0.08, holes=0.01):
    """sites of a sample at slowly varying depth: long runs of plain sites with a few filters changing along the way, holes (sites
    that are not plain) and depth steps"""
    base = depth * (1.0 + 0.6 * np.sin(np.arange(n) / 300.0)) * np.where((np.arange(n) // 900) % 3 == 1, 2.2, 1.0)
    used = np.maximum(0, rng.normal(base, wobble * base)).astype(np.uint32)
    unused = rng.poisson(0.05 * used + 0.2).astype(np.uint32)
    sm = np.zeros(n, capi.GVCF_SITE_SUMMARY_DTYPE)
    sm["flags"] = (rng.random(n) >= holes) & (used > 0)
    sm["gqx"] = np.minimum(99, (2.5 * used + rng.integers(0, 4, n))).astype(np.int32)
    sm["ref_fwd"] = used // 2
    sm["ref_rev"] = used - used // 2
    return sm, used, used + unused, (used + unused + rng.integers(0, 3, n)).astype(np.uint32)


def test_plain_runs_statement_equals_the_site_by_site_statement():
    """the walk that steps over whole tiles of 32 sites (csrc/gvcf_site_core.h plain_run, run on the host through the CPU double) against
    the numpy statement that joins site by site with the reference's stream_stat and check_block_tolerance: blocks of hundreds of sites"""
    L = _double()
    L.sk_init(0)
    rng = np.random.default_rng(5100 + SEED)
    keys, n_long = set(), 0
    for opt in (capi.gvcf_block_options(), capi.gvcf_block_options(is_max_depth=1, max_chrom_depth=95.0, block_percent_tol=10, block_abs_tol=1),
                capi.gvcf_block_options(block_percent_tol=50, block_abs_tol=8, min_homref_gqx=60.0)):
        sm, cc, rc, mq = _run_inputs(rng, 3000)
        got = capi.gvcf_plain_runs(sm, cc, rc, mq, opt, library=L)
        want = pyoracle.gvcf_plain_runs(sm, cc, rc, mq, opt)
        assert got.tobytes() == want.tobytes()
        assert (got["len"] == 0).sum() > 5
        n_long += int((got["len"] > 64).sum())
        keys |= set(got["filter_key"][got["len"] > 0].tolist())
    assert len(keys) >= 3 and n_long > 500


@pytest.mark.gpu
def test_plain_runs_kernel_equals_the_host_statement_at_size():
    """gvcf_site_pod_kernel + gvcf_site_tile_kernel + gvcf_plain_run_kernel against the same statement on the host, 400 000 sites"""
    capi.init(0)
    L = _double()
    L.sk_init(0)
    rng = np.random.default_rng(5200 + SEED)
    for opt, n in ((capi.gvcf_block_options(), 400000), (capi.gvcf_block_options(is_max_depth=1, max_chrom_depth=95.0, block_percent_tol=10, block_abs_tol=1), 50000)):
        sm, cc, rc, mq = _run_inputs(rng, n)
        got = capi.gvcf_plain_runs(sm, cc, rc, mq, opt)
        want = capi.gvcf_plain_runs(sm, cc, rc, mq, opt, library=L)
        assert got.tobytes() == want.tobytes()
        assert (got["len"] > 0).sum() > 0.9 * n and (n < 100000 or (got["len"] > 256).sum() > 1000)
