"""Kernel V2 (gvcf_site_summary_kernel over strelka_amd/csrc/gvcf_site_core.h: "is this position a plain hom-ref site?", GQX, the
counts the writer's block reads) against the REFERENCE's own answers: tests/golden/gvcf_site_reference.npz holds, for every position
of three seeded samples, what the unmodified starling2 printed with a no-compress region over everything (one gVCF record per
position: tests/golden/make_gvcf_site_golden.py).  The drop-in calls the same samples with $STRELKA_AMD_GVCF_SITE_DUMP set and
writes down every position's summary as its window brought it -- on the GPU the kernel's, over the CPU double the statement the
kernel runs (gvcf_site_core.h is one statement for host and device).  Position by position:

    plain site  <=>  the reference printed a hom-ref site without an alternate allele and with used basecalls;
    then GQX, DP (= the cleaned column's size) and DPF (= raw - cleaned) are the reference's.

(the unit oracle of V2 was a numpy statement of the builder's own; the end-to-end runs pinned it only transitively.)"""
import os
import subprocess

import numpy as np
import pytest

from tests import e2e_util as E
from tests.golden import make_gvcf_site_golden as G

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gvcf_site_reference.npz")


def _check(variant, tmp_path):
    gold = np.load(GOLD)
    checked = 0
    for i, s in enumerate(G.SAMPLES):
        d = str(tmp_path / ("sample%d" % i))
        G.make_sample(d, s)
        out = str(tmp_path / ("out%d" % i))
        os.makedirs(out)
        dump = str(tmp_path / ("dump%d.txt" % i))
        # (the drop-in compresses as usual: what is dumped is the window's summary of every position, whatever the writer then does)
        cmd = E.germline_wgs_argv("starling2_" + variant, out + "/", [os.path.join(d, "wgs.bam")], ["chrW:1-%d" % s["length"]], os.path.join(d, "wgs.fa"),
                                  os.path.join(d, "chrom_depth.txt"))
        p = subprocess.run(cmd, env=dict(os.environ, STRELKA_AMD_GVCF_SITE_DUMP=dump), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        got = np.loadtxt(dump, dtype=np.int64).reshape(-1, 7)  # pos flags gqx ref_fwd ref_rev clean raw
        by_pos = {int(r[0]): r for r in got}
        want = gold["sites_%d" % i]
        n_plain = n_other_ploidy = 0
        for pos0, is_homref, gqx, dp, dpf, _ft, diploid in want:
            r = by_pos.get(int(pos0))
            if r is None:
                assert dp + dpf == 0, ("the reference saw basecalls at a position no window covered", i, int(pos0))
                continue
            assert (int(r[5]), int(r[6] - r[5])) == (dp, dpf), (i, int(pos0), r.tolist(), (dp, dpf))  # the window's counts are the reference's DP / DPF
            if not diploid:
                # under a called deletion the reference genotypes the position with a lowered ploidy; the window's summary is for
                # ploidy 2 (the adapter checks spanningIndelPloidyModification before it uses one): nothing to compare
                n_other_ploidy += 1
                continue
            plain = bool(r[1] & 1)
            assert plain == bool(is_homref and dp > 0), (i, int(pos0), r.tolist(), (is_homref, gqx, dp, dpf))
            if plain:
                n_plain += 1
                assert int(r[2]) == gqx, (i, int(pos0), r.tolist(), gqx)
                assert int(r[3] + r[4]) <= dp  # (the reference bases among the used calls; printed by the reference only at variant sites)
        assert n_plain > 0.9 * len(want) and n_other_ploidy < 0.02 * len(want)
        checked += n_plain
    return checked


@pytest.mark.skipif(not E.have("starling2_dbl"), reason="oracle/_ref binaries not built")
def test_site_summaries_equal_the_references_own_records_cpu(tmp_path):
    assert _check("dbl", tmp_path) > 90000


@pytest.mark.gpu
@pytest.mark.skipif(not E.have("starling2_amd"), reason="oracle/_ref binaries not built")
def test_site_summaries_equal_the_references_own_records_gpu(tmp_path):
    assert _check("amd", tmp_path) > 90000
