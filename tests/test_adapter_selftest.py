"""The adapter's host-side restatements against the reference functions they stand in for (adapter/selftest.cpp).

The adapter replaces a few per-base / per-position host loops of the reference that run beside the routed sites
(get_valid_alignment_range, ReferenceRepeatFinder::updateRepeatSpan, add_alignment_to_depth_buffer, the active-region
match / mismatch bookkeeping, checkBamRecord's validity loops) with cheaper statements of the same arithmetic.  The end-to-end
tests check them through whole VCFs; this program drives each one directly against the reference's own function -- linked from
the reference's objects -- on seeded random inputs that hold what the synthetic genomes do not: N runs in the reference, segments
that start at position 0, odd packed offsets, leading / trailing indels and clips, ring buffers that wrap.  CPU only; the program
is built by `make -C adapter double` where /root/reference is present and travels prebuilt with oracle/_ref/."""
import os
import subprocess

import pytest

from tests import e2e_util as E

PROGRAM = os.path.join(E.BIN_DIR, "adapter_selftest")


@pytest.mark.skipif(not os.path.exists(PROGRAM), reason="oracle/_ref/bin/adapter_selftest not built (needs /root/reference)")
def test_host_restatements_match_reference(built):
    p = subprocess.run([PROGRAM], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "adapter selftest: all passed" in out and "FAIL" not in out, out
    # every comparison ran on a non-trivial sample
    for name in ("valid_alignment_range", "repeat_span_update", "depth_buffer_add_alignment", "active_region_insert_aligned_segment",
                 "is_plain_bam_record"):
        assert name + ":" in out, out
