"""the pileup stream's window as the adapter pushes it (2 200 reads at 40x, genotypes + site summaries + runs + EVS words), alone, for a kernel
trace: which launches a window's device time is made of
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o kt -- python tools/diag/stream_window_trace.py"""
import sys

import numpy as np

sys.path.insert(0, ".")
from strelka_amd import capi, synth

capi.init(0)
sb, sb_loci = synth.pileup_reads_flat(1 << 16, np.random.default_rng(77))
stream = capi.PileupStream(capi.pileup_options(report_begin=0, report_end=sb_loci + 200), capi.germline_options(), evs_words=True,
                           gvcf_block_opt=capi.gvcf_block_options(is_max_depth=1, max_chrom_depth=120.0, min_homref_gqx=15.0))
win_reads = 2200
subs = []
for lo in range(0, sb.n_reads, win_reads):
    hi = min(sb.n_reads, lo + win_reads)
    subs.append((synth.ReadBatch(sb.read_off[lo:hi + 1] - sb.read_off[lo], sb.read_code[sb.read_off[lo]:sb.read_off[hi]],
                                 sb.read_qual[sb.read_off[lo]:sb.read_off[hi]], sb.path_off[lo:hi + 1] - sb.path_off[lo],
                                 sb.path[sb.path_off[lo]:sb.path_off[hi]], sb.pos[lo:hi], sb.is_fwd[lo:hi], sb.mapq[lo:hi],
                                 sb.map_level[lo:hi], "", 0), int(sb.pos[hi]) if hi < sb.n_reads else 2**31 - 1))
import time
for rep in range(4):
    stream.begin_region(sb.ref_seq, 0, 0, sb_loci + 200)
    t0 = time.perf_counter()
    for sub, final_to in subs:
        stream.push_raw(sub, final_to)
    dt = time.perf_counter() - t0
    print("rep %d: %d windows, %.3f ms per window" % (rep, len(subs), dt / len(subs) * 1e3), flush=True)
stream.close()
