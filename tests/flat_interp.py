"""Test-only interpreter of a flattened sk_align_batch (pure Python doubles, sequential adds): lets the CPU suite check
the host adapter's integer/byte work (which base is compared with which) without a GPU.  Not part of the product."""
import math

import numpy as np


def score_flat(batch, q2lncompe, q2lne):
    lnthird = -math.log(3.0)
    ln_quarter = math.log(0.25)
    ln_noncand = math.log(1e-5)
    out = np.zeros(batch.n_cals)
    for r in range(batch.n_reads):
        ro, ho = int(batch.read_off[r]), int(batch.hap_off[r])
        for c in range(int(batch.cal_off[r]), int(batch.cal_off[r + 1])):
            lnp = 0.0
            rp = 0
            for k in range(int(batch.op_off[c]), int(batch.op_off[c + 1])):
                op = batch.ops[k]
                ln, kind, flags, src = int(op["length"]), int(op["kind"]), int(op["flags"]), int(op["src"])
                if kind == 0:
                    for j in range(ln):
                        rc = int(batch.read_code[ro + rp + j])
                        if rc == 15:
                            continue
                        q = int(batch.read_qual[ro + rp + j])
                        is_ref = rc == 0 or rc == int(batch.hap_code[ho + src + j])
                        lnp += float(q2lncompe[q]) if is_ref else float(q2lne[q]) + lnthird
                    rp += ln
                elif kind == 1:
                    lnp += ln * ln_quarter
                    rp += ln
                if flags & 1:
                    lnp += ln_noncand
            out[c] = lnp
    return out
