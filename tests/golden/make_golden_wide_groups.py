#!/usr/bin/env python
"""Generate tests/golden/allele_group_wide_reference.npz: the REFERENCE's own getVariantAlleleGroupGenotypeLhoodsForSample
(oracle/_ref/libstrelka_ref.so, ref_allele_group_genotype_lhoods) on seeded allele groups of 4..8 alternate alleles -- the groups a
multi-sample run forms (selectTopOrthogonalAllelesInAllSamples) -- and, with the argument `xwide`, allele_group_xwide_reference.npz: groups of
9..16 (runs of five to eight samples).  Run in the build container: python tests/golden/make_golden_wide_groups.py [xwide]"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from strelka_amd import capi, synth  # noqa: E402


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def reference_lhoods(ab):
    """-> (lhood [n][45], counts [n][2][10]) from the reference's function, group by group"""
    R = pyoracle.ref()
    assert R is not None, "oracle/_ref/libstrelka_ref.so missing"
    W = ab.width
    glh = np.zeros((ab.n_groups, (W + 1) * (W + 2) // 2))
    gcnt = np.zeros((ab.n_groups, 2, W + 2), np.uint32)
    for g in range(ab.n_groups):
        s, e = int(ab.read_off[g]), int(ab.read_off[g + 1])
        A, pl = int(ab.n_alt[g]), int(ab.ploidy[g])
        G = A + 1 if pl == 1 else (A + 1) * (A + 2) // 2
        refl = np.ascontiguousarray(ab.ref_lnp[s:e, :A])
        al = np.ascontiguousarray(ab.allele_lnp[s:e, :A])
        t1 = np.ascontiguousarray(ab.read_flags[s:e] & 1)
        fw = np.ascontiguousarray((ab.read_flags[s:e] >> 1) & 1)
        # (distinct keys: the group is a std::map over IndelKey -- an insertion's sequence tells two alleles of one length apart)
        inss = (C.c_char_p * A)(*[("ACGT"[k % 4] * int(ab.ins_len[g, k])).encode() for k in range(A)])
        dl = np.ascontiguousarray(ab.del_len[g, :A])
        ol, oc = np.zeros(G), np.zeros(2 * (A + 2), np.uint32)
        R.ref_allele_group_genotype_lhoods(e - s, A, _p(refl), _p(al), _p(np.ascontiguousarray(ab.non_ambig[s:e])),
                                           _p(np.ascontiguousarray(ab.read_length[s:e])), _p(t1), _p(fw), _p(dl), inss, pl,
                                           5, C.c_double(0.25), _p(ol), _p(oc))
        glh[g, :G] = ol
        oc = oc.reshape(2, A + 2)
        gcnt[g, :, :A + 1] = oc[:, :A + 1]
        gcnt[g, :, A + 1] = oc[:, A + 1]
    return glh, gcnt


def main():
    pyoracle.build(ref=True)
    # "xwide": groups of 9..16 alternate alleles (runs of five to eight samples), fewer and shallower (153 genotypes a group)
    which = sys.argv[1] if len(sys.argv) > 1 else "wide"
    if which == "xwide":
        rng = np.random.default_rng(20261001)
        ab = synth.allele_group_batch(40, rng, depth_mean=35.0, min_alt=9, max_alt=capi.MAX_ALT_XWIDE, missing_rate=0.01)
    else:
        rng = np.random.default_rng(20260927)
        ab = synth.allele_group_batch(90, rng, depth_mean=45.0, min_alt=4, max_alt=capi.MAX_ALT_WIDE, missing_rate=0.01)
    glh, gcnt = reference_lhoods(ab)
    name = "allele_group_%s_reference.npz" % which
    np.savez_compressed(os.path.join(HERE, name), a_read_off=ab.read_off, a_n_alt=ab.n_alt, a_ploidy=ab.ploidy,
                        a_del=ab.del_len, a_ins=ab.ins_len, a_ref=ab.ref_lnp, a_allele=ab.allele_lnp, a_na=ab.non_ambig, a_rl=ab.read_length,
                        a_flags=ab.read_flags, a_lhood=glh, a_counts=gcnt)
    print("wrote %s: %d groups, n_alt %d..%d, reads used in some group: %s" % (
        name, ab.n_groups, ab.n_alt.min(), ab.n_alt.max(), bool(np.any(glh != 0))))


if __name__ == "__main__":
    main()
