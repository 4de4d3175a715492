"""ctypes loader of the CHECKERS: oracle/libstrelka_oracle.so (plain-C restatement) and, when present,
oracle/_ref/libstrelka_ref.so (the reference's own translation units).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
strelka_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libstrelka_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libstrelka_ref.so")
vp = C.c_void_p


def build(ref=True, quiet=True):
    """(Re)build the checkers.  `make ref` is a no-op when /root/reference is absent."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True, stdout=out)
    if ref and os.path.isdir("/root/reference/src/c++/lib"):
        subprocess.run(["make", "-s", "-C", HERE, "-j8", "ref"], check=True, stdout=out)


class PathSeg(C.Structure):
    _fields_ = [("type", C.c_uint32), ("length", C.c_uint32)]


class Indel(C.Structure):
    _fields_ = [("pos", C.c_int32), ("type", C.c_int32), ("del_len", C.c_uint32), ("ins_len", C.c_uint32),
                ("ins_seq", C.c_char_p), ("is_candidate", C.c_int32)]


class Cal(C.Structure):
    _fields_ = [("pos", C.c_int32), ("n_seg", C.c_int32), ("path", C.POINTER(PathSeg)), ("n_indels", C.c_int32),
                ("indels", C.POINTER(Indel)), ("leading", Indel), ("trailing", Indel)]


class ReadCase(C.Structure):
    _fields_ = [("read_code", vp), ("read_qual", vp), ("read_len", C.c_int32), ("ref_seq", C.c_char_p),
                ("ref_offset", C.c_int32), ("ref_len", C.c_int32), ("cals", C.POINTER(Cal)), ("n_cals", C.c_int32)]


class GermlineOptions(C.Structure):
    _fields_ = [("bsnp_diploid_theta", C.c_double), ("bsnp_ssd_no_mismatch", C.c_double),
                ("bsnp_ssd_one_mismatch", C.c_double), ("is_min_vexp", C.c_int32), ("min_vexp", C.c_double)]


class SomaticSnvOptions(C.Structure):
    _fields_ = [("bsnp_diploid_theta", C.c_double), ("somatic_snv_rate", C.c_double),
                ("shared_site_error_rate", C.c_double), ("shared_site_error_strand_bias_fraction", C.c_double),
                ("ssnv_contam_tolerance", C.c_double)]


def germline_options():
    return GermlineOptions(0.001, 0.35, 0.6, 1, 0.25)


def somatic_snv_options():
    return SomaticSnvOptions(0.001, 1e-4, 5e-10, 0.0, 0.15)


DIGT_RS_DTYPE = np.dtype([("ref_pprob", "<f8"), ("max_gt", "<u4"), ("snp_qphred", "<i4"), ("max_gt_qphred", "<i4"),
                          ("_pad", "<i4")])
DIGT_CALL_DTYPE = np.dtype([("lhood", "<f4", (10,)), ("phredLoghood", "<u4", (10,)), ("genome", DIGT_RS_DTYPE),
                            ("poly", DIGT_RS_DTYPE), ("strand_bias", "<f8"), ("ref_gt", "<u4"), ("is_called", "<u4")])
SOMATIC_CALL_DTYPE = np.dtype([("normal_lhood", "<f4", (30,)), ("tumor_lhood", "<f4", (30,)), ("max_gt", "<u4"),
                               ("qphred", "<i4"), ("from_ntype_qphred", "<i4"), ("ntype", "<u4"),
                               ("strand_bias", "<f4"), ("is_called", "<u4"), ("normal_alt_id", "<u4"),
                               ("tumor_alt_id", "<u4")])

_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        L = C.CDLL(ORACLE_SO)
        L.sko_score_candidate_alignment.restype = C.c_double
        L.sko_log_sum2.restype = C.c_double
        L.sko_log_sum2.argtypes = [C.c_double, C.c_double]
        L.sko_log_sum2f.restype = C.c_float
        L.sko_log_sum2f.argtypes = [C.c_float, C.c_float]
        L.sko_error_prob_to_qphred.argtypes = [C.c_double]
        L.sko_ln_error_prob_to_qphred_f.argtypes = [C.c_float]
        L.sko_germline_lnpriors.argtypes = [C.c_double, vp]
        _oracle = L
    return _oracle


def ref_available():
    return os.path.exists(REF_SO)


def ref():
    """The reference's own code; None when oracle/_ref/libstrelka_ref.so has not been built."""
    global _ref
    if _ref is None and ref_available():
        L = C.CDLL(REF_SO)
        L.ref_pack_base_call.restype = C.c_uint16
        L.ref_log_sum2.restype = C.c_double
        L.ref_log_sum2.argtypes = [C.c_double, C.c_double]
        L.ref_log_sum2f.restype = C.c_float
        L.ref_log_sum2f.argtypes = [C.c_float, C.c_float]
        L.ref_error_prob_to_qphred.argtypes = [C.c_double]
        L.ref_ln_error_prob_to_qphred_f.argtypes = [C.c_float]
        L.ref_germline_lnpriors.argtypes = [C.c_double, vp]
        _ref = L
    return _ref


def _p(a):
    return None if a is None else a.ctypes.data_as(vp)


# ----------------------------------------------------------------------------------------------------------------------
# hot path A

class MarshalledCases:
    """ctypes image of a list of synth.align_cases() dicts; keeps every buffer alive."""

    def __init__(self, cases):
        self.keep = []
        self.n_cals = sum(len(c["cals"]) for c in cases)
        self.arr = (ReadCase * max(len(cases), 1))()
        self.n = len(cases)
        for i, c in enumerate(cases):
            rc = np.ascontiguousarray(c["read_code"], np.uint8)
            rq = np.ascontiguousarray(c["read_qual"], np.uint8)
            ref_b = c["ref_seq"].encode() if isinstance(c["ref_seq"], str) else bytes(c["ref_seq"])
            cals = (Cal * max(len(c["cals"]), 1))()
            for j, cal in enumerate(c["cals"]):
                path = (PathSeg * max(len(cal["path"]), 1))(*[PathSeg(t, l) for t, l in cal["path"]])
                ind = (Indel * max(len(cal["indels"]), 1))(*[self._key(k) for k in cal["indels"]])
                self.keep += [path, ind]
                cals[j] = Cal(cal["pos"], len(cal["path"]), path, len(cal["indels"]), ind, self._key(cal.get("leading")),
                              self._key(cal.get("trailing")))
            self.keep += [rc, rq, ref_b, cals]
            self.arr[i] = ReadCase(_p(rc), _p(rq), len(rc), ref_b, c["ref_offset"], len(ref_b), cals, len(c["cals"]))

    def _key(self, k):
        if k is None:
            return Indel(0, 0, 0, 0, None, 0)
        seq = k.get("ins_seq", "").encode()
        self.keep.append(seq)
        return Indel(k["pos"], k["type"], k.get("del_len", 0), len(seq), seq, int(k.get("is_candidate", 1)))


def score_cases(cases):
    """ln P(read|alignment) of every candidate alignment of every case (concatenated), via the C restatement."""
    m = cases if isinstance(cases, MarshalledCases) else MarshalledCases(cases)
    out = np.zeros(m.n_cals, np.float64)
    oracle().sko_score_cases(m.arr, m.n, _p(out))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# hot path B

def adjust_joint_eprob(batch, opt=None):
    opt = opt or germline_options()
    de = np.zeros(len(batch.calls), np.float32)
    oracle().sko_adjust_joint_eprob_batch(_p(batch.call_off), _p(batch.calls), batch.n_loci, C.byref(opt), _p(de))
    return de


def site_digt_call(batch, de, opt=None):
    opt = opt or germline_options()
    out = np.zeros(batch.n_loci, DIGT_CALL_DTYPE)
    de = np.ascontiguousarray(de, np.float32)
    oracle().sko_site_digt_call_batch(_p(batch.call_off), _p(batch.calls), _p(de), _p(batch.ref_base),
                                      _p(batch.ploidy), batch.n_loci, C.byref(opt), _p(out))
    return out


def somatic_snv_call(normal, tumor, opt=None, is_forced_output=False):
    opt = opt or somatic_snv_options()
    out = np.zeros(normal.n_loci, SOMATIC_CALL_DTYPE)
    oracle().sko_somatic_snv_call_batch(_p(normal.call_off), _p(normal.calls), _p(tumor.call_off), _p(tumor.calls),
                                        _p(normal.ref_base), normal.n_loci, C.byref(opt), int(is_forced_output), _p(out))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# hot path B, indels

def _genotype_dtype():
    from strelka_amd import capi
    return capi.SOMATIC_GENOTYPE_DTYPE


def somatic_snv_call_tiers(n1, t1, n2=None, t2=None, opt=None, is_forced_output=None, is_compute_nonsomatic=False,
                           use_reference=False):
    """the whole of position_somatic_snv_call per locus: the C restatement (sko_position_somatic_snv_call_tiers) or, with
    use_reference, the reference's own function through oracle/_ref (ref_position_somatic_snv_call)."""
    opt = opt or somatic_snv_options()
    n = n1.n_loci
    out = np.zeros(n, _genotype_dtype())
    is_tier2 = n2 is not None
    if not is_tier2:
        n2, t2 = n1, t1
    L = ref() if use_reference else oracle()
    for l in range(n):
        def rng_(b):
            s, e = int(b.call_off[l]), int(b.call_off[l + 1])
            c = np.ascontiguousarray(b.calls[s:e])
            return c, e - s
        c1, k1 = rng_(n1)
        c2, k2 = rng_(t1)
        c3, k3 = rng_(n2)
        c4, k4 = rng_(t2)
        forced = 0 if is_forced_output is None else int(is_forced_output[l])
        rec = np.zeros(1, _genotype_dtype())
        rb = int(n1.ref_base[l])
        if use_reference:
            rc = L.ref_position_somatic_snv_call(_p(c1), k1, _p(c2), k2, _p(c3), k3, _p(c4), k4, int(is_tier2),
                                                 C.c_char(b"ACGTN"[min(rb, 4):min(rb, 4) + 1]), C.byref(opt), forced,
                                                 int(bool(is_compute_nonsomatic)), _p(rec))
            assert rc == 0
        else:
            L.sko_position_somatic_snv_call_tiers(_p(c1), k1, _p(c2), k2, _p(c3), k3, _p(c4), k4, int(is_tier2), rb,
                                                  C.byref(opt), forced, int(bool(is_compute_nonsomatic)), _p(rec))
        out[l] = rec[0]
    return out


class IndelSampleReads(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("ref_lnp", C.c_void_p), ("indel_lnp", C.c_void_p), ("alt_key", C.c_void_p),
                ("alt_lnp", C.c_void_p), ("non_ambig", C.c_void_p), ("read_length", C.c_void_p), ("is_tier1", C.c_void_p)]


class AltKey(C.Structure):
    _fields_ = [("begin_pos", C.c_int32), ("end_pos", C.c_int32), ("is_mismatch", C.c_int32)]


class SomaticIndelParams(C.Structure):
    _fields_ = [("normal_min_read_bp_flank", C.c_int32), ("tumor_min_read_bp_flank", C.c_int32),
                ("random_base_match_prob", C.c_double), ("tier2_random_base_match_prob", C.c_double),
                ("use_tier2_evidence", C.c_int32), ("is_use_alt_indel", C.c_int32), ("bindel_diploid_theta", C.c_double),
                ("somatic_indel_rate", C.c_double), ("shared_indel_error_factor", C.c_double),
                ("indel_contam_tolerance", C.c_double)]


def somatic_indel_params(**kw):
    # strelka_options defaults with the workflow's overrides (configureStrelkaSomaticWorkflow.py.ini)
    p = SomaticIndelParams(1, 5, 0.5, 0.25, 1, 1, 1e-4, 1e-6, 2.2, 0.15)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


SOMATIC_INDEL_GENOTYPE_DTYPE = np.dtype([("sindel_tier", np.uint8), ("sindel_from_ntype_tier", np.uint8),
                                         ("is_forced_output", np.uint8), ("is_overlap", np.uint8), ("ntype", np.uint32),
                                         ("max_gt", np.uint32), ("qphred", np.int32), ("from_ntype_qphred", np.int32)])
assert SOMATIC_INDEL_GENOTYPE_DTYPE.itemsize == 20


def _sample_struct(smp, keep):
    arrs = [np.ascontiguousarray(smp["ref_lnp"], np.float32), np.ascontiguousarray(smp["indel_lnp"], np.float32),
            np.ascontiguousarray(smp["alt_key"], np.int32), np.ascontiguousarray(smp["alt_lnp"], np.float32),
            np.ascontiguousarray(smp["non_ambig"], np.uint16), np.ascontiguousarray(smp["read_length"], np.uint16),
            np.ascontiguousarray(smp["is_tier1"], np.uint8)]
    keep.extend(arrs)
    return IndelSampleReads(len(arrs[0]), *[a.ctypes.data for a in arrs])


def get_somatic_indel(cases, params=None, use_reference=False):
    """cases: list of dicts(normal=sample, tumor=sample, alt_keys=[(begin,end,is_mismatch)], del_len, ins_len, forced,
    indel_to_ref_error_prob) with sample = dict of per-read arrays.  The whole of get_somatic_indel per case: the C
    restatement, or with use_reference the reference's own function (which also reports the error rate it used)."""
    params = params or somatic_indel_params()
    out = np.zeros(len(cases), SOMATIC_INDEL_GENOTYPE_DTYPE)
    used = np.zeros(len(cases))
    L = ref() if use_reference else oracle()
    for i, c in enumerate(cases):
        keep = []
        ns, ts = _sample_struct(c["normal"], keep), _sample_struct(c["tumor"], keep)
        ak = (AltKey * max(1, len(c["alt_keys"])))(*[AltKey(*k) for k in c["alt_keys"]])
        rec = np.zeros(1, SOMATIC_INDEL_GENOTYPE_DTYPE)
        if use_reference:
            u = C.c_double(0)
            rc = L.ref_get_somatic_indel(C.byref(ns), C.byref(ts), ak, len(c["alt_keys"]), int(c["del_len"]), int(c["ins_len"]),
                                         C.byref(params), C.c_double(c["indel_to_ref_error_prob"]), int(c["forced"]), _p(rec),
                                         C.byref(u))
            assert rc == 0
            used[i] = u.value
        else:
            L.sko_get_somatic_indel(C.byref(ns), C.byref(ts), ak, len(c["alt_keys"]), int(c["del_len"]), int(c["ins_len"]), 0,
                                    C.byref(params), C.c_double(c["indel_to_ref_error_prob"]), int(c["forced"]), _p(rec))
            used[i] = c["indel_to_ref_error_prob"]
        out[i] = rec[0]
    return out, used


def indel_grid_lhood(batch, min_read_bp_flank, random_base_match_prob, is_include_tier2, is_use_alt_indel=True):
    """batch: strelka_amd.capi.HostReadScoreBatch -> float64 [n_indels][21] (pass the EFFECTIVE random-base-match
    probability of the pass: the tier2 value for tier2 passes)."""
    out = np.zeros((batch.n_indels, 21))
    alt = batch.alt_lnp if batch.alt_lnp is not None else np.full(len(batch.ref_lnp), np.nan, np.float32)
    for i in range(batch.n_indels):
        s, e = int(batch.read_off[i]), int(batch.read_off[i + 1])
        t1 = np.ascontiguousarray(batch.read_flags[s:e] & 1)
        bp = 0 if batch.is_breakpoint is None else int(batch.is_breakpoint[i])
        args = [np.ascontiguousarray(x[s:e]) for x in (batch.ref_lnp, batch.indel_lnp, alt, batch.non_ambig, batch.read_length)]
        row = np.zeros(21)
        oracle().sko_indel_grid_lhood(e - s, *[_p(x) for x in args], _p(t1), int(batch.del_len[i]), int(batch.ins_len[i]),
                                      bp, int(min_read_bp_flank), C.c_double(random_base_match_prob),
                                      int(is_include_tier2), int(is_use_alt_indel), _p(row))
        out[i] = row
    return out


def somatic_indel_result(normal_lhood, tumor_lhood, indel_to_ref_error_prob, shared_indel_error_factor=2.2,
                         indel_contam_tolerance=0.15, somatic_indel_rate=1e-6, bindel_diploid_theta=1e-4):
    n = len(normal_lhood)
    out = np.zeros(n, np.dtype([("max_gt", "<u4"), ("qphred", "<i4"), ("from_ntype_qphred", "<i4"), ("ntype", "<u4")]))
    for i in range(n):
        mg, q, fq, nt = C.c_uint32(), C.c_int32(), C.c_int32(), C.c_uint32()
        nl = np.ascontiguousarray(normal_lhood[i], np.float64)
        tl = np.ascontiguousarray(tumor_lhood[i], np.float64)
        oracle().sko_somatic_indel_result(_p(nl), _p(tl), C.c_double(indel_to_ref_error_prob[i]),
                                          C.c_double(shared_indel_error_factor), C.c_double(indel_contam_tolerance),
                                          C.c_double(somatic_indel_rate), C.c_double(bindel_diploid_theta), C.byref(mg),
                                          C.byref(q), C.byref(fq), C.byref(nt))
        out[i] = (mg.value, q.value, fq.value, nt.value)
    return out


def allele_group_genotype_lhoods(batch, min_read_bp_flank=5, random_base_match_prob=0.25, threshold=0.51):
    """batch: strelka_amd.capi.HostAlleleGroupBatch -> (lhood [n][10], counts [n][2][5], n_genotypes [n]); for a wide batch (rows of 8
    alternate alleles) [n][45] and [n][2][10]"""
    n = batch.n_groups
    width = getattr(batch, "width", 3)
    lh = np.zeros((n, (width + 1) * (width + 2) // 2))
    counts = np.zeros((n, 2, width + 2), np.uint32)
    ng = np.zeros(n, np.uint32)
    for g in range(n):
        s, e = int(batch.read_off[g]), int(batch.read_off[g + 1])
        A = int(batch.n_alt[g])
        pl = int(batch.ploidy[g])
        G = A + 1 if pl == 1 else (A + 1) * (A + 2) // 2
        refl = np.ascontiguousarray(batch.ref_lnp[s:e, :A])
        al = np.ascontiguousarray(batch.allele_lnp[s:e, :A])
        t1 = np.ascontiguousarray(batch.read_flags[s:e] & 1)
        fw = np.ascontiguousarray((batch.read_flags[s:e] >> 1) & 1)
        na = np.ascontiguousarray(batch.non_ambig[s:e])
        rl = np.ascontiguousarray(batch.read_length[s:e])
        dl = np.ascontiguousarray(batch.del_len[g, :A])
        il = np.ascontiguousarray(batch.ins_len[g, :A])
        ol = np.zeros(G)
        oc = np.zeros(2 * (A + 2), np.uint32)
        oracle().sko_allele_group_genotype_lhoods(e - s, A, _p(refl), _p(al), _p(na), _p(rl), _p(t1), _p(fw), _p(dl), _p(il),
                                                  pl, int(min_read_bp_flank), C.c_double(random_base_match_prob),
                                                  C.c_double(threshold), _p(ol), _p(oc))
        lh[g, :G] = ol
        oc = oc.reshape(2, A + 2)
        counts[g, :, :A + 1] = oc[:, :A + 1]
        counts[g, :, A + 1] = oc[:, A + 1]
        ng[g] = G
    return lh, counts, ng


# ----------------------------------------------------------------------------------------------------------------------
# the reference's own hot path A (oracle/_ref), for pinning the restatement

class RefIndel(C.Structure):
    _fields_ = [("pos", C.c_int32), ("type", C.c_int32), ("del_len", C.c_uint32), ("ins_len", C.c_uint32),
                ("ins_seq", C.c_char_p), ("is_candidate", C.c_int32)]


class RefCal(C.Structure):
    _fields_ = [("pos", C.c_int32), ("n_seg", C.c_int32), ("path", C.POINTER(PathSeg)), ("n_indels", C.c_int32),
                ("indels", C.POINTER(RefIndel)), ("leading", RefIndel), ("trailing", RefIndel)]


_CODE2CHAR = {0: "=", 1: "A", 2: "C", 4: "G", 8: "T", 15: "N"}


def _ref_indel(k):
    if k is None:
        return RefIndel(0, 0, 0, 0, None, 0)
    seq = k.get("ins_seq", "").encode()
    return RefIndel(k["pos"], k["type"], k.get("del_len", 0), len(seq), seq, int(k.get("is_candidate", 1)))


def ref_score_cases(cases, is_somatic=False):
    """scoreCandidateAlignment of the REFERENCE for every candidate alignment of synth.align_cases()-style cases.
    Each case gets its own IndelBuffer session holding exactly the indels its candidates mention."""
    L = ref()
    L.ref_session_create.restype = vp
    L.ref_session_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.ref_session_destroy.argtypes = [vp]
    L.ref_session_add_indel.argtypes = [vp, C.POINTER(RefIndel)]
    L.ref_session_score_cal.argtypes = [vp, C.c_char_p, vp, C.c_int, C.POINTER(RefCal), C.POINTER(C.c_double)]
    out = []
    for c in cases:
        ref_b = c["ref_seq"].encode() if isinstance(c["ref_seq"], str) else bytes(c["ref_seq"])
        s = L.ref_session_create(ref_b, int(c["ref_offset"]), int(is_somatic))
        try:
            seen = set()
            for cal in c["cals"]:
                for k in list(cal["indels"]) + [cal.get("leading"), cal.get("trailing")]:
                    if k is None:
                        continue
                    ident = (k["pos"], k["type"], k.get("del_len", 0), k.get("ins_seq", ""))
                    if ident in seen:
                        continue
                    seen.add(ident)
                    ri = _ref_indel(k)
                    if L.ref_session_add_indel(s, C.byref(ri)) != 0:
                        raise RuntimeError("reference rejected indel %r" % (ident,))
            read = "".join(_CODE2CHAR.get(int(x), "N") for x in c["read_code"]).encode()
            qual = np.ascontiguousarray(c["read_qual"], np.uint8)
            for cal in c["cals"]:
                path = (PathSeg * max(len(cal["path"]), 1))(*[PathSeg(t, l) for t, l in cal["path"]])
                ind = (RefIndel * max(len(cal["indels"]), 1))(*[_ref_indel(k) for k in cal["indels"]])
                rc = RefCal(cal["pos"], len(cal["path"]), path, len(cal["indels"]), ind, _ref_indel(cal.get("leading")),
                            _ref_indel(cal.get("trailing")))
                v = C.c_double()
                if L.ref_session_score_cal(s, read, _p(qual), len(qual), C.byref(rc), C.byref(v)) != 0:
                    raise RuntimeError("reference threw while scoring %r" % (cal,))
                out.append(v.value)
        finally:
            L.ref_session_destroy(s)
    return np.array(out)


class RefReadScore(C.Structure):
    _fields_ = [("pos", C.c_int32), ("type", C.c_int32), ("del_len", C.c_uint32), ("ins_len", C.c_uint32),
                ("ins", C.c_char * 64), ("ref_lnp", C.c_float), ("indel_lnp", C.c_float), ("non_ambig", C.c_uint16),
                ("read_length", C.c_uint16), ("is_tier1_read", C.c_int32), ("is_fwd_strand", C.c_int32),
                ("read_pos", C.c_int32), ("edge_dist", C.c_int32), ("n_alt", C.c_int32), ("alt_pos", C.c_int32 * 2),
                ("alt_type", C.c_int32 * 2), ("alt_del_len", C.c_uint32 * 2), ("alt_ins", (C.c_char * 64) * 2),
                ("alt_lnp", C.c_float * 2), ("is_suboverlap", C.c_int32)]


def ref_realign_scenarios(scenarios, is_somatic=False):
    """realignAndScoreRead of the REFERENCE on synth.realign_scenarios()-style input (one IndelBuffer session per
    scenario).  Fills each indel's log error rates (`r2i`, `i2r`) as the reference's IndelBuffer computed them and
    returns, per scenario, the list of per-read results:
    dict(threw, is_realigned, pos, cigar, scores=[dict(key=(pos,type,del_len,ins), ...)], suboverlap=[key...])."""
    L = ref()
    L.ref_session_create.restype = vp
    L.ref_session_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.ref_session_destroy.argtypes = [vp]
    L.ref_session_add_indel.argtypes = [vp, C.POINTER(RefIndel)]
    L.ref_session_add_indel_observed.argtypes = [vp, C.POINTER(RefIndel), C.c_uint]
    L.ref_session_set_indel_haplotype.argtypes = [vp, C.POINTER(RefIndel)] + [C.c_int] * 5
    L.ref_session_indel_log_error_rates.argtypes = [vp, C.POINTER(RefIndel), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ref_session_realign.argtypes = [vp, C.c_char_p, vp, C.c_int, C.c_int, C.c_int, C.POINTER(PathSeg), C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.c_char_p, C.c_int]
    L.ref_session_read_scores.argtypes = [vp, C.c_uint, C.POINTER(RefReadScore), C.c_int]
    out = []
    for sc in scenarios:
        s = L.ref_session_create(sc["ref_seq"].encode(), int(sc["ref_offset"]), int(is_somatic))
        try:
            for d in sc["indels"]:
                ri = _ref_indel(d)
                if L.ref_session_add_indel(s, C.byref(ri)) != 0:
                    raise RuntimeError("reference rejected indel %r" % (d,))
            for rid, rd in enumerate(sc["reads"]):
                for o in rd["observed"]:
                    ri = _ref_indel(sc["indels"][o])
                    if L.ref_session_add_indel_observed(s, C.byref(ri), rid + 1) != 0:
                        raise RuntimeError("reference rejected indel observation")
            for d in sc["indels"]:
                ri = _ref_indel(d)
                a, b = C.c_double(), C.c_double()
                if L.ref_session_indel_log_error_rates(s, C.byref(ri), C.byref(a), C.byref(b)) != 0:
                    raise RuntimeError("indel missing from the reference session")
                d["r2i"], d["i2r"] = a.value, b.value
                if "arid" in d:
                    L.ref_session_set_indel_haplotype(s, C.byref(ri), d["arid"], d["hap"], d["bypass"], d["forced"], d["ndfr"])
            res = []
            for rid, rd in enumerate(sc["reads"]):
                read = "".join(_CODE2CHAR.get(int(x), "N") for x in rd["code"]).encode()
                qual = np.ascontiguousarray(rd["qual"], np.uint8)
                path = (PathSeg * len(rd["path"]))(*[PathSeg(t, l) for t, l in rd["path"]])
                isr, pos = C.c_int(), C.c_int()
                cig = C.create_string_buffer(512)
                rc = L.ref_session_realign(s, read, _p(qual), len(qual), rd["pos"], len(rd["path"]), path, int(rd["is_fwd"]),
                                           rd["map_level"], rd["realign_range"][0], rd["realign_range"][1], rid + 1,
                                           int(sc.get("is_haplotyping_enabled", 0)), int(sc.get("min_read_bp_flank", 5)),
                                           C.byref(isr), C.byref(pos), cig, 512)
                if rc != 0:
                    res.append(dict(threw=True))
                    continue
                buf = (RefReadScore * 64)()
                n = L.ref_session_read_scores(s, rid + 1, buf, 64)
                if n < 0:
                    raise RuntimeError("score buffer overflow")
                scores, sub = [], []
                for i in range(n):
                    b = buf[i]
                    key = (b.pos, b.type, b.del_len, b.ins.decode())
                    if b.is_suboverlap:
                        sub.append(key)
                        continue
                    scores.append(dict(key=key, ref_lnp=float(b.ref_lnp), indel_lnp=float(b.indel_lnp),
                                       non_ambig=b.non_ambig, read_length=b.read_length,
                                       is_tier1_read=b.is_tier1_read, is_fwd_strand=b.is_fwd_strand, read_pos=b.read_pos,
                                       edge_dist=b.edge_dist,
                                       alt=[((b.alt_pos[a], b.alt_type[a], b.alt_del_len[a], b.alt_ins[a].value.decode()),
                                             float(b.alt_lnp[a])) for a in range(b.n_alt)]))
                res.append(dict(threw=False, is_realigned=bool(isr.value), pos=pos.value, cigar=cig.value.decode(),
                                scores=scores, suboverlap=sub))
            out.append(res)
        finally:
            L.ref_session_destroy(s)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# row a8: pileup

class PileupOptions(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("min_basecall_qscore", "mismatch_density_flank_size", "mismatch_density_max_count",
                                         "use_tier2_evidence", "tier2_mismatch_density_max_count", "is_mapq_adjust",
                                         "min_distance_from_read_edge", "largest_total_indel_ref_span_per_read",
                                         "report_begin", "report_end")]


def pileup_options(**kw):
    o = PileupOptions(17, 20, 2, 0, 10, 1, 0, 49, 0, 0)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class ReadBatchStruct(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("read_off", vp), ("read_code", vp), ("read_qual", vp), ("path_off", vp),
                ("path", vp), ("pos", vp), ("is_fwd", vp), ("mapq", vp), ("map_level", vp), ("ref_seq", C.c_char_p),
                ("ref_offset", C.c_int32), ("ref_len", C.c_int32), ("cand_snv_mask", vp)]


def read_batch_struct(rb, keep):
    ref = rb.ref_seq.encode()
    keep.append(ref)
    return ReadBatchStruct(rb.n_reads, _p(rb.read_off), _p(rb.read_code), _p(rb.read_qual), _p(rb.path_off), _p(rb.path),
                           _p(rb.pos), _p(rb.is_fwd), _p(rb.mapq), _p(rb.map_level), ref, rb.ref_offset, len(ref),
                           None if rb.cand_snv_mask is None else _p(rb.cand_snv_mask))


def pileup_reads(rb, opt, mode):
    """oracle restatement of pileup_read_segment over a strelka_amd.synth.ReadBatch -> (call_off, calls, spandel, submapped)"""
    L = oracle()
    L.sko_pileup_reads.restype = C.c_int64
    L.sko_pileup_reads.argtypes = [C.POINTER(ReadBatchStruct), C.POINTER(PileupOptions), C.c_int, vp, vp, C.c_int64, vp, vp]
    n_loci = opt.report_end - opt.report_begin
    keep = []
    s = read_batch_struct(rb, keep)
    cap = 2 * rb.n_bases + 1
    call_off = np.zeros(n_loci + 1, np.int64)
    calls = np.zeros(cap, np.uint16)
    sd = np.zeros(max(n_loci, 1), np.uint32)
    sm = np.zeros(max(n_loci, 1), np.uint32)
    n = L.sko_pileup_reads(C.byref(s), C.byref(opt), mode, _p(call_off), _p(calls), cap, _p(sd), _p(sm))
    if n < 0:
        raise RuntimeError("sko_pileup_reads failed")
    return call_off, calls[:n].copy(), sd[:n_loci], sm[:n_loci]


def pileup_reads_mapq(rb, opt):
    """raw tier1 columns + MapqTracker per position -> (call_off, calls, spandel, submapped, mapq_count, mapq_zero, mapq_sumsq)"""
    L = oracle()
    L.sko_pileup_reads_mapq.restype = C.c_int64
    L.sko_pileup_reads_mapq.argtypes = [C.POINTER(ReadBatchStruct), C.POINTER(PileupOptions), C.c_int, vp, vp, C.c_int64, vp, vp, vp, vp, vp]
    n_loci = opt.report_end - opt.report_begin
    keep = []
    s = read_batch_struct(rb, keep)
    cap = 2 * rb.n_bases + 1
    call_off = np.zeros(n_loci + 1, np.int64)
    calls = np.zeros(cap, np.uint16)
    sd, sm, mn, mz = (np.zeros(max(n_loci, 1), np.uint32) for _ in range(4))
    sq = np.zeros(max(n_loci, 1), np.uint64)
    n = L.sko_pileup_reads_mapq(C.byref(s), C.byref(opt), 0, _p(call_off), _p(calls), cap, _p(sd), _p(sm), _p(mn), _p(mz), _p(sq))
    if n < 0:
        raise RuntimeError("sko_pileup_reads_mapq failed")
    return call_off, calls[:n].copy(), sd[:n_loci], sm[:n_loci], mn[:n_loci], mz[:n_loci], sq[:n_loci]


def pileup_reads_readpos(rb, opt):
    """raw tier1 columns + (read_pos | read_size << 16) of every call -> (call_off, calls, read_pos)"""
    L = oracle()
    L.sko_pileup_reads_readpos.restype = C.c_int64
    L.sko_pileup_reads_readpos.argtypes = [C.POINTER(ReadBatchStruct), C.POINTER(PileupOptions), vp, vp, C.c_int64, vp]
    n_loci = opt.report_end - opt.report_begin
    keep = []
    s = read_batch_struct(rb, keep)
    cap = 2 * rb.n_bases + 1
    call_off = np.zeros(n_loci + 1, np.int64)
    calls = np.zeros(cap, np.uint16)
    rp = np.zeros(cap, np.uint32)
    n = L.sko_pileup_reads_readpos(C.byref(s), C.byref(opt), _p(call_off), _p(calls), cap, _p(rp))
    if n < 0:
        raise RuntimeError("sko_pileup_reads_readpos failed")
    return call_off, calls[:n].copy(), rp[:n].copy()


def pileup_reads_evs(rb, opt):
    """the germline EVS words of every live match position (updateGermlineScoringMetrics' arguments) -> (evs_off, evs_words)"""
    L = oracle()
    L.sko_pileup_reads_evs.restype = C.c_int64
    L.sko_pileup_reads_evs.argtypes = [C.POINTER(ReadBatchStruct), C.POINTER(PileupOptions), vp, vp, C.c_int64]
    n_loci = opt.report_end - opt.report_begin
    keep = []
    s = read_batch_struct(rb, keep)
    cap = 2 * rb.n_bases + 1
    off = np.zeros(n_loci + 1, np.int64)
    words = np.zeros(cap, np.uint64)
    n = L.sko_pileup_reads_evs(C.byref(s), C.byref(opt), _p(off), _p(words), cap)
    if n < 0:
        raise RuntimeError("sko_pileup_reads_evs failed")
    return off, words[:n].copy()


def ref_germline_metrics(words, ref_base_id):
    """The REFERENCE's rank-sum / mean accumulators (snp_pos_info) fed with the observations one position's EVS words spell out
    (the unpacking of adapter/sk_adapter_pileup.cpp germline_fill_scoring_metrics) -> the six numbers of
    ref_pileup_pipeline(germline_metrics=True)'s `evs`."""
    L = ref()
    L.ref_germline_metrics_from_observations.argtypes = [C.c_int] + [vp] * 7
    w = np.ascontiguousarray(words, np.uint64)
    is_ref = ((w & 7) == ref_base_id).astype(np.uint8)
    mapq = ((w >> 3) & 0xff).astype(np.uint8)
    q = ((w >> 11) & 0x7f).astype(np.uint8)
    cycle = ((w >> 18) & 0x7ff).astype(np.uint16)
    edge = ((w >> 29) & 0x1f).astype(np.uint8)
    sub = ((w >> 34) & 1).astype(np.uint8)
    out = np.zeros(6, np.float64)
    L.ref_germline_metrics_from_observations(len(w), _p(is_ref), _p(mapq), _p(q), _p(cycle), _p(edge), _p(sub), _p(out))
    return out


def mapped_qscore_table():
    L = oracle()
    return np.array([[L.sko_mapped_qscore(q, m) for q in range(71)] for m in range(91)], np.int32)


def ref_pileup_pipeline(reads, ref_seq, ref_offset, opt, candidate_indels=(), return_indels=False, germline_metrics=False):
    """The REFERENCE's position processor end to end (oracle/ref/ref_driver_pileup.cpp): reads (dicts as produced by
    synth.pileup_reads, position-sorted) -> read buffer -> realignment -> pileup.
    Returns (finals, columns): finals = per piled read, in pileup order, dict(read_id = index into `reads`, is_realigned,
    pos, is_fwd, cigar, skipped, input_pos, input_cigar, realign_range); columns = {pos: dict(calls, tier2_calls, spandel,
    submapped)}.  With return_indels=True a third value lists the IndelBuffer as the realigner saw it:
    dict(pos, type, del_len, ins_seq, is_candidate, r2i, i2r (log error rates), read_ids = indices into `reads`).
    With germline_metrics=True the session accumulates the germline EVS metrics (updateGermlineScoringMetrics) and every column
    also holds evs = (MQRankSum, BaseQRankSum, ReadPosRankSum, rawPos, avgBaseQ, meanDistanceFromReadEdge) and mapq_count."""
    L = ref()
    L.refpp_set_germline_metrics.argtypes = [C.c_int]
    L.refpp_column_evs.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_uint32)]
    L.refpp_create.restype = vp
    L.refpp_create.argtypes = [C.c_char_p] + [C.c_int] * 10
    L.refpp_destroy.argtypes = [vp]
    L.refpp_add_read.argtypes = [vp, C.c_char_p, vp, C.c_int, C.c_int, C.POINTER(PathSeg), C.c_int, C.c_int, C.c_int]
    L.refpp_add_candidate_indel.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.refpp_finish.argtypes = [vp]
    L.refpp_n_columns.argtypes = [vp]
    L.refpp_column_info.argtypes = [vp, C.c_int] + [C.POINTER(C.c_int32)] * 3 + [C.POINTER(C.c_uint32)] * 2
    L.refpp_column_calls.argtypes = [vp, C.c_int, vp, vp]
    L.refpp_n_finals.argtypes = [vp]
    L.refpp_final.argtypes = [vp, C.c_int, C.POINTER(C.c_uint32)] + [C.POINTER(C.c_int32)] * 4 + [C.c_char_p, C.c_int]
    L.refpp_final_input.argtypes = [vp, C.c_int] + [C.POINTER(C.c_int32)] * 3 + [C.c_char_p, C.c_int]
    L.refpp_n_indels.argtypes = [vp]
    L.refpp_indel_n_scores.argtypes = [vp, C.c_int]
    L.refpp_indel_score.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_char_p, C.c_int]
    L.refpp_indel.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_char_p, C.c_int,
                              C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                              C.POINTER(C.c_double), vp, C.c_int]
    L.refpp_set_germline_metrics(int(germline_metrics))
    try:
        s = L.refpp_create(ref_seq.encode(), ref_offset, opt.report_begin, opt.report_end, opt.min_basecall_qscore,
                           opt.mismatch_density_flank_size, opt.mismatch_density_max_count, opt.use_tier2_evidence,
                           opt.tier2_mismatch_density_max_count, opt.is_mapq_adjust, opt.min_distance_from_read_edge)
    finally:
        L.refpp_set_germline_metrics(0)
    if not s:
        raise RuntimeError("refpp_create failed")
    try:
        events = [(c["pos"], 0, c) for c in candidate_indels] + [(r["pos"], 1, i) for i, r in enumerate(reads)]
        events.sort(key=lambda e: (e[0], e[1]))
        id_of = {}
        for _, kind, x in events:
            if kind == 0:
                if L.refpp_add_candidate_indel(s, x["pos"], x.get("del_len", 0), x.get("ins_seq", "").encode()):
                    raise RuntimeError("reference rejected candidate indel")
                continue
            r = reads[x]
            seq = "".join(_CODE2CHAR.get(int(c), "N") for c in r["code"]).encode()
            qual = np.ascontiguousarray(r["qual"], np.uint8)
            path = (PathSeg * len(r["path"]))(*[PathSeg(t, l) for t, l in r["path"]])
            rid = L.refpp_add_read(s, seq, _p(qual), r["pos"], len(r["path"]), path, int(r["is_fwd"]), r["mapq"], r["map_level"])
            if rid == -2:
                raise RuntimeError("reference threw while inserting read %d" % x)
            if rid >= 0:
                id_of[rid] = x
        if L.refpp_finish(s):
            raise RuntimeError("reference threw in reset()")
        finals = []
        for i in range(L.refpp_n_finals(s)):
            rid, a, b, c, d = C.c_uint32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
            buf = C.create_string_buffer(1024)
            L.refpp_final(s, i, rid, a, b, c, d, buf, 1024)
            ip, rb_, re_ = C.c_int32(), C.c_int32(), C.c_int32()
            buf2 = C.create_string_buffer(1024)
            L.refpp_final_input(s, i, ip, rb_, re_, buf2, 1024)
            finals.append(dict(read_id=id_of[rid.value], is_realigned=bool(a.value), pos=b.value, is_fwd=bool(c.value),
                               skipped=bool(d.value), cigar=buf.value.decode(), input_pos=ip.value,
                               input_cigar=buf2.value.decode(), realign_range=(rb_.value, re_.value)))
        cols = {}
        for i in range(L.refpp_n_columns(s)):
            pos, n, n2 = C.c_int32(), C.c_int32(), C.c_int32()
            sd, sm = C.c_uint32(), C.c_uint32()
            L.refpp_column_info(s, i, pos, n, n2, sd, sm)
            calls = np.zeros(max(n.value, 1), np.uint16)
            t2 = np.zeros(max(n2.value, 1), np.uint16)
            L.refpp_column_calls(s, i, _p(calls), _p(t2))
            cols[pos.value] = dict(calls=calls[:n.value].copy(), tier2_calls=t2[:n2.value].copy(), spandel=sd.value,
                                   submapped=sm.value)
            if germline_metrics:
                evs, mc = np.zeros(6, np.float64), C.c_uint32()
                L.refpp_column_evs(s, i, _p(evs), mc)
                cols[pos.value].update(evs=evs, mapq_count=mc.value)
        if not return_indels:
            return finals, cols
        indels = []
        for i in range(L.refpp_n_indels(s)):
            pos, ty, ic, nd, fo = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
            dl = C.c_uint32()
            a1, a2 = C.c_double(), C.c_double()
            ins = C.create_string_buffer(256)
            ids = np.zeros(4096, np.uint32)
            n = L.refpp_indel(s, i, pos, ty, dl, ins, 256, ic, nd, fo, a1, a2, _p(ids), 4096)
            scores = []
            for k in range(L.refpp_indel_n_scores(s, i)):
                iv = np.zeros(12, np.int32)
                fv = np.zeros(4, np.float32)
                ab = C.create_string_buffer(256)
                L.refpp_indel_score(s, i, k, _p(iv), _p(fv), ab, 256)
                if int(iv[0]) not in id_of:
                    continue
                ai = ab.value.decode().split("|")
                scores.append(dict(read_id=id_of[int(iv[0])], ref_lnp=float(fv[0]), indel_lnp=float(fv[1]), non_ambig=int(iv[1]),
                                   read_length=int(iv[2]), is_tier1_read=int(iv[3]), is_fwd_strand=int(iv[4]), read_pos=int(iv[5]),
                                   edge_dist=int(iv[6]),
                                   alt=[((int(iv[8 + a]), int(iv[10 + a]), ai[a]), float(fv[2 + a])) for a in range(int(iv[7]))]))
            indels.append(dict(scores=scores, pos=pos.value, type=ty.value, del_len=dl.value, ins_seq=ins.value.decode(),
                               is_candidate=ic.value, ndfr=nd.value, forced=fo.value, r2i=a1.value, i2r=a2.value,
                               read_ids=sorted(set(id_of[int(x)] for x in ids[:min(n, 4096)] if int(x) in id_of))))
        return finals, cols, indels
    finally:
        L.refpp_destroy(s)


# ---------------------------------------------------------------------------------------------------------------------
# GlobalAligner<int>

class AlignScores(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("match", "mismatch", "open", "extend", "offEdge", "insertDelete",
                                         "isAllowEdgeInsertion", "isRequireEdgeDeletion")]


def align_scores(match=1, mismatch=-4, open=-5, extend=-1, off_edge=-100, insert_delete=-5, allow_edge_insertion=1,
                 require_edge_deletion=1):
    """defaults: the active-region detector's aligner (L/starling_common/ActiveRegionDetector.hh:59-63, .cpp:41)"""
    return AlignScores(match, mismatch, open, extend, off_edge, insert_delete, allow_edge_insertion, require_edge_deletion)


_CIGAR = "?MIDNSHP=X"


def global_align(query, ref_seq, sc):
    """oracle restatement -> (score, begin_pos, cigar)"""
    L = oracle()
    L.sko_global_align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(AlignScores), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), C.POINTER(PathSeg), C.c_int]
    q, r = query.encode(), ref_seq.encode()
    cap = len(q) + len(r) + 4
    path = (PathSeg * cap)()
    score, beg = C.c_int32(), C.c_int32()
    n = L.sko_global_align(q, len(q), r, len(r), C.byref(sc), C.byref(score), C.byref(beg), path, cap)
    if n < 0:
        raise RuntimeError("sko_global_align failed")
    return score.value, beg.value, "".join("%d%s" % (path[i].length, _CIGAR[path[i].type]) for i in range(n))


def ref_global_align(query, ref_seq, sc):
    """the REFERENCE's GlobalAligner<int>::align -> (score, begin_pos, cigar)"""
    L = ref()
    L.ref_global_align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int] + [C.c_int] * 8 + [C.POINTER(C.c_int)] * 2 + [C.c_char_p, C.c_int]
    q, r = query.encode(), ref_seq.encode()
    score, beg = C.c_int(), C.c_int()
    buf = C.create_string_buffer(4 * (len(q) + len(r)) + 64)
    rc = L.ref_global_align(q, len(q), r, len(r), sc.match, sc.mismatch, sc.open, sc.extend, sc.offEdge, sc.insertDelete,
                            sc.isAllowEdgeInsertion, sc.isRequireEdgeDeletion, C.byref(score), C.byref(beg), buf, len(buf))
    if rc:
        raise RuntimeError("reference GlobalAligner threw")
    return score.value, beg.value, buf.value.decode()


def ref_discover_indels_and_mismatches(ref_seq, ref_offset, ar_begin, ar_end, prev_ar_end, max_indel_size, haplotype):
    """the REFERENCE's ActiveRegionProcessor::discoverIndelsAndMismatches -> ([(pos, type, del_len, ins_seq)], numIndels)"""
    L = ref()
    L.ref_discover_indels_and_mismatches.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_char_p,
                                                     C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    buf = C.create_string_buffer(64 * (len(haplotype) + len(ref_seq)) + 64)
    n = C.c_int()
    rc = L.ref_discover_indels_and_mismatches(ref_seq.encode(), ref_offset, ar_begin, ar_end, prev_ar_end, max_indel_size,
                                              haplotype.encode(), buf, len(buf), C.byref(n))
    if rc:
        raise RuntimeError("reference discoverIndelsAndMismatches failed (%d)" % rc)
    out = []
    for item in buf.value.decode().split(";"):
        if item:
            pos, typ, dl, ins = item.split(",")
            out.append((int(pos), int(typ), int(dl), ins))
    return out, n.value


def ref_normalize_alignment(ref_seq, ref_offset, read_chars, pos, path):
    """the reference's own normalizeAlignment (oracle/ref/ref_driver_feed.cpp) -> (changed, pos, path)"""
    L = ref()
    L.ref_normalize_alignment.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int32), vp, C.POINTER(C.c_int32), C.c_int32]
    cap = len(path) + 4
    buf = np.zeros(2 * cap, np.uint32)
    for i, (t, l) in enumerate(path):
        buf[2 * i], buf[2 * i + 1] = t, l
    p, n = C.c_int32(int(pos)), C.c_int32(len(path))
    rb, qb = ref_seq.encode(), read_chars.encode()
    rc = L.ref_normalize_alignment(rb, int(ref_offset), len(rb), qb, len(qb), C.byref(p), buf.ctypes.data, C.byref(n), cap)
    if rc < 0:
        raise RuntimeError("ref_normalize_alignment: result does not fit")
    return rc, p.value, [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n.value)]


REF_GVCF_SITE_DTYPE = np.dtype([("pos", "<i4"), ("is_compressible", "u1"), ("is_ref_unknown", "u1"), ("gt_ploidy", "u1"), ("gt_phased", "u1"),
                                ("gt_allele0", "u1"), ("gt_allele1", "u1"), ("ploidy", "u1"), ("flush_before", "u1"), ("locus_filters", "<u4"),
                                ("sample_filters", "<u4"), ("gqx", "<i4"), ("used_basecalls", "<u4"), ("unused_basecalls", "<u4")])
REF_GVCF_BLOCK_DTYPE = np.dtype([("first_site", "<i4"), ("pos", "<i4"), ("count", "<i4"), ("is_gqx_defined", "<i4"), ("gqx_min", "<f8"),
                                 ("dpu_mean", "<f8"), ("dpf_mean", "<f8"), ("dpu_min", "<f8")])


def ref_gvcf_block_sites(sites, block_percent_tol=30, block_abs_tol=3):
    """the reference's own gvcf_block_site_record, driven per site as gvcf_writer::queue_site_record drives it
    (oracle/ref/ref_driver_gvcf_block.cpp).  sites: synth.gvcf_sites records -> (kind[n], blocks[REF_GVCF_BLOCK_DTYPE])"""
    L = ref()
    n = len(sites)
    rs = np.zeros(n, REF_GVCF_SITE_DTYPE)
    for k in ("pos", "is_compressible", "is_ref_unknown", "ploidy", "flush_before", "locus_filters", "sample_filters", "gqx", "used_basecalls", "unused_basecalls"):
        rs[k] = sites[k]
    rs["gt_ploidy"] = sites["gt"] >> 24
    rs["gt_phased"] = (sites["gt"] >> 16) & 1
    rs["gt_allele0"] = (sites["gt"] >> 8) & 0xff
    rs["gt_allele1"] = sites["gt"] & 0xff
    kind = np.zeros(max(n, 1), np.uint8)
    blocks = np.zeros(max(n, 1), REF_GVCF_BLOCK_DTYPE)
    L.ref_gvcf_block_sites.argtypes = [vp, C.c_int32, C.c_uint32, C.c_uint32, vp, vp]
    nb = L.ref_gvcf_block_sites(rs.ctypes.data, n, block_percent_tol, block_abs_tol, kind.ctypes.data, blocks.ctypes.data)
    return kind[:n], blocks[:nb]


def ref_bai_query(bam_path, tid, begin, end):
    """htslib's own sam_itr_queryi on the index of `bam_path` (oracle/ref/ref_driver_bai.cpp) -> [(begin, end)] virtual offsets"""
    L = ref()
    L.ref_bai_query.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32]
    buf = np.zeros(2 * 4096, np.uint64)
    n = L.ref_bai_query(bam_path.encode(), tid, begin, end, buf.ctypes.data, 4096)
    if n < 0:
        raise RuntimeError("ref_bai_query: no index / no iterator")
    return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)]


def gvcf_site_summaries(batch, genotypes):
    """What the gVCF writer's block logic reads of a position (sk_gvcf_site_summary), in numpy / plain Python, from the reference's
    statements: getSiteAltAlleles' rank pass (L/applications/starling/starling_pos_processor.cpp:527-560: per sample the `ploidy` most
    frequent bases, a base ranked only with at least max(1, unsigned(0.10 * depth)) calls; ties go to the lower base index), its pass over
    the two most likely genotypes (:578-611), LocusSampleInfo::setGqx (gvcf_locus_info.hh:356-369) and the AD counts
    (updateSnvLocusWithSampleInfo :452-466).  A position is PLAIN when the reference base is known, the caller ploidy is 2, the cleaned
    column is not empty and no alternate allele would be listed.  -> array of (flags, gqx, ref_fwd, ref_rev)"""
    out = np.zeros(batch.n_loci, np.dtype([("flags", np.uint32), ("gqx", np.int32), ("ref_fwd", np.uint32), ("ref_rev", np.uint32)]))
    off = np.asarray(batch.call_off)
    for i in range(batch.n_loci):
        ref = int(batch.ref_base[i])
        g = genotypes[i]
        if ref > 3 or not int(g["is_called"]):
            continue
        calls = np.asarray(batch.calls[off[i]:off[i + 1]]).astype(np.int64)
        base = (calls >> 6) & 0xf
        fwd = (calls >> 10) & 1
        cnt = np.zeros((4, 2), np.int64)
        for b, f in zip(base, fwd):
            if b <= 3:
                cnt[b, f] += 1
        c = cnt.sum(axis=1).astype(np.float64)
        min_count = max(1, int(float(int(c.sum())) * 0.10))
        ploidy = int(batch.ploidy[i]) if batch.ploidy is not None else 2
        alt = False
        for _ in range(min(ploidy, 2)):
            mb = 0
            for b in range(1, 4):
                if c[b] > c[mb]:
                    mb = b
            if c[mb] >= min_count and mb != ref:
                alt = True
            c[mb] = 0
        hom_ref = int(g["poly"]["max_gt"]) == ref and int(g["genome"]["max_gt"]) == ref
        out["gqx"][i] = min(int(g["genome"]["max_gt_qphred"]), int(g["poly"]["max_gt_qphred"]))
        out["ref_fwd"][i] = cnt[ref, 1]
        out["ref_rev"][i] = cnt[ref, 0]
        if ploidy == 2 and len(calls) > 0 and not alt and hom_ref:
            out["flags"][i] = 1
    return out


def gvcf_plain_runs(summary, clean_count, raw_count, mapq_count, opt):
    """sk_gvcf_run of every site of a window in plain Python: from each plain site the greedy joining of gvcf_writer::queue_site_record
    (L/applications/starling/gvcf_writer.cpp:278-302) over the plain sites that follow -- testCanSiteJoinSampleBlockShared /
    testCanSiteJoinSampleBlock (gvcf_block_site_record.cpp:77-122, :163-182: equal filters; used / unused depth and GQX each within
    tolerance of the block, check_block_tolerance :41-55 on the stream_stat with the new value added) and joinSiteToSampleBlock
    (:126-157); the filters by ScoringModelManager::applyDepthFilter (ScoringModelManager.cpp:234-249) and default_classify_site
    (:270-311) for a homozygous-reference site.  `opt`: an object with sk_gvcf_block_options' fields.  -> structured array as
    capi.GVCF_RUN_DTYPE"""
    import math
    n = len(summary)
    dt = np.dtype([("len", np.int32), ("filter_key", np.uint32), ("gqx_min", np.int32), ("gqx_max", np.int32), ("dpu_min", np.uint32),
                   ("dpu_max", np.uint32), ("dpf_min", np.uint32), ("dpf_max", np.uint32)])
    out = np.zeros(n, dt)
    frac_tol, abs_tol = float(opt.block_percent_tol) / 100., int(opt.block_abs_tol)

    class Stat:  # stream_stat.hh:43-66 without Q
        def __init__(self):
            self.M = self.max = self.min = 0.0
            self.k = 0

        def add(self, x):
            self.k += 1
            if self.k == 1 or x > self.max:
                self.max = x
            if self.k == 1 or x < self.min:
                self.min = x
            self.M += (x - self.M) / float(self.k)

        def with_value(self, x):
            s = Stat()
            s.M, s.max, s.min, s.k = self.M, self.max, self.min, self.k
            s.add(float(x))
            return s

    def compat_round(x):
        return math.floor(x + 0.5) if x >= 0 else math.ceil(x - 0.5)

    def tolerable(ss):
        mn = int(compat_round(ss.min))
        if (mn + abs_tol) >= ss.max / 2.0:
            return True
        ftol = int(math.floor(mn * frac_tol))
        if ftol <= abs_tol:
            return False
        return (mn + ftol) >= ss.max / 2.0
    used = np.asarray(clean_count, np.int64)
    unused = np.asarray(raw_count, np.int64) - used
    key = np.zeros(n, np.int64)
    plain = (np.asarray(summary["flags"]) & 1) != 0
    for i in range(n):
        if not plain[i]:
            continue
        k = 0
        ref_count = int(summary["ref_fwd"][i]) + int(summary["ref_rev"][i])
        if ref_count < opt.min_passed_call_depth or used[i] < opt.min_passed_call_depth:
            k |= 1
        if opt.is_min_homref_gqx and float(summary["gqx"][i]) < opt.min_homref_gqx:
            k |= 2
        if opt.is_max_depth and float(mapq_count[i]) > opt.max_chrom_depth:
            k |= 4
        if opt.is_max_base_filt:
            total = float(used[i] + unused[i])
            if (0.0 if total == 0.0 else float(unused[i]) / total) > opt.max_base_filt:
                k |= 8
        key[i] = k
    for i in range(n):
        if not plain[i]:
            continue
        g, du, df = Stat(), Stat(), Stat()
        j = i
        while j < n:
            if j > i:
                if not plain[j] or key[j] != key[i]:
                    break
                if not tolerable(du.with_value(used[j])) or not tolerable(df.with_value(unused[j])) or not tolerable(g.with_value(int(summary["gqx"][j]))):
                    break
            du.add(float(used[j]))
            df.add(float(unused[j]))
            g.add(float(int(summary["gqx"][j])))
            j += 1
        out[i] = (j - i, key[i], int(g.min), int(g.max), int(du.min), int(du.max), int(df.min), int(df.max))
    return out
