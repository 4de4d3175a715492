"""ctypes binding of include/strelka_amd.h (plumbing for tests and bench.py; the product is the C-ABI library).

There is deliberately no fallback: if libstrelka_amd.so is missing this module raises, and if no gfx950 device is present
`init()` raises with the library's own error text.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libstrelka_amd.so")

c_void_p = C.c_void_p


class ScoreOp(C.Structure):
    _fields_ = [("length", C.c_uint16), ("kind", C.c_uint8), ("flags", C.c_uint8), ("src", C.c_int32)]


SCORE_OP_DTYPE = np.dtype([("length", "<u2"), ("kind", "u1"), ("flags", "u1"), ("src", "<i4")])
OP_BASES, OP_SOFT_CLIP, OP_NOBASE = 0, 1, 2
OPFLAG_PENALTY = 1


class AlignBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_cals", C.c_int32), ("n_ops", C.c_int64),
                ("read_off", c_void_p), ("read_code", c_void_p), ("read_qual", c_void_p),
                ("hap_off", c_void_p), ("hap_code", c_void_p), ("cal_off", c_void_p), ("op_off", c_void_p),
                ("ops", c_void_p), ("max_read_len", C.c_int32), ("max_hap_len", C.c_int32),
                ("entries", c_void_p), ("evmask", c_void_p), ("evmask_words", C.c_int32),
                ("colmat", c_void_p), ("colmat_off", c_void_p), ("addmask", c_void_p)]


class PathSeg(C.Structure):
    _fields_ = [("type", C.c_uint32), ("length", C.c_uint32)]


class IndelKey(C.Structure):
    _fields_ = [("pos", C.c_int32), ("type", C.c_int32), ("del_len", C.c_uint32), ("ins_len", C.c_uint32),
                ("ins_seq", C.c_char_p), ("is_candidate", C.c_int32)]


class CandidateAlignment(C.Structure):
    _fields_ = [("pos", C.c_int32), ("n_seg", C.c_int32), ("path", C.POINTER(PathSeg)), ("n_indels", C.c_int32),
                ("indels", C.POINTER(IndelKey)), ("leading", IndelKey), ("trailing", IndelKey)]


MAX_SAMPLES = 8
SEG = dict(NONE=0, MATCH=1, INSERT=2, DELETE=3, SKIP=4, SOFT_CLIP=5, HARD_CLIP=6, PAD=7, SEQ_MATCH=8, SEQ_MISMATCH=9)
CIGAR_CHARS = "?MIDNSHP=X"
INDEL = dict(NONE=0, INDEL=1, MISMATCH=2, BP_LEFT=3, BP_RIGHT=4)
MAPLEVEL = dict(UNKNOWN=0, TIER1=1, TIER2=2, SUB=3, UNMAPPED=4)


class RealignOptions(C.Structure):
    _fields_ = [("max_read_indel_toggle", C.c_int32), ("max_candidate_indel_density", C.c_double),
                ("max_realignment_candidates", C.c_uint32), ("max_indel_size", C.c_uint32),
                ("is_smoothed_alignments", C.c_int32), ("smoothed_lnp_range", C.c_double),
                ("upstream_oligo_size", C.c_uint32), ("is_haplotyping_enabled", C.c_int32),
                ("min_read_bp_flank", C.c_int32), ("sample_count", C.c_int32), ("host_threads", C.c_int32),
                ("enumeration", C.c_int32)]


class IndelInfo(C.Structure):
    _fields_ = [("key", IndelKey), ("ref_to_indel_log_prob", C.c_double), ("indel_to_ref_log_prob", C.c_double),
                ("active_region_id", C.c_int32), ("haplotype_id", C.c_int8 * MAX_SAMPLES),
                ("is_haplotyping_bypassed", C.c_uint8 * MAX_SAMPLES), ("is_forced_output", C.c_uint8),
                ("not_discovered_from_reads", C.c_uint8)]


class ReadInput(C.Structure):
    _fields_ = [("read_code", c_void_p), ("read_qual", c_void_p), ("read_len", C.c_int32), ("pos", C.c_int32),
                ("n_seg", C.c_int32), ("path", C.POINTER(PathSeg)), ("is_fwd_strand", C.c_int32),
                ("map_level", C.c_int32), ("sample_index", C.c_int32), ("realign_begin", C.c_int32),
                ("realign_end", C.c_int32), ("n_observed", C.c_int32), ("observed", C.POINTER(C.c_int32))]


class ReadPathScores(C.Structure):
    _fields_ = [("indel", C.c_int32), ("ref_lnp", C.c_float), ("indel_lnp", C.c_float), ("non_ambig", C.c_uint16),
                ("read_length", C.c_uint16), ("is_tier1_read", C.c_uint8), ("is_fwd_strand", C.c_uint8),
                ("read_pos", C.c_int16), ("distance_from_closest_read_edge", C.c_int16), ("n_alt", C.c_int32),
                ("alt_indel", C.c_int32 * 2), ("alt_lnp", C.c_float * 2)]


class ReadResult(C.Structure):
    _fields_ = [("n_candidate_alignments", C.c_int32), ("is_realigned", C.c_int32), ("realign_pos", C.c_int32),
                ("realign_n_seg", C.c_int32), ("realign_path", C.POINTER(PathSeg)), ("max_score", C.c_double),
                ("n_scores", C.c_int32), ("scores", C.POINTER(ReadPathScores)), ("n_suboverlap", C.c_int32),
                ("suboverlap", C.POINTER(C.c_int32)), ("warn_origin_skip", C.c_int32),
                ("warn_max_toggle_depth", C.c_int32)]


class AlignScores(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("match", "mismatch", "open", "extend", "off_edge", "insert_delete",
                                         "is_allow_edge_insertion", "is_require_edge_deletion")]


class GlobalAlignBatch(C.Structure):
    _fields_ = [("n", C.c_int32), ("query_off", c_void_p), ("query", c_void_p), ("ref_off", c_void_p), ("ref", c_void_p)]


class PileupOptions(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("min_basecall_qscore", "mismatch_density_flank_size", "mismatch_density_max_count",
                                         "use_tier2_evidence", "tier2_mismatch_density_max_count", "is_mapq_adjust",
                                         "min_distance_from_read_edge", "largest_total_indel_ref_span_per_read",
                                         "report_begin", "report_end")]


class ReadBatchStruct(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("read_off", c_void_p), ("read_code", c_void_p), ("read_qual", c_void_p),
                ("path_off", c_void_p), ("path", c_void_p), ("pos", c_void_p), ("is_fwd", c_void_p), ("mapq", c_void_p),
                ("map_level", c_void_p), ("ref_seq", c_void_p), ("ref_offset", C.c_int32), ("ref_len", C.c_int32),
                ("cand_snv_mask", c_void_p)]


class PileupColumns(C.Structure):
    _fields_ = [("n_loci", C.c_int32), ("capacity", C.c_int64), ("call_off", c_void_p), ("calls", c_void_p),
                ("spandel_count", c_void_p), ("submapped_count", c_void_p)]


PILEUP_RAW_TIER1, PILEUP_RAW_TIER2, PILEUP_CLEAN_TIER1, PILEUP_CLEAN_TIER2 = 0, 1, 2, 3


GVCF_SITE_SUMMARY_DTYPE = np.dtype([("flags", np.uint32), ("gqx", np.int32), ("ref_fwd", np.uint32), ("ref_rev", np.uint32)])


GVCF_RUN_DTYPE = np.dtype([("len", np.int32), ("filter_key", np.uint32), ("gqx_min", np.int32), ("gqx_max", np.int32), ("dpu_min", np.uint32),
                           ("dpu_max", np.uint32), ("dpf_min", np.uint32), ("dpf_max", np.uint32)])
assert GVCF_RUN_DTYPE.itemsize == 32


class GvcfBlockOptions(C.Structure):
    _fields_ = [("min_passed_call_depth", C.c_uint32), ("is_min_homref_gqx", C.c_int32), ("min_homref_gqx", C.c_double), ("is_max_depth", C.c_int32),
                ("is_max_base_filt", C.c_int32), ("max_chrom_depth", C.c_double), ("max_base_filt", C.c_double), ("block_percent_tol", C.c_uint32),
                ("block_abs_tol", C.c_uint32)]


def gvcf_block_options(**kw):
    """the reference's defaults (gvcf_options.hh:57-77); max depth filter off unless max_chrom_depth is given"""
    o = GvcfBlockOptions(3, 1, 30.0, 0, 1, 0.0, 0.4, 30, 3)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class PileupWindow(C.Structure):
    _fields_ = [("begin", C.c_int32), ("end", C.c_int32), ("tier1_off", c_void_p), ("tier1_calls", c_void_p),
                ("tier2_off", c_void_p), ("tier2_calls", c_void_p), ("spandel_count", c_void_p), ("submapped_count", c_void_p),
                ("mapq_count", c_void_p), ("mapq_zero_count", c_void_p), ("mapq_sum_square", c_void_p),
                ("clean_count", c_void_p), ("genotype", c_void_p), ("evs_off", c_void_p), ("evs_words", c_void_p),
                ("site_summary", c_void_p), ("gvcf_runs", c_void_p)]


class SomaticPileupWindow(C.Structure):
    _fields_ = [("normal", PileupWindow), ("tumor", PileupWindow), ("normal_clean_tier2_count", c_void_p),
                ("tumor_clean_tier2_count", c_void_p), ("tumor_tier1_read_pos", c_void_p), ("genotype", c_void_p)]


class PileupBatch(C.Structure):
    _fields_ = [("n_loci", C.c_int32), ("call_off", c_void_p), ("calls", c_void_p), ("de", c_void_p),
                ("ref_base", c_void_p), ("ploidy", c_void_p)]


class GermlineOptions(C.Structure):
    _fields_ = [("bsnp_diploid_theta", C.c_double), ("bsnp_ssd_no_mismatch", C.c_double),
                ("bsnp_ssd_one_mismatch", C.c_double), ("is_min_vexp", C.c_int32), ("min_vexp", C.c_double)]


class SomaticSnvOptions(C.Structure):
    _fields_ = [("bsnp_diploid_theta", C.c_double), ("somatic_snv_rate", C.c_double),
                ("shared_site_error_rate", C.c_double), ("shared_site_error_strand_bias_fraction", C.c_double),
                ("ssnv_contam_tolerance", C.c_double)]


class IndelOptions(C.Structure):
    _fields_ = [("min_read_bp_flank", C.c_int32), ("random_base_match_prob", C.c_double),
                ("tier2_random_base_match_prob", C.c_double), ("read_confident_support_threshold", C.c_double),
                ("is_use_alt_indel", C.c_int32), ("fast_form", C.c_int32)]


class SomaticIndelOptions(C.Structure):
    _fields_ = [("bindel_diploid_theta", C.c_double), ("somatic_indel_rate", C.c_double),
                ("shared_indel_error_factor", C.c_double), ("indel_contam_tolerance", C.c_double)]


class ReadScoreBatch(C.Structure):
    _fields_ = [("n_indels", C.c_int32), ("read_off", c_void_p), ("ref_lnp", c_void_p), ("indel_lnp", c_void_p),
                ("alt_lnp", c_void_p), ("non_ambig", c_void_p), ("read_length", c_void_p), ("read_flags", c_void_p),
                ("del_len", c_void_p), ("ins_len", c_void_p), ("is_breakpoint", c_void_p)]


class AlleleGroupBatch(C.Structure):
    _fields_ = [("n_groups", C.c_int32), ("read_off", c_void_p), ("n_alt", c_void_p), ("ploidy", c_void_p),
                ("del_len", c_void_p), ("ins_len", c_void_p), ("ref_lnp", c_void_p), ("allele_lnp", c_void_p),
                ("non_ambig", c_void_p), ("read_length", c_void_p), ("read_flags", c_void_p)]


MAX_ALT, MAX_INDEL_GT = 3, 10
SOMATIC_INDEL_CALL_DTYPE = np.dtype([("normal_lhood", "<f8", (21,)), ("tumor_lhood", "<f8", (21,)), ("max_gt", "<u4"),
                                     ("qphred", "<i4"), ("from_ntype_qphred", "<i4"), ("ntype", "<u4")])
ALLELE_GROUP_CALL_DTYPE = np.dtype([("lhood", "<f8", (MAX_INDEL_GT,)), ("counts", "<u4", (2, MAX_ALT + 2)),
                                    ("n_genotypes", "<u4"), ("n_reads_used", "<u4")])
assert SOMATIC_INDEL_CALL_DTYPE.itemsize == 352 and ALLELE_GROUP_CALL_DTYPE.itemsize == 128
# multi-sample allele groups (up to ploidy x sample_count alternate alleles): sk_allele_group_call_wide
MAX_ALT_WIDE, MAX_INDEL_GT_WIDE = 8, 45
ALLELE_GROUP_CALL_WIDE_DTYPE = np.dtype([("lhood", "<f8", (MAX_INDEL_GT_WIDE,)), ("counts", "<u4", (2, MAX_ALT_WIDE + 2)),
                                         ("n_genotypes", "<u4"), ("n_reads_used", "<u4")])
assert ALLELE_GROUP_CALL_WIDE_DTYPE.itemsize == 448
# ... of runs of up to eight samples: sk_allele_group_call_xwide
MAX_ALT_XWIDE, MAX_INDEL_GT_XWIDE = 16, 153
ALLELE_GROUP_CALL_XWIDE_DTYPE = np.dtype([("lhood", "<f8", (MAX_INDEL_GT_XWIDE,)), ("counts", "<u4", (2, MAX_ALT_XWIDE + 2)),
                                          ("n_genotypes", "<u4"), ("n_reads_used", "<u4")])
assert ALLELE_GROUP_CALL_XWIDE_DTYPE.itemsize == 1376

DIGT_RS_DTYPE = np.dtype([("ref_pprob", "<f8"), ("max_gt", "<u4"), ("snp_qphred", "<i4"), ("max_gt_qphred", "<i4"),
                          ("_pad", "<i4")])
DIGT_CALL_DTYPE = np.dtype([("lhood", "<f4", (10,)), ("phredLoghood", "<u4", (10,)), ("genome", DIGT_RS_DTYPE),
                            ("poly", DIGT_RS_DTYPE), ("strand_bias", "<f8"), ("ref_gt", "<u4"), ("is_called", "<u4")])
SOMATIC_CALL_DTYPE = np.dtype([("normal_lhood", "<f4", (30,)), ("tumor_lhood", "<f4", (30,)), ("max_gt", "<u4"),
                               ("qphred", "<i4"), ("from_ntype_qphred", "<i4"), ("ntype", "<u4"),
                               ("strand_bias", "<f4"), ("is_called", "<u4"), ("normal_alt_id", "<u4"),
                               ("tumor_alt_id", "<u4")])
assert DIGT_CALL_DTYPE.itemsize == 144 and SOMATIC_CALL_DTYPE.itemsize == 272

# every symbol include/strelka_amd.h declares (tests check the library exports all of them)
EXPORTS = [
    "sk_device_count", "sk_host_alloc", "sk_host_free", "sk_init", "sk_init_strict", "sk_check_device_errors", "sk_debug_force_device_libm", "sk_debug_set_g3_variant", "sk_shutdown", "sk_last_error", "sk_version", "sk_is_initialized", "sk_sync_mode", "sk_libm_restated", "sk_broker_client", "sk_broker_enable", "sk_broker_serve", "sk_broker_selftest", "sk_get_qscore_tables",
    "sk_score_alignments", "sk_score_alignments_dev", "sk_align_evmask_words", "sk_align_prepare", "sk_align_colmat_words", "sk_align_prepare_cols",
    "sk_bgzf_scan", "sk_bgzf_inflate", "sk_bgzf_inflate_prefixed", "sk_bgzf_inflate_dev", "sk_bam_header_end", "sk_bam_scan_records", "sk_bam_decode", "sk_bam_decode_kept", "sk_bam_decode_dev", "sk_normalize_alignments", "sk_normalize_alignments_dev",
    "sk_align_builder_create", "sk_align_builder_destroy", "sk_align_builder_clear", "sk_align_builder_append", "sk_align_builder_add_read",
    "sk_align_builder_finish", "sk_align_builder_error", "sk_align_builder_set_host_threads",
    "sk_align_scores_default", "sk_global_align",
    "sk_pileup_options_default", "sk_pileup_reads", "sk_pileup_reads_dev", "sk_pileup_scratch_bytes",
    "sk_pileup_stream_create", "sk_pileup_stream_destroy", "sk_pileup_stream_begin_region", "sk_pileup_stream_push", "sk_pileup_stream_push_begin", "sk_pileup_stream_push_finish", "sk_pileup_stream_enable_evs_words", "sk_gvcf_site_summaries", "sk_gvcf_site_summaries_dev", "sk_gvcf_plain_runs_dev", "sk_gvcf_plain_runs", "sk_pileup_stream_set_gvcf_block_options",
    "sk_somatic_pileup_stream_create", "sk_somatic_pileup_stream_destroy", "sk_somatic_pileup_stream_begin_region", "sk_somatic_pileup_stream_push", "sk_somatic_pileup_stream_push_begin", "sk_somatic_pileup_stream_push_finish",
    "sk_realign_options_default", "sk_realign_job_create", "sk_realign_job_destroy", "sk_realign_job_error",
    "sk_realign_job_set_reference", "sk_realign_job_set_indels", "sk_realign_job_add_read", "sk_realign_job_add_reads", "sk_realign_job_get_batch",
    "sk_realign_job_finish", "sk_realign_job_run", "sk_realign_job_n_reads", "sk_realign_job_read_result",
    "sk_realign_job_clear_reads", "sk_realign_job_rescore", "sk_realign_job_indels_consulted", "sk_realign_job_enumeration_counts", "sk_realign_device_job_counts", "sk_realign_reference_reads_outside", "sk_realign_job_stage3_counts", "sk_make_start_pos_alignment", "sk_get_end_pin_start_pos",
    "sk_germline_options_default", "sk_dependent_eprob", "sk_dependent_eprob_dev", "sk_site_digt_call",
    "sk_site_digt_call_dev", "sk_site_digt_call_fused", "sk_site_digt_call_fused_dev",
    "sk_somatic_snv_options_default", "sk_somatic_snv_call_batch", "sk_somatic_snv_call_batch_dev",
    "sk_somatic_snv_call_tiers", "sk_somatic_snv_call_tiers_dev", "sk_somatic_snv_tiers_scratch_bytes",
    "sk_indel_options_default", "sk_somatic_indel_options_default", "sk_indel_grid_lhood", "sk_indel_grid_lhood_dev",
    "sk_somatic_indel_call_batch", "sk_somatic_indel_call_tiers", "sk_allele_group_genotype_lhoods", "sk_allele_group_genotype_lhoods_dev",
    "sk_allele_group_genotype_lhoods_wide", "sk_allele_group_genotype_lhoods_wide_dev",
    "sk_allele_group_genotype_lhoods_xwide", "sk_allele_group_genotype_lhoods_xwide_dev",
    "sk_discover_indels_and_mismatches", "sk_global_align_scratch_bytes", "sk_global_align_dev", "sk_bai_query", "sk_bam_region_filter", "sk_gvcf_block_sites", "sk_gvcf_block_sites_dev",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("strelka_amd: %s is missing -- run `python -m strelka_amd.build` (there is no CPU fallback)"
                               % LIB_PATH)
        try:
            # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same soname as /opt/rocm's).  When
            # torch is going to be used in this process (bench.py, device.py) it must be loaded FIRST so that this
            # library binds to the runtime torch initialises; loading /opt/rocm's copy first leaves torch without GPUs.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.sk_last_error.restype = C.c_char_p
        L.sk_align_builder_create.restype = c_void_p
        L.sk_align_builder_error.restype = C.c_char_p
        L.sk_align_builder_error.argtypes = [c_void_p]
        L.sk_align_builder_destroy.argtypes = [c_void_p]
        L.sk_align_builder_clear.argtypes = [c_void_p]
        L.sk_align_builder_add_read.argtypes = [c_void_p, c_void_p, c_void_p, C.c_int32, C.c_char_p, C.c_int32,
                                                C.c_int32, C.POINTER(CandidateAlignment), C.c_int32]
        L.sk_align_builder_finish.argtypes = [c_void_p, C.POINTER(AlignBatch)]
        L.sk_align_scores_default.argtypes = [C.POINTER(AlignScores)]
        L.sk_global_align.argtypes = [C.POINTER(GlobalAlignBatch), C.POINTER(AlignScores), c_void_p, c_void_p, c_void_p, c_void_p]
        L.sk_pileup_options_default.argtypes = [C.POINTER(PileupOptions)]
        L.sk_pileup_reads.argtypes = [C.POINTER(ReadBatchStruct), C.POINTER(PileupOptions), C.c_int, C.POINTER(PileupColumns)]
        L.sk_pileup_scratch_bytes.restype = C.c_int64
        L.sk_pileup_scratch_bytes.argtypes = [C.c_int32, C.c_int64, C.c_int32]
        L.sk_pileup_reads_dev.argtypes = [C.POINTER(ReadBatchStruct), C.c_int64, C.POINTER(PileupOptions), C.c_int,
                                          C.POINTER(PileupColumns), c_void_p, c_void_p]
        L.sk_realign_options_default.argtypes = [C.POINTER(RealignOptions)]
        L.sk_realign_job_create.restype = c_void_p
        L.sk_realign_job_create.argtypes = [C.POINTER(RealignOptions)]
        L.sk_realign_job_destroy.argtypes = [c_void_p]
        L.sk_realign_job_enumeration_counts.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
        L.sk_realign_device_job_counts.argtypes = [c_void_p, c_void_p, c_void_p]
        L.sk_realign_device_job_counts.restype = None
        L.sk_realign_reference_reads_outside.argtypes = []
        L.sk_realign_reference_reads_outside.restype = C.c_int64
        L.sk_realign_job_stage3_counts.argtypes = [c_void_p, c_void_p, c_void_p]
        L.sk_realign_job_indels_consulted.argtypes = [c_void_p, c_void_p, C.c_int32]
        L.sk_realign_job_error.restype = C.c_char_p
        L.sk_realign_job_error.argtypes = [c_void_p]
        L.sk_realign_job_set_reference.argtypes = [c_void_p, C.c_char_p, C.c_int32, C.c_int32]
        L.sk_realign_job_set_indels.argtypes = [c_void_p, C.POINTER(IndelInfo), C.c_int32]
        L.sk_realign_job_add_read.argtypes = [c_void_p, C.POINTER(ReadInput)]
        L.sk_realign_job_add_reads.argtypes = [c_void_p, C.POINTER(ReadInput), C.c_int32]
        L.sk_realign_job_get_batch.argtypes = [c_void_p, C.POINTER(AlignBatch)]
        L.sk_realign_job_finish.argtypes = [c_void_p, c_void_p]
        L.sk_realign_job_run.argtypes = [c_void_p]
        L.sk_realign_job_n_reads.argtypes = [c_void_p]
        L.sk_realign_job_read_result.argtypes = [c_void_p, C.c_int32, C.POINTER(ReadResult)]
        L.sk_realign_job_clear_reads.argtypes = [c_void_p]
        L.sk_make_start_pos_alignment.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(IndelKey),
                                                  C.c_int32, C.POINTER(C.c_int32), C.POINTER(PathSeg), C.c_int32,
                                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.sk_get_end_pin_start_pos.argtypes = [C.POINTER(IndelKey), C.c_int32, C.c_uint32, C.c_int32, C.c_int32,
                                               C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.sk_score_alignments.argtypes = [C.POINTER(AlignBatch), c_void_p]
        L.sk_align_evmask_words.argtypes = [C.c_int32]
        L.sk_bgzf_scan.restype = C.c_int64
        L.sk_bgzf_scan.argtypes = [c_void_p, C.c_int64, c_void_p, c_void_p, C.c_int32]
        L.sk_bgzf_inflate.argtypes = [c_void_p, c_void_p, c_void_p, C.c_int32, c_void_p]
        L.sk_bgzf_inflate_dev.argtypes = [c_void_p, c_void_p, c_void_p, C.c_int32, c_void_p, c_void_p, c_void_p]
        L.sk_bam_header_end.restype = C.c_int64
        L.sk_bam_header_end.argtypes = [c_void_p, C.c_int64]
        L.sk_bam_scan_records.restype = C.c_int64
        L.sk_bam_scan_records.argtypes = [c_void_p, C.c_int64, C.c_int64, c_void_p, c_void_p, c_void_p, C.c_int32]
        L.sk_bam_decode.argtypes = [c_void_p, C.c_int64, c_void_p, C.c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.sk_bam_decode_dev.argtypes = [c_void_p, c_void_p, C.c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.sk_normalize_alignments.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32] + [c_void_p] * 7
        L.sk_normalize_alignments_dev.argtypes = [c_void_p, C.c_int32, C.c_int32, C.c_int32] + [c_void_p] * 8
        L.sk_align_colmat_words.restype = C.c_int64
        L.sk_align_colmat_words.argtypes = [c_void_p]
        L.sk_align_prepare_cols.argtypes = [c_void_p] * 4
        L.sk_align_prepare.argtypes = [C.POINTER(AlignBatch), c_void_p, c_void_p]
        L.sk_score_alignments_dev.argtypes = [C.POINTER(AlignBatch), c_void_p, c_void_p]
        L.sk_score_alignments_dev_generic.argtypes = [C.POINTER(AlignBatch), c_void_p, c_void_p]
        L.sk_dependent_eprob.argtypes = [C.POINTER(PileupBatch), C.POINTER(GermlineOptions), c_void_p]
        L.sk_dependent_eprob_dev.argtypes = [C.POINTER(PileupBatch), C.POINTER(GermlineOptions), c_void_p, c_void_p,
                                             c_void_p]
        L.sk_site_digt_call.argtypes = [C.POINTER(PileupBatch), C.POINTER(GermlineOptions), c_void_p]
        L.sk_site_digt_call_dev.argtypes = [C.POINTER(PileupBatch), C.POINTER(GermlineOptions), c_void_p, c_void_p]
        L.sk_site_digt_call_fused.argtypes = [C.POINTER(PileupBatch), C.POINTER(GermlineOptions), c_void_p, c_void_p]
        L.sk_site_digt_call_fused_dev.argtypes = [C.POINTER(PileupBatch), C.POINTER(GermlineOptions), c_void_p, c_void_p,
                                                  C.c_int, c_void_p, C.c_int64, c_void_p]
        L.sk_somatic_snv_call_batch.argtypes = [C.POINTER(PileupBatch), C.POINTER(PileupBatch),
                                                C.POINTER(SomaticSnvOptions), C.c_int, c_void_p]
        L.sk_somatic_snv_call_batch_dev.argtypes = [C.POINTER(PileupBatch), C.POINTER(PileupBatch),
                                                    C.POINTER(SomaticSnvOptions), C.c_int, c_void_p, c_void_p, c_void_p]
        L.sk_indel_grid_lhood.argtypes = [C.POINTER(ReadScoreBatch), C.POINTER(IndelOptions), C.c_int, c_void_p]
        L.sk_indel_grid_lhood_dev.argtypes = [C.POINTER(ReadScoreBatch), C.POINTER(IndelOptions), C.c_int, c_void_p, c_void_p]
        L.sk_somatic_indel_call_batch.argtypes = [C.POINTER(ReadScoreBatch), C.POINTER(ReadScoreBatch),
                                                  C.POINTER(IndelOptions), C.POINTER(IndelOptions),
                                                  C.POINTER(SomaticIndelOptions), c_void_p, C.c_int, c_void_p]
        L.sk_allele_group_genotype_lhoods.argtypes = [C.POINTER(AlleleGroupBatch), C.POINTER(IndelOptions), c_void_p]
        L.sk_allele_group_genotype_lhoods_dev.argtypes = [C.POINTER(AlleleGroupBatch), C.POINTER(IndelOptions), c_void_p,
                                                          c_void_p]
        L.sk_allele_group_genotype_lhoods_wide_dev.argtypes = [C.POINTER(AlleleGroupBatch), C.POINTER(IndelOptions), c_void_p,
                                                          c_void_p]
        L.sk_allele_group_genotype_lhoods_wide.argtypes = [C.POINTER(AlleleGroupBatch), C.POINTER(IndelOptions), c_void_p]
        _lib = L
    return _lib


class StrelkaAmdError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise StrelkaAmdError(lib().sk_last_error().decode("utf-8", "replace"))


def last_error():
    return lib().sk_last_error().decode("utf-8", "replace")


def init(device=0):
    _check(lib().sk_init(int(device)))


def init_strict(device=0):
    """sk_init_strict: also fails when the host libm is not the one the kernels restate (no bit-exactness otherwise)"""
    _check(lib().sk_init_strict(int(device)))


def shutdown():
    lib().sk_shutdown()


def germline_options():
    o = GermlineOptions()
    lib().sk_germline_options_default(C.byref(o))
    return o


def somatic_snv_options():
    o = SomaticSnvOptions()
    lib().sk_somatic_snv_options_default(C.byref(o))
    return o


def qscore_tables():
    a = [np.zeros(71) for _ in range(3)]
    _check(lib().sk_get_qscore_tables(*[x.ctypes.data_as(c_void_p) for x in a]))
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


# ----------------------------------------------------------------------------------------------------------------------
# numpy-level views of the batches


class HostAlignBatch:
    """Host-side sk_align_batch held as numpy arrays."""

    def __init__(self, read_off, read_code, read_qual, hap_off, hap_code, cal_off, op_off, ops, max_read_len=None,
                 max_hap_len=None):
        self.read_off = np.ascontiguousarray(read_off, np.int64)
        self.read_code = np.ascontiguousarray(read_code, np.uint8)
        self.read_qual = np.ascontiguousarray(read_qual, np.uint8)
        self.hap_off = np.ascontiguousarray(hap_off, np.int64)
        self.hap_code = np.ascontiguousarray(hap_code, np.uint8)
        self.cal_off = np.ascontiguousarray(cal_off, np.int32)
        self.op_off = np.ascontiguousarray(op_off, np.int64)
        self.ops = np.ascontiguousarray(ops, SCORE_OP_DTYPE)
        self.n_reads = len(self.read_off) - 1
        self.n_cals = int(self.cal_off[-1]) if len(self.cal_off) else 0
        self.max_read_len = int(np.diff(self.read_off).max()) if max_read_len is None and self.n_reads else int(max_read_len or 0)
        self.max_hap_len = int(np.diff(self.hap_off).max()) if max_hap_len is None and self.n_reads else int(max_hap_len or 0)

        self.entries = self.evmask = None
        self.evmask_words = 0
        self.colmat = self.colmat_off = self.addmask = None

    def struct(self):
        return AlignBatch(self.n_reads, self.n_cals, len(self.ops), _p(self.read_off), _p(self.read_code),
                          _p(self.read_qual), _p(self.hap_off), _p(self.hap_code), _p(self.cal_off), _p(self.op_off),
                          _p(self.ops), self.max_read_len, self.max_hap_len,
                          None if self.entries is None else _p(self.entries),
                          None if self.evmask is None else _p(self.evmask), self.evmask_words,
                          None if self.colmat is None else _p(self.colmat),
                          None if self.colmat_off is None else _p(self.colmat_off),
                          None if self.addmask is None else _p(self.addmask))

    def prepare(self, columns=True):
        """sk_align_prepare (+ sk_align_prepare_cols): the device-ready forms of this batch -- transition entries + event masks,
        and the column form the streaming kernel reads"""
        if self.entries is None:
            self.evmask_words = lib().sk_align_evmask_words(self.max_read_len)
            ent = np.zeros(len(self.ops) + 2 * self.n_cals + 1, np.uint32)
            msk = np.zeros(self.n_reads * self.evmask_words + 1, np.uint32)
            s = self.struct()
            if lib().sk_align_prepare(C.byref(s), _p(ent), _p(msk)) != 0:
                raise StrelkaAmdError("sk_align_prepare failed")
            self.entries, self.evmask = ent, msk
        if self.colmat is None and columns:
            s = self.struct()
            words = lib().sk_align_colmat_words(C.byref(s))
            cm = np.zeros(words + 1, np.uint32)
            off = np.zeros(self.n_reads + 1, np.int64)
            am = np.zeros(self.n_reads * self.evmask_words + 1, np.uint32)
            if lib().sk_align_prepare_cols(C.byref(s), _p(cm), _p(off), _p(am)) != 0:
                raise StrelkaAmdError("sk_align_prepare_cols failed")
            self.colmat, self.colmat_off, self.addmask = cm, off, am
        return self


def score_alignments(batch):
    """Host-buffer entry point: returns float64[n_cals]."""
    out = np.zeros(batch.n_cals, np.float64)
    s = batch.struct()
    _check(lib().sk_score_alignments(C.byref(s), _p(out)))
    return out


class AlignBuilder:
    """sk_align_builder wrapper: add reads with reference-shaped candidate alignments, get a HostAlignBatch."""

    def __init__(self):
        self._b = lib().sk_align_builder_create()

    def __del__(self):
        try:
            if self._b:
                lib().sk_align_builder_destroy(self._b)
        except Exception:
            pass

    def clear(self):
        lib().sk_align_builder_clear(self._b)

    @staticmethod
    def _key(k, keep):
        if k is None:
            return IndelKey(0, 0, 0, 0, None, 0)
        seq = k.get("ins_seq", "").encode()
        keep.append(seq)
        return IndelKey(k["pos"], k["type"], k.get("del_len", 0), len(seq), seq, int(k.get("is_candidate", 1)))

    def add_read(self, read_code, read_qual, ref_seq, ref_offset, cals):
        """cals: list of dict(pos, path=[(type,len)...], indels=[dict(pos,type,del_len,ins_seq,is_candidate)],
        leading=None|dict, trailing=None|dict)"""
        read_code = np.ascontiguousarray(read_code, np.uint8)
        read_qual = np.ascontiguousarray(read_qual, np.uint8)
        keep = []
        arr = (CandidateAlignment * max(len(cals), 1))()
        for i, c in enumerate(cals):
            path = (PathSeg * max(len(c["path"]), 1))(*[PathSeg(t, l) for t, l in c["path"]])
            ind = (IndelKey * max(len(c["indels"]), 1))(*[self._key(k, keep) for k in c["indels"]])
            keep += [path, ind]
            arr[i] = CandidateAlignment(c["pos"], len(c["path"]), path, len(c["indels"]), ind,
                                        self._key(c.get("leading"), keep), self._key(c.get("trailing"), keep))
        ref = ref_seq.encode() if isinstance(ref_seq, str) else bytes(ref_seq)
        rc = lib().sk_align_builder_add_read(self._b, _p(read_code), _p(read_qual), len(read_code), ref, int(ref_offset),
                                             len(ref), arr, len(cals))
        if rc != 0:
            raise StrelkaAmdError(lib().sk_align_builder_error(self._b).decode())

    def finish(self):
        s = AlignBatch()
        if lib().sk_align_builder_finish(self._b, C.byref(s)) != 0:
            raise StrelkaAmdError("sk_align_builder_finish failed")
        return _host_batch_from_struct(s)


def _host_batch_from_struct(s):
    def arr(ptr, n, dt):
        if n == 0 or not ptr:
            return np.zeros(0, dt)
        buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).copy()

    n, nc = s.n_reads, s.n_cals
    read_off = arr(s.read_off, n + 1, np.int64)
    hap_off = arr(s.hap_off, n + 1, np.int64)
    return HostAlignBatch(read_off, arr(s.read_code, int(read_off[-1]), np.uint8),
                          arr(s.read_qual, int(read_off[-1]), np.uint8), hap_off,
                          arr(s.hap_code, int(hap_off[-1]), np.uint8), arr(s.cal_off, n + 1, np.int32),
                          arr(s.op_off, nc + 1, np.int64), arr(s.ops, s.n_ops, SCORE_OP_DTYPE), s.max_read_len,
                          s.max_hap_len)


class HostPileupBatch:
    def __init__(self, call_off, calls, ref_base, de=None, ploidy=None):
        self.call_off = np.ascontiguousarray(call_off, np.int64)
        self.calls = np.ascontiguousarray(calls, np.uint16)
        self.ref_base = np.ascontiguousarray(ref_base, np.uint8)
        self.de = None if de is None else np.ascontiguousarray(de, np.float32)
        self.ploidy = None if ploidy is None else np.ascontiguousarray(ploidy, np.uint8)
        self.n_loci = len(self.call_off) - 1

    def struct(self):
        return PileupBatch(self.n_loci, _p(self.call_off), _p(self.calls), _p(self.de), _p(self.ref_base), _p(self.ploidy))


def dependent_eprob(batch, opt=None):
    opt = opt or germline_options()
    out = np.zeros(len(batch.calls), np.float32)
    s = batch.struct()
    _check(lib().sk_dependent_eprob(C.byref(s), C.byref(opt), _p(out)))
    return out


def site_digt_call(batch, opt=None):
    opt = opt or germline_options()
    out = np.zeros(batch.n_loci, DIGT_CALL_DTYPE)
    s = batch.struct()
    _check(lib().sk_site_digt_call(C.byref(s), C.byref(opt), _p(out)))
    return out


def site_digt_call_fused(batch, opt=None, want_de=False):
    """a9+a10 in one pass; returns (calls, de or None)."""
    opt = opt or germline_options()
    out = np.zeros(batch.n_loci, DIGT_CALL_DTYPE)
    de = np.zeros(len(batch.calls), np.float32) if want_de else None
    s = batch.struct()
    _check(lib().sk_site_digt_call_fused(C.byref(s), C.byref(opt), _p(out), _p(de)))
    return out, de


def gvcf_site_summaries(batch, genotypes):
    """sk_gvcf_site_summary of every locus of `batch` (its cleaned columns) given the genotype records sk_site_digt_call_fused left"""
    out = np.zeros(batch.n_loci, GVCF_SITE_SUMMARY_DTYPE)
    s = batch.struct()
    g = np.ascontiguousarray(genotypes)
    _check(lib().sk_gvcf_site_summaries(C.byref(s), _p(g), _p(out)))
    return out


def gvcf_plain_runs(summary, clean_count, raw_count, mapq_count, opt, library=None):
    """sk_gvcf_run of every site (library: the ctypes handle to drive -- default the product library; the tests also drive the CPU double)"""
    L = library or lib()
    n = len(summary)
    out = np.zeros(n, GVCF_RUN_DTYPE)
    sm, cc, rc, mq = (np.ascontiguousarray(a, dt) for a, dt in ((summary, GVCF_SITE_SUMMARY_DTYPE), (clean_count, np.uint32), (raw_count, np.uint32),
                                                                (mapq_count, np.uint32)))
    L.sk_gvcf_plain_runs.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_int32, c_void_p]
    L.sk_last_error.restype = C.c_char_p
    if L.sk_gvcf_plain_runs(_p(sm), _p(cc), _p(rc), _p(mq), C.byref(opt), n, _p(out)) != 0:
        raise RuntimeError(L.sk_last_error().decode())
    return out


def somatic_snv_call(normal, tumor, opt=None, is_forced_output=False):
    opt = opt or somatic_snv_options()
    out = np.zeros(normal.n_loci, SOMATIC_CALL_DTYPE)
    sn, st = normal.struct(), tumor.struct()
    _check(lib().sk_somatic_snv_call_batch(C.byref(sn), C.byref(st), C.byref(opt), int(is_forced_output), _p(out)))
    return out


SOMATIC_GENOTYPE_DTYPE = np.dtype([
    ("ref_gt", np.uint32), ("snv_tier", np.uint8), ("snv_from_ntype_tier", np.uint8), ("is_forced_output", np.uint8),
    ("is_computed", np.uint8), ("ntype", np.uint32), ("max_gt", np.uint32), ("qphred", np.int32),
    ("from_ntype_qphred", np.int32), ("nonsomatic_qphred", np.int32), ("normal_alt_id", np.uint32),
    ("tumor_alt_id", np.uint32), ("_pad", np.int32), ("strand_bias", np.float64)], align=True)
assert SOMATIC_GENOTYPE_DTYPE.itemsize == 48


def somatic_snv_call_tiers(normal_t1, tumor_t1, normal_t2=None, tumor_t2=None, opt=None, is_forced_output=None,
                           is_compute_nonsomatic=False):
    """sk_somatic_snv_call_tiers: the whole of position_somatic_snv_call (both tiers)."""
    opt = opt or somatic_snv_options()
    n = normal_t1.n_loci
    out = np.zeros(n, SOMATIC_GENOTYPE_DTYPE)
    s1, s2 = normal_t1.struct(), tumor_t1.struct()
    if normal_t2 is not None:
        s3, s4 = normal_t2.struct(), tumor_t2.struct()
        p3, p4 = C.byref(s3), C.byref(s4)
    else:
        p3 = p4 = None
    forced = None if is_forced_output is None else np.ascontiguousarray(is_forced_output, np.uint8)
    L = lib()
    L.sk_somatic_snv_call_tiers.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_void_p]
    _check(L.sk_somatic_snv_call_tiers(C.cast(C.byref(s1), C.c_void_p), C.cast(C.byref(s2), C.c_void_p),
                                       C.cast(p3, C.c_void_p) if p3 is not None else None,
                                       C.cast(p4, C.c_void_p) if p4 is not None else None,
                                       C.cast(C.byref(opt), C.c_void_p), _p(forced), int(bool(is_compute_nonsomatic)), _p(out)))
    return out


def make_call(q, base, fwd=1, nmm=0, filt=0, tscf=0):
    """SK_MAKE_CALL (vectorised)."""
    q, base, fwd, nmm, filt, tscf = [np.asarray(x, np.uint16) for x in (q, base, fwd, nmm, filt, tscf)]
    return ((q & 0x3f) | ((base & 0xf) << 6) | ((fwd & 1) << 10) | ((nmm & 1) << 11) | ((filt & 1) << 12) |
            ((tscf & 1) << 13)).astype(np.uint16)


# ----------------------------------------------------------------------------------------------------------------------
# indels

def indel_options(is_somatic=False, exact=True):
    """sk_indel_options_default; `exact` (the default HERE: these helpers serve tests that compare doubles bit for bit with the oracle)
    asks for the reference's operation order -- the library's own default is the fast form (sk_indel_options.fast_form)."""
    o = IndelOptions()
    lib().sk_indel_options_default(C.byref(o), int(is_somatic))
    if exact:
        o.fast_form = 0
    return o


def somatic_indel_options():
    o = SomaticIndelOptions()
    lib().sk_somatic_indel_options_default(C.byref(o))
    return o


class HostReadScoreBatch:
    def __init__(self, read_off, ref_lnp, indel_lnp, alt_lnp, non_ambig, read_length, read_flags, del_len, ins_len,
                 is_breakpoint=None):
        self.read_off = np.ascontiguousarray(read_off, np.int64)
        self.ref_lnp = np.ascontiguousarray(ref_lnp, np.float32)
        self.indel_lnp = np.ascontiguousarray(indel_lnp, np.float32)
        self.alt_lnp = None if alt_lnp is None else np.ascontiguousarray(alt_lnp, np.float32)
        self.non_ambig = np.ascontiguousarray(non_ambig, np.uint16)
        self.read_length = np.ascontiguousarray(read_length, np.uint16)
        self.read_flags = np.ascontiguousarray(read_flags, np.uint8)
        self.del_len = np.ascontiguousarray(del_len, np.uint32)
        self.ins_len = np.ascontiguousarray(ins_len, np.uint32)
        self.is_breakpoint = None if is_breakpoint is None else np.ascontiguousarray(is_breakpoint, np.uint8)
        self.n_indels = len(self.read_off) - 1

    def struct(self):
        return ReadScoreBatch(self.n_indels, _p(self.read_off), _p(self.ref_lnp), _p(self.indel_lnp), _p(self.alt_lnp),
                              _p(self.non_ambig), _p(self.read_length), _p(self.read_flags), _p(self.del_len),
                              _p(self.ins_len), _p(self.is_breakpoint))


class AltAllele(C.Structure):
    _fields_ = [("begin_pos", C.c_int32), ("end_pos", C.c_int32), ("is_mismatch", C.c_int32)]


class SomaticIndelBatch(C.Structure):
    _fields_ = [("n_indels", C.c_int32), ("normal", ReadScoreBatch), ("tumor", ReadScoreBatch),
                ("normal_alt_key", c_void_p), ("normal_alt_lnp", c_void_p), ("tumor_alt_key", c_void_p),
                ("tumor_alt_lnp", c_void_p), ("alt_off", c_void_p), ("alt_alleles", c_void_p),
                ("indel_to_ref_error_prob", c_void_p), ("is_forced_output", c_void_p)]


SOMATIC_INDEL_GENOTYPE_DTYPE = np.dtype([("sindel_tier", np.uint8), ("sindel_from_ntype_tier", np.uint8),
                                         ("is_forced_output", np.uint8), ("is_overlap", np.uint8), ("ntype", np.uint32),
                                         ("max_gt", np.uint32), ("qphred", np.int32), ("from_ntype_qphred", np.int32)])
assert SOMATIC_INDEL_GENOTYPE_DTYPE.itemsize == 20


class HostSomaticIndelBatch:
    """sk_somatic_indel_batch from per-indel cases (strelka_amd.synth.somatic_indel_cases layout)."""

    def __init__(self, cases):
        n = len(cases)
        self.n = n

        def sample(name):
            off = np.zeros(n + 1, np.int64)
            np.cumsum([len(c[name]["ref_lnp"]) for c in cases], out=off[1:])
            cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(c[name][k]).reshape(len(c[name]["ref_lnp"]), -1) for c in cases] or [np.zeros((0, 1))]), dt)
            alt_key = cat("alt_key", np.int32).reshape(-1, 2)
            alt_lnp = cat("alt_lnp", np.float32).reshape(-1, 2)
            best = np.where(alt_key >= 0, alt_lnp, -np.inf).max(axis=1) if len(alt_key) else np.zeros(0)
            best = np.where((alt_key >= 0).any(axis=1), best, np.nan).astype(np.float32) if len(alt_key) else np.zeros(0, np.float32)
            t1 = cat("is_tier1", np.uint8).ravel()
            b = HostReadScoreBatch(off, cat("ref_lnp", np.float32).ravel(), cat("indel_lnp", np.float32).ravel(), best,
                                   cat("non_ambig", np.uint16).ravel(), cat("read_length", np.uint16).ravel(),
                                   (t1 & 1) | 2, [c["del_len"] for c in cases], [c["ins_len"] for c in cases])
            return b, np.ascontiguousarray(alt_key), np.ascontiguousarray(alt_lnp)

        self.normal, self.n_alt_key, self.n_alt_lnp = sample("normal")
        self.tumor, self.t_alt_key, self.t_alt_lnp = sample("tumor")
        self.alt_off = np.zeros(n + 1, np.int64)
        np.cumsum([len(c["alt_keys"]) for c in cases], out=self.alt_off[1:])
        keys = [k for c in cases for k in c["alt_keys"]]
        self.alleles = np.ascontiguousarray(np.array(keys, np.int32).reshape(-1, 3))
        self.err = np.ascontiguousarray([c["indel_to_ref_error_prob"] for c in cases], np.float64)
        self.forced = np.ascontiguousarray([c["forced"] for c in cases], np.uint8)

    def struct(self):
        return SomaticIndelBatch(self.n, self.normal.struct(), self.tumor.struct(), _p(self.n_alt_key), _p(self.n_alt_lnp),
                                 _p(self.t_alt_key), _p(self.t_alt_lnp), _p(self.alt_off), _p(self.alleles), _p(self.err),
                                 _p(self.forced))


def somatic_indel_call_tiers(cases, normal_opt=None, tumor_opt=None, sopt=None, use_tier2_evidence=True):
    """sk_somatic_indel_call_tiers: the whole of get_somatic_indel per candidate indel."""
    if normal_opt is None:
        normal_opt = indel_options(True)
        normal_opt.min_read_bp_flank = 1
    tumor_opt = tumor_opt or indel_options(True)
    sopt = sopt or somatic_indel_options()
    b = HostSomaticIndelBatch(cases)
    out = np.zeros(b.n, SOMATIC_INDEL_GENOTYPE_DTYPE)
    s = b.struct()
    _check(lib().sk_somatic_indel_call_tiers(C.byref(s), C.byref(normal_opt), C.byref(tumor_opt), C.byref(sopt),
                                             int(bool(use_tier2_evidence)), _p(out)))
    return out


def indel_grid_lhood(batch, opt=None, is_include_tier2=False):
    opt = opt or indel_options(True)
    out = np.zeros((batch.n_indels, 21), np.float64)
    s = batch.struct()
    _check(lib().sk_indel_grid_lhood(C.byref(s), C.byref(opt), int(is_include_tier2), _p(out)))
    return out


def somatic_indel_call(normal, tumor, indel_to_ref_error_prob, normal_opt=None, tumor_opt=None, sopt=None,
                       is_include_tier2=False):
    if normal_opt is None:
        normal_opt = indel_options(True)
        normal_opt.min_read_bp_flank = 1
    tumor_opt = tumor_opt or indel_options(True)
    sopt = sopt or somatic_indel_options()
    out = np.zeros(normal.n_indels, SOMATIC_INDEL_CALL_DTYPE)
    err = np.ascontiguousarray(indel_to_ref_error_prob, np.float64)
    sn, st = normal.struct(), tumor.struct()
    _check(lib().sk_somatic_indel_call_batch(C.byref(sn), C.byref(st), C.byref(normal_opt), C.byref(tumor_opt),
                                             C.byref(sopt), _p(err), int(is_include_tier2), _p(out)))
    return out


class HostAlleleGroupBatch:
    def __init__(self, read_off, n_alt, ploidy, del_len, ins_len, ref_lnp, allele_lnp, non_ambig, read_length, read_flags, width=MAX_ALT):
        assert width in (MAX_ALT, MAX_ALT_WIDE, MAX_ALT_XWIDE)
        self.width = width  # columns of the per-allele arrays: MAX_ALT, or MAX_ALT_WIDE / MAX_ALT_XWIDE for the wide entry points
        self.read_off = np.ascontiguousarray(read_off, np.int64)
        self.n_alt = np.ascontiguousarray(n_alt, np.uint8)
        self.ploidy = np.ascontiguousarray(ploidy, np.uint8)
        self.del_len = np.ascontiguousarray(del_len, np.uint32).reshape(-1, width)
        self.ins_len = np.ascontiguousarray(ins_len, np.uint32).reshape(-1, width)
        self.ref_lnp = np.ascontiguousarray(ref_lnp, np.float32).reshape(-1, width)
        self.allele_lnp = np.ascontiguousarray(allele_lnp, np.float32).reshape(-1, width)
        self.non_ambig = np.ascontiguousarray(non_ambig, np.uint16)
        self.read_length = np.ascontiguousarray(read_length, np.uint16)
        self.read_flags = np.ascontiguousarray(read_flags, np.uint8)
        self.n_groups = len(self.read_off) - 1

    def struct(self):
        return AlleleGroupBatch(self.n_groups, _p(self.read_off), _p(self.n_alt), _p(self.ploidy), _p(self.del_len),
                                _p(self.ins_len), _p(self.ref_lnp), _p(self.allele_lnp), _p(self.non_ambig),
                                _p(self.read_length), _p(self.read_flags))


def allele_group_genotype_lhoods(batch, opt=None):
    opt = opt or indel_options(False)
    width = getattr(batch, "width", MAX_ALT)
    dtype, fn = {MAX_ALT: (ALLELE_GROUP_CALL_DTYPE, "sk_allele_group_genotype_lhoods"),
                 MAX_ALT_WIDE: (ALLELE_GROUP_CALL_WIDE_DTYPE, "sk_allele_group_genotype_lhoods_wide"),
                 MAX_ALT_XWIDE: (ALLELE_GROUP_CALL_XWIDE_DTYPE, "sk_allele_group_genotype_lhoods_xwide")}[width]
    out = np.zeros(batch.n_groups, dtype)
    s = batch.struct()
    f = getattr(lib(), fn)
    f.argtypes = [C.POINTER(AlleleGroupBatch), C.POINTER(IndelOptions), c_void_p]
    _check(f(C.byref(s), C.byref(opt), _p(out)))
    return out


# ---------------------------------------------------------------------------------------------------- realign job

def cigar_to_path(cigar):
    out, num = [], ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            out.append((CIGAR_CHARS.index(ch), int(num)))
            num = ""
    return out


def path_to_cigar(path):
    return "".join("%d%s" % (l, CIGAR_CHARS[t]) for t, l in path)


def realign_options(**kw):
    o = RealignOptions()
    lib().sk_realign_options_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _ckeys(keys, keep):
    arr = (IndelKey * max(len(keys), 1))()
    for i, k in enumerate(keys):
        arr[i] = AlignBuilder._key(k, keep)
    return arr


def make_start_pos_alignment(ref_start_pos, read_start_pos, is_fwd, read_length, indels):
    """-> dict(pos, path, leading, trailing) (indices into `indels`) or None when the arguments are rejected"""
    keep = []
    arr = _ckeys(indels, keep)
    cap = 2 * len(indels) + 4
    path = (PathSeg * cap)()
    pos, lead, trail = C.c_int32(), C.c_int32(), C.c_int32()
    n = lib().sk_make_start_pos_alignment(ref_start_pos, read_start_pos, int(is_fwd), read_length, arr, len(indels),
                                          C.byref(pos), path, cap, C.byref(lead), C.byref(trail))
    if n < 0:
        return None
    return dict(pos=pos.value, path=[(path[i].type, path[i].length) for i in range(n)], leading=lead.value,
                trailing=trail.value)


def get_end_pin_start_pos(indels, read_length, ref_end_pos, read_end_pos):
    keep = []
    arr = _ckeys(indels, keep)
    a, b = C.c_int32(), C.c_int32()
    if lib().sk_get_end_pin_start_pos(arr, len(indels), read_length, ref_end_pos, read_end_pos, C.byref(a), C.byref(b)):
        return None
    return a.value, b.value


class RealignJob:
    """sk_realign_job wrapper (realignAndScoreRead as a batched job)."""

    def __init__(self, opt=None):
        self.opt = opt or realign_options()
        self._j = lib().sk_realign_job_create(C.byref(self.opt))
        if not self._j:
            raise StrelkaAmdError("sk_realign_job_create failed")
        self._keep = []

    def __del__(self):
        try:
            if self._j:
                lib().sk_realign_job_destroy(self._j)
        except Exception:
            pass

    def _err(self):
        return StrelkaAmdError(lib().sk_realign_job_error(self._j).decode())

    def set_reference(self, seq, offset=0):
        b = seq.encode()
        if lib().sk_realign_job_set_reference(self._j, b, offset, len(b)):
            raise self._err()

    def set_indels(self, indels):
        """indels: dicts(pos,type,del_len,ins_seq,is_candidate, r2i, i2r [, arid, hap, bypass, forced, ndfr])"""
        keep = []
        arr = (IndelInfo * max(len(indels), 1))()
        for i, d in enumerate(indels):
            a = arr[i]
            a.key = AlignBuilder._key(d, keep)
            a.ref_to_indel_log_prob = d.get("r2i", 0.0)
            a.indel_to_ref_log_prob = d.get("i2r", 0.0)
            a.active_region_id = d.get("arid", -1)
            hap = d.get("hap", 0)
            byp = d.get("bypass", 0)
            for s in range(MAX_SAMPLES):
                a.haplotype_id[s] = (hap[s] if s < len(hap) else 0) if isinstance(hap, (list, tuple)) else hap
                a.is_haplotyping_bypassed[s] = (byp[s] if s < len(byp) else 0) if isinstance(byp, (list, tuple)) else byp
            a.is_forced_output = int(d.get("forced", 0))
            a.not_discovered_from_reads = int(d.get("ndfr", 0))
        if lib().sk_realign_job_set_indels(self._j, arr, len(indels)):
            raise self._err()
        self._n_indels = len(indels)

    def add_read(self, read_code, read_qual, pos, path, is_fwd=True, map_level=1, sample=0, realign_range=(0, 1 << 30),
                 observed=()):
        code = np.ascontiguousarray(read_code, np.uint8)
        qual = np.ascontiguousarray(read_qual, np.uint8)
        segs = (PathSeg * max(len(path), 1))(*[PathSeg(t, l) for t, l in path])
        obs = (C.c_int32 * max(len(observed), 1))(*observed)
        r = ReadInput(_p(code), _p(qual), len(code), pos, len(path), segs, int(is_fwd), map_level, sample,
                      realign_range[0], realign_range[1], len(observed), obs)
        i = lib().sk_realign_job_add_read(self._j, C.byref(r))
        if i < 0:
            raise self._err()
        return i

    def add_reads(self, reads):
        """reads: [(read_code, read_qual, pos, path, is_fwd, map_level, sample, realign_range, observed)] -> first index"""
        n = len(reads)
        arr = (ReadInput * max(n, 1))()
        keep = []
        for i, (read_code, read_qual, pos, path, is_fwd, map_level, sample, realign_range, observed) in enumerate(reads):
            code = np.ascontiguousarray(read_code, np.uint8)
            qual = np.ascontiguousarray(read_qual, np.uint8)
            segs = (PathSeg * max(len(path), 1))(*[PathSeg(t, l) for t, l in path])
            obs = (C.c_int32 * max(len(observed), 1))(*observed)
            keep.append((code, qual, segs, obs))
            arr[i] = ReadInput(_p(code), _p(qual), len(code), pos, len(path), segs, int(is_fwd), map_level, sample,
                               realign_range[0], realign_range[1], len(observed), obs)
        first = lib().sk_realign_job_add_reads(self._j, arr, n)
        if first < 0:
            raise self._err()
        return first

    def batch(self):
        s = AlignBatch()
        if lib().sk_realign_job_get_batch(self._j, C.byref(s)):
            raise self._err()
        return _host_batch_from_struct(s)

    def finish(self, scores):
        scores = np.ascontiguousarray(scores, np.float64)
        if lib().sk_realign_job_finish(self._j, _p(scores)):
            raise self._err()

    def run(self):
        if lib().sk_realign_job_run(self._j):
            raise self._err()

    def n_reads(self):
        return lib().sk_realign_job_n_reads(self._j)

    def indels_consulted(self):
        """uint8 per indel of the table as given to set_indels: candidate status consulted by any read so far"""
        out = np.zeros(max(self._n_indels, 1), np.uint8)
        if lib().sk_realign_job_indels_consulted(self._j, _p(out), self._n_indels):
            raise self._err()
        return out[:self._n_indels]

    def enumeration_counts(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        lib().sk_realign_job_enumeration_counts(self._j, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    @staticmethod
    def device_job_counts():
        """(jobs run as one fixed sequence with one wait, of those run again the staged way, jobs run the staged way) of this process"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        lib().sk_realign_device_job_counts(C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def stage3_counts(self):
        a, b = C.c_int64(), C.c_int64()
        lib().sk_realign_job_stage3_counts(self._j, C.byref(a), C.byref(b))
        return a.value, b.value

    def clear_reads(self):
        lib().sk_realign_job_clear_reads(self._j)

    def result(self, i):
        r = ReadResult()
        if lib().sk_realign_job_read_result(self._j, i, C.byref(r)):
            raise StrelkaAmdError("bad read index")
        scores = []
        for q in range(r.n_scores):
            s = r.scores[q]
            scores.append(dict(indel=s.indel, ref_lnp=s.ref_lnp, indel_lnp=s.indel_lnp, non_ambig=s.non_ambig,
                               read_length=s.read_length, is_tier1_read=s.is_tier1_read, is_fwd_strand=s.is_fwd_strand,
                               read_pos=s.read_pos, edge_dist=s.distance_from_closest_read_edge,
                               alt=[(s.alt_indel[a], s.alt_lnp[a]) for a in range(s.n_alt)]))
        return dict(n_cals=r.n_candidate_alignments, is_realigned=bool(r.is_realigned), pos=r.realign_pos,
                    path=[(r.realign_path[i].type, r.realign_path[i].length) for i in range(r.realign_n_seg)],
                    max_score=r.max_score, scores=scores, suboverlap=[r.suboverlap[i] for i in range(r.n_suboverlap)],
                    warn_origin_skip=bool(r.warn_origin_skip), warn_max_toggle_depth=bool(r.warn_max_toggle_depth))


# ---------------------------------------------------------------------------------------------------- pileup (a8)

def pileup_options(**kw):
    o = PileupOptions()
    lib().sk_pileup_options_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def pileup_reads(rb, opt, mode):
    """rb: synth.ReadBatch (host arrays) -> (call_off, calls, spandel_count, submapped_count)"""
    ref = np.frombuffer(rb.ref_seq.encode(), np.uint8).copy()
    s = ReadBatchStruct(rb.n_reads, _p(rb.read_off), _p(rb.read_code), _p(rb.read_qual), _p(rb.path_off), _p(rb.path),
                        _p(rb.pos), _p(rb.is_fwd), _p(rb.mapq), _p(rb.map_level), _p(ref), rb.ref_offset, len(ref),
                        None if rb.cand_snv_mask is None else _p(rb.cand_snv_mask))
    n_loci = opt.report_end - opt.report_begin
    cap = 2 * rb.n_bases + 1
    call_off = np.zeros(n_loci + 1, np.int64)
    calls = np.zeros(cap, np.uint16)
    sd = np.zeros(max(n_loci, 1), np.uint32)
    sm = np.zeros(max(n_loci, 1), np.uint32)
    out = PileupColumns(n_loci, cap, _p(call_off), _p(calls), _p(sd), _p(sm))
    _check(lib().sk_pileup_reads(C.byref(s), C.byref(opt), mode, C.byref(out)))
    return call_off, calls[:call_off[-1]].copy(), sd[:n_loci], sm[:n_loci]


class PileupStream:
    """sk_pileup_stream_*: one sample's pileup over a region, pushed window by window (row a8 chained into a9+a10).
    `library`: the ctypes handle to drive (default: the product library; the tests also drive the CPU double with it)."""

    def __init__(self, opt, germline_opt=None, library=None, evs_words=False, gvcf_block_opt=None):
        self.L = library or lib()
        L = self.L
        self.evs_words = evs_words
        self.gvcf_block_opt = gvcf_block_opt
        L.sk_pileup_stream_enable_evs_words.argtypes = [c_void_p, C.c_int]
        L.sk_pileup_stream_create.restype = c_void_p
        L.sk_pileup_stream_create.argtypes = [C.POINTER(PileupOptions), c_void_p]
        L.sk_pileup_stream_destroy.argtypes = [c_void_p]
        L.sk_pileup_stream_begin_region.argtypes = [c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.sk_pileup_stream_push.argtypes = [c_void_p, C.POINTER(ReadBatchStruct), C.c_int32, C.c_int32, C.c_int32, c_void_p, C.c_int32,
                                            C.c_int32, C.c_int32, c_void_p, C.POINTER(PileupWindow)]
        L.sk_last_error.restype = C.c_char_p
        self.genotype = germline_opt is not None
        self.h = L.sk_pileup_stream_create(C.byref(opt), C.byref(germline_opt) if self.genotype else None)
        if not self.h:
            raise RuntimeError(L.sk_last_error().decode())
        if gvcf_block_opt is not None:
            L.sk_pileup_stream_set_gvcf_block_options.argtypes = [c_void_p, c_void_p]
            if L.sk_pileup_stream_set_gvcf_block_options(self.h, C.byref(gvcf_block_opt)) != 0:
                raise RuntimeError(L.sk_last_error().decode())
        if evs_words and L.sk_pileup_stream_enable_evs_words(self.h, 1) != 0:
            raise RuntimeError(L.sk_last_error().decode())

    def close(self):
        if self.h:
            self.L.sk_pileup_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.sk_last_error().decode())

    def begin_region(self, ref_seq, ref_offset, report_begin, report_end, span=49):
        ref = ref_seq.encode()
        self._check(self.L.sk_pileup_stream_begin_region(self.h, ref, ref_offset, len(ref), report_begin, report_end, span))

    def push_raw(self, rb, final_to, span=49):
        """push without copying the window out (bench): -> (begin, end)"""
        s = ReadBatchStruct(rb.n_reads, _p(rb.read_off), _p(rb.read_code), _p(rb.read_qual), _p(rb.path_off), _p(rb.path),
                            _p(rb.pos), _p(rb.is_fwd), _p(rb.mapq), _p(rb.map_level), None, 0, 0, None)
        w = PileupWindow()
        self._check(self.L.sk_pileup_stream_push(self.h, C.byref(s), span, 0, 0, None, min(final_to, 2**31 - 1), 0, 0, None, C.byref(w)))
        return w.begin, w.end

    def push(self, rb, final_to, mask=None, mask_begin=0, ploidy=None, ploidy_begin=0, span=49, halves=False, between=None):
        """rb: synth.ReadBatch (its reference / mask fields are ignored) -> dict of numpy copies for [begin, end).
        halves: as sk_pileup_stream_push_begin + _finish, `between()` called while the window is in flight"""
        s = ReadBatchStruct(rb.n_reads, _p(rb.read_off), _p(rb.read_code), _p(rb.read_qual), _p(rb.path_off), _p(rb.path),
                            _p(rb.pos), _p(rb.is_fwd), _p(rb.mapq), _p(rb.map_level), None, 0, 0, None)
        w = PileupWindow()
        mask = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        ploidy = None if ploidy is None else np.ascontiguousarray(ploidy, np.uint8)
        if halves:
            self.L.sk_pileup_stream_push_begin.argtypes = self.L.sk_pileup_stream_push.argtypes[:-1]
            self.L.sk_pileup_stream_push_finish.argtypes = [c_void_p, C.POINTER(PileupWindow)]
            self._check(self.L.sk_pileup_stream_push_begin(self.h, C.byref(s), span, mask_begin, 0 if mask is None else len(mask), _p(mask),
                                                            min(final_to, 2**31 - 1), ploidy_begin, 0 if ploidy is None else len(ploidy),
                                                            _p(ploidy)))
            if between is not None:
                between()
            self._check(self.L.sk_pileup_stream_push_finish(self.h, C.byref(w)))
        else:
            self._check(self.L.sk_pileup_stream_push(self.h, C.byref(s), span, mask_begin, 0 if mask is None else len(mask), _p(mask),
                                                      min(final_to, 2**31 - 1), ploidy_begin, 0 if ploidy is None else len(ploidy), _p(ploidy),
                                                      C.byref(w)))
        n = w.end - w.begin

        def arr(ptr, dt, k):
            if k == 0 or not ptr:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(np.dtype(dt)))), shape=(k,)).copy()

        def rec(ptr, dt, k):
            if k == 0 or not ptr:
                return np.zeros(0, dt)
            buf = (C.c_char * (dt.itemsize * k)).from_address(ptr)
            return np.frombuffer(buf, dt, k).copy()
        o1 = arr(w.tier1_off, np.int64, n + 1)
        o2 = arr(w.tier2_off, np.int64, n + 1)
        return dict(begin=w.begin, end=w.end, tier1_off=o1, tier1_calls=arr(w.tier1_calls, np.uint16, int(o1[-1]) if n else 0),
                    tier2_off=o2, tier2_calls=arr(w.tier2_calls, np.uint16, int(o2[-1]) if n else 0),
                    spandel=arr(w.spandel_count, np.uint32, n), submapped=arr(w.submapped_count, np.uint32, n),
                    mapq_count=arr(w.mapq_count, np.uint32, n), mapq_zero=arr(w.mapq_zero_count, np.uint32, n),
                    mapq_sumsq=arr(w.mapq_sum_square, np.uint64, n), clean_count=arr(w.clean_count, np.uint32, n),
                    genotype=rec(w.genotype, DIGT_CALL_DTYPE, n) if self.genotype else None,
                    site_summary=rec(w.site_summary, GVCF_SITE_SUMMARY_DTYPE, n) if self.genotype else None,
                    gvcf_runs=rec(w.gvcf_runs, GVCF_RUN_DTYPE, n) if (self.genotype and self.gvcf_block_opt is not None) else None,
                    evs_off=arr(w.evs_off, np.int64, n + 1) if self.evs_words else None,
                    evs_words=(arr(w.evs_words, np.uint64, int(arr(w.evs_off, np.int64, n + 1)[-1]) if n else 0) if self.evs_words else None))


def _window_arrays(w, genotype_dtype=None):
    n = w.end - w.begin

    def arr(ptr, dt, k):
        if k == 0 or not ptr:
            return np.zeros(0, dt)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(np.dtype(dt)))), shape=(k,)).copy()
    o1 = arr(w.tier1_off, np.int64, n + 1)
    o2 = arr(w.tier2_off, np.int64, n + 1)
    return dict(begin=w.begin, end=w.end, tier1_off=o1, tier1_calls=arr(w.tier1_calls, np.uint16, int(o1[-1]) if n else 0),
                tier2_off=o2, tier2_calls=arr(w.tier2_calls, np.uint16, int(o2[-1]) if n else 0),
                spandel=arr(w.spandel_count, np.uint32, n), submapped=arr(w.submapped_count, np.uint32, n),
                mapq_count=arr(w.mapq_count, np.uint32, n), mapq_zero=arr(w.mapq_zero_count, np.uint32, n),
                mapq_sumsq=arr(w.mapq_sum_square, np.uint64, n), clean_count=arr(w.clean_count, np.uint32, n)), arr


class SomaticPileupStream:
    """sk_somatic_pileup_stream_*: the normal and the tumor sample's pileups over a region, pushed window by window together and
    chained into a12+a13.  `library` as for PileupStream."""

    def __init__(self, opt, somatic_opt=None, with_read_pos=False, library=None):
        self.L = library or lib()
        L = self.L
        L.sk_somatic_pileup_stream_create.restype = c_void_p
        L.sk_somatic_pileup_stream_create.argtypes = [C.POINTER(PileupOptions), c_void_p, C.c_int]
        L.sk_somatic_pileup_stream_destroy.argtypes = [c_void_p]
        L.sk_somatic_pileup_stream_begin_region.argtypes = [c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.sk_somatic_pileup_stream_push.argtypes = [c_void_p, C.POINTER(ReadBatchStruct), C.POINTER(ReadBatchStruct), C.c_int32, C.c_int32,
                                                    C.c_int32, c_void_p, C.c_int32, C.c_int32, C.c_int32, c_void_p, C.c_int,
                                                    C.POINTER(SomaticPileupWindow)]
        L.sk_last_error.restype = C.c_char_p
        self.genotype = somatic_opt is not None
        self.with_read_pos = with_read_pos
        self.h = L.sk_somatic_pileup_stream_create(C.byref(opt), C.byref(somatic_opt) if self.genotype else None, 1 if with_read_pos else 0)
        if not self.h:
            raise RuntimeError(L.sk_last_error().decode())

    def close(self):
        if self.h:
            self.L.sk_somatic_pileup_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.sk_last_error().decode())

    def begin_region(self, ref_seq, ref_offset, report_begin, report_end, span=49):
        ref = ref_seq.encode()
        self._check(self.L.sk_somatic_pileup_stream_begin_region(self.h, ref, ref_offset, len(ref), report_begin, report_end, span))

    def push(self, normal_rb, tumor_rb, final_to, mask=None, mask_begin=0, forced=None, forced_begin=0, is_compute_nonsomatic=False,
             span=49):
        """-> dict(normal=..., tumor=... (as PileupStream.push), clean2_count per sample, read_pos, genotype)"""
        st = [ReadBatchStruct(rb.n_reads, _p(rb.read_off), _p(rb.read_code), _p(rb.read_qual), _p(rb.path_off), _p(rb.path),
                              _p(rb.pos), _p(rb.is_fwd), _p(rb.mapq), _p(rb.map_level), None, 0, 0, None) for rb in (normal_rb, tumor_rb)]
        w = SomaticPileupWindow()
        mask = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        forced = None if forced is None else np.ascontiguousarray(forced, np.uint8)
        self._check(self.L.sk_somatic_pileup_stream_push(self.h, C.byref(st[0]), C.byref(st[1]), span, mask_begin,
                                                          0 if mask is None else len(mask), _p(mask), min(final_to, 2**31 - 1), forced_begin,
                                                          0 if forced is None else len(forced), _p(forced), 1 if is_compute_nonsomatic else 0,
                                                          C.byref(w)))
        normal, arr = _window_arrays(w.normal)
        tumor, _ = _window_arrays(w.tumor)
        n = w.normal.end - w.normal.begin
        normal["clean2_count"] = arr(w.normal_clean_tier2_count, np.uint32, n)
        tumor["clean2_count"] = arr(w.tumor_clean_tier2_count, np.uint32, n)
        geno = None
        if self.genotype:
            if n and w.genotype:
                buf = (C.c_char * (SOMATIC_GENOTYPE_DTYPE.itemsize * n)).from_address(w.genotype)
                geno = np.frombuffer(buf, SOMATIC_GENOTYPE_DTYPE, n).copy()
            else:
                geno = np.zeros(0, SOMATIC_GENOTYPE_DTYPE)
        read_pos = arr(w.tumor_tier1_read_pos, np.uint32, len(tumor["tier1_calls"])) if self.with_read_pos else None
        return dict(begin=w.normal.begin, end=w.normal.end, normal=normal, tumor=tumor, read_pos=read_pos, genotype=geno)


# ---------------------------------------------------------------------------------------------------- GlobalAligner

def align_scores(**kw):
    s = AlignScores()
    lib().sk_align_scores_default(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


def global_align(pairs, scores=None):
    """pairs: [(query, ref)] strings -> [(score, begin_pos, cigar)]"""
    scores = scores or align_scores()
    n = len(pairs)
    qo = np.zeros(n + 1, np.int64)
    ro = np.zeros(n + 1, np.int64)
    for i, (q, r) in enumerate(pairs):
        qo[i + 1] = qo[i] + len(q)
        ro[i + 1] = ro[i] + len(r)
    qb = np.frombuffer("".join(q for q, _ in pairs).encode(), np.uint8).copy() if n else np.zeros(1, np.uint8)
    rb = np.frombuffer("".join(r for _, r in pairs).encode(), np.uint8).copy() if n else np.zeros(1, np.uint8)
    b = GlobalAlignBatch(n, _p(qo), _p(qb), _p(ro), _p(rb))
    score = np.zeros(max(n, 1), np.int32)
    beg = np.zeros(max(n, 1), np.int32)
    nseg = np.zeros(max(n, 1), np.int32)
    path = np.zeros((int(qo[-1] + ro[-1]) + 4 * n + 1, 2), np.uint32)
    _check(lib().sk_global_align(C.byref(b), C.byref(scores), _p(score), _p(beg), _p(path), _p(nseg)))
    out = []
    for i in range(n):
        po = int(qo[i] + ro[i]) + 4 * i
        out.append((int(score[i]), int(beg[i]), "".join("%d%s" % (path[po + k, 1], CIGAR_CHARS[path[po + k, 0]]) for k in range(nseg[i]))))
    return out


class DiscoveredAllele(C.Structure):
    _fields_ = [("pos", C.c_int32), ("type", C.c_int32), ("del_len", C.c_uint32), ("ins_len", C.c_uint32), ("ins_off", C.c_int32)]


def discover_indels_and_mismatches(ref_seq, ref_offset, ar_begin, ar_end, prev_ar_end, max_indel_size, haplotype, begin_pos,
                                   cigar):
    """-> ([(pos, type, del_len, ins_seq)], n_indels); `cigar` is the '='/'X' CIGAR sk_global_align returned"""
    path = cigar_to_path(cigar)
    segs = (PathSeg * max(len(path), 1))(*[PathSeg(t, l) for t, l in path])
    cap = len(haplotype) + len(path) + 1
    out = (DiscoveredAllele * cap)()
    ins = C.create_string_buffer(2 * len(haplotype) + 16)
    n, ni = C.c_int32(), C.c_int32()
    f = lib().sk_discover_indels_and_mismatches
    f.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_char_p, C.c_int32, C.c_int32,
                  C.POINTER(PathSeg), C.c_int32, C.POINTER(DiscoveredAllele), C.c_int32, C.c_char_p, C.c_int32,
                  C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    r, h = ref_seq.encode(), haplotype.encode()
    _check(f(r, ref_offset, len(r), ar_begin, ar_end, prev_ar_end, max_indel_size, h, len(h), begin_pos, segs, len(path), out, cap,
             ins, len(ins), C.byref(n), C.byref(ni)))
    raw = ins.raw
    return [(out[i].pos, out[i].type, out[i].del_len, raw[out[i].ins_off:out[i].ins_off + out[i].ins_len].decode())
            for i in range(n.value)], ni.value


# ---- the feed: BGZF inflation and BAM record decoding ------------------------------------------------------------

BAM_RECORD_DTYPE = np.dtype([("ref_id", "<i4"), ("pos", "<i4"), ("mate_ref_id", "<i4"), ("mate_pos", "<i4"), ("template_size", "<i4"),
                             ("l_seq", "<i4"), ("n_cigar", "<i4"), ("flag", "<u2"), ("mapq", "u1"), ("is_fwd_strand", "u1"), ("pad", "<u4")])
PATH_SEG_DTYPE = np.dtype([("type", "<u4"), ("length", "<u4")])


def bgzf_scan(data):
    """data: uint8 array of whole BGZF blocks -> (block_off[n+1], out_off[n+1])"""
    data = np.ascontiguousarray(data, np.uint8)
    n = lib().sk_bgzf_scan(_p(data), len(data), None, None, 0)
    if n < 0:
        raise StrelkaAmdError("sk_bgzf_scan: malformed BGZF block header")
    block_off, out_off = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
    if lib().sk_bgzf_scan(_p(data), len(data), _p(block_off), _p(out_off), n) != n:
        raise StrelkaAmdError("sk_bgzf_scan failed")
    return block_off, out_off


def bgzf_inflate(data):
    """the inflated stream of a BGZF file image (kernel-backed; CRC-32 and ISIZE of every block checked)"""
    data = np.ascontiguousarray(data, np.uint8)
    block_off, out_off = bgzf_scan(data)
    out = np.zeros(int(out_off[-1]), np.uint8)
    _check(lib().sk_bgzf_inflate(_p(data), _p(block_off), _p(out_off), len(block_off) - 1, _p(out)))
    return out


def bam_scan_records(stream, first=None):
    stream = np.ascontiguousarray(stream, np.uint8)
    if first is None:
        first = lib().sk_bam_header_end(_p(stream), len(stream))
        if first < 0:
            raise StrelkaAmdError("sk_bam_header_end: not a BAM stream")
    n = lib().sk_bam_scan_records(_p(stream), len(stream), first, None, None, None, 0)
    if n < 0:
        raise StrelkaAmdError("sk_bam_scan_records: malformed record")
    rec_off, read_off, path_off = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
    if lib().sk_bam_scan_records(_p(stream), len(stream), first, _p(rec_off), _p(read_off), _p(path_off), n) != n:
        raise StrelkaAmdError("sk_bam_scan_records failed")
    return rec_off[:n].copy(), read_off, path_off


def bam_decode(stream, first=None):
    """-> dict(rec[BAM_RECORD_DTYPE], read_off, read_code, read_qual, path_off, path[PATH_SEG_DTYPE]) of every whole record"""
    stream = np.ascontiguousarray(stream, np.uint8)
    rec_off, read_off, path_off = bam_scan_records(stream, first)
    n = len(rec_off)
    rec = np.zeros(n, BAM_RECORD_DTYPE)
    code = np.zeros(max(int(read_off[-1]), 1), np.uint8)
    qual = np.zeros(max(int(read_off[-1]), 1), np.uint8)
    path = np.zeros(max(int(path_off[-1]), 1), PATH_SEG_DTYPE)
    _check(lib().sk_bam_decode(_p(stream), len(stream), _p(rec_off), n, _p(read_off), _p(path_off), _p(rec), _p(code), _p(qual), _p(path)))
    return dict(rec=rec, rec_off=rec_off, read_off=read_off, read_code=code[:int(read_off[-1])], read_qual=qual[:int(read_off[-1])],
                path_off=path_off, path=path[:int(path_off[-1])])


def normalize_alignments(ref_seq, ref_offset, reads, library=None):
    """reads: dicts(code uint8[], pos, path [(type, length)]) -> list of (changed, pos, path) through sk_normalize_alignments
    (`library`: another build of the ABI, e.g. the CPU double, for the no-GPU test tier)"""
    L = library or lib()
    n = len(reads)
    read_off = np.zeros(n + 1, np.int64)
    path_off = np.zeros(n + 1, np.int64)
    for i, r in enumerate(reads):
        read_off[i + 1] = read_off[i] + len(r["code"])
        path_off[i + 1] = path_off[i] + len(r["path"])
    code = np.concatenate([np.asarray(r["code"], np.uint8) for r in reads]) if n else np.zeros(1, np.uint8)
    path = np.zeros(max(int(path_off[-1]), 1), PATH_SEG_DTYPE)
    k = 0
    for r in reads:
        for t, l in r["path"]:
            path[k] = (t, l)
            k += 1
    n_seg = np.array([len(r["path"]) for r in reads], np.int32)
    pos = np.array([r["pos"] for r in reads], np.int32)
    changed = np.zeros(max(n, 1), np.uint8)
    ref_b = ref_seq.encode() if isinstance(ref_seq, str) else bytes(ref_seq)
    rc = L.sk_normalize_alignments(ref_b, int(ref_offset), len(ref_b), n, _p(read_off), _p(code), _p(path_off), _p(n_seg), _p(path), _p(pos), _p(changed))
    if rc != 0:
        raise StrelkaAmdError(L.sk_last_error().decode("utf-8", "replace"))
    out = []
    for i in range(n):
        seg = path[int(path_off[i]):int(path_off[i]) + int(n_seg[i])]
        out.append((int(changed[i]), int(pos[i]), [(int(t), int(l)) for t, l in seg]))
    return out


BAI_CHUNK_DTYPE = np.dtype([("begin", "<u8"), ("end", "<u8")])


def bai_query(bai, ref_id, begin, end):
    """hts_itr_query over a .bai image -> the chunks (BGZF virtual offsets) that can hold records overlapping [begin, end)"""
    bai = np.ascontiguousarray(bai, np.uint8)
    L = lib()
    L.sk_bai_query.restype = C.c_int32
    L.sk_bai_query.argtypes = [c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_void_p, C.c_int32]
    n = L.sk_bai_query(_p(bai), len(bai), ref_id, begin, end, None, 0)
    if n < 0:
        raise StrelkaAmdError("sk_bai_query: %s" % ("malformed index" if n == -1 else "reference not in the index"))
    out = np.zeros(max(n, 1), BAI_CHUNK_DTYPE)
    if n:
        assert L.sk_bai_query(_p(bai), len(bai), ref_id, begin, end, _p(out), n) == n
    return out[:n]


def bam_region_filter(rec, path_off, path, ref_id, begin, end):
    """hts_itr_next's record test -> (keep[n] bool, how many records the iterator reads before it finishes)"""
    rec = np.ascontiguousarray(rec, BAM_RECORD_DTYPE)
    path_off = np.ascontiguousarray(path_off, np.int64)
    path = np.ascontiguousarray(path, PATH_SEG_DTYPE)
    keep = np.zeros(max(len(rec), 1), np.uint8)
    L = lib()
    L.sk_bam_region_filter.restype = C.c_int32
    L.sk_bam_region_filter.argtypes = [c_void_p, c_void_p, c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_void_p]
    n = L.sk_bam_region_filter(_p(rec), _p(path_off), _p(path), len(rec), ref_id, begin, end, _p(keep))
    if n < 0:
        raise StrelkaAmdError("sk_bam_region_filter: bad argument")
    return keep[:len(rec)].astype(bool), n


def _inflate_blocks(data, block_off, out_off):
    out = np.zeros(max(int(out_off[-1]), 1), np.uint8)
    _check(lib().sk_bgzf_inflate(_p(np.ascontiguousarray(data, np.uint8)), _p(np.ascontiguousarray(block_off, np.int64)),
                                 _p(np.ascontiguousarray(out_off, np.int64)), len(block_off) - 1, _p(out)))
    return out[:int(out_off[-1])]


def bam_fetch_region(bam, bai, ref_id, begin, end, inflate=None, decode=None, blocks=None):
    """What sam_itr_queryi + sam_itr_next hand the reference's bam_streamer for a region, from file images: the index gives the
    chunks (sk_bai_query), their BGZF blocks are inflated and their records decoded (the kernels behind sk_bgzf_inflate /
    sk_bam_decode, unless `inflate(data, block_off, out_off) -> bytes` / `decode(stream, first) -> dict` are given), the iterator's
    record test picks the result (sk_bam_region_filter).
    -> dict(rec, stream_offset (of each record in the inflated file), read_code, read_qual, path: one entry per record)."""
    bam = np.ascontiguousarray(bam, np.uint8)
    block_off, out_off = blocks if blocks is not None else bgzf_scan(bam)
    n_blocks = len(block_off) - 1
    starts = {int(o): i for i, o in enumerate(block_off[:-1])}
    inflate = inflate or _inflate_blocks
    decode = decode or bam_decode
    parts, offsets = [], []
    for ch in bai_query(bai, ref_id, begin, end):
        cb, ub, ce, ue = int(ch["begin"]) >> 16, int(ch["begin"]) & 0xffff, int(ch["end"]) >> 16, int(ch["end"]) & 0xffff
        i0 = starts[cb]
        i1 = starts.get(ce, n_blocks)  # (a chunk may end where the file ends)
        limit = int(out_off[i1] - out_off[i0]) + ue  # the chunk's records start before this offset of its inflated bytes
        last = min(i1, n_blocks - 1)
        while True:  # a record that starts inside the chunk may end in a later block
            stream = inflate(bam[int(block_off[i0]):int(block_off[last + 1])], block_off[i0:last + 2] - block_off[i0],
                             out_off[i0:last + 2] - out_off[i0])
            d = decode(stream, ub)
            if len(d["rec_off"]):
                at = int(d["rec_off"][-1])
                nxt = at + 4 + int(np.frombuffer(np.asarray(stream[at:at + 4], np.uint8).tobytes(), "<i4")[0])
            else:
                nxt = ub
            if nxt >= limit or last == n_blocks - 1:
                break
            last += 1
        n_in = int(np.searchsorted(d["rec_off"], limit))
        keep, n_read = bam_region_filter(d["rec"][:n_in], d["path_off"][:n_in + 1], d["path"], ref_id, begin, end)
        for i in np.nonzero(keep)[0]:
            parts.append((d, int(i)))
            offsets.append(int(out_off[i0]) + int(d["rec_off"][i]))
        if n_read < n_in:
            break  # the iterator has seen a record past the region: finished
    rec = np.array([d["rec"][i] for d, i in parts], BAM_RECORD_DTYPE) if parts else np.zeros(0, BAM_RECORD_DTYPE)
    return dict(rec=rec, stream_offset=np.array(offsets, np.int64),
                read_code=[d["read_code"][int(d["read_off"][i]):int(d["read_off"][i + 1])] for d, i in parts],
                read_qual=[d["read_qual"][int(d["read_off"][i]):int(d["read_off"][i + 1])] for d, i in parts],
                path=[d["path"][int(d["path_off"][i]):int(d["path_off"][i + 1])] for d, i in parts])


GVCF_SITE_DTYPE = np.dtype([("pos", "<i4"), ("is_compressible", "u1"), ("is_gqx", "u1"), ("ploidy", "u1"), ("flush_before", "u1"), ("gt", "<u4"),
                            ("locus_filters", "<u4"), ("sample_filters", "<u4"), ("gqx", "<i4"), ("used_basecalls", "<u4"), ("unused_basecalls", "<u4")])
GVCF_BLOCK_DTYPE = np.dtype([("pos", "<i4"), ("count", "<i4"), ("is_gqx_defined", "<i4"), ("gqx_min", "<i4"), ("dpu_min", "<i4"), ("pad", "<i4"),
                             ("dpu_mean", "<f8"), ("dpf_mean", "<f8")])


def gvcf_block_sites(sites, block_percent_tol=30, block_abs_tol=3, library=None):
    """the gVCF writer's non-variant block logic for one sample's run of sites -> (kind[n], blocks[n]: valid where kind == 1)"""
    sites = np.ascontiguousarray(sites, GVCF_SITE_DTYPE)
    kind = np.zeros(max(len(sites), 1), np.uint8)
    blocks = np.zeros(max(len(sites), 1), GVCF_BLOCK_DTYPE)
    L = library or lib()
    L.sk_gvcf_block_sites.argtypes = [c_void_p, C.c_int32, C.c_uint32, C.c_uint32, c_void_p, c_void_p]
    if L.sk_gvcf_block_sites(_p(sites), len(sites), block_percent_tol, block_abs_tol, _p(kind), _p(blocks)):
        raise StrelkaAmdError(L.sk_last_error().decode() if library else last_error())
    return kind[:len(sites)], blocks[:len(sites)]
