/*
 * strelka_oracle.h -- CPU restatement of the Strelka2 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (strelka_amd/) never links, imports or calls it.
 *
 * Each function restates one reference function (file:line cited at the definition in strelka_oracle.c;
 * `L/` = /root/reference/src/c++/lib/) in plain C with the reference's own data model (CIGAR path segments + indel
 * keys, pileup `base_call`s), NOT the flattened layout of the product's C-ABI -- so a parity test compares
 * "reference algorithm on reference-shaped input" against "host flattener + HIP kernel".
 *
 * Pinning: oracle/ref/ builds the reference's own translation units (oracle/_ref/libstrelka_ref.so) and
 * tests/test_oracle_vs_ref.py + tests/golden/ check this restatement against them.
 */
#ifndef STRELKA_ORACLE_H
#define STRELKA_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- q-score tables (L/blt_util/qscore_cache.cpp:34-50) ---- */
void sko_get_qscore_tables(double* q2p, double* q2lncompe, double* q2lne); /* 71 each */
int sko_error_prob_to_qphred(double p);         /* L/blt_util/qscore.hh:60-66 (double) */
int sko_ln_error_prob_to_qphred_f(float lnp);   /* L/blt_util/qscore.hh:68-74 (float instantiation) */
double sko_log1p_switch(double x);              /* L/blt_util/math_util.hh:33-48 */
double sko_log_sum2(double a, double b);        /* L/blt_util/logSumUtil.hh:33-41 */
float sko_log_sum2f(float a, float b);

/* ---- libstdc++ std::sort restated (introsort + final insertion sort), descending-q comparator of
 *      L/blt_common/adjust_joint_eprob.cpp:41-53.  idx[n] is permuted in place; key[] is indexed by idx values. */
void sko_sort_idx_by_key_desc(uint32_t* idx, int n, const uint16_t* key);

/* ---- hot path A: scoreCandidateAlignment (L/starling_common/starling_read_align_score.cpp:261-499) ---- */
enum { SKO_NONE = 0, SKO_MATCH, SKO_INSERT, SKO_DELETE, SKO_SKIP, SKO_SOFT_CLIP, SKO_HARD_CLIP, SKO_PAD, SKO_SEQ_MATCH,
       SKO_SEQ_MISMATCH }; /* ALIGNPATH::align_t, L/blt_util/align_path.hh:36-48 */
enum { SKO_INDEL_NONE = 0, SKO_INDEL_INDEL, SKO_INDEL_MISMATCH, SKO_INDEL_BP_LEFT, SKO_INDEL_BP_RIGHT };

typedef struct sko_path_seg {
    uint32_t type;
    uint32_t length;
} sko_path_seg;

typedef struct sko_indel { /* IndelKey (L/starling_common/IndelKey.hh:39-199) + its candidate status */
    int32_t pos;
    int32_t type;
    uint32_t del_len;
    uint32_t ins_len;
    const char* ins_seq; /* insert sequence (for breakpoints: the breakpoint insert sequence), ACGTN chars */
    int32_t is_candidate;
} sko_indel;

typedef struct sko_cal { /* CandidateAlignment (L/starling_common/CandidateAlignment.hh:35-78) */
    int32_t pos;         /* al.pos */
    int32_t n_seg;
    const sko_path_seg* path;
    int32_t n_indels;
    const sko_indel* indels; /* IndelKey-sorted indel set of the alignment */
    sko_indel leading;       /* type SKO_INDEL_NONE when absent */
    sko_indel trailing;
} sko_cal;

/* read_code: BAM 4-bit code per byte; ref_seq: reference contig segment chars starting at genomic ref_offset */
double sko_score_candidate_alignment(const uint8_t* read_code, const uint8_t* read_qual, int32_t read_len,
                                     const sko_cal* cal, const char* ref_seq, int32_t ref_offset, int32_t ref_len);

/* ---- hot path B, germline ---- */
typedef struct sko_germline_options {
    double bsnp_diploid_theta, bsnp_ssd_no_mismatch, bsnp_ssd_one_mismatch;
    int32_t is_min_vexp;
    double min_vexp;
} sko_germline_options;

/* adjust_joint_eprob (L/blt_common/adjust_joint_eprob.cpp:201-243); calls = packed base_call (include/strelka_amd.h) */
void sko_adjust_joint_eprob(const uint16_t* calls, int32_t n_calls, const sko_germline_options* opt, float* de);

/* get_diploid_gt_lhood (L/blt_common/position_snp_call_pprob_digt.cpp:328-385), no het-frequency extension */
void sko_diploid_gt_lhood(const uint16_t* calls, const float* de, int32_t n_calls, uint32_t ref_gt,
                          int is_strand_specific, int is_ss_fwd, float* lhood);

typedef struct sko_digt_result_set {
    double ref_pprob;
    uint32_t max_gt;
    int32_t snp_qphred, max_gt_qphred, _pad;
} sko_digt_result_set;
typedef struct sko_digt_call {
    float lhood[10];
    uint32_t phredLoghood[10];
    sko_digt_result_set genome, poly;
    double strand_bias;
    uint32_t ref_gt, is_called;
} sko_digt_call;

/* position_snp_call_pprob_digt(..., is_always_test=true) (L/blt_common/position_snp_call_pprob_digt.cpp:473-539) */
void sko_position_snp_call_pprob_digt(const uint16_t* calls, const float* de, int32_t n_calls, uint32_t ref_base_id,
                                      int ploidy, const sko_germline_options* opt, sko_digt_call* out);

/* ---- hot path B, somatic SNV ---- */
typedef struct sko_somatic_snv_options {
    double bsnp_diploid_theta, somatic_snv_rate, shared_site_error_rate, shared_site_error_strand_bias_fraction,
        ssnv_contam_tolerance;
} sko_somatic_snv_options;
typedef struct sko_somatic_snv_call {
    float normal_lhood[30], tumor_lhood[30];
    uint32_t max_gt;
    int32_t qphred, from_ntype_qphred;
    uint32_t ntype;
    float strand_bias;
    uint32_t is_called, normal_alt_id, tumor_alt_id;
} sko_somatic_snv_call;

/* 21 prestrand states (+9 strand states when with_strand) of one sample
 * (L/applications/strelka/position_somatic_snv_strand_grid_lhood_cached.cpp:41-234, position_somatic_snv_strand_grid.cpp:63-83) */
void sko_somatic_sample_lhood(const uint16_t* calls, int32_t n_calls, uint32_t ref_gt, int with_strand, float* lhood);

/* calculate_result_set_grid (L/applications/strelka/qscore_calculator.cpp:47-209) */
void sko_calculate_result_set_grid(float contam_tolerance, float ln_sse_rate, float ln_csse_rate,
                                   const float* normal_lhood, const float* tumor_lhood, const float* lnprior3,
                                   float lnmatch, float lnmismatch, uint32_t* max_gt, int32_t* qphred,
                                   int32_t* from_ntype_qphred, uint32_t* ntype);

/* position_somatic_snv_call, tier1 only, isComputeNonSomatic=false
 * (L/applications/strelka/position_somatic_snv_strand_grid.cpp:230-363) */
void sko_position_somatic_snv_call(const uint16_t* ncalls, int32_t n_n, const uint16_t* tcalls, int32_t n_t,
                                   uint32_t ref_base_id, const sko_somatic_snv_options* opt, int is_forced_output,
                                   sko_somatic_snv_call* out);

/* the whole of position_somatic_snv_call (both tiers, per-locus forced output, non-somatic quality):
 * somatic_snv_genotype_grid (L/applications/strelka/somatic_result_set.hh:56-79); same layout as sk_somatic_snv_genotype */
typedef struct sko_somatic_snv_genotype {
    uint32_t ref_gt;
    uint8_t snv_tier, snv_from_ntype_tier, is_forced_output, is_computed;
    uint32_t ntype; /* NTYPE::REF/HOM/HET/CONFLICT = 0..3 */
    uint32_t max_gt;
    int32_t qphred, from_ntype_qphred, nonsomatic_qphred;
    uint32_t normal_alt_id, tumor_alt_id;
    int32_t _pad;
    double strand_bias;
} sko_somatic_snv_genotype;
/* n1/t1: CleanPileupFilter(pi,false) of the normal / tumor sample; n2/t2: CleanPileupFilter(pi,true) (used when is_tier2) */
void sko_position_somatic_snv_call_tiers(const uint16_t* n1, int32_t n_n1, const uint16_t* t1, int32_t n_t1,
                                         const uint16_t* n2, int32_t n_n2, const uint16_t* t2, int32_t n_t2, int is_tier2,
                                         uint32_t ref_base_id, const sko_somatic_snv_options* opt, int is_forced_output,
                                         int is_compute_nonsomatic, sko_somatic_snv_genotype* sgt);
int sko_nonsomatic_qphred(const float* normal_lhood21, const float* tumor_lhood21);

/* ---- hot path B, indels ---- */
/* get_het_observed_allele_ratio (L/starling_common/starling_indel_call_pprob_digt.cpp:40-71); outputs untouched when
 * the total path term is 0 */
void sko_het_observed_allele_ratio(unsigned read_length, unsigned min_overlap, unsigned del_len, unsigned ins_len,
                                   double het_allele_ratio, double* log_ref_prob, double* log_indel_prob);
/* integrateOutMappingStatus (L/starling_common/readMappingAdjustmentUtil.hh:29-56) with
 * correctMappingLogPrior = log(1.7e-10) (L/starling_common/starling_base_shared.cpp:64) */
double sko_integrate_out_mapping_status(double randomBaseMatchLogProb, unsigned nonAmbiguousBasesInRead, double lnp);

/* 21 somatic-grid states of one sample for one indel: get_indel_digt_lhood (:240-336) for REF/HOM/HET and
 * get_indel_het_grid_lhood (L/applications/strelka/somatic_indel_grid.cpp:66-89) -> get_high_low_het_ratio_lhood
 * (:75-155) for the 18 grid ratios.  alt_lnp[r] = best alternate-indel score of read r, NaN when it has none. */
void sko_indel_grid_lhood(int32_t n_reads, const float* ref_lnp, const float* indel_lnp, const float* alt_lnp,
                          const uint16_t* non_ambig, const uint16_t* read_length, const uint8_t* is_tier1,
                          unsigned del_len, unsigned ins_len, int is_breakpoint, int min_read_bp_flank,
                          double randomBaseMatchProb, int is_include_tier2, int is_use_alt_indel, double* lhood21);

/* getVariantAlleleGroupGenotypeLhoodsForSample (L/starling_common/AlleleGroupGenotype.cpp:185-258), empty contrast
 * group.  ref_lnp/allele_lnp: [n_reads][n_alt], allele_lnp NaN = read not scored for that allele.
 * out_lhood: ploidy 1 -> n_alt+1, ploidy 2 -> (n_alt+1)(n_alt+2)/2 (VcfGenotypeUtil order).
 * out_counts: [2 strands: fwd, rev][n_alt+2]: confident count per allele (ref first), then non-confident. */
void sko_allele_group_genotype_lhoods(int32_t n_reads, int32_t n_alt, const float* ref_lnp, const float* allele_lnp,
                                      const uint16_t* non_ambig, const uint16_t* read_length, const uint8_t* is_tier1,
                                      const uint8_t* is_fwd, const uint32_t* del_len, const uint32_t* ins_len,
                                      int ploidy, int min_read_bp_flank, double randomBaseMatchProb,
                                      double readSupportThreshold, double* out_lhood, uint32_t* out_counts);

/* somatic indel call from the two samples' 21-state likelihoods (get_somatic_indel :243-291, one tier, no multi-indel
 * filter): float cast, shared error rate = indelToRefErrorProb^shared_indel_error_factor, calculate_result_set_grid */
void sko_somatic_indel_result(const double* normal_lhood21, const double* tumor_lhood21, double indelToRefErrorProb,
                              double shared_indel_error_factor, double indel_contam_tolerance, double somatic_indel_rate,
                              double bindel_diploid_theta, uint32_t* max_gt, int32_t* qphred, int32_t* from_ntype_qphred,
                              uint32_t* ntype);

/* ---- the whole of somatic_indel_caller_grid::get_somatic_indel (L/applications/strelka/somatic_indel_grid.cpp:181-361):
 * multi-indel-allele filter (:102-177, get_sum_path_pprob L/starling_common/starling_indel_call_pprob_digt.cpp:187-236,
 * indel_lnp_to_pprob L/starling_common/AlleleReportInfoUtil.cpp:220-301), both tiers, tier combination ---- */
typedef struct sko_indel_sample_reads { /* IndelSampleData::read_path_lnp in read-id order */
    int32_t n_reads;
    const float* ref_lnp;
    const float* indel_lnp;
    const int32_t* alt_key; /* [n_reads][2] index into the indel's alt-key table, -1 = no entry (entries are front-packed) */
    const float* alt_lnp;   /* [n_reads][2] */
    const uint16_t* non_ambig;
    const uint16_t* read_length;
    const uint8_t* is_tier1;
} sko_indel_sample_reads;
typedef struct sko_alt_key { /* what is_indel_conflict (L/starling_common/indel_util.cpp:27-45) reads of an IndelKey */
    int32_t begin_pos, end_pos; /* pos, right_pos() */
    int32_t is_mismatch;
} sko_alt_key;
typedef struct sko_somatic_indel_params {
    int32_t normal_min_read_bp_flank, tumor_min_read_bp_flank;
    double random_base_match_prob, tier2_random_base_match_prob;
    int32_t use_tier2_evidence, is_use_alt_indel;
    double bindel_diploid_theta, somatic_indel_rate, shared_indel_error_factor, indel_contam_tolerance;
} sko_somatic_indel_params;
typedef struct sko_somatic_indel_genotype { /* somatic_indel_call, somatic_result_set.hh:81-102; same layout as sk_somatic_indel_genotype */
    uint8_t sindel_tier, sindel_from_ntype_tier, is_forced_output, is_overlap;
    uint32_t ntype; /* NTYPE */
    uint32_t max_gt;
    int32_t qphred, from_ntype_qphred;
} sko_somatic_indel_genotype;
void sko_get_somatic_indel(const sko_indel_sample_reads* normal, const sko_indel_sample_reads* tumor,
                           const sko_alt_key* alt_keys, int32_t n_alt_keys, unsigned del_len, unsigned ins_len,
                           int is_breakpoint, const sko_somatic_indel_params* p, double indel_to_ref_error_prob,
                           int is_forced_output, sko_somatic_indel_genotype* out);
/* is_multi_indel_allele alone: returns 1 when the indel is filtered; *is_overlap as the reference sets it */
int sko_is_multi_indel_allele(const sko_indel_sample_reads* normal, const sko_indel_sample_reads* tumor,
                              const sko_alt_key* alt_keys, const sko_somatic_indel_params* p, int is_include_tier2,
                              int* is_overlap);

/* ---- batch drivers (plain loops over the functions above; used for parity tests and the bench CPU baseline) ---- */
typedef struct sko_read_case {
    const uint8_t* read_code;
    const uint8_t* read_qual;
    int32_t read_len;
    const char* ref_seq;
    int32_t ref_offset, ref_len;
    const sko_cal* cals;
    int32_t n_cals;
} sko_read_case;
/* out: concatenated scores, case by case */
void sko_score_cases(const sko_read_case* cases, int32_t n_cases, double* out);

void sko_adjust_joint_eprob_batch(const int64_t* call_off, const uint16_t* calls, int32_t n_loci,
                                  const sko_germline_options* opt, float* de);
void sko_site_digt_call_batch(const int64_t* call_off, const uint16_t* calls, const float* de, const uint8_t* ref_base,
                              const uint8_t* ploidy /* may be NULL */, int32_t n_loci, const sko_germline_options* opt,
                              sko_digt_call* out);
void sko_somatic_snv_call_batch(const int64_t* n_off, const uint16_t* n_calls, const int64_t* t_off,
                                const uint16_t* t_calls, const uint8_t* ref_base, int32_t n_loci,
                                const sko_somatic_snv_options* opt, int is_forced_output, sko_somatic_snv_call* out);
void sko_germline_lnpriors(double theta, float* out /* [2][5][2][10] */);

/* ---- row a8: pileup of aligned reads (restatement of starling_pos_processor_base::pileup_read_segment,
 * L/starling_common/starling_pos_processor_base.cpp:1127-1421, run read after read as pileup_pos_reads does) ---- */
int sko_mapped_qscore(int basecall_q, int mapq); /* qphred_cache::get_mapped_qscore, L/blt_util/qscore_cache.hh:123-134 */
typedef struct sko_pileup_options {
    int32_t min_basecall_qscore, mismatch_density_flank_size, mismatch_density_max_count, use_tier2_evidence,
        tier2_mismatch_density_max_count, is_mapq_adjust, min_distance_from_read_edge,
        largest_total_indel_ref_span_per_read, report_begin, report_end;
} sko_pileup_options;
typedef struct sko_read_batch {
    int32_t n_reads;
    const int64_t* read_off;
    const uint8_t* read_code;
    const uint8_t* read_qual;
    const int64_t* path_off;
    const sko_path_seg* path;
    const int32_t* pos;
    const uint8_t* is_fwd;
    const uint8_t* mapq;
    const uint8_t* map_level; /* MAPLEVEL::index_t */
    const char* ref_seq;
    int32_t ref_offset, ref_len;
    const uint8_t* cand_snv_mask;
} sko_read_batch;
/* mode: 0 raw tier1 calls, 1 raw tier2 calls, 2 CleanPileupFilter(pi,false), 3 CleanPileupFilter(pi,true)
 * (L/starling_common/PileupCleaner.cpp:28-66).  call_off[n_loci+1], calls[capacity]; returns the number of calls or -1
 * when capacity is too small / a read is malformed. */
int64_t sko_pileup_reads(const sko_read_batch* b, const sko_pileup_options* opt, int mode, int64_t* call_off,
                         uint16_t* calls, int64_t capacity, uint32_t* spandel_count, uint32_t* submapped_count);
/* + MapqTracker per position (insert_mapq_count :1346; any of the three may be NULL) */
int64_t sko_pileup_reads_mapq(const sko_read_batch* b, const sko_pileup_options* opt, int mode, int64_t* call_off,
                              uint16_t* calls, int64_t capacity, uint32_t* spandel_count, uint32_t* submapped_count,
                              uint32_t* mapq_count, uint32_t* mapq_zero_count, uint64_t* mapq_sum_square);
/* the raw tier1 column and, parallel to it, read_pos | read_size << 16 of every call (updateSomaticScoringMetrics' arguments,
 * L/starling_common/starling_pos_processor_base.cpp:984-1000, :1360) */
int64_t sko_pileup_reads_readpos(const sko_read_batch* b, const sko_pileup_options* opt, int64_t* call_off, uint16_t* calls,
                                 int64_t capacity, uint32_t* read_pos);
/* one word per live match position of every read (submapped reads included), in pileup order: what updateGermlineScoringMetrics
 * accumulates (L/starling_common/starling_pos_processor_base.cpp:1346-1357, pos_basecall_buffer.cpp:43-70):
 * base id | mapq << 3 | qscore << 11 | cycle << 18 | min(20, distance from read edge) << 29 | is_submapped << 34 */
int64_t sko_pileup_reads_evs(const sko_read_batch* b, const sko_pileup_options* opt, int64_t* evs_off, uint64_t* evs_words, int64_t capacity);

/* ---- GlobalAligner<int>::align (L/alignment/GlobalAlignerImpl.hh:35-228) ---- */
typedef struct sko_align_scores { /* AlignmentScores<int>, L/alignment/AlignmentScores.hh */
    int32_t match, mismatch, open, extend, offEdge, insertDelete, isAllowEdgeInsertion, isRequireEdgeDeletion;
} sko_align_scores;
/* returns the number of path segments written (SEQ_MATCH/SEQ_MISMATCH form) or -1 */
int sko_global_align(const char* query, int query_size, const char* ref, int ref_size, const sko_align_scores* scores,
                     int32_t* out_score, int32_t* out_begin_pos, sko_path_seg* out_path, int path_cap);

#ifdef __cplusplus
}
#endif
#endif
