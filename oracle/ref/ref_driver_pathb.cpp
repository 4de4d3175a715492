// ref_driver_pathb.cpp -- C entry points over the REFERENCE's own translation units (hot path B).
//
// TEST INFRASTRUCTURE ONLY.  This file is compiled together with the unmodified reference sources where they lie
// under /root/reference (see oracle/Makefile, target `ref`) into oracle/_ref/libstrelka_ref.so.  It contains no
// reference code: it only builds the reference's structs (snp_pos_info, extended_pos_info, blt_options ...) from the
// flat batch layout of include/strelka_amd.h and calls the reference functions.

#include "blt_common/adjust_joint_eprob.hh"
#include "blt_common/blt_shared.hh"
#include "blt_common/position_snp_call_pprob_digt.hh"
#include "blt_common/snp_pos_info.hh"
#include "blt_util/logSumUtil.hh"
#include "blt_util/qscore.hh"
#include "applications/strelka/position_somatic_snv_strand_grid_lhood_cached.hh"
#include "applications/strelka/qscore_calculator.hh"
#include "applications/strelka/strelka_digt_states.hh"
#include "strelka_common/het_ratio_cache.hh"

#include <algorithm>
#include <cstdint>
#include <chrono>
#include <cstring>
#include <vector>

namespace
{

struct driver_blt_options : public blt_options
{
    bool is_bsnp_diploid() const override { return true; }
};

void fill_pileup(snp_pos_info& pi, const uint16_t* calls, int n, unsigned ref_base_id)
{
    static const char bases[] = "ACGTN";
    pi.clear();
    pi.set_ref_base(bases[ref_base_id > 4 ? 4 : ref_base_id]);
    for (int i = 0; i < n; ++i) {
        const uint16_t c = calls[i];
        pi.calls.push_back(base_call((c >> 6) & 0xf, c & 0x3f, (c >> 10) & 1, 0, 0, (c >> 12) & 1, (c >> 11) & 1,
                                     (c >> 13) & 1));
    }
}

} // namespace

extern "C" {

int ref_sizeof_base_call() { return (int)sizeof(base_call); }

/// raw bits of a reference base_call built from the same fields: lets the tests check the packed layout
uint16_t ref_pack_base_call(unsigned q, unsigned base, unsigned fwd, unsigned nmm, unsigned filt, unsigned tscf)
{
    base_call bc(base, q, fwd, 0, 0, filt, nmm, tscf);
    uint16_t v;
    std::memcpy(&v, &bc, sizeof(v));
    return v;
}

void ref_get_qscore_tables(double* q2p, double* q2lncompe, double* q2lne)
{
    for (int q = 0; q <= 70; ++q) {
        q2p[q] = qphred_to_error_prob(q);
        q2lncompe[q] = qphred_to_ln_comp_error_prob(q);
        q2lne[q] = qphred_to_ln_error_prob(q);
    }
}

int ref_mapped_qscore(int basecall_q, int mapq) { return qphred_to_mapped_qphred(basecall_q, mapq); }
int ref_error_prob_to_qphred(double p) { return error_prob_to_qphred(p); }
int ref_ln_error_prob_to_qphred_f(float lnp) { return ln_error_prob_to_qphred(lnp); }
double ref_log_sum2(double a, double b) { return getLogSum(a, b); }
float ref_log_sum2f(float a, float b) { return getLogSum(a, b); }

/// std::sort with the reference's comparator (adjust_joint_eprob.cpp:41-53) on an index array
void ref_sort_idx_by_key_desc(uint32_t* idx, int n, const uint16_t* key)
{
    std::sort(idx, idx + n, [key](const uint32_t a, const uint32_t b) { return key[a] > key[b]; });
}

void ref_adjust_joint_eprob(const uint16_t* calls, int n_calls, double ssd_no_mismatch, double ssd_one_mismatch,
                            int is_min_vexp, double min_vexp, float* de)
{
    driver_blt_options opt;
    opt.bsnp_ssd_no_mismatch = ssd_no_mismatch;
    opt.bsnp_ssd_one_mismatch = ssd_one_mismatch;
    opt.is_min_vexp = (is_min_vexp != 0);
    opt.min_vexp = min_vexp;
    dependent_prob_cache dpc;
    snp_pos_info pi;
    fill_pileup(pi, calls, n_calls, 0);
    std::vector<float> dep;
    adjust_joint_eprob(opt, dpc, pi, dep);
    for (int i = 0; i < n_calls; ++i) de[i] = dep[i];
}

void ref_diploid_gt_lhood(const uint16_t* calls, const float* de, int n_calls, unsigned ref_gt, int is_strand_specific,
                          int is_ss_fwd, float* lhood)
{
    driver_blt_options opt;
    snp_pos_info pi;
    fill_pileup(pi, calls, n_calls, ref_gt);
    std::vector<float> dep(de, de + n_calls);
    const extended_pos_info epi(pi, dep);
    pprob_digt_caller::get_diploid_gt_lhood(opt, epi, false, 0, lhood, is_strand_specific != 0, is_ss_fwd != 0);
}

struct ref_digt_result_set
{
    double ref_pprob;
    uint32_t max_gt;
    int32_t snp_qphred, max_gt_qphred, _pad;
};
struct ref_digt_call
{
    float lhood[10];
    uint32_t phredLoghood[10];
    ref_digt_result_set genome, poly;
    double strand_bias;
    uint32_t ref_gt, is_called;
};

void ref_position_snp_call_pprob_digt(const uint16_t* calls, const float* de, int n_calls, unsigned ref_base_id,
                                      int ploidy, double theta, ref_digt_call* out)
{
    static double cached_theta = -1;
    static std::unique_ptr<pprob_digt_caller> caller;
    if (cached_theta != theta) {
        caller.reset(new pprob_digt_caller(theta));
        cached_theta = theta;
    }
    driver_blt_options opt;
    opt.bsnp_diploid_theta = theta;
    snp_pos_info pi;
    fill_pileup(pi, calls, n_calls, ref_base_id);
    std::vector<float> dep(de, de + n_calls);
    const extended_pos_info epi(pi, dep);
    diploid_genotype dgt;
    dgt.ploidy = ploidy;
    std::memset(out, 0, sizeof(*out));
    if (ref_base_id >= 4) {
        caller->position_snp_call_pprob_digt(opt, epi, dgt, true);
        return;
    }
    caller->position_snp_call_pprob_digt(opt, epi, dgt, true);
    pprob_digt_caller::get_diploid_gt_lhood(opt, epi, false, 0, out->lhood);
    for (unsigned gt = 0; gt < 10; ++gt) out->phredLoghood[gt] = dgt.phredLoghood[gt];
    out->genome = { dgt.genome.ref_pprob, dgt.genome.max_gt, dgt.genome.snp_qphred, dgt.genome.max_gt_qphred, 0 };
    out->poly = { dgt.poly.ref_pprob, dgt.poly.max_gt, dgt.poly.snp_qphred, dgt.poly.max_gt_qphred, 0 };
    out->strand_bias = dgt.strand_bias;
    out->ref_gt = dgt.ref_gt;
    out->is_called = 1;
}

/// The reference's own per-locus germline chain over a CSR pileup, as starling_pos_processor.cpp:146-266 runs it:
/// adjust_joint_eprob (PileupCleaner::CleanPileupErrorProb) then position_snp_call_pprob_digt, one locus after the
/// other.  Returns the seconds spent in those two calls (inputs are materialised in the reference's own structs outside
/// the timed region); `checksum` defeats dead-code elimination.  Used only as the "reference" CPU baseline of bench.py.
double ref_time_germline_sites(const int64_t* call_off, const uint16_t* calls, const uint8_t* ref_base, int n_loci, double theta,
                               double* checksum)
{
    pprob_digt_caller caller(theta);
    driver_blt_options opt;
    opt.bsnp_diploid_theta = theta;
    dependent_prob_cache dpc;
    double secs = 0, acc = 0;
    std::vector<float> dep;
    for (int l = 0; l < n_loci; ++l) {
        snp_pos_info pi;
        fill_pileup(pi, calls + call_off[l], int(call_off[l + 1] - call_off[l]), ref_base[l] < 4 ? ref_base[l] : 0);
        diploid_genotype dgt;
        const auto t0 = std::chrono::steady_clock::now();
        adjust_joint_eprob(opt, dpc, pi, dep);
        const extended_pos_info epi(pi, dep);
        caller.position_snp_call_pprob_digt(opt, epi, dgt, true);
        secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        acc += dgt.genome.snp_qphred + dgt.phredLoghood[1];
    }
    *checksum = acc;
    return secs;
}

void ref_germline_lnpriors(double theta, float* out /* [2][5][2][10] */)
{
    pprob_digt_caller caller(theta);
    for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 5; ++r) {
            std::memcpy(out + ((h * 5 + r) * 2 + 0) * 10, caller.lnprior_genomic(r, h != 0), sizeof(float) * 10);
            std::memcpy(out + ((h * 5 + r) * 2 + 1) * 10, caller.lnprior_polymorphic(r, h != 0), sizeof(float) * 10);
        }
}

/// 21 prestrand (+9 strand) states of one sample, via the reference's cached lhood functions
void ref_somatic_sample_lhood(const uint16_t* calls, int n_calls, unsigned ref_gt, int with_strand, float* lhood)
{
    snp_pos_info pi;
    fill_pileup(pi, calls, n_calls, ref_gt);
    for (int i = 0; i < 30; ++i) lhood[i] = 0;
    get_diploid_gt_lhood_cached_simple(pi, ref_gt, lhood);
    get_diploid_het_grid_lhood_cached(pi, ref_gt, DIGT_GRID::HET_RES, lhood + SOMATIC_DIGT::SIZE);
    if (with_strand) {
        // body of the file-static get_diploid_strand_grid_lhood_spi wrapper: one call per ratio with its own cache
        static het_ratio_cache<2> hrcache;
        for (unsigned i = 0; i < DIGT_GRID::HET_RES; ++i) {
            const blt_float_t het_ratio((i + 1) * DIGT_GRID::RATIO_INCREMENT);
            get_strand_ratio_lhood_spi(pi, ref_gt, het_ratio, i, hrcache, lhood + DIGT_GRID::PRESTRAND_SIZE + i);
        }
    }
}

void ref_calculate_result_set_grid(float contam_tolerance, float ln_sse_rate, float ln_csse_rate,
                                   const float* normal_lhood, const float* tumor_lhood, const float* lnprior3,
                                   float lnmatch, float lnmismatch, uint32_t* max_gt, int32_t* qphred,
                                   int32_t* from_ntype_qphred, uint32_t* ntype)
{
    result_set rs;
    rs.ntype = 0;
    rs.max_gt = 0;
    calculate_result_set_grid(contam_tolerance, ln_sse_rate, ln_csse_rate, normal_lhood, tumor_lhood, lnprior3, lnmatch,
                              lnmismatch, rs);
    *max_gt = rs.max_gt;
    *qphred = rs.qphred;
    *from_ntype_qphred = rs.from_ntype_qphred;
    *ntype = rs.ntype;
}

void ref_germline_genotype_log_prior(double theta, float* lnprior3) { calculateGermlineGenotypeLogPrior(theta, lnprior3); }

} // extern "C"
