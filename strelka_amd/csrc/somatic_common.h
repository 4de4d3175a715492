// somatic_common.h -- the 3x2 somatic posterior shared by the SNV (somatic_site.hip) and indel (indel_lhood.hip) callers.
#pragma once

#include "sk_common.h"
#include "libm_dbl64.h"

#include <cmath>
#include <cstring>

namespace
{

enum { SOM_REF = 0, SOM_HOM = 1, SOM_HET = 2, SOM_SIZE = 3, HET_RES = SK_HET_RES, PRESTRAND = 21, GRID = 30 };

struct SomaticDerived
{
    float contam_tolerance;
    float ln_csse_rate, ln_sse_rate;
    float ln_som_match, ln_som_mismatch;
    float lnprior[3];
    float log_error_mod; // -log(PRESTRAND_SIZE-1)
    float ln_one_half;   // (float) std::log(1./2.)
    float grid_frac[PRESTRAND];
    int is_forced_output;
    int exact_libm; // the host libm's float routines are the ones restated in libm_flt32.h (checked by sk_init)
};

// error_prob_to_qphred<double>, L/blt_util/qscore.hh:40-47,60-66
__device__ __forceinline__ int error_prob_to_qphred_d(const double prob, const int exact_libm, const SkLibmTables& lt)
{
    const double minlog10 = -307.;
    const double l = sk_log10_call(prob, exact_libm, lt);
    const double m = (minlog10 < l) ? l : minlog10;
    return static_cast<int>(floor(__dadd_rn(__dmul_rn(-10., m), 0.5)));
}

// calculate_result_set_grid, L/applications/strelka/qscore_calculator.cpp:47-209.  The (Fn,Ft) enumeration order of the
// reference is kept so that the running max / sums see the terms in the same order.
//
// Two shapes of the same arithmetic.  ROLLED = false: Ft is unrolled for all six (ngt,tgt) at once, so every likelihood
// index is a compile-time constant and plain arrays stay in registers (the transcendental routines are then called out of
// line: 150 inlined copies defeat the unrolling).  ROLLED = true: the Ft loop stays a loop, `normal_lhood(i)` /
// `tumor_lhood(i)` are accessors of something indexable at run time (LDS), and the routines are inlined once.
template <bool ROLLED, typename NL, typename TL, typename ResultT>
__device__ void calculate_result_set_grid(const SomaticDerived& d, const SkLibmTables& lt, const NL& normal_lhood,
                                          const TL& tumor_lhood, ResultT& rs)
{
    constexpr int UNROLL_FT = ROLLED ? 1 : int(PRESTRAND);
    auto exp_ = [&](const double x) { return ROLLED ? sk_exp(x, d.exact_libm, lt) : sk_exp_call(x, d.exact_libm, lt); };
    auto log_ = [&](const double x) { return ROLLED ? sk_log(x, d.exact_libm, lt) : sk_log_call(x, d.exact_libm, lt); };
    const double neg_inf = -INFINITY;
    double log_post_prob[SOM_SIZE][2];
    double max_log_prob = neg_inf;
    unsigned max_gt = 0;
    const float RATIO_INCREMENT = 0.5f / static_cast<float>(HET_RES + 1);

    // The reference fills log_sum[] for one (ngt,tgt) at a time, Ft outer / Fn inner, then takes the max and sums the
    // exps in that order (:75-160).  Here Ft is the outer loop for all six (ngt,tgt) at once; each (ngt,tgt) still sees its
    // own terms in the reference's order.  Two passes (max, then sum) instead of the 441-double buffer.
    double mx[SOM_SIZE][2], sm[SOM_SIZE][2];
#pragma unroll
    for (unsigned ngt = 0; ngt < SOM_SIZE; ++ngt) {
        mx[ngt][0] = mx[ngt][1] = neg_inf;
        sm[ngt][0] = sm[ngt][1] = 0.0;
    }
    const double lp_match = static_cast<double>(d.ln_csse_rate);
    const double lp_mismatch = static_cast<double>(__fadd_rn(d.ln_sse_rate, d.log_error_mod));
    const double lp_mod = static_cast<double>(d.log_error_mod);
    const double lp_half = static_cast<double>(__fadd_rn(d.log_error_mod, d.ln_one_half));
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        auto visit = [&](const unsigned ngt, const unsigned tgt, const unsigned nfi, const unsigned tfi, const double lprior_freq) {
            const double lsum = __dadd_rn(__dadd_rn(lprior_freq, static_cast<double>(normal_lhood(nfi))),
                                          static_cast<double>(tumor_lhood(tfi)));
            if (pass == 0) {
                if (lsum > mx[ngt][tgt]) mx[ngt][tgt] = lsum;
            } else {
                // exp() of anything at or below -746 is exactly 0 and adding it changes nothing
                const double dlt = __dsub_rn(lsum, mx[ngt][tgt]);
                if (!(dlt <= -746.)) sm[ngt][tgt] = __dadd_rn(sm[ngt][tgt], exp_(dlt));
            }
        };
#pragma unroll UNROLL_FT
        for (unsigned tfi = 0; tfi < PRESTRAND; ++tfi) {
#pragma unroll
            for (unsigned ngt = 0; ngt < SOM_SIZE; ++ngt) visit(ngt, 0, tfi, tfi, (tfi == ngt) ? lp_match : lp_mismatch);
            // tgt == 1 (:96-137)
            if (!(__fmul_rn(d.contam_tolerance, d.grid_frac[tfi]) >= RATIO_INCREMENT)) {
                if (tfi != 0) visit(SOM_REF, 1, 0, tfi, lp_mod);
            } else {
                if (tfi != 0) visit(SOM_REF, 1, 0, tfi, lp_half);
                if (tfi != SOM_SIZE) visit(SOM_REF, 1, SOM_SIZE, tfi, lp_half);
            }
            if (tfi != SOM_HOM) visit(SOM_HOM, 1, SOM_HOM, tfi, lp_mod);
            if (tfi != SOM_HET) visit(SOM_HET, 1, SOM_HET, tfi, lp_mod);
        }
    }
#pragma unroll
    for (unsigned ngt = 0; ngt < SOM_SIZE; ++ngt) {
#pragma unroll
        for (unsigned tgt = 0; tgt < 2; ++tgt) {
            const double log_genotype_prior = static_cast<double>(__fadd_rn(d.lnprior[ngt], (tgt == 0) ? d.ln_som_match : d.ln_som_mismatch));
            log_post_prob[ngt][tgt] = __dadd_rn(__dadd_rn(log_genotype_prior, mx[ngt][tgt]), log_(sm[ngt][tgt]));
            if (log_post_prob[ngt][tgt] > max_log_prob) {
                max_log_prob = log_post_prob[ngt][tgt];
                max_gt = ngt * 2 + tgt;
            }
        }
    }

    double sum_prob = 0.0;
    for (unsigned ngt = 0; ngt < SOM_SIZE; ++ngt)
        for (unsigned tgt = 0; tgt < 2; ++tgt)
            sum_prob = __dadd_rn(sum_prob, exp_(__dsub_rn(log_post_prob[ngt][tgt], max_log_prob)));
    const double log_sum_prob = log_(sum_prob);
    double min_not_somfrom_sum = INFINITY;
    double nonsom_prob = 0.0;
    int from_ntype_qphred = 0;
    unsigned ntype = 0;
    for (unsigned ngt = 0; ngt < SOM_SIZE; ++ngt) {
        double som_prob_given_ngt = 0;
        for (unsigned tgt = 0; tgt < 2; ++tgt) {
            const double pp = exp_(__dsub_rn(__dsub_rn(log_post_prob[ngt][tgt], max_log_prob), log_sum_prob));
            if (tgt == 0) nonsom_prob = __dadd_rn(nonsom_prob, pp);
            else som_prob_given_ngt = __dadd_rn(som_prob_given_ngt, pp);
        }
        const double err_som_and_ngt = __dsub_rn(1.0, som_prob_given_ngt);
        if (err_som_and_ngt < min_not_somfrom_sum) {
            min_not_somfrom_sum = err_som_and_ngt;
            from_ntype_qphred = error_prob_to_qphred_d(err_som_and_ngt, d.exact_libm, lt);
            ntype = ngt;
        }
    }
    rs.max_gt = max_gt;
    rs.qphred = error_prob_to_qphred_d(nonsom_prob, d.exact_libm, lt);
    rs.from_ntype_qphred = from_ntype_qphred;
    rs.ntype = ntype;
}

} // namespace
