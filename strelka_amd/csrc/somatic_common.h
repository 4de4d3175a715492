// somatic_common.h -- the 3x2 somatic posterior shared by the SNV (somatic_site.hip) and indel (indel_lhood.hip) callers.
#pragma once

#include "sk_common.h"

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
};

// error_prob_to_qphred<double>, L/blt_util/qscore.hh:40-47,60-66
__device__ __forceinline__ int error_prob_to_qphred_d(const double prob)
{
    const double minlog10 = -307.;
    const double l = log10(prob);
    const double m = (minlog10 < l) ? l : minlog10;
    return static_cast<int>(floor(__dadd_rn(__dmul_rn(-10., m), 0.5)));
}

// calculate_result_set_grid, L/applications/strelka/qscore_calculator.cpp:47-209.  The (Fn,Ft) enumeration order of the
// reference is kept so that the running max / sums see the terms in the same order.
template <typename ResultT>
__device__ void calculate_result_set_grid(const SomaticDerived& d, const float* normal_lhood, const float* tumor_lhood,
                                          ResultT& rs)
{
    const double neg_inf = -INFINITY;
    double log_post_prob[SOM_SIZE][2];
    double max_log_prob = neg_inf;
    unsigned max_gt = 0;
    const float RATIO_INCREMENT = 0.5f / static_cast<float>(HET_RES + 1);

    for (unsigned ngt = 0; ngt < SOM_SIZE; ++ngt) {
        for (unsigned tgt = 0; tgt < 2; ++tgt) {
            // two passes over the allowed (Ft,Fn) pairs instead of the reference's log_sum[] buffer: same max, same
            // summation order, no 441-double scratch array
            double max_log_sum = neg_inf;
            double sum = 0.0;
            for (int pass = 0; pass < 2; ++pass) {
                for (unsigned tfi = 0; tfi < PRESTRAND; ++tfi) {
                    const bool consider_norm_contam = (__fmul_rn(d.contam_tolerance, d.grid_frac[tfi]) >= RATIO_INCREMENT);
                    for (unsigned nfi = 0; nfi < PRESTRAND; ++nfi) {
                        double lprior_freq;
                        if (tgt == 0) {
                            if (nfi != tfi) continue;
                            lprior_freq = (nfi == ngt) ? static_cast<double>(d.ln_csse_rate)
                                                       : static_cast<double>(__fadd_rn(d.ln_sse_rate, d.log_error_mod));
                        } else {
                            if (nfi == tfi) continue;
                            if (ngt != SOM_REF) {
                                if (nfi != ngt) continue;
                                lprior_freq = d.log_error_mod;
                            } else {
                                if (!consider_norm_contam) {
                                    if (nfi == 0) lprior_freq = d.log_error_mod;
                                    else continue;
                                } else {
                                    if ((nfi == ngt) || (nfi == SOM_SIZE))
                                        lprior_freq = static_cast<double>(__fadd_rn(d.log_error_mod, d.ln_one_half));
                                    else continue;
                                }
                            }
                        }
                        const double lsum = __dadd_rn(__dadd_rn(lprior_freq, static_cast<double>(normal_lhood[nfi])),
                                                      static_cast<double>(tumor_lhood[tfi]));
                        if (pass == 0) {
                            if (lsum > max_log_sum) max_log_sum = lsum;
                        } else {
                            sum = __dadd_rn(sum, exp(__dsub_rn(lsum, max_log_sum)));
                        }
                    }
                }
            }
            const double log_genotype_prior = static_cast<double>(__fadd_rn(d.lnprior[ngt], (tgt == 0) ? d.ln_som_match : d.ln_som_mismatch));
            log_post_prob[ngt][tgt] = __dadd_rn(__dadd_rn(log_genotype_prior, max_log_sum), log(sum));
            if (log_post_prob[ngt][tgt] > max_log_prob) {
                max_log_prob = log_post_prob[ngt][tgt];
                max_gt = ngt * 2 + tgt;
            }
        }
    }

    double sum_prob = 0.0;
    for (unsigned ngt = 0; ngt < SOM_SIZE; ++ngt)
        for (unsigned tgt = 0; tgt < 2; ++tgt)
            sum_prob = __dadd_rn(sum_prob, exp(__dsub_rn(log_post_prob[ngt][tgt], max_log_prob)));
    const double log_sum_prob = log(sum_prob);
    double min_not_somfrom_sum = INFINITY;
    double nonsom_prob = 0.0;
    int from_ntype_qphred = 0;
    unsigned ntype = 0;
    for (unsigned ngt = 0; ngt < SOM_SIZE; ++ngt) {
        double som_prob_given_ngt = 0;
        for (unsigned tgt = 0; tgt < 2; ++tgt) {
            const double pp = exp(__dsub_rn(__dsub_rn(log_post_prob[ngt][tgt], max_log_prob), log_sum_prob));
            if (tgt == 0) nonsom_prob = __dadd_rn(nonsom_prob, pp);
            else som_prob_given_ngt = __dadd_rn(som_prob_given_ngt, pp);
        }
        const double err_som_and_ngt = __dsub_rn(1.0, som_prob_given_ngt);
        if (err_som_and_ngt < min_not_somfrom_sum) {
            min_not_somfrom_sum = err_som_and_ngt;
            from_ntype_qphred = error_prob_to_qphred_d(err_som_and_ngt);
            ntype = ngt;
        }
    }
    rs.max_gt = max_gt;
    rs.qphred = error_prob_to_qphred_d(nonsom_prob);
    rs.from_ntype_qphred = from_ntype_qphred;
    rs.ntype = ntype;
}

} // namespace
