// germline_common.h -- pieces shared by germline_site.hip (per-call-site kernels) and germline_fused.hip.
#pragma once

#include "sk_common.h"
#include "libm_dbl64.h"

#include <cmath>
#include <cstring>

namespace
{

struct GermlineDerived
{
    float lnprior[2][5][2][10]; // [haploid][ref base incl N][genome,poly][gt], pprob_digt_caller ctor (:241-258)
    float depmin[SK_NQ6];       // get_dependent_eprob(q, (float)min_vexp)  (dependent_prob_cache, adjust_joint_eprob.cpp:73-86)
    float v0min[SK_NQ6];        // logf(depmin[q]) + log_one_third : val[0] of a call whose de is the cached floor value
    float v0e[SK_NQ6];          // logf((float)error_prob(q)) + log_one_third : val[0] of a call left at its raw error prob
    float v0r0[SK_NQ6][3];      // val[0] of the call ranked 2, 3, 4 in a (strand, base) group WITHOUT a neighbouring mismatch: there
                                // mismatch_frac is exactly 0, the exponent chain of adjust_icalls_eprob (:146-178) a constant of the options
    double ssd_no_mismatch, ssd_one_mismatch;
    float min_vexp;
    int is_min_vexp;
    int is_dependent_eprob;
    float ln10f; // std::log(10.f) for ln_error_prob_to_phred<float> (qscore.hh:54)
    int exact_libm; // the host libm's powf/logf are the routines restated in libm_flt32.h (checked by sk_init)
};

// ---------------------------------------------------------------------------------------------------------------------
// scalar helpers (device restatements; the oracle has the same functions on the CPU)

// error_prob_to_qphred<double>, L/blt_util/qscore.hh:40-47,60-66
__device__ __forceinline__ int error_prob_to_qphred_d(const double prob, const int exact_libm, const SkLibmTables& lt)
{
    const double minlog10 = -307.; // std::numeric_limits<double>::min_exponent10
    const double l = sk_log10(prob, exact_libm, lt);
    const double m = (minlog10 < l) ? l : minlog10;
    return static_cast<int>(floor(__dadd_rn(__dmul_rn(-10., m), 0.5)));
}

// ln_error_prob_to_qphred<float>, L/blt_util/qscore.hh:49-57,68-74
__device__ __forceinline__ int ln_error_prob_to_qphred_f(const float lnProb, const float ln10f)
{
    const float minlog10 = -37.f; // std::numeric_limits<float>::min_exponent10
    const float q = __fdiv_rn(lnProb, ln10f);
    const float m = (minlog10 < q) ? q : minlog10;
    const float phred = static_cast<float>(__dmul_rn(-10., static_cast<double>(m)));
    return static_cast<int>(floor(__dadd_rn(static_cast<double>(phred), 0.5)));
}

// std::log(float) of the reference = the host libm's logf: restated bit for bit (libm_flt32.h); the device library's
// double log rounded once is the stand-in when the host libm is not the implementation restated there
__device__ __forceinline__ float logf_ref(const float x, const int exact_libm)
{
    float r;
    if (exact_libm && sk_libm::logf_glibc(x, r)) return r;
    return static_cast<float>(log(static_cast<double>(x)));
}

// get_dependent_eprob, adjust_joint_eprob.cpp:58-69 (float throughout; std::pow(float,float) = the host libm's powf)
__device__ __forceinline__ float get_dependent_eprob(const float eprob, const float vexp, const int exact_libm)
{
    const float dep_converge_prob = 0.75f;
    float val;
    if (!(exact_libm && sk_libm::powf_glibc(eprob, vexp, val)))
        val = static_cast<float>(pow(static_cast<double>(eprob), static_cast<double>(vexp)));
    const float frac = __fdiv_rn(__fsub_rn(1.f, val), __fsub_rn(1.f, eprob));
    const float dep = __fadd_rn(__fmul_rn(frac, val), __fmul_rn(__fsub_rn(1.f, frac), dep_converge_prob));
    return (eprob < dep) ? dep : eprob;
}


// genotype index -> its two alleles (DIGT::get_allele, L/blt_util/digt.hh:148-170)
__device__ __forceinline__ unsigned digt_a0(const unsigned gt)
{
    constexpr unsigned char A0[10] = { 0, 1, 2, 3, 0, 0, 0, 1, 1, 2 };
    return A0[gt];
}
__device__ __forceinline__ unsigned digt_a1(const unsigned gt)
{
    constexpr unsigned char A1[10] = { 0, 1, 2, 3, 1, 2, 3, 2, 3, 3 };
    return A1[gt];
}

// calculate_result_set, position_snp_call_pprob_digt.cpp:412-433 (normalizeLogDistro + prob_comp, prob_util.hh:177-237)
__device__ void calculate_result_set(const float* lhood, const float* lnprior, const unsigned ref_gt, const int exact_libm,
                                     const SkLibmTables& lt, sk_digt_result_set& rs)
{
    double pprob[10];
#pragma unroll
    for (int gt = 0; gt < 10; ++gt) pprob[gt] = static_cast<double>(__fadd_rn(lhood[gt], lnprior[gt]));
    unsigned max_idx = 0;
    double mx = pprob[0];
#pragma unroll
    for (int i = 1; i < 10; ++i)
        if (pprob[i] > mx) {
            mx = pprob[i];
            max_idx = i;
        }
    double sum = 0.;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        pprob[i] = sk_exp(__dsub_rn(pprob[i], mx), exact_libm, lt);
        sum = __dadd_rn(sum, pprob[i]);
    }
    sum = __ddiv_rn(1., sum);
    double refp = 0., comp = 0.;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        pprob[i] = __dmul_rn(pprob[i], sum);
        if (unsigned(i) == ref_gt) refp = pprob[i];
        if (unsigned(i) != max_idx) comp = __dadd_rn(comp, pprob[i]);
    }
    rs.max_gt = max_idx;
    rs.ref_pprob = refp;
    rs.snp_qphred = error_prob_to_qphred_d(refp, exact_libm, lt);
    rs.max_gt_qphred = error_prob_to_qphred_d(comp, exact_libm, lt);
    rs._pad = 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// host: option-derived constants, built with the host libm from the reference's expressions

inline void build_priors(const float theta, float out[2][5][2][10])
{
    // get_genomic_prior/get_poly_prior/get_haploid_* + finish_prior, position_snp_call_pprob_digt.cpp:50-258
    static const unsigned char A0[10] = { 0, 1, 2, 3, 0, 0, 0, 1, 1, 2 }, A1[10] = { 0, 1, 2, 3, 1, 2, 3, 2, 3, 3 };
    const float one_third(1. / 3.);
    std::memset(out, 0, sizeof(float) * 200);
    for (unsigned ref_gt = 0; ref_gt < 4; ++ref_gt) {
        auto has_ref = [&](unsigned gt) { return A0[gt] == ref_gt || A1[gt] == ref_gt; };
        {
            float* prior = out[0][ref_gt][0];
            float prior_sum(0.);
            for (unsigned gt = 0; gt < 10; ++gt) {
                if (gt == ref_gt) continue;
                prior[gt] = (theta * one_third);
                if (gt >= 4) {
                    if (!has_ref(gt)) prior[gt] *= theta;
                } else {
                    prior[gt] *= .5;
                }
                prior_sum += prior[gt];
            }
            prior[ref_gt] = (1. - prior_sum);
        }
        {
            float* prior = out[0][ref_gt][1];
            const float ctheta(1. - theta);
            for (unsigned gt = 0; gt < 10; ++gt) {
                if (gt == ref_gt) prior[gt] = 0.25 * (ctheta);
                else if (gt >= 4) {
                    if (!has_ref(gt)) prior[gt] = theta * one_third;
                    else prior[gt] = 0.5 * one_third * ctheta;
                } else prior[gt] = 0.25 * one_third * ctheta;
            }
        }
        {
            float* prior = out[1][ref_gt][0];
            float prior_sum(0.);
            for (unsigned gt = 0; gt < 10; ++gt) {
                if (gt == ref_gt) continue;
                if (gt >= 4) prior[gt] = 0;
                else prior[gt] = (theta * one_third);
                prior_sum += prior[gt];
            }
            prior[ref_gt] = (1. - prior_sum);
        }
        {
            float* prior = out[1][ref_gt][1];
            for (unsigned gt = 0; gt < 10; ++gt) {
                if (gt == ref_gt) prior[gt] = 0.5;
                else if (gt >= 4) prior[gt] = 0;
                else prior[gt] = 0.5 * one_third;
            }
        }
    }
    for (int h = 0; h < 2; ++h) {
        for (unsigned i = 0; i < 4; ++i)
            for (int w = 0; w < 2; ++w)
                for (unsigned gt = 0; gt < 10; ++gt) out[h][4][w][gt] += out[h][i][w][gt];
        for (int w = 0; w < 2; ++w) {
            float* x = out[h][4][w];
            float sum(0);
            for (unsigned gt = 0; gt < 10; ++gt) sum += x[gt];
            sum = 1. / sum;
            for (unsigned gt = 0; gt < 10; ++gt) x[gt] *= sum;
        }
        for (unsigned i = 0; i < 5; ++i)
            for (int w = 0; w < 2; ++w)
                for (unsigned gt = 0; gt < 10; ++gt) out[h][i][w][gt] = std::log(out[h][i][w][gt]);
    }
}

inline float host_dependent_eprob(const float eprob, const float vexp)
{
    const float dep_converge_prob(0.75);
    const float val(std::pow(eprob, vexp));
    const float frac((1 - val) / (1 - eprob));
    return std::max(eprob, frac * val + (1 - frac) * dep_converge_prob);
}

inline void derive(const sk_germline_options& opt, GermlineDerived& d)
{
    std::memset(&d, 0, sizeof(d));
    build_priors(static_cast<float>(opt.bsnp_diploid_theta), d.lnprior);
    d.ssd_no_mismatch = opt.bsnp_ssd_no_mismatch;
    d.ssd_one_mismatch = opt.bsnp_ssd_one_mismatch;
    d.min_vexp = static_cast<float>(opt.min_vexp);
    d.is_min_vexp = opt.is_min_vexp ? 1 : 0;
    d.is_dependent_eprob = (opt.bsnp_ssd_no_mismatch > 0. || opt.bsnp_ssd_one_mismatch > 0) ? 1 : 0;
    const SkTables& t = sk_ctx().host_tables;
    for (int q = 0; q < SK_NQ6; ++q) {
        d.depmin[q] = host_dependent_eprob(t.g_eprob[q], d.min_vexp);
        d.v0min[q] = std::log(d.depmin[q]) + t.g_log_one_third; // position_snp_call_pprob_digt.cpp:352, float logf + float add
        d.v0e[q] = std::log(t.g_eprob[q]) + t.g_log_one_third;
    }
    {
        // a group none of whose calls has a neighbouring mismatch: mismatch_frac = 0 / den = 0 exactly (adjust_joint_eprob.cpp:128-133),
        // so vexp_frac and the exponents of its sorted calls (:146-178) do not depend on the data; the same expressions as the kernels'
        const float mismatch_frac(0.f);
        const float vexp_frac(static_cast<float>(static_cast<double>(1.f - mismatch_frac) * d.ssd_no_mismatch +
                                                 static_cast<double>(mismatch_frac) * d.ssd_one_mismatch));
        const float m(1.f - vexp_frac);
        float vexp(1.f);
        for (int rank = 1; rank <= 4; ++rank) {
            if (rank >= 2) {
                for (int q = 0; q < SK_NQ6; ++q)
                    d.v0r0[q][rank - 2] = std::log(host_dependent_eprob(t.g_eprob[q], vexp)) + t.g_log_one_third;
            }
            const float next_vexp(vexp * m);
            vexp = d.is_min_vexp ? ((d.min_vexp < next_vexp) ? next_vexp : d.min_vexp) : next_vexp;
        }
    }
    volatile float ten = 10.f;
    d.ln10f = std::log(static_cast<float>(ten));
    d.exact_libm = sk_ctx().libm_restated ? 1 : 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// libstdc++ std::sort on an index array with comp(a,b) = q[a] > q[b]   (adjust_joint_eprob.cpp:41-53,145)
// (GCC bits/stl_algo.h __introsort_loop/__final_insertion_sort, threshold 16; bits/stl_heap.h for the depth fallback)

struct SortKey
{
    const uint16_t* calls; // locus-local packed calls
    __device__ __forceinline__ bool gt(const uint32_t a, const uint32_t b) const { return SKC_Q(calls[a]) > SKC_Q(calls[b]); }
};

__device__ void s_unguarded_linear_insert(uint32_t* last, const SortKey& key)
{
    const uint32_t val = *last;
    uint32_t* next = last - 1;
    while (key.gt(val, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

__device__ void s_insertion_sort(uint32_t* first, uint32_t* last, const SortKey& key)
{
    if (first == last) return;
    for (uint32_t* i = first + 1; i != last; ++i) {
        if (key.gt(*i, *first)) {
            const uint32_t val = *i;
            for (uint32_t* p = i; p != first; --p) *p = *(p - 1);
            *first = val;
        } else {
            s_unguarded_linear_insert(i, key);
        }
    }
}

__device__ void s_adjust_heap(uint32_t* first, long hole, const long len, const uint32_t value, const SortKey& key)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (key.gt(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    long parent = (hole - 1) / 2; // __push_heap
    while (hole > top && key.gt(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

__device__ void s_heap_sort(uint32_t* first, uint32_t* last, const SortKey& key)
{
    const long len = last - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            const uint32_t value = first[parent];
            s_adjust_heap(first, parent, len, value, key);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {
        --last;
        const uint32_t value = *last;
        *last = *first;
        s_adjust_heap(first, 0, last - first, value, key);
    }
}

__device__ void std_sort_emulated(uint32_t* idx, const int n, const SortKey& key)
{
    if (n <= 0) return;
    if (n > 16) {
        long lg = 0;
        for (unsigned m = unsigned(n); m > 1; m >>= 1) ++lg;
        // explicit stack for the (cut,last) recursion of __introsort_loop
        int st_first[64], st_last[64], st_depth[64];
        int sp = 0;
        st_first[0] = 0;
        st_last[0] = n;
        st_depth[0] = int(lg * 2);
        sp = 1;
        while (sp > 0) {
            --sp;
            int first = st_first[sp], last = st_last[sp], depth = st_depth[sp];
            while (last - first > 16) {
                if (depth == 0) {
                    s_heap_sort(idx + first, idx + last, key);
                    break;
                }
                --depth;
                // __move_median_to_first(first, first+1, mid, last-1)
                uint32_t* a = idx + first + 1;
                uint32_t* b = idx + first + (last - first) / 2;
                uint32_t* c = idx + last - 1;
                uint32_t* pick;
                if (key.gt(*a, *b)) {
                    if (key.gt(*b, *c)) pick = b;
                    else if (key.gt(*a, *c)) pick = c;
                    else pick = a;
                } else if (key.gt(*a, *c)) pick = a;
                else if (key.gt(*b, *c)) pick = c;
                else pick = b;
                {
                    const uint32_t t = idx[first];
                    idx[first] = *pick;
                    *pick = t;
                }
                // __unguarded_partition(first+1, last, pivot=first)
                uint32_t* lo = idx + first + 1;
                uint32_t* hi = idx + last;
                const uint32_t* pivot = idx + first;
                for (;;) {
                    while (key.gt(*lo, *pivot)) ++lo;
                    --hi;
                    while (key.gt(*pivot, *hi)) --hi;
                    if (!(lo < hi)) break;
                    const uint32_t t = *lo;
                    *lo = *hi;
                    *hi = t;
                    ++lo;
                }
                const int cut = int(lo - idx);
                // recurse on [cut,last) first (as the reference does), then continue with [first,cut):
                // emulate by finishing the right part now via the stack: push the LEFT remainder, iterate the right.
                // Order of processing disjoint ranges does not affect the result.
                st_first[sp] = first;
                st_last[sp] = cut;
                st_depth[sp] = depth;
                ++sp;
                first = cut;
            }
        }
        s_insertion_sort(idx, idx + 16, key);
        for (uint32_t* i = idx + 16; i != idx + n; ++i) s_unguarded_linear_insert(i, key);
    } else {
        s_insertion_sort(idx, idx + n, key);
    }
}

// adjust_joint_eprob for one locus, global-memory version (L/blt_common/adjust_joint_eprob.cpp:201-243)
__device__ void locus_dependent_eprob_global(const sk_pileup_batch& B, const SkTables* tab, const GermlineDerived& D,
                                             float* out_de, uint32_t* scratch, const int l)
{
    const int64_t off = B.call_off[l];
    const int n = int(B.call_off[l + 1] - off);
    const uint16_t* __restrict__ calls = B.calls + off;
    float* __restrict__ de = out_de + off;
    const SkTables* __restrict__ T = tab;

    for (int i = 0; i < n; ++i) de[i] = T->g_eprob[SKC_Q(calls[i])];
    if (!D.is_dependent_eprob || n == 0) return;

    // bucket the usable calls into the 8 (strand, base) groups, pileup order preserved inside a group (:215-234)
    uint32_t* ic = scratch + off;
    int start[9];
    {
        int cnt[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int i = 0; i < n; ++i) {
            const uint16_t b = calls[i];
            if (SKC_FILTER(b) || SKC_Q(b) < 3 || SKC_BASE(b) > 3) continue;
            ++cnt[SKC_FWD(b) + 2 * SKC_BASE(b)];
        }
        start[0] = 0;
        for (int g = 0; g < 8; ++g) start[g + 1] = start[g] + cnt[g];
        int fill[8];
        for (int g = 0; g < 8; ++g) fill[g] = start[g];
        for (int i = 0; i < n; ++i) {
            const uint16_t b = calls[i];
            if (SKC_FILTER(b) || SKC_Q(b) < 3 || SKC_BASE(b) > 3) continue;
            ic[fill[SKC_FWD(b) + 2 * SKC_BASE(b)]++] = uint32_t(i);
        }
    }
    const SortKey key{ calls };
    for (int g = 0; g < 8; ++g) {
        uint32_t* gi = ic + start[g];
        const int ic_size = start[g + 1] - start[g];
        if (ic_size == 0) continue;
        // weighted fraction of calls with a neighbouring mismatch (:112-133)
        float num = 0.f, den = 0.f;
        for (int i = 0; i < ic_size; ++i) {
            const uint16_t bi = calls[gi[i]];
            const float weight = T->g_weight[SKC_Q(bi)];
            den = __fadd_rn(den, weight);
            if (SKC_NMM(bi)) num = __fadd_rn(num, weight);
        }
        float mismatch_frac = 0.f;
        if (den > 0.) mismatch_frac = __fdiv_rn(num, den);
        const float vexp_frac = static_cast<float>(
            __dadd_rn(__dmul_rn(static_cast<double>(__fsub_rn(1.f, mismatch_frac)), D.ssd_no_mismatch),
                      __dmul_rn(static_cast<double>(mismatch_frac), D.ssd_one_mismatch)));

        std_sort_emulated(gi, ic_size, key);

        bool is_min_vexp = false;
        float vexp = 1.f;
        for (int i = 0; i < ic_size; ++i) {
            const uint32_t ci = gi[i];
            const unsigned q = SKC_Q(calls[ci]);
            if (!is_min_vexp) {
                de[ci] = get_dependent_eprob(T->g_eprob[q], vexp, D.exact_libm);
                const float next_vexp = __fmul_rn(vexp, __fsub_rn(1.f, vexp_frac));
                if (D.is_min_vexp) {
                    is_min_vexp = (next_vexp <= D.min_vexp);
                    vexp = (D.min_vexp < next_vexp) ? next_vexp : D.min_vexp;
                } else {
                    vexp = next_vexp;
                }
            } else {
                de[ci] = D.depmin[q]; // dependent_prob_cache: always filled at vexp == min_vexp
            }
        }
    }
}

// position_snp_call_pprob_digt for one locus, global-memory version with de[] given
// (L/blt_common/position_snp_call_pprob_digt.cpp:473-539)
__device__ void locus_site_digt_call_global(const sk_pileup_batch& B, const float* de_all, const SkTables* tab,
                                            const GermlineDerived& D, sk_digt_call* out, const int l)
{
    const int64_t off = B.call_off[l];
    const int n = int(B.call_off[l + 1] - off);
    const uint16_t* __restrict__ calls = B.calls + off;
    const float* __restrict__ de = de_all + off;
    const SkTables* __restrict__ T = tab;
    const unsigned ref = B.ref_base[l];
    const int ploidy = B.ploidy ? int(B.ploidy[l]) : 2;

    sk_digt_call res;
    memset(&res, 0, sizeof(res));
    if (ref >= 4) { // 'N' (:481)
        out[l] = res;
        return;
    }
    res.is_called = 1;
    res.ref_gt = ref;
    const bool is_haploid = (ploidy == 1);
    const float log_one_third = T->g_log_one_third;

    float lh[10];
#pragma unroll
    for (int gt = 0; gt < 10; ++gt) lh[gt] = 0.f;
    for (int i = 0; i < n; ++i) {
        const uint16_t bc = calls[i];
        const unsigned q = SKC_Q(bc), obs = SKC_BASE(bc);
        const float v0 = __fadd_rn(logf_ref(de[i], D.exact_libm), log_one_third); // val[0] (:352)
        const float v1 = T->g_v1[q];                             // val[1] (:353)
        const float v2 = T->g_v2[q];                             // val[2] (:354)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) lh[gt] = __fadd_rn(lh[gt], (obs == unsigned(gt)) ? v2 : v0);
#pragma unroll
        for (int gt = 4; gt < 10; ++gt)
            lh[gt] = __fadd_rn(lh[gt], (obs == digt_a0(gt) || obs == digt_a1(gt)) ? v1 : v0);
    }
#pragma unroll
    for (int gt = 0; gt < 10; ++gt) res.lhood[gt] = lh[gt];

    { // PLs (:499-511)
        const int gtcount = is_haploid ? 4 : 10;
        int maxIndex = 0;
        for (int gt = 1; gt < gtcount; ++gt)
            if (lh[gt] > lh[maxIndex]) maxIndex = gt;
        for (int gt = 0; gt < gtcount; ++gt)
            res.phredLoghood[gt] = unsigned(ln_error_prob_to_qphred_f(__fsub_rn(lh[gt], lh[maxIndex]), D.ln10f));
    }
    const float* pg = D.lnprior[is_haploid ? 1 : 0][ref][0];
    const float* pp = D.lnprior[is_haploid ? 1 : 0][ref][1];
    const SkLibmTables lt = sk_libm_tables_default();
    calculate_result_set(lh, pg, ref, D.exact_libm, lt, res.genome);
    calculate_result_set(lh, pp, ref, D.exact_libm, lt, res.poly);

    if (res.genome.snp_qphred != 0) { // strand bias (:520-534): only lhood_{fwd,rev}[max_gt] are consumed
        const unsigned tgt = res.genome.max_gt;
        const unsigned t0 = digt_a0(tgt), t1 = digt_a1(tgt);
        float lf = 0.f, lr = 0.f;
        for (int i = 0; i < n; ++i) {
            const uint16_t bc = calls[i];
            const unsigned q = SKC_Q(bc), obs = SKC_BASE(bc);
            const float v0 = __fadd_rn(logf_ref(de[i], D.exact_libm), log_one_third);
            const float v1 = T->g_v1[q];
            const float v2 = T->g_v2[q];
            const float val_ref = (obs == ref) ? v2 : v0; // expect2(obs, ref_gt): ref_gt is homozygous
            const float val_tgt = (tgt < 4) ? ((obs == tgt) ? v2 : v0) : ((obs == t0 || obs == t1) ? v1 : v0);
            const bool fwd = SKC_FWD(bc);
            lf = __fadd_rn(lf, fwd ? val_tgt : val_ref); // is_ss_fwd=true: other-strand calls forced to ref (:356)
            lr = __fadd_rn(lr, fwd ? val_ref : val_tgt);
        }
        const float m = (lf < lr) ? lr : lf;
        res.strand_bias = static_cast<double>(__fsub_rn(m, lh[tgt]));
    }
    out[l] = res;
}

} // namespace
