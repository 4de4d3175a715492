// indel_lhood.hip -- hot path B (indels): reductions over IndelSampleData::read_path_lnp.
//
//   I1 `indel_grid_lhood_kernel`      : get_indel_digt_lhood (L/starling_common/starling_indel_call_pprob_digt.cpp:240-336)
//                                       + get_indel_het_grid_lhood (L/applications/strelka/somatic_indel_grid.cpp:66-89)
//                                       = the 21 somatic-grid states of one sample per indel
//   I2 `somatic_indel_posterior_kernel`: float cast + calculate_result_set_grid with the indel priors
//                                       (somatic_indel_grid.cpp:264-287, qscore_calculator.cpp:47-209)
//   I3 `allele_group_kernel`          : getVariantAlleleGroupGenotypeLhoodsForSample
//                                       (L/starling_common/AlleleGroupGenotype.cpp:185-258)
//
// Mapping: one 64-lane wavefront per indel / allele group.  The reference adds one double term per read to each state
// (sequentially, in read-id order); the TERMS are independent, so lanes evaluate them for 64 reads at a time into LDS,
// then one lane per state performs the adds in read order -- the order and rounding of every add is the reference's.
//
// Every term is evaluated operation for operation as the reference does (getLogSum / log1p_switch / integrateOutMappingStatus
// / get_het_observed_allele_ratio), and its double exp / log / log1p calls are the host libm's routines restated in
// libm_dbl64.h: the likelihoods are bit-identical to the reference's.  (An algebraically equal form with one log per state
// and the read's exponentials shared by its states is 2.5x faster, 2.3 instead of ~6 ms per 2^18 indels, but only agrees to
// ~1e-15; indel loci are a thousandth of all loci, so the exact form is the one that ships.)

#include "somatic_common.h"

#include <unordered_map>
#include <vector>

namespace
{

constexpr int WAVE = 64;
constexpr int N_STATES = 21;

struct MapParams
{
    double correct_mapping_log_prior; // log(1.7e-10), starling_base_shared.cpp:64
    double random_base_match_log_prob; // log(randomBaseMatchProb) of the pass (tier2 passes use the tier2 value)
};

struct GridArgs
{
    sk_readscore_batch b;
    MapParams map;
    int min_read_bp_flank;
    int is_include_tier2;
    int is_use_alt_indel;
    double* out; // [n_indels][21]
    int exact_libm;
    double het_ratio[SK_HET_RES], chet_ratio[SK_HET_RES], log_het_ratio[SK_HET_RES], log_chet_ratio[SK_HET_RES];
    double loghalf;
};

// log1p_switch / getLogSum, L/blt_util/math_util.hh:33-48, L/blt_util/logSumUtil.hh:33-41 (double)
__device__ __forceinline__ double log1p_switch_d(const double x, const int ex, const SkLibmTables& lt)
{
    return (fabs(x) < 0.01) ? sk_log1p(x, ex) : sk_log(__dadd_rn(1., x), ex, lt);
}
__device__ __forceinline__ double log_sum2(double x1, double x2, const int ex, const SkLibmTables& lt)
{
    if (x1 < x2) {
        const double t = x1;
        x1 = x2;
        x2 = t;
    }
    return __dadd_rn(x1, log1p_switch_d(sk_exp(__dsub_rn(x2, x1), ex, lt), ex, lt));
}

// integrateOutMappingStatus, L/starling_common/readMappingAdjustmentUtil.hh:29-56
__device__ __forceinline__ double integrate_out_mapping(const MapParams& m, const unsigned non_ambig, const double lnp, const int ex,
                                                        const SkLibmTables& lt)
{
    return log_sum2(__dadd_rn(lnp, m.correct_mapping_log_prior), __dmul_rn(m.random_base_match_log_prob, double(non_ambig)), ex, lt);
}

// get_het_observed_allele_ratio, starling_indel_call_pprob_digt.cpp:40-71
__device__ __forceinline__ void het_observed_allele_ratio(const unsigned read_length, const unsigned min_overlap,
                                                          const unsigned del_len, const unsigned ins_len,
                                                          const double het_allele_ratio, double& log_ref_prob,
                                                          double& log_indel_prob, const int ex, const SkLibmTables& lt)
{
    const unsigned base_expect = ((read_length + 1) < (2 * min_overlap)) ? 0 : (read_length + 1) - (2 * min_overlap);
    const double ref_path_expect = double(base_expect + min(del_len, base_expect));
    const double indel_path_expect = double(base_expect + min(ins_len, base_expect));
    const double ref_path_term = __dmul_rn(__dsub_rn(1., het_allele_ratio), ref_path_expect);
    const double indel_path_term = __dmul_rn(het_allele_ratio, indel_path_expect);
    const double total_path_term = __dadd_rn(ref_path_term, indel_path_term);
    if (total_path_term > 0) {
        const double indel_prob = __ddiv_rn(indel_path_term, total_path_term);
        log_ref_prob = sk_log(__dsub_rn(1., indel_prob), ex, lt);
        log_indel_prob = sk_log(indel_prob, ex, lt);
    }
}

// ---- the "fast form" (sk_indel_options.fast_form): algebraically the same terms with two exp per read shared by its 21
// states and one log per state.  Not the reference's operation order: likelihoods agree to ~1e-15 relative, not bit for bit
// (north_star allows 1e-5 on log-likelihoods); the default is the exact form.
__device__ __forceinline__ bool het_observed_indel_prob(const unsigned read_length, const unsigned min_overlap, const unsigned del_len,
                                                        const unsigned ins_len, const double het_allele_ratio, double& ref_prob,
                                                        double& indel_prob)
{
    const unsigned base_expect = ((read_length + 1) < (2 * min_overlap)) ? 0 : (read_length + 1) - (2 * min_overlap);
    const double ref_path_expect = double(base_expect + min(del_len, base_expect));
    const double indel_path_expect = double(base_expect + min(ins_len, base_expect));
    const double ref_path_term = __dmul_rn(__dsub_rn(1., het_allele_ratio), ref_path_expect);
    const double indel_path_term = __dmul_rn(het_allele_ratio, indel_path_expect);
    const double total_path_term = __dadd_rn(ref_path_term, indel_path_term);
    if (!(total_path_term > 0)) return false;
    indel_prob = __ddiv_rn(indel_path_term, total_path_term);
    ref_prob = __dsub_rn(1., indel_prob);
    return true;
}

// One read's term of a het state, integrateOutMappingStatus(logsum(noindel + log(1-p), hom + log(p))), evaluated as
//     T + log( (e^(A-T) * ((1-p) e^(a-m) + p e^(b-m))) + e^(M-T) )
// with a = noindel, b = hom, m = max(a,b), A = m + correct_mapping_log_prior, M = random_base_match_log_prob * nonAmbig,
// T = max(A, M)
struct ReadExp
{
    double T, wa, wb, wm; // wa = e^(A-T) e^(a-m), wb = e^(A-T) e^(b-m), wm = e^(M-T)
};
__device__ __forceinline__ ReadExp read_exponentials(const MapParams& mp, const unsigned non_ambig, const double a, const double b,
                                                     const int ex, const SkLibmTables& lt)
{
    const bool a_lt_b = (a < b);
    const double m = a_lt_b ? b : a;
    const double A = __dadd_rn(m, mp.correct_mapping_log_prior);
    const double M = __dmul_rn(mp.random_base_match_log_prob, double(non_ambig));
    const bool A_lt_M = (A < M);
    ReadExp r;
    r.T = A_lt_M ? M : A;
    double e1 = sk_exp(-fabs(__dsub_rn(a, b)), ex, lt);
    if (!(e1 == e1)) e1 = 1.; // a == b == -inf
    const double eT = sk_exp(-fabs(__dsub_rn(A, M)), ex, lt);
    const double sA = A_lt_M ? eT : 1.;
    r.wm = A_lt_M ? 1. : eT;
    r.wa = __dmul_rn(sA, a_lt_b ? e1 : 1.);
    r.wb = __dmul_rn(sA, a_lt_b ? 1. : e1);
    return r;
}
__device__ __forceinline__ double mix_term(const ReadExp& r, const double ref_prob, const double indel_prob, const int ex,
                                           const SkLibmTables& lt)
{
    const double mix = __dadd_rn(__dmul_rn(ref_prob, r.wa), __dmul_rn(indel_prob, r.wb));
    return __dadd_rn(r.T, sk_log(__dadd_rn(mix, r.wm), ex, lt));
}

template <bool FAST>
__global__ __launch_bounds__(WAVE) void indel_grid_lhood_kernel(const GridArgs a)
{
    __shared__ double s_term[N_STATES][WAVE];
    const int ind = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t r0 = a.b.read_off[ind];
    const int n = int(a.b.read_off[ind + 1] - r0);
    const unsigned del_len = a.b.del_len[ind], ins_len = a.b.ins_len[ind];
    const bool is_breakpoint = a.b.is_breakpoint ? (a.b.is_breakpoint[ind] != 0) : false;
    const unsigned flank = unsigned(a.min_read_bp_flank);
    const int ex = a.exact_libm;
    const SkLibmTables lt = sk_libm_tables_default();

    // The two logs of get_het_observed_allele_ratio depend on (read length, the indel's lengths, the state's ratio) and on nothing else
    // of the read: when all reads of the indel have one length -- every WGS sample -- they are 19 pairs per indel, evaluated once, by
    // the operations the per-read form performs (the same values bit for bit), instead of 19 pairs per read: two of the six
    // transcendentals of a (read, het state).
    __shared__ double s_lr[N_STATES], s_li[N_STATES];
    bool one_length = false;
    if (!FAST) {
        unsigned lo = 0xffffffffu, hi = 0;
        for (int r = lane; r < n; r += WAVE) {
            const unsigned rl = a.b.read_length[r0 + r];
            lo = min(lo, rl);
            hi = max(hi, rl);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo = min(lo, unsigned(__shfl_xor(int(lo), d)));
            hi = max(hi, unsigned(__shfl_xor(int(hi), d)));
        }
        one_length = (n > 0 && lo == hi);
        if (one_length && lane < 2 * SK_HET_RES + 1) {
            // lane 0: the 0.5 state (2); lanes 1..HET_RES: het_lhood_low of i = lane - 1 (state 3 + i); the rest: het_lhood_high of i
            // (state 3 + 2*HET_RES - (i + 1))
            const bool mid = (lane == 0), low = (lane >= 1 && lane <= SK_HET_RES);
            const int i = mid ? 0 : low ? lane - 1 : lane - 1 - SK_HET_RES;
            double lr = mid ? a.loghalf : low ? a.log_chet_ratio[i] : a.log_het_ratio[i];
            double li = mid ? a.loghalf : low ? a.log_het_ratio[i] : a.log_chet_ratio[i];
            const double ratio = mid ? 0.5 : low ? a.het_ratio[i] : a.chet_ratio[i];
            if (!is_breakpoint) het_observed_allele_ratio(lo, flank, del_len, ins_len, ratio, lr, li, ex, lt);
            const int state = mid ? 2 : low ? 3 + i : 3 + (2 * SK_HET_RES - (i + 1));
            s_lr[state] = lr;
            s_li[state] = li;
        }
        __syncthreads();
    }

    double acc = 0.; // lanes 0..20: the running sum of state `lane`
    for (int base = 0; base < n; base += WAVE) {
        const int r = base + lane;
        const int cnt = min(WAVE, n - base);
        if (r < n) {
            const int64_t g = r0 + r;
            const unsigned flags = a.b.read_flags[g];
            const bool use = a.is_include_tier2 || (flags & SK_READ_TIER1);
            if (!use) {
#pragma unroll
                for (int s = 0; s < N_STATES; ++s) s_term[s][lane] = 0.; // skipped read: x + 0.0 == x
            } else {
                double alt_path_lnp = double(a.b.ref_lnp[g]);
                if (a.is_use_alt_indel && a.b.alt_lnp) {
                    const float al = a.b.alt_lnp[g];
                    if (al == al && double(al) > alt_path_lnp) alt_path_lnp = double(al);
                }
                const double noindel_lnp = alt_path_lnp;
                const double hom_lnp = double(a.b.indel_lnp[g]);
                const unsigned na = a.b.non_ambig[g];
                const unsigned rl = a.b.read_length[g];
                if (FAST) {
                    const ReadExp e = read_exponentials(a.map, na, noindel_lnp, hom_lnp, ex, lt);
                    s_term[0][lane] = mix_term(e, 1., 0., ex, lt);
                    s_term[1][lane] = mix_term(e, 0., 1., ex, lt);
                    {
                        double pr = 0.5, pi = 0.5;
                        if (!is_breakpoint) het_observed_indel_prob(rl, flank, del_len, ins_len, 0.5, pr, pi);
                        s_term[2][lane] = mix_term(e, pr, pi, ex, lt);
                    }
                    for (int i = 0; i < SK_HET_RES; ++i) {
                        {
                            double pr = a.chet_ratio[i], pi = a.het_ratio[i];
                            if (!is_breakpoint) het_observed_indel_prob(rl, flank, del_len, ins_len, a.het_ratio[i], pr, pi);
                            s_term[3 + i][lane] = mix_term(e, pr, pi, ex, lt);
                        }
                        {
                            double pr = a.het_ratio[i], pi = a.chet_ratio[i];
                            if (!is_breakpoint) het_observed_indel_prob(rl, flank, del_len, ins_len, a.chet_ratio[i], pr, pi);
                            s_term[3 + (2 * SK_HET_RES - (i + 1))][lane] = mix_term(e, pr, pi, ex, lt);
                        }
                    }
                } else {
                // SOMATIC_DIGT / STAR_DIINDEL: 0 = REF/NOINDEL, 1 = HOM, 2 = HET
                s_term[0][lane] = integrate_out_mapping(a.map, na, noindel_lnp, ex, lt);
                s_term[1][lane] = integrate_out_mapping(a.map, na, hom_lnp, ex, lt);
                if (one_length) { // (wave-uniform)
                    for (int s = 2; s < N_STATES; ++s)
                        s_term[s][lane] =
                            integrate_out_mapping(a.map, na, log_sum2(__dadd_rn(noindel_lnp, s_lr[s]), __dadd_rn(hom_lnp, s_li[s]), ex, lt), ex, lt);
                } else {
                {
                    double lr = a.loghalf, li = a.loghalf;
                    if (!is_breakpoint) het_observed_allele_ratio(rl, flank, del_len, ins_len, 0.5, lr, li, ex, lt);
                    s_term[2][lane] = integrate_out_mapping(a.map, na, log_sum2(__dadd_rn(noindel_lnp, lr), __dadd_rn(hom_lnp, li), ex, lt), ex, lt);
                }
                for (int i = 0; i < SK_HET_RES; ++i) {
                    { // het_lhood_low -> grid[i]
                        double lr = a.log_chet_ratio[i], li = a.log_het_ratio[i];
                        if (!is_breakpoint) het_observed_allele_ratio(rl, flank, del_len, ins_len, a.het_ratio[i], lr, li, ex, lt);
                        s_term[3 + i][lane] =
                            integrate_out_mapping(a.map, na, log_sum2(__dadd_rn(noindel_lnp, lr), __dadd_rn(hom_lnp, li), ex, lt), ex, lt);
                    }
                    { // het_lhood_high -> grid[2*HET_RES-(i+1)]
                        double lr = a.log_het_ratio[i], li = a.log_chet_ratio[i];
                        if (!is_breakpoint) het_observed_allele_ratio(rl, flank, del_len, ins_len, a.chet_ratio[i], lr, li, ex, lt);
                        s_term[3 + (2 * SK_HET_RES - (i + 1))][lane] =
                            integrate_out_mapping(a.map, na, log_sum2(__dadd_rn(noindel_lnp, lr), __dadd_rn(hom_lnp, li), ex, lt), ex, lt);
                    }
                }
                }
                } // exact form
            }
        }
        __syncthreads();
        if (lane < N_STATES)
            for (int j = 0; j < cnt; ++j) acc = __dadd_rn(acc, s_term[lane][j]);
        __syncthreads();
    }
    if (lane < N_STATES) a.out[size_t(ind) * N_STATES + lane] = acc;
}

struct PostArgs
{
    const double* normal_lhood; // [n][21]
    const double* tumor_lhood;
    const float* ln_sse;  // [n] (float) log(indelToRef^factor)
    const float* ln_csse; // [n] (float) log1p_switch(-indelToRef^factor)
    sk_somatic_indel_call* out;
    int n;
    SomaticDerived d;
};

// (launched 64 lanes to a block: the bound lets a lane keep its two rows of 21 likelihoods and the posterior's working set in registers -- with the
// default bound of 1 024 the compiler capped it at 128 and spilled 254 of them to 1 KB of scratch a lane)
__global__ __launch_bounds__(64) void somatic_indel_posterior_kernel(const PostArgs a)
{
    const SkLibmTables lt = sk_libm_tables_default();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    sk_somatic_indel_call res;
    float nf[30], tf[30];
#pragma unroll
    for (int j = 0; j < N_STATES; ++j) {
        res.normal_lhood[j] = a.normal_lhood[size_t(i) * N_STATES + j];
        res.tumor_lhood[j] = a.tumor_lhood[size_t(i) * N_STATES + j];
        nf[j] = static_cast<float>(res.normal_lhood[j]); // "temporary solution" float cast, somatic_indel_grid.cpp:264-270
        tf[j] = static_cast<float>(res.tumor_lhood[j]);
    }
    SomaticDerived d = a.d;
    d.ln_sse_rate = a.ln_sse[i];
    d.ln_csse_rate = a.ln_csse[i];
    calculate_result_set_grid<false>(d, lt, [&](const unsigned i) { return nf[i]; }, [&](const unsigned i) { return tf[i]; }, res);
    a.out[i] = res;
}

// --------------------------------------------------------------------------------------------------------------------

struct GroupArgs
{
    sk_allele_group_batch b;
    MapParams map;
    int min_read_bp_flank;
    double support_threshold;
    double loghalf;
    void* out; // sk_allele_group_call or sk_allele_group_call_wide records
    int exact_libm;
};

// MAXA = alternate alleles a record holds: MAXA (3) or MAXA_WIDE (8: a multi-sample group, ploidy x samples).  Rows of
// del_len / ins_len / ref_lnp / allele_lnp are MAXA wide.
// Round 6 (1.85 -> 1.08 ms per 2^18 groups, bit-identical): every allele's integrateOutMappingStatus is evaluated once for the hom
// genotypes AND the supporting-read statistics (it was evaluated twice: a third of the kernel's transcendentals); the statistics are
// counted by ballots over the chunk's lanes instead of a walk by lane 0 over the chunk (~1 000 instructions with one lane active, as
// many as the parallel part); four waves per SIMD for the narrow record (111 VGPRs; five spill and are slower: 1.36 ms).
// MAXA = 16 (SK_MAX_ALT_XWIDE: ploidy x 8 samples, 153 genotypes -- more than a wave has lanes): a lane holds the sums of genotypes
// lane, lane + 64, lane + 128; the loops over alleles and genotypes are left rolled (run-time indices into the per-lane arrays: they live in
// scratch memory -- a group this wide is a rarity of multi-sample runs, what matters is that it is computed here and not refused).
template <int MAXA, typename CallT>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(MAXA <= 3 ? 4 : (MAXA <= 8 ? 2 : 1), MAXA <= 3 ? 4 : (MAXA <= 8 ? 2 : 1)))) void allele_group_kernel(const GroupArgs a)
{
    constexpr int MAXGT = (MAXA + 1) * (MAXA + 2) / 2;
    constexpr int NACC = (MAXGT + WAVE - 1) / WAVE; // genotypes a lane sums
    constexpr int UNROLL_A = (MAXA <= 8) ? MAXA + 2 : 1; // (the narrow and the wide record: fully unrolled, as before)
    __shared__ double s_term[MAXGT][WAVE];
    const int grp = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t r0 = a.b.read_off[grp];
    const int n = int(a.b.read_off[grp + 1] - r0);
    const int n_alt = a.b.n_alt[grp];
    const int ploidy = a.b.ploidy[grp];
    const int full = n_alt + 1;
    const int gcount = (ploidy == 1) ? full : full * (full + 1) / 2;
    const unsigned flank = unsigned(a.min_read_bp_flank);
    const int ex = a.exact_libm;
    const SkLibmTables lt = sk_libm_tables_default();
    unsigned del_len[MAXA], ins_len[MAXA];
#pragma unroll UNROLL_A
    for (int k = 0; k < MAXA; ++k) {
        del_len[k] = a.b.del_len[size_t(grp) * MAXA + k];
        ins_len[k] = a.b.ins_len[size_t(grp) * MAXA + k];
    }

    // The allele-ratio priors of a het genotype (get_het_observed_allele_ratio once or twice and, for two alternate alleles, their
    // renormalisation) depend on (read length, the alleles' lengths) only: with one read length in the group -- every WGS sample -- a
    // pair per genotype and group, evaluated once by the per-read form's own operations, not per read (as I1 does).
    __shared__ double s_lp0[MAXGT], s_lp1[MAXGT];
    bool one_length = false;
    {
        unsigned lo = 0xffffffffu, hi = 0;
        for (int r = lane; r < n; r += WAVE) {
            const unsigned rl = a.b.read_length[r0 + r];
            lo = min(lo, rl);
            hi = max(hi, rl);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo = min(lo, unsigned(__shfl_xor(int(lo), d)));
            hi = max(hi, unsigned(__shfl_xor(int(hi), d)));
        }
        one_length = (n > 0 && lo == hi && ploidy != 1);
        for (int gl = lane; one_length && gl < gcount; gl += WAVE) {
            int a1 = 0;
            while ((a1 + 1) * (a1 + 2) / 2 <= gl) ++a1;
            const int a0 = gl - a1 * (a1 + 1) / 2;
            double lp0 = a.loghalf, lp1 = a.loghalf;
            if (a0 != a1) {
                unsigned d1 = 0, i1 = 0, d0 = 0, i0 = 0; // (run-time allele indices: selected, not indexed, so that the arrays stay in registers)
#pragma unroll UNROLL_A
                for (int k = 0; k < MAXA; ++k) {
                    if (k == a1 - 1) { d1 = del_len[k]; i1 = ins_len[k]; }
                    if (k == a0 - 1) { d0 = del_len[k]; i0 = ins_len[k]; }
                }
                het_observed_allele_ratio(lo, flank, d1, i1, 0.5, lp0, lp1, ex, lt);
                if (a0 > 0) {
                    double log_ref_prior = a.loghalf;
                    lp0 = a.loghalf;
                    het_observed_allele_ratio(lo, flank, d0, i0, 0.5, log_ref_prior, lp0, ex, lt);
                    const double norm = log_sum2(lp0, lp1, ex, lt);
                    lp0 = __dsub_rn(lp0, norm);
                    lp1 = __dsub_rn(lp1, norm);
                }
            }
            s_lp0[gl] = lp0;
            s_lp1[gl] = lp1;
        }
        __syncthreads();
    }

    double acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = 0.;
    unsigned cnt_f[MAXA + 2], cnt_r[MAXA + 2]; // (wave-uniform: counted by ballots over the chunk's lanes)
#pragma unroll UNROLL_A
    for (int k = 0; k < MAXA + 2; ++k) cnt_f[k] = cnt_r[k] = 0;
    unsigned used = 0;

    for (int base = 0; base < n; base += WAVE) {
        const int r = base + lane;
        const int cnt = min(WAVE, n - base);
        unsigned slot = 0xffu; // this lane's read supports allele `slot` (n_alt + 1: none with confidence); 0xff: the read is not used
        bool fwd = false;
        if (r < n) {
            const int64_t g = r0 + r;
            const unsigned flags = a.b.read_flags[g];
            // intersection of the tier1 reads scored for every allele (OrthogonalVariantAlleleCandidateGroupUtil.cpp:64-113)
            bool use = (flags & SK_READ_TIER1) != 0;
            double L[MAXA + 1];
            L[0] = 0.;
#pragma unroll UNROLL_A
            for (int k = 0; k < MAXA; ++k) {
                L[k + 1] = 0.;
                if (k < n_alt) {
                    const float s = a.b.allele_lnp[g * MAXA + k];
                    if (!(s == s)) use = false;
                    const double rl = double(a.b.ref_lnp[g * MAXA + k]);
                    L[0] = (k == 0) ? rl : ((L[0] < rl) ? rl : L[0]); // getAlleleLogLhoodFromRead :159-167
                    L[k + 1] = double(s);
                }
            }
            if (!use) {
#pragma unroll (MAXA <= 8 ? MAXGT : 1)
                for (int gi = 0; gi < MAXGT; ++gi) s_term[gi][lane] = 0.;
            } else {
                const unsigned na = a.b.non_ambig[g];
                const unsigned rlen = a.b.read_length[g];
                // integrateOutMappingStatus of every allele's own likelihood: the hom genotypes' terms (AlleleGroupGenotype.cpp:97-104) AND
                // what updateSupportingReadStats starts from (:125-131) -- the same function of the same arguments, evaluated once
                double H[MAXA + 1];
#pragma unroll UNROLL_A
                for (int k = 0; k <= MAXA; ++k) H[k] = (k < full) ? integrate_out_mapping(a.map, na, L[k], ex, lt) : 0.;
                // updateGenotypeLogLhoodFromAlleleLogLhood, AlleleGroupGenotype.cpp:36-114
                if (ploidy == 1) {
#pragma unroll UNROLL_A
                    for (int a0 = 0; a0 <= MAXA; ++a0)
                        if (a0 < full) s_term[a0][lane] = H[a0];
                } else {
#pragma unroll UNROLL_A
                    for (int a1 = 0; a1 <= MAXA; ++a1) {
#pragma unroll UNROLL_A
                        for (int a0 = 0; a0 <= a1; ++a0) {
                            if (a1 >= full) continue;
                            const int gi = a0 + (a1 * (a1 + 1) / 2);
                            double raw;
                            if (a0 != a1 && one_length) { // (wave-uniform)
                                raw = log_sum2(__dadd_rn(L[a0], s_lp0[gi]), __dadd_rn(L[a1], s_lp1[gi]), ex, lt);
                            } else if (a0 != a1) {
                                double lp0 = a.loghalf, lp1 = a.loghalf;
                                het_observed_allele_ratio(rlen, flank, del_len[a1 - 1], ins_len[a1 - 1], 0.5, lp0, lp1, ex, lt);
                                if (a0 > 0) { // het-alt: both alleles' indel ratios, renormalised (:83-95)
                                    double log_ref_prior = a.loghalf;
                                    lp0 = a.loghalf;
                                    het_observed_allele_ratio(rlen, flank, del_len[a0 - 1], ins_len[a0 - 1], 0.5, log_ref_prior, lp0, ex, lt);
                                    const double norm = log_sum2(lp0, lp1, ex, lt);
                                    lp0 = __dsub_rn(lp0, norm);
                                    lp1 = __dsub_rn(lp1, norm);
                                }
                                raw = log_sum2(__dadd_rn(L[a0], lp0), __dadd_rn(L[a1], lp1), ex, lt);
                            } else {
                                s_term[gi][lane] = H[a0];
                                continue;
                            }
                            s_term[gi][lane] = integrate_out_mapping(a.map, na, raw, ex, lt);
                        }
                    }
                }
                // updateSupportingReadStats, :125-155 (normalizeLogDistro: first maximum, exp, 1/sum)
                double Lm[MAXA + 1];
                double mx = 0.;
#pragma unroll UNROLL_A
                for (int k = 0; k <= MAXA; ++k) {
                    Lm[k] = H[k];
                    if (k < full) mx = (k == 0) ? Lm[0] : ((Lm[k] > mx) ? Lm[k] : mx);
                }
                double sum = 0.;
#pragma unroll UNROLL_A
                for (int k = 0; k <= MAXA; ++k)
                    if (k < full) {
                        Lm[k] = sk_exp(__dsub_rn(Lm[k], mx), ex, lt);
                        sum = __dadd_rn(sum, Lm[k]);
                    }
                sum = __ddiv_rn(1., sum);
                unsigned which = 0xfeu;
#pragma unroll UNROLL_A
                for (int k = MAXA; k >= 0; --k)
                    if (k < full && !(__dmul_rn(Lm[k], sum) < a.support_threshold)) which = unsigned(k); // first such allele
                slot = (which == 0xfeu) ? unsigned(n_alt + 1) : which;
                fwd = (flags & SK_READ_FWD) != 0;
            }
        }
        // LocusSupportingReadStats: the chunk's reads counted per (allele, strand) by ballots -- every lane of the wave, no serial walk
        // (counts do not depend on the order of the reads)
        used += unsigned(__popcll(__ballot(slot != 0xffu)));
#pragma unroll UNROLL_A
        for (unsigned k = 0; k < MAXA + 2; ++k) {
            const unsigned long long all = __ballot(slot == k), f = __ballot(slot == k && fwd);
            cnt_f[k] += unsigned(__popcll(f));
            cnt_r[k] += unsigned(__popcll(all)) - unsigned(__popcll(f));
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NACC; ++t) {
            const int gl = lane + t * WAVE;
            if (gl < gcount)
                for (int j = 0; j < cnt; ++j) acc[t] = __dadd_rn(acc[t], s_term[gl][j]);
        }
        __syncthreads();
    }
    CallT* o = static_cast<CallT*>(a.out) + grp;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        const int gl = lane + t * WAVE;
        if (gl < MAXGT) o->lhood[gl] = (gl < gcount) ? acc[t] : 0.;
    }
    if (lane == 0) {
#pragma unroll UNROLL_A
        for (int k = 0; k < MAXA + 2; ++k) {
            o->counts[0][k] = cnt_f[k];
            o->counts[1][k] = cnt_r[k];
        }
        o->n_genotypes = unsigned(gcount);
        o->n_reads_used = used;
    }
}

// ---- the whole of get_somatic_indel (sk_somatic_indel_call_tiers) -------------------------------------------------------

// M1: is_multi_indel_allele (somatic_indel_grid.cpp:102-177) for both tiers of every indel, one wave per indel.
//   lanes  : indel_lnp_to_pprob (L/starling_common/AlleleReportInfoUtil.cpp:220-301) of 64 reads at a time into LDS --
//            ReadPathScores::score_t is float, so every store to a pprob field rounds to float while the arithmetic in
//            between is double wherever a double operand takes part (restated literally);
//   lane 0 : get_sum_path_pprob (L/starling_common/starling_indel_call_pprob_digt.cpp:187-236) -- float sums in read order,
//            alternates accumulated per key with a key -> slot map that is local to each of the two calls (so the tumor
//            sample's alternates follow the normal sample's even where a key repeats) -- then the sort and the decisions.
struct MultiArgs
{
    sk_readscore_batch n, t;
    const int32_t* alt_key[2]; // [reads][2]
    const float* alt_lnp[2];
    const int64_t* alt_off;
    const sk_alt_allele* alt_alleles;
    double correct_mapping_log_prior;
    double random_base_match_log_prob[2]; // per tier
    double allele_lnprior[3];             // log(1./n_alleles), n_alleles = 2, 3, 4
    uint8_t* filter;  // [2][n_indels]
    uint8_t* overlap; // [2][n_indels]
    int n_indels;
    int exact_libm;
};

struct ReadPprob
{
    float ref, indel, alt[2];
    int32_t alt_key[2];
    int32_t use; // the read takes part in this tier's pass
};

constexpr int MAX_SLOTS = 2 * SK_MAX_ALT_ALLELES;

__global__ __launch_bounds__(WAVE) void multi_indel_allele_kernel(const MultiArgs a)
{
    __shared__ ReadPprob s_pp[WAVE];
    const int ind = blockIdx.x;
    const int lane = threadIdx.x;
    const SkLibmTables lt = sk_libm_tables_default();
    const int ex = a.exact_libm;
    const sk_alt_allele* alleles = a.alt_alleles + a.alt_off[ind];

    for (int tier = 0; tier < 2; ++tier) {
        float tot_ref = 0.f, tot_indel = 0.f;
        int n_slots = 0;
        int32_t slot_key[MAX_SLOTS];
        float slot_val[MAX_SLOTS];
        for (int smp = 0; smp < 2; ++smp) {
            const sk_readscore_batch& b = smp ? a.t : a.n;
            const int64_t r0 = b.read_off[ind];
            const int n = int(b.read_off[ind + 1] - r0);
            const int first_slot = n_slots;
            for (int base = 0; base < n; base += WAVE) {
                const int r = base + lane;
                if (r < n) {
                    const int64_t g = r0 + r;
                    ReadPprob pp;
                    pp.use = (tier == 1) || ((b.read_flags[g] & SK_READ_TIER1) != 0);
                    int n_in = 0;
                    if (a.alt_key[smp][2 * g] >= 0) ++n_in;
                    if (a.alt_key[smp][2 * g + 1] >= 0) ++n_in;
                    const double alp = a.allele_lnprior[n_in];
                    const double cmlp = a.correct_mapping_log_prior;
                    float incorrect = float(__dmul_rn(a.random_base_match_log_prob[tier], double(b.non_ambig[g])));
                    pp.ref = float(__dadd_rn(__dadd_rn(double(b.ref_lnp[g]), cmlp), alp));
                    pp.indel = float(__dadd_rn(__dadd_rn(double(b.indel_lnp[g]), cmlp), alp));
                    pp.alt_key[0] = pp.alt_key[1] = -1;
                    pp.alt[0] = pp.alt[1] = 0.f;
                    for (int k = 0; k < n_in; ++k) {
                        pp.alt_key[k] = a.alt_key[smp][2 * g + k];
                        pp.alt[k] = float(__dadd_rn(__dadd_rn(double(a.alt_lnp[smp][2 * g + k]), cmlp), alp));
                    }
                    const float m1 = (pp.ref < pp.indel) ? pp.indel : pp.ref;
                    double scale = double((incorrect < m1) ? m1 : incorrect);
                    for (int k = 0; k < n_in; ++k) if (scale < double(pp.alt[k])) scale = double(pp.alt[k]);
                    incorrect = float(sk_exp(__dsub_rn(double(incorrect), scale), ex, lt));
                    pp.ref = float(sk_exp(__dsub_rn(double(pp.ref), scale), ex, lt));
                    pp.indel = float(sk_exp(__dsub_rn(double(pp.indel), scale), ex, lt));
                    for (int k = 0; k < n_in; ++k) pp.alt[k] = float(sk_exp(__dsub_rn(double(pp.alt[k]), scale), ex, lt));
                    double sum = double(__fadd_rn(__fadd_rn(incorrect, pp.ref), pp.indel));
                    for (int k = 0; k < n_in; ++k) sum = __dadd_rn(sum, double(pp.alt[k]));
                    pp.ref = float(__ddiv_rn(double(pp.ref), sum));
                    pp.indel = float(__ddiv_rn(double(pp.indel), sum));
                    for (int k = 0; k < n_in; ++k) pp.alt[k] = float(__ddiv_rn(double(pp.alt[k]), sum));
                    s_pp[lane] = pp;
                }
                __syncthreads();
                if (lane == 0) {
                    const int cnt = min(WAVE, n - base);
                    for (int j = 0; j < cnt; ++j) {
                        const ReadPprob& pp = s_pp[j];
                        if (!pp.use) continue;
                        tot_indel = __fadd_rn(tot_indel, pp.indel);
                        tot_ref = __fadd_rn(tot_ref, pp.ref);
                        for (int k = 0; k < 2; ++k) {
                            if (pp.alt_key[k] < 0) break;
                            int slot = -1;
                            for (int q = first_slot; q < n_slots; ++q)
                                if (slot_key[q] == pp.alt_key[k]) { slot = q; break; }
                            if (slot < 0) {
                                if (n_slots < MAX_SLOTS) {
                                    slot_key[n_slots] = pp.alt_key[k];
                                    slot_val[n_slots] = pp.alt[k];
                                    ++n_slots;
                                }
                            } else {
                                slot_val[slot] = __fadd_rn(slot_val[slot], pp.alt[k]);
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (lane == 0) {
            // scores: (-pprob, id) with id -2 = the indel, -1 = the reference, >= 0 = alternate slot; std::sort ascending
            int n = 2 + n_slots;
            double sc[MAX_SLOTS + 2];
            int id[MAX_SLOTS + 2];
            sc[0] = -double(tot_indel); id[0] = -2;
            sc[1] = -double(tot_ref); id[1] = -1;
            for (int i = 0; i < n_slots; ++i) { sc[2 + i] = -double(slot_val[i]); id[2 + i] = i; }
            for (int i = 1; i < n; ++i) { // insertion sort on the total order of std::pair<double,int>
                const double v = sc[i];
                const int vi = id[i];
                int j = i - 1;
                while (j >= 0 && (v < sc[j] || (!(sc[j] < v) && vi < id[j]))) {
                    sc[j + 1] = sc[j];
                    id[j + 1] = id[j];
                    --j;
                }
                sc[j + 1] = v;
                id[j + 1] = vi;
            }
            while (id[0] >= 0 && id[1] >= 0) {
                const sk_alt_allele& k1 = alleles[slot_key[id[0]]];
                const sk_alt_allele& k2 = alleles[slot_key[id[1]]];
                const bool either_mm = k1.is_mismatch || k2.is_mismatch; // is_indel_conflict, indel_util.cpp:27-45
                const int e1 = k1.end_pos + (either_mm ? 0 : 1), e2 = k2.end_pos + (either_mm ? 0 : 1);
                if ((e2 > k1.begin_pos) && (k2.begin_pos < e1)) break;
                for (int i = 1; i + 1 < n; ++i) { sc[i] = sc[i + 1]; id[i] = id[i + 1]; }
                --n;
            }
            bool filtered = (id[0] != -2) && (id[1] != -2);
            if (!filtered && n >= 3) {
                const double top_prob = __dadd_rn(sc[0], sc[1]);
                const double top_frac = __ddiv_rn(top_prob, __dadd_rn(top_prob, sc[2]));
                if (top_frac < .9) filtered = true;
            }
            a.filter[tier * a.n_indels + ind] = filtered ? 1 : 0;
            a.overlap[tier * a.n_indels + ind] = (!filtered && (id[0] != -1) && (id[1] != -1)) ? 1 : 0;
        }
        __syncthreads();
    }
}

// M2: skip rules, tier selection and NTYPE conflict (somatic_indel_grid.cpp:220-232, 239-254, 289-360)
struct IndelCombineArgs
{
    const sk_somatic_indel_call* call[2]; // per tier: likelihoods + posterior of every indel
    const uint8_t* filter;
    const uint8_t* overlap;
    const uint8_t* forced;
    int use_tier2;
    int n_indels;
    sk_somatic_indel_genotype* out;
};

__global__ __launch_bounds__(256) void somatic_indel_combine_kernel(const IndelCombineArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_indels) return;
    const bool forced = (a.forced != nullptr) && (a.forced[i] != 0);
    struct Rs { uint32_t ntype, max_gt; int32_t qphred, from_ntype_qphred; bool is_overlap; } rs[2];
    for (int t = 0; t < 2; ++t) { rs[t].ntype = rs[t].max_gt = 0; rs[t].qphred = rs[t].from_ntype_qphred = 0; rs[t].is_overlap = false; }
    for (int t = 0; t < 2; ++t) {
        if (t == 1) {
            if (!a.use_tier2) continue;
            if (rs[0].qphred == 0 && !forced) continue;
        }
        const bool filtered = a.filter[t * a.n_indels + i] != 0;
        rs[t].is_overlap = a.overlap[t * a.n_indels + i] != 0;
        if (filtered && !forced) continue; // (qphred stays 0)
        const sk_somatic_indel_call& c = a.call[t][i];
        rs[t].ntype = c.ntype;
        rs[t].max_gt = c.max_gt;
        rs[t].qphred = filtered ? 0 : c.qphred;
        rs[t].from_ntype_qphred = c.from_ntype_qphred;
    }
    sk_somatic_indel_genotype g;
    memset(&g, 0, sizeof(g));
    g.is_forced_output = forced ? 1 : 0;
    if (forced || !(rs[0].qphred == 0 || rs[1].qphred == 0)) {
        uint8_t tier = 0, from_tier = 0;
        if (a.use_tier2 && rs[0].qphred > rs[1].qphred) tier = 1;
        if (a.use_tier2 && rs[0].from_ntype_qphred > rs[1].from_ntype_qphred) from_tier = 1;
        g.sindel_tier = tier;
        g.sindel_from_ntype_tier = from_tier;
        g.max_gt = rs[from_tier].max_gt;
        g.from_ntype_qphred = rs[from_tier].from_ntype_qphred;
        g.is_overlap = rs[from_tier].is_overlap ? 1 : 0;
        if (rs[0].ntype != rs[1].ntype) {
            g.ntype = 3u; // NTYPE::CONFLICT
            g.from_ntype_qphred = 0;
        } else {
            const uint32_t nt = rs[from_tier].ntype;
            g.ntype = (nt == SOM_REF) ? 0u : ((nt == SOM_HOM) ? 1u : 2u);
        }
        g.qphred = rs[tier].qphred;
    }
    a.out[i] = g;
}

double h_log1p_switch(const double x)
{
    if (std::abs(x) < 0.01) return ::log1p(x);
    return std::log(1 + x);
}

MapParams make_map(const sk_indel_options& opt, const bool tier2_pass)
{
    MapParams m;
    volatile double p = 1.7e-10;
    m.correct_mapping_log_prior = std::log(p);
    volatile double rb = tier2_pass ? opt.tier2_random_base_match_prob : opt.random_base_match_prob;
    m.random_base_match_log_prob = std::log(rb);
    return m;
}

} // namespace

template <int MAXA, typename CallT>
static int allele_group_dev_t(const sk_allele_group_batch* b, const sk_indel_options* opt, CallT* dev_out, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!b || !opt || !dev_out) return sk_fail("sk_allele_group_genotype_lhoods_dev: null argument");
    if (b->n_groups <= 0) return 0;
    GroupArgs a;
    a.b = *b;
    a.map = make_map(*opt, false); // isTier2Pass(false), AlleleGroupGenotype.cpp:46
    a.min_read_bp_flank = opt->min_read_bp_flank;
    a.support_threshold = opt->read_confident_support_threshold;
    volatile double half = 0.5;
    a.loghalf = std::log(half); // :75
    a.out = dev_out;
    a.exact_libm = sk_ctx().libm_restated ? 1 : 0;
    SK_LAUNCH((allele_group_kernel<MAXA, CallT>), dim3(b->n_groups), dim3(WAVE), 0, static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(skrt::getLastError());
    return 0;
}

template <int MAXA, typename CallT>
static int allele_group_host_t(const sk_allele_group_batch* hb, const sk_indel_options* opt, CallT* out)
{
    SK_REQUIRE_INIT();
    if (!hb || !opt || !out) return sk_fail("sk_allele_group_genotype_lhoods: null argument");
    const int n = hb->n_groups;
    if (n <= 0) return 0;
    if (hb->read_off[0] != 0) return sk_fail("allele group batch: read_off must start at 0");
    for (int g = 0; g < n; ++g) {
        if (hb->n_alt[g] < 1 || hb->n_alt[g] > MAXA) return sk_fail("allele group batch: n_alt outside 1..SK_MAX_ALT (SK_MAX_ALT_WIDE / SK_MAX_ALT_XWIDE for the wide entries)");
        if (hb->ploidy[g] != 1 && hb->ploidy[g] != 2) return sk_fail("Unexpected ploidy value"); // AlleleGroupGenotype.cpp:112
    }
    const int64_t tr = hb->read_off[n];
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    // one block in, one block out, moved by launches (SkStage, sk_common.h): the adapter calls this once per indel locus
    SkStage sg;
    const size_t in_bytes = 8 * size_t(n + 1) + 2 * size_t(n) + 2 * 4 * size_t(MAXA) * size_t(n) + 2 * 4 * size_t(MAXA) * size_t(tr) + 2 * 2 * size_t(tr) + size_t(tr);
    if (sg.begin(in_bytes, sizeof(CallT) * size_t(n), 0, 11)) return 1;
    sk_allele_group_batch d = *hb;
    hipStream_t st = ctx.stream;
    d.read_off = sg.put(hb->read_off, size_t(n + 1));
    d.n_alt = sg.put(hb->n_alt, size_t(n));
    d.ploidy = sg.put(hb->ploidy, size_t(n));
    d.del_len = sg.put(hb->del_len, size_t(n) * MAXA);
    d.ins_len = sg.put(hb->ins_len, size_t(n) * MAXA);
    d.ref_lnp = sg.put(hb->ref_lnp, size_t(tr) * MAXA);
    d.allele_lnp = sg.put(hb->allele_lnp, size_t(tr) * MAXA);
    d.non_ambig = sg.put(hb->non_ambig, size_t(tr));
    d.read_length = sg.put(hb->read_length, size_t(tr));
    d.read_flags = sg.put(hb->read_flags, size_t(tr));
    CallT* dout = sg.out<CallT>(size_t(n));
    if (sg.upload(st)) return 1;
    if (allele_group_dev_t<MAXA, CallT>(&d, opt, dout, st)) return 1;
    if (sg.download_and_wait(st)) return 1;
    sg.fetch(out, dout, size_t(n));
    return 0;
}


extern "C" {

void sk_indel_options_default(sk_indel_options* opt, int is_somatic)
{
    opt->min_read_bp_flank = 5;
    opt->random_base_match_prob = is_somatic ? 0.5 : 0.25;
    opt->tier2_random_base_match_prob = 0.25;
    opt->read_confident_support_threshold = 0.51;
    opt->is_use_alt_indel = 1;
    // The 21-state grid likelihoods in the fast form (two exp per read shared by its states, one log per state: 3x).  north_star's
    // bar for log-likelihoods is 1e-5; this form agrees with the reference's operation order to ~1e-13, every integer the path derives
    // from them (QSI, QSI_NT, NT, the tier decisions) is the reference's on 10^6 fuzzed indels (tests/test_gpu_parity.py) and the
    // somatic end-to-end outputs stay byte-identical (tests/test_bench_e2e.py, tools/fuzz/e2e_seeds.py somatic).  fast_form = 0 is the
    // exact form: bit-identical doubles ($STRELKA_AMD_INDEL_EXACT=1 in the adapter).  DESIGN.md section 5.
    opt->fast_form = 1;
}

void sk_somatic_indel_options_default(sk_somatic_indel_options* opt)
{
    opt->bindel_diploid_theta = 1e-4;
    opt->somatic_indel_rate = 1e-6;
    opt->shared_indel_error_factor = 2.2;
    opt->indel_contam_tolerance = 0.15;
}

int sk_indel_grid_lhood_dev(const sk_readscore_batch* b, const sk_indel_options* opt, int is_include_tier2,
                            double* dev_out_lhood, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!b || !opt || !dev_out_lhood) return sk_fail("sk_indel_grid_lhood_dev: null argument");
    if (b->n_indels <= 0) return 0;
    GridArgs a;
    a.b = *b;
    a.map = make_map(*opt, is_include_tier2 != 0);
    a.min_read_bp_flank = opt->min_read_bp_flank;
    a.is_include_tier2 = is_include_tier2 ? 1 : 0;
    a.is_use_alt_indel = opt->is_use_alt_indel ? 1 : 0;
    a.out = dev_out_lhood;
    a.exact_libm = sk_ctx().libm_restated ? 1 : 0;
    const float RATIO_INCREMENT = 0.5f / static_cast<float>(SK_HET_RES + 1);
    for (unsigned i = 0; i < SK_HET_RES; ++i) {
        // get_indel_het_grid_lhood :79 / get_high_low_het_ratio_lhood :88-91
        const double het_ratio((i + 1) * RATIO_INCREMENT);
        volatile double hr = het_ratio;
        const double chet_ratio(1. - hr);
        volatile double chr = chet_ratio;
        a.het_ratio[i] = hr;
        a.chet_ratio[i] = chet_ratio;
        a.log_het_ratio[i] = std::log(hr);
        a.log_chet_ratio[i] = std::log(chr);
    }
    volatile double two = 2.;
    a.loghalf = -std::log(two); // :251
    if (opt->fast_form) SK_LAUNCH(indel_grid_lhood_kernel<true>, dim3(b->n_indels), dim3(WAVE), 0, static_cast<hipStream_t>(hip_stream), a);
    else SK_LAUNCH(indel_grid_lhood_kernel<false>, dim3(b->n_indels), dim3(WAVE), 0, static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(skrt::getLastError());
    return 0;
}

static int upload_readscores(const sk_readscore_batch* hb, SkStage& sg, sk_readscore_batch& d)
{
    const int n = hb->n_indels;
    if (hb->read_off[0] != 0) return sk_fail("readscore batch: read_off must start at 0");
    const int64_t tr = hb->read_off[n];
    d = *hb;
#define UPA(field, count, optional)                                \
    if (hb->field) {                                               \
        d.field = sg.put(hb->field, count);                        \
    } else if (!(optional)) {                                      \
        return sk_fail("readscore batch: missing array " #field); \
    }
    UPA(read_off, size_t(n + 1), false);
    UPA(ref_lnp, size_t(tr), false);
    UPA(indel_lnp, size_t(tr), false);
    UPA(alt_lnp, size_t(tr), true);
    UPA(non_ambig, size_t(tr), false);
    UPA(read_length, size_t(tr), false);
    UPA(read_flags, size_t(tr), false);
    UPA(del_len, size_t(n), false);
    UPA(ins_len, size_t(n), false);
    UPA(is_breakpoint, size_t(n), true);
#undef UPA
    return 0;
}

static size_t readscore_bytes(const sk_readscore_batch* hb)
{
    const int64_t tr = hb->read_off[hb->n_indels];
    return sk_align256(8 * (hb->n_indels + 1)) + 3 * sk_align256(4 * tr) + 2 * sk_align256(2 * tr) + sk_align256(tr) +
           2 * sk_align256(4 * hb->n_indels) + sk_align256(hb->n_indels) + 16 * 256;
}


// per-indel shared error rate, host libm (somatic_indel_grid.cpp:273-275)
static int upload_shared_error_rates(const double* indel_to_ref_error_prob, const int n, const sk_somatic_indel_options& sopt,
                                     SkStage& sg, float*& dsse, float*& dcsse)
{
    // the error model hands out a few dozen distinct rates (one per repeat context), so the host libm's pow / log / log1p
    // run once per distinct rate, not once per indel
    std::vector<float> ln_sse(n), ln_csse(n);
    std::unordered_map<uint64_t, std::pair<float, float>> memo;
    for (int i = 0; i < n; ++i) {
        uint64_t key;
        std::memcpy(&key, &indel_to_ref_error_prob[i], 8);
        auto it = memo.find(key);
        if (it == memo.end()) {
            const double sharedIndelErrorRate(std::pow(indel_to_ref_error_prob[i], sopt.shared_indel_error_factor));
            it = memo.emplace(key, std::make_pair((float)std::log(sharedIndelErrorRate), (float)h_log1p_switch(-sharedIndelErrorRate))).first;
        }
        ln_sse[i] = it->second.first;
        ln_csse[i] = it->second.second;
    }
    dsse = sg.put(ln_sse.data(), size_t(n));
    dcsse = sg.put(ln_csse.data(), size_t(n));
    return 0;
}

static void fill_post_args(const sk_somatic_indel_options& sopt, const double* dnl, const double* dtl, const float* dsse,
                           const float* dcsse, sk_somatic_indel_call* dout, const int n, PostArgs& p)
{
    p.normal_lhood = dnl;
    p.tumor_lhood = dtl;
    p.ln_sse = dsse;
    p.ln_csse = dcsse;
    p.out = dout;
    p.n = n;
    std::memset(&p.d, 0, sizeof(p.d));
    // somatic_indel_caller_grid ctor, somatic_indel_grid.cpp:58-64
    p.d.contam_tolerance = (float)sopt.indel_contam_tolerance;
    p.d.exact_libm = sk_ctx().libm_restated ? 1 : 0;
    p.d.ln_som_match = h_log1p_switch(-sopt.somatic_indel_rate);
    p.d.ln_som_mismatch = std::log(sopt.somatic_indel_rate);
    p.d.lnprior[SOM_REF] = (float)h_log1p_switch(-(3. * sopt.bindel_diploid_theta) / 2.);
    p.d.lnprior[SOM_HOM] = (float)std::log(sopt.bindel_diploid_theta / 2.);
    p.d.lnprior[SOM_HET] = (float)std::log(sopt.bindel_diploid_theta);
    volatile double half = 1. / 2., pm1 = static_cast<double>(PRESTRAND - 1);
    p.d.ln_one_half = std::log(half);
    p.d.log_error_mod = -std::log(pm1);
    const float RATIO_INCREMENT = 0.5f / static_cast<float>(HET_RES + 1);
    for (int index = 0; index < PRESTRAND; ++index) {
        float f;
        if (index == SOM_REF) f = 0.f;
        else if (index == SOM_HOM) f = 1.f;
        else if (index == SOM_HET) f = 0.5f;
        else if (index < SOM_SIZE + HET_RES) f = RATIO_INCREMENT * (index - SOM_SIZE + 1);
        else f = RATIO_INCREMENT * (index - SOM_SIZE + 2);
        p.d.grid_frac[index] = f;
    }
}

int sk_indel_grid_lhood(const sk_readscore_batch* hb, const sk_indel_options* opt, int is_include_tier2, double* out_lhood)
{
    SK_REQUIRE_INIT();
    if (!hb || !opt || !out_lhood) return sk_fail("sk_indel_grid_lhood: null argument");
    if (hb->n_indels <= 0) return 0;
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    // (one block in, one block out, moved by launches: SkStage, sk_common.h)
    SkStage sg;
    const size_t out_bytes = sizeof(double) * N_STATES * size_t(hb->n_indels);
    if (sg.begin(readscore_bytes(hb), out_bytes, 0, 12)) return 1;
    sk_readscore_batch d;
    if (upload_readscores(hb, sg, d)) return 1;
    double* dout = sg.out<double>(size_t(hb->n_indels) * N_STATES);
    if (sg.upload(ctx.stream)) return 1;
    if (sk_indel_grid_lhood_dev(&d, opt, is_include_tier2, dout, ctx.stream)) return 1;
    if (sg.download_and_wait(ctx.stream)) return 1;
    sg.fetch(out_lhood, dout, size_t(hb->n_indels) * N_STATES);
    return 0;
}

int sk_somatic_indel_call_batch(const sk_readscore_batch* hn, const sk_readscore_batch* ht, const sk_indel_options* nopt,
                                const sk_indel_options* topt, const sk_somatic_indel_options* sopt,
                                const double* indel_to_ref_error_prob, int is_include_tier2, sk_somatic_indel_call* out)
{
    SK_REQUIRE_INIT();
    if (!hn || !ht || !nopt || !topt || !sopt || !indel_to_ref_error_prob || !out)
        return sk_fail("sk_somatic_indel_call_batch: null argument");
    if (hn->n_indels != ht->n_indels) return sk_fail("sk_somatic_indel_call_batch: normal/tumor n_indels differ");
    const int n = hn->n_indels;
    if (n <= 0) return 0;
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    SkStage sg;
    const size_t lh_bytes = sizeof(double) * N_STATES * size_t(n);
    if (sg.begin(readscore_bytes(hn) + readscore_bytes(ht) + 2 * 4 * size_t(n), sizeof(sk_somatic_indel_call) * size_t(n), 2 * sk_align256(lh_bytes) + 1024, 26))
        return 1;
    sk_readscore_batch dn, dt;
    if (upload_readscores(hn, sg, dn) || upload_readscores(ht, sg, dt)) return 1;
    float *dsse = nullptr, *dcsse = nullptr;
    if (upload_shared_error_rates(indel_to_ref_error_prob, n, *sopt, sg, dsse, dcsse)) return 1;
    sk_somatic_indel_call* dout = sg.out<sk_somatic_indel_call>(size_t(n));
    double* dnl = sg.ar.take<double>(size_t(n) * N_STATES);
    double* dtl = sg.ar.take<double>(size_t(n) * N_STATES);
    if (sg.upload(ctx.stream)) return 1;
    if (sk_indel_grid_lhood_dev(&dn, nopt, is_include_tier2, dnl, ctx.stream)) return 1;
    if (sk_indel_grid_lhood_dev(&dt, topt, is_include_tier2, dtl, ctx.stream)) return 1;
    PostArgs p;
    fill_post_args(*sopt, dnl, dtl, dsse, dcsse, dout, n, p);
    SK_LAUNCH(somatic_indel_posterior_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx.stream, p);
    if (sg.download_and_wait(ctx.stream)) return 1;
    sg.fetch(out, dout, size_t(n));
    return 0;
}

int sk_somatic_indel_call_tiers(const sk_somatic_indel_batch* hb, const sk_indel_options* nopt, const sk_indel_options* topt,
                                const sk_somatic_indel_options* sopt, int use_tier2_evidence, sk_somatic_indel_genotype* out)
{
    SK_REQUIRE_INIT();
    if (!hb || !nopt || !topt || !sopt || !out) return sk_fail("sk_somatic_indel_call_tiers: null argument");
    const int n = hb->n_indels;
    if (n <= 0) return 0;
    if (hb->normal.n_indels != n || hb->tumor.n_indels != n) return sk_fail("sk_somatic_indel_call_tiers: n_indels differ");
    if (!hb->normal_alt_key || !hb->normal_alt_lnp || !hb->tumor_alt_key || !hb->tumor_alt_lnp || !hb->alt_off ||
        !hb->indel_to_ref_error_prob)
        return sk_fail("sk_somatic_indel_call_tiers: missing array");
    const sk_readscore_batch* smp[2] = { &hb->normal, &hb->tumor };
    const int32_t* h_alt_key[2] = { hb->normal_alt_key, hb->tumor_alt_key };
    for (int i = 0; i < n; ++i) {
        const int64_t na = hb->alt_off[i + 1] - hb->alt_off[i];
        if (na < 0 || na > SK_MAX_ALT_ALLELES) return sk_fail("sk_somatic_indel_call_tiers: more alternate alleles at one indel than SK_MAX_ALT_ALLELES");
        for (int s = 0; s < 2; ++s)
            for (int64_t r = smp[s]->read_off[i]; r < smp[s]->read_off[i + 1]; ++r)
                for (int k = 0; k < 2; ++k) {
                    const int32_t key = h_alt_key[s][2 * r + k];
                    if (key >= na) return sk_fail("sk_somatic_indel_call_tiers: alternate-allele index out of range");
                    if (k == 1 && key >= 0 && h_alt_key[s][2 * r] < 0) return sk_fail("sk_somatic_indel_call_tiers: alt entries must be front-packed");
                }
    }
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    hipStream_t st = ctx.stream;
    // one block in, one block out, moved by launches (SkStage, sk_common.h): the adapter calls this once per indel locus
    SkStage sg;
    const size_t lh_bytes = sizeof(double) * N_STATES * size_t(n);
    const int64_t trn = hb->normal.read_off[n], trt = hb->tumor.read_off[n];
    const int64_t n_alleles = hb->alt_off[n];
    const size_t in_bytes = readscore_bytes(&hb->normal) + readscore_bytes(&hb->tumor) + 2 * 4 * size_t(n) + 2 * 8 * size_t(trn) + 2 * 8 * size_t(trt) +
                            8 * (size_t(n) + 1) + sizeof(sk_alt_allele) * size_t(n_alleles) + size_t(n);
    const size_t extra = 2 * sk_align256(lh_bytes) + 2 * sk_align256(sizeof(sk_somatic_indel_call) * n) + sk_align256(4 * size_t(n)) + 8 * 256;
    if (sg.begin(in_bytes, sizeof(sk_somatic_indel_genotype) * size_t(n), extra, 34)) return 1;
    SkArena& ar = sg.ar;
    sk_readscore_batch dn, dt;
    if (upload_readscores(&hb->normal, sg, dn) || upload_readscores(&hb->tumor, sg, dt)) return 1;
    float *dsse = nullptr, *dcsse = nullptr;
    if (upload_shared_error_rates(hb->indel_to_ref_error_prob, n, *sopt, sg, dsse, dcsse)) return 1;

    MultiArgs m;
    m.n = dn;
    m.t = dt;
    auto up = [&](const void* src, const size_t bytes) -> void* { return sg.put(static_cast<const char*>(src), bytes); };
    m.alt_key[0] = static_cast<const int32_t*>(up(hb->normal_alt_key, 8 * size_t(trn)));
    m.alt_lnp[0] = static_cast<const float*>(up(hb->normal_alt_lnp, 8 * size_t(trn)));
    m.alt_key[1] = static_cast<const int32_t*>(up(hb->tumor_alt_key, 8 * size_t(trt)));
    m.alt_lnp[1] = static_cast<const float*>(up(hb->tumor_alt_lnp, 8 * size_t(trt)));
    m.alt_off = static_cast<const int64_t*>(up(hb->alt_off, 8 * (size_t(n) + 1)));
    m.alt_alleles = static_cast<const sk_alt_allele*>(up(hb->alt_alleles, sizeof(sk_alt_allele) * size_t(n_alleles)));
    uint8_t* dforced = nullptr;
    if (hb->is_forced_output) dforced = static_cast<uint8_t*>(up(hb->is_forced_output, size_t(n)));
    sk_somatic_indel_genotype* dout = sg.out<sk_somatic_indel_genotype>(size_t(n));
    double* dnl = ar.take<double>(size_t(n) * N_STATES);
    double* dtl = ar.take<double>(size_t(n) * N_STATES);
    sk_somatic_indel_call* dcall[2] = { ar.take<sk_somatic_indel_call>(n), ar.take<sk_somatic_indel_call>(n) };
    if (sg.upload(st)) return 1;
    {
        const MapParams m0 = make_map(*topt, false), m1 = make_map(*topt, true);
        m.correct_mapping_log_prior = m0.correct_mapping_log_prior;
        m.random_base_match_log_prob[0] = m0.random_base_match_log_prob;
        m.random_base_match_log_prob[1] = m1.random_base_match_log_prob;
        for (int k = 0; k < 3; ++k) {
            volatile double prior = 1. / static_cast<double>(2 + k);
            m.allele_lnprior[k] = std::log(prior);
        }
    }
    uint8_t* dflags = ar.take<uint8_t>(4 * size_t(n));
    m.filter = dflags;
    m.overlap = dflags + 2 * size_t(n);
    m.n_indels = n;
    m.exact_libm = sk_ctx().libm_restated ? 1 : 0;
    SK_LAUNCH(multi_indel_allele_kernel, dim3(n), dim3(WAVE), 0, st, m);
    SK_HIP(skrt::getLastError());

    for (int tier = 0; tier < 2; ++tier) {
        if (tier == 1 && !use_tier2_evidence) break;
        if (sk_indel_grid_lhood_dev(&dn, nopt, tier, dnl, st)) return 1;
        if (sk_indel_grid_lhood_dev(&dt, topt, tier, dtl, st)) return 1;
        PostArgs p;
        fill_post_args(*sopt, dnl, dtl, dsse, dcsse, dcall[tier], n, p);
        SK_LAUNCH(somatic_indel_posterior_kernel, dim3((n + 63) / 64), dim3(64), 0, st, p);
        SK_HIP(skrt::getLastError());
    }
    IndelCombineArgs c;
    c.call[0] = dcall[0];
    c.call[1] = dcall[1];
    c.filter = m.filter;
    c.overlap = m.overlap;
    c.forced = dforced;
    c.use_tier2 = use_tier2_evidence ? 1 : 0;
    c.n_indels = n;
    c.out = dout;
    SK_LAUNCH(somatic_indel_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c);
    if (sg.download_and_wait(st)) return 1;
    sg.fetch(out, dout, size_t(n));
    return 0;
}

int sk_allele_group_genotype_lhoods_dev(const sk_allele_group_batch* b, const sk_indel_options* opt,
                                        sk_allele_group_call* dev_out, void* hip_stream)
{
    return allele_group_dev_t<SK_MAX_ALT, sk_allele_group_call>(b, opt, dev_out, hip_stream);
}

int sk_allele_group_genotype_lhoods_wide_dev(const sk_allele_group_batch* b, const sk_indel_options* opt,
                                             sk_allele_group_call_wide* dev_out, void* hip_stream)
{
    return allele_group_dev_t<SK_MAX_ALT_WIDE, sk_allele_group_call_wide>(b, opt, dev_out, hip_stream);
}

int sk_allele_group_genotype_lhoods_xwide_dev(const sk_allele_group_batch* b, const sk_indel_options* opt,
                                              sk_allele_group_call_xwide* dev_out, void* hip_stream)
{
    return allele_group_dev_t<SK_MAX_ALT_XWIDE, sk_allele_group_call_xwide>(b, opt, dev_out, hip_stream);
}

int sk_allele_group_genotype_lhoods_xwide(const sk_allele_group_batch* hb, const sk_indel_options* opt, sk_allele_group_call_xwide* out)
{
    return allele_group_host_t<SK_MAX_ALT_XWIDE, sk_allele_group_call_xwide>(hb, opt, out);
}

int sk_allele_group_genotype_lhoods(const sk_allele_group_batch* hb, const sk_indel_options* opt, sk_allele_group_call* out)
{
    return allele_group_host_t<SK_MAX_ALT, sk_allele_group_call>(hb, opt, out);
}

int sk_allele_group_genotype_lhoods_wide(const sk_allele_group_batch* hb, const sk_indel_options* opt, sk_allele_group_call_wide* out)
{
    return allele_group_host_t<SK_MAX_ALT_WIDE, sk_allele_group_call_wide>(hb, opt, out);
}

} // extern "C"
