// somatic_site.hip -- hot path B (somatic SNV): per-locus 30-state frequency-grid likelihoods of the normal and tumor
// pileups and the 3x2 (normal genotype x {non-somatic, somatic}) posterior.
//
//   position_somatic_snv_call       L/applications/strelka/position_somatic_snv_strand_grid.cpp:230-363 (one tier)
//   get_diploid_gt_lhood_cached_simple / get_diploid_het_grid_lhood_cached / get_strand_ratio_lhood_spi
//                                   L/applications/strelka/position_somatic_snv_strand_grid_lhood_cached.cpp:41-234
//   calculate_result_set_grid       L/applications/strelka/qscore_calculator.cpp:47-209
//
// One thread per locus; the calls of a block's loci are staged through LDS with coalesced loads (normal sample, then
// tumor, through the same buffer) and the memo tables live in LDS as one row per q-score.  The reference memoises every per-call term by (qscore, ratio index); those memo tables are
// built on the host (SkTables) so each of the 21 (+2x9 strand) accumulators is the same sequential float32 sum over the
// calls in pileup order as in the reference -- bit-identical.  Only the 9 strand states' final float logsum and the
// double-precision posterior evaluate device transcendentals.

#include "somatic_common.h"

int sk_upload_pileup_internal(const sk_pileup_batch* hb, bool need_de, SkArena& ar, size_t extra_bytes,
                              sk_pileup_batch& d, hipStream_t st, int64_t& total_calls);

namespace
{

struct SomArgs
{
    sk_pileup_batch n, t;
    const SkTables* tab;
    sk_somatic_snv_call* out;
    SomaticDerived d;
};

// getLogSum<float>, L/blt_util/logSumUtil.hh:33-41 with log1p_switch<float>, L/blt_util/math_util.hh:33-48
__device__ __forceinline__ float log_sum2f(float x1, float x2)
{
    if (x1 < x2) {
        const float t = x1;
        x1 = x2;
        x2 = t;
    }
    // expf/log1pf/logf of glibc are evaluated in double and rounded once; so are these
    const float e = static_cast<float>(exp(static_cast<double>(__fsub_rn(x2, x1))));
    const float l = (fabsf(e) < 0.01f) ? static_cast<float>(log1p(static_cast<double>(e)))
                                       : static_cast<float>(log(static_cast<double>(__fadd_rn(1.f, e))));
    return __fadd_rn(x1, l);
}

// per-q row of every memoised term, one LDS row per q-score: the per-call loop indexes it with a data-dependent q
struct QRow
{
    float v0, v1, v2, off_ref;            // simple genotypes (:56-64); off-strand ref term (:213)
    float c0[HET_RES], c1[HET_RES];        // het grid (:104-110)
    float t0[HET_RES], t1[HET_RES];        // strand states, on-strand (:197-206)
    float off_alt, pad[3];                 // off-strand alt term (:221)
};
static_assert(sizeof(QRow) % 16 == 0, "rows are read with 128-bit LDS loads");

constexpr int SOM_THREADS = 64;
constexpr int SOM_CAP = 8192; // calls of one sample staged per sub-batch (16 KiB)

// one sample's 30 likelihoods for the locus whose calls sit at calls[0..n) in LDS
template <bool WITH_STRAND>
__device__ __forceinline__ void sample_lhood(const uint16_t* calls, const int n, const unsigned ref_gt, const QRow* Q,
                                             const float ln_one_half, float* __restrict__ lhood, bool& allref, unsigned& alt_id)
{
    float acc[PRESTRAND];
#pragma unroll
    for (int i = 0; i < PRESTRAND; ++i) acc[i] = 0.f;
    float sf[HET_RES], sr[HET_RES];
#pragma unroll
    for (int r = 0; r < HET_RES; ++r) sf[r] = sr[r] = 0.f;
    unsigned alt_count[4] = { 0, 0, 0, 0 };
    allref = true;

    for (int i = 0; i < n; ++i) {
        const uint16_t bc = calls[i];
        const unsigned q = SKC_Q(bc), obs = SKC_BASE(bc);
        const bool is_ref = (obs == ref_gt);
        if (!is_ref) {
            allref = false;
#pragma unroll
            for (unsigned b = 0; b < 4; ++b) alt_count[b] += (obs == b) ? 1u : 0u;
        }
        const QRow& R = Q[q];
        const float v0 = R.v0, v1 = R.v1, v2 = R.v2;
        acc[SOM_REF] = __fadd_rn(acc[SOM_REF], is_ref ? v2 : v0);
        acc[SOM_HET] = __fadd_rn(acc[SOM_HET], v1);
        acc[SOM_HOM] = __fadd_rn(acc[SOM_HOM], is_ref ? v0 : v2);
#pragma unroll
        for (int r = 0; r < HET_RES; ++r) {
            const float c0 = R.c0[r], c1 = R.c1[r];
            // lhood_high = grid[2*HET_RES-(r+1)], lhood_low = grid[r]   (…_lhood_cached.cpp:149-151)
            acc[SOM_SIZE + (2 * HET_RES - (r + 1))] = __fadd_rn(acc[SOM_SIZE + (2 * HET_RES - (r + 1))], is_ref ? c0 : c1);
            acc[SOM_SIZE + r] = __fadd_rn(acc[SOM_SIZE + r], is_ref ? c1 : c0);
        }
        if (WITH_STRAND) {
            const bool fwd = SKC_FWD(bc);
            const float off = is_ref ? R.off_ref : R.off_alt;
#pragma unroll
            for (int r = 0; r < HET_RES; ++r) {
                const float on = is_ref ? R.t0[r] : R.t1[r];
                sf[r] = __fadd_rn(sf[r], fwd ? on : off);
                sr[r] = __fadd_rn(sr[r], fwd ? off : on);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PRESTRAND; ++i) lhood[i] = acc[i];
#pragma unroll
    for (int r = 0; r < HET_RES; ++r)
        lhood[PRESTRAND + r] = WITH_STRAND ? __fadd_rn(log_sum2f(sf[r], sr[r]), ln_one_half) : 0.f;

    // snp_pos_info::get_most_frequent_alt_id, L/blt_common/snp_pos_info.hh:164-190
    alt_id = ref_gt;
    unsigned max_count = 0;
#pragma unroll
    for (unsigned b = 0; b < 4; ++b) {
        if (alt_count[b] > max_count && b != ref_gt) {
            max_count = alt_count[b];
            alt_id = b;
        }
    }
}

// stage the calls of the block's loci [s, e) of one sample into LDS with coalesced loads; sub-batched like the germline
// kernel when the span exceeds the buffer.  `f(t, calls, n)` runs for every locus t of the block exactly once; a locus
// deeper than the whole buffer is handed its global pointer instead.
template <typename F>
__device__ __forceinline__ void for_each_locus_staged(const sk_pileup_batch& b, const int l0, const int nl, uint16_t* s_calls,
                                                      int64_t* s_off, F&& f)
{
    const int tid = threadIdx.x;
    for (int j = tid; j <= nl; j += SOM_THREADS) s_off[j] = b.call_off[l0 + j];
    __syncthreads();
    int s = 0;
    while (s < nl) {
        const int64_t c0 = s_off[s];
        const bool fits = (tid >= s) && (tid < nl) && (s_off[tid + 1] - c0 <= SOM_CAP);
        const int cnt = __syncthreads_count(fits);
        if (cnt == 0) { // one locus deeper than the buffer: straight from global memory
            if (tid == s) f(s, b.calls + c0, int(s_off[s + 1] - c0));
            s += 1;
            __syncthreads();
            continue;
        }
        const int e = s + cnt;
        const int span = int(s_off[e] - c0);
        const uint16_t* __restrict__ g = b.calls + c0;
        for (int j = tid; j < span; j += SOM_THREADS) s_calls[j] = g[j];
        __syncthreads();
        const int t = s + tid;
        if (t < e) f(t, s_calls + int(s_off[t] - c0), int(s_off[t + 1] - s_off[t]));
        s = e;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SOM_THREADS) void somatic_snv_kernel(const SomArgs a)
{
    __shared__ uint16_t s_calls[SOM_CAP];
    __shared__ int64_t s_off[SOM_THREADS + 1];
    __shared__ __attribute__((aligned(16))) QRow s_q[SK_NQ6];

    const int tid = threadIdx.x;
    const int l0 = blockIdx.x * SOM_THREADS;
    const int nl = min(SOM_THREADS, a.n.n_loci - l0);
    for (int q = tid; q < SK_NQ6; q += SOM_THREADS) {
        const SkTables* __restrict__ T = a.tab;
        QRow r;
        r.v0 = T->s_v0[q];
        r.v1 = T->s_v1[q];
        r.v2 = T->s_v2[q];
        r.off_ref = T->t_off_ref[q];
        r.off_alt = T->t_off_alt[q];
        r.pad[0] = r.pad[1] = r.pad[2] = 0.f;
#pragma unroll
        for (int k = 0; k < HET_RES; ++k) {
            r.c0[k] = T->s_c0[k][q];
            r.c1[k] = T->s_c1[k][q];
            r.t0[k] = T->t_c0[k][q];
            r.t1[k] = T->t_c1[k][q];
        }
        s_q[q] = r;
    }
    const float ln_one_half = a.tab->s_ln_one_half;
    const int l = l0 + tid;
    const unsigned ref = (tid < nl) ? a.n.ref_base[l] : 4u;

    sk_somatic_snv_call res;
    memset(&res, 0, sizeof(res));
    bool n_allref = true, t_allref = true;
    // (the first __syncthreads inside for_each_locus_staged also publishes s_q)
    for_each_locus_staged(a.n, l0, nl, s_calls, s_off, [&](const int, const uint16_t* calls, const int n) {
        if (ref < 4) sample_lhood<false>(calls, n, ref, s_q, ln_one_half, res.normal_lhood, n_allref, res.normal_alt_id);
    });
    for_each_locus_staged(a.t, l0, nl, s_calls, s_off, [&](const int, const uint16_t* calls, const int n) {
        if (ref < 4) sample_lhood<true>(calls, n, ref, s_q, ln_one_half, res.tumor_lhood, t_allref, res.tumor_alt_id);
    });
    if (tid >= nl) return;
    if (ref >= 4 || (!a.d.is_forced_output && n_allref && t_allref)) { // N reference / early-out (:251-254)
        memset(&res, 0, sizeof(res));
        a.out[l] = res;
        return;
    }
    res.is_called = 1;
    calculate_result_set_grid(a.d, res.normal_lhood, res.tumor_lhood, res);
    if (a.d.is_forced_output || res.qphred != 0) { // strand bias (:216-225), skipped by the early return at :184
        float symm = res.tumor_lhood[SOM_SIZE];
        for (int i = SOM_SIZE; i < PRESTRAND; ++i) symm = (symm < res.tumor_lhood[i]) ? res.tumor_lhood[i] : symm;
        float strand = res.tumor_lhood[PRESTRAND];
        for (int i = PRESTRAND; i < GRID; ++i) strand = (strand < res.tumor_lhood[i]) ? res.tumor_lhood[i] : strand;
        const float dd = __fsub_rn(strand, symm);
        res.strand_bias = (0.f < dd) ? dd : 0.f;
    }
    a.out[l] = res;
}

double log1p_switch(const double x)
{
    if (std::abs(x) < 0.01) return ::log1p(x);
    return std::log(1 + x);
}

void derive(const sk_somatic_snv_options& opt, int is_forced_output, SomaticDerived& d)
{
    std::memset(&d, 0, sizeof(d));
    // somatic_snv_caller_strand_grid ctor, position_somatic_snv_strand_grid.cpp:42-54
    d.contam_tolerance = opt.ssnv_contam_tolerance;
    d.ln_csse_rate = log1p_switch(-opt.shared_site_error_rate);
    d.ln_som_match = log1p_switch(-opt.somatic_snv_rate);
    d.ln_som_mismatch = std::log(opt.somatic_snv_rate);
    // calculateGermlineGenotypeLogPrior, qscore_calculator.cpp:33-42
    d.lnprior[SOM_REF] = (float)log1p_switch(-(3. * opt.bsnp_diploid_theta) / 2.);
    d.lnprior[SOM_HOM] = (float)std::log(opt.bsnp_diploid_theta / 2.);
    d.lnprior[SOM_HET] = (float)std::log(opt.bsnp_diploid_theta);
    const float strand_sse_rate(opt.shared_site_error_rate * opt.shared_site_error_strand_bias_fraction);
    const float nostrand_sse_rate(opt.shared_site_error_rate - strand_sse_rate);
    d.ln_sse_rate = std::log(nostrand_sse_rate);
    // qscore_calculator.cpp:59-60
    volatile double half = 1. / 2., pm1 = static_cast<double>(PRESTRAND - 1);
    d.ln_one_half = std::log(half);
    d.log_error_mod = -std::log(pm1);
    // DIGT_GRID::get_fraction_from_index, strelka_digt_states.cpp:33-41
    const float RATIO_INCREMENT = 0.5f / static_cast<float>(HET_RES + 1);
    for (int index = 0; index < PRESTRAND; ++index) {
        float f;
        if (index == SOM_REF) f = 0.f;
        else if (index == SOM_HOM) f = 1.f;
        else if (index == SOM_HET) f = 0.5f;
        else if (index < SOM_SIZE + HET_RES) f = RATIO_INCREMENT * (index - SOM_SIZE + 1);
        else f = RATIO_INCREMENT * (index - SOM_SIZE + 2);
        d.grid_frac[index] = f;
    }
    d.is_forced_output = is_forced_output ? 1 : 0;
}

} // namespace

extern "C" {

int sk_somatic_snv_call_batch_dev(const sk_pileup_batch* n, const sk_pileup_batch* t, const sk_somatic_snv_options* opt,
                                  int is_forced_output, sk_somatic_snv_call* dev_out, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!n || !t || !opt || !dev_out) return sk_fail("sk_somatic_snv_call_batch_dev: null argument");
    if (n->n_loci != t->n_loci) return sk_fail("sk_somatic_snv_call_batch_dev: normal/tumor n_loci differ");
    if (n->n_loci <= 0) return 0;
    SomArgs a;
    a.n = *n;
    a.t = *t;
    a.tab = sk_ctx().dev_tables;
    a.out = dev_out;
    derive(*opt, is_forced_output, a.d);
    const int threads = SOM_THREADS;
    hipLaunchKernelGGL(somatic_snv_kernel, dim3((n->n_loci + threads - 1) / threads), dim3(threads), 0,
                       static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(hipGetLastError());
    return 0;
}

int sk_somatic_snv_call_batch(const sk_pileup_batch* hn, const sk_pileup_batch* ht, const sk_somatic_snv_options* opt,
                              int is_forced_output, sk_somatic_snv_call* out)
{
    SK_REQUIRE_INIT();
    if (!hn || !ht || !opt || !out) return sk_fail("sk_somatic_snv_call_batch: null argument");
    if (hn->n_loci != ht->n_loci) return sk_fail("sk_somatic_snv_call_batch: normal/tumor n_loci differ");
    const int n = hn->n_loci;
    if (n <= 0) return 0;
    if (std::memcmp(hn->ref_base, ht->ref_base, n) != 0) return sk_fail("sk_somatic_snv_call_batch: ref_base differs");
    SkContext& ctx = sk_ctx();
    SK_HIP(hipSetDevice(ctx.device));
    // the two uploads and the output share one arena
    SkArena ar;
    const size_t need = 2 * (sk_align256(sizeof(int64_t) * (n + 1)) + sk_align256(n) * 2 + 16 * 256) +
                        sk_align256(2 * hn->call_off[n]) + sk_align256(2 * ht->call_off[n]) +
                        sk_align256(sizeof(sk_somatic_snv_call) * n) + 4096;
    if (ar.reserve(need)) return 1;
    auto up = [&](const sk_pileup_batch* hb, sk_pileup_batch& d) -> int {
        if (hb->call_off[0] != 0) return sk_fail("pileup batch: call_off must start at 0");
        const int64_t tc = hb->call_off[n];
        for (int64_t i = 0; i < tc; ++i)
            if (SKC_BASE(hb->calls[i]) > 3) return sk_fail("sk_somatic_snv_call_batch: basecall with base_id > 3");
        d = *hb;
        int64_t* off = ar.take<int64_t>(n + 1);
        SK_HIP(hipMemcpyAsync(off, hb->call_off, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, ctx.stream));
        d.call_off = off;
        uint16_t* calls = ar.take<uint16_t>(tc);
        if (tc) SK_HIP(hipMemcpyAsync(calls, hb->calls, 2 * tc, hipMemcpyHostToDevice, ctx.stream));
        d.calls = calls;
        d.de = nullptr;
        d.ploidy = nullptr;
        uint8_t* rb = ar.take<uint8_t>(n);
        SK_HIP(hipMemcpyAsync(rb, hb->ref_base, n, hipMemcpyHostToDevice, ctx.stream));
        d.ref_base = rb;
        return 0;
    };
    sk_pileup_batch dn, dt;
    if (up(hn, dn) || up(ht, dt)) return 1;
    sk_somatic_snv_call* dout = ar.take<sk_somatic_snv_call>(n);
    if (sk_somatic_snv_call_batch_dev(&dn, &dt, opt, is_forced_output, dout, ctx.stream)) return 1;
    SK_HIP(hipMemcpyAsync(out, dout, sizeof(sk_somatic_snv_call) * n, hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(hipStreamSynchronize(ctx.stream));
    return 0;
}

} // extern "C"
