// germline_fused.hip -- hot path B (germline SNV), fused: dependent error probabilities (adjust_joint_eprob,
// L/blt_common/adjust_joint_eprob.cpp:201-243) followed by the diploid genotype call (position_snp_call_pprob_digt,
// L/blt_common/position_snp_call_pprob_digt.cpp:473-539) for every locus in ONE pass over the pileup.
//
// Layout / mapping
//   * a workgroup owns LOCI_PER_BLOCK consecutive loci; their calls are one contiguous span of the CSR `calls` array,
//     which the block copies into LDS with coalesced loads (sub-batched when the span exceeds the LDS budget);
//   * one THREAD per locus walks its calls in LDS.  Per call the LDS holds 4 bytes: the packed basecall (u16) and one
//     u16 slot of the per-group sort array -- `de` itself is never stored.
//   * phase 1 (adjust_joint_eprob): per (strand, base) group the calls' indices are sorted by descending q with the
//     reference's std::sort (emulated step for step: its tie order decides which calls get the first exponents).  Only
//     the first few sorted calls of a group (until the exponent reaches min_vexp) have a de that is not a pure function
//     of q; their RANK (1..7) is written into the three spare bits of the LDS copy of the basecall.
//   * phase 2 (get_diploid_gt_lhood): calls are visited in pileup order; val[0] = logf(de)+ln(1/3) comes from a host-built
//     table unless the call carries a rank, in which case de is recomputed from (q, group exponent chain).  The ten
//     genotype sums are sequential float32 adds in pileup order, as in the reference.
//   Loci deeper than 1023 calls, groups needing more than 7 ranked calls, or spans that do not fit the LDS budget take
//   the global-memory routines of germline_common.h (same arithmetic).
//
// Roofline: HBM-bound by 2 B/call in + 144 B/locus out (+4 B/call when `de` is requested); SURVEY.md 8d prices the two
// call sites separately at 6 B/call + 121 B/locus.

#include "germline_common.h"

#include <cstdlib>

namespace
{

constexpr int LOCI_PER_BLOCK = 128;
constexpr int FUSED_THREADS = 128;
constexpr int CAP_CALLS = 5312;       // LDS budget: 4 B/call -> 22 KiB (+12 KiB of per-locus ranked-call terms) per block
constexpr int MAX_RANK = 4;           // ranked calls per group on the LDS path (defaults need <= 4: 1, .65, .4225, .2746)
constexpr int V0R_PER_LOCUS = 12;    // LDS floats per locus for val[0] of ranks 2..MAX_RANK (rank 1 has de == e_q), packed
                                     // group after group; a locus needing more takes the global pass
constexpr int MAX_PACKED_DEPTH = 1022; // index fits 10 bits beside the 6-bit q in a u16 sort key
constexpr unsigned RANK_SHIFT = 13;    // bits 13..15 of the LDS basecall copy hold the rank (bit 13 = tscf, unused here)
constexpr unsigned CALL_MASK = 0x1fffu;
constexpr uint32_t NEEDS_GLOBAL_PASS = 0xffffffffu; // sentinel in sk_digt_call::is_called between the two passes

// per-q float tables, copied into LDS once per block: the per-call loops index them with a data-dependent q, and an
// LDS read costs a fraction of a (cached) global load's latency
struct QTab
{
    float4 v[SK_NQ6]; // {v0e, v0min, v1, v2}: one ds_read_b128 per call in the likelihood loop
    float weight[SK_NQ6], eprob[SK_NQ6], depmin[SK_NQ6];
};

struct FusedArgs
{
    sk_pileup_batch b;
    const SkTables* tab;
    sk_digt_call* out;
    float* de_tmp;      // global de (output when want_de, scratch for the deep-locus path)
    uint32_t* scratch;  // global sort scratch for the deep-locus path
    int want_de;
    GermlineDerived d;
};

// ---- libstdc++ std::sort on packed u16 keys (q << 10 | idx), comp(a,b) = q(a) > q(b) ----
__device__ __forceinline__ bool kgt(const uint16_t a, const uint16_t b) { return (a >> 10) > (b >> 10); }

__device__ __forceinline__ void k_unguarded_linear_insert(uint16_t* last)
{
    const uint16_t val = *last;
    uint16_t* next = last - 1;
    while (kgt(val, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

__device__ void k_insertion_sort(uint16_t* first, uint16_t* last)
{
    if (first == last) return;
    for (uint16_t* i = first + 1; i != last; ++i) {
        if (kgt(*i, *first)) {
            const uint16_t val = *i;
            for (uint16_t* p = i; p != first; --p) *p = *(p - 1);
            *first = val;
        } else {
            k_unguarded_linear_insert(i);
        }
    }
}

__device__ void k_adjust_heap(uint16_t* first, int hole, const int len, const uint16_t value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (kgt(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && kgt(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

__device__ void k_heap_sort(uint16_t* first, uint16_t* last)
{
    const int len = int(last - first);
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            const uint16_t value = first[parent];
            k_adjust_heap(first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {
        --last;
        const uint16_t value = *last;
        *last = *first;
        k_adjust_heap(first, 0, int(last - first), value);
    }
}

constexpr int SORT_STACK = 4; // pending left parts kept per thread (LDS); deeper recursion -> caller falls back

// returns false when the pending-part stack would overflow (n in the hundreds with adversarial splits)
__device__ bool k_std_sort(uint16_t* idx, const int n, uint32_t* stack)
{
    if (n <= 16) {
        k_insertion_sort(idx, idx + n);
        return true;
    }
    int lg = 0;
    for (unsigned m = unsigned(n); m > 1; m >>= 1) ++lg;
    // entry = first | last << 10 | depth << 20   (n <= 1023, depth <= 18)
    int sp = 1;
    stack[0] = 0u | (unsigned(n) << 10) | (unsigned(lg * 2) << 20);
    while (sp > 0) {
        --sp;
        const uint32_t ent = stack[sp];
        int first = int(ent & 0x3ffu), last = int((ent >> 10) & 0x3ffu), depth = int(ent >> 20);
        while (last - first > 16) {
            if (depth == 0) {
                k_heap_sort(idx + first, idx + last);
                break;
            }
            --depth;
            uint16_t* a = idx + first + 1;
            uint16_t* b = idx + first + (last - first) / 2;
            uint16_t* c = idx + last - 1;
            uint16_t* pick;
            if (kgt(*a, *b)) {
                if (kgt(*b, *c)) pick = b;
                else if (kgt(*a, *c)) pick = c;
                else pick = a;
            } else if (kgt(*a, *c)) pick = a;
            else if (kgt(*b, *c)) pick = c;
            else pick = b;
            {
                const uint16_t t = idx[first];
                idx[first] = *pick;
                *pick = t;
            }
            uint16_t* lo = idx + first + 1;
            uint16_t* hi = idx + last;
            const uint16_t pivot = idx[first];
            for (;;) {
                while (kgt(*lo, pivot)) ++lo;
                --hi;
                while (kgt(pivot, *hi)) --hi;
                if (!(lo < hi)) break;
                const uint16_t t = *lo;
                *lo = *hi;
                *hi = t;
                ++lo;
            }
            const int cut = int(lo - idx);
            if (sp >= SORT_STACK) return false;
            stack[sp] = unsigned(first) | (unsigned(cut) << 10) | (unsigned(depth) << 20);
            ++sp;
            first = cut;
        }
    }
    k_insertion_sort(idx, idx + 16);
    for (uint16_t* i = idx + 16; i != idx + n; ++i) k_unguarded_linear_insert(i);
    return true;
}

// exponent of the call with rank r (1-based) in a group with fraction f: the chain of adjust_icalls_eprob :146-178
__device__ __forceinline__ float vexp_of_rank(const unsigned r, const float vexp_frac, const GermlineDerived& D)
{
    float vexp = 1.f;
    const float m = __fsub_rn(1.f, vexp_frac);
    for (unsigned k = 1; k < r; ++k) {
        const float next_vexp = __fmul_rn(vexp, m);
        vexp = D.is_min_vexp ? ((D.min_vexp < next_vexp) ? next_vexp : D.min_vexp) : next_vexp;
    }
    return vexp;
}

__device__ __forceinline__ float select8(const float (&f)[8], const unsigned g)
{
    float r = f[0];
#pragma unroll
    for (unsigned k = 1; k < 8; ++k) r = (g == k) ? f[k] : r;
    return r;
}

// de of one call (adjust_joint_eprob semantics) from its LDS copy
__device__ __forceinline__ float call_de(const uint16_t c, const float (&vfrac)[8], const QTab& Q,
                                         const GermlineDerived& D)
{
    const unsigned q = SKC_Q(c), rank = c >> RANK_SHIFT;
    if (!D.is_dependent_eprob || SKC_FILTER(c) || q < 3) return Q.eprob[q];
    if (rank == 0) return Q.depmin[q];
    const unsigned g = SKC_FWD(c) + 2 * SKC_BASE(c);
    return get_dependent_eprob(Q.eprob[q], vexp_of_rank(rank, select8(vfrac, g), D));
}

// val[0] = logf(de) + ln(1/3) of one call: a table value unless the call is one of the few ranked ones, whose terms
// phase 1 left in LDS (v0r).  Branch-free: lanes of a wave hit ranked calls at different loop iterations.
__device__ __forceinline__ float call_v0(const uint16_t c, const float4 qv, const float* v0r, const unsigned gbase,
                                         const GermlineDerived& D)
{
    const unsigned q = SKC_Q(c), rank = c >> RANK_SHIFT;
    const bool raw = (!D.is_dependent_eprob) || SKC_FILTER(c) || q < 3 || rank == 1; // de == (float)error_prob(q)
    const unsigned g = SKC_FWD(c) + 2 * SKC_BASE(c);
    const unsigned slot = (rank >= 2) ? (((gbase >> (4 * g)) & 15u) + rank - 2) : 0u; // gbase: 4-bit slot base per group
    const float ranked = v0r[slot];
    const float tab = raw ? qv.x : qv.y;
    return (rank >= 2 && !raw) ? ranked : tab;
}

// phase 1 for one locus in LDS.  Returns false when a group needs more than MAX_RANK ranked calls (caller falls back).
__device__ bool locus_rank_calls(uint16_t* calls, uint16_t* keys, const int n, const SkTables* __restrict__ T,
                                 const GermlineDerived& D, const QTab& Q, float (&vfrac)[8], float* v0r,
                                 unsigned& gbase, uint32_t* sort_stack)
{
    gbase = 0;
    unsigned nslots = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) vfrac[g] = 0.f;
    if (!D.is_dependent_eprob) return true;
    // which (strand, base) groups are present
    unsigned present = 0;
    for (int i = 0; i < n; ++i) {
        const uint16_t b = calls[i];
        if (SKC_FILTER(b) || SKC_Q(b) < 3) continue;
        present |= 1u << (SKC_FWD(b) + 2 * SKC_BASE(b));
    }
    bool ok = true;
    // lanes walk their k-th present group together (groups are independent; ascending order as in the reference)
    while (present) {
        const unsigned g = __builtin_ctz(present);
        present &= present - 1;
        int gs = 0;
        float num = 0.f, den = 0.f;
        for (int i = 0; i < n; ++i) {
            const uint16_t b = calls[i];
            if (SKC_FILTER(b) || SKC_Q(b) < 3) continue;
            if (SKC_FWD(b) + 2 * SKC_BASE(b) != g) continue;
            keys[gs++] = uint16_t((SKC_Q(b) << 10) | unsigned(i));
            const float weight = Q.weight[SKC_Q(b)];
            den = __fadd_rn(den, weight);
            if (SKC_NMM(b)) num = __fadd_rn(num, weight);
        }
        float mismatch_frac = 0.f;
        if (den > 0.) mismatch_frac = __fdiv_rn(num, den);
        const float vexp_frac = static_cast<float>(
            __dadd_rn(__dmul_rn(static_cast<double>(__fsub_rn(1.f, mismatch_frac)), D.ssd_no_mismatch),
                      __dmul_rn(static_cast<double>(mismatch_frac), D.ssd_one_mismatch)));
#pragma unroll
        for (unsigned k = 0; k < 8; ++k)
            if (k == g) vfrac[k] = vexp_frac;

        if (!k_std_sort(keys, gs, sort_stack)) {
            ok = false;
            break;
        }

        bool is_min = false;
        float vexp = 1.f;
        const float m = __fsub_rn(1.f, vexp_frac);
        gbase |= nslots << (4 * g);
        for (int i = 0; i < gs && !is_min; ++i) {
            if (i >= MAX_RANK || (i >= 1 && nslots >= unsigned(V0R_PER_LOCUS))) {
                ok = false;
                break;
            }
            const unsigned ci = keys[i] & 0x3ffu;
            const uint16_t c = calls[ci];
            calls[ci] = uint16_t(c | ((unsigned(i) + 1u) << RANK_SHIFT));
            if (i >= 1) { // rank 1 has vexp == 1 -> de == e_q exactly, a table term
                const float de = get_dependent_eprob(Q.eprob[SKC_Q(c)], vexp);
                v0r[nslots++] = __fadd_rn(logf_via_double(de), T->g_log_one_third);
            }
            const float next_vexp = __fmul_rn(vexp, m);
            if (D.is_min_vexp) {
                is_min = (next_vexp <= D.min_vexp);
                vexp = (D.min_vexp < next_vexp) ? next_vexp : D.min_vexp;
            } else {
                vexp = next_vexp;
            }
        }
    }
    return ok;
}

// phase 2 for one locus in LDS
__device__ void locus_call_lds(const uint16_t* calls, const int n, const unsigned ref, const int ploidy,
                               const float* v0r, const unsigned gbase, const SkTables* __restrict__ T,
                               const GermlineDerived& D, const QTab& Q, sk_digt_call& res)
{
    memset(&res, 0, sizeof(res));
    if (ref >= 4) return;
    res.is_called = 1;
    res.ref_gt = ref;
    const bool is_haploid = (ploidy == 1);

    float lh[10];
#pragma unroll
    for (int gt = 0; gt < 10; ++gt) lh[gt] = 0.f;
    for (int i = 0; i < n; ++i) {
        const uint16_t bc = calls[i];
        const unsigned q = SKC_Q(bc), obs = SKC_BASE(bc);
        const float4 qv = Q.v[q];
        const float v0 = call_v0(bc, qv, v0r, gbase, D);
        const float v1 = qv.z;
        const float v2 = qv.w;
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) lh[gt] = __fadd_rn(lh[gt], (obs == unsigned(gt)) ? v2 : v0);
#pragma unroll
        for (int gt = 4; gt < 10; ++gt) lh[gt] = __fadd_rn(lh[gt], (obs == digt_a0(gt) || obs == digt_a1(gt)) ? v1 : v0);
    }
#pragma unroll
    for (int gt = 0; gt < 10; ++gt) res.lhood[gt] = lh[gt];
    {
        const int gtcount = is_haploid ? 4 : 10;
        float best = lh[0]; // lhood[maxIndex]: first maximum, strict > as in the reference
#pragma unroll
        for (int gt = 1; gt < 10; ++gt)
            if (gt < gtcount && lh[gt] > best) best = lh[gt];
#pragma unroll
        for (int gt = 0; gt < 10; ++gt)
            res.phredLoghood[gt] = (gt < gtcount) ? unsigned(ln_error_prob_to_qphred_f(__fsub_rn(lh[gt], best), D.ln10f)) : 0u;
    }
    calculate_result_set(lh, D.lnprior[is_haploid ? 1 : 0][ref][0], ref, res.genome);
    calculate_result_set(lh, D.lnprior[is_haploid ? 1 : 0][ref][1], ref, res.poly);

    if (res.genome.snp_qphred != 0) {
        const unsigned tgt = res.genome.max_gt;
        const unsigned t0 = digt_a0(tgt), t1 = digt_a1(tgt);
        float lf = 0.f, lr = 0.f;
        for (int i = 0; i < n; ++i) {
            const uint16_t bc = calls[i];
            const unsigned q = SKC_Q(bc), obs = SKC_BASE(bc);
            const float4 qv = Q.v[q];
            const float v0 = call_v0(bc, qv, v0r, gbase, D);
            const float v1 = qv.z;
            const float v2 = qv.w;
            const float val_ref = (obs == ref) ? v2 : v0;
            const float val_tgt = (tgt < 4) ? ((obs == tgt) ? v2 : v0) : ((obs == t0 || obs == t1) ? v1 : v0);
            const bool fwd = SKC_FWD(bc);
            lf = __fadd_rn(lf, fwd ? val_tgt : val_ref);
            lr = __fadd_rn(lr, fwd ? val_ref : val_tgt);
        }
        const float m = (lf < lr) ? lr : lf;
        float lht = lh[0];
#pragma unroll
        for (int gt = 1; gt < 10; ++gt) lht = (tgt == unsigned(gt)) ? lh[gt] : lht;
        res.strand_bias = static_cast<double>(__fsub_rn(m, lht));
    }
}

__global__ __launch_bounds__(FUSED_THREADS) void germline_site_fused_kernel(const FusedArgs a)
{
    __shared__ uint16_t s_calls[CAP_CALLS];
    __shared__ uint16_t s_keys[CAP_CALLS];
    __shared__ int64_t s_off[LOCI_PER_BLOCK + 1];
    __shared__ float s_v0r[LOCI_PER_BLOCK * V0R_PER_LOCUS];
    __shared__ uint32_t s_stack[FUSED_THREADS * SORT_STACK];
    __shared__ QTab s_q;

    const int tid = threadIdx.x;
    const int l0 = blockIdx.x * LOCI_PER_BLOCK;
    const int nl = min(LOCI_PER_BLOCK, a.b.n_loci - l0);
    for (int j = tid; j <= nl; j += FUSED_THREADS) s_off[j] = a.b.call_off[l0 + j];
    for (int q = tid; q < SK_NQ6; q += FUSED_THREADS) {
        s_q.v[q] = make_float4(a.d.v0e[q], a.d.v0min[q], a.tab->g_v1[q], a.tab->g_v2[q]);
        s_q.weight[q] = a.tab->g_weight[q];
        s_q.eprob[q] = a.tab->g_eprob[q];
        s_q.depmin[q] = a.d.depmin[q];
    }
    __syncthreads();

    const SkTables* __restrict__ T = a.tab;
    int s = 0; // first locus (block-relative) of the current sub-batch
    while (s < nl) {
        // sub-batch = the longest run of loci starting at s whose calls fit the LDS budget (call_off is monotone)
        const int64_t c0 = s_off[s];
        const bool fits = (tid >= s) && (tid < nl) && (s_off[tid + 1] - c0 <= CAP_CALLS);
        const int cnt = __syncthreads_count(fits);
        if (cnt == 0) {
            // a single locus deeper than the LDS budget: left to the global-memory pass
            if (tid == 0) a.out[l0 + s].is_called = NEEDS_GLOBAL_PASS;
            s += 1;
            __syncthreads();
            continue;
        }
        const int e = s + cnt;
        const int span = int(s_off[e] - c0);
        const uint16_t* __restrict__ gcalls = a.b.calls + c0;
        for (int j = tid; j < span; j += FUSED_THREADS) s_calls[j] = gcalls[j] & CALL_MASK;
        __syncthreads();

        const int t = s + tid;
        if (t < e) {
            const int l = l0 + t;
            const int off = int(s_off[t] - c0);
            const int n = int(s_off[t + 1] - s_off[t]);
            const unsigned ref = a.b.ref_base[l];
            const int ploidy = a.b.ploidy ? int(a.b.ploidy[l]) : 2;
            float vfrac[8];
            bool ok = (n <= MAX_PACKED_DEPTH);
            float* v0r = s_v0r + tid * V0R_PER_LOCUS;
            unsigned gbase = 0;
            if (ok) ok = locus_rank_calls(s_calls + off, s_keys + off, n, T, a.d, s_q, vfrac, v0r, gbase, s_stack + tid * SORT_STACK);
            if (ok) {
                sk_digt_call res;
                locus_call_lds(s_calls + off, n, ref, ploidy, v0r, gbase, T, a.d, s_q, res);
                a.out[l] = res;
                if (a.want_de) {
                    float* __restrict__ de = a.de_tmp + s_off[t];
                    for (int i = 0; i < n; ++i) de[i] = call_de(s_calls[off + i], vfrac, s_q, a.d);
                }
            } else {
                a.out[l].is_called = NEEDS_GLOBAL_PASS;
            }
        }
        s = e;
        __syncthreads();
    }
}

// second pass: the few loci the LDS kernel declined (deeper than 1022 calls / the LDS budget, or needing more ranked
// calls / sort stack than the fast path holds) through the global-memory routines -- same arithmetic
__global__ void germline_site_global_pass_kernel(const FusedArgs a)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.b.n_loci) return;
    if (a.out[l].is_called != NEEDS_GLOBAL_PASS) return;
    locus_dependent_eprob_global(a.b, a.tab, a.d, a.de_tmp, a.scratch, l);
    locus_site_digt_call_global(a.b, a.de_tmp, a.tab, a.d, a.out, l);
}

} // namespace

int sk_upload_pileup_internal(const sk_pileup_batch* hb, bool need_de, SkArena& ar, size_t extra_bytes,
                              sk_pileup_batch& d, hipStream_t st, int64_t& total_calls);

extern "C" {

int sk_site_digt_call_fused_dev(const sk_pileup_batch* b, const sk_germline_options* opt, sk_digt_call* dev_out,
                                float* dev_de_tmp, int want_de, void* dev_scratch, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!b || !opt || !dev_out || !dev_de_tmp || !dev_scratch) return sk_fail("sk_site_digt_call_fused_dev: null argument");
    if (b->n_loci <= 0) return 0;
    FusedArgs a;
    a.b = *b;
    a.tab = sk_ctx().dev_tables;
    a.out = dev_out;
    a.de_tmp = dev_de_tmp;
    a.scratch = static_cast<uint32_t*>(dev_scratch);
    a.want_de = want_de ? 1 : 0;
    derive(*opt, a.d);
    const int blocks = (b->n_loci + LOCI_PER_BLOCK - 1) / LOCI_PER_BLOCK;
    hipLaunchKernelGGL(germline_site_fused_kernel, dim3(blocks), dim3(FUSED_THREADS), 0,
                       static_cast<hipStream_t>(hip_stream), a);
    if (!getenv("SK_DEBUG_SKIP_GLOBAL_PASS")) // debugging aid: leaves the sentinel visible in is_called
        hipLaunchKernelGGL(germline_site_global_pass_kernel, dim3((b->n_loci + 255) / 256), dim3(256), 0,
                           static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(hipGetLastError());
    return 0;
}

int sk_site_digt_call_fused(const sk_pileup_batch* hb, const sk_germline_options* opt, sk_digt_call* out, float* out_de)
{
    SK_REQUIRE_INIT();
    if (!hb || !opt || !out) return sk_fail("sk_site_digt_call_fused: null argument");
    if (hb->n_loci <= 0) return 0;
    SkContext& ctx = sk_ctx();
    SK_HIP(hipSetDevice(ctx.device));
    const int64_t tc = hb->call_off[hb->n_loci];
    for (int64_t i = 0; i < tc; ++i)
        if (SKC_BASE(hb->calls[i]) > 3) return sk_fail("sk_site_digt_call_fused: basecall with base_id > 3 in cleaned pileup");
    SkArena ar;
    sk_pileup_batch d;
    int64_t total = 0;
    if (sk_upload_pileup_internal(hb, false, ar, sk_align256(sizeof(sk_digt_call) * hb->n_loci) + 2 * sk_align256(4 * tc) + 1024,
                                  d, ctx.stream, total))
        return 1;
    sk_digt_call* dout = ar.take<sk_digt_call>(hb->n_loci);
    float* dde = ar.take<float>(total);
    uint32_t* scratch = ar.take<uint32_t>(total);
    if (sk_site_digt_call_fused_dev(&d, opt, dout, dde, out_de ? 1 : 0, scratch, ctx.stream)) return 1;
    SK_HIP(hipMemcpyAsync(out, dout, sizeof(sk_digt_call) * hb->n_loci, hipMemcpyDeviceToHost, ctx.stream));
    if (out_de && total) SK_HIP(hipMemcpyAsync(out_de, dde, 4 * total, hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(hipStreamSynchronize(ctx.stream));
    return 0;
}

} // extern "C"
