// germline_fused.hip -- hot path B (germline SNV), fused: dependent error probabilities (adjust_joint_eprob,
// L/blt_common/adjust_joint_eprob.cpp:201-243) followed by the diploid genotype call (position_snp_call_pprob_digt,
// L/blt_common/position_snp_call_pprob_digt.cpp:473-539) for every locus in ONE pass over the pileup.
//
// Layout / mapping
//   * a workgroup owns LOCI_PER_BLOCK consecutive loci; their calls are one contiguous span of the CSR `calls` array,
//     which the block copies into LDS with coalesced loads (sub-batched when the span exceeds the LDS budget);
//   * one THREAD per locus walks its calls in LDS.  Per call the LDS holds 4 bytes: the packed basecall (u16) and one
//     u16 slot of the per-group sort array -- `de` itself is never stored.
//   * phase 1 (adjust_joint_eprob): the reference sorts every (strand, base) group by descending q with std::sort and
//     walks the sorted calls with a decaying exponent.  Only the first few sorted calls (until the exponent reaches
//     min_vexp) have a de that is not a pure function of q, so only THEIR identity and order are needed: one counting
//     and one filling pass build all groups' key lists, then per group the leading elements of the std::sort result are
//     found without sorting (k_top_ranked: the introsort partition steps on the leftmost chain + a stable top-k; the
//     tie order of libstdc++'s algorithm is reproduced exactly).  Their RANK (1..4) goes into spare bits of the LDS
//     copy of the basecall.  Groups are visited largest first so that the lanes of a wave stay in step.
//   * phase 2 (get_diploid_gt_lhood): calls are visited in pileup order; val[0] = logf(de)+ln(1/3) comes from a host-built
//     table unless the call carries a rank, in which case de is recomputed from (q, group exponent chain).  The ten
//     genotype sums are sequential float32 adds in pileup order, as in the reference.
//   Loci deeper than 511 calls, groups needing more than 4 ranked calls, or spans that do not fit the LDS budget are
//   queued on a work-list and redone by the global-memory routines of germline_common.h (same arithmetic, full sort).
//
// Roofline: HBM-bound by 2 B/call in + 144 B/locus out (+4 B/call when `de` is requested); SURVEY.md 8d prices the two
// call sites separately at 6 B/call + 121 B/locus.

#include "germline_common.h"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

#ifndef G3_LIBM_LDS
#define G3_LIBM_LDS 1
#endif
#ifndef G3_DEFAULT_VARIANT
#define G3_DEFAULT_VARIANT 1
#endif

namespace
{

constexpr int LOCI_PER_BLOCK = 128;
constexpr int FUSED_THREADS = 128;
constexpr int CAP_CALLS = 5312;       // LDS budget: 4 B/call (call + sort key) = 20.75 KiB; with the 3.5 KiB term pool, offsets and
                                      // tables a block takes 26.5 KiB: 6 blocks = 12 waves per CU
constexpr int V0R_POOL = 864;         // LDS floats per block for the ranked calls' val[0] terms, handed out exact-fit
constexpr int MAX_RANK = 4;           // ranked calls per group on the LDS path (defaults need <= 4: 1, .65, .4225, .2746)
constexpr int V0R_PER_LOCUS = 16;    // most terms one locus may hold (ranks 2..MAX_RANK of each group, packed group after group,
                                     // 4-bit slot bases); a locus needing more takes the global pass
constexpr int MAX_PACKED_DEPTH = 511;  // sort key u16 = q << 10 | neighbor-mismatch << 9 | 9-bit call index
constexpr unsigned RANK_SHIFT = 13;    // bits 13..15 of the LDS basecall copy hold the rank (bit 13 = tscf, unused here)
constexpr unsigned CALL_MASK = 0x1fffu;
constexpr uint32_t NEEDS_GLOBAL_PASS = 0xffffffffu; // is_called of a locus queued for the global-memory pass

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
    uint32_t* work_count; // number of loci left to the global-memory pass, and their indices
    uint32_t* worklist;
    int want_de;
    GermlineDerived d;
};

// ---- libstdc++ std::sort on packed u16 keys (q << 10 | idx), comp(a,b) = q(a) > q(b) ----
__device__ __forceinline__ bool kgt(const uint16_t a, const uint16_t b) { return (a >> 10) > (b >> 10); }

// The first `need` (<= MAX_RANK) elements of std::sort(keys, keys+n, q descending) WITHOUT sorting.
//
// libstdc++'s std::sort = __introsort_loop (median-of-3 pivot, unguarded partition, recursing into the RIGHT part and
// looping on the left, ranges of <= 16 left alone) followed by one insertion sort over the whole array.  That insertion
// sort is stable (strict comparisons), and after the partitioning every element of an earlier final range compares >=
// every element of a later one, so the sorted array is the concatenation of the stably sorted final ranges.  Its first
// elements therefore need only (a) the partition steps along the LEFTMOST chain of ranges -- partitioning a range never
// touches elements outside it, so the right parts can be left unpartitioned until they are reached -- and (b) a stable
// top-k selection inside each final range reached.  Both are emulated step for step (the tie order decides which calls
// get the first exponents).
// A right part is only ever entered when the left part beside it holds fewer elements than are still needed (then that
// left part is at most 3 elements, i.e. already a final range), so no stack of pending ranges is required.
// Returns false when the reference would enter its heap-sort fallback (depth limit exhausted): the caller then takes
// the full emulation in the global-memory pass.
__device__ bool k_top_ranked(uint16_t* keys, const int n, const int need, uint16_t (&res)[MAX_RANK], int& nres)
{
    nres = 0;
    int lg = 0;
    for (unsigned m = unsigned(n); m > 1; m >>= 1) ++lg;
    int depth = 2 * lg;
    int first = 0, last = n;
    // stable selection of the best `want` keys of the final range [lo, hi) appended to res; q >= 3 for every real key,
    // so a zero key never displaces one
    auto take_from = [&](const int lo, const int hi, const int want) {
        uint16_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        for (int i = lo; i < hi; ++i) {
            const uint16_t x = keys[i];
            const bool c0 = kgt(x, t0), c1 = kgt(x, t1), c2 = kgt(x, t2), c3 = kgt(x, t3);
            t3 = c2 ? t2 : (c3 ? x : t3);
            t2 = c1 ? t1 : (c2 ? x : t2);
            t1 = c0 ? t0 : (c1 ? x : t1);
            t0 = c0 ? x : t0;
        }
        const int take = min(hi - lo, want);
        const uint16_t sel[4] = { t0, t1, t2, t3 };
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < take) {
#pragma unroll
                for (int r = 0; r < MAX_RANK; ++r)
                    if (r == nres + k) res[r] = sel[k];
            }
        nres += take;
    };
    while (last - first > 16) {
        if (depth == 0) return false;
        --depth;
        uint16_t* a = keys + first + 1;
        uint16_t* b = keys + first + (last - first) / 2;
        uint16_t* c = keys + last - 1;
        uint16_t* pick;
        if (kgt(*a, *b)) {
            if (kgt(*b, *c)) pick = b;
            else if (kgt(*a, *c)) pick = c;
            else pick = a;
        } else if (kgt(*a, *c)) pick = a;
        else if (kgt(*b, *c)) pick = c;
        else pick = b;
        {
            const uint16_t t = keys[first];
            keys[first] = *pick;
            *pick = t;
        }
        uint16_t* lo = keys + first + 1;
        uint16_t* hi = keys + last;
        const uint16_t pivot = keys[first];
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
        const int cut = int(lo - keys);
        if (cut - first >= need - nres) {
            last = cut; // everything still needed lies in the left part
        } else {
            take_from(first, cut, need - nres); // a final range of < 4 elements; then on into the right part
            first = cut;
        }
    }
    if (nres < need) take_from(first, last, need - nres);
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
    return get_dependent_eprob(Q.eprob[q], vexp_of_rank(rank, select8(vfrac, g), D), D.exact_libm);
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

// phase 1 for one locus in LDS.  Returns false when the locus needs the full emulation (caller falls back).
//   pass A  count the members of the eight (strand, base) groups           (16-bit fields of two 64-bit registers)
//   pass B  write every member's sort key into its group's slice of `keys` (pileup order inside a group)
//   then per present group: mismatch fraction -> exponent chain length -> the first ranked calls -> their val[0] terms
__device__ bool locus_rank_calls(uint16_t* calls, uint16_t* keys, const int n, const SkTables* __restrict__ T,
                                 const GermlineDerived& D, const QTab& Q, float (&vfrac)[8], float* pool,
                                 unsigned* pool_ctr, float*& v0r, unsigned& gbase)
{
    gbase = 0;
    v0r = pool;
    unsigned nslots = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) vfrac[g] = 0.f;
    if (!D.is_dependent_eprob) return true;

    uint64_t cntA = 0, cntB = 0; // groups 0-3 / 4-7
    for (int i = 0; i < n; ++i) {
        const uint16_t b = calls[i];
        const bool valid = !(SKC_FILTER(b) || SKC_Q(b) < 3);
        const unsigned g = SKC_FWD(b) + 2 * SKC_BASE(b);
        const uint64_t inc = valid ? (uint64_t(1) << (16 * (g & 3))) : 0;
        cntA += (g < 4) ? inc : 0;
        cntB += (g >= 4) ? inc : 0;
    }
    // term slots: ranks 2..4 of every group, i.e. at most min(count - 1, MAX_RANK - 1) each; taken exact-fit from the block pool
    unsigned need = 0;
#pragma unroll
    for (unsigned g = 0; g < 8; ++g) {
        const unsigned cg = unsigned(((g < 4) ? cntA : cntB) >> (16 * (g & 3))) & 0xffffu;
        need += (cg > 1u) ? ((cg - 1u < unsigned(MAX_RANK - 1)) ? cg - 1u : unsigned(MAX_RANK - 1)) : 0u;
    }
    if (need > unsigned(V0R_PER_LOCUS)) return false;
    if (need) {
        const unsigned at = atomicAdd(pool_ctr, need);
        if (at + need > unsigned(V0R_POOL)) return false;
        v0r = pool + at;
    }
    // exclusive prefix sums of the eight counts, same packing (no field overflows: n <= 511)
    const uint64_t inclA = cntA + (cntA << 16) + (cntA << 32) + (cntA << 48);
    const uint64_t inclB = cntB + (cntB << 16) + (cntB << 32) + (cntB << 48);
    const uint64_t totalA = inclA >> 48;
    uint64_t posA = inclA << 16, posB = (inclB << 16) + totalA * 0x0001000100010001ull;
    for (int i = 0; i < n; ++i) {
        const uint16_t b = calls[i];
        const bool valid = !(SKC_FILTER(b) || SKC_Q(b) < 3);
        const unsigned g = SKC_FWD(b) + 2 * SKC_BASE(b);
        const unsigned sh = 16 * (g & 3);
        const unsigned off = unsigned(((g < 4) ? posA : posB) >> sh) & 0xffffu;
        if (valid) keys[off] = uint16_t((SKC_Q(b) << 10) | (SKC_NMM(b) << 9) | unsigned(i));
        const uint64_t inc = valid ? (uint64_t(1) << sh) : 0;
        posA += (g < 4) ? inc : 0;
        posB += (g >= 4) ? inc : 0;
    }
    // posA/posB now hold the END offset of every group's slice

    bool ok = true;
    // Groups are independent, so the order in which a locus visits them is free.  Lanes visit their groups LARGEST FIRST:
    // the lanes of a wave then work on groups of similar size at the same time (the two big reference-base groups,
    // then the one- or two-call error groups) instead of pairing one lane's 20-call group with another's single call.
    unsigned present = 0;
#pragma unroll
    for (unsigned g = 0; g < 8; ++g)
        if ((((g < 4) ? cntA : cntB) >> (16 * (g & 3))) & 0xffffu) present |= 1u << g;
    while (present) {
        unsigned g = 0, best = 0;
#pragma unroll
        for (unsigned k = 0; k < 8; ++k) {
            const unsigned ck = unsigned(((k < 4) ? cntA : cntB) >> (16 * (k & 3))) & 0xffffu;
            const bool take = ((present >> k) & 1u) && ck > best;
            g = take ? k : g;
            best = take ? ck : best;
        }
        present &= ~(1u << g);
        const unsigned sh = 16 * (g & 3);
        const int gs = int(((g < 4) ? cntA : cntB) >> sh) & 0xffff;
        uint16_t* gk = keys + ((int(((g < 4) ? posA : posB) >> sh) & 0xffff) - gs);

        float num = 0.f, den = 0.f; // adjust_icalls_eprob :110-127, pileup order
        for (int i = 0; i < gs; ++i) {
            const unsigned k = gk[i];
            const float weight = Q.weight[k >> 10];
            den = __fadd_rn(den, weight);
            if (k & 0x200u) num = __fadd_rn(num, weight);
        }
        float mismatch_frac = 0.f;
        if (den > 0.) mismatch_frac = __fdiv_rn(num, den);
        const float vexp_frac = static_cast<float>(
            __dadd_rn(__dmul_rn(static_cast<double>(__fsub_rn(1.f, mismatch_frac)), D.ssd_no_mismatch),
                      __dmul_rn(static_cast<double>(mismatch_frac), D.ssd_one_mismatch)));
#pragma unroll
        for (unsigned k = 0; k < 8; ++k)
            if (k == g) vfrac[k] = vexp_frac;

        // how many calls of the sorted group get an exponent above the floor (:146-178): those need their rank
        const float m = __fsub_rn(1.f, vexp_frac);
        int nrank = 0;
        {
            float vexp = 1.f;
            bool is_min = false;
            while (!is_min && nrank < gs && nrank <= MAX_RANK) {
                ++nrank;
                const float next_vexp = __fmul_rn(vexp, m);
                if (D.is_min_vexp) {
                    is_min = (next_vexp <= D.min_vexp);
                    vexp = (D.min_vexp < next_vexp) ? next_vexp : D.min_vexp;
                } else {
                    vexp = next_vexp;
                }
            }
        }
        if (nrank > MAX_RANK) {
            ok = false;
            break;
        }
        uint16_t top[MAX_RANK];
        int ntop = 0;
        if (!k_top_ranked(gk, gs, nrank, top, ntop)) {
            ok = false;
            break;
        }
        if (ntop > 1) gbase |= nslots << (4 * g); // (a group without ranks >= 2 owns no slots; nslots may be 16 here)
        if (nslots + unsigned(ntop > 1 ? ntop - 1 : 0) > need) {
            ok = false;
            break;
        }
        float vexp = 1.f;
#pragma unroll
        for (int i = 0; i < MAX_RANK; ++i) {
            if (i < ntop) {
                const unsigned ci = top[i] & 0x1ffu;
                const uint16_t c = calls[ci];
                calls[ci] = uint16_t(c | ((unsigned(i) + 1u) << RANK_SHIFT));
                if (i >= 1) { // rank 1 has vexp == 1 -> de == e_q exactly, a table term
                    const float de = get_dependent_eprob(Q.eprob[SKC_Q(c)], vexp, D.exact_libm);
                    v0r[nslots++] = __fadd_rn(logf_ref(de, D.exact_libm), T->g_log_one_third);
                }
                const float next_vexp = __fmul_rn(vexp, m);
                vexp = D.is_min_vexp ? ((D.min_vexp < next_vexp) ? next_vexp : D.min_vexp) : next_vexp;
            }
        }
    }
    return ok;
}

// phase 2 for one locus in LDS
__device__ void locus_call_lds(const uint16_t* calls, const int n, const unsigned ref, const int ploidy,
                               const float* v0r, const unsigned gbase, const SkTables* __restrict__ T,
                               const GermlineDerived& D, const QTab& Q, const SkLibmTables& lt, sk_digt_call& res)
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
    calculate_result_set(lh, D.lnprior[is_haploid ? 1 : 0][ref][0], ref, D.exact_libm, lt, res.genome);
    calculate_result_set(lh, D.lnprior[is_haploid ? 1 : 0][ref][1], ref, D.exact_libm, lt, res.poly);

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

// phase 2 for one locus in LDS, four calls read ahead (see locus_rank_calls_v2)
__device__ void locus_call_lds_v2(const uint16_t* calls, const int n, const unsigned ref, const int ploidy,
                               const float* v0r, const unsigned gbase, const SkTables* __restrict__ T,
                               const GermlineDerived& D, const QTab& Q, const SkLibmTables& lt, sk_digt_call& res)
{
    memset(&res, 0, sizeof(res));
    if (ref >= 4) return;
    res.is_called = 1;
    res.ref_gt = ref;
    const bool is_haploid = (ploidy == 1);

    float lh[10];
#pragma unroll
    for (int gt = 0; gt < 10; ++gt) lh[gt] = 0.f;
    auto add_call = [&](const uint16_t bc, const float4 qv, const float v0) {
        const unsigned obs = SKC_BASE(bc);
        const float v1 = qv.z;
        const float v2 = qv.w;
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) lh[gt] = __fadd_rn(lh[gt], (obs == unsigned(gt)) ? v2 : v0);
#pragma unroll
        for (int gt = 4; gt < 10; ++gt) lh[gt] = __fadd_rn(lh[gt], (obs == digt_a0(gt) || obs == digt_a1(gt)) ? v1 : v0);
    };
    {
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            const uint16_t b0 = calls[i], b1 = calls[i + 1], b2 = calls[i + 2], b3 = calls[i + 3];
            const float4 q0 = Q.v[SKC_Q(b0)], q1 = Q.v[SKC_Q(b1)], q2 = Q.v[SKC_Q(b2)], q3 = Q.v[SKC_Q(b3)];
            const float t0 = call_v0(b0, q0, v0r, gbase, D), t1 = call_v0(b1, q1, v0r, gbase, D), t2 = call_v0(b2, q2, v0r, gbase, D),
                        t3 = call_v0(b3, q3, v0r, gbase, D);
            add_call(b0, q0, t0);
            add_call(b1, q1, t1);
            add_call(b2, q2, t2);
            add_call(b3, q3, t3);
        }
        for (; i < n; ++i) {
            const uint16_t bc = calls[i];
            const float4 qv = Q.v[SKC_Q(bc)];
            add_call(bc, qv, call_v0(bc, qv, v0r, gbase, D));
        }
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
    calculate_result_set(lh, D.lnprior[is_haploid ? 1 : 0][ref][0], ref, D.exact_libm, lt, res.genome);
    calculate_result_set(lh, D.lnprior[is_haploid ? 1 : 0][ref][1], ref, D.exact_libm, lt, res.poly);

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

__global__ __launch_bounds__(FUSED_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void germline_site_fused_kernel(const FusedArgs a)
{
    __shared__ uint16_t s_calls[CAP_CALLS];
    __shared__ uint16_t s_keys[CAP_CALLS];
    __shared__ int32_t s_off[LOCI_PER_BLOCK + 1]; // call offsets relative to the block's first call
    __shared__ float s_pool[V0R_POOL];
    __shared__ unsigned s_pool_ctr;
    __shared__ QTab s_q;
#if G3_LIBM_LDS
    __shared__ uint64_t s_libm[512];
#endif

    const int tid = threadIdx.x;
    const int l0 = blockIdx.x * LOCI_PER_BLOCK;
    const int nl = min(LOCI_PER_BLOCK, a.b.n_loci - l0);
    const int64_t block_c0 = a.b.call_off[l0];
    const bool huge = (a.b.call_off[l0 + nl] - block_c0) > int64_t(0x7fff0000); // offsets below are 32-bit
    for (int j = tid; j <= nl; j += FUSED_THREADS) s_off[j] = huge ? 0 : int32_t(a.b.call_off[l0 + j] - block_c0);
    for (int q = tid; q < SK_NQ6; q += FUSED_THREADS) {
        s_q.v[q] = make_float4(a.d.v0e[q], a.d.v0min[q], a.tab->g_v1[q], a.tab->g_v2[q]);
        s_q.weight[q] = a.tab->g_weight[q];
        s_q.eprob[q] = a.tab->g_eprob[q];
        s_q.depmin[q] = a.d.depmin[q];
    }
#if G3_LIBM_LDS
    const SkLibmTables lt = sk_libm_tables_to_lds(s_libm, tid, FUSED_THREADS);
#else
    const SkLibmTables lt = sk_libm_tables_default();
#endif
    __syncthreads();
    if (huge) { // (a block spanning > 2^31 calls: every locus to the global-memory pass)
        if (tid < nl) {
            a.out[l0 + tid].is_called = NEEDS_GLOBAL_PASS;
            a.worklist[atomicAdd(a.work_count, 1u)] = unsigned(l0 + tid);
        }
        return;
    }

    const SkTables* __restrict__ T = a.tab;
    int s = 0; // first locus (block-relative) of the current sub-batch
    while (s < nl) {
        // sub-batch = the longest run of loci starting at s whose calls fit the LDS budget (call_off is monotone)
        const int c0 = s_off[s];
        const bool fits = (tid >= s) && (tid < nl) && (s_off[tid + 1] - c0 <= CAP_CALLS);
        if (tid == 0) s_pool_ctr = 0;
        const int cnt = __syncthreads_count(fits);
        if (cnt == 0) {
            // a single locus deeper than the LDS budget: left to the global-memory pass
            if (tid == 0) {
                a.out[l0 + s].is_called = NEEDS_GLOBAL_PASS;
                a.worklist[atomicAdd(a.work_count, 1u)] = unsigned(l0 + s);
            }
            s += 1;
            __syncthreads();
            continue;
        }
        const int e = s + cnt;
        const int span = s_off[e] - c0;
        const uint16_t* __restrict__ gcalls = a.b.calls + block_c0 + c0;
        for (int j = tid; j < span; j += FUSED_THREADS) s_calls[j] = gcalls[j] & CALL_MASK;
        __syncthreads();

        const int t = s + tid;
        if (t < e) {
            const int l = l0 + t;
            const int off = s_off[t] - c0;
            const int n = s_off[t + 1] - s_off[t];
            const unsigned ref = a.b.ref_base[l];
            const int ploidy = a.b.ploidy ? int(a.b.ploidy[l]) : 2;
            float vfrac[8];
            bool ok = (n <= MAX_PACKED_DEPTH);
            float* v0r = s_pool;
            unsigned gbase = 0;
            if (ok) ok = locus_rank_calls(s_calls + off, s_keys + off, n, T, a.d, s_q, vfrac, s_pool, &s_pool_ctr, v0r, gbase);
            if (ok) {
                sk_digt_call res;
                locus_call_lds(s_calls + off, n, ref, ploidy, v0r, gbase, T, a.d, s_q, lt, res);
                a.out[l] = res;
                if (a.want_de) {
                    float* __restrict__ de = a.de_tmp + block_c0 + s_off[t];
                    for (int i = 0; i < n; ++i) de[i] = call_de(s_calls[off + i], vfrac, s_q, a.d);
                }
            } else {
                a.out[l].is_called = NEEDS_GLOBAL_PASS;
                a.worklist[atomicAdd(a.work_count, 1u)] = unsigned(l);
            }
        }
        s = e;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// G3 v2.  Counters of the kernel above on 2^24 loci at 40x (profiles/r04_v1_g3_sq_counters.txt): it is bound by VALU issue --
// 11 900 vector instructions per wave of 64 loci, the vector ALUs busy 60-70 % of the time with three waves per SIMD -- not by LDS
// (21 % busy) or memory.  A third of those instructions are the powf + logf of the ranked calls' terms: three per big group, inside
// the per-group loop, where a wave pays for them whenever ANY of its lanes has a ranked call in that round.  Here
//   * a group none of whose calls has a neighbouring mismatch has mismatch_frac == 0 exactly, hence a constant exponent chain: its
//     ranked terms are a host-built table of (q, rank) (GermlineDerived::v0r0) -- two groups in three at human mismatch densities;
//   * the others leave (slot, q) on a block-wide list and their exponent in the slot; after a barrier the whole block works the list
//     off, one entry per lane: the transcendentals are evaluated once per entry instead of once per (wave, round, rank);
//   * every per-call loop reads four calls ahead of their use (one LDS wait per four calls instead of one or two per call);
//   * the log table of the restated double-precision libm stays in global memory (L1-resident): six blocks per CU instead of five.
// Same arithmetic, same records; loci the fast path declines take the global-memory pass as before.
template <int LOCI>
struct G3Cfg
{
    static constexpr int CAP = (CAP_CALLS * LOCI) / 128;
    static constexpr int POOL = (V0R_POOL * LOCI) / 128;
    static constexpr int PEND = POOL; // (every pooled term may be pending -- pileups made of reads carry the neighbouring-mismatch flag in runs)
    // a list entry = pool slot | q << QSHIFT: 16 bits while the pool has at most 1024 slots
    static constexpr int QSHIFT = (POOL <= 1024) ? 10 : 16;
    typedef typename std::conditional<(POOL <= 1024), uint16_t, uint32_t>::type pend_t;
};

// phase 1 of a locus, as locus_rank_calls, except that a ranked call's term is either taken from the table (group without a
// neighbouring mismatch) or left pending: its exponent in the slot, (slot | q << 16) on the block's list
template <typename PendT, int QSHIFT>
__device__ bool locus_rank_calls_v2(uint16_t* calls, uint16_t* keys, const int n, const GermlineDerived& D, const QTab& Q, const float* tab,
                                    float (&vfrac)[8], float* pool, unsigned* pool_ctr, const unsigned pool_cap, PendT* pend,
                                    unsigned* pend_ctr, const unsigned pend_cap, float*& v0r, unsigned& gbase)
{
    gbase = 0;
    v0r = pool;
    unsigned nslots = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) vfrac[g] = 0.f;
    if (!D.is_dependent_eprob) return true;

    // (the per-call loops of this kernel are latency chains -- an LDS read, a table read that depends on it, an accumulator -- and a
    // SIMD holds only two or three waves of it: four calls are read ahead of their use in every loop, so that a loop waits for LDS once
    // per four calls instead of once or twice per call)
    uint64_t cntA = 0, cntB = 0; // groups 0-3 / 4-7
    auto count_call = [&](const uint16_t b) {
        const bool valid = !(SKC_FILTER(b) || SKC_Q(b) < 3);
        const unsigned g = SKC_FWD(b) + 2 * SKC_BASE(b);
        const uint64_t inc = valid ? (uint64_t(1) << (16 * (g & 3))) : 0;
        cntA += (g < 4) ? inc : 0;
        cntB += (g >= 4) ? inc : 0;
    };
    {
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            const uint16_t b0 = calls[i], b1 = calls[i + 1], b2 = calls[i + 2], b3 = calls[i + 3];
            count_call(b0);
            count_call(b1);
            count_call(b2);
            count_call(b3);
        }
        for (; i < n; ++i) count_call(calls[i]);
    }
    unsigned need = 0;
#pragma unroll
    for (unsigned g = 0; g < 8; ++g) {
        const unsigned cg = unsigned(((g < 4) ? cntA : cntB) >> (16 * (g & 3))) & 0xffffu;
        need += (cg > 1u) ? ((cg - 1u < unsigned(MAX_RANK - 1)) ? cg - 1u : unsigned(MAX_RANK - 1)) : 0u;
    }
    if (need > unsigned(V0R_PER_LOCUS)) return false;
    if (need) {
        const unsigned at = atomicAdd(pool_ctr, need);
        if (at + need > pool_cap) return false;
        v0r = pool + at;
    }
    const uint64_t inclA = cntA + (cntA << 16) + (cntA << 32) + (cntA << 48);
    const uint64_t inclB = cntB + (cntB << 16) + (cntB << 32) + (cntB << 48);
    const uint64_t totalA = inclA >> 48;
    uint64_t posA = inclA << 16, posB = (inclB << 16) + totalA * 0x0001000100010001ull;
    auto place_call = [&](const uint16_t b, const int i) {
        const bool valid = !(SKC_FILTER(b) || SKC_Q(b) < 3);
        const unsigned g = SKC_FWD(b) + 2 * SKC_BASE(b);
        const unsigned sh = 16 * (g & 3);
        const unsigned off = unsigned(((g < 4) ? posA : posB) >> sh) & 0xffffu;
        if (valid) keys[off] = uint16_t((SKC_Q(b) << 10) | (SKC_NMM(b) << 9) | unsigned(i));
        const uint64_t inc = valid ? (uint64_t(1) << sh) : 0;
        posA += (g < 4) ? inc : 0;
        posB += (g >= 4) ? inc : 0;
    };
    {
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            const uint16_t b0 = calls[i], b1 = calls[i + 1], b2 = calls[i + 2], b3 = calls[i + 3];
            place_call(b0, i);
            place_call(b1, i + 1);
            place_call(b2, i + 2);
            place_call(b3, i + 3);
        }
        for (; i < n; ++i) place_call(calls[i], i);
    }

    bool ok = true;
    unsigned present = 0;
#pragma unroll
    for (unsigned g = 0; g < 8; ++g)
        if ((((g < 4) ? cntA : cntB) >> (16 * (g & 3))) & 0xffffu) present |= 1u << g;
    while (present) {
        unsigned g = 0, best = 0;
#pragma unroll
        for (unsigned k = 0; k < 8; ++k) {
            const unsigned ck = unsigned(((k < 4) ? cntA : cntB) >> (16 * (k & 3))) & 0xffffu;
            const bool take = ((present >> k) & 1u) && ck > best;
            g = take ? k : g;
            best = take ? ck : best;
        }
        present &= ~(1u << g);
        const unsigned sh = 16 * (g & 3);
        const int gs = int(((g < 4) ? cntA : cntB) >> sh) & 0xffff;
        uint16_t* gk = keys + ((int(((g < 4) ? posA : posB) >> sh) & 0xffff) - gs);

        float num = 0.f, den = 0.f; // adjust_icalls_eprob :110-127, pileup order
        unsigned any_nmm = 0;
        auto weigh = [&](const unsigned k, const float weight) {
            den = __fadd_rn(den, weight);
            if (k & 0x200u) num = __fadd_rn(num, weight);
            any_nmm |= k & 0x200u;
        };
        {
            int i = 0;
            for (; i + 4 <= gs; i += 4) {
                const unsigned k0 = gk[i], k1 = gk[i + 1], k2 = gk[i + 2], k3 = gk[i + 3];
                const float w0 = Q.weight[k0 >> 10], w1 = Q.weight[k1 >> 10], w2 = Q.weight[k2 >> 10], w3 = Q.weight[k3 >> 10];
                weigh(k0, w0);
                weigh(k1, w1);
                weigh(k2, w2);
                weigh(k3, w3);
            }
            for (; i < gs; ++i) {
                const unsigned k = gk[i];
                weigh(k, Q.weight[k >> 10]);
            }
        }
        float mismatch_frac = 0.f;
        if (den > 0.) mismatch_frac = __fdiv_rn(num, den);
        const float vexp_frac = static_cast<float>(
            __dadd_rn(__dmul_rn(static_cast<double>(__fsub_rn(1.f, mismatch_frac)), D.ssd_no_mismatch),
                      __dmul_rn(static_cast<double>(mismatch_frac), D.ssd_one_mismatch)));
#pragma unroll
        for (unsigned k = 0; k < 8; ++k)
            if (k == g) vfrac[k] = vexp_frac;

        const float m = __fsub_rn(1.f, vexp_frac);
        int nrank = 0;
        {
            float vexp = 1.f;
            bool is_min = false;
            while (!is_min && nrank < gs && nrank <= MAX_RANK) {
                ++nrank;
                const float next_vexp = __fmul_rn(vexp, m);
                if (D.is_min_vexp) {
                    is_min = (next_vexp <= D.min_vexp);
                    vexp = (D.min_vexp < next_vexp) ? next_vexp : D.min_vexp;
                } else {
                    vexp = next_vexp;
                }
            }
        }
        if (nrank > MAX_RANK) {
            ok = false;
            break;
        }
        uint16_t top[MAX_RANK];
        int ntop = 0;
        if (!k_top_ranked(gk, gs, nrank, top, ntop)) {
            ok = false;
            break;
        }
        if (ntop > 1) gbase |= nslots << (4 * g);
        if (nslots + unsigned(ntop > 1 ? ntop - 1 : 0) > need) {
            ok = false;
            break;
        }
        // a group whose calls carry no neighbouring mismatch has the constant chain the table was built for -- unless the weights'
        // sum is not positive (then mismatch_frac is 0 as well, :128-133)
        const bool tabled = (any_nmm == 0u);
        float vexp = 1.f;
#pragma unroll
        for (int i = 0; i < MAX_RANK; ++i) {
            if (i < ntop) {
                const unsigned ci = top[i] & 0x1ffu;
                const uint16_t c = calls[ci];
                calls[ci] = uint16_t(c | ((unsigned(i) + 1u) << RANK_SHIFT));
                if (i >= 1) { // rank 1 has vexp == 1 -> de == e_q exactly, a table term
                    const unsigned q = SKC_Q(c);
                    if (tabled) {
                        v0r[nslots] = tab[q * 3u + unsigned(i - 1)];
                    } else {
                        v0r[nslots] = vexp;
                        const unsigned at = atomicAdd(pend_ctr, 1u);
                        if (at < pend_cap) pend[at] = PendT((unsigned(v0r - pool) + nslots) | (q << QSHIFT));
                        else ok = false; // (list full: this locus takes the global-memory pass)
                    }
                    ++nslots;
                }
                const float next_vexp = __fmul_rn(vexp, m);
                vexp = D.is_min_vexp ? ((D.min_vexp < next_vexp) ? next_vexp : D.min_vexp) : next_vexp;
            }
        }
    }
    return ok;
}

template <int LOCI>
__global__ __launch_bounds__(LOCI) __attribute__((amdgpu_waves_per_eu(3, 3))) void germline_site_fused_v2_kernel(const FusedArgs a)
{
    typedef G3Cfg<LOCI> C;
    __shared__ __attribute__((aligned(16))) uint16_t s_calls[C::CAP];
    __shared__ __attribute__((aligned(16))) uint16_t s_keys[C::CAP];
    __shared__ int32_t s_off[LOCI + 1];
    __shared__ float s_pool[C::POOL];
    __shared__ typename C::pend_t s_pend[C::PEND];
    __shared__ unsigned s_pool_ctr, s_pend_ctr;
    __shared__ QTab s_q;
    __shared__ float s_tab[SK_NQ6 * 3];
    __shared__ uint64_t s_exp[256];

    const int tid = threadIdx.x;
    const int l0 = blockIdx.x * LOCI;
    const int nl = min(LOCI, a.b.n_loci - l0);
    const int64_t block_c0 = a.b.call_off[l0];
    const bool huge = (a.b.call_off[l0 + nl] - block_c0) > int64_t(0x7fff0000);
    for (int j = tid; j <= nl; j += LOCI) s_off[j] = huge ? 0 : int32_t(a.b.call_off[l0 + j] - block_c0);
    for (int q = tid; q < SK_NQ6; q += LOCI) {
        s_q.v[q] = make_float4(a.d.v0e[q], a.d.v0min[q], a.tab->g_v1[q], a.tab->g_v2[q]);
        s_q.weight[q] = a.tab->g_weight[q];
        s_q.eprob[q] = a.tab->g_eprob[q];
        s_q.depmin[q] = a.d.depmin[q];
    }
    for (int j = tid; j < SK_NQ6 * 3; j += LOCI) s_tab[j] = a.d.v0r0[j / 3][j % 3];
    {
        const uint64_t* e = sk_libm::exp_table();
        for (int j = tid; j < 256; j += LOCI) s_exp[j] = e[j];
    }
    const SkLibmTables lt = SkLibmTables{ s_exp, sk_libm::log_table() };
    __syncthreads();
    if (huge) {
        if (tid < nl) {
            a.out[l0 + tid].is_called = NEEDS_GLOBAL_PASS;
            a.worklist[atomicAdd(a.work_count, 1u)] = unsigned(l0 + tid);
        }
        return;
    }

    const SkTables* __restrict__ T = a.tab;
    int s = 0;
    while (s < nl) {
        const int c0 = s_off[s];
        const bool fits = (tid >= s) && (tid < nl) && (s_off[tid + 1] - c0 <= C::CAP);
        if (tid == 0) {
            s_pool_ctr = 0;
            s_pend_ctr = 0;
        }
        const int cnt = __syncthreads_count(fits);
        if (cnt == 0) {
            if (tid == 0) {
                a.out[l0 + s].is_called = NEEDS_GLOBAL_PASS;
                a.worklist[atomicAdd(a.work_count, 1u)] = unsigned(l0 + s);
            }
            s += 1;
            __syncthreads();
            continue;
        }
        const int e = s + cnt;
        const int span = s_off[e] - c0;
        const uint16_t* __restrict__ gcalls = a.b.calls + block_c0 + c0;
        for (int j = tid; j < span; j += LOCI) s_calls[j] = gcalls[j] & CALL_MASK;

        const int t = s + tid;
        __syncthreads();

        const bool active = (t < e);
        int off = 0, n = 0;
        float vfrac[8];
        float* v0r = s_pool;
        unsigned gbase = 0;
        bool ok = false;
        if (active) {
            off = s_off[t] - c0;
            n = s_off[t + 1] - s_off[t];
            ok = (n <= MAX_PACKED_DEPTH);
            if (ok)
                ok = locus_rank_calls_v2<typename C::pend_t, C::QSHIFT>(s_calls + off, s_keys + off, n, a.d, s_q, s_tab, vfrac, s_pool, &s_pool_ctr, unsigned(C::POOL), s_pend,
                                         &s_pend_ctr, unsigned(C::PEND), v0r, gbase);
        }
        __syncthreads();
        {
            // the pending terms, one per lane: val[0] = logf(de) + ln(1/3) with de from (e_q, exponent) -- position_snp_call_pprob_digt.cpp:352,
            // adjust_joint_eprob.cpp:58-69
            const unsigned np = min(s_pend_ctr, unsigned(C::PEND));
            for (unsigned i = unsigned(tid); i < np; i += unsigned(LOCI)) {
                const unsigned ent = s_pend[i];
                const unsigned slot = ent & ((1u << C::QSHIFT) - 1u), q = ent >> C::QSHIFT;
                const float de = get_dependent_eprob(s_q.eprob[q], s_pool[slot], a.d.exact_libm);
                s_pool[slot] = __fadd_rn(logf_ref(de, a.d.exact_libm), T->g_log_one_third);
            }
        }
        __syncthreads();
        if (active) {
            const int l = l0 + t;
            if (ok) {
                const unsigned ref = a.b.ref_base[l];
                const int ploidy = a.b.ploidy ? int(a.b.ploidy[l]) : 2;
                sk_digt_call res;
                locus_call_lds_v2(s_calls + off, n, ref, ploidy, v0r, gbase, T, a.d, s_q, lt, res);
                a.out[l] = res;
                if (a.want_de) {
                    float* __restrict__ de = a.de_tmp + block_c0 + s_off[t];
                    for (int i = 0; i < n; ++i) de[i] = call_de(s_calls[off + i], vfrac, s_q, a.d);
                }
            } else {
                a.out[l].is_called = NEEDS_GLOBAL_PASS;
                a.worklist[atomicAdd(a.work_count, 1u)] = unsigned(l);
            }
        }
        s = e;
        __syncthreads();
    }
}

// second pass: the few loci the LDS kernel declined (deeper than 1022 calls / the LDS budget, or needing more ranked
// calls / sort stack than the fast path holds) through the global-memory routines -- same arithmetic
__global__ __launch_bounds__(64) void germline_site_global_pass_kernel(const FusedArgs a) // (launched 64 lanes to a block: no register cap of 128, no spills)
{
    const unsigned n = *a.work_count;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int l = int(a.worklist[i]);
        locus_dependent_eprob_global(a.b, a.tab, a.d, a.de_tmp, a.scratch, l);
        locus_site_digt_call_global(a.b, a.de_tmp, a.tab, a.d, a.out, l);
    }
}

} // namespace

int sk_upload_pileup_internal(const sk_pileup_batch* hb, bool need_de, SkArena& ar, size_t extra_bytes,
                              sk_pileup_batch& d, hipStream_t st, int64_t& total_calls);

static int g_g3_variant_override = -1;

extern "C" {

int sk_debug_set_g3_variant(int variant)
{
    g_g3_variant_override = variant;
    return 0;
}

int sk_site_digt_call_fused_dev(const sk_pileup_batch* b, const sk_germline_options* opt, sk_digt_call* dev_out,
                                float* dev_de_tmp, int want_de, void* dev_scratch, int64_t n_calls, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!b || !opt || !dev_out || !dev_de_tmp || !dev_scratch) return sk_fail("sk_site_digt_call_fused_dev: null argument");
    if (b->n_loci <= 0) return 0;
    if (n_calls < 0) return sk_fail("sk_site_digt_call_fused_dev: negative n_calls");
    FusedArgs a;
    a.b = *b;
    a.tab = sk_ctx().dev_tables;
    a.out = dev_out;
    a.de_tmp = dev_de_tmp;
    a.scratch = static_cast<uint32_t*>(dev_scratch);
    a.work_count = a.scratch + n_calls;
    a.worklist = a.work_count + 4;
    a.want_de = want_de ? 1 : 0;
    derive(*opt, a.d);
    SK_HIP(skrt::memsetAsync(a.work_count, 0, sizeof(uint32_t), static_cast<hipStream_t>(hip_stream)));
    // $SK_G3_VARIANT = 0 (A-B runs, tests): round 1's kernel; otherwise the second statement.  Same records from both.
    static const int env_variant = []() { const char* v = std::getenv("SK_G3_VARIANT"); return (v && *v) ? std::atoi(v) : G3_DEFAULT_VARIANT; }();
    const int variant = (g_g3_variant_override >= 0) ? g_g3_variant_override : env_variant;
    const hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int n = b->n_loci;
    if (variant == 0) SK_LAUNCH(germline_site_fused_kernel, dim3((n + LOCI_PER_BLOCK - 1) / LOCI_PER_BLOCK), dim3(FUSED_THREADS), 0, st, a);
    else SK_LAUNCH((germline_site_fused_v2_kernel<128>), dim3((n + 127) / 128), dim3(128), 0, st, a);
    SK_LAUNCH(germline_site_global_pass_kernel, dim3(std::min(2048, (b->n_loci + 63) / 64)), dim3(64), 0,
                       static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(skrt::getLastError());
    return 0;
}

int sk_site_digt_call_fused(const sk_pileup_batch* hb, const sk_germline_options* opt, sk_digt_call* out, float* out_de)
{
    SK_REQUIRE_INIT();
    if (!hb || !opt || !out) return sk_fail("sk_site_digt_call_fused: null argument");
    if (hb->n_loci <= 0) return 0;
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    const int64_t tc = hb->call_off[hb->n_loci];
    for (int64_t i = 0; i < tc; ++i)
        if (SKC_BASE(hb->calls[i]) > 3) return sk_fail("sk_site_digt_call_fused: basecall with base_id > 3 in cleaned pileup");
    SkArena ar;
    sk_pileup_batch d;
    int64_t total = 0;
    if (sk_upload_pileup_internal(hb, false, ar, sk_align256(sizeof(sk_digt_call) * hb->n_loci) + 2 * sk_align256(4 * tc) + sk_align256(4 * (int64_t(hb->n_loci) + 4)) + 1024,
                                  d, ctx.stream, total))
        return 1;
    sk_digt_call* dout = ar.take<sk_digt_call>(hb->n_loci);
    float* dde = ar.take<float>(total);
    uint32_t* scratch = ar.take<uint32_t>(total + hb->n_loci + 4);
    if (sk_site_digt_call_fused_dev(&d, opt, dout, dde, out_de ? 1 : 0, scratch, total, ctx.stream)) return 1;
    SK_HIP(skrt::memcpyAsync(out, dout, sizeof(sk_digt_call) * hb->n_loci, hipMemcpyDeviceToHost, ctx.stream));
    if (out_de && total) SK_HIP(skrt::memcpyAsync(out_de, dde, 4 * total, hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(skrt::streamSynchronize(ctx.stream));
    return 0;
}

} // extern "C"
