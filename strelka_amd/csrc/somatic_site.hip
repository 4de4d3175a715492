// somatic_site.hip -- hot path B (somatic SNV): per-locus 30-state frequency-grid likelihoods of the normal and tumor
// pileups and the 3x2 (normal genotype x {non-somatic, somatic}) posterior.
//
//   position_somatic_snv_call       L/applications/strelka/position_somatic_snv_strand_grid.cpp:230-363 (one tier)
//   get_diploid_gt_lhood_cached_simple / get_diploid_het_grid_lhood_cached / get_strand_ratio_lhood_spi
//                                   L/applications/strelka/position_somatic_snv_strand_grid_lhood_cached.cpp:41-234
//   calculate_result_set_grid       L/applications/strelka/qscore_calculator.cpp:47-209
//
// Two launches: `somatic_classify_kernel` streams the calls once and queues the loci that are not skipped by the
// reference's early return (reference base 'N', or both pileups all-reference); `somatic_snv_kernel` evaluates the queued
// loci, one thread per locus, 64 at a time, their calls staged through LDS with coalesced loads (normal sample, then
// tumor, through the same buffer) and the memo tables in LDS as one row per q-score.  The reference memoises every per-call term by (qscore, ratio index); those memo tables are
// built on the host (SkTables) so each of the 21 (+2x9 strand) accumulators is the same sequential float32 sum over the
// calls in pileup order as in the reference -- bit-identical.  Only the 9 strand states' final float logsum and the
// double-precision posterior evaluate device transcendentals.

#include "somatic_common.h"
#include "libm_flt32.h"

#include <cstddef>

int sk_upload_pileup_internal(const sk_pileup_batch* hb, bool need_de, SkArena& ar, size_t extra_bytes,
                              sk_pileup_batch& d, hipStream_t st, int64_t& total_calls);

namespace
{

struct SomArgs
{
    sk_pileup_batch n, t;
    const SkTables* tab;
    sk_somatic_snv_call* out;
    unsigned* work; // [0] = number of loci to evaluate, [1..] = their indices (any order)
    SomaticDerived d;
    // the whole-wrapper entry (sk_somatic_snv_call_tiers): per-locus forced output, non-somatic quality
    const uint8_t* forced;  // [n_loci] or nullptr
    int nonsomatic;         // isComputeNonSomatic: acts like forced output on every early return (:251,:184,:315)
    int* nonsom_q;          // [n_loci] nonsomatic_qphred of this tier's records, or nullptr
};

__device__ __forceinline__ bool locus_is_forced(const SomArgs& a, const int l)
{
    return a.d.is_forced_output || a.nonsomatic || (a.forced != nullptr && a.forced[l] != 0);
}

// getLogSum<float>, L/blt_util/logSumUtil.hh:33-41 with log1p_switch<float>, L/blt_util/math_util.hh:33-48: the reference's
// expf / log1pf / logf are the host libm's, restated for the device in libm_flt32.h; the device library's double-precision
// functions rounded once stand in when the host libm is another implementation (sk_libm_restated() == 0)
__device__ __forceinline__ float log_sum2f(float x1, float x2, const int exact_libm)
{
    if (x1 < x2) {
        const float t = x1;
        x1 = x2;
        x2 = t;
    }
    const float d = __fsub_rn(x2, x1);
    float e, l;
    if (!(exact_libm && sk_libm::expf_glibc(d, e))) e = static_cast<float>(exp(static_cast<double>(d)));
    if (fabsf(e) < 0.01f) {
        if (!(exact_libm && sk_libm::log1pf_glibc(e, l))) l = static_cast<float>(log1p(static_cast<double>(e)));
    } else {
        const float one_e = __fadd_rn(1.f, e);
        if (!(exact_libm && sk_libm::logf_glibc(one_e, l))) l = static_cast<float>(log(static_cast<double>(one_e)));
    }
    return __fadd_rn(x1, l);
}

// Every memoised term of one call, one LDS row per (q-score, call == reference base): 31 floats in the order the accumulators use
// them -- ref, het, hom (is_ref ? (v2, v1, v0) : (v0, v1, v2), :56-64), the off-strand term (is_ref ? ln_comp_error_prob :
// ln_error_prob + ln(1/3), :213,:221), hi[HET_RES] (-> grid[2*HET_RES-(r+1)]: is_ref ? c0 : c1, :104-110,:149-151), lo[HET_RES]
// (-> grid[r]: is_ref ? c1 : c0), on[HET_RES] (the on-strand term of the strand states: is_ref ? t0 : t1, :197-206) -- so that the
// per-call loop is loads + adds with no selects except the strand one.  The loop reads a row as six (with the strand terms: eight)
// `ds_read_b128`, written out: from source the compiler reads the same bytes in 16 pieces of 8 and 12, and the LDS pipe is what bounds
// the kernel (busy 80 % of its time, profiles/r04_v26_som_sq_counters.txt).
constexpr int QROW_HI = 4, QROW_LO = 4 + HET_RES, QROW_ON = 4 + 2 * HET_RES;
struct alignas(16) QRow
{
    float f[32];
    float pad[4]; // 144-byte rows = 36 banks: sixteen consecutive rows tile the 64 banks
};
static_assert(sizeof(QRow) == 144 && 4 + 3 * HET_RES <= 32 && 4 + 2 * HET_RES <= 24, "eight / six 128-bit loads hold the fields");
// row of (q, is_ref): one further row per eight q-scores, so that q and q + 8 (binned qualities: 32 / 40) do not share banks
__device__ __forceinline__ int qrow_index(const unsigned q, const bool is_ref) { return int(2u * q + (is_ref ? 1u : 0u) + (q >> 3)); }
constexpr int N_QROWS = 2 * SK_NQ6 + SK_NQ6 / 8;

__device__ __forceinline__ void fill_qrows(const SkTables* __restrict__ T, QRow* s_q, const int tid, const int nthreads)
{
    for (int i = tid; i < 2 * SK_NQ6; i += nthreads) {
        const int q = i >> 1;
        const bool is_ref = (i & 1) != 0;
        QRow r;
        r.f[0] = is_ref ? T->s_v2[q] : T->s_v0[q];
        r.f[1] = T->s_v1[q];
        r.f[2] = is_ref ? T->s_v0[q] : T->s_v2[q];
        r.f[3] = is_ref ? T->t_off_ref[q] : T->t_off_alt[q];
#pragma unroll
        for (int k = 0; k < HET_RES; ++k) {
            r.f[QROW_HI + k] = is_ref ? T->s_c0[k][q] : T->s_c1[k][q];
            r.f[QROW_LO + k] = is_ref ? T->s_c1[k][q] : T->s_c0[k][q];
            r.f[QROW_ON + k] = is_ref ? T->t_c0[k][q] : T->t_c1[k][q];
        }
        for (int k = 4 + 3 * HET_RES; k < 32; ++k) r.f[k] = 0.f;
        r.pad[0] = r.pad[1] = r.pad[2] = r.pad[3] = 0.f;
        s_q[qrow_index(unsigned(q), is_ref)] = r;
    }
}

// measured on the bench input (ms per 2^20 loci, S1 / S2): (4 waves/SIMD, 32-call chunks) 0.78, (4, 16) 0.75,
// (5, 16) 0.71; S2 at 1 / 2 / 3 waves per SIMD 0.34 / 0.27 / 0.30
#ifndef SOM_WPE
#define SOM_WPE 4       // S1: waves per SIMD the register allocation is held to (128 VGPRs: a row of ten float4s + 39 accumulators)
#endif
#ifndef SOM_CHUNK
#define SOM_CHUNK 16
#endif
#ifndef SOM_POST_WPE
#define SOM_POST_WPE 3  // S2: three 256-thread workgroups (44 KB of LDS rows each) per CU
#endif
constexpr int SOM_WAVES = 4;               // waves per block; each works through its own 64 queued loci
constexpr int SOM_THREADS = 64 * SOM_WAVES;
constexpr int CHUNK = SOM_CHUNK;           // calls of one locus staged per round
constexpr int ROW_DW = CHUNK / 2 + 1;      // row stride in dwords: odd, so the lanes' row reads fall on distinct LDS banks
constexpr int LOCI_PER_TRIP = 64 / CHUNK;  // loci one staging trip of the wave fetches (CHUNK lanes each)

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct LhoodAcc
{
    float acc[PRESTRAND];
    float sf[HET_RES], sr[HET_RES];
    unsigned alt_count[4];
};

// a row as 128-bit LDS reads, issued together and waited for once
typedef float som_v4f __attribute__((ext_vector_type(4)));
template <bool WITH_STRAND>
__device__ __forceinline__ void read_qrow(const QRow* row, som_v4f (&v)[8])
{
    const unsigned at = unsigned(uintptr_t((const __attribute__((address_space(3))) QRow*)row)); // (the LDS address: the low half of the generic one)
    if (WITH_STRAND) {
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                     "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\t"
                     "ds_read_b128 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                     : "v"(at)
                     : "memory");
    } else {
        asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\tds_read_b128 %3, %6 offset:48\n\t"
                     "ds_read_b128 %4, %6 offset:64\n\tds_read_b128 %5, %6 offset:80\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                     : "v"(at)
                     : "memory");
    }
}

// the per-call updates of one sample's accumulators for calls row[0..cnt), in pileup order
template <bool WITH_STRAND>
__device__ __forceinline__ void accumulate_calls(const uint16_t* row, const int cnt, const unsigned ref_gt, const QRow* Q, LhoodAcc& A)
{
    for (int i = 0; i < cnt; ++i) {
        const uint16_t bc = row[i];
        const unsigned q = SKC_Q(bc), obs = SKC_BASE(bc);
        const bool is_ref = (obs == ref_gt);
        if (!is_ref) {
#pragma unroll
            for (unsigned b = 0; b < 4; ++b) A.alt_count[b] += (obs == b) ? 1u : 0u;
        }
        som_v4f R4[8];
        read_qrow<WITH_STRAND>(Q + qrow_index(q, is_ref), R4);
        auto f = [&](const int k) -> float { return R4[k >> 2][k & 3]; }; // (constant k after unrolling)
        A.acc[SOM_REF] = __fadd_rn(A.acc[SOM_REF], f(0));
        A.acc[SOM_HET] = __fadd_rn(A.acc[SOM_HET], f(1));
        A.acc[SOM_HOM] = __fadd_rn(A.acc[SOM_HOM], f(2));
#pragma unroll
        for (int r = 0; r < HET_RES; ++r) {
            A.acc[SOM_SIZE + (2 * HET_RES - (r + 1))] = __fadd_rn(A.acc[SOM_SIZE + (2 * HET_RES - (r + 1))], f(QROW_HI + r));
            A.acc[SOM_SIZE + r] = __fadd_rn(A.acc[SOM_SIZE + r], f(QROW_LO + r));
        }
        if (WITH_STRAND) {
            const bool fwd = SKC_FWD(bc);
            const float off = f(3);
#pragma unroll
            for (int r = 0; r < HET_RES; ++r) {
                const float on = f(QROW_ON + r);
                A.sf[r] = __fadd_rn(A.sf[r], fwd ? on : off);
                A.sr[r] = __fadd_rn(A.sr[r], fwd ? off : on);
            }
        }
    }
}

// one sample's 30 likelihoods for the wave's 64 queued loci (lane t owns locus `l`, -1 = none).  The calls are staged
// CHUNK per locus at a time into the wave's LDS rows: 32 lanes fetch 64 contiguous bytes of one locus, so every global
// request is a full segment and the buffer stays small enough for the register file to bound the occupancy.
template <bool WITH_STRAND>
__device__ __forceinline__ void sample_lhood(const sk_pileup_batch& b, const int l, const unsigned ref_gt, uint32_t* rows,
                                             const QRow* Q, const float ln_one_half, const int exact_libm, float* __restrict__ lhood, unsigned& alt_id)
{
    const int lane = threadIdx.x & 63;
    int64_t g0 = 0;
    int n = 0;
    if (l >= 0) {
        g0 = b.call_off[l];
        n = int(b.call_off[l + 1] - g0);
    }
    int maxn = n;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxn = max(maxn, __shfl_xor(maxn, d));
    LhoodAcc A;
#pragma unroll
    for (int i = 0; i < PRESTRAND; ++i) A.acc[i] = 0.f;
#pragma unroll
    for (int r = 0; r < HET_RES; ++r) A.sf[r] = A.sr[r] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) A.alt_count[k] = 0;

    const uint16_t* __restrict__ calls = b.calls;
    uint16_t* rows16 = reinterpret_cast<uint16_t*>(rows);
    const uint16_t* own = rows16 + lane * (2 * ROW_DW);
    const int part = lane / CHUNK, k = lane % CHUNK;
    for (int c = 0; c < maxn; c += CHUNK) {
#pragma unroll 8
        for (int j2 = 0; j2 < 64 / LOCI_PER_TRIP; ++j2) {
            const int j = LOCI_PER_TRIP * j2 + part;
            const int64_t gj = __shfl(g0, j);
            const int nj = __shfl(n, j);
            if (c + k < nj) rows16[j * (2 * ROW_DW) + k] = calls[gj + c + k];
        }
        wave_sync();
        accumulate_calls<WITH_STRAND>(own, min(max(n - c, 0), CHUNK), ref_gt, Q, A);
        wave_sync();
    }
#pragma unroll
    for (int i = 0; i < PRESTRAND; ++i) lhood[i] = A.acc[i];
#pragma unroll
    for (int r = 0; r < HET_RES; ++r)
        lhood[PRESTRAND + r] = WITH_STRAND ? __fadd_rn(log_sum2f(A.sf[r], A.sr[r], exact_libm), ln_one_half) : 0.f;

    // snp_pos_info::get_most_frequent_alt_id, L/blt_common/snp_pos_info.hh:164-190
    alt_id = ref_gt;
    unsigned max_count = 0;
#pragma unroll
    for (unsigned bb = 0; bb < 4; ++bb) {
        if (A.alt_count[bb] > max_count && bb != ref_gt) {
            max_count = A.alt_count[bb];
            alt_id = bb;
        }
    }
}

constexpr int CLS_THREADS = 256;

// S0: which loci need evaluating at all.  The reference returns before any likelihood is computed when the reference base
// is 'N' or both pileups show only the reference base (position_somatic_snv_strand_grid.cpp:244-254); on 40x + 110x
// data that is most loci.  Calls are streamed with 16-byte loads; a locus is queued when any of its calls differs from
// its reference base (or unconditionally for forced output).  The queue order is arbitrary -- loci are independent.
__device__ __forceinline__ void flag_nonref_loci(const sk_pileup_batch& b, const int l0, const int nl, int64_t* s_off,
                                                 const unsigned char* s_ref, unsigned* s_flag)
{
    const int tid = threadIdx.x;
    for (int j = tid; j <= nl; j += CLS_THREADS) s_off[j] = b.call_off[l0 + j];
    __syncthreads();
    const int64_t c0 = s_off[0], c1 = s_off[nl];
    const uint16_t* __restrict__ g = b.calls;
    const int64_t a0 = c0 - int64_t((reinterpret_cast<uintptr_t>(g + c0) & 15u) >> 1); // 16-byte aligned start
    for (int64_t i = a0 + int64_t(tid) * 8; i < c1; i += CLS_THREADS * 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(g + i); // stays inside the 16-byte blocks of valid calls
        const unsigned w[4] = { v.x, v.y, v.z, v.w };
        int lo = 0, hi = nl; // locus of call max(i, c0): the last j with s_off[j] <= it
        const int64_t first = (i < c0) ? c0 : i;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= first) lo = mid; else hi = mid;
        }
        int loc = lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t idx = i + k;
            if (idx < c0 || idx >= c1) continue;
            while (idx >= s_off[loc + 1]) ++loc;
            const unsigned bc = (w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
            if (SKC_BASE(bc) != unsigned(s_ref[loc])) s_flag[loc] = 1u;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(CLS_THREADS) void somatic_classify_kernel(const SomArgs a)
{
    __shared__ int64_t s_off[CLS_THREADS + 1];
    __shared__ unsigned char s_ref[CLS_THREADS];
    __shared__ unsigned s_flag[CLS_THREADS];
    __shared__ unsigned s_wave_cnt[CLS_THREADS / 64], s_base;
    const int tid = threadIdx.x;
    const int l0 = blockIdx.x * CLS_THREADS;
    const int nl = min(CLS_THREADS, a.n.n_loci - l0);
    const unsigned ref = (tid < nl) ? a.n.ref_base[l0 + tid] : 4u;
    s_ref[tid] = (unsigned char)ref;
    s_flag[tid] = (tid < nl && locus_is_forced(a, l0 + tid)) ? 1u : 0u;
    __syncthreads();
    if (!(a.d.is_forced_output || a.nonsomatic)) {
        flag_nonref_loci(a.n, l0, nl, s_off, s_ref, s_flag);
        flag_nonref_loci(a.t, l0, nl, s_off, s_ref, s_flag);
    }
    const bool active = (tid < nl) && (ref < 4u) && (s_flag[tid] != 0u);
    const unsigned long long m = __ballot(active);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) s_wave_cnt[wave] = unsigned(__popcll(m));
    __syncthreads();
    if (tid == 0) {
        unsigned tot = 0;
        for (int w = 0; w < CLS_THREADS / 64; ++w) {
            const unsigned c = s_wave_cnt[w];
            s_wave_cnt[w] = tot;
            tot += c;
        }
        s_base = tot ? atomicAdd(a.work, tot) : 0u;
    }
    __syncthreads();
    if (active) a.work[1 + s_base + s_wave_cnt[wave] + unsigned(__popcll(m & ((1ull << lane) - 1ull)))] = unsigned(l0 + tid);
}

// write one 30-float block of the wave's 64 result records (at `dword_off` inside the record) through the wave's LDS rows,
// so that 15 consecutive lanes store the 120 contiguous bytes of one record instead of every lane scattering dwords
__device__ __forceinline__ void emit_lhood_block(uint32_t* rows, const int l, const float* v, sk_somatic_snv_call* out,
                                                 const int dword_off)
{
    constexpr int PER = (64 * ROW_DW) / GRID >= 32 ? 32 : 16; // records per pass that fit the rows buffer
    static_assert(PER * GRID <= 64 * ROW_DW, "a pass fits the rows buffer");
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int h = 0; h < 64 / PER; ++h) {
        wave_sync();
        if (lane / PER == h) {
#pragma unroll
            for (int i = 0; i < GRID; ++i) rows[(lane % PER) * GRID + i] = __float_as_uint(v[i]);
        }
        wave_sync();
        for (int i0 = 0; i0 < PER * (GRID / 2); i0 += 64) { // all lanes take every trip: __shfl reads 0 from an idle lane
            const int idx = i0 + lane;
            const int j = min(idx / (GRID / 2), PER - 1), c = idx - j * (GRID / 2);
            const int lj = __shfl(l, PER * h + j);
            if (idx < PER * (GRID / 2) && lj >= 0) {
                const uint2 w = *reinterpret_cast<const uint2*>(rows + j * GRID + 2 * c);
                *reinterpret_cast<uint2*>(reinterpret_cast<uint32_t*>(out + lj) + dword_off + 2 * c) = w;
            }
        }
    }
}

// S1: the 2 x 30 likelihoods of the queued loci, one thread per locus, 64 queue entries per wave
__global__ __launch_bounds__(SOM_THREADS) __attribute__((amdgpu_waves_per_eu(SOM_WPE, SOM_WPE))) void somatic_lhood_kernel(const SomArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_rows[SOM_WAVES][64 * ROW_DW];
    __shared__ __attribute__((aligned(16))) QRow s_q[N_QROWS];

    const int tid = threadIdx.x;
    const unsigned n_work = a.work[0];
    if (unsigned(blockIdx.x) * SOM_THREADS >= n_work) return;
    fill_qrows(a.tab, s_q, tid, SOM_THREADS);
    __syncthreads();
    const float ln_one_half = a.tab->s_ln_one_half;
    const unsigned w = unsigned(blockIdx.x) * SOM_THREADS + unsigned(tid);
    const int l = (w < n_work) ? int(a.work[1 + w]) : -1;
    const unsigned ref = (l >= 0) ? a.n.ref_base[l] : 0u;
    uint32_t* rows = s_rows[tid >> 6];
    unsigned alt_n, alt_t;
    {
        float lh[GRID];
        sample_lhood<false>(a.n, l, ref, rows, s_q, ln_one_half, a.d.exact_libm, lh, alt_n);
        emit_lhood_block(rows, l, lh, a.out, 0);
    }
    {
        float lh[GRID];
        sample_lhood<true>(a.t, l, ref, rows, s_q, ln_one_half, a.d.exact_libm, lh, alt_t);
        emit_lhood_block(rows, l, lh, a.out, GRID);
    }
    if (l >= 0) {
        a.out[l].normal_alt_id = alt_n;
        a.out[l].tumor_alt_id = alt_t;
    }
}

// S2: posterior of the queued loci from the likelihoods S1 left in their records (a13).  The 2 x 21 likelihoods of a thread's
// locus sit in an LDS row (odd stride: conflict-free), so the (Fn,Ft) enumeration can stay a loop over Ft with run-time
// indices and the table-driven exp is inlined a handful of times instead of called ~150 times.
constexpr int POST_THREADS = 256;
constexpr int POST_ROW = 2 * PRESTRAND + 1;
__global__ __launch_bounds__(POST_THREADS) __attribute__((amdgpu_waves_per_eu(SOM_POST_WPE, SOM_POST_WPE))) void somatic_posterior_kernel(const SomArgs a)
{
    __shared__ float s_lh[POST_THREADS * POST_ROW];
    const unsigned n_work = a.work[0];
    const unsigned w = blockIdx.x * unsigned(POST_THREADS) + threadIdx.x;
    if (w >= n_work) return;
    const SkLibmTables lt = sk_libm_tables_default();
    const int l = int(a.work[1 + w]);
    sk_somatic_snv_call* o = a.out + l;
    float* row = s_lh + threadIdx.x * POST_ROW;
    {
        const float4* p = reinterpret_cast<const float4*>(o->normal_lhood); // records are 16-byte aligned (272 = 17 x 16)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float4 v = p[i];
            row[4 * i] = v.x;
            row[4 * i + 1] = v.y;
            row[4 * i + 2] = v.z;
            row[4 * i + 3] = v.w;
        }
        row[20] = o->normal_lhood[20];
    }
    float tl_strand[GRID - PRESTRAND + 1]; // tumor states 20..29 (20 goes to the row, 21..29 are the strand states)
    {
        const float2* t = reinterpret_cast<const float2*>(o->tumor_lhood); // 8-byte aligned (offset 120)
#pragma unroll
        for (int i = 0; i < GRID / 2; ++i) {
            const float2 v = t[i];
            if (2 * i < PRESTRAND) row[PRESTRAND + 2 * i] = v.x;
            else tl_strand[2 * i - (PRESTRAND - 1)] = v.x;
            if (2 * i + 1 < PRESTRAND) row[PRESTRAND + 2 * i + 1] = v.y;
            else tl_strand[2 * i + 1 - (PRESTRAND - 1)] = v.y;
        }
    }
    // (each thread reads back only its own row: no barrier needed)
    struct
    {
        uint32_t max_gt;
        int32_t qphred, from_ntype_qphred;
        uint32_t ntype;
    } rs;
    calculate_result_set_grid<true>(a.d, lt, [&](const unsigned i) { return row[i]; }, [&](const unsigned i) { return row[PRESTRAND + i]; }, rs);
    float strand_bias = 0.f;
    if (locus_is_forced(a, l) || rs.qphred != 0) { // strand bias (:216-225), skipped by the early return at :184
        float symm = row[PRESTRAND + SOM_SIZE];
        for (int i = SOM_SIZE; i < PRESTRAND; ++i) symm = (symm < row[PRESTRAND + i]) ? row[PRESTRAND + i] : symm;
        float strand = tl_strand[1];
#pragma unroll
        for (int i = 1; i <= GRID - PRESTRAND; ++i) strand = (strand < tl_strand[i]) ? tl_strand[i] : strand;
        const float dd = __fsub_rn(strand, symm);
        strand_bias = (0.f < dd) ? dd : 0.f;
    }
    // two wide stores (the tail of the record is 16-byte aligned) instead of six scattered dwords
    static_assert(offsetof(sk_somatic_snv_call, max_gt) % 16 == 0 && offsetof(sk_somatic_snv_call, strand_bias) == offsetof(sk_somatic_snv_call, max_gt) + 16 &&
                      offsetof(sk_somatic_snv_call, is_called) == offsetof(sk_somatic_snv_call, strand_bias) + 4,
                  "record tail layout");
    *reinterpret_cast<uint4*>(&o->max_gt) = make_uint4(rs.max_gt, uint32_t(rs.qphred), uint32_t(rs.from_ntype_qphred), rs.ntype);
    *reinterpret_cast<uint2*>(&o->strand_bias) = make_uint2(__float_as_uint(strand_bias), 1u);
}

// ---- the whole wrapper (sk_somatic_snv_call_tiers) ----------------------------------------------------------------------

struct TierArgs
{
    const unsigned* work1;            // loci past the early return
    unsigned* work2;                  // those of them whose tier1 result has qphred != 0: tier2 is evaluated for these
    const sk_somatic_snv_call* rec1;  // tier records (likelihoods, posterior, alt ids)
    const sk_somatic_snv_call* rec2;
    const int* nonsom_q1;
    const uint8_t* ref_base;
    const uint8_t* forced;
    int is_tier2, nonsomatic;
    int n_loci;
    sk_somatic_snv_genotype* out;
};

// every locus: the record the reference leaves on its early returns (:244-254)
__global__ __launch_bounds__(256) void somatic_prefill_kernel(const TierArgs a)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= a.n_loci) return;
    const unsigned ref = a.ref_base[l];
    sk_somatic_snv_genotype g;
    memset(&g, 0, sizeof(g));
    if (ref < 4u) {
        g.ref_gt = ref;
        g.is_forced_output = (a.forced != nullptr && a.forced[l] != 0) ? 1 : 0;
    }
    a.out[l] = g;
}

// tier2 is evaluated only where tier1 found something (:274-282)
__global__ __launch_bounds__(256) void somatic_tier2_queue_kernel(const TierArgs a)
{
    const unsigned n_work = a.work1[0];
    const unsigned w = blockIdx.x * 256u + threadIdx.x;
    const bool active = (w < n_work) && (a.rec1[a.work1[1 + w]].qphred != 0);
    const unsigned long long m = __ballot(active);
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == 0 && m != 0ull) base = atomicAdd(a.work2, unsigned(__popcll(m)));
    base = __shfl(base, 0);
    if (active) a.work2[1 + base + unsigned(__popcll(m & ((1ull << lane) - 1ull)))] = a.work1[1 + w];
}

// the non-somatic quality of the wrapper (:186-214): 21 x 21 joint states with gvcf_nonsomatic_gvcf_prior (:120-155),
// opt_normalize_ln_distro (L/blt_util/prob_util.hh:248-311; its "opt max" is compared with the running max that has just
// been updated, so it is the first predicate-true entry, i.e. state (REF,REF)), then the sum over fn == ft
struct NonsomArgs
{
    const unsigned* work;
    const sk_somatic_snv_call* rec;
    int* nonsom_q;
    unsigned valid_mask; // isValidNonsomaticIndex per tumor state (:93-113)
    float ln_half;       // std::log(0.5f)
    int exact_libm;
};

constexpr int NS_THREADS = 256;
__global__ __launch_bounds__(NS_THREADS) void somatic_nonsomatic_kernel(const NonsomArgs a)
{
    __shared__ float s_lh[NS_THREADS * POST_ROW];
    const unsigned n_work = a.work[0];
    const unsigned w = blockIdx.x * unsigned(NS_THREADS) + threadIdx.x;
    if (w >= n_work) return;
    const SkLibmTables lt = sk_libm_tables_default();
    const int l = int(a.work[1 + w]);
    const sk_somatic_snv_call* o = a.rec + l;
    float* row = s_lh + threadIdx.x * POST_ROW;
    for (int i = 0; i < PRESTRAND; ++i) {
        row[i] = o->normal_lhood[i];
        row[PRESTRAND + i] = o->tumor_lhood[i];
    }
    const float neg_inf = -INFINITY;
    auto value = [&](const unsigned fn, const unsigned ft) -> float {
        float prior;
        if (!((a.valid_mask >> ft) & 1u)) prior = neg_inf;
        else if (fn == ft) prior = 0.f; // std::log(1.f)
        else if (fn == SOM_REF || fn == SOM_HOM) prior = a.ln_half;
        else prior = neg_inf;
        return __fadd_rn(__fadd_rn(row[fn], row[PRESTRAND + ft]), prior);
    };
    float maxv = value(0, 0);
    const double opt_max = double(maxv);
    for (unsigned ft = 0; ft < PRESTRAND; ++ft)
        for (unsigned fn = 0; fn < PRESTRAND; ++fn) {
            const float v = value(fn, ft);
            if (v > maxv) maxv = v;
        }
    const double mx = double(maxv);
    double sum = 0.;
    for (unsigned ft = 0; ft < PRESTRAND; ++ft)
        for (unsigned fn = 0; fn < PRESTRAND; ++fn) {
            const double p = double(value(fn, ft));
            const double mdiff = __dsub_rn(mx, p);
            if (mdiff > 20.) {
                if (fn != ft) continue;
                if (__dsub_rn(opt_max, p) > 5.) continue;
            }
            sum = __dadd_rn(sum, sk_exp(-mdiff, a.exact_libm, lt));
        }
    const double inv = __ddiv_rn(1., sum);
    double nonsomatic_sum = 0.;
    for (unsigned f = 0; f < PRESTRAND; ++f) {
        const double p = double(value(f, f));
        const double mdiff = __dsub_rn(mx, p);
        double e = 0.;
        if (!(mdiff > 20. && __dsub_rn(opt_max, p) > 5.)) e = sk_exp(-mdiff, a.exact_libm, lt);
        nonsomatic_sum = __dadd_rn(nonsomatic_sum, __dmul_rn(e, inv));
    }
    a.nonsom_q[l] = error_prob_to_qphred_d(__dsub_rn(1., nonsomatic_sum), a.exact_libm, lt);
}

// tier selection, NTYPE conflict and the final record (:315-362)
__global__ __launch_bounds__(256) void somatic_combine_kernel(const TierArgs a)
{
    const unsigned n_work = a.work1[0];
    const unsigned w = blockIdx.x * 256u + threadIdx.x;
    if (w >= n_work) return;
    const int l = int(a.work1[1 + w]);
    const sk_somatic_snv_call& r0 = a.rec1[l];
    // tier_rs[1] = tier_rs[0] where tier1 found nothing (:277-281)
    const sk_somatic_snv_call& r1 = (a.is_tier2 && r0.qphred != 0) ? a.rec2[l] : r0;
    const bool forced = a.nonsomatic || (a.forced != nullptr && a.forced[l] != 0);
    sk_somatic_snv_genotype g = a.out[l];
    g.is_computed = 1;
    if (forced || !((r0.qphred == 0) || (a.is_tier2 && r1.qphred == 0))) {
        uint8_t snv_tier = 0, from_tier = 0;
        if (a.is_tier2) {
            if (r0.qphred > r1.qphred) snv_tier = 1;
            if (r0.from_ntype_qphred > r1.from_ntype_qphred) from_tier = 1;
        }
        const sk_somatic_snv_call& rs = from_tier ? r1 : r0;
        g.snv_tier = snv_tier;
        g.snv_from_ntype_tier = from_tier;
        g.max_gt = rs.max_gt;
        g.from_ntype_qphred = rs.from_ntype_qphred;
        g.normal_alt_id = rs.normal_alt_id;
        g.tumor_alt_id = rs.tumor_alt_id;
        g.strand_bias = double(rs.strand_bias);
        if (a.is_tier2 && r0.ntype != r1.ntype) {
            g.ntype = 3u; // NTYPE::CONFLICT
            g.from_ntype_qphred = 0;
        } else {
            g.ntype = (rs.ntype == SOM_REF) ? 0u : ((rs.ntype == SOM_HOM) ? 1u : 2u);
        }
        g.qphred = (snv_tier ? r1 : r0).qphred;
        g.nonsomatic_qphred = a.nonsomatic ? a.nonsom_q1[l] : 0;
    }
    a.out[l] = g;
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
    d.exact_libm = sk_ctx().libm_restated ? 1 : 0;
}

} // namespace


// one sample's cleaned pileup columns to the device (calls validated: base ids 0..3)
static int upload_somatic_pileup(const sk_pileup_batch* hb, const int n, SkArena& ar, hipStream_t st, sk_pileup_batch& d)
{
    if (hb->call_off[0] != 0) return sk_fail("pileup batch: call_off must start at 0");
    const int64_t tc = hb->call_off[n];
    for (int64_t i = 0; i < tc; ++i)
        if (SKC_BASE(hb->calls[i]) > 3) return sk_fail("somatic pileup batch: basecall with base_id > 3");
    d = *hb;
    int64_t* off = ar.take<int64_t>(n + 1);
    SK_HIP(skrt::memcpyAsync(off, hb->call_off, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, st));
    d.call_off = off;
    uint16_t* calls = ar.take<uint16_t>(tc);
    if (tc) SK_HIP(skrt::memcpyAsync(calls, hb->calls, 2 * tc, hipMemcpyHostToDevice, st));
    d.calls = calls;
    d.de = nullptr;
    d.ploidy = nullptr;
    uint8_t* rb = ar.take<uint8_t>(n);
    SK_HIP(skrt::memcpyAsync(rb, hb->ref_base, n, hipMemcpyHostToDevice, st));
    d.ref_base = rb;
    return 0;
}

static size_t pileup_upload_bytes(const sk_pileup_batch* hb, const int n)
{
    return sk_align256(sizeof(int64_t) * (size_t(n) + 1)) + sk_align256(size_t(n)) + sk_align256(2 * size_t(hb->call_off[n])) + 3 * 256;
}

extern "C" {

int sk_somatic_snv_call_batch_dev(const sk_pileup_batch* n, const sk_pileup_batch* t, const sk_somatic_snv_options* opt,
                                  int is_forced_output, sk_somatic_snv_call* dev_out, void* dev_scratch, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!n || !t || !opt || !dev_out || !dev_scratch) return sk_fail("sk_somatic_snv_call_batch_dev: null argument");
    if (n->n_loci != t->n_loci) return sk_fail("sk_somatic_snv_call_batch_dev: normal/tumor n_loci differ");
    if (n->n_loci <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    SomArgs a;
    a.n = *n;
    a.t = *t;
    a.tab = sk_ctx().dev_tables;
    a.out = dev_out;
    a.work = static_cast<unsigned*>(dev_scratch);
    a.forced = nullptr;
    a.nonsomatic = 0;
    a.nonsom_q = nullptr;
    derive(*opt, is_forced_output, a.d);
    // skipped loci report an all-zero record (is_called = 0)
    SK_HIP(skrt::memsetAsync(dev_out, 0, sizeof(sk_somatic_snv_call) * size_t(n->n_loci), st));
    SK_HIP(skrt::memsetAsync(a.work, 0, sizeof(unsigned), st));
    SK_LAUNCH(somatic_classify_kernel, dim3((n->n_loci + CLS_THREADS - 1) / CLS_THREADS), dim3(CLS_THREADS), 0, st, a);
    SK_HIP(skrt::getLastError());
    // sized for every locus being queued; blocks past the end of the queue exit at once
    SK_LAUNCH(somatic_lhood_kernel, dim3((n->n_loci + SOM_THREADS - 1) / SOM_THREADS), dim3(SOM_THREADS), 0, st, a);
    SK_HIP(skrt::getLastError());
    SK_LAUNCH(somatic_posterior_kernel, dim3((n->n_loci + POST_THREADS - 1) / POST_THREADS), dim3(POST_THREADS), 0, st, a);
    SK_HIP(skrt::getLastError());
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
    SK_HIP(skrt::setDevice(ctx.device));
    // the two uploads and the output share one arena
    SkArena ar;
    const size_t need = 2 * (sk_align256(sizeof(int64_t) * (n + 1)) + sk_align256(n) * 2 + 16 * 256) +
                        sk_align256(2 * hn->call_off[n]) + sk_align256(2 * ht->call_off[n]) +
                        sk_align256(sizeof(sk_somatic_snv_call) * n) + sk_align256(4 * (size_t(n) + 4)) + 4096;
    if (ar.reserve(need)) return 1;
    auto up = [&](const sk_pileup_batch* hb, sk_pileup_batch& d) -> int { return upload_somatic_pileup(hb, n, ar, ctx.stream, d); };
    sk_pileup_batch dn, dt;
    if (up(hn, dn) || up(ht, dt)) return 1;
    sk_somatic_snv_call* dout = ar.take<sk_somatic_snv_call>(n);
    unsigned* work = ar.take<unsigned>(size_t(n) + 4);
    if (sk_somatic_snv_call_batch_dev(&dn, &dt, opt, is_forced_output, dout, work, ctx.stream)) return 1;
    SK_HIP(skrt::memcpyAsync(out, dout, sizeof(sk_somatic_snv_call) * n, hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(skrt::streamSynchronize(ctx.stream));
    return 0;
}

// ---- sk_somatic_snv_call_tiers: the whole of position_somatic_snv_call ----------------------------------------------------

size_t sk_somatic_snv_tiers_scratch_bytes(int32_t n_loci)
{
    const size_t n = size_t(n_loci > 0 ? n_loci : 0);
    return 2 * sk_align256(4 * (n + 4)) + 2 * sk_align256(sizeof(sk_somatic_snv_call) * n) + sk_align256(4 * n) + 1024;
}

int sk_somatic_snv_call_tiers_dev(const sk_pileup_batch* n1, const sk_pileup_batch* t1, const sk_pileup_batch* n2,
                                  const sk_pileup_batch* t2, const sk_somatic_snv_options* opt, const uint8_t* dev_forced,
                                  int is_compute_nonsomatic, sk_somatic_snv_genotype* dev_out, void* dev_scratch, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!n1 || !t1 || !opt || !dev_out || !dev_scratch) return sk_fail("sk_somatic_snv_call_tiers_dev: null argument");
    if ((n2 == nullptr) != (t2 == nullptr)) return sk_fail("sk_somatic_snv_call_tiers_dev: tier2 needs both samples");
    const int n = n1->n_loci;
    if (t1->n_loci != n || (n2 && (n2->n_loci != n || t2->n_loci != n))) return sk_fail("sk_somatic_snv_call_tiers_dev: n_loci differ");
    if (n <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const bool is_tier2 = (n2 != nullptr);

    char* p = static_cast<char*>(dev_scratch);
    auto take = [&](const size_t bytes) { char* r = p; p += sk_align256(bytes); return r; };
    unsigned* work1 = reinterpret_cast<unsigned*>(take(4 * (size_t(n) + 4)));
    unsigned* work2 = reinterpret_cast<unsigned*>(take(4 * (size_t(n) + 4)));
    sk_somatic_snv_call* rec1 = reinterpret_cast<sk_somatic_snv_call*>(take(sizeof(sk_somatic_snv_call) * size_t(n)));
    sk_somatic_snv_call* rec2 = reinterpret_cast<sk_somatic_snv_call*>(take(sizeof(sk_somatic_snv_call) * size_t(n)));
    int* nonsom_q = reinterpret_cast<int*>(take(4 * size_t(n)));

    SomArgs a;
    a.n = *n1;
    a.t = *t1;
    a.tab = sk_ctx().dev_tables;
    a.out = rec1;
    a.work = work1;
    a.forced = dev_forced;
    a.nonsomatic = is_compute_nonsomatic ? 1 : 0;
    a.nonsom_q = nonsom_q;
    derive(*opt, 0, a.d);

    TierArgs ta;
    ta.work1 = work1;
    ta.work2 = work2;
    ta.rec1 = rec1;
    ta.rec2 = rec2;
    ta.nonsom_q1 = nonsom_q;
    ta.ref_base = n1->ref_base;
    ta.forced = dev_forced;
    ta.is_tier2 = is_tier2 ? 1 : 0;
    ta.nonsomatic = a.nonsomatic;
    ta.n_loci = n;
    ta.out = dev_out;

    const dim3 g256((n + 255) / 256), b256(256);
    SK_HIP(skrt::memsetAsync(work1, 0, sizeof(unsigned), st));
    SK_HIP(skrt::memsetAsync(work2, 0, sizeof(unsigned), st));
    SK_LAUNCH(somatic_prefill_kernel, g256, b256, 0, st, ta);
    SK_LAUNCH(somatic_classify_kernel, dim3((n + CLS_THREADS - 1) / CLS_THREADS), dim3(CLS_THREADS), 0, st, a);
    SK_LAUNCH(somatic_lhood_kernel, dim3((n + SOM_THREADS - 1) / SOM_THREADS), dim3(SOM_THREADS), 0, st, a);
    SK_LAUNCH(somatic_posterior_kernel, dim3((n + POST_THREADS - 1) / POST_THREADS), dim3(POST_THREADS), 0, st, a);
    SK_HIP(skrt::getLastError());
    if (a.nonsomatic) {
        NonsomArgs na;
        na.work = work1;
        na.rec = rec1;
        na.nonsom_q = nonsom_q;
        na.exact_libm = a.d.exact_libm;
        // isValidNonsomaticIndex (:93-113) and the two finite priors of gvcf_nonsomatic_gvcf_prior (:120-155), host floats
        {
            const float nonSomaticMinFrac(0.1f);
            const float nonSomaticMinFracComp(1. - nonSomaticMinFrac);
            const float epsilon(0.0001f);
            const float lo(nonSomaticMinFrac - epsilon), hi(nonSomaticMinFracComp + epsilon);
            unsigned mask = 0;
            for (unsigned f = 0; f < PRESTRAND; ++f) {
                bool ok = true;
                if (!(f == SOM_REF || f == SOM_HOM)) {
                    const float frac = a.d.grid_frac[f];
                    if (frac < lo || frac > hi) ok = false;
                }
                if (ok) mask |= (1u << f);
            }
            na.valid_mask = mask;
            volatile float half = 0.5f;
            na.ln_half = std::log(half);
        }
        SK_LAUNCH(somatic_nonsomatic_kernel, dim3((n + NS_THREADS - 1) / NS_THREADS), dim3(NS_THREADS), 0, st, na);
        SK_HIP(skrt::getLastError());
    }
    if (is_tier2) {
        SK_LAUNCH(somatic_tier2_queue_kernel, g256, b256, 0, st, ta);
        SomArgs a2 = a;
        a2.n = *n2;
        a2.t = *t2;
        a2.out = rec2;
        a2.work = work2;
        a2.nonsom_q = nullptr;
        SK_LAUNCH(somatic_lhood_kernel, dim3((n + SOM_THREADS - 1) / SOM_THREADS), dim3(SOM_THREADS), 0, st, a2);
        SK_LAUNCH(somatic_posterior_kernel, dim3((n + POST_THREADS - 1) / POST_THREADS), dim3(POST_THREADS), 0, st, a2);
        SK_HIP(skrt::getLastError());
    }
    SK_LAUNCH(somatic_combine_kernel, g256, b256, 0, st, ta);
    SK_HIP(skrt::getLastError());
    return 0;
}

int sk_somatic_snv_call_tiers(const sk_pileup_batch* hn1, const sk_pileup_batch* ht1, const sk_pileup_batch* hn2,
                              const sk_pileup_batch* ht2, const sk_somatic_snv_options* opt, const uint8_t* is_forced_output,
                              int is_compute_nonsomatic, sk_somatic_snv_genotype* out)
{
    SK_REQUIRE_INIT();
    if (!hn1 || !ht1 || !opt || !out) return sk_fail("sk_somatic_snv_call_tiers: null argument");
    if ((hn2 == nullptr) != (ht2 == nullptr)) return sk_fail("sk_somatic_snv_call_tiers: tier2 needs both samples");
    const int n = hn1->n_loci;
    if (ht1->n_loci != n || (hn2 && (hn2->n_loci != n || ht2->n_loci != n))) return sk_fail("sk_somatic_snv_call_tiers: n_loci differ");
    if (n <= 0) return 0;
    if (std::memcmp(hn1->ref_base, ht1->ref_base, n) != 0 ||
        (hn2 && (std::memcmp(hn1->ref_base, hn2->ref_base, n) != 0 || std::memcmp(hn1->ref_base, ht2->ref_base, n) != 0)))
        return sk_fail("sk_somatic_snv_call_tiers: ref_base differs between the batches");
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    SkArena ar;
    size_t need = pileup_upload_bytes(hn1, n) + pileup_upload_bytes(ht1, n) + sk_align256(size_t(n)) +
                  sk_align256(sizeof(sk_somatic_snv_genotype) * size_t(n)) + sk_somatic_snv_tiers_scratch_bytes(n) + 4096;
    if (hn2) need += pileup_upload_bytes(hn2, n) + pileup_upload_bytes(ht2, n);
    if (ar.reserve(need)) return 1;
    sk_pileup_batch dn1, dt1, dn2, dt2;
    if (upload_somatic_pileup(hn1, n, ar, ctx.stream, dn1) || upload_somatic_pileup(ht1, n, ar, ctx.stream, dt1)) return 1;
    if (hn2 && (upload_somatic_pileup(hn2, n, ar, ctx.stream, dn2) || upload_somatic_pileup(ht2, n, ar, ctx.stream, dt2))) return 1;
    uint8_t* dforced = nullptr;
    if (is_forced_output) {
        dforced = ar.take<uint8_t>(n);
        SK_HIP(skrt::memcpyAsync(dforced, is_forced_output, n, hipMemcpyHostToDevice, ctx.stream));
    }
    sk_somatic_snv_genotype* dout = ar.take<sk_somatic_snv_genotype>(n);
    void* scratch = ar.take<char>(sk_somatic_snv_tiers_scratch_bytes(n));
    if (sk_somatic_snv_call_tiers_dev(&dn1, &dt1, hn2 ? &dn2 : nullptr, hn2 ? &dt2 : nullptr, opt, dforced, is_compute_nonsomatic,
                                      dout, scratch, ctx.stream))
        return 1;
    SK_HIP(skrt::memcpyAsync(out, dout, sizeof(sk_somatic_snv_genotype) * size_t(n), hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(skrt::streamSynchronize(ctx.stream));
    return 0;
}

} // extern "C"
