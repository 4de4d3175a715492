// score_alignments.hip -- hot path A: ln P(read | candidate alignment) for batches of (read, candidate alignment) pairs.
//
// Reproduces scoreCandidateAlignment (L/starling_common/starling_read_align_score.cpp:261-499) bit-for-bit: one running
// double per pair, terms added in path order, every term taken from the host-built tables (SkTables), no FMA
// contraction, no reassociation.
//
// Kernel A1 `score_wave_per_read` (the fast path)
//   * one 64-lane wavefront per read, one LANE per candidate alignment of that read (64 per pass), 4 waves per block;
//   * the read is expanded once per wave into LDS as a per-position ROW of 6 doubles {term vs hap A, C, G, T, other,
//     0.0}: M = ln(1-e_q) in the column of the read base, X = ln(e_q)-ln3 elsewhere; all 0.0 for an 'N' read base (the
//     reference's `continue`), all M for '=' ("always matches").  The read's haplotype source pool (reference window +
//     insert sequences) is stored as row-column byte offsets;
//   * all lanes sweep the read positions i = 0..L-1 together: per cell one ds_read_u8 (the lane's haplotype column,
//     prefetched one position ahead), one ds_read_b64 of row[i][column] -- the row address is wave-uniform, so the 64
//     reads of a row are LDS broadcasts -- and one dependent v_add_f64 one position behind.  A lane leaves the lock-step
//     sweep only at its own op boundaries (soft-clip term, non-candidate-indel penalty, next op's hap offset); its ops
//     are prefetched from global memory one transition ahead;
//   * soft-clipped and finished lanes read the 0.0 column: x + 0.0 == x for every x the sum can hold, so they stay in
//     lock-step without changing a bit of the result.
//
// Kernel A2 `score_thread_per_cal` (generic fallback: reads/pools too long for LDS, or unknown bounds)
//   one thread per candidate alignment, straight from global memory.
//
// Roofline: HBM-bound by the algorithmic bytes of SURVEY.md 8d (2L + H*(L_h+8) per read); this layout moves far fewer
// (the haplotypes are never materialised: 2L + pool + 8*ops + 8H per read), so the sweep is LDS/VALU-issue bound.

#include "sk_common.h"

#include <algorithm>
#include <vector>

namespace
{

constexpr int WAVES_PER_BLOCK = 4;
constexpr int WAVE = 64;
constexpr int OPS_CAP = 0;     // >0: stage up to this many scoring ops per wave per pass in LDS (0: prefetch from global)
constexpr int ROW_BYTES = 48;  // per read position: {A, C, G, T, other, 0.0} doubles
constexpr int ZERO_COL = 40;   // byte offset of the 0.0 column

__device__ __forceinline__ double dadd(double a, double b) { return __dadd_rn(a, b); }

// BAM 4-bit code of a haplotype base -> byte offset of its column in a table row
__device__ __forceinline__ unsigned hap_col_offset(const unsigned code)
{
    return code == SK_BAM_A ? 0u : code == SK_BAM_C ? 8u : code == SK_BAM_G ? 16u : code == SK_BAM_T ? 24u : 32u;
}

struct ScoreArgs
{
    sk_align_batch b;
    const SkTables* tab;
    double* out;
    int lds_tab_bytes; // per wave: ROW_BYTES * maxL
    int lds_hap_bytes; // per wave: align16(max(maxPool, maxL))
};

// Kernel A1.  LDS slab per wave: [table rows][hap column offsets][staged ops]
__global__ __launch_bounds__(WAVES_PER_BLOCK* WAVE) void score_wave_per_read(const ScoreArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / WAVE); // wave-uniform: L, P, offsets live in SGPRs
    const size_t per_wave = size_t(a.lds_tab_bytes) + a.lds_hap_bytes + OPS_CAP * sizeof(sk_score_op);
    unsigned char* slab = smem + per_wave * wave;
    unsigned char* tabb = slab;
    unsigned char* hap = slab + a.lds_tab_bytes;
    sk_score_op* lops = reinterpret_cast<sk_score_op*>(hap + a.lds_hap_bytes);

    const int r = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (r >= a.b.n_reads) return; // whole wave; waves never synchronise with each other

    const int64_t ro = a.b.read_off[r];
    const int L = int(a.b.read_off[r + 1] - ro);
    const int64_t ho = a.b.hap_off[r];
    const int P = int(a.b.hap_off[r + 1] - ho);
    const int cal_begin = a.b.cal_off[r], cal_end = a.b.cal_off[r + 1];
    {
        // per-position table: row[col] = the term added when the haplotype base of that column faces read base i
        //   'N' read base -> 0.0 everywhere (the reference `continue`s, :125/:158)
        //   '=' read base -> M everywhere   (always a match, :127/:160)
        const SkTables* __restrict__ T = a.tab;
        double* tab = reinterpret_cast<double*>(tabb);
        for (int j = lane; j < L; j += WAVE) {
            const unsigned rc = a.b.read_code[ro + j];
            unsigned q = a.b.read_qual[ro + j];
            q = q > 70u ? 70u : q;
            const bool any = (rc == SK_BAM_ANY);
            const double M = any ? 0.0 : T->q2lncompe[q];
            const double X = any ? 0.0 : (rc == SK_BAM_REF ? M : T->q2mis[q]);
            double* row = tab + 6 * j;
            row[0] = (rc == SK_BAM_A) ? M : X;
            row[1] = (rc == SK_BAM_C) ? M : X;
            row[2] = (rc == SK_BAM_G) ? M : X;
            row[3] = (rc == SK_BAM_T) ? M : X;
            row[4] = X; // haplotype 'N'/other never equals a non-N read code
            row[5] = 0.0;
        }
        for (int j = lane; j < P; j += WAVE) hap[j] = (unsigned char)hap_col_offset(a.b.hap_code[ho + j]);
        for (int j = P + lane; j < a.lds_hap_bytes; j += WAVE) hap[j] = 32;
    }

    const double ln_quarter = a.tab->ln_quarter;
    const double ln_noncand = a.tab->ln_noncand;
    const sk_score_op* __restrict__ gops = a.b.ops;

    for (int cbase = cal_begin; cbase < cal_end; cbase += WAVE) {
        const int c = cbase + lane;
        const bool has_cal = (c < cal_end);
        const int clast = (cbase + WAVE < cal_end) ? cbase + WAVE : cal_end;
        const int64_t kbase = a.b.op_off[cbase];
        const int nstage = int(a.b.op_off[clast] - kbase);
        const bool in_lds = (OPS_CAP > 0) && (nstage <= OPS_CAP);
        __builtin_amdgcn_wave_barrier();
        if (in_lds) {
            // consecutive candidates' ops are contiguous: one coalesced copy for the whole wave
            const uint2* __restrict__ src = reinterpret_cast<const uint2*>(gops + kbase);
            uint2* dst = reinterpret_cast<uint2*>(lops);
            for (int j = lane; j < nstage; j += WAVE) dst[j] = src[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        int k = 0, kend = 0; // op cursor relative to kbase
        if (has_cal) {
            k = int(a.b.op_off[c] - kbase);
            kend = int(a.b.op_off[c + 1] - kbase);
        }
        auto load_op = [&](const int kk) -> uint2 {
            if (kk >= kend) return make_uint2(0u, 0u);
            return in_lds ? reinterpret_cast<const uint2*>(lops)[kk] : reinterpret_cast<const uint2*>(gops + kbase)[kk];
        };
        double lnp = 0.0;
        double pend = 0.0;  // term of the previous read position, not yet added (software pipeline of depth 1)
        int op_end = 0;     // read position at which the current op ends
        int delta = 0;      // hap index = delta + i inside a BASES op
        unsigned colmin = ZERO_COL; // 0 inside a BASES op, ZERO_COL otherwise (max() forces the 0.0 column)
        unsigned cur_flags = 0;
        uint2 nop = load_op(k); // the next op, prefetched one transition ahead

        // advance(): finish the current op (penalty), start following ops until one that spans read bases.
        // Called by exactly the lanes whose current op ends at read position i.
        auto advance = [&](const int i) {
            lnp = dadd(lnp, pend); // the last base term of the finished op precedes its penalty
            pend = 0.0;
            for (;;) {
                if (cur_flags & SK_OPFLAG_NONCANDIDATE_PENALTY) lnp = dadd(lnp, ln_noncand);
                cur_flags = 0;
                if (k >= kend) {
                    op_end = 0x7fffffff;
                    colmin = ZERO_COL;
                    delta = -i; // keeps the (ignored) hap read of an idle lane in range
                    return;
                }
                const unsigned w0 = nop.x;
                const int src = int(nop.y);
                ++k;
                nop = load_op(k);
                cur_flags = (w0 >> 24) & 0xffu;
                const unsigned kind = (w0 >> 16) & 0xffu;
                int len = int(w0 & 0xffffu);
                if (kind == SK_OP_BASES) {
                    colmin = 0;
                    delta = src - i;
                } else {
                    colmin = ZERO_COL;
                    delta = -i;
                    if (kind == SK_OP_SOFT_CLIP) {
                        lnp = dadd(lnp, __dmul_rn(double(unsigned(len)), ln_quarter));
                    } else {
                        len = 0;
                    }
                }
                op_end = i + len;
                if (len > 0) return;
            }
        };

        if (op_end == 0) advance(0);
        unsigned col_next = (L > 0) ? hap[delta] : 0u;
#pragma unroll 2
        for (int i = 0; i < L; ++i) {
            if (op_end == i && i > 0) { // a lane leaves the lock-step sweep only at its own op boundaries
                advance(i);
                col_next = hap[delta + i];
            }
            unsigned col = col_next;
            col = col > colmin ? col : colmin;
            const double v = *reinterpret_cast<const double*>(tabb + ROW_BYTES * i + col);
            col_next = hap[delta + i + 1]; // prefetch (hap slab is padded by 16 bytes beyond max(P, L))
            lnp = dadd(lnp, pend);
            pend = v;
        }
        if (op_end == L && L > 0) advance(L); // trailing penalty / NOBASE ops
        lnp = dadd(lnp, pend);
        if (has_cal) a.out[c] = lnp;
    }
}

__global__ void score_thread_per_cal(const ScoreArgs a)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.b.n_cals) return;
    // read index: last r with cal_off[r] <= c
    int lo = 0, hi = a.b.n_reads;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.b.cal_off[mid] <= c) lo = mid; else hi = mid;
    }
    const int r = lo;
    const int64_t ro = a.b.read_off[r];
    const int64_t ho = a.b.hap_off[r];
    const SkTables* __restrict__ T = a.tab;
    double lnp = 0.0;
    int rp = 0;
    for (int64_t k = a.b.op_off[c]; k < a.b.op_off[c + 1]; ++k) {
        const sk_score_op op = a.b.ops[k];
        if (op.kind == SK_OP_BASES) {
            for (int j = 0; j < int(op.length); ++j) {
                const unsigned rc = a.b.read_code[ro + rp + j];
                if (rc == SK_BAM_ANY) continue;
                unsigned q = a.b.read_qual[ro + rp + j];
                q = q > 70u ? 70u : q;
                const bool is_ref = (rc == SK_BAM_REF) || (rc == a.b.hap_code[ho + op.src + j]);
                lnp = dadd(lnp, is_ref ? T->q2lncompe[q] : T->q2mis[q]);
            }
            rp += op.length;
        } else if (op.kind == SK_OP_SOFT_CLIP) {
            lnp = dadd(lnp, __dmul_rn(double(unsigned(op.length)), T->ln_quarter));
            rp += op.length;
        }
        if (op.flags & SK_OPFLAG_NONCANDIDATE_PENALTY) lnp = dadd(lnp, T->ln_noncand);
    }
    a.out[c] = lnp;
}

inline int align16(int n) { return (n + 15) & ~15; }

} // namespace

extern "C" int sk_score_alignments_dev(const sk_align_batch* b, double* dev_out_lnp, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!b || !dev_out_lnp) return sk_fail("sk_score_alignments_dev: null argument");
    if (b->n_reads < 0 || b->n_cals < 0) return sk_fail("sk_score_alignments_dev: negative count");
    if (b->n_reads == 0 || b->n_cals == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    ScoreArgs a;
    a.b = *b;
    a.tab = sk_ctx().dev_tables;
    a.out = dev_out_lnp;
    const int maxL = b->max_read_len, maxP = b->max_hap_len;
    a.lds_tab_bytes = ROW_BYTES * std::max(maxL, 1);
    a.lds_hap_bytes = align16(std::max(std::max(maxP, maxL), 1)) + 16; // +16: the sweep prefetches one byte ahead
    const size_t per_wave = size_t(a.lds_tab_bytes) + a.lds_hap_bytes + OPS_CAP * sizeof(sk_score_op);
    const size_t lds = per_wave * WAVES_PER_BLOCK;
    // fast path when 4 waves' slabs leave room for >= 2 workgroups per CU (160 KiB LDS)
    if (maxL > 0 && maxP > 0 && lds <= 64 * 1024) {
        const int blocks = (b->n_reads + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
        hipLaunchKernelGGL(score_wave_per_read, dim3(blocks), dim3(WAVES_PER_BLOCK * WAVE), lds, st, a);
    } else {
        const int threads = 256;
        const int blocks = (b->n_cals + threads - 1) / threads;
        hipLaunchKernelGGL(score_thread_per_cal, dim3(blocks), dim3(threads), 0, st, a);
    }
    SK_HIP(hipGetLastError());
    return 0;
}

// test hook: force the generic kernel regardless of the bounds (parity of the two kernels is tested against each other)
extern "C" int sk_score_alignments_dev_generic(const sk_align_batch* b, double* dev_out_lnp, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!b || !dev_out_lnp) return sk_fail("sk_score_alignments_dev_generic: null argument");
    if (b->n_reads <= 0 || b->n_cals <= 0) return 0;
    ScoreArgs a;
    a.b = *b;
    a.tab = sk_ctx().dev_tables;
    a.out = dev_out_lnp;
    a.lds_tab_bytes = a.lds_hap_bytes = 0;
    const int threads = 256;
    hipLaunchKernelGGL(score_thread_per_cal, dim3((b->n_cals + threads - 1) / threads), dim3(threads), 0,
                       static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(hipGetLastError());
    return 0;
}

extern "C" int sk_score_alignments(const sk_align_batch* hb, double* out_lnp)
{
    SK_REQUIRE_INIT();
    if (!hb || !out_lnp) return sk_fail("sk_score_alignments: null argument");
    if (hb->n_reads < 0) return sk_fail("sk_score_alignments: negative n_reads");
    if (hb->n_reads == 0) return 0;
    const int n = hb->n_reads;
    if (hb->read_off[0] != 0 || hb->hap_off[0] != 0 || hb->cal_off[0] != 0)
        return sk_fail("sk_score_alignments: CSR offsets must start at 0");
    const int n_cals = hb->cal_off[n];
    if (n_cals != hb->n_cals) return sk_fail("sk_score_alignments: n_cals != cal_off[n_reads]");
    if (n_cals == 0) return 0;
    if (hb->op_off[0] != 0) return sk_fail("sk_score_alignments: op_off must start at 0");
    const int64_t n_ops = hb->op_off[n_cals];
    if (n_ops != hb->n_ops) return sk_fail("sk_score_alignments: n_ops != op_off[n_cals]");
    const int64_t n_bases = hb->read_off[n], n_hap = hb->hap_off[n];

    // validation the reference performs by throwing: q <= 70 (qscore_cache.cpp:53-75); plus structural checks that
    // keep the kernel's LDS indexing in range
    int maxL = 0, maxP = 0;
    for (int r = 0; r < n; ++r) {
        const int64_t L = hb->read_off[r + 1] - hb->read_off[r], P = hb->hap_off[r + 1] - hb->hap_off[r];
        if (L < 0 || P < 0 || L > 0x7fff0000 || P > 0x7fff0000) return sk_fail("sk_score_alignments: bad CSR offsets");
        maxL = std::max<int>(maxL, int(L));
        maxP = std::max<int>(maxP, int(P));
        for (int c = hb->cal_off[r]; c < hb->cal_off[r + 1]; ++c) {
            int64_t covered = 0;
            for (int64_t k = hb->op_off[c]; k < hb->op_off[c + 1]; ++k) {
                const sk_score_op& op = hb->ops[k];
                if (op.kind == SK_OP_BASES) {
                    if (op.src < 0 || int64_t(op.src) + op.length > P)
                        return sk_fail("sk_score_alignments: op source range outside the read's hap pool");
                    covered += op.length;
                } else if (op.kind == SK_OP_SOFT_CLIP) {
                    covered += op.length;
                } else if (op.kind != SK_OP_NOBASE) {
                    return sk_fail("sk_score_alignments: unknown op kind");
                }
            }
            if (covered != L) return sk_fail("sk_score_alignments: candidate alignment does not span its read");
        }
    }
    for (int64_t i = 0; i < n_bases; ++i)
        if (hb->read_qual[i] > 70) return sk_fail("Attempting to lookup basecall quality score which exceeds the maximum cached score of 70");

    SkContext& ctx = sk_ctx();
    SK_HIP(hipSetDevice(ctx.device));
    SkArena ar;
    const size_t need = sk_align256(sizeof(int64_t) * (n + 1)) * 2 + sk_align256(sizeof(int32_t) * (n + 1)) +
                        sk_align256(sizeof(int64_t) * (n_cals + 1)) + sk_align256(n_bases) * 2 + sk_align256(n_hap) +
                        sk_align256(sizeof(sk_score_op) * n_ops) + sk_align256(sizeof(double) * n_cals) + 16 * 256;
    if (ar.reserve(need)) return 1;
    sk_align_batch d = *hb;
    d.max_read_len = maxL;
    d.max_hap_len = std::max(maxP, 1);
    hipStream_t st = ctx.stream;
#define UP(field, T, count)                                                                              \
    {                                                                                                    \
        T* p = ar.take<T>(count);                                                                        \
        if (count) SK_HIP(hipMemcpyAsync(p, hb->field, sizeof(T) * (count), hipMemcpyHostToDevice, st)); \
        d.field = p;                                                                                     \
    }
    UP(read_off, int64_t, size_t(n + 1));
    UP(read_code, uint8_t, size_t(n_bases));
    UP(read_qual, uint8_t, size_t(n_bases));
    UP(hap_off, int64_t, size_t(n + 1));
    UP(hap_code, uint8_t, size_t(n_hap));
    UP(cal_off, int32_t, size_t(n + 1));
    UP(op_off, int64_t, size_t(n_cals + 1));
    UP(ops, sk_score_op, size_t(n_ops));
#undef UP
    double* dout = ar.take<double>(n_cals);
    if (sk_score_alignments_dev(&d, dout, st)) return 1;
    SK_HIP(hipMemcpyAsync(out_lnp, dout, sizeof(double) * n_cals, hipMemcpyDeviceToHost, st));
    SK_HIP(hipStreamSynchronize(st));
    return 0;
}
