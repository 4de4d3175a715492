// global_align.hip -- GlobalAligner<int>::align (L/alignment/GlobalAlignerImpl.hh:35-228) for batches of
// (haplotype, reference segment) pairs: the one genuine wavefront DP of the product.
//
// One 64-lane wave per problem.  The query is cut into strips of 64 symbols, one lane per query symbol; inside a strip the
// wave sweeps the reference with a skew of one column per lane, so that at every step the 64 lanes sit on one
// anti-diagonal of the (query x reference) matrix and all three recurrences have their inputs ready:
//     match (q,r)  <- (q-1,r-1)   the upper lane's value two steps ago      (kept from the previous DPP shuffle)
//     delete(q,r)  <- (q,  r-1)   the lane's own value one step ago
//     insert(q,r)  <- (q-1,r)     the upper lane's value one step ago       (one __shfl_up per state)
// The last lane of a strip leaves its row in LDS for lane 0 of the next strip.  Scores are plain int32 (the reference's
// ScoreType for this aligner), including its -10000 "bad" value that is added to like any other number; max3 keeps the
// reference's tie order (first of equals; L/alignment/AlignerBase.hh:75-95).  Back-pointers (3 x 2 bits per cell) are one
// byte per cell in anti-diagonal-major order -- the 64 lanes write 64 consecutive bytes per step -- in global scratch
// (L2-resident): keeping the whole matrix in LDS left one wave per CU, and the sweep is a chain of dependent shuffles
// that needs several waves per SIMD to hide.  The traceback start is picked exactly in the order the reference offers
// candidates (updateBacktrace, L/alignment/Alignment.hh); the walk back (SingleRefAlignerSharedImpl.hh:96-195) reads the
// pointers through a 64-anti-diagonal LDS window that the whole wave refills with coalesced loads (the walk only ever
// moves to lower anti-diagonals inside a strip); matches are expanded to '=' / 'X' (L/blt_util/align_path_impl.hh:36-86).
//
// Roofline: a latency-bound integer DP (~30 instructions per anti-diagonal step); bytes are negligible.

#include "sk_common.h"

#include <algorithm>
#include <vector>

namespace
{

constexpr int WAVE = 64;
constexpr int BAD = -10000;
constexpr int MAX_LEN = 1024;
constexpr int WIN = 64; // anti-diagonals of one strip held in LDS during the traceback
enum { ST_MATCH = 0, ST_DELETE = 1, ST_INSERT = 2 };

struct GaArgs
{
    sk_global_align_batch b;
    sk_align_scores sc;
    int32_t* out_score;
    int32_t* out_begin;
    sk_path_seg* out_path;
    sk_path_seg* tmp_path; // same layout as out_path: reversed raw segments
    int32_t* out_nseg;
    uint8_t* ptr_scratch;  // back-pointers
    const int64_t* ptr_off; // per-problem offsets into ptr_scratch, or NULL: problem p starts at p * ptr_stride
    int64_t ptr_stride;
    int max_ref;           // LDS sizing: boundary rows and the reference copy
    int max_query;         // LDS sizing: last-column scores
};

__device__ __forceinline__ unsigned max3(int& mx, const int v0, const int v1, const int v2)
{
    unsigned p = 0;
    mx = v0;
    if (v1 > v0) {
        mx = v1;
        p = 1;
    }
    if (v2 > mx) {
        mx = v2;
        p = 2;
    }
    return p;
}

struct Bt
{
    int max, state, q, r;
    bool init;
};
__device__ __forceinline__ void bt_update(Bt& bt, const int v, const int r, const int q, const int state)
{
    if (!bt.init || v > bt.max) {
        bt.max = v;
        bt.r = r;
        bt.q = q;
        bt.init = true;
        bt.state = state;
    }
}

__global__ __launch_bounds__(WAVE) void global_align_kernel(const GaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int p = blockIdx.x;
    const int64_t qo = a.b.query_off[p], ro = a.b.ref_off[p];
    const int Q = int(a.b.query_off[p + 1] - qo), R = int(a.b.ref_off[p + 1] - ro);
    const char* __restrict__ gq = a.b.query + qo;
    const char* __restrict__ gr = a.b.ref + ro;
    const sk_align_scores sc = a.sc;
    const bool allow_ins = sc.is_allow_edge_insertion != 0, req_del = sc.is_require_edge_deletion != 0;

    // LDS: [back-pointer window][reference copy][last-column match scores (Q+1)][two boundary rows of (R+1) x {m,d,i}]
    const int RW = a.max_ref + 1;
    uint8_t* s_win = smem;
    char* s_ref = reinterpret_cast<char*>(smem + WIN * WAVE);
    int* s_last = reinterpret_cast<int*>(smem + WIN * WAVE + ((RW + 3) & ~3));
    int* s_rowA = s_last + (a.max_query + 1);
    int* s_rowB = s_rowA + 3 * RW;
    const int strips = (Q + WAVE - 1) / WAVE;
    uint8_t* ptr = a.ptr_scratch + (a.ptr_off ? a.ptr_off[p] : int64_t(p) * a.ptr_stride);

    for (int i = lane; i < R; i += WAVE) s_ref[i] = gr[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // the reference's candidates for the traceback start that are found during the sweep:
    Bt col_bt = { 0, ST_MATCH, 0, 0, false }; // (Q, r) for every column r, kept by the lane that owns query Q
    int fin_m = BAD, fin_d = BAD, fin_i = BAD; // scores of (Q, R)

    for (int s = 0; s < strips; ++s) {
        const int q = s * WAVE + lane + 1; // 1-based query index of this lane's row
        const bool row_ok = (q <= Q);
        const char qc = row_ok ? gq[q - 1] : char(0);
        int* row_in = (s & 1) ? s_rowB : s_rowA;  // row (64 s) written by the previous strip
        int* row_out = (s & 1) ? s_rowA : s_rowB; // row (64 s + 64) for the next strip
        // own previous column (q, r-1); starts with column 0 (:66-83)
        int lm = q * sc.off_edge, ld = BAD, li = allow_ins ? sc.open + q * sc.extend : BAD;
        // upper neighbour (q-1, r-1): column 0 of row q-1
        int dm = (q - 1) * sc.off_edge, dd = BAD, di = allow_ins ? sc.open + (q - 1) * sc.extend : BAD;
        int cm = 0, cd = 0, ci = 0; // this lane's newest cell
        uint8_t* sp = ptr + int64_t(s) * (R + WAVE) * WAVE;
        const int nstep = R + WAVE - 1;
        for (int t = 1; t <= nstep; ++t) {
            const int r = t - lane; // 1-based reference index of this lane's cell at this step
            // upper neighbour (q-1, r): lane-1's newest cell; lane 0 reads the boundary row
            int um = __shfl_up(cm, 1, WAVE), ud = __shfl_up(cd, 1, WAVE), ui = __shfl_up(ci, 1, WAVE);
            if (lane == 0) {
                if (s == 0) { // row 0 (:98-117)
                    um = req_del ? BAD : 0;
                    ud = req_del ? sc.open + r * sc.extend : BAD;
                    ui = BAD;
                } else if (r >= 1 && r <= R) {
                    um = row_in[3 * r + 0];
                    ud = row_in[3 * r + 1];
                    ui = row_in[3 * r + 2];
                }
            }
            const bool on = row_ok && r >= 1 && r <= R;
            if (on) {
                int mx;
                const unsigned pm = max3(mx, dm, dd, di);
                cm = mx + ((qc == s_ref[r - 1]) ? sc.match : sc.mismatch);
                const unsigned pd = max3(mx, lm + sc.open, ld, li + sc.insert_delete);
                cd = (r == 1) ? BAD : mx + sc.extend;
                const unsigned pi = max3(mx, um + sc.open, BAD, ui);
                ci = (q == 1) ? BAD : mx + sc.extend;
                sp[int64_t(t) * WAVE + lane] = uint8_t(pm | (pd << 2) | (pi << 4));
                if (q == Q) {
                    if (!req_del) bt_update(col_bt, cm, r, Q, ST_MATCH); // (:206-211)
                    if (r == R) {
                        fin_m = cm;
                        fin_d = cd;
                        fin_i = ci;
                    }
                }
                if (r == R) s_last[q] = cm;
                if (lane == WAVE - 1) {
                    row_out[3 * r + 0] = cm;
                    row_out[3 * r + 1] = cd;
                    row_out[3 * r + 2] = ci;
                }
                lm = cm;
                ld = cd;
                li = ci;
            }
            // next step's diagonal neighbour is this step's upper neighbour; before the lane's first cell the diagonal
            // stays at column 0 of row q-1 (for row 0 that is cell (0,0), which belongs to column 0)
            if (r >= 1) {
                dm = um;
                dd = ud;
                di = ui;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }

    // hand the sweep's findings to lane 0
    const int owner = (Q - 1) % WAVE;
    const int b_max = __shfl(col_bt.max, owner, WAVE), b_r = __shfl(col_bt.r, owner, WAVE);
    const int b_init = __shfl(int(col_bt.init), owner, WAVE);
    fin_m = __shfl(fin_m, owner, WAVE);
    fin_d = __shfl(fin_d, owner, WAVE);
    fin_i = __shfl(fin_i, owner, WAVE);
    // every lane walks the (wave-uniform) traceback so that the wave can refill the pointer window together; lane 0 writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

    Bt bt = { b_max, ST_MATCH, Q, b_r, b_init != 0 };
    if (req_del) { // (:214-220)
        bt_update(bt, fin_m, R, Q, ST_MATCH);
        bt_update(bt, fin_d, R, Q, ST_DELETE);
    }
    if (allow_ins) bt_update(bt, fin_i, R, Q, ST_INSERT); // (:222-227)
    s_last[0] = req_del ? BAD : 0;                          // row 0 at the last column
    for (int q = 0; q < Q; ++q) bt_update(bt, s_last[q] + (Q - q) * sc.off_edge, R, q, ST_MATCH); // (:229-235)

    if (lane == 0) a.out_score[p] = bt.max;
    const int64_t po = qo + ro + 4 * int64_t(p);
    sk_path_seg* rev = a.tmp_path + po;
    int nrev = 0;
    uint32_t ps_type = SK_SEG_NONE, ps_len = 0;
    if (bt.q < Q) {
        ps_type = SK_SEG_SOFT_CLIP;
        ps_len = uint32_t(Q - bt.q);
    }
    auto update_path = [&](const uint32_t atype) { // AlignerUtil::updatePath
        if (ps_type == atype) return;
        if (ps_type != SK_SEG_NONE) {
            if (lane == 0) rev[nrev] = sk_path_seg{ ps_type, ps_len };
            nrev++;
        }
        ps_type = atype;
        ps_len = 0;
    };
    int win_s = -1, win_lo = 0; // the window holds anti-diagonals [win_lo, win_lo + WIN) of strip win_s
    auto state_ptr = [&](const int q, const int r, const int state) -> int {
        if (r == 0) return (state == ST_INSERT && allow_ins) ? ST_INSERT : ST_MATCH; // column 0 (:72-83)
        if (q == 0) return (state == ST_DELETE && req_del) ? ST_DELETE : ST_MATCH;   // row 0 (:101-116)
        const int s = (q - 1) / WAVE, j = (q - 1) % WAVE, t = r + j;
        if (s != win_s || t < win_lo || t >= win_lo + WIN) {
            win_s = s;
            win_lo = max(0, t - (WIN - 1));
            const uint4* __restrict__ src = reinterpret_cast<const uint4*>(ptr + (int64_t(s) * (R + WAVE) + win_lo) * WAVE);
            uint4* dst = reinterpret_cast<uint4*>(s_win);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < WIN * WAVE / 16 / WAVE; ++k) dst[k * WAVE + lane] = src[k * WAVE + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const unsigned v = s_win[(t - win_lo) * WAVE + j];
        return int((v >> (2 * state)) & 3u);
    };
    for (;;) {
        const int next = state_ptr(bt.q, bt.r, bt.state);
        if (bt.state == ST_MATCH) {
            if (bt.q < 1 || bt.r < 1) break;
            update_path(SK_SEG_MATCH);
            bt.q--;
            bt.r--;
        } else if (bt.state == ST_DELETE) {
            if (bt.r < 1) break;
            update_path(SK_SEG_DELETE);
            bt.r--;
        } else {
            if (bt.q < 1) break;
            update_path(SK_SEG_INSERT);
            bt.q--;
        }
        bt.state = next;
        ps_len++;
    }
    if (lane != 0) return;
    if (ps_type != SK_SEG_NONE) rev[nrev++] = sk_path_seg{ ps_type, ps_len };
    if (bt.q != 0) rev[nrev++] = sk_path_seg{ SK_SEG_SOFT_CLIP, uint32_t(bt.q) };
    a.out_begin[p] = bt.r;

    // reverse + apath_add_seqmatch
    sk_path_seg* out = a.out_path + po;
    int n_out = 0, qi = 0, ri = bt.r;
    for (int k = nrev - 1; k >= 0; --k) {
        const uint32_t t = rev[k].type, len = rev[k].length;
        if (t == SK_SEG_MATCH) {
            for (uint32_t j = 0; j < len; ++j) {
                const char x = gq[qi], y = s_ref[ri];
                const bool same = (x == y) && x != 'N' && y != 'N';
                const uint32_t st = same ? SK_SEG_SEQ_MATCH : SK_SEG_SEQ_MISMATCH;
                if (n_out > 0 && out[n_out - 1].type == st) out[n_out - 1].length++;
                else out[n_out++] = sk_path_seg{ st, 1u };
                ++qi;
                ++ri;
            }
        } else {
            out[n_out++] = sk_path_seg{ t, len };
            if (t == SK_SEG_INSERT || t == SK_SEG_SOFT_CLIP) qi += int(len);
            if (t == SK_SEG_DELETE) ri += int(len);
        }
    }
    a.out_nseg[p] = n_out;
}

} // namespace

extern "C" {

void sk_align_scores_default(sk_align_scores* s)
{
    s->match = 1;
    s->mismatch = -4;
    s->open = -5;
    s->extend = -1;
    s->off_edge = -100;
    s->insert_delete = -5;
    s->is_allow_edge_insertion = 1;
    s->is_require_edge_deletion = 1;
}

// the dynamic LDS a launch may ask for is an attribute of the kernel, not of the launch: raised when a call needs more than any before
// (the adapter calls once per haplotype: the driver call every time was a tenth of the call)
static int ga_allow_lds(const size_t lds)
{
    size_t& allowed = sk_ctx().global_align_lds_allowed; // (of the context: the attribute belongs to the device sk_init chose)
    if (lds > allowed) {
        SK_HIP(skrt::funcSetAttribute(reinterpret_cast<const void*>(global_align_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        allowed = lds;
    }
    return 0;
}

int sk_global_align(const sk_global_align_batch* hb, const sk_align_scores* sc, int32_t* out_score, int32_t* out_begin_pos,
                    sk_path_seg* out_path, int32_t* out_n_seg)
{
    SK_REQUIRE_INIT();
    if (!hb || !sc || !out_score || !out_begin_pos || !out_path || !out_n_seg) return sk_fail("sk_global_align: null argument");
    const int n = hb->n;
    if (n < 0) return sk_fail("sk_global_align: negative n");
    if (n == 0) return 0;
    if (hb->query_off[0] != 0 || hb->ref_off[0] != 0) return sk_fail("sk_global_align: CSR offsets must start at 0");
    int maxQ = 0, maxR = 0;
    std::vector<int64_t> ptr_off(size_t(n) + 1, 0);
    for (int p = 0; p < n; ++p) {
        const int64_t Q = hb->query_off[p + 1] - hb->query_off[p], R = hb->ref_off[p + 1] - hb->ref_off[p];
        if (Q < 1 || R < 1 || Q > MAX_LEN || R > MAX_LEN)
            return sk_fail("sk_global_align: query and reference lengths must be in 1..1024");
        maxQ = std::max(maxQ, int(Q));
        maxR = std::max(maxR, int(R));
        ptr_off[size_t(p) + 1] = ptr_off[size_t(p)] + ((Q + WAVE - 1) / WAVE) * (R + WAVE) * WAVE;
    }
    const int64_t nq = hb->query_off[n], nr = hb->ref_off[n], npath = nq + nr + 4 * int64_t(n);
    // LDS: pointer window, reference copy, last column, two boundary rows
    const int RW = maxR + 1;
    const size_t lds = size_t(WIN) * WAVE + size_t((RW + 3) & ~3) + 4 * size_t(maxQ + 1) + 2 * 12 * size_t(RW) + 16;

    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    // one block in, one block out, moved by launches (SkStage, sk_common.h): the adapter calls this once per haplotype
    SkStage sg;
    const size_t in_bytes = 3 * 8 * (size_t(n) + 1) + size_t(nq) + size_t(nr);
    const size_t out_bytes = 8 * size_t(npath) + 3 * 4 * size_t(n);
    const size_t extra = sk_align256(8 * size_t(npath)) + sk_align256(size_t(ptr_off[size_t(n)]) + 16 * WIN * WAVE) + 8 * 256;
    if (sg.begin(in_bytes, out_bytes, extra, 9)) return 1;
    hipStream_t st = ctx.stream;
    GaArgs a;
    a.b.n = n;
    a.b.query_off = sg.put(hb->query_off, size_t(n) + 1);
    a.b.query = sg.put(hb->query, size_t(nq));
    a.b.ref_off = sg.put(hb->ref_off, size_t(n) + 1);
    a.b.ref = sg.put(hb->ref, size_t(nr));
    a.ptr_off = sg.put(ptr_off.data(), size_t(n) + 1);
    a.sc = *sc;
    a.out_path = sg.out<sk_path_seg>(size_t(npath));
    a.out_score = sg.out<int32_t>(size_t(n));
    a.out_begin = sg.out<int32_t>(size_t(n));
    a.out_nseg = sg.out<int32_t>(size_t(n));
    a.tmp_path = sg.ar.take<sk_path_seg>(size_t(npath));
    a.ptr_scratch = sg.ar.take<uint8_t>(size_t(ptr_off[size_t(n)]) + size_t(WIN) * WAVE); // + slack: a window read may run past the end
    a.ptr_stride = 0;
    a.max_ref = maxR;
    a.max_query = maxQ;
    if (ga_allow_lds(lds)) return 1;
    if (sg.upload(st)) return 1;
    SK_LAUNCH(global_align_kernel, dim3(n), dim3(WAVE), lds, st, a);
    if (sg.download_and_wait(st)) return 1;
    sg.fetch(out_score, a.out_score, size_t(n));
    sg.fetch(out_begin_pos, a.out_begin, size_t(n));
    sg.fetch(out_n_seg, a.out_nseg, size_t(n));
    sg.fetch(out_path, a.out_path, size_t(npath));
    return 0;
}

static int64_t ga_ptr_stride(const int max_query_len, const int max_ref_len)
{
    return int64_t((max_query_len + WAVE - 1) / WAVE) * (max_ref_len + WAVE) * WAVE;
}

size_t sk_global_align_scratch_bytes(int32_t n, int64_t total_query_len, int64_t total_ref_len, int32_t max_query_len,
                                     int32_t max_ref_len)
{
    if (n <= 0) return 256;
    const int64_t npath = total_query_len + total_ref_len + 4 * int64_t(n);
    return sk_align256(8 * size_t(npath)) + sk_align256(size_t(n) * size_t(ga_ptr_stride(max_query_len, max_ref_len)) + size_t(WIN) * WAVE) + 512;
}

int sk_global_align_dev(const sk_global_align_batch* db, int64_t total_query_len, int64_t total_ref_len, int32_t max_query_len,
                        int32_t max_ref_len, const sk_align_scores* sc, int32_t* dev_out_score, int32_t* dev_out_begin_pos,
                        sk_path_seg* dev_out_path, int32_t* dev_out_n_seg, void* dev_scratch, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!db || !sc || !dev_out_score || !dev_out_begin_pos || !dev_out_path || !dev_out_n_seg || !dev_scratch)
        return sk_fail("sk_global_align_dev: null argument");
    if (db->n < 0) return sk_fail("sk_global_align_dev: negative n");
    if (db->n == 0) return 0;
    if (max_query_len < 1 || max_ref_len < 1 || max_query_len > MAX_LEN || max_ref_len > MAX_LEN)
        return sk_fail("sk_global_align_dev: query and reference lengths must be in 1..1024");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int64_t npath = total_query_len + total_ref_len + 4 * int64_t(db->n);
    GaArgs a;
    a.b = *db;
    a.sc = *sc;
    a.out_score = dev_out_score;
    a.out_begin = dev_out_begin_pos;
    a.out_path = dev_out_path;
    a.out_nseg = dev_out_n_seg;
    char* base = static_cast<char*>(dev_scratch);
    base += (256 - (reinterpret_cast<uintptr_t>(base) & 255)) & 255;
    a.tmp_path = reinterpret_cast<sk_path_seg*>(base);
    a.ptr_scratch = reinterpret_cast<uint8_t*>(base + sk_align256(8 * size_t(npath)));
    a.ptr_off = nullptr;
    a.ptr_stride = ga_ptr_stride(max_query_len, max_ref_len);
    a.max_ref = max_ref_len;
    a.max_query = max_query_len;
    const int RW = max_ref_len + 1;
    const size_t lds = size_t(WIN) * WAVE + size_t((RW + 3) & ~3) + 4 * size_t(max_query_len + 1) + 2 * 12 * size_t(RW) + 16;
    if (ga_allow_lds(lds)) return 1;
    SK_LAUNCH(global_align_kernel, dim3(db->n), dim3(WAVE), lds, st, a);
    SK_HIP(skrt::getLastError());
    return 0;
}

} // extern "C"
