// sk_common.h -- internal declarations shared by the HIP translation units of libstrelka_amd.so (gfx950 only).
#pragma once

#include "strelka_amd.h"

#include "sk_rt.h"

#include <cstdint>
#include <cstring>
#include <string>

// ---------------------------------------------------------------------------------------------------------------------
// Device-resident tables.  Every value is computed ON THE HOST with the host libm, using the reference's own
// expressions (file:line on each builder in sk_context.hip), then uploaded: the kernels never evaluate a
// transcendental for a quantity the reference takes from a table or memoises, so those terms are bit-identical.
// ---------------------------------------------------------------------------------------------------------------------
enum { SK_NQ = 71, SK_NQ6 = 64, SK_HET_RES = 9 };

struct SkTables
{
    // hot path A (double): L/blt_util/qscore_cache.cpp:34-50, L/starling_common/starling_read_align_score.cpp:119,133-135
    double q2lncompe[SK_NQ + 1]; // ln(1-e_q)
    double q2mis[SK_NQ + 1];     // ln(e_q) + (-ln 3)
    double ln_quarter;           // std::log(0.25)            (:453)
    double ln_noncand;           // std::log(1e-5)            (:483)

    // hot path B germline (float): L/blt_common/position_snp_call_pprob_digt.cpp:43-46,345-354
    float g_eprob[SK_NQ6];  // (float) error_prob(q)          (adjust_joint_eprob.cpp:210)
    float g_v1[SK_NQ6];     // (float)(log(ce+(1.-ce)/3)+ln 1/2)
    float g_v2[SK_NQ6];     // (float) ln_comp_error_prob(q)
    float g_weight[SK_NQ6]; // (float)(ln0.75 - ln_error_prob(q))  (adjust_joint_eprob.cpp:116,123)
    float g_log_one_third;

    // hot path B somatic (float): L/applications/strelka/position_somatic_snv_strand_grid_lhood_cached.cpp
    float s_v0[SK_NQ6], s_v1[SK_NQ6], s_v2[SK_NQ6];           // :56-64
    float s_c0[SK_HET_RES][SK_NQ6], s_c1[SK_HET_RES][SK_NQ6]; // :104-110  (het grid)
    float t_c0[SK_HET_RES][SK_NQ6], t_c1[SK_HET_RES][SK_NQ6]; // :197-206  (strand states, on-strand)
    float t_off_ref[SK_NQ6];                                  // (float) ln_comp_error_prob(q)            (:213)
    float t_off_alt[SK_NQ6];                                  // (float)(ln_error_prob(q) + ln_one_third) (:221)
    float s_ln_one_half;

    // a8: qphred_cache::mappedq[mapq 0..90][q 0..70] (qscore_cache.cpp:46-49, qscore.hh:104-113)
    uint8_t mappedq[91][SK_NQ + 1];
};

struct SkContext
{
    int device = -1;
    bool ready = false;
    bool libm_restated = false; // the host libm's powf/logf match csrc/libm_flt32.h (checked at sk_init)
    bool blocking_sync = false; // hipDeviceScheduleBlockingSync took effect on this process's device (sk_sync_mode)
    SkTables host_tables;
    SkTables* dev_tables = nullptr;
    hipStream_t stream = nullptr; // used by the host-buffer entry points
    // staging arena for the host-buffer entry points (grown on demand, never shrunk)
    void* arena = nullptr;
    size_t arena_bytes = 0;
    // sticky error bits raised by kernels on input the reference would have thrown on (sk_check_device_errors)
    unsigned* dev_error_flags = nullptr;
    // per-kernel attributes raised on this context's device (cleared with the context: another device starts from the defaults)
    size_t global_align_lds_allowed = 0;
    bool inflate_scalar_lds_allowed = false;
};
enum { SK_DEVERR_QSCORE = 1u }; // a basecall quality above 70 reached a scoring kernel (qscore_cache.cpp:53-75 throws)

SkContext& sk_ctx();
void sk_set_error(const std::string& msg);
int sk_fail(const std::string& msg);

#define SK_HIP(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return sk_fail(std::string(#expr) + ": " + skrt::errorString(_e));              \
    } while (0)

#define SK_REQUIRE_INIT()                                                                    \
    do {                                                                                     \
        if (!sk_ctx().ready) return sk_fail("strelka_amd: sk_init() has not succeeded");     \
    } while (0)

// simple bump allocator over the context arena (host-buffer entry points)
struct SkArena
{
    char* base = nullptr;
    size_t cap = 0, used = 0;
    int reserve(size_t bytes);
    template <typename T> T* take(size_t n)
    {
        size_t off = (used + 255) & ~size_t(255);
        used = off + n * sizeof(T);
        return reinterpret_cast<T*>(base + off);
    }
};

static inline size_t sk_align256(size_t n) { return (n + 255) & ~size_t(255); }

// The small host-buffer entry points (one indel group, one haplotype, a few loci per call: the adapter's per-unit sites) move their
// arrays as ONE page-locked block in and ONE out, carried by two launches in the process's own queue.  A copy call per array from the
// caller's pageable memory is a synchronous staged copy each (~10 us, ten of them per call), and with several caller processes on a
// device every copy call queues behind the others' (profiles/r05_enum_job_history.txt: the realignment job's way in and out).
//   SkStage st; st.begin(in_bytes, out_bytes);  T* d = st.put(host_array, n); ... U* dout = st.out<U>(n);
//   st.upload(stream); <launch kernels on d..., dout>; st.download_and_wait(stream); st.fetch(host_dst, dout, n);
struct SkStage
{
    SkArena ar;                 // device side: [in block | out block | whatever the caller takes after begin()]
    char* h_in = nullptr;       // page-locked mirrors (the context's, grown on demand)
    char* h_out = nullptr;
    size_t in_cap = 0, out_cap = 0, in_used = 0, out_used = 0;
    char* d_in = nullptr;
    char* d_out = nullptr;
    hipStream_t in_flight = nullptr; // the stream of an upload() that no download_and_wait() has followed yet
    /// An entry point that fails between upload() and download_and_wait() leaves kernels running over the page-locked mirrors, which
    /// the next call refills or frees: the stage waits for them before it goes.
    ~SkStage()
    {
        if (in_flight) (void)skrt::streamSynchronize(in_flight);
    }
    /// room for `in_bytes` of inputs and `out_bytes` of outputs (sums of the arrays' sizes; alignment padding is added here) plus
    /// `extra_bytes` of device scratch the caller takes from `ar` afterwards
    int begin(size_t in_bytes, size_t out_bytes, size_t extra_bytes, int n_arrays);
    template <typename T> T* put(const T* src, const size_t n)
    {
        const size_t off = (in_used + 255) & ~size_t(255);
        in_used = off + n * sizeof(T);
        if (n && in_used <= in_cap) std::memcpy(h_in + off, src, n * sizeof(T)); // (past the room asked for: upload() reports it)
        return reinterpret_cast<T*>(d_in + off);
    }
    template <typename T> T* out(const size_t n)
    {
        const size_t off = (out_used + 255) & ~size_t(255);
        out_used = off + n * sizeof(T);
        return reinterpret_cast<T*>(d_out + off);
    }
    int upload(hipStream_t st);            // in block: page-locked mirror -> device (one launch)
    int download_and_wait(hipStream_t st); // out block: device -> page-locked mirror (one launch), then the wait
    template <typename T> void fetch(T* dst, const T* dev, const size_t n) const
    {
        if (n) std::memcpy(dst, h_out + (reinterpret_cast<const char*>(dev) - d_out), n * sizeof(T));
    }
};

// packed base_call accessors (device + host)
#define SKC_Q(c) ((unsigned)((c) & 0x3f))
#define SKC_BASE(c) ((unsigned)(((c) >> 6) & 0xf))
#define SKC_FWD(c) ((unsigned)(((c) >> 10) & 1))
#define SKC_NMM(c) ((unsigned)(((c) >> 11) & 1))
#define SKC_FILTER(c) ((unsigned)(((c) >> 12) & 1))
