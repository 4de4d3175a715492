// sk_context.hip -- lifecycle, error channel and host-built tables of libstrelka_amd.so.
//
// The tables are evaluated at run time with the host libm (the same glibc the reference links), never constant-folded
// by the compiler: inputs pass through `rt()` (a volatile round trip).

#include "sk_common.h"

#include <algorithm>
#include "libm_dbl64.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace
{
thread_local std::string g_last_error;
SkContext g_ctx;

inline double rt(double x)
{
    volatile double v = x;
    return v;
}
inline float rtf(float x)
{
    volatile float v = x;
    return v;
}

// log1p_switch, L/blt_util/math_util.hh:33-48
double log1p_switch(const double x)
{
    if (std::abs(x) < 0.01) return ::log1p(x);
    return std::log(1 + x);
}

void build_tables(SkTables& t)
{
    std::memset(&t, 0, sizeof(t));
    double q2p[SK_NQ], q2lne[SK_NQ];
    // qphred_cache::qphred_cache, L/blt_util/qscore_cache.cpp:34-50
    const double q2lnp(-std::log(rt(10.)) / 10.);
    const double lnthird(-std::log(rt(3.))); // starling_read_align_score.cpp:119
    for (int i = 0; i < SK_NQ; ++i) {
        q2p[i] = std::pow(rt(10.), -static_cast<double>(i) / 10.); // phred_to_error_prob, qscore.hh:77-82
        t.q2lncompe[i] = log1p_switch(-q2p[i]);
        q2lne[i] = static_cast<double>(i) * q2lnp;
        t.q2mis[i] = q2lne[i] + lnthird; // qphred_to_ln_error_prob(qscore)+lnthird (:135)
    }
    // mappedq[j][i] = error_prob_to_qphred(phred_to_mapped_error_prob(i, j)), qscore_cache.cpp:46-49:
    //   be = 10^(-i/10), me = 10^(-j/10), p = (1-me)*be + me*0.75 (qscore.hh:104-113); floor(-10*log10(p) + 0.5) (:40-66)
    for (int i = 0; i < SK_NQ; ++i)
        for (int j = 0; j <= 90; ++j) {
            const double be(std::pow(rt(10.), -static_cast<double>(i) / 10.));
            const double me(std::pow(rt(10.), -static_cast<double>(j) / 10.));
            const double p(((1. - me) * be) + (me * 0.75));
            t.mappedq[j][i] = static_cast<uint8_t>(static_cast<int>(std::floor((-10. * std::log10(p)) + 0.5)));
        }
    t.ln_quarter = std::log(rt(0.25));
    t.ln_noncand = std::log(rt(1e-5));

    // germline, position_snp_call_pprob_digt.cpp:43-46
    const float one_third(rt(1.) / 3.);
    const float log_one_third(std::log(rtf(one_third)));
    const float one_half(rt(1.) / 2.);
    const float log_one_half(std::log(rtf(one_half)));
    t.g_log_one_third = log_one_third;
    const float lnran(std::log(rt(0.75))); // adjust_joint_eprob.cpp:116
    for (int q = 0; q < SK_NQ6; ++q) {
        t.g_eprob[q] = static_cast<float>(q2p[q]);
        const float ceprob(1. - q2p[q]);                                                  // :347
        t.g_v1[q] = std::log((ceprob) + ((1. - ceprob) * one_third)) + log_one_half;      // :353 (double, narrowed)
        t.g_v2[q] = static_cast<float>(t.q2lncompe[q]);                                   // :348,354
        t.g_weight[q] = lnran - q2lne[q];                                                 // adjust_joint_eprob.cpp:123
    }

    // somatic, position_somatic_snv_strand_grid_lhood_cached.cpp:34-37
    const float ln_one_third(std::log(rtf(one_third)));
    const float ln_one_half(std::log(rtf(one_half)));
    t.s_ln_one_half = ln_one_half;
    const float RATIO_INCREMENT = 0.5f / static_cast<float>(SK_HET_RES + 1); // strelka_digt_states.hh:94
    for (int q = 0; q < SK_NQ6; ++q) {
        {
            // get_diploid_gt_lhood_cached_simple :53-64
            const float eprob(q2p[q]);
            const float ceprob(1 - eprob);
            const float lne(q2lne[q]);
            const float lnce(t.q2lncompe[q]);
            t.s_v0[q] = lne + ln_one_third;
            t.s_v1[q] = std::log((ceprob) + ((eprob)*one_third)) + ln_one_half;
            t.s_v2[q] = lnce;
        }
        for (unsigned r = 0; r < SK_HET_RES; ++r) {
            const float het_ratio((r + 1) * RATIO_INCREMENT);
            const float chet_ratio(1. - het_ratio);
            {
                // get_high_low_het_ratio_lhood_cached :94-110
                const float eprob(q2p[q]);
                const float ceprob(1 - eprob);
                t.s_c0[r][q] = std::log((ceprob)*het_ratio + ((eprob)*one_third) * chet_ratio);
                t.s_c1[r][q] = std::log((ceprob)*chet_ratio + ((eprob)*one_third) * het_ratio);
            }
            {
                // get_strand_ratio_lhood_spi :178-206
                const float eprob(q2p[q]);
                const float ceprob(1. - eprob);
                t.t_c0[r][q] = (std::log((ceprob)*chet_ratio + ((eprob)*one_third) * het_ratio));
                t.t_c1[r][q] = (std::log((ceprob)*het_ratio + ((eprob)*one_third) * chet_ratio));
            }
        }
        t.t_off_ref[q] = static_cast<float>(t.q2lncompe[q]);  // :213
        t.t_off_alt[q] = q2lne[q] + ln_one_third;             // :221 (double add, narrowed)
    }
}
} // namespace

SkContext& sk_ctx() { return g_ctx; }
void sk_set_error(const std::string& msg) { g_last_error = msg; }
int sk_fail(const std::string& msg)
{
    g_last_error = msg;
    return 1;
}

int SkArena::reserve(size_t bytes)
{
    SkContext& c = sk_ctx();
    bytes = sk_align256(bytes) + 4096;
    if (c.arena_bytes < bytes) {
        if (c.arena) (void)skrt::free_(c.arena);
        c.arena = nullptr;
        c.arena_bytes = 0;
        size_t want = bytes + bytes / 4;
        SK_HIP(skrt::malloc_(&c.arena, want));
        c.arena_bytes = want;
    }
    base = static_cast<char*>(c.arena);
    cap = c.arena_bytes;
    used = 0;
    return 0;
}

namespace
{
// (16-byte pieces, a grid-stride loop: the blocks are a few KB to a few hundred KB)
__global__ __launch_bounds__(256) void stage_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const uint32_t n16)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}
struct StageMirrors // the context's page-locked mirrors of SkStage's two blocks
{
    void* in = nullptr;
    void* out = nullptr;
    size_t in_cap = 0, out_cap = 0;
};
StageMirrors& stage_mirrors()
{
    static StageMirrors m;
    return m;
}
int grow_pinned(void*& p, size_t& cap, const size_t bytes)
{
    if (bytes <= cap) return 0;
    if (p) (void)skrt::hostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    SK_HIP(skrt::hostMalloc(&p, want));
    cap = want;
    return 0;
}
}

int SkStage::begin(const size_t in_bytes, const size_t out_bytes, const size_t extra_bytes, const int n_arrays)
{
    skrt::wakeHint(); // (the caller fills the in block next)
    const size_t pad = 256 * size_t(n_arrays + 2);
    const size_t in_room = sk_align256(in_bytes + pad), out_room = sk_align256(out_bytes + pad);
    StageMirrors& m = stage_mirrors();
    if (grow_pinned(m.in, m.in_cap, in_room) || grow_pinned(m.out, m.out_cap, out_room)) return 1;
    if (ar.reserve(in_room + out_room + extra_bytes + 256 * size_t(n_arrays + 2))) return 1;
    h_in = static_cast<char*>(m.in);
    h_out = static_cast<char*>(m.out);
    in_cap = in_room;
    out_cap = out_room;
    in_used = out_used = 0;
    d_in = ar.take<char>(in_room);
    d_out = ar.take<char>(out_room);
    return 0;
}

int SkStage::upload(hipStream_t st)
{
    if (in_used > in_cap || out_used > out_cap) return sk_fail("strelka_amd: a staged call's arrays outgrew the room asked for");
    const uint32_t n16 = uint32_t((in_used + 15) / 16);
    if (n16) SK_LAUNCH(stage_copy_kernel, dim3(std::min<uint32_t>((n16 + 255) / 256, 256u)), dim3(256), 0, st, reinterpret_cast<const uint4*>(h_in),
                                reinterpret_cast<uint4*>(d_in), n16);
    in_flight = st;
    return 0;
}

int SkStage::download_and_wait(hipStream_t st)
{
    const uint32_t n16 = uint32_t((out_used + 15) / 16);
    if (n16) SK_LAUNCH(stage_copy_kernel, dim3(std::min<uint32_t>((n16 + 255) / 256, 256u)), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_out),
                                reinterpret_cast<uint4*>(h_out), n16);
    in_flight = nullptr;
    SK_HIP(skrt::getLastError());
    SK_HIP(skrt::streamSynchronize(st));
    return 0;
}

// The kernels evaluate the reference's powf/logf calls with csrc/libm_flt32.h, a restatement of glibc's routines.  That is
// only "the same float as the reference" while the host's libm IS those routines: compare them on a spread of arguments
// (every q-score's error probability and neighbours, exponents across (0,1]); on any difference the kernels fall back to
// the device library's double-precision pow/log (agreement to the last ulp or two instead of bit for bit).
static bool host_libm_matches_restatement()
{
    unsigned long long st = 0x9e3779b97f4a7c15ull;
    for (int it = 0; it < 20000; ++it) {
        st ^= st << 13;
        st ^= st >> 7;
        st ^= st << 17;
        const int q = 3 + int(st % 68);
        float e = static_cast<float>(std::pow(10.0, -0.1 * q));
        e = sk_libm::as_f32(sk_libm::as_u32(e) + uint32_t((st >> 20) % 64));
        const float v = static_cast<float>(double((st >> 32) % 100000 + 1) / 100000.0);
        volatile float ve = e, vv = v;
        float mine;
        if (!sk_libm::powf_glibc(e, v, mine) || sk_libm::as_u32(mine) != sk_libm::as_u32(std::pow(ve, vv))) return false;
        volatile float x = sk_libm::as_f32(0x33000000u + uint32_t((st >> 8) % 0x0c800000u));
        if (!sk_libm::logf_glibc(x, mine) || sk_libm::as_u32(mine) != sk_libm::as_u32(std::log(x))) return false;
        volatile float xe = -static_cast<float>(double(st % 11000000) / 100000.0);
        if (!sk_libm::expf_glibc(xe, mine) || sk_libm::as_u32(mine) != sk_libm::as_u32(std::exp(xe))) return false;
        volatile float x1 = sk_libm::as_f32(0x20000000u + uint32_t((st >> 16) % (0x3ed413d7u - 0x20000000u)));
        if (!sk_libm::log1pf_glibc(x1, mine) || sk_libm::as_u32(mine) != sk_libm::as_u32(std::log1p(x1))) return false;
        double dm;
        volatile double xd = -double(st % 760000000) / 1.0e6;
        if (!sk_libm::exp_glibc(xd, dm) || sk_libm::as_u64(dm) != sk_libm::as_u64(std::exp(xd))) return false;
        volatile double xl = (it & 1) ? 0.9375 + double(st % 12720000) / 1.0e8
                                      : sk_libm::as_f64(0x0010000000000000ull + (st % 0x7fe0000000000000ull));
        if (!sk_libm::log_glibc(xl, dm) || sk_libm::as_u64(dm) != sk_libm::as_u64(std::log(xl))) return false;
        if (!sk_libm::log10_glibc(xl, dm) || sk_libm::as_u64(dm) != sk_libm::as_u64(std::log10(xl))) return false;
        volatile double xp = double(st % 1000000000) / 1.0e11;
        if (!sk_libm::log1p_glibc(xp, dm) || sk_libm::as_u64(dm) != sk_libm::as_u64(std::log1p(xp))) return false;
    }
    return true;
}

extern "C" {

int sk_version(void) { return SK_VERSION; }
int sk_libm_restated(void) { return g_ctx.libm_restated ? 1 : 0; }
const char* sk_last_error(void) { return g_last_error.c_str(); }
int sk_is_initialized(void) { return g_ctx.ready ? 1 : 0; }
int sk_sync_mode(void) { return g_ctx.ready ? (g_ctx.blocking_sync ? 1 : 0) : -1; }

// One hardware queue per process unless the caller says otherwise: this library runs everything on one stream, and its
// processes are many per GPU (one per genome segment) -- with HIP's default of four queues per process, sixteen caller
// processes oversubscribe the device's queue slots and every wait turns into a scheduler time slice
// (profiles/r03_v2_gpu_sharing.txt).  The HIP runtime reads the variable when it is first touched, so every entry point that
// can be a process's first HIP call (sk_device_count, sk_init) comes through here first.
static void sk_pre_runtime_env()
{
    (void)setenv("GPU_MAX_HW_QUEUES", "1", 0);
}

int sk_device_count(void)
{
    if (skrt::remote()) { // a broker client asks the broker of device 0 (started on demand)
        std::string why;
        const int n = skrt::r_device_count(&why);
        if (n <= 0) sk_set_error(why);
        return n;
    }
    sk_pre_runtime_env();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sk_broker_client(void) { return skrt::remote() ? 1 : 0; }

int sk_broker_enable(const int on)
{
    if (g_ctx.ready) return sk_fail("sk_broker_enable: decided before sk_init (this process already has its device side)");
    skrt::set_remote(on != 0);
    return 0;
}

void* sk_host_alloc(size_t bytes)
{
    if (!g_ctx.ready) {
        sk_fail("strelka_amd: sk_init() has not succeeded");
        return nullptr;
    }
    void* p = nullptr;
    const hipError_t e = skrt::hostMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        sk_fail(std::string("sk_host_alloc: ") + skrt::errorString(e));
        return nullptr;
    }
    return p;
}

void sk_host_free(void* p)
{
    if (p) (void)skrt::hostFree(p);
}

int sk_init(int device)
{
    SkContext& c = g_ctx;
    if (c.ready && c.device == device) {
        // (the current device is a property of the calling THREAD: a caller that initialised on a worker thread calls again from the
        // thread that will make the calls)
        SK_HIP(skrt::setDevice(device));
        return 0;
    }
    if (c.ready) sk_shutdown();
    sk_pre_runtime_env();
    const bool timing = std::getenv("SK_INIT_TIMING") != nullptr; // diagnostics: where a caller process's start-up goes
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (timing) std::fprintf(stderr, "[sk_init] %-34s t=%.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    if (skrt::remote()) {
        // a client of the device's broker: no GPU context in this process (sk_rt.h); the broker has checked the device
        std::string why;
        if (skrt::r_connect(device, &why)) return sk_fail(why);
        c.blocking_sync = true; // (a client's wait is a futex wait)
        lap("broker connection");
    } else {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return sk_fail(std::string("strelka_amd: no HIP device available (") + hipGetErrorString(e) +
                       "); this library has no CPU fallback");
    if (device < 0 || device >= n) return sk_fail("strelka_amd: device index out of range");
    lap("hipGetDeviceCount (runtime start)");
    SK_HIP(skrt::setDevice(device));
    // A process that waits for the device SLEEPS: sixteen caller processes share a GPU and a CPU quota, and a wait spent
    // spinning is a core taken from a process that has host work to do (profiles/r03_v5_thread_cpu_seconds.txt: under contention
    // the callers' CPU seconds doubled, all of it in their main threads).  The flags belong to the CURRENT device: set after
    // hipSetDevice, so that farm processes on devices 1..N-1 get them too.
    if (std::getenv("STRELKA_AMD_SPIN_WAIT") == nullptr) {
        (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
        unsigned flags = 0;
        if (hipGetDeviceFlags(&flags) == hipSuccess) c.blocking_sync = (flags & hipDeviceScheduleMask) == hipDeviceScheduleBlockingSync;
    }
    hipDeviceProp_t prop;
    SK_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return sk_fail(std::string("strelka_amd: built for gfx950 only, device is ") + prop.gcnArchName);
    lap("device properties");
    }
    build_tables(c.host_tables);
    lap("host tables");
    c.libm_restated = host_libm_matches_restatement();
    lap("libm comparison");
    SK_HIP(skrt::malloc_(reinterpret_cast<void**>(&c.dev_tables), sizeof(SkTables)));
    lap("first hipMalloc (context)");
    SK_HIP(skrt::memcpy_(c.dev_tables, &c.host_tables, sizeof(SkTables), hipMemcpyHostToDevice));
    if (skrt::remote()) c.stream = skrt::r_stream();
    else SK_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    SK_HIP(skrt::malloc_(reinterpret_cast<void**>(&c.dev_error_flags), sizeof(unsigned)));
    SK_HIP(skrt::memset_(c.dev_error_flags, 0, sizeof(unsigned)));
    lap("tables up, stream, flags");
    c.device = device;
    c.ready = true;
    return 0;
}

void sk_shutdown(void)
{
    SkContext& c = g_ctx;
    if (!c.ready) return;
    (void)skrt::setDevice(c.device);
    if (c.stream && !skrt::remote()) (void)hipStreamDestroy(c.stream);
    if (c.dev_tables) (void)skrt::free_(c.dev_tables);
    if (c.dev_error_flags) (void)skrt::free_(c.dev_error_flags);
    if (c.arena) (void)skrt::free_(c.arena);
    // (a broker client keeps its connection: the process's grown buffers -- the job's, the streams', the staged calls' -- stay valid
    // across sk_shutdown / sk_init as they do in a process with a context of its own; the broker frees them when the process ends)
    c = SkContext();
}

int sk_check_device_errors(void)
{
    SK_REQUIRE_INIT();
    SkContext& c = sk_ctx();
    SK_HIP(skrt::setDevice(c.device));
    unsigned flags = 0;
    SK_HIP(skrt::memcpy_(&flags, c.dev_error_flags, sizeof(unsigned), hipMemcpyDeviceToHost)); // synchronises the device
    if (flags == 0) return 0;
    SK_HIP(skrt::memset_(c.dev_error_flags, 0, sizeof(unsigned)));
    if (flags & SK_DEVERR_QSCORE)
        return sk_fail("Attempting to lookup basecall quality score which exceeds the maximum cached score of 70 (seen by a *_dev kernel)");
    return sk_fail("strelka_amd: a kernel reported an error");
}

int sk_debug_force_device_libm(int on)
{
    SK_REQUIRE_INIT();
    SkContext& c = sk_ctx();
    c.libm_restated = on ? false : host_libm_matches_restatement();
    return 0;
}

int sk_init_strict(int device)
{
    if (sk_init(device)) return 1;
    if (!sk_ctx().libm_restated) {
        sk_shutdown();
        return sk_fail("strelka_amd: the host C library is not the one the kernels restate (glibc >= 2.28 x86-64 FMA build "
                       "expected): results would agree with the reference only to 1e-5, not bit for bit");
    }
    return 0;
}

int sk_get_qscore_tables(double* q2p, double* q2lncompe, double* q2lne)
{
    // usable without a GPU: the tables are host-built
    SkTables t;
    build_tables(t);
    const double lnthird(-std::log(rt(3.)));
    const double q2lnp(-std::log(rt(10.)) / 10.);
    (void)lnthird;
    for (int i = 0; i < SK_NQ; ++i) {
        q2p[i] = std::pow(rt(10.), -static_cast<double>(i) / 10.);
        q2lncompe[i] = t.q2lncompe[i];
        q2lne[i] = static_cast<double>(i) * q2lnp;
    }
    return 0;
}

void sk_germline_options_default(sk_germline_options* opt)
{
    opt->bsnp_diploid_theta = 0.001;
    opt->bsnp_ssd_no_mismatch = 0.35;
    opt->bsnp_ssd_one_mismatch = 0.6;
    opt->is_min_vexp = 1;
    opt->min_vexp = 0.25;
}

void sk_somatic_snv_options_default(sk_somatic_snv_options* opt)
{
    opt->bsnp_diploid_theta = 0.001;
    opt->somatic_snv_rate = 1e-4;
    opt->shared_site_error_rate = 5e-10;
    opt->shared_site_error_strand_bias_fraction = 0.0;
    opt->ssnv_contam_tolerance = 0.15;
}

} // extern "C"
