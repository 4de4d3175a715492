// germline_site.hip -- hot path B (germline SNV):
//   G1 `dependent_eprob_kernel` : adjust_joint_eprob            (L/blt_common/adjust_joint_eprob.cpp:201-243)
//   G2 `site_digt_call_kernel`  : position_snp_call_pprob_digt  (L/blt_common/position_snp_call_pprob_digt.cpp:473-539)
//
// One thread per locus; each of the 10 genotype likelihoods is a sequential float32 sum over the locus' calls in pileup
// order, exactly as in the reference (float accumulation order is part of the result).  Terms that the reference derives
// from the q-score alone come from the host-built tables; the only transcendental evaluated on the device per call is
// logf(de[i]) (and powf for the first few calls of each strand/base group in G1).
//
// G1 reproduces the reference's std::sort (libstdc++ introsort + final insertion sort, unstable) step for step on an
// index array, because WHICH of several equal-quality calls receives which exponent changes `de[]`.

#include "germline_common.h"

namespace
{

struct DepArgs
{
    sk_pileup_batch b;
    const SkTables* tab;
    float* out_de;
    uint32_t* scratch; // one uint32 per call
    GermlineDerived d;
};

__global__ __launch_bounds__(64) void dependent_eprob_kernel(const DepArgs a)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.b.n_loci) return;
    locus_dependent_eprob_global(a.b, a.tab, a.d, a.out_de, a.scratch, l);
}

// ---------------------------------------------------------------------------------------------------------------------

struct SiteArgs
{
    sk_pileup_batch b;
    const SkTables* tab;
    sk_digt_call* out;
    GermlineDerived d;
};

__global__ __launch_bounds__(64) void site_digt_call_kernel(const SiteArgs a)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.b.n_loci) return;
    locus_site_digt_call_global(a.b, a.b.de, a.tab, a.d, a.out, l);
}

int upload_pileup(const sk_pileup_batch* hb, bool need_de, SkArena& ar, size_t extra_bytes, sk_pileup_batch& d,
                  hipStream_t st, int64_t& total_calls)
{
    const int n = hb->n_loci;
    if (hb->call_off[0] != 0) return sk_fail("pileup batch: call_off must start at 0");
    total_calls = hb->call_off[n];
    for (int l = 0; l < n; ++l)
        if (hb->call_off[l + 1] < hb->call_off[l]) return sk_fail("pileup batch: call_off not monotone");
    const size_t need = sk_align256(sizeof(int64_t) * (n + 1)) + sk_align256(2 * total_calls) +
                        sk_align256(4 * total_calls) + sk_align256(n) * 2 + extra_bytes + 16 * 256;
    if (ar.reserve(need)) return 1;
    d = *hb;
    int64_t* off = ar.take<int64_t>(n + 1);
    SK_HIP(skrt::memcpyAsync(off, hb->call_off, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, st));
    d.call_off = off;
    uint16_t* calls = ar.take<uint16_t>(total_calls);
    if (total_calls) SK_HIP(skrt::memcpyAsync(calls, hb->calls, 2 * total_calls, hipMemcpyHostToDevice, st));
    d.calls = calls;
    d.de = nullptr;
    if (need_de) {
        if (!hb->de) return sk_fail("pileup batch: de is required");
        float* de = ar.take<float>(total_calls);
        if (total_calls) SK_HIP(skrt::memcpyAsync(de, hb->de, 4 * total_calls, hipMemcpyHostToDevice, st));
        d.de = de;
    }
    uint8_t* rb = ar.take<uint8_t>(n);
    SK_HIP(skrt::memcpyAsync(rb, hb->ref_base, n, hipMemcpyHostToDevice, st));
    d.ref_base = rb;
    d.ploidy = nullptr;
    if (hb->ploidy) {
        uint8_t* pl = ar.take<uint8_t>(n);
        SK_HIP(skrt::memcpyAsync(pl, hb->ploidy, n, hipMemcpyHostToDevice, st));
        d.ploidy = pl;
    }
    return 0;
}

} // namespace

int sk_upload_pileup_internal(const sk_pileup_batch* hb, bool need_de, SkArena& ar, size_t extra_bytes,
                              sk_pileup_batch& d, hipStream_t st, int64_t& total_calls)
{
    return upload_pileup(hb, need_de, ar, extra_bytes, d, st, total_calls);
}

extern "C" {

int sk_dependent_eprob_dev(const sk_pileup_batch* b, const sk_germline_options* opt, float* dev_out_de,
                           void* dev_scratch, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!b || !opt || !dev_out_de || !dev_scratch) return sk_fail("sk_dependent_eprob_dev: null argument");
    if (b->n_loci <= 0) return 0;
    DepArgs a;
    a.b = *b;
    a.tab = sk_ctx().dev_tables;
    a.out_de = dev_out_de;
    a.scratch = static_cast<uint32_t*>(dev_scratch);
    derive(*opt, a.d);
    const int threads = 64;
    SK_LAUNCH(dependent_eprob_kernel, dim3((b->n_loci + threads - 1) / threads), dim3(threads), 0,
                       static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(skrt::getLastError());
    return 0;
}

int sk_dependent_eprob(const sk_pileup_batch* hb, const sk_germline_options* opt, float* out_de)
{
    SK_REQUIRE_INIT();
    if (!hb || !opt || !out_de) return sk_fail("sk_dependent_eprob: null argument");
    if (hb->n_loci <= 0) return 0;
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    for (int64_t i = 0, e = hb->call_off[hb->n_loci]; i < e; ++i)
        if (SKC_BASE(hb->calls[i]) > 3) return sk_fail("sk_dependent_eprob: basecall with base_id > 3 in cleaned pileup");
    SkArena ar;
    sk_pileup_batch d;
    int64_t total = 0;
    const int64_t tc = hb->call_off[hb->n_loci];
    if (upload_pileup(hb, false, ar, sk_align256(4 * tc) * 2 + 512, d, ctx.stream, total)) return 1;
    float* dde = ar.take<float>(total);
    uint32_t* scratch = ar.take<uint32_t>(total);
    if (sk_dependent_eprob_dev(&d, opt, dde, scratch, ctx.stream)) return 1;
    if (total) SK_HIP(skrt::memcpyAsync(out_de, dde, 4 * total, hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(skrt::streamSynchronize(ctx.stream));
    return 0;
}

int sk_site_digt_call_dev(const sk_pileup_batch* b, const sk_germline_options* opt, sk_digt_call* dev_out,
                          void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!b || !opt || !dev_out) return sk_fail("sk_site_digt_call_dev: null argument");
    if (!b->de) return sk_fail("sk_site_digt_call_dev: batch.de is required");
    if (b->n_loci <= 0) return 0;
    SiteArgs a;
    a.b = *b;
    a.tab = sk_ctx().dev_tables;
    a.out = dev_out;
    derive(*opt, a.d);
    const int threads = 64;
    SK_LAUNCH(site_digt_call_kernel, dim3((b->n_loci + threads - 1) / threads), dim3(threads), 0,
                       static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(skrt::getLastError());
    return 0;
}

int sk_site_digt_call(const sk_pileup_batch* hb, const sk_germline_options* opt, sk_digt_call* out)
{
    SK_REQUIRE_INIT();
    if (!hb || !opt || !out) return sk_fail("sk_site_digt_call: null argument");
    if (hb->n_loci <= 0) return 0;
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    for (int64_t i = 0, e = hb->call_off[hb->n_loci]; i < e; ++i)
        if (SKC_BASE(hb->calls[i]) > 3) return sk_fail("sk_site_digt_call: basecall with base_id > 3 in cleaned pileup");
    SkArena ar;
    sk_pileup_batch d;
    int64_t total = 0;
    if (upload_pileup(hb, true, ar, sk_align256(sizeof(sk_digt_call) * hb->n_loci) + 512, d, ctx.stream, total)) return 1;
    sk_digt_call* dout = ar.take<sk_digt_call>(hb->n_loci);
    if (sk_site_digt_call_dev(&d, opt, dout, ctx.stream)) return 1;
    SK_HIP(skrt::memcpyAsync(out, dout, sizeof(sk_digt_call) * hb->n_loci, hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(skrt::streamSynchronize(ctx.stream));
    return 0;
}

} // extern "C"
