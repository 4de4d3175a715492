// gvcf_block.hip -- the gVCF writer's non-variant block logic as a kernel (csrc/gvcf_block_core.h has the statement and the reference
// lines): the stretches between hard block ends are independent, one lane walks each.

#include "sk_common.h"

#include "gvcf_block_core.h"

namespace
{

struct GvcfArgs
{
    const sk_gvcf_site* sites;
    int32_t n;
    double frac_tol;
    int32_t abs_tol;
    uint8_t* kind;
    sk_gvcf_block* blocks;
};

// one lane per site; the lane of a stretch's first site walks the stretch (its sites are contiguous: the wave's other lanes are done)
__global__ __launch_bounds__(256) void gvcf_block_kernel(const GvcfArgs a)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (!skgvcf::starts_stretch(a.sites, i)) return;
    skgvcf::walk_stretch(a.sites, a.n, i, a.frac_tol, a.abs_tol, a.kind, a.blocks);
}

struct GvcfBuffers
{
    void* p[3] = { nullptr, nullptr, nullptr };
    size_t cap[3] = { 0, 0, 0 };
    int reserve(const int i, const size_t bytes)
    {
        if (bytes <= cap[i]) return 0;
        if (p[i]) (void)hipFree(p[i]);
        p[i] = nullptr;
        cap[i] = 0;
        SK_HIP(hipMalloc(&p[i], bytes + bytes / 4 + 256));
        cap[i] = bytes + bytes / 4 + 256;
        return 0;
    }
};
GvcfBuffers& gvcf_bufs()
{
    static GvcfBuffers b;
    return b;
}

} // namespace

extern "C" {

int sk_gvcf_block_sites_dev(const sk_gvcf_site* dev_sites, int32_t n_sites, uint32_t block_percent_tol, uint32_t block_abs_tol, uint8_t* dev_kind,
                            sk_gvcf_block* dev_blocks, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (n_sites < 0) return sk_fail("sk_gvcf_block_sites_dev: negative count");
    if (n_sites == 0) return 0;
    if (!dev_sites || !dev_kind || !dev_blocks) return sk_fail("sk_gvcf_block_sites_dev: null argument");
    GvcfArgs a;
    a.sites = dev_sites;
    a.n = n_sites;
    a.frac_tol = static_cast<double>(block_percent_tol) / 100.; // gvcf_block_site_record.hh:41
    a.abs_tol = int32_t(block_abs_tol);
    a.kind = dev_kind;
    a.blocks = dev_blocks;
    hipLaunchKernelGGL(gvcf_block_kernel, dim3((n_sites + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(hipGetLastError());
    return 0;
}

int sk_gvcf_block_sites(const sk_gvcf_site* sites, int32_t n_sites, uint32_t block_percent_tol, uint32_t block_abs_tol, uint8_t* kind,
                        sk_gvcf_block* blocks)
{
    SK_REQUIRE_INIT();
    if (n_sites < 0) return sk_fail("sk_gvcf_block_sites: negative count");
    if (n_sites == 0) return 0;
    if (!sites || !kind || !blocks) return sk_fail("sk_gvcf_block_sites: null argument");
    SkContext& ctx = sk_ctx();
    SK_HIP(hipSetDevice(ctx.device));
    hipStream_t st = ctx.stream;
    GvcfBuffers& B = gvcf_bufs();
    if (B.reserve(0, sizeof(sk_gvcf_site) * size_t(n_sites)) || B.reserve(1, size_t(n_sites)) || B.reserve(2, sizeof(sk_gvcf_block) * size_t(n_sites))) return 1;
    SK_HIP(hipMemcpyAsync(B.p[0], sites, sizeof(sk_gvcf_site) * size_t(n_sites), hipMemcpyHostToDevice, st));
    SK_HIP(hipMemsetAsync(B.p[2], 0, sizeof(sk_gvcf_block) * size_t(n_sites), st));
    if (sk_gvcf_block_sites_dev(static_cast<sk_gvcf_site*>(B.p[0]), n_sites, block_percent_tol, block_abs_tol, static_cast<uint8_t*>(B.p[1]),
                                static_cast<sk_gvcf_block*>(B.p[2]), st))
        return 1;
    SK_HIP(hipMemcpyAsync(kind, B.p[1], size_t(n_sites), hipMemcpyDeviceToHost, st));
    SK_HIP(hipMemcpyAsync(blocks, B.p[2], sizeof(sk_gvcf_block) * size_t(n_sites), hipMemcpyDeviceToHost, st));
    SK_HIP(hipStreamSynchronize(st));
    return 0;
}

} // extern "C"
