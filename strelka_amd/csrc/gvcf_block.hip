// gvcf_block.hip -- the gVCF writer's non-variant block logic as a kernel (csrc/gvcf_block_core.h has the statement and the reference
// lines): the stretches between hard block ends are independent, one lane walks each.

#include "sk_common.h"

#include <cstring>
#include <vector>

#include "gvcf_block_core.h"
#include "gvcf_site_core.h"

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

// one lane per position of a stream window: its cleaned column (a few dozen 16-bit calls) and its genotype record -> sk_gvcf_site_summary
__global__ __launch_bounds__(256) void gvcf_site_summary_kernel(const sk_pileup_batch b, const sk_digt_call* __restrict__ geno, sk_gvcf_site_summary* __restrict__ out)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n_loci) return;
    const int64_t c0 = b.call_off[i], c1 = b.call_off[i + 1];
    out[i] = skgvcf::site_summary(b.calls + c0, c1 - c0, b.ref_base[i], b.ploidy ? b.ploidy[i] : 2u, geno[i]);
}

// what the join test reads of every site of a window (skgvcf::SitePod), from the summaries and the window's column sizes
struct PodArgs
{
    const sk_gvcf_site_summary* summary;
    const int64_t* clean_off; // [n + 1] the cleaned tier1 columns
    const int64_t* raw_off;   // [n + 1] the raw tier1 columns
    const uint32_t* mapq_count;
    sk_gvcf_block_options opt;
    int32_t n;
    skgvcf::SitePod* pod;
};
__global__ __launch_bounds__(256) void gvcf_site_pod_kernel(const PodArgs a)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const sk_gvcf_site_summary s = a.summary[i];
    skgvcf::SitePod p;
    p.gqx = s.gqx;
    p.used = uint32_t(a.clean_off[i + 1] - a.clean_off[i]);
    p.unused = uint32_t(a.raw_off[i + 1] - a.raw_off[i]) - p.used;
    p.key_plain = (s.flags & skgvcf::SITE_PLAIN) ? (uint32_t(skgvcf::POD_PLAIN) | skgvcf::site_filter_key(a.opt, s.gqx, p.used, p.unused, s.ref_fwd + s.ref_rev, a.mapq_count[i])) : 0u;
    a.pod[i] = p;
}
// a lane per tile of 32 sites: the tile's common key and extremes
__global__ __launch_bounds__(64) void gvcf_site_tile_kernel(const skgvcf::SitePod* __restrict__ pod, const int32_t n, skgvcf::SiteTile* __restrict__ tiles)
{
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t * skgvcf::TILE_SITES >= n) return;
    tiles[t] = skgvcf::make_tile(pod, n, t);
}
// a lane per site: the block that would start at it -- to the next tile boundary site by site, over whole tiles while they join, then
// site by site to its end (~len / 32 + 32 steps; the lanes of a wave walk neighbouring sites, whose blocks mostly end at the same place)
__global__ __launch_bounds__(64) void gvcf_plain_run_kernel(const skgvcf::SitePod* __restrict__ pod, const skgvcf::SiteTile* __restrict__ tiles, const int32_t n,
                                                            const double frac_tol, const int abs_tol, sk_gvcf_run* __restrict__ runs)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    runs[i] = skgvcf::plain_run(pod, tiles, n, i, frac_tol, abs_tol);
}

struct GvcfBuffers
{
    void* p[3] = { nullptr, nullptr, nullptr };
    size_t cap[3] = { 0, 0, 0 };
    int reserve(const int i, const size_t bytes)
    {
        if (bytes <= cap[i]) return 0;
        if (p[i]) (void)skrt::free_(p[i]);
        p[i] = nullptr;
        cap[i] = 0;
        SK_HIP(skrt::malloc_(&p[i], bytes + bytes / 4 + 256));
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

int sk_gvcf_site_summaries_dev(const sk_pileup_batch* dev_batch, const sk_digt_call* dev_genotypes, sk_gvcf_site_summary* dev_out, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!dev_batch || dev_batch->n_loci < 0) return sk_fail("sk_gvcf_site_summaries_dev: bad batch");
    if (dev_batch->n_loci == 0) return 0;
    if (!dev_genotypes || !dev_out) return sk_fail("sk_gvcf_site_summaries_dev: null argument");
    SK_LAUNCH(gvcf_site_summary_kernel, dim3((dev_batch->n_loci + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(hip_stream), *dev_batch, dev_genotypes,
                       dev_out);
    SK_HIP(skrt::getLastError());
    return 0;
}

int sk_gvcf_plain_runs_dev(const sk_gvcf_site_summary* dev_summary, const int64_t* dev_clean_off, const int64_t* dev_raw_off, const uint32_t* dev_mapq_count,
                           const sk_gvcf_block_options* opt, int32_t n, void* dev_pod_scratch, sk_gvcf_run* dev_runs, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (n < 0 || !opt) return sk_fail("sk_gvcf_plain_runs_dev: bad argument");
    if (n == 0) return 0;
    if (!dev_summary || !dev_clean_off || !dev_raw_off || !dev_mapq_count || !dev_pod_scratch || !dev_runs) return sk_fail("sk_gvcf_plain_runs_dev: null argument");
    PodArgs a;
    a.summary = dev_summary;
    a.clean_off = dev_clean_off;
    a.raw_off = dev_raw_off;
    a.mapq_count = dev_mapq_count;
    a.opt = *opt;
    a.n = n;
    a.pod = static_cast<skgvcf::SitePod*>(dev_pod_scratch);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    SK_LAUNCH(gvcf_site_pod_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a);
    // (the scratch holds the pods and, behind them, the tiles: 16 bytes per site + 32 per 32 sites)
    skgvcf::SiteTile* tiles = reinterpret_cast<skgvcf::SiteTile*>(static_cast<char*>(dev_pod_scratch) + ((size_t(n) * sizeof(skgvcf::SitePod) + 255) & ~size_t(255)));
    const int n_tiles = (n + skgvcf::TILE_SITES - 1) / skgvcf::TILE_SITES;
    SK_LAUNCH(gvcf_site_tile_kernel, dim3((n_tiles + 63) / 64), dim3(64), 0, st, a.pod, n, tiles);
    SK_LAUNCH(gvcf_plain_run_kernel, dim3((n + 63) / 64), dim3(64), 0, st, a.pod, tiles, n, static_cast<double>(opt->block_percent_tol) / 100.,
                       int(opt->block_abs_tol), dev_runs);
    SK_HIP(skrt::getLastError());
    return 0;
}

int sk_gvcf_plain_runs(const sk_gvcf_site_summary* summary, const uint32_t* clean_count, const uint32_t* raw_count, const uint32_t* mapq_count,
                       const sk_gvcf_block_options* opt, int32_t n, sk_gvcf_run* runs)
{
    SK_REQUIRE_INIT();
    if (n < 0 || !opt) return sk_fail("sk_gvcf_plain_runs: bad argument");
    if (n == 0) return 0;
    if (!summary || !clean_count || !raw_count || !mapq_count || !runs) return sk_fail("sk_gvcf_plain_runs: null argument");
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    hipStream_t st = ctx.stream;
    std::vector<int64_t> off(2 * (size_t(n) + 1), 0);
    int64_t* clean_off = off.data();
    int64_t* raw_off = off.data() + n + 1;
    for (int32_t i = 0; i < n; ++i) {
        clean_off[i + 1] = clean_off[i] + clean_count[i];
        raw_off[i + 1] = raw_off[i] + raw_count[i];
    }
    auto up = [](const size_t b) { return (b + 255) & ~size_t(255); };
    const size_t N = size_t(n);
    const size_t o_sum = 0, o_off = o_sum + up(sizeof(sk_gvcf_site_summary) * N), o_mq = o_off + up(16 * (N + 1)), o_pod = o_mq + up(4 * N),
                 o_runs = o_pod + up(17 * N + 512), total = o_runs + up(sizeof(sk_gvcf_run) * N);
    GvcfBuffers& B = gvcf_bufs();
    if (B.reserve(0, total)) return 1;
    char* d = static_cast<char*>(B.p[0]);
    SK_HIP(skrt::memcpyAsync(d + o_sum, summary, sizeof(sk_gvcf_site_summary) * N, hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(d + o_off, off.data(), 16 * (N + 1), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(d + o_mq, mapq_count, 4 * N, hipMemcpyHostToDevice, st));
    if (sk_gvcf_plain_runs_dev(reinterpret_cast<const sk_gvcf_site_summary*>(d + o_sum), reinterpret_cast<const int64_t*>(d + o_off),
                               reinterpret_cast<const int64_t*>(d + o_off) + n + 1, reinterpret_cast<const uint32_t*>(d + o_mq), opt, n, d + o_pod,
                               reinterpret_cast<sk_gvcf_run*>(d + o_runs), st))
        return 1;
    SK_HIP(skrt::memcpyAsync(runs, d + o_runs, sizeof(sk_gvcf_run) * N, hipMemcpyDeviceToHost, st));
    SK_HIP(skrt::streamSynchronize(st)); // (`off` is pageable: the copies above have finished by now)
    return 0;
}

int sk_gvcf_site_summaries(const sk_pileup_batch* hb, const sk_digt_call* genotypes, sk_gvcf_site_summary* out)
{
    SK_REQUIRE_INIT();
    if (!hb || hb->n_loci < 0) return sk_fail("sk_gvcf_site_summaries: bad batch");
    if (hb->n_loci == 0) return 0;
    if (!genotypes || !out || !hb->call_off || !hb->calls || !hb->ref_base) return sk_fail("sk_gvcf_site_summaries: null argument");
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    hipStream_t st = ctx.stream;
    const size_t n = size_t(hb->n_loci);
    const size_t n_calls = size_t(hb->call_off[n]);
    auto up = [](const size_t b) { return (b + 255) & ~size_t(255); };
    const size_t o_off = 0, o_calls = o_off + up(8 * (n + 1)), o_ref = o_calls + up(2 * n_calls + 2), o_pl = o_ref + up(n), o_g = o_pl + up(n),
                 o_out = o_g + up(sizeof(sk_digt_call) * n), total = o_out + up(sizeof(sk_gvcf_site_summary) * n);
    GvcfBuffers& B = gvcf_bufs();
    if (B.reserve(0, total)) return 1;
    char* d = static_cast<char*>(B.p[0]);
    SK_HIP(skrt::memcpyAsync(d + o_off, hb->call_off, 8 * (n + 1), hipMemcpyHostToDevice, st));
    if (n_calls) SK_HIP(skrt::memcpyAsync(d + o_calls, hb->calls, 2 * n_calls, hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(d + o_ref, hb->ref_base, n, hipMemcpyHostToDevice, st));
    if (hb->ploidy) SK_HIP(skrt::memcpyAsync(d + o_pl, hb->ploidy, n, hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(d + o_g, genotypes, sizeof(sk_digt_call) * n, hipMemcpyHostToDevice, st));
    sk_pileup_batch db;
    std::memset(&db, 0, sizeof(db));
    db.n_loci = hb->n_loci;
    db.call_off = reinterpret_cast<const int64_t*>(d + o_off);
    db.calls = reinterpret_cast<const uint16_t*>(d + o_calls);
    db.ref_base = reinterpret_cast<const uint8_t*>(d + o_ref);
    db.ploidy = hb->ploidy ? reinterpret_cast<const uint8_t*>(d + o_pl) : nullptr;
    if (sk_gvcf_site_summaries_dev(&db, reinterpret_cast<const sk_digt_call*>(d + o_g), reinterpret_cast<sk_gvcf_site_summary*>(d + o_out), st)) return 1;
    SK_HIP(skrt::memcpyAsync(out, d + o_out, sizeof(sk_gvcf_site_summary) * n, hipMemcpyDeviceToHost, st));
    SK_HIP(skrt::streamSynchronize(st));
    return 0;
}

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
    SK_LAUNCH(gvcf_block_kernel, dim3((n_sites + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(skrt::getLastError());
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
    SK_HIP(skrt::setDevice(ctx.device));
    hipStream_t st = ctx.stream;
    GvcfBuffers& B = gvcf_bufs();
    if (B.reserve(0, sizeof(sk_gvcf_site) * size_t(n_sites)) || B.reserve(1, size_t(n_sites)) || B.reserve(2, sizeof(sk_gvcf_block) * size_t(n_sites))) return 1;
    SK_HIP(skrt::memcpyAsync(B.p[0], sites, sizeof(sk_gvcf_site) * size_t(n_sites), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memsetAsync(B.p[2], 0, sizeof(sk_gvcf_block) * size_t(n_sites), st));
    if (sk_gvcf_block_sites_dev(static_cast<sk_gvcf_site*>(B.p[0]), n_sites, block_percent_tol, block_abs_tol, static_cast<uint8_t*>(B.p[1]),
                                static_cast<sk_gvcf_block*>(B.p[2]), st))
        return 1;
    SK_HIP(skrt::memcpyAsync(kind, B.p[1], size_t(n_sites), hipMemcpyDeviceToHost, st));
    SK_HIP(skrt::memcpyAsync(blocks, B.p[2], sizeof(sk_gvcf_block) * size_t(n_sites), hipMemcpyDeviceToHost, st));
    SK_HIP(skrt::streamSynchronize(st));
    return 0;
}

} // extern "C"
