// gvcf_site_core.h -- what the gVCF writer's non-variant block logic needs of a position, made from the position's cleaned pileup column
// and its genotype record where they lie (the stream's window), for the device and the host (the output side).
//
// For a position of one diploid sample the reference builds a GermlineDiploidSiteLocusInfo (process_pos_snp_digt,
// L/applications/starling/starling_pos_processor.cpp:619-701): the candidate alternate alleles (getSiteAltAlleles :508-612), the genotype
// translated to them, PLs, AD counts (updateSnvLocusWithSampleInfo :344-500) -- and, for almost every position of a genome, finds a
// homozygous-reference site with no alternate allele, which the writer joins to the open block of the sample
// (gvcf_writer::queue_site_record, gvcf_writer.cpp:278-302) after reading five numbers of it.  This header says, per position, whether it
// is such a PLAIN site and gives the numbers:
//
//   plain  <=>  the reference base is known, the caller ploidy is 2, the cleaned column is not empty, getSiteAltAlleles would list no
//               alternate allele (the rank pass :527-560 gives no rank to a base other than the reference's; both most likely genotypes
//               -- genomic and polymorphic prior -- are the homozygous-reference one, so the pass over them :578-611 adds none)
//   gqx            LocusSampleInfo::setGqx (gvcf_locus_info.hh:356-369) with maxGenotypeIndex == maxGenotypeIndexPolymorphic == 0/0:
//                  min(genome.max_gt_qphred, poly.max_gt_qphred)
//   ref_fwd/rev    the AD counts of the reference allele by strand (:452-466; with no alternate allele every other base is skipped)
//
// Everything that depends on state the window cannot know (the ploidy as indel calls have lowered it since, forced output, an open
// active region or overlapping indel in the writer's pipe) is the caller's to check when the position is reached.
#pragma once

#include "strelka_amd.h"

#include "gvcf_block_core.h"

#ifdef __HIPCC__
#define SKGS_HD __host__ __device__
#else
#define SKGS_HD
#endif

namespace skgvcf
{

enum { SITE_PLAIN = 1 };

// calls: the cleaned tier1 column (CleanPileupFilter(pi, false)) in the reference's base_call bit layout (q:6, base:4, fwd, ...)
SKGS_HD inline sk_gvcf_site_summary site_summary(const uint16_t* calls, const int64_t n, const unsigned ref_base, const unsigned ploidy, const sk_digt_call& g)
{
    sk_gvcf_site_summary s;
    s.flags = 0;
    s.gqx = 0;
    s.ref_fwd = s.ref_rev = 0;
    if (ref_base > 3 || !g.is_called) return s;
    uint32_t cnt[4][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
    for (int64_t i = 0; i < n; ++i) {
        const unsigned b = (unsigned(calls[i]) >> 6) & 0xfu;
        if (b > 3) continue; // snp_pos_info::getBasecallCounts :166-174: unknown bases are not counted
        ++cnt[b][(unsigned(calls[i]) >> 10) & 1u];
    }
    s.ref_fwd = cnt[ref_base][1];
    s.ref_rev = cnt[ref_base][0];
    // getSiteAltAlleles :527-560 for one sample
    double c[4];
    unsigned min_count = 0;
    for (int b = 0; b < 4; ++b) {
        c[b] = double(cnt[b][0] + cnt[b][1]);
        min_count = unsigned(double(min_count) + c[b]); // (minCount += sampleBaseCounts[baseIndex])
    }
    min_count = unsigned(double(min_count) * 0.10);      // (minCount *= minAlleleFraction)
    if (min_count < 1u) min_count = 1u;
    bool alt = false;
    for (unsigned k = 0; k < ploidy && k < 2; ++k) {
        unsigned mb = 0;
        for (unsigned b = 1; b < 4; ++b)
            if (c[b] > c[mb]) mb = b;
        if (c[mb] >= double(min_count) && mb != ref_base) alt = true;
        c[mb] = 0;
    }
    // the pass over the most likely genotypes :578-611: hom-ref under both priors adds nothing (DIGT: genotypes 0..3 are AA,CC,GG,TT)
    const bool hom_ref = (g.poly.max_gt == ref_base) && (g.genome.max_gt == ref_base);
    const int32_t gq = int32_t(g.genome.max_gt_qphred), gqp = int32_t(g.poly.max_gt_qphred);
    s.gqx = gq < gqp ? gq : gqp;
    if (ploidy == 2 && n > 0 && !alt && hom_ref) s.flags |= SITE_PLAIN;
    return s;
}

// ---- the block that starts at a plain site (sk_gvcf_run) -------------------------------------------------------------------------
// what the join test reads of a plain site: its filters (as a key: equal filters <=> equal keys), GQX and the two depth counts
struct SitePod
{
    uint32_t key_plain; // bit 31: plain site; bits 0-3: LowDepth, LowGQX, HighDepth, HighBaseFilt
    int32_t gqx;
    uint32_t used, unused;
};
enum { POD_PLAIN = 0x80000000u };

// ScoringModelManager::applyDepthFilter :234-249 and default_classify_site :270-311 for a homozygous-reference site of one sample whose
// locus has no alternate allele (totalConfidentCounts = the reference allele's AD; allSampleLocusDepth = the sample's MapqTracker count)
SKGS_HD inline uint32_t site_filter_key(const sk_gvcf_block_options& o, const int32_t gqx, const uint32_t used, const uint32_t unused, const uint32_t ref_count,
                                        const uint32_t total_read_depth)
{
    uint32_t key = 0;
    if (ref_count < o.min_passed_call_depth || used < o.min_passed_call_depth) key |= 1u;                  // LowDepth
    if (o.is_min_homref_gqx && double(gqx) < o.min_homref_gqx) key |= 2u;                                  // LowGQX
    if (o.is_max_depth && double(total_read_depth) > o.max_chrom_depth) key |= 4u;                         // HighDepth
    if (o.is_max_base_filt) {                                                                              // HighBaseFilt
        const double total = double(used + unused);                                                        // (safeFrac, math_util.hh:108-113)
        const double frac = (total <= 0. && total >= 0.) ? 0. : double(unused) / total;
        if (frac > o.max_base_filt) key |= 8u;
    }
    return key;
}

// is_new_value_blockable :59-73 reads of a stream_stat only its extremes: the statistic with the new value added has min' = min(min, v),
// max' = max(max, v), and check_block_tolerance :41-55 reads nothing else (compat_round of an integer-valued minimum is the minimum).
// the extremes of a block's three accumulators (all the join test reads of them)
struct Extremes
{
    int g0, g1, u0, u1, f0, f1; // GQX, used depth, unused depth: minimum, maximum
    SKGS_HD void start(const SitePod& s)
    {
        g0 = g1 = s.gqx;
        u0 = u1 = int(s.used);
        f0 = f1 = int(s.unused);
    }
    SKGS_HD void take(const int g_lo, const int g_hi, const int u_lo, const int u_hi, const int f_lo, const int f_hi)
    {
        g0 = g_lo < g0 ? g_lo : g0;
        g1 = g_hi > g1 ? g_hi : g1;
        u0 = u_lo < u0 ? u_lo : u0;
        u1 = u_hi > u1 ? u_hi : u1;
        f0 = f_lo < f0 ? f_lo : f0;
        f1 = f_hi > f1 ? f_hi : f1;
    }
};
// would the three accumulators, with values between these bounds added, still pass check_block_tolerance?  (For one site the bounds are
// its values.  The test is monotone: lowering a minimum or raising a maximum can only make it fail -- so if it passes for the extremes
// of a whole run of sites it passes for every prefix of the run.)
SKGS_HD inline bool extremes_join(const Extremes& e, const int g_lo, const int g_hi, const int u_lo, const int u_hi, const int f_lo, const int f_hi, const double frac_tol,
                                  const int abs_tol)
{
    auto ok = [&](const int lo_new, const int hi_new, const int lo, const int hi) {
        const int mn = lo_new < lo ? lo_new : lo, mx = hi_new > hi ? hi_new : hi;
        const double half_max = double(mx) / 2.0;
        if (double(mn + abs_tol) >= half_max) return true;
        const int ftol = int(__builtin_floor(double(mn) * frac_tol));
        if (ftol <= abs_tol) return false;
        return double(mn + ftol) >= half_max;
    };
    return ok(u_lo, u_hi, e.u0, e.u1) && ok(f_lo, f_hi, e.f0, e.f1) && ok(g_lo, g_hi, e.g0, e.g1);
}

// 32 consecutive sites at once: their common key (TILE_MIXED when they are not all plain with one key) and the extremes of their values
struct SiteTile
{
    uint32_t key;
    int g0, g1, u0, u1, f0, f1;
    uint32_t pad;
};
enum { TILE_SITES = 32, TILE_MIXED = 0x7fffffffu };
SKGS_HD inline SiteTile make_tile(const SitePod* pod, const int64_t n, const int64_t t)
{
    SiteTile T;
    T.pad = 0;
    const int64_t b = t * TILE_SITES;
    T.key = TILE_MIXED;
    T.g0 = T.g1 = T.u0 = T.u1 = T.f0 = T.f1 = 0;
    if (b + TILE_SITES > n) return T; // (a partial tile at the window's end is walked site by site)
    Extremes e;
    e.start(pod[b]);
    uint32_t key = pod[b].key_plain;
    if (!(key & POD_PLAIN)) return T;
    for (int64_t j = b + 1; j < b + TILE_SITES; ++j) {
        if (pod[j].key_plain != key) return T;
        e.take(pod[j].gqx, pod[j].gqx, int(pod[j].used), int(pod[j].used), int(pod[j].unused), int(pod[j].unused));
    }
    T.key = key;
    T.g0 = e.g0; T.g1 = e.g1; T.u0 = e.u0; T.u1 = e.u1; T.f0 = e.f0; T.f1 = e.f1;
    return T;
}

// gvcf_writer::queue_site_record's joining from an empty block at site i (testCanSiteJoinSampleBlockShared :77-122 for two plain
// sites: equal filters; depth and filtered depth within tolerance of the block; both covered; both 0/0, diploid; GQX within tolerance
// :163-182; joinSiteToSampleBlock :126-157).  Site by site to the next tile boundary, then whole tiles while every site of a tile joins
// (see extremes_join: the test passes for the tile's extremes), then site by site to the block's end.  tiles: null = site by site.
SKGS_HD inline sk_gvcf_run plain_run(const SitePod* pod, const SiteTile* tiles, const int64_t n, const int64_t i, const double frac_tol, const int abs_tol)
{
    sk_gvcf_run r;
    r.len = 0;
    r.filter_key = 0;
    r.gqx_min = r.gqx_max = 0;
    r.dpu_min = r.dpu_max = r.dpf_min = r.dpf_max = 0;
    const SitePod first = pod[i];
    if (!(first.key_plain & POD_PLAIN)) return r;
    const uint32_t key = first.key_plain;
    Extremes e;
    e.start(first);
    int64_t j = i + 1;
    bool open = true;
    auto site = [&]() { // site j: joins (true) or ends the block
        const SitePod s = pod[j];
        if (s.key_plain != key) return false; // (not plain, or other filters)
        if (!extremes_join(e, s.gqx, s.gqx, int(s.used), int(s.used), int(s.unused), int(s.unused), frac_tol, abs_tol)) return false;
        e.take(s.gqx, s.gqx, int(s.used), int(s.used), int(s.unused), int(s.unused));
        return true;
    };
    if (tiles) {
        while (open && j < n && (j % TILE_SITES) != 0) {
            if (site()) ++j; else open = false;
        }
        while (open && j + TILE_SITES <= n) {
            const SiteTile T = tiles[j / TILE_SITES];
            if (T.key != key || !extremes_join(e, T.g0, T.g1, T.u0, T.u1, T.f0, T.f1, frac_tol, abs_tol)) break;
            e.take(T.g0, T.g1, T.u0, T.u1, T.f0, T.f1);
            j += TILE_SITES;
        }
    }
    while (open && j < n) {
        if (site()) ++j; else open = false;
    }
    r.len = int32_t(j - i);
    r.filter_key = key & 0xfu;
    r.gqx_min = e.g0;
    r.gqx_max = e.g1;
    r.dpu_min = uint32_t(e.u0);
    r.dpu_max = uint32_t(e.u1);
    r.dpf_min = uint32_t(e.f0);
    r.dpf_max = uint32_t(e.f1);
    return r;
}

} // namespace skgvcf
