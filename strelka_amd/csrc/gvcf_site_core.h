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

// is_new_value_blockable :59-73 on a stream_stat of which only the extremes matter: the statistic with the new value added has
// min' = min(min, v), max' = max(max, v), and check_block_tolerance :41-55 reads nothing else (compat_round of an integer-valued
// minimum is the minimum)
SKGS_HD inline bool extremes_blockable(const int v, const int mn, const int mx, const double frac_tol, const int abs_tol)
{
    const int lo = v < mn ? v : mn, hi = v > mx ? v : mx;
    const double half_max = double(hi) / 2.0;
    if (double(lo + abs_tol) >= half_max) return true;
    const int ftol = int(__builtin_floor(double(lo) * frac_tol));
    if (ftol <= abs_tol) return false;
    return double(lo + ftol) >= half_max;
}

// gvcf_writer::queue_site_record's joining from an empty block at site i (testCanSiteJoinSampleBlockShared :77-122 for two plain
// sites: equal filters; depth and filtered depth within tolerance of the block; both covered; both 0/0, diploid; GQX within tolerance
// :163-182; joinSiteToSampleBlock :126-157)
SKGS_HD inline sk_gvcf_run plain_run(const SitePod* pod, const int64_t n, const int64_t i, const double frac_tol, const int abs_tol)
{
    sk_gvcf_run r;
    r.len = 0;
    r.filter_key = 0;
    r.gqx_min = r.gqx_max = 0;
    r.dpu_min = r.dpu_max = r.dpf_min = r.dpf_max = 0;
    const SitePod first = pod[i];
    if (!(first.key_plain & POD_PLAIN)) return r;
    const uint32_t key = first.key_plain;
    int g0 = first.gqx, g1 = first.gqx, u0 = int(first.used), u1 = int(first.used), f0 = int(first.unused), f1 = int(first.unused);
    int64_t j = i + 1;
    for (; j < n; ++j) {
        const SitePod s = pod[j];
        if (s.key_plain != key) break; // (not plain, or other filters)
        if (!extremes_blockable(int(s.used), u0, u1, frac_tol, abs_tol)) break;
        if (!extremes_blockable(int(s.unused), f0, f1, frac_tol, abs_tol)) break;
        if (!extremes_blockable(s.gqx, g0, g1, frac_tol, abs_tol)) break;
        u0 = int(s.used) < u0 ? int(s.used) : u0;
        u1 = int(s.used) > u1 ? int(s.used) : u1;
        f0 = int(s.unused) < f0 ? int(s.unused) : f0;
        f1 = int(s.unused) > f1 ? int(s.unused) : f1;
        g0 = s.gqx < g0 ? s.gqx : g0;
        g1 = s.gqx > g1 ? s.gqx : g1;
    }
    r.len = int32_t(j - i);
    r.filter_key = key & 0xfu;
    r.gqx_min = g0;
    r.gqx_max = g1;
    r.dpu_min = uint32_t(u0);
    r.dpu_max = uint32_t(u1);
    r.dpf_min = uint32_t(f0);
    r.dpf_max = uint32_t(f1);
    return r;
}

} // namespace skgvcf
