// sk_adapter_gvcf.cpp -- site 10: the gVCF writer's non-variant blocks fed from the pileup stream's window (SURVEY.md section 8f rank 4,
// the output side).
//
// For every covered position the reference builds a GermlineDiploidSiteLocusInfo (process_pos_snp_digt, L/applications/starling/
// starling_pos_processor.cpp:619-701: candidate alternate alleles, translated genotype, PLs, AD counts, per-sample site info), sends it
// through the variant pipe (prefilter -> phaser -> overlap resolver) and, for almost every position of a genome, the writer then reads
// five numbers of it and joins it to the sample's open non-variant block (gvcf_writer::queue_site_record, gvcf_writer.cpp:278-302).
// The device has already said which positions are such PLAIN sites (sk_gvcf_site_summary, made beside the genotype record from the same
// cleaned column: csrc/gvcf_site_core.h) and holds their numbers.  For a plain position whose state at call time is the window's --
// ploidy not lowered by an indel call made since, not forced, no active region open in the phaser, no variant indel buffered in the
// resolver or still overlapping in the writer -- the adapter
//
//   * does not clean the pileup (process_pos_sample_stats takes the two depth counts from the window: sample_stats_counts),
//   * does not build a locus: it patches ONE kept GermlineDiploidSiteLocusInfo with the position's numbers, lets the reference's own
//     ScoringModelManager set its filters (applyDepthFilter, classify_site: the calls variant_prefilter_stage makes) and hands it to the
//     reference's own gvcf_writer::skip_to_pos / add_site_internal -- the block joining, the block records and their text are the
//     reference's code on a locus that holds, field for field, what the writer reads of the one process_pos_snp_digt would have built.
//
// Anything else (a variant or filtered-looking site, an alternate allele in the column, ploidy 1 or 0, reference N, an empty cleaned
// column, a forced position, any buffered locus in the pipe, call regions) takes the reference's path as before.  With several samples a
// position is routed when it is a plain site of EVERY sample's window (the locus then has no alternate allele: getSiteAltAlleles ranks
// the samples' columns one by one and adds the bases of their most likely genotypes); the writer's own loop joins it to each sample's
// block.  Whole blocks (below) are installed in single-sample runs only.
// $STRELKA_AMD_GVCF_FAST=0 switches the site off.
#include "sk_adapter_access.hh"

#include "applications/starling/VariantOverlapResolver.hh"
#include "applications/starling/VariantPhaser.hh"
#include "applications/starling/gvcf_aggregator.hh"
#include "applications/starling/gvcf_writer.hh"
#include "applications/starling/starling_pos_processor.hh"
#include "blt_util/seq_util.hh"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <vector>

namespace sk_adapter
{

struct GvcfAccess
{
    static gvcf_aggregator* aggregator(starling_pos_processor& pp) { return pp._gvcfer.get(); }
    static const starling_options& opt(const starling_pos_processor& pp) { return pp._opt; }
    static const starling_deriv_options& dopt(const starling_pos_processor& pp) { return pp._dopt; }
    static const ScoringModelManager& models(const gvcf_aggregator& a) { return a._scoringModels; }
    static gvcf_writer* writer(gvcf_aggregator& a) { return a._gvcfWriterPtr.get(); }
    static VariantPhaser* phaser(gvcf_aggregator& a) { return a._variantPhaserPtr.get(); }
    static bool isPhaserEmpty(const VariantPhaser& p) { return p._locusBuffer.empty(); }
    static VariantOverlapResolver* resolverBehind(VariantPhaser& p) { return dynamic_cast<VariantOverlapResolver*>(p._sink.get()); }
    static bool isResolverEmpty(const VariantOverlapResolver& r)
    {
        return r._variantIndelBuffer.empty() && r._nonvariantIndelBuffer.empty() && r._siteBuffer.empty();
    }
    /// the writer's last variant indel still reaches this position (modifySiteForConsistencyWithUpstreamIndels would change the site)
    static bool isUnderLastVariantIndel(const gvcf_writer& w, const pos_t pos)
    {
        return w._lastVariantIndelWritten && pos < w._lastVariantIndelWritten->end();
    }
    static bool isCompressible(const gvcf_writer& w, const GermlineSiteLocusInfo& locus) { return w._gvcf_comp.is_site_compressible(locus); }
    static gvcf_block_site_record& sampleBlock(gvcf_writer& w, const unsigned sampleIndex) { return w._blockPerSample[sampleIndex]; }
    static bool isRangeCompressible(const gvcf_writer& w, const pos_t begin, const pos_t end)
    {
        return w._gvcf_comp.is_range_compressible(known_pos_range2(begin, end));
    }
    static void setHeadPos(gvcf_writer& w, const pos_t pos) { w._headPos = pos; }
    /// gvcf_writer::process(site) without the ownership (gvcf_writer.cpp:180-197)
    static void writerProcessSite(gvcf_writer& w, GermlineSiteLocusInfo& locus)
    {
        w.skip_to_pos(locus.pos);
        w.add_site_internal(locus);
    }
};

namespace
{

struct GvcfFast
{
    bool decided = false, enabled = false;
    unsigned sampleCount = 1;
    uint32_t cleanSkipped = 0; ///< bit s: process_pos_sample_stats took sample s's counts of cleanSkippedPos from the window and did not clean its pileup
    pos_t cleanSkippedPos = 0;
    std::unique_ptr<GermlineDiploidSiteLocusInfo> scratch;
    std::vector<const SiteChunk*> sampleChunks; ///< (scratch of gvcf_plain_site)
    bool isBlocks = false; ///< the window brings the block that would start at every plain site (sk_gvcf_run): whole blocks are installed
    pos_t blockTo = std::numeric_limits<pos_t>::min(); ///< positions below are members of the block installed last (their process_pos_snp has nothing left to do)
    /// What a member of the installed block still owes the reference when its turn comes -- the two depth counts of process_pos_sample_stats
    /// -- is written down HERE when the block is installed, member by member from blockFrom on: from then on a member depends on nothing
    /// that can change or go away (its window's chunk, the position's pileup), so there is no state left to find broken mid-genome.
    pos_t blockFrom = std::numeric_limits<pos_t>::min();
    std::vector<uint32_t> blockUsed, blockUnused;
    unsigned long plainSites = 0, referenceSites = 0, declinedByState = 0;
    unsigned long blocksInstalled = 0, blockSites = 0, blocksDeclined = 0, filterKeyMismatches = 0;
    ~GvcfFast()
    {
        const char* v(std::getenv("STRELKA_AMD_VERBOSE"));
        if (v && std::atoi(v) != 0)
        {
            std::cerr << "strelka_amd adapter gvcf: gvcf_plain_sites=" << plainSites << " gvcf_reference_sites=" << referenceSites
                      << " gvcf_plain_declined_by_state=" << declinedByState << " gvcf_blocks_installed=" << blocksInstalled
                      << " gvcf_block_sites=" << blockSites << " gvcf_blocks_declined=" << blocksDeclined
                      << " gvcf_filter_key_mismatches=" << filterKeyMismatches << "\n";
        }
    }
};

GvcfFast& gf()
{
    static GvcfFast g;
    return g;
}

bool decide(starling_pos_processor& pp)
{
    GvcfFast& g(gf());
    if (g.decided) return g.enabled;
    g.decided = true;
    const starling_options& opt(GvcfAccess::opt(pp));
    const char* v(std::getenv("STRELKA_AMD_GVCF_FAST"));
    const bool isOn(! (v && *v == '0'));
    State& s(state());
    {
        const char* b(std::getenv("STRELKA_AMD_GVCF_BLOCKS"));
        g.isBlocks = ! (b && *b == '0');
    }
    g.sampleCount = Access::sampleCount(pp);
    g.enabled = isOn && s.pileup.enabled && s.pileup.isGenotyping && (g.sampleCount >= 1) && (g.sampleCount <= 32) && opt.is_bsnp_diploid() &&
                opt.gvcf.is_gvcf_output() && (! opt.isUseCallRegions()) && (GvcfAccess::aggregator(pp) != nullptr);
    if (g.sampleCount != 1) g.isBlocks = false;
    if (g.enabled)
    {
        // the one locus that stands for every plain site: what never changes is set here (updateSnvLocusWithSampleInfo :344-500 for
        // diploid samples whose most likely genotype is 0/0 under both priors and whose locus has no alternate allele)
        g.scratch.reset(new GermlineDiploidSiteLocusInfo(GvcfAccess::dopt(pp).gvcf, g.sampleCount));
        for (unsigned sampleIndex(0); sampleIndex < g.sampleCount; ++sampleIndex)
        {
            LocusSampleInfo& sampleInfo(g.scratch->getSample(sampleIndex));
            sampleInfo.setPloidy(2);
            sampleInfo.setActiveRegionId(-1);
            sampleInfo.maxGenotypeIndex.setGenotypeFromAlleleIndices(0, 0);
            sampleInfo.maxGenotypeIndexPolymorphic.setGenotypeFromAlleleIndices(0, 0);
            sampleInfo.supportCounts.setAltCount(0);
        }
    }
    return g.enabled;
}

/// the chunk of a sample that holds pos (chunks leave from the front as POST_ALIGN moves forward, as in site_diploid_genotype)
const SiteChunk* chunkAt(const pos_t pos, const unsigned sampleIndex = 0)
{
    State& s(state());
    if (s.pileup.chunks.size() <= sampleIndex) return nullptr;
    std::deque<SiteChunk>& chunks(s.pileup.chunks[sampleIndex]);
    while ((! chunks.empty()) && chunks.front().end <= pos) chunks.pop_front();
    if (chunks.empty() || chunks.front().begin > pos) return nullptr;
    return &chunks.front();
}

/// plain by the window's account, with the column the window wrote still in place
bool isPlainInWindow(const SiteChunk& c, const size_t k, const snp_pos_info& pi)
{
    if (c.summary.empty()) return false;
    return (c.summary[k].flags & 1u) != 0 && c.ploidy[k] == 2 && c.rawCount[k] == pi.calls.size();
}

}

void gvcf_reset_region()
{
    // (a process may call several regions, on any chromosome: nothing of the last one's installed block carries over)
    GvcfFast& g(gf());
    g.blockTo = g.blockFrom = std::numeric_limits<pos_t>::min();
    g.cleanSkipped = 0;
}

void gvcf_configure_stream(starling_pos_processor_base& pp, const unsigned sampleIndex, sk_pileup_stream* stream)
{
    // (before the first position of the region is called: the decision of decide() is not made yet -- a stream that turns out not to
    // feed the writer only carries the runs along)
    starling_pos_processor* spp(dynamic_cast<starling_pos_processor*>(&pp));
    if (spp == nullptr || sampleIndex != 0 || stream == nullptr) return;
    const starling_options& opt(GvcfAccess::opt(*spp));
    gvcf_aggregator* agg(GvcfAccess::aggregator(*spp));
    {
        const char* b(std::getenv("STRELKA_AMD_GVCF_BLOCKS"));
        const char* f(std::getenv("STRELKA_AMD_GVCF_FAST"));
        if ((b && *b == '0') || (f && *f == '0') || agg == nullptr || (! opt.gvcf.is_gvcf_output()) || Access::sampleCount(pp) != 1) return;
    }
    sk_gvcf_block_options bo;
    std::memset(&bo, 0, sizeof(bo));
    bo.min_passed_call_depth = opt.gvcf.minPassedCallDepth;
    bo.is_min_homref_gqx = opt.gvcf.is_min_homref_gqx ? 1 : 0;
    bo.min_homref_gqx = opt.gvcf.min_homref_gqx;
    bo.is_max_base_filt = opt.gvcf.is_max_base_filt ? 1 : 0;
    bo.max_base_filt = opt.gvcf.max_base_filt;
    // ScoringModelManager::default_classify_site :293-302: the depth ceiling of the chromosome the region lies on (resetChrom :80-97,
    // called by gvcf_aggregator::resetRegion before the region's first read)
    bo.is_max_depth = GvcfAccess::dopt(*spp).gvcf.is_max_depth() ? 1 : 0;
    bo.max_chrom_depth = bo.is_max_depth ? agg->getMaxDepth() : 0.;
    bo.block_percent_tol = opt.gvcf.block_percent_tol;
    bo.block_abs_tol = opt.gvcf.block_abs_tol;
    check(sk_pileup_stream_set_gvcf_block_options(stream, &bo), "sk_pileup_stream_set_gvcf_block_options");
}

bool germline_sample_stats_counts(starling_pos_processor_base& pp, const pos_t pos, const unsigned sampleIndex, unsigned& used, unsigned& unused)
{
    GvcfFast& g(gf());
    if (g.cleanSkippedPos != pos) g.cleanSkipped = 0; // (the samples of one position are asked one after the other)
    g.cleanSkippedPos = pos;
    g.cleanSkipped &= ~(1u << (sampleIndex & 31u));
    if (! (g.decided && g.enabled) || sampleIndex >= g.sampleCount) return false;
    if (pos < g.blockTo && pos >= g.blockFrom && sampleIndex == 0)
    {
        // a member of the installed block (single-sample runs): the counts noted at the install
        used = g.blockUsed[static_cast<size_t>(pos - g.blockFrom)];
        unused = g.blockUnused[static_cast<size_t>(pos - g.blockFrom)];
        g.cleanSkipped |= 1u;
        return true;
    }
    const SiteChunk* c(chunkAt(pos, sampleIndex));
    if (c == nullptr) return false;
    const snp_pos_info& pi(pp.sample(sampleIndex).basecallBuffer.get_pos(pos));
    const size_t k(static_cast<size_t>(pos - c->begin));
    if (! isPlainInWindow(*c, k, pi)) return false;
    // CleanedPileup::usedBasecallCount / unusedBasecallCount of CleanPileupFilter(pi, false) (PileupCleaner.hh:48-58)
    used = c->cleanCount[k];
    unused = static_cast<unsigned>(pi.calls.size()) - used;
    g.cleanSkipped |= (1u << sampleIndex);
    return true;
}

bool gvcf_plain_site(starling_pos_processor& pp, const pos_t pos)
{
    if (! decide(pp)) return false;
    GvcfFast& g(gf());
    const unsigned sampleCount(g.sampleCount);
    const uint32_t allSamples((sampleCount >= 32) ? 0xffffffffu : ((1u << sampleCount) - 1u));
    const uint32_t cleanSkipped((g.cleanSkippedPos == pos) ? g.cleanSkipped : 0u);
    const bool isCleanSkipped(cleanSkipped == allSamples); // (every sample's counts came from its window: plain in all of them)
    g.cleanSkipped = 0;
    if (pos < g.blockTo)
    {
        // a member of the block installed at its first site: joined already (nine positions in ten end here).  Everything this position
        // needed was checked and noted when the block was installed (blockUsed / blockUnused); nothing is looked up again.
        return true;
    }
    starling_pos_processor_base& base(pp);
    starling_pos_processor_base::sample_info& sif(base.sample(0));
    const snp_pos_info& pi(sif.basecallBuffer.get_pos(pos));

    auto referencePath = [&]() -> bool
    {
        // the reference's process_pos_snp takes it from here; a sample's cleaned pileup is made now if process_pos_sample_stats left it out
        for (unsigned sampleIndex(0); sampleIndex < sampleCount; ++sampleIndex)
        {
            if ((cleanSkipped >> sampleIndex) & 1u)
            {
                starling_pos_processor_base::sample_info& ssif(base.sample(sampleIndex));
                Access::pileupCleaner(base).CleanPileupFilter(ssif.basecallBuffer.get_pos(pos), false, ssif.cleanedPileup);
            }
        }
        if (! pi.calls.empty()) g.referenceSites++;
        return false;
    };
    if (! isCleanSkipped) return referencePath(); // (not plain in every window, or the counts did not come from them)

    const SiteChunk* c(chunkAt(pos));
    if (c == nullptr) return referencePath();
    const size_t k(static_cast<size_t>(pos - c->begin));
    if (! isPlainInWindow(*c, k, pi)) return referencePath();
    std::vector<const SiteChunk*>& sampleChunks(g.sampleChunks);
    sampleChunks.assign(sampleCount, c);
    for (unsigned sampleIndex(1); sampleIndex < sampleCount; ++sampleIndex)
    {
        const SiteChunk* sc(chunkAt(pos, sampleIndex));
        if (sc == nullptr || (! isPlainInWindow(*sc, static_cast<size_t>(pos - sc->begin), base.sample(sampleIndex).basecallBuffer.get_pos(pos))))
        {
            return referencePath();
        }
        sampleChunks[sampleIndex] = sc;
    }

    // ---- the state at call time (process_pos_snp_digt "prep step 2" :637-651; is_forced_output_pos :152)
    bool isStateTheWindows(! Access::isForcedOutputPos(base, pos));
    for (unsigned sampleIndex(0); isStateTheWindows && sampleIndex < sampleCount; ++sampleIndex)
    {
        isStateTheWindows = (static_cast<int>(Access::ploidy(base, pos, sampleIndex)) == 2) &&
                            (base.sample(sampleIndex).basecallBuffer.get_pos(pos).spanningIndelPloidyModification == 0);
    }
    if (! isStateTheWindows)
    {
        g.declinedByState++;
        return referencePath();
    }
    const reference_contig_segment& ref(Access::ref(base));
    const uint8_t refBaseIndex(base_to_id(ref.get_base(pos)));
    if (refBaseIndex == BASE_ID::ANY) return referencePath();

    // ---- the pipe between process_pos_snp_digt and the writer must be empty: a locus waiting there would be overtaken
    gvcf_aggregator& agg(*GvcfAccess::aggregator(pp));
    gvcf_writer* writer(GvcfAccess::writer(agg));
    VariantPhaser* phaser(GvcfAccess::phaser(agg));
    if (writer == nullptr) return referencePath();
    if (phaser != nullptr)
    {
        VariantOverlapResolver* resolver(GvcfAccess::resolverBehind(*phaser));
        const bool isInActiveRegion(GvcfAccess::opt(pp).isUseVariantPhaser && (Access::activeRegionId(base, pos) >= 0));
        if ((! GvcfAccess::isPhaserEmpty(*phaser)) || isInActiveRegion || resolver == nullptr || (! GvcfAccess::isResolverEmpty(*resolver)))
        {
            g.declinedByState++;
            return referencePath();
        }
    }
    if (GvcfAccess::isUnderLastVariantIndel(*writer, pos))
    {
        g.declinedByState++;
        return referencePath();
    }

    // ---- the locus process_pos_snp_digt would have built, as far as anything downstream reads it
    GermlineDiploidSiteLocusInfo& locus(*g.scratch);
    locus.pos = pos;
    locus.refBaseIndex = refBaseIndex;
    locus.isForcedOutput = false;
    locus.filters.clear();
    for (unsigned sampleIndex(0); sampleIndex < sampleCount; ++sampleIndex)
    {
        const SiteChunk& sc(*sampleChunks[sampleIndex]);
        const size_t sk(static_cast<size_t>(pos - sc.begin));
        const sk_digt_call& sdgt(sc.calls[sk]);
        const sk_gvcf_site_summary& ssm(sc.summary[sk]);
        const snp_pos_info& spi(base.sample(sampleIndex).basecallBuffer.get_pos(pos));
        LocusSampleInfo& sampleInfo(locus.getSample(sampleIndex));
        sampleInfo.filters.clear();
        sampleInfo.genotypeQuality = sdgt.genome.max_gt_qphred;           // :388
        sampleInfo.genotypeQualityPolymorphic = sdgt.poly.max_gt_qphred;   // :391
        sampleInfo.setGqx();                                               // :394
        sampleInfo.supportCounts.setAltCount(0);                           // :452 (clears the counts)
        sampleInfo.supportCounts.getCounts(true).incrementAlleleCount(0, ssm.ref_fwd);  // :455-465
        sampleInfo.supportCounts.getCounts(false).incrementAlleleCount(0, ssm.ref_rev);
        // updateSiteSampleInfo :200-250 (the EVS metrics belong to variant or forced sites only)
        GermlineSiteSampleInfo siteSampleInfo;
        siteSampleInfo.isOverlappingHomAltDeletion = false;
        // (updateSiteSampleInfo reads the count from the CLEANED pileup, and CleanPileupFilter does not copy it: it is always 0 there.
        //  Nothing of the diploid writer reads the field; the locus holds what the reference's would.)
        siteSampleInfo.spanningDeletionReadCount = 0;
        siteSampleInfo.usedBasecallCount = sc.cleanCount[sk];
        siteSampleInfo.unusedBasecallCount = static_cast<unsigned>(spi.calls.size()) - sc.cleanCount[sk];
        siteSampleInfo.mapqTracker = spi.mapqTracker;
        const double maxBias(GvcfAccess::opt(pp).maxAbsSampleVariantStrandBias);
        siteSampleInfo.strandBias = std::min(maxBias, std::max(-maxBias, sdgt.strand_bias));
        locus.setSiteSampleInfo(sampleIndex, siteSampleInfo);
    }
    LocusSampleInfo& sampleInfo(locus.getSample(0)); // (the whole-block step below: single-sample runs)
    // variant_prefilter_stage::process (variant_prefilter_stage.cpp:52-71): no ploidy conflict; the depth filter; the site's filters
    const ScoringModelManager& models(GvcfAccess::models(agg));
    models.applyDepthFilter(locus);
    models.classify_site(locus);
    if (! GvcfAccess::isCompressible(*writer, locus)) return referencePath(); // (a no-compress region: the writer would print the site)

    GvcfAccess::writerProcessSite(*writer, locus);
    g.plainSites++;

    // ---- whole blocks: when this site has just STARTED the sample's block, the device has already walked the writer's greedy joining
    // from it over the plain sites that follow (sk_gvcf_run, gvcf_plain_run_kernel: testCanSiteJoinSampleBlock / joinSiteToSampleBlock
    // site after site).  If nothing can happen at those positions that the window did not know -- no indel key, no forced position, no
    // active region, the ploidy and the columns the window's -- the block is brought to the state those joins leave (count, the three
    // running statistics, the writer's head) and the positions have nothing left to do when process_pos_snp reaches them.  The site
    // that ends the block, plain or not, meets that block through the reference's own test.  (What the device has done is the join
    // TESTS, one after the other from the block's first site; the joins themselves are three additions per member.)
    if (g.isBlocks && (! c->runs.empty()))
    {
        gvcf_block_site_record& block(GvcfAccess::sampleBlock(*writer, 0));
        const sk_gvcf_run& run(c->runs[k]);
        const pos_t end(std::min(pos + static_cast<pos_t>(run.len), std::min(c->end, Access::reportRange(base).end_pos())));
        if (block.count == 1 && block.pos == pos && end > pos + 1)
        {
            bool isSafe(true);
            // the filters the device gave the block's sites are the ones the reference has just given this one
            {
                const GermlineFilterKeeper& f(sampleInfo.filters);
                GermlineFilterKeeper known;
                uint32_t key(0);
                if (f.test(GERMLINE_VARIANT_VCF_FILTERS::LowDepth)) { key |= 1u; known.set(GERMLINE_VARIANT_VCF_FILTERS::LowDepth); }
                if (f.test(GERMLINE_VARIANT_VCF_FILTERS::LowGQX)) { key |= 2u; known.set(GERMLINE_VARIANT_VCF_FILTERS::LowGQX); }
                if (f.test(GERMLINE_VARIANT_VCF_FILTERS::HighDepth)) { key |= 4u; known.set(GERMLINE_VARIANT_VCF_FILTERS::HighDepth); }
                if (f.test(GERMLINE_VARIANT_VCF_FILTERS::HighBaseFilt)) { key |= 8u; known.set(GERMLINE_VARIANT_VCF_FILTERS::HighBaseFilt); }
                if (key != run.filter_key || (! (known == f)) || (! locus.filters.none()))
                {
                    g.filterKeyMismatches++;
                    isSafe = false;
                }
            }
            if (isSafe)
            {
                IndelBuffer& indelBuffer(Access::indelBuffer(base));
                isSafe = (indelBuffer.positionIterator(pos + 1) == indelBuffer.positionIterator(end)) &&
                         (! Access::isAnyForcedOutputPos(base, pos + 1, end)) && GvcfAccess::isRangeCompressible(*writer, pos, end);
            }
            const bool isPloidyRegions(Access::hasPloidyRegions(base, 0));
            const bool isPhasing(phaser != nullptr && GvcfAccess::opt(pp).isUseVariantPhaser);
            for (pos_t p(pos + 1); isSafe && p < end; ++p)
            {
                const snp_pos_info& ppi(sif.basecallBuffer.get_pos(p));
                isSafe = isPlainInWindow(*c, static_cast<size_t>(p - c->begin), ppi) && ppi.spanningIndelPloidyModification == 0 &&
                         ((! isPloidyRegions) || Access::ploidy(base, p, 0) == 2) && ((! isPhasing) || Access::activeRegionId(base, p) < 0);
            }
            if (isSafe)
            {
                const unsigned n(static_cast<unsigned>(end - pos));
                if (n == static_cast<unsigned>(run.len))
                {
                    // joinSiteToSampleBlock (gvcf_block_site_record.cpp:149-156) for every member after the first: the three
                    // accumulators take the member's numbers through the reference's own stream_stat::add, the count goes up
                    g.blockUsed.resize(n);
                    g.blockUnused.resize(n);
                    for (pos_t p(pos + 1); p < end; ++p)
                    {
                        const size_t kk(static_cast<size_t>(p - c->begin));
                        block.block_dpu.add(c->cleanCount[kk]);
                        block.block_dpf.add(c->rawCount[kk] - c->cleanCount[kk]);
                        block.block_gqx.add(c->summary[kk].gqx);
                        // (rawCount == the position's pileup size: isPlainInWindow above)
                        g.blockUsed[static_cast<size_t>(p - pos)] = c->cleanCount[kk];
                        g.blockUnused[static_cast<size_t>(p - pos)] = c->rawCount[kk] - c->cleanCount[kk];
                    }
                    g.blockFrom = pos;
                    block.count = static_cast<int>(n);
                    GvcfAccess::setHeadPos(*writer, end); // add_site_internal's _headPos = locus.pos + 1 of the last member
                    g.blockTo = end;
                    g.blocksInstalled++;
                    g.blockSites += (n - 1);
                }
                else
                {
                    g.blocksDeclined++; // (the window or the report range ends inside the block: its sites go one by one)
                }
            }
            else
            {
                g.blocksDeclined++;
            }
        }
    }
    return true;
}

}
