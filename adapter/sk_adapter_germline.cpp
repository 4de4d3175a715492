// sk_adapter_germline.cpp -- sites 2+3 of the germline caller: adjust_joint_eprob (L/starling_common/PileupCleaner.cpp:73)
// and pprob_digt_caller::position_snp_call_pprob_digt (L/applications/starling/starling_pos_processor.cpp:256-267) for a
// whole stage window in one sk_site_digt_call_fused call.
//
// When the (deferred) POST_ALIGN stage reaches position P the pileup columns of [P, P+W] are complete (sk_adapter.hh), so
// their cleaned columns (CleanPileupFilter, PileupCleaner.cpp:28-66) are packed into one sk_pileup_batch and genotyped in
// one launch.  The reference's per-position control flow is unchanged: process_pos_snp_digt(P+k) asks for its sample's
// diploid_genotype through site_diploid_genotype() and gets the cached record.  The one input that the window cannot
// know in advance is the site ploidy, which the indel calls of earlier positions may lower
// (updateDiploidIndelLocusWithSampleInfo -> decrementSpanningIndelPloidy, starling_pos_processor.cpp:1205-1226): each
// cache entry records the ploidy (and call count) it was computed with, and an entry that no longer matches is
// recomputed on the spot -- same entry point, one locus.
#include "sk_adapter_access.hh"

#include "applications/starling/starling_pos_processor.hh"
#include "blt_common/position_snp_call_pprob_digt.hh"
#include "blt_util/seq_util.hh"
#include "starling_common/LocusSupportingReadStats.hh"
#include "starling_common/PileupCleaner.hh"

#include <cstdlib>
#include <cstring>

namespace sk_adapter
{

namespace
{

void germlineOptions(const starling_base_options& opt, sk_germline_options& go)
{
    sk_germline_options_default(&go);
    go.bsnp_diploid_theta = opt.bsnp_diploid_theta;
    go.bsnp_ssd_no_mismatch = opt.bsnp_ssd_no_mismatch;
    go.bsnp_ssd_one_mismatch = opt.bsnp_ssd_one_mismatch;
    go.is_min_vexp = opt.is_min_vexp ? 1 : 0;
    go.min_vexp = opt.min_vexp;
    if (opt.isHetVariantFrequencyExtensionDefined())
    {
        throw blt_exception("strelka_amd adapter: --het-variant-frequency-extension (RNA) is not supported on this path");
    }
}

/// CleanPileupFilter(pi, is_include_tier2=false): the tier1 calls that are not filtered, in pileup order
void appendCleanedCalls(const snp_pos_info& pi, std::vector<uint16_t>& calls)
{
    static_assert(sizeof(base_call) == 2, "base_call is the 16-bit record the kernels decode");
    for (const base_call& bc : pi.calls)
    {
        if (bc.is_call_filter) continue;
        uint16_t v;
        std::memcpy(&v, &bc, 2);
        calls.push_back(v);
    }
}

uint8_t refBaseId(const char refBase)
{
    const unsigned id(base_to_id(refBase));
    return static_cast<uint8_t>(id < 4 ? id : 4);
}

unsigned callerPloidy(const starling_pos_processor_base& pp, const pos_t pos, const unsigned sampleIndex, const snp_pos_info& pi)
{
    // process_pos_snp_digt, "prep step 2" (starling_pos_processor.cpp:637-651)
    const int regionPloidy(Access::ploidy(pp, pos, sampleIndex));
    const int ploidy(std::max(0, regionPloidy + pi.spanningIndelPloidyModification));
    return (ploidy == 0) ? 2 : static_cast<unsigned>(ploidy);
}

void toDiploidGenotype(const sk_digt_call& c, const unsigned ploidy, diploid_genotype& dgt)
{
    dgt.ploidy = static_cast<int>(ploidy);
    if (! c.is_called) return; // reference base 'N': the reference returns before touching dgt (:481)
    dgt.ref_gt = c.ref_gt;
    const unsigned gtcount(ploidy == 1 ? 4u : 10u);
    for (unsigned gt(0); gt < gtcount; ++gt) dgt.phredLoghood[gt] = c.phredLoghood[gt];
    dgt.genome.max_gt = c.genome.max_gt;
    dgt.genome.ref_pprob = c.genome.ref_pprob;
    dgt.genome.snp_qphred = c.genome.snp_qphred;
    dgt.genome.max_gt_qphred = c.genome.max_gt_qphred;
    dgt.poly.max_gt = c.poly.max_gt;
    dgt.poly.ref_pprob = c.poly.ref_pprob;
    dgt.poly.snp_qphred = c.poly.snp_qphred;
    dgt.poly.max_gt_qphred = c.poly.max_gt_qphred;
    dgt.strand_bias = c.strand_bias;
}

void genotypeLoci(const starling_base_options& opt, const std::vector<int64_t>& callOff, const std::vector<uint16_t>& calls,
                  const std::vector<uint8_t>& refBase, const std::vector<uint8_t>& ploidy, sk_digt_call* out)
{
    sk_germline_options go;
    germlineOptions(opt, go);
    sk_pileup_batch pb;
    std::memset(&pb, 0, sizeof(pb));
    pb.n_loci = static_cast<int32_t>(refBase.size());
    pb.call_off = callOff.data();
    static const uint16_t none(0);
    pb.calls = calls.empty() ? &none : calls.data();
    pb.de = nullptr;
    pb.ref_base = refBase.data();
    pb.ploidy = ploidy.data();
    check(sk_site_digt_call_fused(&pb, &go, out, nullptr), "sk_site_digt_call_fused");
}

}

void before_process_pos_variants(starling_pos_processor_base& pp, const pos_t pos)
{
    const starling_base_options& opt(Access::opt(pp));
    if (opt.isSomaticCallingMode)
    {
        somatic_window(pp, pos);
        return;
    }
    pileup_before_variants(pp, pos);
    if (! opt.is_bsnp_diploid()) return; // the continuous-frequency caller stays on the reference's path

    State& s(state());
    if (s.pileup.isGenotyping) return; // the genotypes came with the pileup (site 9)
    SiteCache& cache(s.sites);
    if (pos >= cache.begin && pos < cache.end) return;
    AccumTimer hookTimer(s.tSiteHook);

    const unsigned sampleCount(Access::sampleCount(pp));
    const pos_t begin(pos), end(pos + static_cast<pos_t>(post_align_defer()) + 1);
    const size_t n(static_cast<size_t>(end - begin) * sampleCount);
    cache.begin = begin;
    cache.end = end;
    cache.isValid.assign(n, 0);
    cache.ploidy.assign(n, 2);
    cache.callCount.assign(n, 0);
    cache.calls.resize(n);

    std::vector<int64_t> callOff(1, 0);
    std::vector<uint16_t> calls;
    std::vector<uint8_t> refBase, ploidy;
    std::vector<size_t> slot;
    for (pos_t p(begin); p < end; ++p)
    {
        if (! Access::isPosReportable(pp, p)) continue;
        for (unsigned sampleIndex(0); sampleIndex < sampleCount; ++sampleIndex)
        {
            const snp_pos_info& pi(pp.sample(sampleIndex).basecallBuffer.get_pos(p));
            const size_t before(calls.size());
            appendCleanedCalls(pi, calls);
            callOff.push_back(static_cast<int64_t>(calls.size()));
            refBase.push_back(refBaseId(pi.get_ref_base()));
            const unsigned pl(callerPloidy(pp, p, sampleIndex, pi));
            ploidy.push_back(static_cast<uint8_t>(pl));
            const size_t k(static_cast<size_t>(p - begin) * sampleCount + sampleIndex);
            cache.ploidy[k] = static_cast<uint8_t>(pl);
            cache.callCount[k] = static_cast<uint32_t>(calls.size() - before);
            slot.push_back(k);
        }
    }
    if (slot.empty()) return;
    std::vector<sk_digt_call> out(slot.size());
    {
        AccumTimer abiTimer(s.tSiteAbi);
        genotypeLoci(opt, callOff, calls, refBase, ploidy, out.data());
    }
    for (size_t i(0); i < slot.size(); ++i)
    {
        cache.calls[slot[i]] = out[i];
        cache.isValid[slot[i]] = 1;
    }
    s.siteBatches++;
    s.siteLoci += slot.size();
}

void site_diploid_genotype(starling_pos_processor& pp, const pos_t pos, const unsigned sampleIndex, const unsigned ploidy,
                           diploid_genotype& dgt)
{
    State& s(state());
    SiteCache& cache(s.sites);
    const starling_pos_processor_base& base(pp);
    const CleanedPileup& cpi(base.sample(sampleIndex).cleanedPileup);
    const size_t cleanedCount(used_basecall_count(sampleIndex, cpi));
    const unsigned sampleCount(Access::sampleCount(base));
    if (s.pileup.isGenotyping)
    {
        std::deque<SiteChunk>& chunks(s.pileup.chunks[sampleIndex]);
        while ((! chunks.empty()) && chunks.front().end <= pos) chunks.pop_front(); // POST_ALIGN only moves forward
        if ((! chunks.empty()) && chunks.front().begin <= pos)
        {
            const SiteChunk& c(chunks.front());
            const size_t k(static_cast<size_t>(pos - c.begin));
            if (c.ploidy[k] == ploidy && c.cleanCount[k] == cleanedCount)
            {
                toDiploidGenotype(c.calls[k], ploidy, dgt);
                return;
            }
        }
    }
    else if (pos >= cache.begin && pos < cache.end)
    {
        const size_t k(static_cast<size_t>(pos - cache.begin) * sampleCount + sampleIndex);
        if (cache.isValid[k] && cache.ploidy[k] == ploidy && cache.callCount[k] == cleanedCount)
        {
            toDiploidGenotype(cache.calls[k], ploidy, dgt);
            return;
        }
    }
    // the window's assumption about this locus no longer holds (its ploidy was lowered by an indel call made since):
    // genotype the locus as it is now
    std::vector<int64_t> callOff(1, 0);
    std::vector<uint16_t> calls;
    appendCleanedCalls(cpi.rawPileup(), calls); // (the cleaned tier1 column, whether or not the copy was made)
    callOff.push_back(static_cast<int64_t>(calls.size()));
    const std::vector<uint8_t> refBase(1, refBaseId(cpi.rawPileup().get_ref_base()));
    const std::vector<uint8_t> pl(1, static_cast<uint8_t>(ploidy));
    sk_digt_call out;
    genotypeLoci(Access::opt(base), callOff, calls, refBase, pl, &out);
    toDiploidGenotype(out, ploidy, dgt);
    s.siteRecomputed++;
}

namespace
{

bool isCleanSummaryEnabled()
{
    static const char* const v(std::getenv("STRELKA_AMD_CLEAN_SUMMARY"));
    static const bool isEnabled(v == nullptr || *v == 0 || std::strtoul(v, nullptr, 10) != 0);
    return isEnabled;
}

/// the sample's summary if it was taken from the pileup this CleanedPileup points at
const CleanSummary* summaryOf(const unsigned sampleIndex, const CleanedPileup& cpi)
{
    const State& s(state());
    if (sampleIndex >= s.cleanSummary.size()) return nullptr;
    const CleanSummary& cs(s.cleanSummary[sampleIndex]);
    return (cs.pi != nullptr && cs.pi == &(cpi.rawPileup())) ? &cs : nullptr;
}

}

void germline_arm_clean_summary(const starling_pos_processor_base& pp, const pos_t pos, const unsigned sampleIndex)
{
    const starling_base_options& opt(Access::opt(pp));
    if (opt.isSomaticCallingMode || (! opt.is_bsnp_diploid()) || (! isCleanSummaryEnabled())) return;
    State& s(state());
    if (s.cleanSummary.size() <= sampleIndex) s.cleanSummary.resize(sampleIndex + 1);
    s.cleanSummaryArmed = static_cast<int>(sampleIndex);
    s.cleanSummaryArmedPos = pos;
    if (s.pileup.isGenotyping) return; // (the counts come with the stream's windows: the column itself is not read)
    // the columns were written a window ago and each is a small block of its own: ask for the one two positions on
    __builtin_prefetch(&(pp.sample(sampleIndex).basecallBuffer.get_pos(pos + 6).calls));
    const snp_pos_info& ahead(pp.sample(sampleIndex).basecallBuffer.get_pos(pos + 2));
    if (! ahead.calls.empty())
    {
        __builtin_prefetch(ahead.calls.data());
        __builtin_prefetch(reinterpret_cast<const char*>(ahead.calls.data()) + 64);
    }
}

bool clean_pileup_summary(const snp_pos_info& pi, const bool isIncludeTier2)
{
    State& s(state());
    const int armed(s.cleanSummaryArmed);
    s.cleanSummaryArmed = -1;
    if (armed < 0 || isIncludeTier2)
    {
        // cleaned by copy: no summary may go on describing this pileup object (the basecall buffer reuses its entries)
        for (CleanSummary& cs : s.cleanSummary)
        {
            if (cs.pi == &pi) cs.pi = nullptr;
        }
        return false;
    }
    CleanSummary& cs(s.cleanSummary[static_cast<size_t>(armed)]);
    for (CleanSummary& other : s.cleanSummary)
    {
        if (other.pi == &pi) other.pi = nullptr;
    }
    cs.pi = &pi;
    if (s.pileup.isGenotyping && static_cast<size_t>(armed) < s.pileup.chunks.size())
    {
        // the counts were taken when the stream delivered the window (sk_adapter_pileup.cpp), if this is still that column
        const pos_t pos(s.cleanSummaryArmedPos);
        std::deque<SiteChunk>& chunks(s.pileup.chunks[static_cast<size_t>(armed)]);
        while ((! chunks.empty()) && chunks.front().end <= pos) chunks.pop_front(); // (as site_diploid_genotype: POST_ALIGN only moves forward)
        if ((! chunks.empty()) && chunks.front().begin <= pos)
        {
            const SiteChunk& c(chunks.front());
            const size_t k(static_cast<size_t>(pos - c.begin));
            if ((! c.rawCount.empty()) && c.rawCount[k] == pi.calls.size())
            {
                const uint32_t* const count(c.strandBase.data() + k * 10);
                cs.used = 0;
                for (unsigned strand(0); strand < 2; ++strand)
                {
                    for (unsigned b(0); b < 5; ++b)
                    {
                        cs.count[strand][b] = count[strand * 5 + b];
                        cs.used += count[strand * 5 + b];
                    }
                }
                return true;
            }
        }
    }
    // CleanPileupFilter's tier1 test (PileupCleaner.cpp:40-50): what it would have copied, counted.  Most calls of a position land
    // on the same counter (the reference base of one strand or the other): two sets of counters, alternating, halve that chain.
    uint32_t count[2][2][8] = {};
    const base_call* const calls(pi.calls.data());
    const size_t n(pi.calls.size());
    size_t i(0);
    for (; i + 2 <= n; i += 2)
    {
        const base_call& a(calls[i]);
        const base_call& b(calls[i + 1]);
        count[0][a.is_fwd_strand ? 1 : 0][a.base_id & 7u] += a.is_call_filter ? 0u : 1u;
        count[1][b.is_fwd_strand ? 1 : 0][b.base_id & 7u] += b.is_call_filter ? 0u : 1u;
    }
    if (i < n) count[0][calls[i].is_fwd_strand ? 1 : 0][calls[i].base_id & 7u] += calls[i].is_call_filter ? 0u : 1u;
    cs.used = 0;
    for (unsigned strand(0); strand < 2; ++strand)
    {
        for (unsigned b(0); b < 5; ++b) cs.count[strand][b] = 0;
        for (unsigned b(0); b < 8; ++b)
        {
            // (base ids are 0..4, BASE_ID::ANY = 4)
            const uint32_t c(count[0][strand][b] + count[1][strand][b]);
            cs.count[strand][std::min(b, 4u)] += c;
            cs.used += c;
        }
    }
    return true;
}

unsigned used_basecall_count(const unsigned sampleIndex, const CleanedPileup& cpi)
{
    const CleanSummary* cs(summaryOf(sampleIndex, cpi));
    return (cs != nullptr) ? cs->used : cpi.usedBasecallCount();
}

unsigned unused_basecall_count(const unsigned sampleIndex, const CleanedPileup& cpi)
{
    const CleanSummary* cs(summaryOf(sampleIndex, cpi));
    return (cs != nullptr) ? (cpi.totalBasecallCount() - cs->used) : cpi.unusedBasecallCount();
}

bool summary_basecall_counts(const unsigned sampleIndex, const CleanedPileup& cpi, double* baseCount)
{
    const CleanSummary* cs(summaryOf(sampleIndex, cpi));
    if (cs == nullptr) return false;
    // snp_pos_info::getBasecallCounts (snp_pos_info.hh:162-174): unknown bases are not counted
    for (unsigned b(0); b < 4; ++b) baseCount[b] = static_cast<double>(cs->count[0][b] + cs->count[1][b]);
    return true;
}

bool summary_allele_counts(const unsigned sampleIndex, const CleanedPileup& cpi, const uint8_t* baseIndexToAlleleIndex, const uint8_t fullAlleleCount,
                           LocusSupportingReadStats& supportCounts)
{
    const CleanSummary* cs(summaryOf(sampleIndex, cpi));
    if (cs == nullptr) return false;
    for (unsigned b(0); b < 4; ++b)
    {
        const uint8_t alleleIndex(baseIndexToAlleleIndex[b]);
        if (alleleIndex == fullAlleleCount) continue;
        for (unsigned strand(0); strand < 2; ++strand)
        {
            if (cs->count[strand][b] != 0) supportCounts.getCounts(strand != 0).incrementAlleleCount(alleleIndex, cs->count[strand][b]);
        }
    }
    return true;
}

std::vector<int>& scratch_ploidy_vector(const unsigned which)
{
    static std::vector<int> v[2];
    v[which & 1u].clear();
    return v[which & 1u];
}

std::vector<diploid_genotype>& scratch_site_genotypes(const unsigned sampleCount)
{
    static std::vector<diploid_genotype> v;
    if (v.size() != sampleCount) v.assign(sampleCount, diploid_genotype());
    else
    {
        for (diploid_genotype& dgt : v) dgt.reset();
    }
    return v;
}

void empty_site_genotype(const starling_pos_processor_base& pp, const unsigned refBaseIndex, diploid_genotype& dgt)
{
    init();
    const std::vector<int64_t> callOff(2, 0);
    const std::vector<uint16_t> calls;
    const std::vector<uint8_t> refBase(1, static_cast<uint8_t>(refBaseIndex));
    const std::vector<uint8_t> pl(1, 2);
    sk_digt_call out;
    genotypeLoci(Access::opt(pp), callOff, calls, refBase, pl, &out);
    toDiploidGenotype(out, 2, dgt);
}

}
