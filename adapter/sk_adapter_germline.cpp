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
    const snp_pos_info& cleaned(base.sample(sampleIndex).cleanedPileup.cleanedPileup());
    const unsigned sampleCount(Access::sampleCount(base));
    if (s.pileup.isGenotyping)
    {
        std::deque<SiteChunk>& chunks(s.pileup.chunks[sampleIndex]);
        while ((! chunks.empty()) && chunks.front().end <= pos) chunks.pop_front(); // POST_ALIGN only moves forward
        if ((! chunks.empty()) && chunks.front().begin <= pos)
        {
            SiteChunk& c(chunks.front());
            const size_t k(static_cast<size_t>(pos - c.begin));
            if (c.ploidy[k] == ploidy && c.cleanCount[k] == cleaned.calls.size())
            {
                toDiploidGenotype(c.calls[k], ploidy, dgt);
                return;
            }
            // An indel call made since the window was genotyped has lowered the ploidy here -- and, as a rule, over the whole span of
            // the deletion (updateDiploidIndelLocusWithSampleInfo, starling_pos_processor.cpp:1215-1226, decrements every position of
            // it at once): the run of following positions of this chunk whose ploidy no longer is the window's is genotyped again in
            // ONE call and the chunk's records replaced, instead of one device round trip per position.
            if (c.cleanCount[k] == cleaned.calls.size())
            {
                std::vector<int64_t> callOff(1, 0);
                std::vector<uint16_t> calls;
                std::vector<uint8_t> refBase, pl;
                std::vector<size_t> slot;
                const size_t kEnd(std::min(c.calls.size(), k + 512));
                for (size_t kk(k); kk < kEnd; ++kk)
                {
                    const pos_t p(c.begin + static_cast<pos_t>(kk));
                    const snp_pos_info& pi(base.sample(sampleIndex).basecallBuffer.get_pos(p));
                    const unsigned nowPloidy((kk == k) ? ploidy : callerPloidy(base, p, sampleIndex, pi));
                    if (nowPloidy == c.ploidy[kk]) break;
                    const size_t before(calls.size());
                    appendCleanedCalls(pi, calls);
                    if ((calls.size() - before) != c.cleanCount[kk]) // (the column is not the window's: left to its own call)
                    {
                        calls.resize(before);
                        break;
                    }
                    callOff.push_back(static_cast<int64_t>(calls.size()));
                    refBase.push_back(refBaseId(pi.get_ref_base()));
                    pl.push_back(static_cast<uint8_t>(nowPloidy));
                    slot.push_back(kk);
                }
                if (! slot.empty())
                {
                    std::vector<sk_digt_call> out(slot.size());
                    genotypeLoci(Access::opt(base), callOff, calls, refBase, pl, out.data());
                    for (size_t i(0); i < slot.size(); ++i)
                    {
                        c.calls[slot[i]] = out[i];
                        c.ploidy[slot[i]] = pl[i];
                    }
                    s.siteRecomputed += slot.size();
                    s.siteRecomputeCalls++;
                    toDiploidGenotype(c.calls[k], ploidy, dgt);
                    return;
                }
            }
        }
    }
    else if (pos >= cache.begin && pos < cache.end)
    {
        const size_t k(static_cast<size_t>(pos - cache.begin) * sampleCount + sampleIndex);
        if (cache.isValid[k] && cache.ploidy[k] == ploidy && cache.callCount[k] == cleaned.calls.size())
        {
            toDiploidGenotype(cache.calls[k], ploidy, dgt);
            return;
        }
    }
    // the window's assumption about this locus no longer holds (its ploidy was lowered by an indel call made since):
    // genotype the locus as it is now
    std::vector<int64_t> callOff(1, 0);
    std::vector<uint16_t> calls;
    for (const base_call& bc : cleaned.calls)
    {
        uint16_t v;
        std::memcpy(&v, &bc, 2);
        calls.push_back(v);
    }
    callOff.push_back(static_cast<int64_t>(calls.size()));
    const std::vector<uint8_t> refBase(1, refBaseId(cleaned.get_ref_base()));
    const std::vector<uint8_t> pl(1, static_cast<uint8_t>(ploidy));
    sk_digt_call out;
    genotypeLoci(Access::opt(base), callOff, calls, refBase, pl, &out);
    toDiploidGenotype(out, ploidy, dgt);
    s.siteRecomputed++;
    s.siteRecomputeCalls++;
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
