// sk_adapter_germline_indel.cpp -- site 4: getVariantAlleleGroupGenotypeLhoodsForSample
// (L/starling_common/AlleleGroupGenotype.cpp:185-258, called at L/applications/starling/starling_pos_processor.cpp:1384-1386)
// through sk_allele_group_genotype_lhoods.
//
// apply_hooks.py renames the reference's definition to ..._reference (it stays in the hooked translation unit, never called);
// this file provides the function under the original name, so every caller in the reference reaches the C-ABI.  The
// allele group's reads are resolved exactly as the reference does (getAlleleGroupSupportingReadIds: tier1 reads scored for
// every allele of the group, ascending read id; per allele the read's ReadPathScores ref / indel floats) and handed over
// as one CSR row; genotype likelihoods and the supporting-read counts come back.
//
// One group per call: indel loci are ~1e-3 of all loci and each call site sits inside position-ordered host logic whose
// inputs (which alleles form the group) depend on the calls made at earlier positions (_variantLocusAlreadyOutputToPos,
// starling_pos_processor.cpp:1618,1797), so the groups of a window are not known ahead of time.
#include "sk_adapter_access.hh"

#include "starling_common/AlleleGroupGenotype.hh"
#include "starling_common/OrthogonalVariantAlleleCandidateGroupUtil.hh"
#include "htsapi/vcf_util.hh"

#include <cmath>
#include <cstring>
#include <limits>

void
getVariantAlleleGroupGenotypeLhoodsForSample(
    const starling_base_options& opt,
    const starling_base_deriv_options& dopt,
    const starling_sample_options& sampleOptions,
    const unsigned callerPloidy,
    const unsigned sampleIndex,
    const OrthogonalVariantAlleleCandidateGroup& alleleGroup,
    const OrthogonalVariantAlleleCandidateGroup& contrastGroup,
    std::vector<double>& genotypeLogLhood,
    LocusSupportingReadStats& locusReadStats)
{
    using namespace sk_adapter;
    assert(callerPloidy > 0u);
    assert(callerPloidy < 3u);

    const uint8_t nonRefAlleleCount(alleleGroup.size());
    const uint8_t fullAlleleCount(nonRefAlleleCount + 1);
    const unsigned genotypeCount(VcfGenotypeUtil::getGenotypeCount(callerPloidy, fullAlleleCount));
    genotypeLogLhood.resize(genotypeCount);
    std::fill(genotypeLogLhood.begin(), genotypeLogLhood.end(), 0.);
    if (nonRefAlleleCount == 0) return;

    // A contrast group (AlleleGroupGenotype.cpp:225-252: alleles "intended for an 'other' category", e.g. a forced-output indel that
    // overlaps the called group) only ever raises a read's REFERENCE likelihood: the read's allele likelihoods are taken over the
    // extended group (getAlleleLogLhoodFromRead: the reference entry is the largest ReadPathScores::ref of every allele, contrast
    // ones included, that scored the read), then every contrast allele's own score is "maxed down into the reference" and dropped.
    // The reads are those scored for EVERY allele of the group proper (getAlleleGroupIntersectionReadIds), so none of its alleles is
    // ever missing for a read and the whole step is one more candidate for the maximum the kernel takes over the row's ref entries:
    // it is folded, as the float it is, into the first allele's ref entry below.
    // A multi-sample run can put up to ploidy x sample-count alternate alleles into one group (selectTopOrthogonalAllelesInAllSamples,
    // L/starling_common/OrthogonalVariantAlleleCandidateGroupUtil.cpp:285-340: the union of every sample's top alleles).  Groups of up to
    // SK_MAX_ALT alleles go through the narrow record, wider ones (several distinct overlapping indels that differ between the
    // samples) through sk_allele_group_genotype_lhoods_wide: the same kernel with rows of SK_MAX_ALT_WIDE.
    // Runs of five to eight samples can go on to SK_MAX_ALT_XWIDE = 16 (sk_allele_group_genotype_lhoods_xwide: 153 genotypes, three to a
    // lane); a group wider still -- nine or more samples whose top alleles at one locus are all distinct -- is the one refusal left here.
    if (nonRefAlleleCount > SK_MAX_ALT_XWIDE)
    {
        throw blt_exception("strelka_amd adapter: allele group with more than SK_MAX_ALT_XWIDE (16 = ploidy x 8 samples) alternate alleles");
    }
    const bool isWide(nonRefAlleleCount > SK_MAX_ALT);
    const bool isXWide(nonRefAlleleCount > SK_MAX_ALT_WIDE);
    const size_t width(isXWide ? SK_MAX_ALT_XWIDE : (isWide ? SK_MAX_ALT_WIDE : SK_MAX_ALT));
    init();

    locusReadStats.setAltCount(nonRefAlleleCount);

    static const bool isTier1Only(true);
    std::set<unsigned> readIds;
    getAlleleGroupSupportingReadIds(sampleIndex, alleleGroup, readIds, isTier1Only);

    const size_t readCount(readIds.size());
    const float notScored(std::numeric_limits<float>::quiet_NaN());
    std::vector<float> refLnp(readCount * width, 0.f), alleleLnp(readCount * width, notScored);
    std::vector<uint16_t> nonAmbig(readCount), readLength(readCount);
    std::vector<uint8_t> flags(readCount);
    size_t r(0);
    for (const unsigned readId : readIds)
    {
        bool isExemplarSet(false);
        for (unsigned a(0); a < nonRefAlleleCount; ++a)
        {
            const IndelSampleData& isd(alleleGroup.data(a).getSampleData(sampleIndex));
            const auto it(isd.read_path_lnp.find(readId));
            if (it == isd.read_path_lnp.end()) continue;
            const ReadPathScores& rps(it->second);
            refLnp[r * width + a] = rps.ref;
            alleleLnp[r * width + a] = rps.indel;
            if (! isExemplarSet)
            {
                // getExemplarReadScore (AlleleGroupGenotype.cpp:157-181): the first allele that scored the read
                nonAmbig[r] = rps.nonAmbiguousBasesInRead;
                readLength[r] = rps.read_length;
                flags[r] = static_cast<uint8_t>((rps.is_tier1_read ? SK_READ_TIER1 : 0) | (rps.is_fwd_strand ? SK_READ_FWD : 0));
                isExemplarSet = true;
            }
        }
        assert(isExemplarSet);
        if (contrastGroup.size() != 0)
        {
            float best(refLnp[r * width]);
            for (unsigned a(0); a < nonRefAlleleCount; ++a)
            {
                // (with the intersection of the alleles' reads every entry is set; were one missing, the reference's two-step maximum
                // could not be written as one: that read set is a compile-time choice of the reference, USE_GERMLINE_SUPPORTING_READ_UNION)
                if (std::isnan(alleleLnp[r * width + a])) throw blt_exception("strelka_amd adapter: a read of an allele group's intersection is missing from one of its alleles");
            }
            for (unsigned c(0); c < contrastGroup.size(); ++c)
            {
                const IndelSampleData& isd(contrastGroup.data(c).getSampleData(sampleIndex));
                const auto it(isd.read_path_lnp.find(readId));
                if (it == isd.read_path_lnp.end()) continue; // (an allele that did not score the read takes the reference entry: no new candidate)
                best = std::max(best, std::max(it->second.ref, it->second.indel));
            }
            refLnp[r * width] = best;
        }
        ++r;
    }

    const int64_t readOff[2] = {0, static_cast<int64_t>(readCount)};
    const uint8_t nAlt(nonRefAlleleCount), ploidy(static_cast<uint8_t>(callerPloidy));
    uint32_t delLen[SK_MAX_ALT_XWIDE] = {0}, insLen[SK_MAX_ALT_XWIDE] = {0};
    for (unsigned a(0); a < nonRefAlleleCount; ++a)
    {
        const IndelKey& k(alleleGroup.key(a));
        delLen[a] = k.delete_length();
        insLen[a] = k.insert_length();
    }
    static const float noFloat(0.f);
    static const uint16_t noU16(0);
    static const uint8_t noU8(0);
    sk_allele_group_batch b;
    std::memset(&b, 0, sizeof(b));
    b.n_groups = 1;
    b.read_off = readOff;
    b.n_alt = &nAlt;
    b.ploidy = &ploidy;
    b.del_len = delLen;
    b.ins_len = insLen;
    b.ref_lnp = readCount ? refLnp.data() : &noFloat;
    b.allele_lnp = readCount ? alleleLnp.data() : &noFloat;
    b.non_ambig = readCount ? nonAmbig.data() : &noU16;
    b.read_length = readCount ? readLength.data() : &noU16;
    b.read_flags = readCount ? flags.data() : &noU8;

    sk_indel_options io;
    sk_indel_options_default(&io, opt.isSomaticCallingMode ? 1 : 0);
    io.min_read_bp_flank = sampleOptions.min_read_bp_flank;
    io.random_base_match_prob = opt.randomBaseMatchProb;
    io.tier2_random_base_match_prob = opt.tier2.randomBaseMatchProb;
    io.read_confident_support_threshold = opt.readConfidentSupportThreshold.numval();

    // (the three records differ in their array sizes only: read through one view)
    sk_allele_group_call narrow;
    sk_allele_group_call_wide wide;
    static sk_allele_group_call_xwide xwide; // (1.4 KB: not on the stack of a function called per indel locus)
    {
        AccumTimer abiTimer(state().tIndelAbi);
        if (isXWide) check(sk_allele_group_genotype_lhoods_xwide(&b, &io, &xwide), "sk_allele_group_genotype_lhoods_xwide");
        else if (isWide) check(sk_allele_group_genotype_lhoods_wide(&b, &io, &wide), "sk_allele_group_genotype_lhoods_wide");
        else check(sk_allele_group_genotype_lhoods(&b, &io, &narrow), "sk_allele_group_genotype_lhoods");
    }
    const unsigned outGenotypes(isXWide ? xwide.n_genotypes : (isWide ? wide.n_genotypes : narrow.n_genotypes));
    const double* const outLhood(isXWide ? xwide.lhood : (isWide ? wide.lhood : narrow.lhood));
    if (outGenotypes != genotypeCount)
    {
        throw blt_exception("strelka_amd adapter: genotype count mismatch in sk_allele_group_genotype_lhoods");
    }
    for (unsigned g(0); g < genotypeCount; ++g) genotypeLogLhood[g] = outLhood[g];
    for (unsigned s(0); s < 2; ++s)
    {
        auto& counts(locusReadStats.getCounts(s == 0));
        const uint32_t* const outCounts(isXWide ? xwide.counts[s] : (isWide ? wide.counts[s] : narrow.counts[s]));
        for (unsigned a(0); a < fullAlleleCount; ++a) counts.incrementAlleleCount(a, outCounts[a]);
        counts.nonConfidentCount += outCounts[fullAlleleCount];
    }
    if (isWide) state().indelGroupsWide++;
    if (isXWide) state().indelGroupsXWide++;
    state().indelGroups++;
}
