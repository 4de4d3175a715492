// sk_adapter_somatic.cpp -- site 5 of the somatic caller: somatic_snv_caller_strand_grid::position_somatic_snv_call
// (L/applications/strelka/strelka_pos_processor.cpp:213-219) for a whole stage window in one sk_somatic_snv_call_tiers call.
//
// When the (deferred) POST_ALIGN stage reaches position P the pileup columns of [P, P+W] are complete, so the four cleaned
// columns per locus -- normal / tumor, CleanPileupFilter(pi,false) and, with tier2 evidence, CleanPileupFilter(pi,true)
// (L/starling_common/PileupCleaner.cpp:28-66) -- are packed into four sk_pileup_batch and the whole wrapper (early return,
// both tiers, tier selection, NTYPE conflict, forced output, non-somatic quality) runs in one call.  The reference's
// per-position control flow is unchanged: process_pos_snp_somatic(P+k) asks for its somatic_snv_genotype_grid and gets the
// cached record after a check that the columns it was computed from are the ones the reference holds now.
#include "sk_adapter_access.hh"

#include "applications/strelka/strelka_pos_processor.hh"
#include "applications/strelka/strelka_shared.hh"
#include "blt_util/seq_util.hh"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>

namespace sk_adapter
{

namespace
{

unsigned env_unsigned(const char* name, const unsigned def)
{
    const char* v(std::getenv(name));
    if (v == nullptr || *v == 0) return def;
    return static_cast<unsigned>(std::strtoul(v, nullptr, 10));
}

const strelka_options& strelkaOptions(const starling_pos_processor_base& pp)
{
    return dynamic_cast<const strelka_options&>(Access::opt(pp));
}

void somaticSnvOptions(const strelka_options& opt, sk_somatic_snv_options& so)
{
    sk_somatic_snv_options_default(&so);
    so.bsnp_diploid_theta = opt.bsnp_diploid_theta;
    so.somatic_snv_rate = opt.somatic_snv_rate;
    so.shared_site_error_rate = opt.shared_site_error_rate;
    so.shared_site_error_strand_bias_fraction = opt.shared_site_error_strand_bias_fraction;
    so.ssnv_contam_tolerance = opt.ssnv_contam_tolerance;
}

inline uint16_t bits(const base_call& bc)
{
    uint16_t v;
    std::memcpy(&v, &bc, 2);
    return v;
}

/// CleanPileupFilter(pi, is_include_tier2) (PileupCleaner.cpp:28-66)
void appendCleaned(const snp_pos_info& pi, const bool isIncludeTier2, std::vector<uint16_t>& calls)
{
    for (const base_call& bc : pi.calls)
    {
        if (bc.is_call_filter)
        {
            if (! (isIncludeTier2 && bc.is_tier_specific_call_filter)) continue;
        }
        calls.push_back(bits(bc));
    }
    if (isIncludeTier2)
    {
        for (const base_call& bc : pi.tier2_calls)
        {
            if (bc.is_call_filter) continue;
            calls.push_back(bits(bc));
        }
    }
}

struct Columns
{
    std::vector<int64_t> off;
    std::vector<uint16_t> calls;
    Columns() : off(1, 0) {}
    void close() { off.push_back(static_cast<int64_t>(calls.size())); }
    void fill(sk_pileup_batch& b, const std::vector<uint8_t>& refBase) const
    {
        static const uint16_t none(0);
        std::memset(&b, 0, sizeof(b));
        b.n_loci = static_cast<int32_t>(refBase.size());
        b.call_off = off.data();
        b.calls = calls.empty() ? &none : calls.data();
        b.ref_base = refBase.data();
    }
};

void callLoci(const strelka_options& opt, const Columns col[4], const std::vector<uint8_t>& refBase,
              const std::vector<uint8_t>& forced, const bool isComputeNonSomatic, sk_somatic_snv_genotype* out)
{
    sk_somatic_snv_options so;
    somaticSnvOptions(opt, so);
    sk_pileup_batch b[4];
    for (unsigned i(0); i < 4; ++i) col[i].fill(b[i], refBase);
    const bool isTier2(opt.useTier2Evidence);
    check(sk_somatic_snv_call_tiers(&b[0], &b[1], isTier2 ? &b[2] : nullptr, isTier2 ? &b[3] : nullptr, &so, forced.data(),
                                    isComputeNonSomatic ? 1 : 0, out), "sk_somatic_snv_call_tiers");
}

uint8_t refBaseId(const char refBase)
{
    const unsigned id(base_to_id(refBase));
    return static_cast<uint8_t>(id < 4 ? id : 4);
}

void toGenotypeGrid(const sk_somatic_snv_genotype& g, somatic_snv_genotype_grid& sgt)
{
    sgt.snv_tier = (g.snv_tier != 0);
    sgt.snv_from_ntype_tier = (g.snv_from_ntype_tier != 0);
    sgt.ref_gt = g.ref_gt;
    sgt.is_forced_output = (g.is_forced_output != 0);
    sgt.rs.ntype = g.ntype;
    sgt.rs.max_gt = g.max_gt;
    sgt.rs.qphred = g.qphred;
    sgt.rs.from_ntype_qphred = g.from_ntype_qphred;
    sgt.rs.nonsomatic_qphred = g.nonsomatic_qphred;
    sgt.rs.normal_alt_id = g.normal_alt_id;
    sgt.rs.tumor_alt_id = g.tumor_alt_id;
    sgt.rs.strandBias = g.strand_bias;
}

}

bool somatic_stream_options(const starling_pos_processor_base& pp, sk_somatic_snv_options& so, bool& isComputeNonSomatic)
{
    const strelka_options& opt(strelkaOptions(pp));
    somaticSnvOptions(opt, so);
    isComputeNonSomatic = opt.is_somatic_callable();
    return opt.is_somatic_snv();
}

void somatic_window(starling_pos_processor_base& pp, const pos_t pos)
{
    const strelka_options& opt(strelkaOptions(pp));
    pileup_before_variants(pp, pos);
    if (! opt.is_somatic_snv()) return;

    State& s(state());
    if (s.pileup.isGenotyping) return; // the records came with the pileup (site 9)
    SomaticSiteCache& cache(s.somaticSites);
    if (pos >= cache.begin && pos < cache.end) return;
    AccumTimer hookTimer(s.tSiteHook);

    using namespace STRELKA_SAMPLE_TYPE;
    const pos_t begin(pos), end(pos + static_cast<pos_t>(post_align_defer()) + 1);
    const size_t n(static_cast<size_t>(end - begin));
    cache.begin = begin;
    cache.end = end;
    cache.isValid.assign(n, 0);
    cache.forced.assign(n, 0);
    cache.callCount.assign(n * 4, 0);
    cache.genotypes.resize(n);

    Columns col[4]; // normal t1, tumor t1, normal t2, tumor t2
    std::vector<uint8_t> refBase, forced;
    std::vector<size_t> slot;
    const bool isTier2(opt.useTier2Evidence);
    for (pos_t p(begin); p < end; ++p)
    {
        if (! Access::isPosReportable(pp, p)) continue;
        const snp_pos_info& npi(pp.sample(NORMAL).basecallBuffer.get_pos(p));
        const snp_pos_info& tpi(pp.sample(TUMOR).basecallBuffer.get_pos(p));
        const size_t k(static_cast<size_t>(p - begin));
        const snp_pos_info* pis[2] = {&npi, &tpi};
        for (unsigned t(0); t < 2; ++t)
        {
            if (t == 1 && ! isTier2) continue;
            for (unsigned si(0); si < 2; ++si)
            {
                Columns& c(col[2 * t + si]);
                const size_t before(c.calls.size());
                appendCleaned(*pis[si], t == 1, c.calls);
                c.close();
                cache.callCount[4 * k + 2 * t + si] = static_cast<uint32_t>(c.calls.size() - before);
            }
        }
        refBase.push_back(refBaseId(npi.get_ref_base()));
        const uint8_t isForced(Access::isForcedOutputPos(pp, p) ? 1 : 0);
        forced.push_back(isForced);
        cache.forced[k] = isForced;
        slot.push_back(k);
    }
    if (slot.empty()) return;
    if (! isTier2)
    {
        col[2] = col[0];
        col[3] = col[1];
    }
    std::vector<sk_somatic_snv_genotype> out(slot.size());
    {
        AccumTimer abiTimer(s.tSiteAbi);
        callLoci(opt, col, refBase, forced, opt.is_somatic_callable(), out.data());
    }
    for (size_t i(0); i < slot.size(); ++i)
    {
        cache.genotypes[slot[i]] = out[i];
        cache.isValid[slot[i]] = 1;
    }
    s.siteBatches++;
    s.siteLoci += slot.size();
}

namespace
{

/// the chunk of the pileup stream that covers pos (POST_ALIGN only moves forward: chunks behind it are dropped)
SomaticChunk* somaticChunkFor(const pos_t pos)
{
    std::deque<SomaticChunk>& chunks(state().pileup.somaticChunks);
    while ((! chunks.empty()) && chunks.front().end <= pos) chunks.pop_front();
    if ((! chunks.empty()) && chunks.front().begin <= pos) return &chunks.front();
    return nullptr;
}

}

bool somatic_defer_clean(starling_pos_processor_base& /*pp*/, const pos_t /*pos*/)
{
    static const bool isLazy(env_unsigned("STRELKA_AMD_LAZY_CLEAN", 1) != 0);
    const PileupState& ps(state().pileup);
    return isLazy && ps.enabled && ps.isSomatic && ps.isGenotyping;
}

void somatic_clean_now(starling_pos_processor_base& pp, const pos_t pos, CleanedPileup* const* normalCpi, CleanedPileup* const* tumorCpi)
{
    using namespace STRELKA_SAMPLE_TYPE;
    const starling_base_options& opt(Access::opt(pp));
    const PileupCleaner& cleaner(Access::pileupCleaner(pp));
    for (unsigned t(0); t < 2; ++t) // strelka_pos_processor.cpp:180-186
    {
        const bool isIncludeTier2(t != 0);
        if (isIncludeTier2 && (! opt.useTier2Evidence)) continue;
        cleaner.CleanPileup(pp.sample(NORMAL).basecallBuffer.get_pos(pos), isIncludeTier2, *(normalCpi[t]));
        cleaner.CleanPileup(pp.sample(TUMOR).basecallBuffer.get_pos(pos), isIncludeTier2, *(tumorCpi[t]));
    }
}

bool sample_stats_counts(starling_pos_processor_base& pp, const pos_t pos, const unsigned sampleIndex, unsigned& used, unsigned& unused)
{
    // the germline caller: a plain site's counts come from the window (site 10, sk_adapter_gvcf.cpp)
    if (! Access::opt(pp).isSomaticCallingMode) return germline_sample_stats_counts(pp, pos, sampleIndex, used, unused);
    if (! somatic_defer_clean(pp, pos)) return false;
    const snp_pos_info& pi(pp.sample(sampleIndex).basecallBuffer.get_pos(pos));
    if (pi.calls.empty())
    {
        used = unused = 0;
        return true;
    }
    const SomaticChunk* c(somaticChunkFor(pos));
    if (c == nullptr) return false;
    const size_t k(static_cast<size_t>(pos - c->begin));
    if (sampleIndex > 1 || c->rawCount[sampleIndex][k] != pi.calls.size()) return false;
    // CleanedPileup::usedBasecallCount / unusedBasecallCount of CleanPileupFilter(pi, false) (PileupCleaner.hh:48-58)
    used = c->count[sampleIndex][k];
    unused = static_cast<unsigned>(pi.calls.size()) - used;
    return true;
}

void somatic_snv_genotype(starling_pos_processor_base& pp, const pos_t pos, CleanedPileup* const* normalCpi, CleanedPileup* const* tumorCpi,
                          bool& isCleanDeferred, const bool isComputeNonSomatic, somatic_snv_genotype_grid& sgt)
{
    using namespace STRELKA_SAMPLE_TYPE;
    State& s(state());
    SomaticSiteCache& cache(s.somaticSites);
    const strelka_options& opt(strelkaOptions(pp));
    const bool isTier2(opt.useTier2Evidence);
    const uint8_t isForced(sgt.is_forced_output ? 1 : 0);
    if (s.pileup.isGenotyping)
    {
        const SomaticChunk* c(somaticChunkFor(pos));
        if (c != nullptr)
        {
            const size_t k(static_cast<size_t>(pos - c->begin));
            bool ok(c->forced[k] == isForced);
            if (isCleanDeferred)
            {
                // the record was computed from the columns the stream wrote into the reference's buffers: still those?
                const snp_pos_info& npi(pp.sample(NORMAL).basecallBuffer.get_pos(pos));
                const snp_pos_info& tpi(pp.sample(TUMOR).basecallBuffer.get_pos(pos));
                ok = ok && c->rawCount[0][k] == npi.calls.size() && c->rawCount[1][k] == tpi.calls.size() &&
                     c->rawCount[2][k] == npi.tier2_calls.size() && c->rawCount[3][k] == tpi.tier2_calls.size();
            }
            else
            {
                const CleanedPileup* cpis[4] = {normalCpi[0], tumorCpi[0], isTier2 ? normalCpi[1] : nullptr, isTier2 ? tumorCpi[1] : nullptr};
                for (unsigned i(0); ok && i < 4; ++i)
                {
                    if (cpis[i] == nullptr) continue;
                    ok = (c->count[i][k] == cpis[i]->cleanedPileup().calls.size());
                }
            }
            if (ok)
            {
                toGenotypeGrid(c->genotypes[k], sgt);
                return;
            }
        }
    }
    if (isCleanDeferred)
    {
        somatic_clean_now(pp, pos, normalCpi, tumorCpi);
        isCleanDeferred = false;
    }
    const CleanedPileup* const cpis[4] = {normalCpi[0], tumorCpi[0], normalCpi[1], tumorCpi[1]};
    const snp_pos_info* cleaned[4] = {&cpis[0]->cleanedPileup(), &cpis[1]->cleanedPileup(),
                                      isTier2 ? &cpis[2]->cleanedPileup() : nullptr, isTier2 ? &cpis[3]->cleanedPileup() : nullptr};
    if ((! s.pileup.isGenotyping) && pos >= cache.begin && pos < cache.end)
    {
        const size_t k(static_cast<size_t>(pos - cache.begin));
        bool ok(cache.isValid[k] && cache.forced[k] == isForced);
        for (unsigned i(0); ok && i < 4; ++i)
        {
            if (cleaned[i] == nullptr) continue;
            ok = (cache.callCount[4 * k + i] == cleaned[i]->calls.size());
        }
        if (ok)
        {
            toGenotypeGrid(cache.genotypes[k], sgt);
            return;
        }
    }
    // not covered by the window (or the site changed since): call the locus as the reference holds it now
    Columns col[4];
    for (unsigned i(0); i < 4; ++i)
    {
        const snp_pos_info* pi(cleaned[i] ? cleaned[i] : cleaned[i - 2]);
        for (const base_call& bc : pi->calls) col[i].calls.push_back(bits(bc));
        col[i].close();
    }
    const std::vector<uint8_t> refBase(1, refBaseId(cleaned[0]->get_ref_base()));
    const std::vector<uint8_t> forced(1, isForced);
    sk_somatic_snv_genotype out;
    callLoci(opt, col, refBase, forced, isComputeNonSomatic, &out);
    toGenotypeGrid(out, sgt);
    s.siteRecomputed++;
}


// ---- site 6: somatic_indel_caller_grid::get_somatic_indel at strelka_pos_processor.cpp:343-349 ------------------------------
//
// One candidate indel per call, as the reference calls it (candidate indels with read support are ~1e-3 of the loci and the
// call sits inside position-ordered host logic).  The two samples' IndelSampleData::read_path_lnp rows go over as CSR rows
// in read-id order, the reads' alternate-indel lists as indices into a table of the distinct alternate IndelKeys.
void somatic_indel(const strelka_options& opt, const starling_sample_options& normalOpt, const starling_sample_options& tumorOpt,
                   const IndelKey& indelKey, const IndelData& indelData, const unsigned normalSampleIndex,
                   const unsigned tumorSampleIndex, const bool isUseAltIndel, somatic_indel_call& sindel)
{
    init();
    if (indelKey.is_breakpoint()) throw blt_exception("strelka_amd adapter: breakpoint alleles are not supported on this path");

    std::map<IndelKey, int32_t> altIndex;
    std::vector<sk_alt_allele> alleles;
    struct Rows
    {
        std::vector<float> ref, indel, best;
        std::vector<int32_t> altKey;
        std::vector<float> altLnp;
        std::vector<uint16_t> nonAmbig, readLength;
        std::vector<uint8_t> flags;
    } rows[2];
    const unsigned sampleIndex[2] = {normalSampleIndex, tumorSampleIndex};
    for (unsigned s(0); s < 2; ++s)
    {
        Rows& R(rows[s]);
        for (const auto& val : indelData.getSampleData(sampleIndex[s]).read_path_lnp)
        {
            const ReadPathScores& rps(val.second);
            R.ref.push_back(rps.ref);
            R.indel.push_back(rps.indel);
            R.nonAmbig.push_back(rps.nonAmbiguousBasesInRead);
            R.readLength.push_back(rps.read_length);
            R.flags.push_back(static_cast<uint8_t>((rps.is_tier1_read ? SK_READ_TIER1 : 0) | (rps.is_fwd_strand ? SK_READ_FWD : 0)));
            float best(std::numeric_limits<float>::quiet_NaN());
            if (rps.alt_indel.size() > 2) throw blt_exception("strelka_amd adapter: more than two alternate indels on a read");
            for (unsigned a(0); a < 2; ++a)
            {
                if (a >= rps.alt_indel.size())
                {
                    R.altKey.push_back(-1);
                    R.altLnp.push_back(0.f);
                    continue;
                }
                const IndelKey& ak(rps.alt_indel[a].first);
                if (ak.is_breakpoint()) throw blt_exception("strelka_amd adapter: breakpoint alleles are not supported on this path");
                auto it(altIndex.find(ak));
                if (it == altIndex.end())
                {
                    it = altIndex.insert(std::make_pair(ak, static_cast<int32_t>(alleles.size()))).first;
                    sk_alt_allele al;
                    al.begin_pos = ak.pos;
                    al.end_pos = ak.right_pos();
                    al.is_mismatch = ak.isMismatch() ? 1 : 0;
                    alleles.push_back(al);
                }
                R.altKey.push_back(it->second);
                R.altLnp.push_back(rps.alt_indel[a].second);
                const float v(rps.alt_indel[a].second);
                if (!(best == best) || best < v) best = v;
            }
            R.best.push_back(best);
        }
    }

    const uint32_t delLen(indelKey.delete_length()), insLen(indelKey.insert_length());
    static const float noFloat(0.f);
    static const uint16_t noU16(0);
    static const uint8_t noU8(0);
    static const int32_t noI32(-1);
    int64_t readOff[2][2];
    auto fill = [&](const unsigned s, sk_readscore_batch& b)
    {
        const Rows& R(rows[s]);
        std::memset(&b, 0, sizeof(b));
        readOff[s][0] = 0;
        readOff[s][1] = static_cast<int64_t>(R.ref.size());
        b.n_indels = 1;
        b.read_off = readOff[s];
        const bool any(! R.ref.empty());
        b.ref_lnp = any ? R.ref.data() : &noFloat;
        b.indel_lnp = any ? R.indel.data() : &noFloat;
        b.alt_lnp = any ? R.best.data() : &noFloat;
        b.non_ambig = any ? R.nonAmbig.data() : &noU16;
        b.read_length = any ? R.readLength.data() : &noU16;
        b.read_flags = any ? R.flags.data() : &noU8;
        b.del_len = &delLen;
        b.ins_len = &insLen;
        b.is_breakpoint = nullptr;
    };
    sk_somatic_indel_batch sb;
    std::memset(&sb, 0, sizeof(sb));
    sb.n_indels = 1;
    fill(0, sb.normal);
    fill(1, sb.tumor);
    sb.normal_alt_key = rows[0].altKey.empty() ? &noI32 : rows[0].altKey.data();
    sb.normal_alt_lnp = rows[0].altLnp.empty() ? &noFloat : rows[0].altLnp.data();
    sb.tumor_alt_key = rows[1].altKey.empty() ? &noI32 : rows[1].altKey.data();
    sb.tumor_alt_lnp = rows[1].altLnp.empty() ? &noFloat : rows[1].altLnp.data();
    const int64_t altOff[2] = {0, static_cast<int64_t>(alleles.size())};
    static const sk_alt_allele noAllele = {0, 0, 0};
    sb.alt_off = altOff;
    sb.alt_alleles = alleles.empty() ? &noAllele : alleles.data();
    const double indelToRef(indelData.getSampleData(tumorSampleIndex).getErrorRates().indelToRefErrorProb.getValue());
    sb.indel_to_ref_error_prob = &indelToRef;
    const uint8_t forced(indelData.isForcedOutput ? 1 : 0);
    sb.is_forced_output = &forced;

    sk_indel_options no, to;
    sk_indel_options_default(&no, 1);
    no.min_read_bp_flank = normalOpt.min_read_bp_flank;
    no.random_base_match_prob = opt.randomBaseMatchProb;
    no.tier2_random_base_match_prob = opt.tier2.randomBaseMatchProb;
    no.read_confident_support_threshold = opt.readConfidentSupportThreshold.numval();
    no.is_use_alt_indel = isUseAltIndel ? 1 : 0;
    {
        // the library's default is the fast form of the 21-state likelihoods (include/strelka_amd.h, sk_indel_options.fast_form);
        // $STRELKA_AMD_INDEL_EXACT=1: the reference's operation order, bit-identical doubles
        static const bool isExact([]() { const char* v(std::getenv("STRELKA_AMD_INDEL_EXACT")); return v && *v && *v != '0'; }());
        if (isExact) no.fast_form = 0;
    }
    to = no;
    to.min_read_bp_flank = tumorOpt.min_read_bp_flank;
    sk_somatic_indel_options so;
    sk_somatic_indel_options_default(&so);
    so.bindel_diploid_theta = opt.bindel_diploid_theta;
    so.somatic_indel_rate = opt.somatic_indel_rate;
    so.shared_indel_error_factor = opt.shared_indel_error_factor;
    so.indel_contam_tolerance = opt.indel_contam_tolerance;

    sk_somatic_indel_genotype g;
    check(sk_somatic_indel_call_tiers(&sb, &no, &to, &so, opt.useTier2Evidence ? 1 : 0, &g), "sk_somatic_indel_call_tiers");
    sindel.is_forced_output = indelData.isForcedOutput;
    sindel.sindel_tier = (g.sindel_tier != 0);
    sindel.sindel_from_ntype_tier = (g.sindel_from_ntype_tier != 0);
    sindel.rs.ntype = g.ntype;
    sindel.rs.max_gt = g.max_gt;
    sindel.rs.qphred = g.qphred;
    sindel.rs.from_ntype_qphred = g.from_ntype_qphred;
    sindel.rs.is_overlap = (g.is_overlap != 0);
    state().indelGroups++;
}

}
