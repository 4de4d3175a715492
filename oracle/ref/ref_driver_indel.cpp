// ref_driver_indel.cpp -- C entry points over the REFERENCE's own indel genotype-likelihood code (hot path B, indels).
// TEST INFRASTRUCTURE ONLY; contains no reference code (see ref_driver_pathb.cpp).

#include "applications/strelka/qscore_calculator.hh"
#include "applications/strelka/strelka_digt_states.hh"
#include "starling_common/AlleleGroupGenotype.hh"
#include "starling_common/IndelData.hh"
#include "starling_common/IndelKey.hh"
#include "starling_common/OrthogonalVariantAlleleCandidateGroup.hh"
#include "starling_common/starling_indel_call_pprob_digt.hh"
#include "test/starling_base_options_test.hh"

#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace
{
struct Env
{
    starling_base_options_test opt;
    std::unique_ptr<starling_base_deriv_options> dopt;
    Env(const double randomBaseMatchProb)
    {
        opt.randomBaseMatchProb = randomBaseMatchProb;
        dopt.reset(new starling_base_deriv_options(opt));
    }
};

Env& env(const double randomBaseMatchProb)
{
    static std::map<double, std::unique_ptr<Env>> cache;
    auto& p = cache[randomBaseMatchProb];
    if (!p) p.reset(new Env(randomBaseMatchProb));
    return *p;
}

IndelKey make_key(int pos, unsigned del_len, const char* ins)
{
    return IndelKey(pos, INDEL::INDEL, del_len, ins ? ins : "");
}
} // namespace

extern "C" {

/// get_het_observed_allele_ratio (starling_indel_call_pprob_digt.cpp:40-71)
void ref_het_observed_allele_ratio(unsigned read_length, unsigned min_overlap, unsigned del_len, const char* ins_seq,
                                   double het_allele_ratio, double* log_ref_prob, double* log_indel_prob)
{
    const IndelKey key(make_key(10, del_len, ins_seq));
    get_het_observed_allele_ratio(read_length, min_overlap, key, het_allele_ratio, *log_ref_prob, *log_indel_prob);
}

/// one sample's 21 somatic-grid indel likelihoods: get_indel_digt_lhood (3 states) + the 18 het-grid states through
/// get_high_low_het_ratio_lhood, exactly as get_indel_het_grid_lhood (somatic_indel_grid.cpp:66-89, file-static) does.
/// alt_lnp: per-read best alternate-indel score, NaN = the read has no alt_indel entry.
void ref_indel_grid_lhood(int n_reads, const float* ref_lnp, const float* indel_lnp, const float* alt_lnp,
                          const uint16_t* non_ambig, const uint16_t* read_length, const uint8_t* is_tier1,
                          unsigned del_len, const char* ins_seq, int min_read_bp_flank, double randomBaseMatchProb,
                          int is_include_tier2, int is_use_alt_indel, double* lhood21)
{
    Env& e = env(randomBaseMatchProb);
    const IndelKey key(make_key(100, del_len, ins_seq));
    const IndelKey altKey(make_key(101, 1, ""));
    IndelData id(1, key);
    IndelSampleData& isd(id.getSampleData(0));
    for (int r = 0; r < n_reads; ++r) {
        ReadPathScores rps(ref_lnp[r], indel_lnp[r], non_ambig[r], read_length[r], is_tier1[r] != 0, true, 0, 0);
        if (alt_lnp[r] == alt_lnp[r]) rps.insertAlt(altKey, alt_lnp[r]);
        isd.read_path_lnp[r] = rps;
    }
    starling_sample_options sopt(e.opt);
    sopt.min_read_bp_flank = min_read_bp_flank;
    get_indel_digt_lhood(e.opt, *e.dopt, sopt, key, isd, is_include_tier2 != 0, is_use_alt_indel != 0, lhood21);
    const unsigned lsize(DIGT_GRID::HET_RES * 2);
    double* grid = lhood21 + SOMATIC_DIGT::SIZE;
    for (unsigned i = 0; i < DIGT_GRID::HET_RES; ++i) {
        const double het_ratio((i + 1) * DIGT_GRID::RATIO_INCREMENT);
        get_high_low_het_ratio_lhood(e.opt, *e.dopt, sopt, key, isd, het_ratio, is_include_tier2 != 0,
                                     is_use_alt_indel != 0, grid[lsize - (i + 1)], grid[i]);
    }
}

/// getVariantAlleleGroupGenotypeLhoodsForSample (AlleleGroupGenotype.cpp:185-258) for one allele group.
/// allele_lnp: [n_reads][n_alt] indel scores, NaN = read not scored for that allele; ref_lnp: [n_reads][n_alt].
/// out_counts: [2 strands][n_alt+2]: per-allele confident counts (ref first) then the non-confident count.
void ref_allele_group_genotype_lhoods(int n_reads, int n_alt, const float* ref_lnp, const float* allele_lnp,
                                      const uint16_t* non_ambig, const uint16_t* read_length, const uint8_t* is_tier1,
                                      const uint8_t* is_fwd, const unsigned* del_len, const char* const* ins_seq,
                                      int ploidy, int min_read_bp_flank, double randomBaseMatchProb, double* out_lhood,
                                      unsigned* out_counts)
{
    Env& e = env(randomBaseMatchProb);
    typedef std::map<IndelKey, IndelData> map_t;
    map_t buffer;
    std::vector<map_t::iterator> iters;
    for (int a = 0; a < n_alt; ++a) {
        const IndelKey key(make_key(100 + a, del_len[a], ins_seq[a]));
        iters.push_back(buffer.insert(std::make_pair(key, IndelData(1, key))).first);
    }
    for (int a = 0; a < n_alt; ++a) {
        IndelSampleData& isd(iters[a]->second.getSampleData(0));
        for (int r = 0; r < n_reads; ++r) {
            const float s = allele_lnp[r * n_alt + a];
            if (!(s == s)) continue;
            isd.read_path_lnp[r] = ReadPathScores(ref_lnp[r * n_alt + a], s, non_ambig[r], read_length[r],
                                                  is_tier1[r] != 0, is_fwd[r] != 0, 0, 0);
        }
    }
    OrthogonalVariantAlleleCandidateGroup group, contrast;
    for (int a = 0; a < n_alt; ++a) group.addVariantAllele(iters[a]);
    starling_sample_options sopt(e.opt);
    sopt.min_read_bp_flank = min_read_bp_flank;
    std::vector<double> lhood;
    LocusSupportingReadStats stats;
    getVariantAlleleGroupGenotypeLhoodsForSample(e.opt, *e.dopt, sopt, ploidy, 0, group, contrast, lhood, stats);
    for (size_t i = 0; i < lhood.size(); ++i) out_lhood[i] = lhood[i];
    for (int s = 0; s < 2; ++s) {
        const auto& c(stats.getCounts(s == 0));
        for (int a = 0; a <= n_alt; ++a) out_counts[s * (n_alt + 2) + a] = c.confidentAlleleCount(a);
        out_counts[s * (n_alt + 2) + n_alt + 1] = c.nonConfidentCount;
    }
}

} // extern "C"
