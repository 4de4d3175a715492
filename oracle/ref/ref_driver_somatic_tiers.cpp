// ref_driver_somatic_tiers.cpp -- C entry points over the REFERENCE's own somatic callers, whole wrappers:
//   somatic_snv_caller_strand_grid::position_somatic_snv_call  (L/applications/strelka/position_somatic_snv_strand_grid.cpp:230-363)
//     with tier-2 pileups, per-site forced output and isComputeNonSomatic
//   somatic_indel_caller_grid::get_somatic_indel               (L/applications/strelka/somatic_indel_grid.cpp:181-361)
//     with the multi-indel-allele filter and the tier combination
//
// TEST INFRASTRUCTURE ONLY; contains no reference code.  Compiled with the unmodified reference sources into
// oracle/_ref/libstrelka_ref.so (oracle/Makefile, target `ref`).

#include "applications/strelka/position_somatic_snv_strand_grid.hh"
// included as a translation unit (like the reference's own unit tests do) to reach the file-static is_multi_indel_allele
#include "applications/strelka/somatic_indel_grid.cpp"
#include "applications/strelka/strelka_shared.hh"
#include "blt_common/snp_pos_info.hh"
#include "blt_util/reference_contig_segment.hh"
#include "starling_common/IndelData.hh"

#include <chrono>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace
{

void fill_pileup(snp_pos_info& pi, const uint16_t* calls, int n, char ref_base)
{
    pi.clear();
    pi.set_ref_base(ref_base);
    for (int i = 0; i < n; ++i) {
        const uint16_t c = calls[i];
        pi.calls.push_back(base_call((c >> 6) & 0xf, c & 0x3f, (c >> 10) & 1, 0, 0, (c >> 12) & 1, (c >> 11) & 1,
                                     (c >> 13) & 1));
    }
}

struct SnvOptions
{
    double bsnp_diploid_theta, somatic_snv_rate, shared_site_error_rate, shared_site_error_strand_bias_fraction,
        ssnv_contam_tolerance;
};

struct SnvGenotype // same layout as sko_somatic_snv_genotype
{
    uint32_t ref_gt;
    uint8_t snv_tier, snv_from_ntype_tier, is_forced_output, is_computed;
    uint32_t ntype, max_gt;
    int32_t qphred, from_ntype_qphred, nonsomatic_qphred;
    uint32_t normal_alt_id, tumor_alt_id;
    int32_t _pad;
    double strand_bias;
};

} // namespace

extern "C" {

/// one locus through the reference's position_somatic_snv_call.  n1/t1: cleaned tier1 pileups (normal, tumor); n2/t2:
/// cleaned tier1+tier2 pileups, used when is_tier2.  ref_base: 'A','C','G','T' or 'N'.
int ref_position_somatic_snv_call(const uint16_t* n1, int n_n1, const uint16_t* t1, int n_t1, const uint16_t* n2, int n_n2,
                                  const uint16_t* t2, int n_t2, int is_tier2, char ref_base, const SnvOptions* o,
                                  int is_forced_output, int is_compute_nonsomatic, SnvGenotype* out)
{
    try {
        strelka_options opt;
        opt.bsnp_diploid_theta = o->bsnp_diploid_theta;
        opt.somatic_snv_rate = o->somatic_snv_rate;
        opt.shared_site_error_rate = o->shared_site_error_rate;
        opt.shared_site_error_strand_bias_fraction = o->shared_site_error_strand_bias_fraction;
        opt.ssnv_contam_tolerance = o->ssnv_contam_tolerance;
        const somatic_snv_caller_strand_grid caller(opt);

        snp_pos_info pn1, pt1, pn2, pt2;
        fill_pileup(pn1, n1, n_n1, ref_base);
        fill_pileup(pt1, t1, n_t1, ref_base);
        fill_pileup(pn2, n2, n_n2, ref_base);
        fill_pileup(pt2, t2, n_t2, ref_base);
        static const std::vector<float> no_de;
        const extended_pos_info en1(pn1, no_de), et1(pt1, no_de), en2(pn2, no_de), et2(pt2, no_de);

        somatic_snv_genotype_grid sgt;
        sgt.is_forced_output = (is_forced_output != 0);
        // members without initialisers that the caller leaves untouched on its early returns
        sgt.rs.ntype = 0;
        sgt.rs.max_gt = 0;
        sgt.rs.normal_alt_id = 0;
        sgt.rs.tumor_alt_id = 0;
        caller.position_somatic_snv_call(en1, et1, is_tier2 ? &en2 : nullptr, is_tier2 ? &et2 : nullptr,
                                         is_compute_nonsomatic != 0, sgt);
        std::memset(out, 0, sizeof(*out));
        out->ref_gt = sgt.ref_gt;
        out->snv_tier = sgt.snv_tier;
        out->snv_from_ntype_tier = sgt.snv_from_ntype_tier;
        out->is_forced_output = sgt.is_forced_output;
        out->ntype = sgt.rs.ntype;
        out->max_gt = sgt.rs.max_gt;
        out->qphred = sgt.rs.qphred;
        out->from_ntype_qphred = sgt.rs.from_ntype_qphred;
        out->nonsomatic_qphred = sgt.rs.nonsomatic_qphred;
        out->normal_alt_id = sgt.rs.normal_alt_id;
        out->tumor_alt_id = sgt.rs.tumor_alt_id;
        out->strand_bias = sgt.rs.strandBias;
        return 0;
    } catch (...) {
        return 1;
    }
}

struct IndelSampleReads // same layout as sko_indel_sample_reads
{
    int32_t n_reads;
    const float* ref_lnp;
    const float* indel_lnp;
    const int32_t* alt_key;
    const float* alt_lnp;
    const uint16_t* non_ambig;
    const uint16_t* read_length;
    const uint8_t* is_tier1;
};
struct AltKey
{
    int32_t begin_pos, end_pos, is_mismatch;
};
struct SomaticIndelParams // same layout as sko_somatic_indel_params
{
    int32_t normal_min_read_bp_flank, tumor_min_read_bp_flank;
    double random_base_match_prob, tier2_random_base_match_prob;
    int32_t use_tier2_evidence, is_use_alt_indel;
    double bindel_diploid_theta, somatic_indel_rate, shared_indel_error_factor, indel_contam_tolerance;
};
struct IndelEnv
{
    strelka_options opt;
    std::unique_ptr<strelka_deriv_options> dopt;
    explicit IndelEnv(const SomaticIndelParams& p)
    {
        opt.randomBaseMatchProb = p.random_base_match_prob;
        opt.tier2.randomBaseMatchProb = p.tier2_random_base_match_prob;
        opt.useTier2Evidence = (p.use_tier2_evidence != 0);
        opt.bindel_diploid_theta = p.bindel_diploid_theta;
        opt.somatic_indel_rate = p.somatic_indel_rate;
        opt.shared_indel_error_factor = p.shared_indel_error_factor;
        opt.indel_contam_tolerance = p.indel_contam_tolerance;
        opt.alignFileOpt.alignmentFilenames = {"normal.bam", "tumor.bam"};
        opt.alignFileOpt.isAlignmentTumor = {false, true};
        dopt.reset(new strelka_deriv_options(opt));
    }
};

struct SomaticIndelGenotype
{
    uint8_t sindel_tier, sindel_from_ntype_tier, is_forced_output, is_overlap;
    uint32_t ntype, max_gt;
    int32_t qphred, from_ntype_qphred;
};

/// one candidate indel through the reference's get_somatic_indel.  The indel is a deletion of del_len / an insertion of
/// ins_len 'A's at position 1000; alternate keys are rebuilt from (begin_pos, end_pos, is_mismatch).  The tumor sample's
/// indelToRef error rate is written straight into its IndelSampleData (private member reached through the public
/// initializer being bypassed: see below).
int ref_get_somatic_indel(const IndelSampleReads* normal, const IndelSampleReads* tumor, const AltKey* alt_keys, int n_alt_keys,
                          unsigned del_len, unsigned ins_len, const SomaticIndelParams* p, double indel_to_ref_error_prob,
                          int is_forced_output, SomaticIndelGenotype* out, double* used_indel_to_ref_error_prob)
{
    try {
        static std::map<std::string, std::unique_ptr<IndelEnv>> envs;
        const std::string envKey(std::string(reinterpret_cast<const char*>(p), sizeof(*p)));
        std::unique_ptr<IndelEnv>& envp(envs[envKey]);
        if (!envp) envp.reset(new IndelEnv(*p));
        const strelka_options& opt(envp->opt);
        const strelka_deriv_options& dopt(*envp->dopt);
#if 0
        strelka_options opt;
        opt.randomBaseMatchProb = p->random_base_match_prob;
        opt.tier2.randomBaseMatchProb = p->tier2_random_base_match_prob;
        opt.useTier2Evidence = (p->use_tier2_evidence != 0);
        opt.bindel_diploid_theta = p->bindel_diploid_theta;
        opt.somatic_indel_rate = p->somatic_indel_rate;
        opt.shared_indel_error_factor = p->shared_indel_error_factor;
        opt.indel_contam_tolerance = p->indel_contam_tolerance;
        opt.alignFileOpt.alignmentFilenames = {"normal.bam", "tumor.bam"};
        opt.alignFileOpt.isAlignmentTumor = {false, true};
        const strelka_deriv_options dopt(opt);
#endif

        const std::string ins(ins_len, 'A');
        const IndelKey key(1000, INDEL::INDEL, del_len, ins.c_str());
        IndelData id(2, key);
        id.isForcedOutput = (is_forced_output != 0);
        // error rates: the reference's own model for this key in an 'N' context (initializeAuxInfo), reported back so that
        // the restatement is given the same value
        reference_contig_segment ref;
        ref.seq() = std::string(4000, 'N');
        id.initializeAuxInfo(opt, dopt, ref);
        *used_indel_to_ref_error_prob = id.getSampleData(1).getErrorRates().indelToRefErrorProb.getValue();
        (void)indel_to_ref_error_prob;

        std::vector<IndelKey> alts;
        for (int k = 0; k < n_alt_keys; ++k) {
            if (alt_keys[k].is_mismatch) alts.push_back(IndelKey(alt_keys[k].begin_pos, INDEL::MISMATCH, 1, "C"));
            else alts.push_back(IndelKey(alt_keys[k].begin_pos, INDEL::INDEL, unsigned(alt_keys[k].end_pos - alt_keys[k].begin_pos),
                                         (alt_keys[k].end_pos == alt_keys[k].begin_pos) ? "G" : ""));
        }
        const IndelSampleReads* smp[2] = {normal, tumor};
        for (unsigned s = 0; s < 2; ++s) {
            IndelSampleData& isd(id.getSampleData(s));
            for (int r = 0; r < smp[s]->n_reads; ++r) {
                ReadPathScores rps(smp[s]->ref_lnp[r], smp[s]->indel_lnp[r], smp[s]->non_ambig[r], smp[s]->read_length[r],
                                   smp[s]->is_tier1[r] != 0, true, 0, 0);
                for (int a = 0; a < 2; ++a) {
                    const int32_t k = smp[s]->alt_key[2 * r + a];
                    if (k < 0) continue;
                    rps.alt_indel.push_back(std::make_pair(alts[k], smp[s]->alt_lnp[2 * r + a]));
                }
                isd.read_path_lnp[static_cast<align_id_t>(r + 1 + 100000 * s)] = rps;
            }
        }
        starling_sample_options nopt(opt), topt(opt);
        nopt.min_read_bp_flank = p->normal_min_read_bp_flank;
        topt.min_read_bp_flank = p->tumor_min_read_bp_flank;
        somatic_indel_call sindel;
        sindel.sindel_tier = false;
        sindel.sindel_from_ntype_tier = false;
        sindel.rs.ntype = 0;
        sindel.rs.max_gt = 0;
        dopt.sicaller_grid().get_somatic_indel(opt, dopt, nopt, topt, key, id, 0, 1, p->is_use_alt_indel != 0, sindel);
        std::memset(out, 0, sizeof(*out));
        out->sindel_tier = sindel.sindel_tier;
        out->sindel_from_ntype_tier = sindel.sindel_from_ntype_tier;
        out->is_forced_output = sindel.is_forced_output;
        out->is_overlap = sindel.rs.is_overlap;
        out->ntype = sindel.rs.ntype;
        out->max_gt = sindel.rs.max_gt;
        out->qphred = sindel.rs.qphred;
        out->from_ntype_qphred = sindel.rs.from_ntype_qphred;
        return 0;
    } catch (...) {
        return 1;
    }
}

/// is_multi_indel_allele alone (file-static in somatic_indel_grid.cpp:102-177); returns 0/1, -1 on error
int ref_is_multi_indel_allele(const IndelSampleReads* normal, const IndelSampleReads* tumor, const AltKey* alt_keys, int n_alt_keys,
                              const SomaticIndelParams* p, int is_include_tier2, int* is_overlap)
{
    try {
        static std::map<std::string, std::unique_ptr<IndelEnv>> envs;
        std::unique_ptr<IndelEnv>& envp(envs[std::string(reinterpret_cast<const char*>(p), sizeof(*p))]);
        if (!envp) envp.reset(new IndelEnv(*p));
        const IndelKey key(1000, INDEL::INDEL, 1, "");
        IndelData id(2, key);
        std::vector<IndelKey> alts;
        for (int k = 0; k < n_alt_keys; ++k) {
            if (alt_keys[k].is_mismatch) alts.push_back(IndelKey(alt_keys[k].begin_pos, INDEL::MISMATCH, 1, "C"));
            else alts.push_back(IndelKey(alt_keys[k].begin_pos, INDEL::INDEL, unsigned(alt_keys[k].end_pos - alt_keys[k].begin_pos),
                                         (alt_keys[k].end_pos == alt_keys[k].begin_pos) ? "G" : ""));
        }
        const IndelSampleReads* smp[2] = {normal, tumor};
        for (unsigned s = 0; s < 2; ++s) {
            IndelSampleData& isd(id.getSampleData(s));
            for (int r = 0; r < smp[s]->n_reads; ++r) {
                ReadPathScores rps(smp[s]->ref_lnp[r], smp[s]->indel_lnp[r], smp[s]->non_ambig[r], smp[s]->read_length[r],
                                   smp[s]->is_tier1[r] != 0, true, 0, 0);
                for (int a = 0; a < 2; ++a) {
                    const int32_t k = smp[s]->alt_key[2 * r + a];
                    if (k < 0) continue;
                    rps.alt_indel.push_back(std::make_pair(alts[k], smp[s]->alt_lnp[2 * r + a]));
                }
                isd.read_path_lnp[static_cast<align_id_t>(r + 1 + 100000 * s)] = rps;
            }
        }
        bool overlap(*is_overlap != 0);
        const bool res(is_multi_indel_allele(*envp->dopt, id.getSampleData(0), id.getSampleData(1), is_include_tier2 != 0, overlap));
        *is_overlap = overlap ? 1 : 0;
        return res ? 1 : 0;
    } catch (...) {
        return -1;
    }
}

/// the reference's position_somatic_snv_call (tier1 columns only, as sk_somatic_snv_call_batch) over n loci; returns the
/// seconds spent inside the calls (pileups are materialised in snp_pos_info objects outside the clock)
double ref_time_somatic_sites(const int64_t* n_off, const uint16_t* n_calls, const int64_t* t_off, const uint16_t* t_calls,
                              const uint8_t* ref_base, int n_loci, const SnvOptions* o, double* checksum)
{
    strelka_options opt;
    opt.bsnp_diploid_theta = o->bsnp_diploid_theta;
    opt.somatic_snv_rate = o->somatic_snv_rate;
    opt.shared_site_error_rate = o->shared_site_error_rate;
    opt.shared_site_error_strand_bias_fraction = o->shared_site_error_strand_bias_fraction;
    opt.ssnv_contam_tolerance = o->ssnv_contam_tolerance;
    const somatic_snv_caller_strand_grid caller(opt);
    static const std::vector<float> no_de;
    static const char bases[] = "ACGTN";
    double secs = 0, acc = 0;
    for (int l = 0; l < n_loci; ++l) {
        snp_pos_info pn, pt;
        const char rb(bases[ref_base[l] > 4 ? 4 : ref_base[l]]);
        fill_pileup(pn, n_calls + n_off[l], int(n_off[l + 1] - n_off[l]), rb);
        fill_pileup(pt, t_calls + t_off[l], int(t_off[l + 1] - t_off[l]), rb);
        const extended_pos_info en(pn, no_de), et(pt, no_de);
        somatic_snv_genotype_grid sgt;
        const auto t0 = std::chrono::steady_clock::now();
        caller.position_somatic_snv_call(en, et, nullptr, nullptr, false, sgt);
        secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        acc += sgt.rs.qphred;
    }
    *checksum = acc;
    return secs;
}

} // extern "C"
