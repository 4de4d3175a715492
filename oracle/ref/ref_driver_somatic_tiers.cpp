// ref_driver_somatic_tiers.cpp -- C entry points over the REFERENCE's own somatic callers, whole wrappers:
//   somatic_snv_caller_strand_grid::position_somatic_snv_call  (L/applications/strelka/position_somatic_snv_strand_grid.cpp:230-363)
//     with tier-2 pileups, per-site forced output and isComputeNonSomatic
//   somatic_indel_caller_grid::get_somatic_indel               (L/applications/strelka/somatic_indel_grid.cpp:181-361)
//     with the multi-indel-allele filter and the tier combination
//
// TEST INFRASTRUCTURE ONLY; contains no reference code.  Compiled with the unmodified reference sources into
// oracle/_ref/libstrelka_ref.so (oracle/Makefile, target `ref`).

#include "applications/strelka/position_somatic_snv_strand_grid.hh"
#include "applications/strelka/somatic_indel_grid.hh"
#include "applications/strelka/strelka_shared.hh"
#include "blt_common/snp_pos_info.hh"
#include "starling_common/IndelData.hh"

#include <cstdint>
#include <cstring>
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

} // extern "C"
