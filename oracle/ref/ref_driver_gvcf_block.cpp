// ref_driver_gvcf_block.cpp -- C entry point over the REFERENCE's own non-variant block logic of the gVCF writer:
// gvcf_block_site_record::testCanSiteJoinSampleBlock / joinSiteToSampleBlock (L/applications/starling/gvcf_block_site_record.cpp),
// driven per sample as gvcf_writer::queue_site_record (gvcf_writer.cpp:278-302) drives them.
//
// TEST INFRASTRUCTURE ONLY; contains no reference code: it fills the reference's own locus objects from plain arrays, calls the two
// member functions and reads the block's public members where write_site_record (gvcf_writer.cpp:749-806) reads them.

#include "applications/starling/gvcf_block_site_record.hh"
#include "applications/starling/gvcf_locus_info.hh"
#include "applications/starling/gvcf_options.hh"

#include <cstdint>

extern "C" {

struct RefGvcfSite
{
    int32_t pos;
    uint8_t is_compressible, is_ref_unknown, gt_ploidy, gt_phased;
    uint8_t gt_allele0, gt_allele1, ploidy, flush_before;
    uint32_t locus_filters, sample_filters;
    int32_t gqx;
    uint32_t used_basecalls, unused_basecalls;
};
struct RefGvcfBlock
{
    int32_t first_site, pos, count, is_gqx_defined;
    double gqx_min, dpu_mean, dpf_mean, dpu_min;
};

/// kind[i]: 0 = site i continues the block of the site before it, 1 = it starts a block, 2 = it is written as a record of its own.
/// Returns the number of blocks written to `blocks` (capacity n).
int ref_gvcf_block_sites(const RefGvcfSite* sites, int32_t n, uint32_t block_percent_tol, uint32_t block_abs_tol, uint8_t* kind,
                         RefGvcfBlock* blocks)
{
    gvcf_options opt;
    opt.block_percent_tol = block_percent_tol;
    opt.block_abs_tol = block_abs_tol;
    gvcf_block_site_record block(opt);
    int n_blocks = 0;
    int first_site = -1;
    auto flush = [&]() {
        if (block.count <= 0) return;
        RefGvcfBlock& b = blocks[n_blocks++];
        b.first_site = first_site;
        b.pos = block.pos;
        b.count = block.count;
        b.is_gqx_defined = block.isBlockGqxDefined ? 1 : 0;
        b.gqx_min = block.isBlockGqxDefined ? block.block_gqx.min() : 0.;
        b.dpu_mean = block.block_dpu.mean();
        b.dpf_mean = block.block_dpf.mean();
        b.dpu_min = block.block_dpu.min();
        block.reset();
    };
    for (int32_t i = 0; i < n; ++i) {
        const RefGvcfSite& s = sites[i];
        if (s.flush_before) flush();
        GermlineSiteLocusInfo locus(1, s.pos, s.is_ref_unknown ? BASE_ID::ANY : BASE_ID::A);
        for (unsigned f = 0; f < GERMLINE_VARIANT_VCF_FILTERS::SIZE; ++f) {
            if ((s.locus_filters >> f) & 1u) locus.filters.set(static_cast<GERMLINE_VARIANT_VCF_FILTERS::index_t>(f));
            if ((s.sample_filters >> f) & 1u) locus.getSample(0).filters.set(static_cast<GERMLINE_VARIANT_VCF_FILTERS::index_t>(f));
        }
        LocusSampleInfo& sample = locus.getSample(0);
        if (s.gt_ploidy == 1) sample.max_gt().setGenotypeFromAlleleIndices(s.gt_allele0);
        else if (s.gt_ploidy == 2) sample.max_gt().setGenotypeFromAlleleIndices(s.gt_allele0, s.gt_allele1, s.gt_phased != 0);
        else sample.max_gt().setGenotypeFromAlleleIndices();
        sample.setPloidy(s.ploidy);
        sample.gqx = s.gqx;
        GermlineSiteSampleInfo siteSample;
        siteSample.usedBasecallCount = s.used_basecalls;
        siteSample.unusedBasecallCount = s.unused_basecalls;
        locus.setSiteSampleInfo(0, siteSample);
        if (!s.is_compressible) { // queue_site_record: writeAllNonVariantBlockRecords + write_site_record
            flush();
            kind[i] = 2;
            continue;
        }
        if (!block.testCanSiteJoinSampleBlock(locus, 0)) flush();
        kind[i] = (block.count == 0) ? 1 : 0;
        if (block.count == 0) first_site = i;
        block.joinSiteToSampleBlock(locus, 0);
    }
    flush();
    return n_blocks;
}

} // extern "C"
