// gvcf_block_core.h -- the non-variant block logic of the gVCF writer (the output side) for the device and
// the host:
//
//   gvcf_block_site_record::testCanSiteJoinSampleBlock / testCanSiteJoinSampleBlockShared / joinSiteToSampleBlock
//                              L/applications/starling/gvcf_block_site_record.cpp:30-184
//   stream_stat::add           L/blt_util/stream_stat.hh:56-66 (min, max and the running mean M; Q is not read by the writer)
//   the per-sample loop        gvcf_writer::queue_site_record, gvcf_writer.cpp:278-302
//
// A site that is not compressible (gvcf_compressor::is_site_compressible), a gap in positions and a flush the caller asks for (an
// indel record, the end of a region: writeAllNonVariantBlockRecords) end the open block whatever it holds; between two such places
// the greedy joining depends on the block's running minimum / maximum, so a stretch is walked in order -- and different stretches are
// independent of each other: one lane each (gvcf_block_kernel), or one loop over all of them on a host.
#pragma once

#include "strelka_amd.h"

#ifdef __HIPCC__
#define SKG_HD __host__ __device__
#else
#define SKG_HD
#endif

namespace skgvcf
{

struct Stat // stream_stat without Q
{
    double M, max, min;
    unsigned k;
    SKG_HD void reset()
    {
        M = max = min = 0;
        k = 0;
    }
    SKG_HD void add(const double x)
    {
        k++;
        if (k == 1 || x > max) max = x;
        if (k == 1 || x < min) min = x;
        const double delta = x - M;
        M += delta / static_cast<double>(k);
    }
};

SKG_HD inline double compat_round(const double x) { return x >= 0. ? __builtin_floor(x + 0.5) : __builtin_ceil(x - 0.5); } // compat_util.cpp:32-42

SKG_HD inline bool single_tolerance(const Stat& ss, const int min, const int tol) { return (min + tol) >= ss.max / 2.0; } // :30-37
SKG_HD inline bool block_tolerance(const Stat& ss, const double frac_tol, const int abs_tol)                              // :41-55
{
    const int min = static_cast<int>(compat_round(ss.min));
    if (single_tolerance(ss, min, abs_tol)) return true;
    const int ftol = static_cast<int>(__builtin_floor(min * frac_tol));
    if (ftol <= abs_tol) return false;
    return single_tolerance(ss, min, ftol);
}
SKG_HD inline bool new_value_blockable(const int new_val, const Stat& ss, const double frac_tol, const int abs_tol, const bool is_new_val = true,
                                       const bool is_old_val = true) // :59-73
{
    if (!(is_new_val && is_old_val)) return is_new_val == is_old_val;
    Stat ss2 = ss;
    ss2.add(new_val);
    return block_tolerance(ss2, frac_tol, abs_tol);
}

SKG_HD inline bool gt_is_variant(const uint32_t gt) { return (gt & 0xffffu) != 0; } // VcfGenotype::isVariant: an allele index is not 0

// is site i the first of a stretch (see the header)?
SKG_HD inline bool starts_stretch(const sk_gvcf_site* s, const int32_t i)
{
    if (i == 0 || s[i].flush_before) return true;
    if (!s[i].is_compressible || !s[i - 1].is_compressible) return true;
    return s[i].pos != s[i - 1].pos + 1;
}

// the stretch that starts at site `first`: kind[] for its sites, blocks[] at the first site of every block
SKG_HD inline void walk_stretch(const sk_gvcf_site* s, const int32_t n, const int32_t first, const double frac_tol, const int abs_tol, uint8_t* kind,
                                sk_gvcf_block* blocks)
{
    if (!s[first].is_compressible) { // written as a record of its own (write_site_record)
        kind[first] = 2;
        return;
    }
    Stat gqx, dpu, dpf;
    int32_t start = -1, count = 0; // the open block: its first site, and the members set from that site (:119-147)
    auto flush = [&]() {
        if (count <= 0) return;
        sk_gvcf_block& b = blocks[start];
        b.pos = s[start].pos;
        b.count = count;
        b.is_gqx_defined = s[start].is_gqx ? 1 : 0;
        b.gqx_min = s[start].is_gqx ? static_cast<int32_t>(gqx.min) : 0;
        b.dpu_min = static_cast<int32_t>(dpu.min);
        b.dpu_mean = dpu.M;
        b.dpf_mean = dpf.M;
        count = 0;
    };
    for (int32_t i = first; i < n && (i == first || !starts_stretch(s, i)); ++i) {
        const sk_gvcf_site& in = s[i];
        bool can_join = true;
        if (count > 0) {
            const sk_gvcf_site& b0 = s[start];
            // testCanSiteJoinSampleBlockShared :77-122 ((pos + count) == locus.pos holds inside a stretch)
            can_join = in.locus_filters == b0.locus_filters && in.sample_filters == b0.sample_filters && !gt_is_variant(b0.gt) && !gt_is_variant(in.gt) &&
                       new_value_blockable(static_cast<int>(in.used_basecalls), dpu, frac_tol, abs_tol) &&
                       new_value_blockable(static_cast<int>(in.unused_basecalls), dpf, frac_tol, abs_tol) &&
                       ((b0.used_basecalls != 0 || b0.unused_basecalls != 0) == (in.used_basecalls != 0 || in.unused_basecalls != 0)) &&
                       ((b0.used_basecalls != 0) == (in.used_basecalls != 0)) && in.gt == b0.gt && in.ploidy == b0.ploidy &&
                       // :163-182
                       new_value_blockable(in.gqx, gqx, frac_tol, abs_tol, in.is_gqx != 0, b0.is_gqx != 0);
        }
        if (!can_join) flush();
        if (count == 0) {
            start = i;
            gqx.reset();
            dpu.reset();
            dpf.reset();
            kind[i] = 1;
        } else {
            kind[i] = 0;
        }
        dpu.add(in.used_basecalls); // joinSiteToSampleBlock :126-157
        dpf.add(in.unused_basecalls);
        if (in.is_gqx) gqx.add(in.gqx);
        ++count;
    }
    flush();
}

} // namespace skgvcf
