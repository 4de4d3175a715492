// sk_adapter_active_region.cpp -- site 7: ActiveRegionProcessor::discoverIndelsAndMismatches
// (L/starling_common/ActiveRegionProcessor.cpp:572-705): the haplotype-to-reference GlobalAligner<int> call (:591) through
// sk_global_align and the walk / left-shift of the resulting path (:594-705) through sk_discover_indels_and_mismatches.
// The reference calls it once per selected haplotype (at most two per active region); the first call of a region aligns all of the
// region's alternate haplotypes in one sk_global_align batch.
#include "sk_adapter_access.hh"

#include <cstdlib>

#include "blt_util/reference_contig_segment.hh"
#include "starling_common/IndelKey.hh"

#include <cstring>

namespace sk_adapter
{

namespace
{

/// the alignments of one active region's alternate haplotypes against its reference segment, made in one batch
struct RegionAlignments
{
    pos_t regionBegin = 0, regionEnd = 0;
    std::string refSegment;
    std::vector<std::string> haplotypes;           ///< as selected (the reference segment itself may be among them)
    std::vector<int32_t> score, beginPos, segCount; ///< per haplotype; segCount < 0 = not aligned (equal to the reference segment)
    std::vector<std::vector<sk_path_seg>> path;
    bool matches(const std::vector<std::string>& selected, const std::string& ref, const pos_t begin, const pos_t end) const
    {
        return begin == regionBegin && end == regionEnd && ref == refSegment && selected == haplotypes;
    }
};

void alignRegion(const std::vector<std::string>& selected, const std::string& refSegment, const pos_t regionBegin, const pos_t regionEnd,
                 RegionAlignments& ra)
{
    ra.regionBegin = regionBegin;
    ra.regionEnd = regionEnd;
    ra.refSegment = refSegment;
    ra.haplotypes = selected;
    const size_t count(selected.size());
    ra.score.assign(count, 0);
    ra.beginPos.assign(count, 0);
    ra.segCount.assign(count, -1);
    ra.path.assign(count, std::vector<sk_path_seg>());
    std::vector<size_t> which;
    std::vector<int64_t> queryOff(1, 0), refOff(1, 0);
    std::string query, ref;
    for (size_t i(0); i < count; ++i)
    {
        const std::string& haplotypeSeq(selected[i]);
        if (haplotypeSeq == refSegment) continue; // (processSelectedHaplotypes skips it, :530)
        if (haplotypeSeq.empty() || refSegment.empty() || haplotypeSeq.size() > 1024 || refSegment.size() > 1024)
        {
            throw blt_exception("strelka_amd adapter: haplotype / reference segment length outside 1..1024");
        }
        which.push_back(i);
        query += haplotypeSeq;
        ref += refSegment;
        queryOff.push_back(static_cast<int64_t>(query.size()));
        refOff.push_back(static_cast<int64_t>(ref.size()));
    }
    if (which.empty()) return;
    sk_align_scores scores;
    sk_align_scores_default(&scores); // = ActiveRegionDetector's AlignmentScores<int>(1,-4,-5,-1,-100) (ActiveRegionDetector.hh:59-63)
    sk_global_align_batch gb;
    gb.n = static_cast<int32_t>(which.size());
    gb.query_off = queryOff.data();
    gb.query = query.data();
    gb.ref_off = refOff.data();
    gb.ref = ref.data();
    std::vector<int32_t> score(which.size()), beginPos(which.size()), segCount(which.size());
    // pair p's path lies at query_off[p] + ref_off[p] + 4 p (include/strelka_amd.h, sk_global_align)
    std::vector<sk_path_seg> path(query.size() + ref.size() + 4 * which.size());
    {
        AccumTimer abiTimer(state().tHaplotypeAbi);
        check(sk_global_align(&gb, &scores, score.data(), beginPos.data(), path.data(), segCount.data()), "sk_global_align");
    }
    for (size_t p(0); p < which.size(); ++p)
    {
        const size_t i(which[p]);
        ra.score[i] = score[p];
        ra.beginPos[i] = beginPos[p];
        ra.segCount[i] = segCount[p];
        const sk_path_seg* const first(path.data() + queryOff[p] + refOff[p] + 4 * static_cast<int64_t>(p));
        ra.path[i].assign(first, first + segCount[p]);
    }
    state().haplotypeBatches++;
}

}

bool discover_indels_and_mismatches(const std::vector<std::string>& selectedHaplotypes, const unsigned selectedHaplotypeIndex,
                                    const std::string& refSegment, const reference_contig_segment& ref, const pos_t regionBegin,
                                    const pos_t regionEnd, const pos_t prevActiveRegionEnd, const unsigned maxIndelSize,
                                    std::vector<IndelKey>& discovered, int& numIndels)
{
    init();
    static RegionAlignments ra;
    // $STRELKA_AMD_HAPLOTYPE_BATCH=0: one sk_global_align problem per call, as the reference calls its aligner (diagnosis)
    static const bool isBatched([]() { const char* v(std::getenv("STRELKA_AMD_HAPLOTYPE_BATCH")); return ! (v && *v == '0'); }());
    const std::string& haplotypeSeq(selectedHaplotypes[selectedHaplotypeIndex]);
    unsigned at(selectedHaplotypeIndex);
    if (isBatched)
    {
        if (! ra.matches(selectedHaplotypes, refSegment, regionBegin, regionEnd)) alignRegion(selectedHaplotypes, refSegment, regionBegin, regionEnd, ra);
    }
    else
    {
        alignRegion(std::vector<std::string>(1, haplotypeSeq), refSegment, regionBegin, regionEnd, ra);
        at = 0;
    }
    if (ra.segCount[at] < 0) throw blt_exception("strelka_amd adapter: haplotype equal to the reference segment");
    const int32_t beginPos(ra.beginPos[at]), segCount(ra.segCount[at]);
    const std::vector<sk_path_seg>& path(ra.path[at]);

    const int32_t cap(static_cast<int32_t>(haplotypeSeq.size() + refSegment.size() + 4));
    std::vector<sk_discovered_allele> found(static_cast<size_t>(cap));
    std::vector<char> insSeq(haplotypeSeq.size() * 2 + 16);
    int32_t n(0), indelCount(0);
    check(sk_discover_indels_and_mismatches(ref.seq().data(), static_cast<int32_t>(ref.get_offset()),
                                            static_cast<int32_t>(ref.seq().size()), regionBegin, regionEnd, prevActiveRegionEnd,
                                            maxIndelSize, haplotypeSeq.data(), static_cast<int32_t>(haplotypeSeq.size()), beginPos,
                                            path.data(), segCount, found.data(), cap, insSeq.data(),
                                            static_cast<int32_t>(insSeq.size()), &n, &indelCount),
          "sk_discover_indels_and_mismatches");
    for (int32_t k(0); k < n; ++k)
    {
        const sk_discovered_allele& a(found[static_cast<size_t>(k)]);
        const std::string ins(insSeq.data() + a.ins_off, a.ins_len);
        discovered.push_back(IndelKey(a.pos, static_cast<INDEL::index_t>(a.type), a.del_len, ins.c_str()));
    }
    numIndels = indelCount;
    state().haplotypes++;
    return true;
}

}
