// sk_adapter_active_region.cpp -- site 7: ActiveRegionProcessor::discoverIndelsAndMismatches
// (L/starling_common/ActiveRegionProcessor.cpp:572-705): the haplotype-to-reference GlobalAligner<int> call (:591) through
// sk_global_align and the walk / left-shift of the resulting path (:594-705) through sk_discover_indels_and_mismatches.
// One selected haplotype per call, as the reference calls it (at most two per active region).
#include "sk_adapter_access.hh"

#include "blt_util/reference_contig_segment.hh"
#include "starling_common/IndelKey.hh"

#include <cstring>

namespace sk_adapter
{

bool discover_indels_and_mismatches(const std::string& haplotypeSeq, const std::string& refSegment,
                                    const reference_contig_segment& ref, const pos_t regionBegin, const pos_t regionEnd,
                                    const pos_t prevActiveRegionEnd, const unsigned maxIndelSize,
                                    std::vector<IndelKey>& discovered, int& numIndels)
{
    init();
    if (haplotypeSeq.empty() || refSegment.empty() || haplotypeSeq.size() > 1024 || refSegment.size() > 1024)
    {
        throw blt_exception("strelka_amd adapter: haplotype / reference segment length outside 1..1024");
    }
    sk_align_scores scores;
    sk_align_scores_default(&scores); // = ActiveRegionDetector's AlignmentScores<int>(1,-4,-5,-1,-100) (ActiveRegionDetector.hh:59-63)
    const int64_t queryOff[2] = {0, static_cast<int64_t>(haplotypeSeq.size())};
    const int64_t refOff[2] = {0, static_cast<int64_t>(refSegment.size())};
    sk_global_align_batch gb;
    gb.n = 1;
    gb.query_off = queryOff;
    gb.query = haplotypeSeq.data();
    gb.ref_off = refOff;
    gb.ref = refSegment.data();
    int32_t score(0), beginPos(0), segCount(0);
    std::vector<sk_path_seg> path(haplotypeSeq.size() + refSegment.size() + 4);
    {
        AccumTimer abiTimer(state().tHaplotypeAbi);
        check(sk_global_align(&gb, &scores, &score, &beginPos, path.data(), &segCount), "sk_global_align");
    }

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
