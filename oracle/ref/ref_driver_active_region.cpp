// ref_driver_active_region.cpp -- C entry point over the REFERENCE's ActiveRegionProcessor::discoverIndelsAndMismatches
// (L/starling_common/ActiveRegionProcessor.cpp:572-697): haplotype -> GlobalAligner -> left-shifted primitive alleles.
// TEST INFRASTRUCTURE ONLY; contains no reference code.  The method is private: the class is included with its private
// members opened so that the driver can hand it a haplotype directly instead of synthesising reads that assemble to it.
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#define private public
#include "starling_common/ActiveRegionProcessor.hh"
#undef private

#include "starling_common/ActiveRegionDetector.hh"
#include "starling_common/CandidateSnvBuffer.hh"
#include "starling_common/IndelBuffer.hh"
#include "test/starling_base_options_test.hh"

#include <cstring>

extern "C" int ref_discover_indels_and_mismatches(const char* ref_seq, int ref_offset, int ar_begin, int ar_end, int prev_ar_end,
                                                  unsigned max_indel_size, const char* haplotype, char* out, int out_cap,
                                                  int* num_indels)
{
    try {
        starling_base_options_test opt;
        opt.is_candidate_indel_signal_test = false;
        starling_base_deriv_options dopt(opt);
        reference_contig_segment ref;
        ref.seq() = ref_seq;
        ref.set_offset(ref_offset);
        IndelBuffer indelBuffer(opt, dopt, ref);
        depth_buffer db, db2;
        indelBuffer.registerSample(db, db2, false);
        indelBuffer.finalizeSamples();
        CandidateSnvBuffer snvBuffer(1);
        ActiveRegionReadBuffer readBuffer(ref, 0.2f, indelBuffer);
        const GlobalAligner<int> aligner(AlignmentScores<int>(ActiveRegionDetector::ScoreMatch, ActiveRegionDetector::ScoreMismatch,
                                                              ActiveRegionDetector::ScoreOpen, ActiveRegionDetector::ScoreExtend,
                                                              ActiveRegionDetector::ScoreOffEdge, ActiveRegionDetector::ScoreOpen,
                                                              true, true));
        ActiveRegionProcessor arp(known_pos_range2(ar_begin, ar_end), prev_ar_end, ref, max_indel_size, 0, 2, aligner, readBuffer,
                                  indelBuffer, snvBuffer);
        arp._selectedHaplotypes.push_back(haplotype);
        std::vector<IndelKey> found;
        int n = 0;
        arp.discoverIndelsAndMismatches(0, found, n);
        *num_indels = n;
        std::ostringstream os;
        for (const IndelKey& k : found) os << k.pos << "," << int(k.type) << "," << k.deletionLength << "," << k.insertSequence << ";";
        const std::string s(os.str());
        if (int(s.size()) + 1 > out_cap) return 2;
        std::strcpy(out, s.c_str());
        return 0;
    } catch (...) {
        return 1;
    }
}
