// ref_driver_aligner.cpp -- C entry point over the REFERENCE's GlobalAligner<int> (L/alignment/GlobalAligner.hh), the
// haplotype-to-reference aligner of the active-region code.  TEST INFRASTRUCTURE ONLY; contains no reference code.
#include "alignment/GlobalAligner.hh"
#include "blt_util/align_path.hh"

#include <cstring>
#include <string>

extern "C" int ref_global_align(const char* query, int query_size, const char* ref, int ref_size, int match, int mismatch,
                                int open, int extend, int off_edge, int insert_delete, int allow_edge_insertion,
                                int require_edge_deletion, int* out_score, int* out_begin_pos, char* out_cigar, int cigar_cap)
{
    try {
        const AlignmentScores<int> scores(match, mismatch, open, extend, off_edge, insert_delete, allow_edge_insertion != 0,
                                          require_edge_deletion != 0);
        const GlobalAligner<int> aligner(scores);
        AlignmentResult<int> result;
        const std::string q(query, query_size), r(ref, ref_size);
        aligner.align(q.begin(), q.end(), r.begin(), r.end(), result);
        *out_score = result.score;
        *out_begin_pos = result.align.beginPos;
        const std::string cigar(ALIGNPATH::apath_to_cigar(result.align.apath));
        std::strncpy(out_cigar, cigar.c_str(), cigar_cap - 1);
        out_cigar[cigar_cap - 1] = 0;
        return 0;
    } catch (...) {
        return 1;
    }
}
