// ref_driver_feed.cpp -- C entry point over the REFERENCE's own normalizeAlignment (L/starling_common/normalizeAlignment.cpp:647-703).
//
// TEST INFRASTRUCTURE ONLY; contains no reference code: it builds the reference's own argument types (reference_contig_segment behind
// rc_segment_bam_seq, a string_bam_seq for the read, an `alignment`) and calls the function.

#include "starling_common/normalizeAlignment.hh"

#include "blt_util/align_path.hh"
#include "blt_util/reference_contig_segment.hh"
#include "htsapi/bam_seq.hh"

#include <cstdint>
#include <string>

extern "C" {

/// path: (type, length) pairs in/out (capacity cap_seg); returns the function's result (1 = changed), -1 when the result does not fit
int ref_normalize_alignment(const char* ref_seq, int32_t ref_offset, int32_t ref_len, const char* read_seq, int32_t read_len, int32_t* pos,
                            uint32_t* path, int32_t* n_seg, int32_t cap_seg)
{
    reference_contig_segment ref;
    ref.seq() = std::string(ref_seq, ref_seq + ref_len);
    ref.set_offset(ref_offset);
    const rc_segment_bam_seq refBamSeq(ref);
    const std::string read(read_seq, read_seq + read_len);
    const string_bam_seq readBamSeq(read);
    alignment al;
    al.pos = *pos;
    for (int32_t i = 0; i < *n_seg; ++i)
        al.path.push_back(ALIGNPATH::path_segment(static_cast<ALIGNPATH::align_t>(path[2 * i]), path[2 * i + 1]));
    const bool changed(normalizeAlignment(refBamSeq, readBamSeq, al));
    if (static_cast<int32_t>(al.path.size()) > cap_seg) return -1;
    *pos = al.pos;
    *n_seg = static_cast<int32_t>(al.path.size());
    for (size_t i = 0; i < al.path.size(); ++i) {
        path[2 * i] = static_cast<uint32_t>(al.path[i].type);
        path[2 * i + 1] = al.path[i].length;
    }
    return changed ? 1 : 0;
}

} // extern "C"
