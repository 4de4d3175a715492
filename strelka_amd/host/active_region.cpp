// active_region.cpp -- host stage after the GlobalAligner kernel: decompose an aligned haplotype into left-shifted
// primitive alleles.  Mirrors ActiveRegionProcessor::discoverIndelsAndMismatches
// (L/starling_common/ActiveRegionProcessor.cpp:572-697); integer/byte work, results identical to the reference's.
#include "strelka_amd.h"

#include <string>

int sk_fail(const std::string& msg);

namespace
{

struct RefSeg // reference_contig_segment::get_base: 'N' outside the segment (L/blt_util/reference_contig_segment.hh)
{
    const char* seq;
    int32_t off, len;
    char base(const int32_t pos) const { return (pos < off || pos >= off + len) ? 'N' : seq[pos - off]; }
};

} // namespace

extern "C" int sk_discover_indels_and_mismatches(const char* ref_seq, int32_t ref_offset, int32_t ref_len, int32_t ar_begin,
                                                 int32_t ar_end, int32_t prev_ar_end, uint32_t max_indel_size,
                                                 const char* haplotype, int32_t hap_len, int32_t align_begin_pos,
                                                 const sk_path_seg* path, int32_t n_seg, sk_discovered_allele* out,
                                                 int32_t out_cap, char* out_ins_seq, int32_t ins_cap, int32_t* n_out,
                                                 int32_t* n_indels)
{
    if (!ref_seq || !haplotype || (!path && n_seg > 0) || !out || !out_ins_seq || !n_out || !n_indels)
        return sk_fail("sk_discover_indels_and_mismatches: null argument");
    if (align_begin_pos > 0) return sk_fail("sk_discover_indels_and_mismatches: unexpected alignment segment"); // :597-600
    const RefSeg ref = { ref_seq, ref_offset, ref_len };
    int32_t reference_pos = ar_begin, hap_off = 0, n = 0, ni = 0, ins_used = 0;
    auto emit = [&](const int32_t pos, const int32_t type, const uint32_t del_len, const std::string& ins) -> int {
        if (n >= out_cap || ins_used + int32_t(ins.size()) > ins_cap)
            return sk_fail("sk_discover_indels_and_mismatches: output capacity exceeded");
        out[n].pos = pos;
        out[n].type = type;
        out[n].del_len = del_len;
        out[n].ins_len = uint32_t(ins.size());
        out[n].ins_off = ins_used;
        ins.copy(out_ins_seq + ins_used, ins.size());
        ins_used += int32_t(ins.size());
        ++n;
        return 0;
    };
    for (int32_t i = 0; i < n_seg; ++i) {
        const int32_t len = int32_t(path[i].length);
        switch (path[i].type) {
        case SK_SEG_SEQ_MATCH:
            reference_pos += len;
            hap_off += len;
            break;
        case SK_SEG_SEQ_MISMATCH: // one MISMATCH key per base; two regions may share a base (:617-633)
            if (hap_off + len > hap_len) return sk_fail("sk_discover_indels_and_mismatches: path longer than the haplotype");
            for (int32_t k = 0; k < len; ++k) {
                if (reference_pos >= prev_ar_end && emit(reference_pos, SK_INDEL_MISMATCH, 1, std::string(1, haplotype[hap_off])))
                    return 1;
                ++reference_pos;
                ++hap_off;
            }
            break;
        case SK_SEG_INSERT:
            if (hap_off + len > hap_len) return sk_fail("sk_discover_indels_and_mismatches: path longer than the haplotype");
            if (uint32_t(len) <= max_indel_size) { // left-align: move while the last inserted base equals the base before (:640-664)
                int32_t insert_pos = reference_pos;
                std::string ins(haplotype + hap_off, size_t(len));
                char prev = ref.base(insert_pos - 1);
                while (ins.back() == prev) {
                    ins.insert(ins.begin(), prev);
                    ins.pop_back();
                    --insert_pos;
                    prev = ref.base(insert_pos - 1);
                    if (insert_pos < ref_offset - len) // all-N insert left of an N reference: the reference would not stop
                        return sk_fail("sk_discover_indels_and_mismatches: unbounded left shift");
                }
                if (prev != 'N' && insert_pos >= prev_ar_end && insert_pos < ar_end) {
                    if (emit(insert_pos, SK_INDEL_INDEL, 0, ins)) return 1;
                    ++ni;
                }
            }
            hap_off += len;
            break;
        case SK_SEG_DELETE:
            if (uint32_t(len) <= max_indel_size) { // (:668-695)
                int32_t delete_pos = reference_pos;
                char prev = ref.base(delete_pos - 1);
                char last_del = ref.base(delete_pos + len - 1);
                while (last_del == prev) {
                    --delete_pos;
                    last_del = ref.base(delete_pos + len - 1);
                    prev = ref.base(delete_pos - 1);
                    if (delete_pos < ref_offset - len) return sk_fail("sk_discover_indels_and_mismatches: unbounded left shift");
                }
                if (prev != 'N' && delete_pos >= prev_ar_end && delete_pos < ar_end) {
                    if (emit(delete_pos, SK_INDEL_INDEL, uint32_t(len), std::string())) return 1;
                    ++ni;
                }
            }
            reference_pos += len;
            break;
        case SK_SEG_SOFT_CLIP:
            reference_pos += len; // as the reference does (:697-701)
            break;
        default:
            return sk_fail("sk_discover_indels_and_mismatches: unexpected alignment segment");
        }
    }
    *n_out = n;
    *n_indels = ni;
    return 0;
}
