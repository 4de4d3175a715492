// normalize_core.h -- normalizeAlignment (L/starling_common/normalizeAlignment.cpp:647-703) as container-free code for the device and
// the host alike: indel collapsing (:379-461, collapseInsert :211-262), left-shifting (:463-538, findLeftShift :78-117,
// leftShiftIndel :119-161), edge-indel normalisation (:540-645, :264-377) and apath_cleaner (L/blt_util/align_path.cpp:553-651),
// iterated as the reference iterates them.  The path is edited in place: no step adds a segment, the cleaner only merges and drops.
#pragma once

#include "strelka_amd.h"

#include <stdint.h>

#if defined(__HIP__)
#define SKN_HD __host__ __device__
#else
#define SKN_HD
#endif

namespace sknorm
{

struct Seqs // the two sequences normalizeAlignment compares, with the reference's out-of-range behaviour
{
    const char* ref; // reference_contig_segment: get_base(i) is 'N' outside [ref_offset, ref_offset + ref_len)
    int32_t ref_offset, ref_len;
    const uint8_t* read_code; // BAM 4-bit codes, one per byte (bam_seq::get_char: '=', A, C, G, T, everything else 'N'; 'N' out of range)
    int32_t read_len;
};
SKN_HD inline char ref_char(const Seqs& s, const int32_t p)
{
    return (p < s.ref_offset || p >= s.ref_offset + s.ref_len) ? 'N' : s.ref[p - s.ref_offset];
}
SKN_HD inline char read_char(const Seqs& s, const int32_t i)
{
    if (i < 0 || i >= s.read_len) return 'N';
    switch (s.read_code[i]) {
    case 0: return '=';
    case 1: return 'A';
    case 2: return 'C';
    case 4: return 'G';
    case 8: return 'T';
    default: return 'N';
    }
}
SKN_HD inline bool same(const Seqs& s, const int32_t read_pos, const int32_t ref_pos) { return read_char(s, read_pos) == ref_char(s, ref_pos); }

SKN_HD inline bool seg_match(const uint32_t t) { return t == SK_SEG_MATCH || t == SK_SEG_SEQ_MATCH || t == SK_SEG_SEQ_MISMATCH; }
SKN_HD inline bool seg_read_len(const uint32_t t) { return seg_match(t) || t == SK_SEG_INSERT || t == SK_SEG_SOFT_CLIP; }
SKN_HD inline bool seg_ref_len(const uint32_t t) { return seg_match(t) || t == SK_SEG_DELETE || t == SK_SEG_SKIP; }

struct Info // AlignmentInfo :43-58
{
    int32_t refPos = 0, readPos = 0;
    unsigned startPriorMatchSegment = 0, endPriorMatchSegment = 0, startPriorIndelSegment = 0, endPriorIndelSegment = 0;
    int32_t priorMatchLength = 0, priorDeleteLength = 0, priorInsertLength = 0;
    bool isChanged = false;
};

struct Aln
{
    int32_t pos;
    sk_path_seg* path;
    int n_seg;
};

// apath_cleaner, align_path.cpp:553-651
SKN_HD inline bool apath_cleaner(Aln& al)
{
    bool is_cleaned = false;
    const int as = al.n_seg;
    int insertIndex = as, deleteIndex = as, otherIndex = as;
    for (int i = 0; i < as; ++i) {
        sk_path_seg& ps = al.path[i];
        if (ps.length == 0) {
            is_cleaned = true;
        } else if (ps.type == SK_SEG_PAD) {
            ps.length = 0;
            is_cleaned = true;
        } else if (ps.type == SK_SEG_INSERT) {
            if (insertIndex < as) {
                al.path[insertIndex].length += ps.length;
                ps.length = 0;
                is_cleaned = true;
            } else {
                insertIndex = i;
            }
        } else if (ps.type == SK_SEG_DELETE) {
            if (deleteIndex < as) {
                al.path[deleteIndex].length += ps.length;
                ps.length = 0;
                is_cleaned = true;
            } else {
                deleteIndex = i;
            }
        } else {
            if (insertIndex < as || deleteIndex < as) {
                insertIndex = as;
                deleteIndex = as;
                otherIndex = as;
            }
            if (otherIndex < as && al.path[otherIndex].type == ps.type) {
                al.path[otherIndex].length += ps.length;
                ps.length = 0;
                is_cleaned = true;
            } else {
                otherIndex = i;
            }
        }
    }
    for (int i = 0; i < as; ++i) { // NDN -> N
        sk_path_seg& ps = al.path[i];
        if (ps.type == SK_SEG_SKIP && i + 2 < as && al.path[i + 1].type == SK_SEG_DELETE && al.path[i + 2].type == SK_SEG_SKIP) {
            for (int j = 1; j < 3; ++j) {
                ps.length += al.path[i + j].length;
                al.path[i + j].length = 0;
            }
            is_cleaned = true;
        }
    }
    if (is_cleaned) {
        int n = 0;
        for (int i = 0; i < as; ++i)
            if (al.path[i].length != 0) al.path[n++] = al.path[i];
        al.n_seg = n;
    }
    return is_cleaned;
}

SKN_HD inline int32_t find_left_shift(const Seqs& s, const Info& ai) // :78-117
{
    int32_t shift = 0;
    const int32_t refRightPos = ai.refPos - 1, readRightPos = ai.readPos - 1;
    const int32_t refLeftPos = refRightPos - ai.priorDeleteLength, readLeftPos = readRightPos - ai.priorInsertLength;
    while (shift < ai.priorMatchLength) {
        const bool isLeftMatch = same(s, readLeftPos - shift, refLeftPos - shift);
        const bool isRightMatch = same(s, readRightPos - shift, refRightPos - shift);
        if (!isRightMatch && isLeftMatch) break;
        shift++;
    }
    return shift;
}

SKN_HD inline void left_shift_indel(const Seqs& s, const unsigned currentSegment, Aln& al, Info& ai) // :119-161
{
    const int32_t shiftSize = find_left_shift(s, ai);
    if (shiftSize <= 0) return;
    ai.isChanged = true;
    al.path[currentSegment].type = SK_SEG_MATCH;
    al.path[currentSegment].length += uint32_t(shiftSize);
    for (unsigned m = ai.startPriorMatchSegment; m < ai.endPriorMatchSegment; ++m) {
        al.path[m].type = SK_SEG_MATCH;
        al.path[m].length = (m == ai.startPriorMatchSegment) ? uint32_t(ai.priorMatchLength - shiftSize) : 0u;
    }
    ai.refPos -= shiftSize;
    ai.readPos -= shiftSize;
}

SKN_HD inline void set_indel_lengths(Aln& al, const Info& ai, const bool keep_delete) // the loops at :245-261, :305-321, :362-376
{
    bool isFirstInsert = true, isFirstDelete = true;
    for (unsigned i = ai.startPriorIndelSegment; i < ai.endPriorIndelSegment; ++i) {
        sk_path_seg& ps = al.path[i];
        if (ps.type == SK_SEG_INSERT) {
            ps.length = isFirstInsert ? uint32_t(ai.priorInsertLength) : 0u;
            isFirstInsert = false;
        } else if (ps.type == SK_SEG_DELETE) {
            ps.length = (keep_delete && isFirstDelete) ? uint32_t(ai.priorDeleteLength) : 0u;
            isFirstDelete = false;
        }
    }
}

SKN_HD inline void collapse_insert(const Seqs& s, const unsigned currentSegment, Aln& al, Info& ai) // :163-262
{
    int32_t leftCollapse = 0, rightCollapse = 0;
    const int32_t maxCollapse = ai.priorDeleteLength < ai.priorInsertLength ? ai.priorDeleteLength : ai.priorInsertLength;
    {
        const int32_t refRightPos = ai.refPos - 1, readRightPos = ai.readPos - 1;
        while (rightCollapse < maxCollapse) {
            if (!same(s, readRightPos - rightCollapse, refRightPos - rightCollapse)) break;
            rightCollapse++;
        }
    }
    {
        const int32_t refLeftPos = ai.refPos - ai.priorDeleteLength, readLeftPos = ai.readPos - ai.priorInsertLength;
        while (leftCollapse + rightCollapse < maxCollapse) {
            if (!same(s, readLeftPos + leftCollapse, refLeftPos + leftCollapse)) break;
            leftCollapse++;
        }
    }
    if (leftCollapse + rightCollapse <= 0) return;
    ai.isChanged = true;
    if (leftCollapse > 0) {
        al.path[ai.startPriorMatchSegment].type = SK_SEG_MATCH;
        al.path[ai.startPriorMatchSegment].length += uint32_t(leftCollapse);
        ai.priorMatchLength += leftCollapse;
    }
    if (rightCollapse > 0) {
        al.path[currentSegment].type = SK_SEG_MATCH;
        al.path[currentSegment].length += uint32_t(rightCollapse);
    }
    ai.priorInsertLength -= leftCollapse + rightCollapse;
    ai.priorDeleteLength -= leftCollapse + rightCollapse;
    set_indel_lengths(al, ai, true);
}

SKN_HD inline void left_edge_indel_collapse(const Seqs& s, const unsigned currentSegment, Aln& al, Info& ai) // :264-322
{
    if (ai.priorDeleteLength > 0) {
        ai.isChanged = true;
        al.pos += ai.priorDeleteLength;
    }
    if (ai.priorInsertLength > 0) {
        int32_t rightCollapse = 0;
        const int32_t refRightPos = ai.refPos - 1, readRightPos = ai.readPos - 1;
        while (rightCollapse < ai.priorInsertLength && refRightPos > 0) {
            if (!same(s, readRightPos - rightCollapse, refRightPos - rightCollapse)) break;
            rightCollapse++;
        }
        if (rightCollapse > 0) {
            ai.isChanged = true;
            al.pos -= rightCollapse;
            al.path[currentSegment].type = SK_SEG_MATCH;
            al.path[currentSegment].length += uint32_t(rightCollapse);
            ai.priorInsertLength -= rightCollapse;
        }
    }
    set_indel_lengths(al, ai, false);
}

SKN_HD inline void right_edge_indel_collapse(const Seqs& s, Aln& al, Info& ai) // :324-377
{
    if (ai.priorDeleteLength > 0) ai.isChanged = true;
    if (ai.priorInsertLength > 0) {
        int32_t leftCollapse = 0;
        const int32_t refLeftPos = ai.refPos - ai.priorDeleteLength, readLeftPos = ai.readPos - ai.priorInsertLength;
        while (leftCollapse < ai.priorInsertLength) {
            if (!same(s, readLeftPos + leftCollapse, refLeftPos + leftCollapse)) break;
            leftCollapse++;
        }
        if (leftCollapse > 0) {
            ai.isChanged = true;
            al.path[ai.startPriorMatchSegment].type = SK_SEG_MATCH;
            al.path[ai.startPriorMatchSegment].length += uint32_t(leftCollapse);
            ai.priorInsertLength -= leftCollapse;
        }
    }
    set_indel_lengths(al, ai, false);
}

SKN_HD inline void advance(const sk_path_seg& ps, Info& ai)
{
    if (seg_read_len(ps.type)) ai.readPos += int32_t(ps.length);
    if (seg_ref_len(ps.type)) ai.refPos += int32_t(ps.length);
}

SKN_HD inline bool collapse_alignment_indels(const Seqs& s, Aln& al) // :379-461
{
    Info ai;
    bool isInsideIndel = false;
    ai.refPos = al.pos;
    for (int i = 0; i < al.n_seg; ++i) {
        sk_path_seg& ps = al.path[i];
        if (ps.length == 0) continue;
        if (seg_match(ps.type)) {
            if (ai.priorMatchLength > 0 && isInsideIndel) {
                if (ai.priorDeleteLength > 0 && ai.priorInsertLength > 0) collapse_insert(s, unsigned(i), al, ai);
                ai.priorMatchLength = 0;
            }
            if (ai.priorMatchLength == 0) ai.startPriorMatchSegment = unsigned(i);
            isInsideIndel = false;
            ai.priorMatchLength += int32_t(ps.length);
            ai.endPriorMatchSegment = unsigned(i) + 1;
        } else if (ps.type == SK_SEG_DELETE || ps.type == SK_SEG_INSERT) {
            if (!isInsideIndel) {
                isInsideIndel = true;
                ai.priorDeleteLength = 0;
                ai.priorInsertLength = 0;
                ai.startPriorIndelSegment = unsigned(i);
            }
            ai.endPriorIndelSegment = unsigned(i) + 1;
            if (ps.type == SK_SEG_DELETE) ai.priorDeleteLength += int32_t(ps.length);
            else ai.priorInsertLength += int32_t(ps.length);
        } else {
            isInsideIndel = false;
            ai.priorMatchLength = 0;
        }
        advance(ps, ai);
    }
    return ai.isChanged;
}

SKN_HD inline bool left_shift_alignment_indels(const Seqs& s, Aln& al) // :463-538
{
    Info ai;
    bool isInsideIndel = false;
    ai.refPos = al.pos;
    for (int i = 0; i < al.n_seg; ++i) {
        sk_path_seg& ps = al.path[i];
        if (seg_match(ps.type)) {
            if (ai.priorMatchLength > 0 && isInsideIndel) {
                left_shift_indel(s, unsigned(i), al, ai);
                ai.priorMatchLength = 0;
            }
            if (ai.priorMatchLength == 0) ai.startPriorMatchSegment = unsigned(i);
            isInsideIndel = false;
            ai.priorMatchLength += int32_t(ps.length);
            ai.endPriorMatchSegment = unsigned(i) + 1;
        } else if (ps.type == SK_SEG_DELETE || ps.type == SK_SEG_INSERT) {
            if (!isInsideIndel) {
                isInsideIndel = true;
                ai.priorDeleteLength = 0;
                ai.priorInsertLength = 0;
            }
            if (ps.type == SK_SEG_DELETE) ai.priorDeleteLength += int32_t(ps.length);
            else ai.priorInsertLength += int32_t(ps.length);
        } else {
            isInsideIndel = false;
            ai.priorMatchLength = 0;
        }
        advance(ps, ai);
    }
    return ai.isChanged;
}

SKN_HD inline bool normalize_edge_indels(const Seqs& s, Aln& al) // :540-645
{
    Info ai;
    bool isInsideIndel = false, isFirstMatch = true;
    ai.refPos = al.pos;
    for (int i = 0; i < al.n_seg; ++i) {
        sk_path_seg& ps = al.path[i];
        if (ps.length == 0) continue;
        if (seg_match(ps.type)) {
            if (isInsideIndel && isFirstMatch) left_edge_indel_collapse(s, unsigned(i), al, ai);
            isFirstMatch = false;
            if (ai.priorMatchLength > 0 && isInsideIndel) ai.priorMatchLength = 0;
            if (ai.priorMatchLength == 0) ai.startPriorMatchSegment = unsigned(i);
            isInsideIndel = false;
            ai.priorMatchLength += int32_t(ps.length);
            ai.endPriorMatchSegment = unsigned(i) + 1;
        } else if (ps.type == SK_SEG_DELETE || ps.type == SK_SEG_INSERT) {
            if (!isInsideIndel) {
                isInsideIndel = true;
                ai.priorDeleteLength = 0;
                ai.priorInsertLength = 0;
                ai.startPriorIndelSegment = unsigned(i);
            }
            ai.endPriorIndelSegment = unsigned(i) + 1;
            if (ps.type == SK_SEG_DELETE) ai.priorDeleteLength += int32_t(ps.length);
            else ai.priorInsertLength += int32_t(ps.length);
        } else {
            if ((ps.type == SK_SEG_SOFT_CLIP || ps.type == SK_SEG_HARD_CLIP) && ai.priorMatchLength && isInsideIndel &&
                (ai.priorDeleteLength > 0 || ai.priorInsertLength > 0))
                right_edge_indel_collapse(s, al, ai);
            isInsideIndel = false;
            ai.priorMatchLength = 0;
        }
        advance(ps, ai);
    }
    if (ai.priorMatchLength && isInsideIndel && (ai.priorDeleteLength > 0 || ai.priorInsertLength > 0)) right_edge_indel_collapse(s, al, ai);
    return ai.isChanged;
}

// normalizeAlignment :647-703
SKN_HD inline bool normalize_alignment(const Seqs& s, Aln& al)
{
    bool isAlignmentChanged = false, isCollapseAgain = true, isLeftShiftAgain = true;
    while (isCollapseAgain || isLeftShiftAgain) {
        if (isCollapseAgain) {
            if (collapse_alignment_indels(s, al)) {
                isAlignmentChanged = true;
                isLeftShiftAgain = true;
                isCollapseAgain = apath_cleaner(al);
            } else {
                isCollapseAgain = false;
            }
        }
        if (isLeftShiftAgain) {
            if (left_shift_alignment_indels(s, al)) {
                isAlignmentChanged = true;
                isCollapseAgain = true;
                isLeftShiftAgain = apath_cleaner(al);
            } else {
                isLeftShiftAgain = false;
            }
        }
    }
    if (normalize_edge_indels(s, al)) {
        isAlignmentChanged = true;
        (void)apath_cleaner(al);
    }
    return isAlignmentChanged;
}

} // namespace sknorm
