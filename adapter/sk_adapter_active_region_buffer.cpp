// sk_adapter_active_region_buffer.cpp -- the active-region detector's per-basecall bookkeeping for one aligned segment of a read
// (addAlignmentIndelsToPosProcessor, L/starling_common/starling_pos_processor_indel_util.cpp:463-483) in one call.
//
// The reference walks every match segment of every input read base by base and calls ActiveRegionReadBuffer::insertMatch or
// insertMismatch (ActiveRegionReadBuffer.cpp:26-50) -- an out-of-line call that takes three ring-buffer slots by `%` and two vector
// rows by double indirection for each base.  This is host work the device path does not touch (the detector's ring buffers feed
// the reference's own haplotype assembly), but once the pileup and the likelihoods are off the host it is the largest single item
// of what remains (profiles/r03_v15_host_remainder.txt), so the segment is entered here with the same stores in the same order,
// the row pointers and the ring index taken once.  A maintainer would add this as a member function
// (ActiveRegionReadBuffer::insertAlignedSegment); the adapter cannot edit the class (its header is included by unhooked headers of
// the reference through relative paths), so this one translation unit sees the class with its private section opened.
#include <algorithm>
#include <cassert>
#include <iosfwd>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include <cstdint>
#include <cstring>
#include <ostream>
#include <sstream>

#include "blt_util/blt_types.hh"
#include "blt_util/PolymorphicObject.hh"
#include "blt_util/known_pos_range2.hh"
#include "blt_util/reference_contig_segment.hh"
#include "blt_util/seq_util.hh"
#define private public
#include "htsapi/bam_seq.hh"
#undef private
#include "starling_common/IndelBuffer.hh"
#include "starling_common/ReferenceRepeatFinder.hh"
#include "starling_common/indel.hh"
#include "starling_common/starling_types.hh"

#define private public
#include "starling_common/ActiveRegionReadBuffer.hh"
#undef private

#include "sk_adapter.hh"

#include "blt_util/blt_exception.hh"
#include "blt_util/pos_range.hh"
#include "starling_common/alignment.hh"

#include <sstream>

namespace sk_adapter
{

namespace
{

/// bam_seq::get_char for positions [from, from + n) of a packed read (bam_seq.hh:193-206: high nibble first, 'N' past the end), two
/// bases per table look-up instead of a virtual size() and a switch per base
void unpackReadChars(const bam_seq& seq, const unsigned from, const unsigned n, char* dst)
{
    static char pairChar[256][2];
    static bool isTable(false);
    if (! isTable)
    {
        for (unsigned b(0); b < 256; ++b)
        {
            pairChar[b][0] = get_bam_seq_char(static_cast<uint8_t>(b >> 4));
            pairChar[b][1] = get_bam_seq_char(static_cast<uint8_t>(b & 15u));
        }
        isTable = true;
    }
    const unsigned size(seq._size);
    const unsigned live((from < size) ? std::min(n, size - from) : 0u);
    const uint8_t* packed(seq._s);
    unsigned i(static_cast<unsigned>(seq._offset) + from), k(0);
    if ((i & 1u) && k < live)
    {
        dst[k++] = pairChar[packed[i >> 1]][1];
        ++i;
    }
    for (; k + 2 <= live; k += 2, i += 2) std::memcpy(dst + k, pairChar[packed[i >> 1]], 2);
    if (k < live) dst[k++] = pairChar[packed[i >> 1]][0];
    for (; k < n; ++k) dst[k] = 'N';
}

/// reference_contig_segment::get_base for [pos, pos + n): the segment's own characters where the range lies inside it
const char* referenceChars(const reference_contig_segment& ref, const pos_t pos, const unsigned n, char* buffer)
{
    if (pos >= ref.get_offset() && pos + static_cast<pos_t>(n) <= ref.end()) return ref.seq().data() + (pos - ref.get_offset());
    for (unsigned j(0); j < n; ++j) buffer[j] = ref.get_base(pos + static_cast<pos_t>(j));
    return buffer;
}

}

// get_valid_alignment_range (L/starling_common/starling_read_util.cpp:218-329), called for every input read at
// starling_pos_processor_indel_util.cpp:335: the same scores, sums and tie rules, with the two per-read vectors kept between calls
// and the base look-ups made without virtual dispatch (the read is a bam_seq, the reference the contig segment itself).
void valid_alignment_range(const alignment& al, const reference_contig_segment& ref, const bam_seq_base& readSeq, pos_range& validRange)
{
    static const int matchScore(2), mismatchScore(-5), minSegmentScore(-11);
    const bam_seq* packed(dynamic_cast<const bam_seq*>(&readSeq));
    const unsigned readSize(readSeq.size());
    static std::vector<int> fwdScore, revScore;
    static std::vector<char> readChars, refChars;
    using namespace ALIGNPATH;
    if (packed != nullptr)
    {
        // Each mismatch, insertion or deletion takes 5 from the running sums below and nothing else lowers them, so with at most two
        // such events no prefix (or suffix) sum reaches minSegmentScore = -11 and the whole read is valid -- most reads.  Counting the
        // events needs neither the score arrays nor the scan.
        unsigned eventCount(0);
        bool isCountable(true);
        pos_t refPos(al.pos);
        unsigned readPos(0);
        for (const path_segment& ps : al.path)
        {
            if (ps.type == INSERT || ps.type == SOFT_CLIP)
            {
                if (ps.type == INSERT) ++eventCount;
                readPos += ps.length;
            }
            else if (ps.type == DELETE)
            {
                ++eventCount;
                refPos += ps.length;
            }
            else if (is_segment_align_match(ps.type))
            {
                if (readPos + ps.length > readSize)
                {
                    isCountable = false;
                    break;
                }
                readChars.resize(ps.length);
                refChars.resize(ps.length);
                unpackReadChars(*packed, readPos, ps.length, readChars.data());
                const char* const refChar(referenceChars(ref, refPos, ps.length, refChars.data()));
                const char* const readChar(readChars.data());
                unsigned mismatchCount(0);
                for (unsigned j(0); j < ps.length; ++j)
                {
                    mismatchCount += static_cast<unsigned>(readChar[j] != refChar[j]) & static_cast<unsigned>(readChar[j] != 'N') &
                                     static_cast<unsigned>(refChar[j] != 'N');
                }
                eventCount += mismatchCount;
                readPos += ps.length;
                refPos += ps.length;
            }
            else if (ps.type != HARD_CLIP)
            {
                isCountable = false; // (the full pass below reports it)
                break;
            }
            if (eventCount > 2) break;
        }
        if (isCountable && eventCount <= 2)
        {
            validRange.set_begin_pos(0);
            validRange.set_end_pos(readSize);
            if (validRange.end_pos <= validRange.begin_pos)
            {
                validRange.begin_pos = 0;
                validRange.end_pos = 0;
            }
            return;
        }
    }
    fwdScore.assign(readSize, 0);
    revScore.assign(readSize, 0);
    pos_t refHeadPos(al.pos);
    unsigned readHeadPos(0);
    for (const path_segment& ps : al.path)
    {
        if ((ps.type == INSERT) || (ps.type == SOFT_CLIP))
        {
            if (ps.type == INSERT)
            {
                fwdScore[readHeadPos] += mismatchScore;
                revScore[readHeadPos + ps.length - 1] += mismatchScore;
            }
            readHeadPos += ps.length;
        }
        else if (ps.type == DELETE)
        {
            if (readHeadPos > 0) fwdScore[readHeadPos - 1] += mismatchScore;       // (leading deletions do not count)
            if (readHeadPos < readSize) revScore[readHeadPos] += mismatchScore;    // (nor trailing ones)
            refHeadPos += ps.length;
        }
        else if (is_segment_align_match(ps.type))
        {
            if (packed != nullptr && readHeadPos + ps.length <= readSize)
            {
                readChars.resize(ps.length);
                refChars.resize(ps.length);
                unpackReadChars(*packed, readHeadPos, ps.length, readChars.data());
                const char* const refChar(referenceChars(ref, refHeadPos, ps.length, refChars.data()));
                const char* const readChar(readChars.data());
                int* const fwd(fwdScore.data() + readHeadPos);
                int* const rev(revScore.data() + readHeadPos);
                for (unsigned j(0); j < ps.length; ++j)
                {
                    const int v((readChar[j] != refChar[j]) ? mismatchScore : matchScore);
                    const int w(((readChar[j] != 'N') && (refChar[j] != 'N')) ? v : 0);
                    fwd[j] += w;
                    rev[j] += w;
                }
            }
            else
            {
                for (unsigned j(0); j < ps.length; ++j)
                {
                    const unsigned readPos(readHeadPos + j);
                    const char readChar(readSeq.get_char(static_cast<pos_t>(readPos)));
                    const char refChar(ref.get_base(refHeadPos + static_cast<pos_t>(j)));
                    if ((readChar != 'N') && (refChar != 'N'))
                    {
                        const int v((readChar != refChar) ? mismatchScore : matchScore);
                        fwdScore[readPos] += v;
                        revScore[readPos] += v;
                    }
                }
            }
            readHeadPos += ps.length;
            refHeadPos += ps.length;
        }
        else if (ps.type == HARD_CLIP)
        {
        }
        else
        {
            std::ostringstream oss;
            oss << "Can't handle cigar code: " << segment_type_to_cigar_code(ps.type) << "\n";
            throw blt_exception(oss.str().c_str());
        }
    }
    validRange.set_begin_pos(0);
    validRange.set_end_pos(readSize);
    int fwdSum(0), fwdMin(minSegmentScore), revSum(0), revMin(minSegmentScore);
    for (unsigned i(0); i < readSize; ++i)
    {
        fwdSum += fwdScore[i];
        if (fwdSum <= fwdMin)
        {
            validRange.begin_pos = i + 1;
            fwdMin = fwdSum;
        }
        revSum += revScore[readSize - i - 1];
        if (revSum <= revMin)
        {
            validRange.end_pos = readSize - i - 1;
            revMin = revSum;
        }
    }
    if (validRange.end_pos <= validRange.begin_pos)
    {
        validRange.begin_pos = 0;
        validRange.end_pos = 0;
    }
}

void active_region_insert_aligned_segment(ActiveRegionReadBuffer& buffer, const unsigned alignId, const reference_contig_segment& ref,
                                          const bam_seq_base& readSeq, const unsigned readOffset, const pos_t refHeadPos, const unsigned length)
{
    const bam_seq* packed(dynamic_cast<const bam_seq*>(&readSeq));
    if (refHeadPos < 0 || packed == nullptr)
    {
        // (a segment that starts before position 0 wraps the ring index through the unsigned conversion: the reference's own loop)
        for (unsigned j(0); j < length; ++j)
        {
            const pos_t refPos(refHeadPos + static_cast<pos_t>(j));
            const char baseChar(readSeq.get_char(static_cast<pos_t>(readOffset + j)));
            if (ref.get_base(refPos) != baseChar) buffer.insertMismatch(alignId, refPos, baseChar);
            else buffer.insertMatch(alignId, refPos);
        }
        return;
    }
    static const unsigned ringSize(ActiveRegionReadBuffer::MaxBufferSize);
    const unsigned idIndex(alignId % ActiveRegionReadBuffer::MaxDepth);
    ActiveRegionReadBuffer::VariantType* const variantRow(buffer._variantInfo[idIndex].data());
    char* const snvRow(buffer._snvBuffer[idIndex]);
    // the rows are 4 kB apart and come round once per thousand reads: ask for the next read's slots (the next id, about here) while
    // this read's are written
    const ActiveRegionReadBuffer::VariantType* const nextRow(buffer._variantInfo[(idIndex + 1u) % ActiveRegionReadBuffer::MaxDepth].data());
    unsigned* const variantCounter(buffer._variantCounter.data());
    unsigned* const depth(buffer._depth.data());
    std::vector<align_id_t>* const alignIds(buffer._positionToAlignIds.data());
    // the segment in runs that do not wrap the ring: per run the bases are unpacked and compared first, then each of the buffer's
    // arrays gets its stores in one pass over consecutive slots (the per-position result is the same whatever the order between arrays)
    static const unsigned maxRun(512);
    char baseChar[maxRun], refBuffer[maxRun];
    unsigned char isMismatch[maxRun];
    unsigned done(0);
    while (done < length)
    {
        const unsigned index0(static_cast<unsigned>(refHeadPos + static_cast<pos_t>(done)) % ringSize);
        const unsigned run(std::min(std::min(length - done, ringSize - index0), maxRun));
        const pos_t refPos0(refHeadPos + static_cast<pos_t>(done));
        for (unsigned j(0); j < run; j += 16) __builtin_prefetch(nextRow + index0 + j, 1);
        unpackReadChars(*packed, readOffset + done, run, baseChar);
        const char* const refChar(referenceChars(ref, refPos0, run, refBuffer));
        for (unsigned j(0); j < run; ++j) isMismatch[j] = (refChar[j] != baseChar[j]) ? 1 : 0;
        // insertMismatch / insertMatch: addVariantCount ...
        for (unsigned j(0); j < run; ++j)
        {
            variantCounter[index0 + j] += isMismatch[j] * static_cast<unsigned>(ActiveRegionReadBuffer::MismatchWeight);
            ++depth[index0 + j];
        }
        // ... setMismatch / setMatch ...
        for (unsigned j(0); j < run; ++j)
        {
            variantRow[index0 + j] = isMismatch[j] ? ActiveRegionReadBuffer::MISMATCH : ActiveRegionReadBuffer::MATCH;
        }
        for (unsigned j(0); j < run; ++j)
        {
            if (isMismatch[j]) snvRow[index0 + j] = baseChar[j];
        }
        // ... addAlignIdToPos
        for (unsigned j(0); j < run; ++j)
        {
            std::vector<align_id_t>& ids(alignIds[index0 + j]);
            if (ids.empty() || ids.back() != alignId) ids.push_back(alignId);
        }
        done += run;
    }
}

bool repeat_span_update(const reference_contig_segment& ref, const pos_t pos, const unsigned maxRepeatUnitLength, const unsigned ringSize,
                        const unsigned minRepeatSpan, std::vector<std::vector<unsigned>>& repeatSpan, std::vector<bool>& isAnchor)
{
    // the reference evaluates, for each of the 50 unit lengths, two bounds-checked base look-ups and two ring indices by division;
    // here: the segment's characters directly (so only where [pos - maxRepeatUnitLength, pos] lies inside it), the two rows once
    static const unsigned maxUnits(64);
    const unsigned lookBack((maxRepeatUnitLength + 7u) & ~7u);
    if (pos < 0 || ringSize < 2 || lookBack > maxUnits || pos - static_cast<pos_t>(lookBack) < ref.get_offset() || pos >= ref.end()) return false;
    const char* const refChar(ref.seq().data() + (pos - ref.get_offset()));
    const char base(refChar[0]);
    // prevBase[unit - 1] = the base `unit` positions back: eight at a time, byte-swapped, so that the loop below runs forward over
    // three plain arrays and the compiler can vectorise it
    unsigned char prevBase[maxUnits];
    for (unsigned k(0); k < lookBack; k += 8)
    {
        uint64_t v;
        std::memcpy(&v, refChar - static_cast<int>(k) - 8, 8);
        v = __builtin_bswap64(v);
        std::memcpy(prevBase + k, &v, 8);
    }
    const unsigned posIndex(static_cast<unsigned>(pos) % ringSize);
    const unsigned* __restrict const prevRow(repeatSpan[static_cast<unsigned>(pos - 1) % ringSize].data());
    unsigned* __restrict const row(repeatSpan[posIndex].data());
    isAnchor[posIndex] = true;
    // (the comparison outcome is as good as random: selected by mask, not by branch)
    unsigned isAnyRepeat(0);
    for (unsigned k(0); k < maxRepeatUnitLength; ++k)
    {
        const unsigned unit(k + 1u);
        const unsigned isSame(static_cast<unsigned>(prevBase[k] == static_cast<unsigned char>(base)) & static_cast<unsigned>(prevBase[k] != 'N'));
        const unsigned mask(0u - isSame);
        const unsigned span((mask & (prevRow[k] + 1u)) | (~mask & unit));
        row[k] = span;
        isAnyRepeat |= static_cast<unsigned>(span >= unit * 2u) & static_cast<unsigned>(span >= minRepeatSpan);
    }
    if (! isAnyRepeat) return true;
    for (unsigned unit(1); unit <= maxRepeatUnitLength; ++unit)
    {
        const unsigned span(row[unit - 1]);
        if (! (span >= unit * 2u && span >= minRepeatSpan)) continue;
        if (span == unit * 2u || span == minRepeatSpan)
        {
            // (the flags only ever go to false here: the order between unit lengths does not matter)
            for (pos_t prevPos(pos - 1u); prevPos > static_cast<pos_t>(pos - span); --prevPos)
            {
                const pos_t prevPosIndex(prevPos % ringSize);
                isAnchor[prevPosIndex] = false;
            }
        }
        isAnchor[posIndex] = false;
    }
    return true;
}

}
