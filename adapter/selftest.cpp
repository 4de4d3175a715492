// selftest.cpp -- the adapter's host-side restatements against the reference functions they stand in for, on seeded random inputs
// (tests/test_adapter_selftest.py runs the program; it is linked like the `*_dbl` adapter programs, with this main()).
//
// The end-to-end tests check the same functions through whole VCFs; their synthetic genomes hold no 'N' stretches and their regions
// do not start at position 0, so the corners are exercised here: reference segments with N runs and short offsets, reads with N
// bases, odd packed offsets, leading / trailing indels and soft clips, rings that wrap, keys that move a RangeMap's bounds either way.
//
//   valid_alignment_range            vs get_valid_alignment_range           (L/starling_common/starling_read_util.cpp:218-329)
//   repeat_span_update               vs ReferenceRepeatFinder::updateRepeatSpan (L/starling_common/ReferenceRepeatFinder.cpp:26-59)
//   depth_buffer_add_alignment       vs add_alignment_to_depth_buffer       (L/blt_util/depth_buffer_util.cpp:29-48)
//   active_region_insert_aligned_segment vs ActiveRegionReadBuffer::insertMatch / insertMismatch (ActiveRegionReadBuffer.cpp:26-50)
//   is_plain_bam_record              vs the two validity loops of checkBamRecord (L/starling_common/starling_pos_processor_util.cpp:141-240)
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iosfwd>
#include <iostream>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "blt_util/blt_exception.hh"
#include "blt_util/blt_types.hh"
#include "blt_util/known_pos_range2.hh"
#include "blt_util/pos_range.hh"
#include "blt_util/reference_contig_segment.hh"
#include "blt_util/depth_buffer.hh"
#include "blt_util/depth_buffer_util.hh"
#include "htsapi/bam_seq.hh"
#include "starling_common/IndelBuffer.hh"
#include "starling_common/ReferenceRepeatFinder.hh"
#include "starling_common/alignment.hh"
#include "starling_common/indel.hh"
#include "starling_common/starling_types.hh"

#define private public
#include "starling_common/ActiveRegionReadBuffer.hh"
#undef private

#include "starling_common/starling_read_util.hh"

#include "selftest_repeat_finder.hh"
#include "sk_adapter.hh"

namespace
{

typedef std::mt19937_64 Rng;

unsigned uniform(Rng& rng, const unsigned n) { return static_cast<unsigned>(rng() % n); }

std::string randomReference(Rng& rng, const unsigned length, const bool isRepetitive)
{
    static const char bases[] = "ACGT";
    std::string s;
    while (s.size() < length)
    {
        const unsigned kind(uniform(rng, 10));
        if (kind == 0)
        {
            s.append(1 + uniform(rng, 70), 'N');
        }
        else if (isRepetitive && kind < 5)
        {
            // a short tandem repeat: unit of 1..12, 2..40 copies
            std::string unit;
            for (unsigned i(0), n(1 + uniform(rng, 12)); i < n; ++i) unit.push_back(bases[uniform(rng, 4)]);
            for (unsigned i(0), n(2 + uniform(rng, 39)); i < n; ++i) s += unit;
        }
        else
        {
            for (unsigned i(0), n(1 + uniform(rng, 60)); i < n; ++i) s.push_back(bases[uniform(rng, 4)]);
        }
    }
    s.resize(length);
    return s;
}

/// a read of the reference at `pos` with the given path, packed as a BAM record packs it (+ `offset` leading bases that are not the read's)
struct PackedRead
{
    std::vector<uint8_t> packed;
    std::string chars;
    unsigned offset;
};

uint8_t codeOf(const char c)
{
    switch (c)
    {
    case 'A': return 1;
    case 'C': return 2;
    case 'G': return 4;
    case 'T': return 8;
    default: return 15;
    }
}

PackedRead packRead(Rng& rng, const std::string& chars)
{
    PackedRead r;
    r.chars = chars;
    r.offset = uniform(rng, 4);
    std::string all(r.offset, 'A');
    all += chars;
    r.packed.assign((all.size() + 1) / 2 + 1, 0);
    for (size_t i(0); i < all.size(); ++i)
    {
        const uint8_t code(codeOf(all[i]));
        r.packed[i / 2] |= static_cast<uint8_t>((i % 2) ? code : (code << 4));
    }
    return r;
}

alignment randomAlignment(Rng& rng, const reference_contig_segment& ref, std::string& readChars)
{
    using namespace ALIGNPATH;
    static const char bases[] = "ACGTN";
    alignment al;
    al.is_fwd_strand = (uniform(rng, 2) == 0);
    const pos_t refBegin(ref.get_offset()), refEnd(ref.end());
    al.pos = refBegin - 20 + static_cast<pos_t>(uniform(rng, static_cast<unsigned>(refEnd - refBegin) + 40));
    readChars.clear();
    pos_t refPos(al.pos);
    const unsigned segmentCount(1 + uniform(rng, 6));
    static const double rates[3] = {0.35, 0.1, 0.01};
    const double mismatchRate(rates[uniform(rng, 3)]);
    if (uniform(rng, 4) == 0) al.path.push_back(path_segment(HARD_CLIP, 1 + uniform(rng, 5)));
    if (uniform(rng, 4) == 0)
    {
        const unsigned n(1 + uniform(rng, 10));
        al.path.push_back(path_segment(SOFT_CLIP, n));
        for (unsigned i(0); i < n; ++i) readChars.push_back(bases[uniform(rng, 5)]);
    }
    for (unsigned k(0); k < segmentCount; ++k)
    {
        const unsigned kind((k % 2 == 0) ? 0u : 1u + uniform(rng, 2));
        if (kind == 0)
        {
            const unsigned n(1 + uniform(rng, 80));
            al.path.push_back(path_segment(MATCH, n));
            for (unsigned i(0); i < n; ++i)
            {
                char c(ref.get_base(refPos + static_cast<pos_t>(i)));
                if (uniform(rng, 1000) < static_cast<unsigned>(mismatchRate * 1000)) c = bases[uniform(rng, 5)];
                readChars.push_back(c);
            }
            refPos += static_cast<pos_t>(n);
        }
        else if (kind == 1)
        {
            const unsigned n(1 + uniform(rng, 8));
            al.path.push_back(path_segment(INSERT, n));
            for (unsigned i(0); i < n; ++i) readChars.push_back(bases[uniform(rng, 4)]);
        }
        else
        {
            const unsigned n(1 + uniform(rng, 12));
            al.path.push_back(path_segment(DELETE, n));
            refPos += static_cast<pos_t>(n);
        }
    }
    if (uniform(rng, 4) == 0)
    {
        const unsigned n(1 + uniform(rng, 10));
        al.path.push_back(path_segment(SOFT_CLIP, n));
        for (unsigned i(0); i < n; ++i) readChars.push_back(bases[uniform(rng, 5)]);
    }
    return al;
}

unsigned g_failures(0);

void fail(const char* what, const unsigned trial)
{
    std::cout << "FAIL " << what << " trial " << trial << "\n";
    ++g_failures;
}

void testValidAlignmentRange(Rng& rng)
{
    unsigned fullCount(0), trimmedCount(0);
    for (unsigned trial(0); trial < 20000; ++trial)
    {
        reference_contig_segment ref;
        ref.seq() = randomReference(rng, 300 + uniform(rng, 200), false);
        ref.set_offset(static_cast<pos_t>(uniform(rng, 3) == 0 ? 0 : uniform(rng, 5000)));
        std::string readChars;
        const alignment al(randomAlignment(rng, ref, readChars));
        const PackedRead pr(packRead(rng, readChars));
        const bam_seq readSeq(pr.packed.data(), static_cast<uint16_t>(readChars.size()), static_cast<uint16_t>(pr.offset));
        const rc_segment_bam_seq refSeq(ref);
        pos_range want, got;
        get_valid_alignment_range(al, refSeq, readSeq, want);
        sk_adapter::valid_alignment_range(al, ref, readSeq, got);
        if (! (want == got)) fail("valid_alignment_range", trial);
        if (want.begin_pos == 0 && want.end_pos == static_cast<pos_t>(readChars.size())) ++fullCount;
        else ++trimmedCount;
        // the same read as plain characters: the path without the packed fast route
        const string_bam_seq plainSeq(readChars);
        pos_range gotPlain;
        sk_adapter::valid_alignment_range(al, ref, plainSeq, gotPlain);
        if (! (want == gotPlain)) fail("valid_alignment_range (string_bam_seq)", trial);
    }
    std::cout << "valid_alignment_range: full " << fullCount << " trimmed " << trimmedCount << "\n";
    if (fullCount < 2000 || trimmedCount < 2000) fail("valid_alignment_range coverage", 0);
}

void testRepeatSpan(Rng& rng)
{
    unsigned anchorCount(0), repeatCount(0);
    for (unsigned trial(0); trial < 60; ++trial)
    {
        reference_contig_segment ref;
        ref.seq() = randomReference(rng, 3000 + uniform(rng, 3000), true);
        ref.set_offset(static_cast<pos_t>(trial % 3 == 0 ? 0 : uniform(rng, 100000)));
        static const unsigned maxUnit(50), ringSize(1000), minSpan(3);
        OriginalRepeatFinder original(ref, maxUnit, ringSize, minSpan);
        std::vector<std::vector<unsigned>> repeatSpan(ringSize, std::vector<unsigned>(maxUnit));
        std::vector<bool> isAnchor(ringSize);
        // the way ActiveRegionReadBuffer::setEndPos drives it (ActiveRegionReadBuffer.cpp:176-187): initRepeatSpan at the first
        // position, then one update per position -- positions before and after the segment included
        const pos_t begin(ref.get_offset() - (ref.get_offset() >= 30 ? 30 : 0)), end(ref.end() + 120);
        // (initRepeatSpan is the original's own driver over updateRepeatSpan: replayed here over both)
        pos_t minPos(begin - 2 * static_cast<pos_t>(maxUnit) + 1);
        if (minPos < ref.get_offset()) minPos = ref.get_offset();
        original.initRepeatSpan(begin);
        for (unsigned u(1); u <= maxUnit; ++u) repeatSpan[static_cast<unsigned>(minPos) % ringSize][u - 1] = u;
        struct Mirror
        {
            static void update(const reference_contig_segment& r, const pos_t pos, std::vector<std::vector<unsigned>>& span, std::vector<bool>& anchor)
            {
                if (sk_adapter::repeat_span_update(r, pos, maxUnit, ringSize, minSpan, span, anchor)) return;
                // not handled (look-back outside the segment): the reference's loop, on the mirror's tables
                const char base(r.get_base(pos));
                const unsigned posIndex(static_cast<unsigned>(pos) % ringSize);
                anchor[posIndex] = true;
                for (unsigned unit(1); unit <= maxUnit; ++unit)
                {
                    const char prevBase(r.get_base(pos - unit));
                    unsigned s;
                    if (prevBase != 'N' && base == prevBase) s = span[static_cast<unsigned>(pos - 1) % ringSize][unit - 1] + 1;
                    else s = unit;
                    span[posIndex][unit - 1] = s;
                    if (s >= unit * 2u && s >= minSpan)
                    {
                        if (s == unit * 2u || s == minSpan)
                        {
                            for (pos_t prevPos(pos - 1u); prevPos > static_cast<pos_t>(pos - s); --prevPos) anchor[static_cast<pos_t>(prevPos % ringSize)] = false;
                        }
                        anchor[posIndex] = false;
                    }
                }
            }
        };
        for (pos_t p(minPos); p < static_cast<pos_t>(begin + maxUnit * 2u); ++p) Mirror::update(ref, p, repeatSpan, isAnchor);
        for (pos_t p(begin); p < end; ++p)
        {
            const pos_t q(p + static_cast<pos_t>(maxUnit * 2u));
            original.updateRepeatSpan(q);
            Mirror::update(ref, q, repeatSpan, isAnchor);
            const bool want(original.isAnchor(p)), got(isAnchor[static_cast<unsigned>(p) % ringSize]);
            if (want != got)
            {
                fail("repeat_span_update (anchor flag)", trial);
                break;
            }
            if (want) ++anchorCount;
            else ++repeatCount;
        }
    }
    std::cout << "repeat_span_update: anchors " << anchorCount << " repeat positions " << repeatCount << "\n";
    if (anchorCount < 20000 || repeatCount < 20000) fail("repeat_span_update coverage", 0);
}

void testDepthBuffer(Rng& rng)
{
    uint64_t total(0);
    for (unsigned trial(0); trial < 300; ++trial)
    {
        depth_buffer want, got;
        pos_t head(static_cast<pos_t>(uniform(rng, 100000)));
        pos_t cleared(head - 400);
        for (unsigned step(0); step < 400; ++step)
        {
            using namespace ALIGNPATH;
            path_t path;
            if (uniform(rng, 5) == 0) path.push_back(path_segment(SOFT_CLIP, 1 + uniform(rng, 9)));
            const unsigned segmentCount(1 + uniform(rng, 4));
            for (unsigned k(0); k < segmentCount; ++k)
            {
                if (k % 2 == 0) path.push_back(path_segment(MATCH, 1 + uniform(rng, (trial % 10 == 0) ? 3000 : 150)));
                else if (uniform(rng, 2)) path.push_back(path_segment(DELETE, 1 + uniform(rng, 40)));
                else path.push_back(path_segment(INSERT, 1 + uniform(rng, 10)));
            }
            // mostly forward in small steps; now and then a start before everything buffered so far
            pos_t pos(head + static_cast<pos_t>(uniform(rng, 12)));
            if (uniform(rng, 25) == 0) pos = head - static_cast<pos_t>(uniform(rng, 300));
            else head = pos;
            add_alignment_to_depth_buffer(pos, path, want);
            sk_adapter::depth_buffer_add_alignment(pos, path, got);
            // the buffers are cleared position by position behind the head (POST_CALL stage)
            const pos_t clearTo(head - 200 - static_cast<pos_t>(uniform(rng, 100)));
            for (; cleared < clearTo; ++cleared)
            {
                want.clear_pos(cleared);
                got.clear_pos(cleared);
            }
            if (step % 16 == 0)
            {
                for (pos_t p(head - 700); p < head + 3400; ++p)
                {
                    if (want.val(p) != got.val(p))
                    {
                        fail("depth_buffer_add_alignment", trial);
                        step = 1000000;
                        break;
                    }
                    total += want.val(p);
                }
            }
        }
    }
    std::cout << "depth_buffer_add_alignment: compared depth sum " << total << "\n";
    if (total < 1000000) fail("depth_buffer coverage", 0);
}

void testActiveRegionBuffer(Rng& rng)
{
    uint64_t mismatches(0), bases(0);
    for (unsigned trial(0); trial < 40; ++trial)
    {
        reference_contig_segment ref;
        ref.seq() = randomReference(rng, 4000, false);
        ref.set_offset(static_cast<pos_t>(trial % 4 == 0 ? 0 : uniform(rng, 50000)));
        // (the buffer keeps a reference to the indel buffer for its other members; the bookkeeping under test never follows it)
        alignas(IndelBuffer) static char indelBufferStorage[sizeof(IndelBuffer)];
        IndelBuffer& indelBuffer(*reinterpret_cast<IndelBuffer*>(indelBufferStorage));
        std::unique_ptr<ActiveRegionReadBuffer> want(new ActiveRegionReadBuffer(ref, 0.f, indelBuffer));
        std::unique_ptr<ActiveRegionReadBuffer> got(new ActiveRegionReadBuffer(ref, 0.f, indelBuffer));
        unsigned alignId(uniform(rng, 5000));
        pos_t pos(ref.get_offset() - (ref.get_offset() > 0 ? 5 : 0));
        while (pos < ref.end() - 200)
        {
            pos += static_cast<pos_t>(uniform(rng, 6));
            std::string readChars;
            alignment al;
            {
                // a read starting at pos, all matches and a few clips / indels
                al = randomAlignment(rng, ref, readChars);
                al.pos = pos;
                // the characters were drawn for another position: redraw the match segments here
                readChars.clear();
                pos_t refPos(pos);
                static const char basesN[] = "ACGTN";
                for (const ALIGNPATH::path_segment& ps : al.path)
                {
                    using namespace ALIGNPATH;
                    if (is_segment_align_match(ps.type))
                    {
                        for (unsigned i(0); i < ps.length; ++i)
                        {
                            char c(ref.get_base(refPos + static_cast<pos_t>(i)));
                            if (uniform(rng, 40) == 0) c = basesN[uniform(rng, 5)];
                            readChars.push_back(c);
                        }
                        refPos += static_cast<pos_t>(ps.length);
                    }
                    else if (ps.type == INSERT || ps.type == SOFT_CLIP)
                    {
                        for (unsigned i(0); i < ps.length; ++i) readChars.push_back(basesN[uniform(rng, 4)]);
                    }
                    else if (ps.type == DELETE) refPos += static_cast<pos_t>(ps.length);
                }
            }
            const PackedRead pr(packRead(rng, readChars));
            const bam_seq readSeq(pr.packed.data(), static_cast<uint16_t>(readChars.size()), static_cast<uint16_t>(pr.offset));
            ++alignId;
            unsigned readOffset(0);
            pos_t refHeadPos(al.pos);
            for (const ALIGNPATH::path_segment& ps : al.path)
            {
                using namespace ALIGNPATH;
                if (is_segment_align_match(ps.type))
                {
                    for (unsigned j(0); j < ps.length; ++j) // starling_pos_processor_indel_util.cpp:466-483
                    {
                        const pos_t refPos(refHeadPos + static_cast<pos_t>(j));
                        const char baseChar(readSeq.get_char(static_cast<pos_t>(readOffset + j)));
                        if (ref.get_base(refPos) != baseChar)
                        {
                            want->insertMismatch(alignId, refPos, baseChar);
                            ++mismatches;
                        }
                        else want->insertMatch(alignId, refPos);
                        ++bases;
                    }
                    sk_adapter::active_region_insert_aligned_segment(*got, alignId, ref, readSeq, readOffset, refHeadPos, ps.length);
                }
                if (is_segment_type_read_length(ps.type)) readOffset += ps.length;
                if (is_segment_type_ref_length(ps.type)) refHeadPos += ps.length;
            }
            // behind the reads the ring is cleared position by position (ActiveRegionReadBuffer::clearPos)
            if (pos - 400 >= 0)
            {
                want->clearPos(pos - 400);
                got->clearPos(pos - 400);
            }
        }
        bool isSame(want->_variantCounter == got->_variantCounter && want->_depth == got->_depth && want->_positionToAlignIds == got->_positionToAlignIds &&
                    want->_variantInfo == got->_variantInfo);
        for (unsigned i(0); isSame && i < ActiveRegionReadBuffer::MaxDepth; ++i)
        {
            // (only the slots of mismatches are ever written or read in the character table)
            for (unsigned j(0); j < ActiveRegionReadBuffer::MaxBufferSize; ++j)
            {
                if (want->_variantInfo[i][j] == ActiveRegionReadBuffer::MISMATCH && want->_snvBuffer[i][j] != got->_snvBuffer[i][j]) isSame = false;
            }
        }
        if (! isSame) fail("active_region_insert_aligned_segment", trial);
    }
    std::cout << "active_region_insert_aligned_segment: bases " << bases << " mismatches " << mismatches << "\n";
    if (bases < 1000000 || mismatches < 10000) fail("active_region coverage", 0);
}

void testPlainRecord(Rng& rng)
{
    unsigned plain(0), notPlain(0);
    for (unsigned trial(0); trial < 200000; ++trial)
    {
        const unsigned readSize(1 + uniform(rng, trial % 50 == 0 ? 400 : 160));
        std::vector<uint8_t> record((readSize + 1) / 2 + readSize + 8, 0);
        uint8_t* const packed(record.data() + 3);
        uint8_t* const qual(packed + ((readSize + 1) / 2));
        static const uint8_t valid[5] = {1, 2, 4, 8, 15};
        const unsigned badBase(uniform(rng, 6) == 0 ? uniform(rng, readSize) : readSize);
        const unsigned badQual(uniform(rng, 6) == 0 ? uniform(rng, readSize) : readSize);
        bool want(true);
        for (unsigned i(0); i < readSize; ++i)
        {
            uint8_t code(valid[uniform(rng, 5)]);
            if (i == badBase)
            {
                static const uint8_t invalid[11] = {0, 3, 5, 6, 7, 9, 10, 11, 12, 13, 14};
                code = invalid[uniform(rng, 11)];
                want = false;
            }
            packed[i / 2] |= static_cast<uint8_t>((i % 2) ? code : (code << 4));
            qual[i] = static_cast<uint8_t>(uniform(rng, 71));
            if (i == badQual)
            {
                qual[i] = static_cast<uint8_t>(uniform(rng, 2) ? 255 : 71 + uniform(rng, 100));
                want = false;
            }
        }
        if ((readSize % 2) && uniform(rng, 2)) packed[readSize / 2] |= static_cast<uint8_t>(uniform(rng, 16)); // (the pad nibble is not part of the read)
        const bool got(sk_adapter::is_plain_bam_record(qual, readSize));
        if (want != got) fail("is_plain_bam_record", trial);
        if (want) ++plain;
        else ++notPlain;
    }
    if (sk_adapter::is_plain_bam_record(nullptr, 0)) fail("is_plain_bam_record (empty read)", 0);
    std::cout << "is_plain_bam_record: plain " << plain << " not plain " << notPlain << "\n";
}

}

int main(int argc, char** argv)
{
    // (the test suite runs the default seed; `adapter_selftest SEED` for other inputs)
    Rng rng(argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 20260926ull);
    testValidAlignmentRange(rng);
    testRepeatSpan(rng);
    testDepthBuffer(rng);
    testActiveRegionBuffer(rng);
    testPlainRecord(rng);
    std::cout << (g_failures == 0 ? "adapter selftest: all passed\n" : "adapter selftest: FAILED\n");
    return (g_failures == 0) ? 0 : 1;
}
