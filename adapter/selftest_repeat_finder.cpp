// selftest_repeat_finder.cpp -- the reference's own ReferenceRepeatFinder compiled under another name, for adapter/selftest.cpp (the
// adapter programs link the hooked copy under the real name; the class is renamed by macro while its header and source are read,
// both from where they lie in the reference tree: -DSK_REFERENCE_REPEAT_FINDER_CPP=... in adapter/Makefile)
#include <vector>

#include "blt_util/blt_types.hh"
#include "blt_util/reference_contig_segment.hh"

#define ReferenceRepeatFinder ReferenceRepeatFinderOriginal
#include SK_REFERENCE_REPEAT_FINDER_CPP
#undef ReferenceRepeatFinder

#include "selftest_repeat_finder.hh"

OriginalRepeatFinder::OriginalRepeatFinder(const reference_contig_segment& ref, const unsigned maxRepeatUnitLength, const unsigned ringSize,
                                           const unsigned minRepeatSpan)
    : _impl(new ReferenceRepeatFinderOriginal(ref, maxRepeatUnitLength, ringSize, minRepeatSpan))
{}

OriginalRepeatFinder::~OriginalRepeatFinder() { delete static_cast<ReferenceRepeatFinderOriginal*>(_impl); }
void OriginalRepeatFinder::initRepeatSpan(const pos_t pos) { static_cast<ReferenceRepeatFinderOriginal*>(_impl)->initRepeatSpan(pos); }
void OriginalRepeatFinder::updateRepeatSpan(const pos_t pos) { static_cast<ReferenceRepeatFinderOriginal*>(_impl)->updateRepeatSpan(pos); }
bool OriginalRepeatFinder::isAnchor(const pos_t pos) const { return static_cast<const ReferenceRepeatFinderOriginal*>(_impl)->isAnchor(pos); }
