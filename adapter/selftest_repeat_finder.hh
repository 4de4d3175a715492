// selftest_repeat_finder.hh -- see selftest_repeat_finder.cpp
#pragma once

#include "blt_util/blt_types.hh"

struct reference_contig_segment;

struct OriginalRepeatFinder
{
    OriginalRepeatFinder(const reference_contig_segment& ref, const unsigned maxRepeatUnitLength, const unsigned ringSize, const unsigned minRepeatSpan);
    ~OriginalRepeatFinder();
    void initRepeatSpan(const pos_t pos);
    void updateRepeatSpan(const pos_t pos);
    bool isAnchor(const pos_t pos) const;

private:
    OriginalRepeatFinder(const OriginalRepeatFinder&);
    void* _impl;
};
