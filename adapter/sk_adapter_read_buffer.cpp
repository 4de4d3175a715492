// sk_adapter_read_buffer.cpp -- the read segments buffered in a stage window, in the read buffer's order, in one walk.
//
// The reference visits its read buffer one position at a time (starling_read_buffer::get_pos_read_segment_iter,
// L/starling_common/starling_read_buffer.cpp:117-123: a look-up in the position map per position, then read_segment_iter::get_ptr,
// :187-198: a look-up in the read map per segment).  The adapter's stage windows cover thousands of positions, so the same
// segments in the same order come from one ordered walk of the position map, with the read map entered at the entry after the
// previous hit (reads of a window carry consecutive ids).  The two maps are private members; like the active-region ring buffer
// (sk_adapter_active_region_buffer.cpp) this one translation unit sees the class with its private section opened -- a maintainer
// would add the walk as a member function.
#include <algorithm>
#include <cassert>
#include <iosfwd>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "starling_common/starling_read.hh"
#include "boost/utility.hpp"

#define private public
#include "starling_common/starling_read_buffer.hh"
#undef private

#include "sk_adapter_access.hh"

namespace sk_adapter
{

void collect_window_segments(starling_read_buffer& buffer, const pos_t begin, const pos_t end, std::vector<WindowSegment>& segments)
{
    starling_read_buffer::read_data_t& readData(buffer._read_data);
    starling_read_buffer::read_data_t::iterator cursor(readData.end());
    const starling_read_buffer::pos_group_t::const_iterator groupEnd(buffer._pos_group.end());
    for (starling_read_buffer::pos_group_t::const_iterator group(buffer._pos_group.lower_bound(begin)); group != groupEnd && group->first < end; ++group)
    {
        for (const starling_read_buffer::segment_t& key : group->second)
        {
            if (cursor != readData.end()) ++cursor;
            if (cursor == readData.end() || cursor->first != key.first) cursor = readData.find(key.first);
            if (cursor == readData.end()) break; // (get_ptr returns null for a read that is gone, which ends the position's loop)
            if (key.second != 0) throw blt_exception("strelka_amd adapter: spliced (RNA) read segments are not supported on this path");
            WindowSegment ws;
            ws.rseg = &(cursor->second->get_segment(key.second));
            ws.bufferPos = group->first;
            segments.push_back(ws);
        }
    }
}

}
