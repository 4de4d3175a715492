// sk_adapter_depth_buffer.cpp -- an input read's aligned positions counted into the estimated-depth buffer, a run at a time.
//
// add_alignment_to_depth_buffer (L/blt_util/depth_buffer_util.cpp:29-48) calls depth_buffer::inc for every matched position of every
// input read, and each inc is a RangeMap::getRef (L/blt_util/RangeMap.hh:107-138): two range tests, a ring index, an occupancy test.
// Nearly all of a read's positions lie inside the key range the buffer already holds (the reads arrive sorted; a read adds a few new
// positions at the front), so those are visited here as consecutive ring slots with the range tests and the index taken once per
// run; every position outside the present key range still goes through getRef itself, which moves the bounds exactly as before.
// The map is a private member of the buffer (and its fields private to the map); like the read buffer and the active-region ring
// (sk_adapter_read_buffer.cpp, sk_adapter_active_region_buffer.cpp) this translation unit sees the two classes with their private
// sections opened -- a maintainer would add depth_buffer::inc(pos, length).
#include <algorithm>
#include <cassert>
#include <iostream>
#include <sstream>
#include <vector>

#include "blt_util/blt_exception.hh"
#include "blt_util/blt_types.hh"
#include "boost/dynamic_bitset.hpp"

#define private public
#define protected public
#include "blt_util/RangeMap.hh"
#include "blt_util/depth_buffer.hh"
#undef protected
#undef private

#include "sk_adapter.hh"

#include "blt_util/align_path.hh"

namespace sk_adapter
{

namespace
{

void incrementRange(depth_buffer& buffer, const pos_t begin, const unsigned length)
{
    RangeMap<pos_t, unsigned>& map(buffer._data);
    const pos_t end(begin + static_cast<pos_t>(length));
    pos_t key(begin);
    while (key < end)
    {
        if (map._isEmpty || key < map._minKey || key > map._maxKey)
        {
            map.getRef(key) += 1; // (a key that moves the map's bounds: the reference's own path)
            ++key;
            continue;
        }
        // keys [key, runEnd) lie inside [_minKey, _maxKey] and on consecutive slots of the ring
        const unsigned dataSize(static_cast<unsigned>(map._data.size()));
        const unsigned index(map.getKeyIndex(key));
        const pos_t insideEnd(std::min(end, map._maxKey + 1));
        const unsigned run(std::min(static_cast<unsigned>(insideEnd - key), dataSize - index));
        unsigned* const data(map._data.data() + index);
        for (unsigned j(0); j < run; ++j)
        {
            if (! map._occup.test(index + j))
            {
                data[j] = 0; // (ZeroT)
                map._occup.set(index + j);
            }
            data[j] += 1;
        }
        key += static_cast<pos_t>(run);
    }
}

}

void depth_buffer_add_alignment(const pos_t pos, const ALIGNPATH::path_t& path, depth_buffer& buffer)
{
    using namespace ALIGNPATH;
    pos_t refHeadPos(pos);
    for (const path_segment& ps : path)
    {
        if (is_segment_align_match(ps.type)) incrementRange(buffer, refHeadPos, ps.length);
        if (is_segment_type_ref_length(ps.type)) refHeadPos += ps.length;
    }
}

}
