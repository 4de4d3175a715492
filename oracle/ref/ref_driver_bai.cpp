// ref_driver_bai.cpp -- C entry point over htslib's own index query (hts_idx_load + sam_itr_queryi, the calls of
// L/htsapi/bam_streamer.cpp:154-183, :228), htslib 1.7-6-g6d2bfb7 from the reference's redist/ tarball.
//
// TEST INFRASTRUCTURE ONLY; contains no reference code: it loads the index of a BAM file, asks for a region and copies the chunk list
// the iterator holds (hts_itr_t::off), which is what sk_bai_query restates.

extern "C" {
#include "htslib/hts.h"
#include "htslib/sam.h"
}

#include <cstdint>

extern "C" {

/// chunks: (begin, end) virtual offsets, capacity cap pairs.  Returns the number of chunks, -1 = no index / no iterator.
int ref_bai_query(const char* bam_path, int32_t tid, int32_t begin, int32_t end, uint64_t* chunks, int32_t cap)
{
    hts_idx_t* idx = hts_idx_load(bam_path, HTS_FMT_BAI);
    if (!idx) return -1;
    hts_itr_t* it = sam_itr_queryi(idx, tid, begin, end);
    int n = -1;
    if (it) {
        n = it->finished ? 0 : it->n_off;
        for (int i = 0; i < n && i < cap; ++i) {
            chunks[2 * i] = it->off[i].u;
            chunks[2 * i + 1] = it->off[i].v;
        }
        hts_itr_destroy(it);
    }
    hts_idx_destroy(idx);
    return n;
}

} // extern "C"
