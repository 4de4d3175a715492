// bam_feed.cpp -- host side of the feed: finding the BGZF blocks of a file image and the BAM
// records of an inflated stream.  Both are chains of length fields (each block / record says where the next one starts), a few
// nanoseconds per element on one core; the bytes themselves are inflated and decoded on the device (csrc/bam_feed.hip).
// Formats: SAM specification v1, sections 4.1 (BGZF) and 4.2 (BAM); in the reference this is htslib (bgzf.c bgzf_read_block,
// sam.c bam_read1) behind L/htsapi/bam_streamer.cpp:268.

#include "strelka_amd.h"

#include <cstring>

namespace
{
inline uint32_t le16(const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8); }
inline uint32_t le32(const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }
} // namespace

extern "C" {

int64_t sk_bgzf_scan(const uint8_t* data, int64_t n_bytes, int64_t* block_off, int64_t* out_off, int32_t max_blocks)
{
    if (!data || n_bytes < 0 || max_blocks < 0 || (max_blocks > 0 && (!block_off || !out_off))) return -1;
    int64_t at = 0, out = 0;
    int32_t n = 0;
    while (at < n_bytes) {
        if (n_bytes - at < 28) return -1;
        const uint8_t* h = data + at;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return -1;
        const uint32_t xlen = le16(h + 10);
        // the BC subfield (SI1 = 66, SI2 = 67, SLEN = 2) holds BSIZE = total block size - 1
        int64_t bsize = -1;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t* sf = h + 12 + x;
            const uint32_t slen = le16(sf + 2);
            if (sf[0] == 66 && sf[1] == 67 && slen == 2) bsize = int64_t(le16(sf + 4)) + 1;
            x += 4 + slen;
        }
        if (bsize < 0 || at + bsize > n_bytes || bsize < int64_t(12 + xlen + 8)) return -1;
        const uint32_t isize = le32(h + bsize - 4);
        if (isize > 65536u) return -1;
        if (n < max_blocks) {
            block_off[n] = at;
            out_off[n] = out;
        }
        ++n;
        at += bsize;
        out += isize;
    }
    if (n <= max_blocks && max_blocks > 0) { // the closing offsets (arrays are [max_blocks + 1])
        block_off[n] = at;
        out_off[n] = out;
    }
    return n;
}

int64_t sk_bam_scan_records(const uint8_t* stream, int64_t stream_len, int64_t first, int64_t* rec_off, int64_t* read_off, int64_t* path_off,
                            int32_t max_records)
{
    if (!stream || stream_len < 0 || first < 0 || max_records < 0 || (max_records > 0 && (!rec_off || !read_off || !path_off))) return -1;
    int64_t at = first, bases = 0, segs = 0;
    int32_t n = 0;
    while (at + 4 <= stream_len) {
        const int64_t block_size = int64_t(le32(stream + at));
        if (block_size < 32 || at + 4 + block_size > stream_len) break; // (a record cut by the end of the stream is left to the next call)
        const uint8_t* p = stream + at;
        const uint32_t l_read_name = p[12], n_cigar = le16(p + 16);
        const int64_t l_seq = int64_t(int32_t(le32(p + 20)));
        if (l_seq < 0 || 32 + int64_t(l_read_name) + 4 * int64_t(n_cigar) + (l_seq + 1) / 2 + l_seq > block_size) return -1;
        if (n < max_records) {
            rec_off[n] = at;
            read_off[n] = bases;
            path_off[n] = segs;
        }
        ++n;
        bases += l_seq;
        segs += n_cigar;
        at += 4 + block_size;
    }
    if (n <= max_records && max_records > 0) {
        read_off[n] = bases; // (arrays are [max_records + 1])
        path_off[n] = segs;
    }
    return n;
}

int64_t sk_bam_header_end(const uint8_t* stream, int64_t stream_len)
{
    // magic "BAM\1", l_text, text, n_ref, then per reference l_name, name, l_ref
    if (!stream || stream_len < 12 || std::memcmp(stream, "BAM\1", 4) != 0) return -1;
    int64_t at = 8 + int64_t(le32(stream + 4));
    if (at + 4 > stream_len) return -1;
    const uint32_t n_ref = le32(stream + at);
    at += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (at + 4 > stream_len) return -1;
        at += 4 + int64_t(le32(stream + at)) + 4;
        if (at > stream_len) return -1;
    }
    return at;
}

} // extern "C"
