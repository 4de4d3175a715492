// bam_feed.cpp -- host side of the feed: finding the BGZF blocks of a file image and the BAM
// records of an inflated stream.  Both are chains of length fields (each block / record says where the next one starts), a few
// nanoseconds per element on one core; the bytes themselves are inflated and decoded on the device (csrc/bam_feed.hip).
// Formats: SAM specification v1, sections 4.1 (BGZF) and 4.2 (BAM); in the reference this is htslib (bgzf.c bgzf_read_block,
// sam.c bam_read1) behind L/htsapi/bam_streamer.cpp:268.

#include "strelka_amd.h"

#include <algorithm>
#include <cstring>
#include <vector>

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
        // the extra field and the 8-byte trailer must lie inside the buffer before a subfield is read (a header that claims
        // XLEN = 0xFFFF in a 28-byte buffer is malformed input, not a reason to read past the end)
        if (int64_t(12) + int64_t(xlen) + 8 > n_bytes - at) return -1;
        // the BC subfield (SI1 = 66, SI2 = 67, SLEN = 2) holds BSIZE = total block size - 1
        int64_t bsize = -1;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t* sf = h + 12 + x;
            const uint32_t slen = le16(sf + 2);
            if (x + 4 + slen > xlen) return -1; // a subfield that runs past XLEN
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
        if (block_size < 32) return -1; // shorter than its fixed fields: malformed (htslib's bam_read1 returns -4, the reference throws)
        if (at + 4 + block_size > stream_len) break; // (a record cut by the end of the stream is left to the next call)
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

// ---- the index: hts_itr_query over a .bai image (htslib hts.c:2066-2169; BAI = 14-bit smallest bins, 5 levels, SAM spec 5.2) ----
namespace
{

inline uint64_t le64(const uint8_t* p) { return uint64_t(le32(p)) | (uint64_t(le32(p + 4)) << 32); }

enum { BAI_MIN_SHIFT = 14, BAI_LEVELS = 5, BAI_N_BINS = ((1 << (3 * BAI_LEVELS + 3)) - 1) / 7 }; // 37 449; the metadata bin is + 1

struct BaiBin
{
    uint32_t bin;
    int32_t n_chunk;
    const uint8_t* chunks; // n_chunk x (begin, end) little-endian
};
inline int bin_first(const int level) { return ((1 << (3 * level)) - 1) / 7; }      // hts_bin_first
inline int bin_parent(const int bin) { return (bin - 1) >> 3; }                     // hts_bin_parent
inline int bin_bottom(const int bin)                                                // hts_bin_bot: its first smallest bin
{
    int level = 0;
    for (int b = bin; b; b = bin_parent(b)) ++level;
    return (bin - bin_first(level)) << ((BAI_LEVELS - level) * 3);
}

struct BaiRef // one reference of the index, as parsed in place
{
    std::vector<BaiBin> bins; // sorted by bin number
    int32_t n_intv = 0;
    const uint8_t* intv = nullptr;
    const BaiBin* find(const uint32_t bin) const
    {
        size_t lo = 0, hi = bins.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (bins[mid].bin < bin) lo = mid + 1; else hi = mid;
        }
        return (lo < bins.size() && bins[lo].bin == bin) ? &bins[lo] : nullptr;
    }
    // update_loff (hts.c:1379-1408): the linear index entry of the bin's first smallest bin; zeros in the file mean "as before"
    // (hts_idx_load_core :1784-1785); no linear index that far, or the metadata bin: 0
    uint64_t loff(const uint32_t bin) const
    {
        if (bin >= uint32_t(BAI_N_BINS)) return 0;
        int l = bin_bottom(int(bin));
        if (l >= n_intv) return 0;
        uint64_t v = le64(intv + 8 * size_t(l));
        while (v == 0 && l > 0) v = le64(intv + 8 * size_t(--l));
        return v;
    }
};

// 0 = parsed, 1 = malformed, 2 = ref_id beyond the index
int bai_reference(const uint8_t* bai, const int64_t len, const int32_t ref_id, BaiRef& out)
{
    if (!bai || len < 8 || std::memcmp(bai, "BAI\1", 4) != 0) return 1;
    const int32_t n_ref = int32_t(le32(bai + 4));
    if (ref_id < 0 || ref_id >= n_ref) return 2;
    int64_t at = 8;
    for (int32_t r = 0; r <= ref_id; ++r) {
        if (at + 4 > len) return 1;
        const int32_t n_bin = int32_t(le32(bai + at));
        at += 4;
        if (n_bin < 0) return 1;
        for (int32_t b = 0; b < n_bin; ++b) {
            if (at + 8 > len) return 1;
            BaiBin e;
            e.bin = le32(bai + at);
            e.n_chunk = int32_t(le32(bai + at + 4));
            at += 8;
            if (e.n_chunk < 0 || at + 16 * int64_t(e.n_chunk) > len) return 1;
            e.chunks = bai + at;
            at += 16 * int64_t(e.n_chunk);
            if (r == ref_id) out.bins.push_back(e);
        }
        if (at + 4 > len) return 1;
        const int32_t n_intv = int32_t(le32(bai + at));
        at += 4;
        if (n_intv < 0 || at + 8 * int64_t(n_intv) > len) return 1;
        if (r == ref_id) {
            out.n_intv = n_intv;
            out.intv = bai + at;
        }
        at += 8 * int64_t(n_intv);
    }
    std::sort(out.bins.begin(), out.bins.end(), [](const BaiBin& a, const BaiBin& b) { return a.bin < b.bin; });
    for (size_t i = 1; i < out.bins.size(); ++i)
        if (out.bins[i].bin == out.bins[i - 1].bin) return 1; // (hts_idx_load_core refuses duplicate bins)
    return 0;
}

} // namespace

extern "C" {

int32_t sk_bai_query(const uint8_t* bai, int64_t bai_len, int32_t ref_id, int32_t begin, int32_t end, sk_bai_chunk* out, int32_t max_chunks)
{
    BaiRef ref;
    const int rc = bai_reference(bai, bai_len, ref_id, ref);
    if (rc != 0) return -rc;
    if (begin < 0) begin = 0;
    if (end < begin || ref.bins.empty()) return 0;
    // min_off: the linear-index offset of the smallest bin holding `begin`, or of the nearest bin before it / above it that exists
    uint64_t min_off = 0;
    {
        int bin = bin_first(BAI_LEVELS) + (begin >> BAI_MIN_SHIFT);
        const BaiBin* k = nullptr;
        do {
            if ((k = ref.find(uint32_t(bin))) != nullptr) break;
            const int first = (bin_parent(bin) << 3) + 1;
            if (bin > first) --bin; else bin = bin_parent(bin);
        } while (bin);
        if (bin == 0) k = ref.find(0);
        min_off = k ? ref.loff(k->bin) : 0;
    }
    // max_off: where the first bin to the right of `end` that exists begins
    uint64_t max_off;
    {
        int bin = bin_first(BAI_LEVELS) + ((end - 1) >> BAI_MIN_SHIFT) + 1;
        if (bin >= BAI_N_BINS) bin = 0;
        for (;;) {
            while (bin % 8 == 1) bin = bin_parent(bin);
            if (bin == 0) {
                max_off = ~uint64_t(0);
                break;
            }
            const BaiBin* k = ref.find(uint32_t(bin));
            if (k && k->n_chunk > 0) {
                max_off = le64(k->chunks);
                break;
            }
            ++bin;
        }
    }
    // reg2bins (hts.c:1939-1955) and the chunks of those bins between the two offsets
    std::vector<sk_bai_chunk> off;
    if (begin < end) {
        int64_t e = end;
        int s = BAI_MIN_SHIFT + 3 * BAI_LEVELS;
        if (e >= (int64_t(1) << s)) e = int64_t(1) << s;
        --e;
        for (int l = 0, t = 0; l <= BAI_LEVELS; s -= 3, t += 1 << (3 * l), ++l)
            for (int b = t + int(begin >> s); b <= t + int(e >> s); ++b) {
                const BaiBin* k = ref.find(uint32_t(b));
                if (!k) continue;
                for (int32_t j = 0; j < k->n_chunk; ++j) {
                    const uint64_t u = le64(k->chunks + 16 * size_t(j)), v = le64(k->chunks + 16 * size_t(j) + 8);
                    if (v > min_off && u < max_off) off.push_back(sk_bai_chunk{ u, v });
                }
            }
    }
    if (off.empty()) return 0;
    // (ks_introsort by begin is not stable; chunks with equal begins differ only in their ends, and the first pass below keeps the
    // longest whatever their order)
    std::sort(off.begin(), off.end(), [](const sk_bai_chunk& a, const sk_bai_chunk& b) { return a.begin < b.begin || (a.begin == b.begin && a.end > b.end); });
    size_t l = 0;
    for (size_t i = 1; i < off.size(); ++i) // chunks contained in the previous one
        if (off[l].end < off[i].end) off[++l] = off[i];
    size_t n = l + 1;
    for (size_t i = 1; i < n; ++i) // overlaps between neighbours
        if (off[i - 1].end >= off[i].begin) off[i - 1].end = off[i].begin;
    l = 0;
    for (size_t i = 1; i < n; ++i) { // neighbours that meet in one BGZF block
        if ((off[l].end >> 16) == (off[i].begin >> 16)) off[l].end = off[i].end;
        else off[++l] = off[i];
    }
    n = l + 1;
    for (size_t i = 0; i < n && int32_t(i) < max_chunks; ++i) out[i] = off[i];
    return int32_t(n);
}

int32_t sk_bam_region_filter(const sk_bam_record* rec, const int64_t* path_off, const sk_path_seg* path, int32_t n_records, int32_t ref_id,
                             int32_t begin, int32_t end, uint8_t* keep)
{
    if (n_records < 0 || (n_records > 0 && (!rec || !path_off || !keep))) return -1;
    if (begin < 0) begin = 0;
    for (int32_t i = 0; i < n_records; ++i) keep[i] = 0;
    for (int32_t i = 0; i < n_records; ++i) {
        const sk_bam_record& r = rec[i];
        if (r.ref_id != ref_id || r.pos >= end) return i; // hts_itr_next :2623: "no need to proceed"
        int32_t rec_end = r.pos + 1;                       // bam_endpos, sam.c:359-365
        if (!(r.flag & 0x4) && r.n_cigar > 0) {
            int32_t rlen = 0;
            for (int64_t k = path_off[i]; k < path_off[i + 1]; ++k) {
                const uint16_t t = path[k].type;
                if (t == SK_SEG_MATCH || t == SK_SEG_DELETE || t == SK_SEG_SKIP || t == SK_SEG_SEQ_MATCH || t == SK_SEG_SEQ_MISMATCH)
                    rlen += int32_t(path[k].length);
            }
            rec_end = r.pos + rlen;
        }
        if (rec_end > begin && end > r.pos) keep[i] = 1;
    }
    return n_records;
}

} // extern "C"
