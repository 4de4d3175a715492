// sk_adapter_feed.cpp -- site 8: the reads of a region, from the file to bam_record, through the feed entry points of the C-ABI.
//
// bam_streamer::resetRegion (L/htsapi/bam_streamer.cpp:211-243) asks htslib for an iterator (sam_itr_queryi) and next() (:246-287)
// pulls one record at a time through it (sam_itr_next: bgzf seek, zlib inflate block by block, bam_read1).  Here resetRegion has the
// index answer the region (sk_bai_query), reads the byte ranges of the chunks, inflates all their BGZF blocks in one call
// (sk_bgzf_inflate: the kernels), finds and decodes the records (sk_bam_scan_records, sk_bam_decode) and applies the iterator's test
// (sk_bam_region_filter); next() then hands the reference its bam1_t one record after the other, filled from the record's bytes the
// way bam_read1 (htslib sam.c:435-500) fills it.  Everything after that -- bam_record's accessors, the read filters,
// normalizeBamRecordAlignment -- is the reference's, untouched.
//
// Falls back to the reference's own iterator (returns false from feed_reset_region) for anything that is not a BAM file with a .bai
// next to it, and with STRELKA_AMD_FEED=0.

#include "sk_adapter.hh"
#include "sk_adapter_access.hh"

#include "blt_util/align_path.hh"
#include "blt_util/blt_exception.hh"
#include "blt_util/reference_contig_segment.hh"
#include "starling_common/alignment.hh"

#include "strelka_amd.h"

extern "C" {
#include "htslib/hts.h"
#include "htslib/sam.h"
}

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <chrono>
#include <vector>

namespace sk_adapter
{

namespace
{

struct BamFile
{
    int fd = -1;
    int64_t size = 0;
    std::vector<uint8_t> bai;
    bool usable = false;
};

struct Feed
{
    // ---- the region as a stream: the reference's iterator holds one record at a time (hts_itr_next); this one holds one SLICE of
    // at most SLICE_BLOCKS BGZF blocks -- inflated, scanned, decoded, filtered and normalised in one call each -- and refills from
    // next() when the slice is used up, carrying a record the slice end cut over to the next slice
    BamFile* file = nullptr;
    std::string name;
    int tid = 0, begin = 0, end = 0;
    std::vector<sk_bai_chunk> chunks;
    size_t chunk = 0;          // the chunk being read
    bool chunk_open = false;
    int64_t file_at = 0;       // file offset of the next BGZF block of the chunk
    int64_t first_offset = 0;  // where the first record starts in the next slice's stream (the chunk's in-block offset, then 0)
    int64_t limit = -1;        // the chunk's end in the next slice's stream, once the block holding it has been read; -1: not yet
    std::vector<uint8_t> carry; // the bytes of the record the last slice ended in
    bool finished = false;     // the iterator's end: a record past the region was seen, or every chunk is used up
    // ---- the current slice
    std::vector<uint8_t> bytes;  // its records, raw (block_size field included), one after the other
    std::vector<size_t> rec_at;
    size_t next = 0;
    // what sk_bam_decode made of them: bases (BAM codes, one per byte), CIGAR as path segments, position
    std::vector<int64_t> read_off{ 0 }, path_off{ 0 };
    std::vector<uint8_t> code;
    std::vector<sk_path_seg> path;
    std::vector<int32_t> pos;
    std::vector<uint8_t> mapped;
    // normalizeAlignment of all of them at once (feed_normalize_current), made when the first record asks
    bool normalized = false;
    std::vector<int32_t> in_batch; // record -> its place in the batch, -1: not in it
    std::vector<int64_t> n_path_off;
    std::vector<sk_path_seg> n_path_in, n_path;
    std::vector<int32_t> n_pos_in, n_pos, n_seg_in, n_seg;
    std::vector<uint8_t> n_changed;
};

struct FeedState
{
    std::map<std::string, BamFile> files;
    std::map<const void*, Feed> feeds;
    unsigned long regions = 0, records = 0, blocks = 0, inflated = 0, norm_batches = 0, norm_reads = 0, norm_changed = 0, norm_declined = 0;
    double t_refill = 0, t_abi = 0; // wall seconds in feed_refill / feed_normalize_current's batch, of which inside the C-ABI
    ~FeedState()
    {
        if (std::getenv("STRELKA_AMD_VERBOSE") && std::atoi(std::getenv("STRELKA_AMD_VERBOSE")) != 0)
            std::cerr << "strelka_amd adapter feed: regions=" << regions << " records=" << records << " bgzf_blocks=" << blocks
                      << " inflated_bytes=" << inflated << " normalize_batches=" << norm_batches << " normalized=" << norm_reads
                      << " normalize_changed=" << norm_changed << " normalize_declined=" << norm_declined << " seconds=" << t_refill
                      << " abi_seconds=" << t_abi << "\n";
        for (auto& f : files)
            if (f.second.fd >= 0) ::close(f.second.fd);
    }
};
FeedState& fs()
{
    static FeedState s;
    return s;
}

bool read_file(const std::string& path, std::vector<uint8_t>& out)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(size_t(n));
    const bool ok = (n == 0) || std::fread(out.data(), 1, size_t(n), f) == size_t(n);
    std::fclose(f);
    return ok;
}

BamFile& bam_file(const std::string& name)
{
    auto it = fs().files.find(name);
    if (it != fs().files.end()) return it->second;
    BamFile& b = fs().files[name];
    if (name.size() < 4 || name.compare(name.size() - 4, 4, ".bam") != 0) return b; // (CRAM and the rest: the reference's own path)
    if (!read_file(name + ".bai", b.bai) && !read_file(name.substr(0, name.size() - 4) + ".bai", b.bai)) return b;
    b.fd = ::open(name.c_str(), O_RDONLY);
    struct stat st;
    if (b.fd < 0 || ::fstat(b.fd, &st) != 0) return b;
    b.size = int64_t(st.st_size);
    b.usable = true;
    return b;
}

void fail(const std::string& what, const char* name)
{
    std::ostringstream oss;
    oss << "strelka_amd feed: " << what << " (file '" << name << "')";
    throw blt_exception(oss.str().c_str());
}

inline uint32_t le32(const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }

} // namespace

// BGZF blocks per slice (at most 32 MiB of inflated bytes in flight per stream) and bytes read from the file at a time (the blocks of
// a BAM are ~15-25 kB: ~400 of them); $STRELKA_AMD_FEED_SLICE_BLOCKS makes slices small enough for the tests' inputs to need many
static int32_t slice_blocks()
{
    static const int32_t v([]() { const char* e(std::getenv("STRELKA_AMD_FEED_SLICE_BLOCKS")); return (e && std::atoi(e) > 0) ? std::atoi(e) : 512; }());
    return v;
}
#define SLICE_BLOCKS slice_blocks()
static int64_t slice_raw_bytes() { return std::max<int64_t>(int64_t(2) * 65536, std::min<int64_t>(int64_t(8) << 20, int64_t(slice_blocks()) * 65536)); }
#define SLICE_RAW_BYTES slice_raw_bytes()

/// the next slice of the region into `feed` (records, decoded fields); false: the region is used up
/// a grow-only buffer of page-locked host memory (sk_host_alloc): what crosses the C-ABI again and again
template <typename T>
struct Pinned
{
    T* p = nullptr;
    size_t cap = 0, n = 0;
    T* data() { return p; }
    size_t size() const { return n; }
    void resize(const size_t count) // (contents are not kept across a growth: every user refills)
    {
        if (count > cap) {
            if (p) sk_host_free(p);
            cap = count + count / 2 + 64;
            p = static_cast<T*>(sk_host_alloc(cap * sizeof(T)));
            if (p == nullptr) throw blt_exception((std::string("strelka_amd feed: sk_host_alloc: ") + sk_last_error()).c_str());
        }
        n = count;
    }
    T& operator[](const size_t i) { return p[i]; }
};

struct FeedTimer
{
    double& acc;
    std::chrono::steady_clock::time_point t0;
    explicit FeedTimer(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~FeedTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

bool feed_refill(Feed& feed)
{
    FeedTimer whole(fs().t_refill);
    static Pinned<uint8_t> raw, stream, code, qual;
    static Pinned<sk_bam_record> rec;
    static Pinned<sk_path_seg> path;
    static std::vector<uint8_t> keep;
    static std::vector<int64_t> block_off, out_off, rec_off, read_off, path_off;
    BamFile& bf(*feed.file);
    const char* name(feed.name.c_str());
    feed.bytes.clear();
    feed.rec_at.clear();
    feed.next = 0;
    feed.read_off.assign(1, 0);
    feed.path_off.assign(1, 0);
    feed.code.clear();
    feed.path.clear();
    feed.pos.clear();
    feed.mapped.clear();
    feed.normalized = false;
    feed.n_path_in.clear();
    feed.n_seg_in.clear();
    feed.n_pos_in.clear();
    while (feed.rec_at.empty() && !feed.finished) {
        if (!feed.chunk_open) {
            if (feed.chunk >= feed.chunks.size()) {
                feed.finished = true;
                break;
            }
            feed.file_at = int64_t(feed.chunks[feed.chunk].begin >> 16);
            feed.first_offset = int64_t(feed.chunks[feed.chunk].begin & 0xffff);
            feed.limit = -1;
            feed.carry.clear();
            feed.chunk_open = true;
        }
        const int64_t ce = int64_t(feed.chunks[feed.chunk].end >> 16), ue = int64_t(feed.chunks[feed.chunk].end & 0xffff);
        // the slice's blocks: whole blocks from file_at on, SLICE_BLOCKS at most; once the chunk's last block is among them, only the
        // few more a record cut by it can reach into
        const int64_t want = std::min<int64_t>(bf.size - feed.file_at, SLICE_RAW_BYTES);
        bool chunk_done = false;
        if (want <= 0) {
            chunk_done = true;
        } else {
            raw.resize(size_t(want));
            if (::pread(bf.fd, raw.data(), raw.size(), off_t(feed.file_at)) != ssize_t(raw.size())) fail("short read", name);
            block_off.assign(1, 0);
            out_off.assign(1, 0);
            int64_t at = 0;
            int32_t past_end = 0;
            while (int32_t(block_off.size()) - 1 < SLICE_BLOCKS && at + 18 <= int64_t(raw.size())) {
                if (raw[size_t(at)] != 31 || raw[size_t(at) + 1] != 139) fail("not a BGZF block where the index points", name);
                int64_t bsize = -1;
                const int64_t xlen = raw[size_t(at) + 10] | (raw[size_t(at) + 11] << 8);
                if (at + 12 + xlen > int64_t(raw.size())) break; // (the header itself is cut: not a whole block)
                for (int64_t x = 0; x + 4 <= xlen;) {
                    const uint8_t* sf = raw.data() + at + 12 + x;
                    const int64_t slen = sf[2] | (sf[3] << 8);
                    if (x + 4 + slen > xlen) fail("malformed BGZF extra field", name);
                    if (sf[0] == 66 && sf[1] == 67 && slen == 2) bsize = (sf[4] | (sf[5] << 8)) + 1;
                    x += 4 + slen;
                }
                if (bsize < 0) fail("BGZF block without its BC subfield", name);
                if (bsize < 12 + xlen + 8) fail("BGZF block shorter than its header and trailer", name); // (before the trailer is read)
                if (at + bsize > int64_t(raw.size())) break;
                if (feed.file_at + at > ce && ++past_end > 2) break;
                out_off.push_back(out_off.back() + int64_t(le32(raw.data() + at + bsize - 4)));
                at += bsize;
                block_off.push_back(at);
            }
            const int32_t nb = int32_t(block_off.size()) - 1;
            if (nb == 0) {
                chunk_done = true; // (the file ends in a cut block, or the chunk's end was passed: nothing more to read for it)
            } else {
                const int64_t carried = int64_t(feed.carry.size());
                stream.resize(size_t(carried + out_off.back()) + 8);
                {
                    FeedTimer abi(fs().t_abi);
                    if (sk_bgzf_inflate_prefixed(raw.data(), block_off.data(), out_off.data(), nb, feed.carry.data(), carried, stream.data()))
                        fail(std::string("inflate: ") + sk_last_error(), name);
                }
                fs().blocks += unsigned(nb);
                fs().inflated += (unsigned long)out_off.back();
                const int64_t stream_len = carried + out_off.back();
                if (feed.limit < 0)
                    for (int32_t b = 0; b <= nb; ++b)
                        if (feed.file_at + block_off[size_t(b)] == ce) feed.limit = carried + out_off[size_t(b)] + ue;
                const bool at_eof = (feed.file_at + block_off[size_t(nb)] >= bf.size);
                const int64_t start = carried ? 0 : feed.first_offset;
                int64_t n_rec = sk_bam_scan_records(stream.data(), stream_len, start, nullptr, nullptr, nullptr, 0);
                if (n_rec < 0) fail("malformed BAM record", name);
                rec_off.assign(size_t(n_rec) + 1, 0);
                read_off.assign(size_t(n_rec) + 1, 0);
                path_off.assign(size_t(n_rec) + 1, 0);
                if (n_rec > 0 && sk_bam_scan_records(stream.data(), stream_len, start, rec_off.data(), read_off.data(), path_off.data(), int32_t(n_rec)) != n_rec)
                    fail("record scan", name);
                int64_t next_at = start;
                if (n_rec > 0) next_at = rec_off[size_t(n_rec) - 1] + 4 + int64_t(le32(stream.data() + rec_off[size_t(n_rec) - 1]));
                // the records of this chunk among them: those that start before its end
                int32_t n_in = int32_t(n_rec);
                if (feed.limit >= 0) {
                    n_in = 0;
                    while (n_in < n_rec && rec_off[size_t(n_in)] < feed.limit) ++n_in;
                }
                if (n_in > 0) {
                    rec.resize(size_t(n_in));
                    code.resize(size_t(read_off[size_t(n_in)]) + 1);
                    qual.resize(size_t(read_off[size_t(n_in)]) + 1);
                    path.resize(size_t(path_off[size_t(n_in)]) + 1);
                    keep.resize(size_t(n_in));
                    {
                        FeedTimer abi(fs().t_abi);
                        if (sk_bam_decode_kept(stream.data(), stream_len, rec_off.data(), n_in, read_off.data(), path_off.data(), rec.data(), code.data(),
                                               qual.data(), path.data()))
                            fail(std::string("decode: ") + sk_last_error(), name);
                    }
                    const int32_t n_read = sk_bam_region_filter(rec.data(), path_off.data(), path.data(), n_in, feed.tid, feed.begin, feed.end, keep.data());
                    if (n_read < 0) fail("region filter", name);
                    for (int32_t i = 0; i < n_in; ++i)
                        if (keep[size_t(i)]) {
                            const uint8_t* r = stream.data() + rec_off[size_t(i)];
                            const size_t len = 4 + size_t(le32(r));
                            feed.rec_at.push_back(feed.bytes.size());
                            feed.bytes.insert(feed.bytes.end(), r, r + len);
                            feed.code.insert(feed.code.end(), code.data() + read_off[size_t(i)], code.data() + read_off[size_t(i) + 1]);
                            feed.read_off.push_back(int64_t(feed.code.size()));
                            feed.path.insert(feed.path.end(), path.data() + path_off[size_t(i)], path.data() + path_off[size_t(i) + 1]);
                            feed.path_off.push_back(int64_t(feed.path.size()));
                            feed.pos.push_back(rec[size_t(i)].pos);
                            feed.mapped.push_back((rec[size_t(i)].flag & 0x4) ? 0 : 1);
                        }
                    if (n_read < n_in) feed.finished = true; // a record past the region's end: hts_itr_next stops there
                }
                // is the chunk used up?  Its end is known and the next record starts at or past it; or the file is
                if ((feed.limit >= 0 && (next_at >= feed.limit || n_in < n_rec)) || at_eof) {
                    chunk_done = true;
                } else {
                    // the record the slice ended in (if any) opens the next slice's stream
                    feed.carry.assign(stream.data() + next_at, stream.data() + stream_len);
                    if (feed.limit >= 0) feed.limit -= next_at;
                    feed.first_offset = 0;
                    feed.file_at += block_off[size_t(nb)];
                    if (next_at > stream_len) fail("record scan ran past the slice", name);
                }
            }
        }
        if (chunk_done) {
            feed.chunk_open = false;
            ++feed.chunk;
        }
    }
    fs().records += feed.rec_at.size();
    return !feed.rec_at.empty();
}

bool feed_reset_region(const void* streamer, const char* name, const int tid, const int begin, const int end)
{
    fs().feeds.erase(streamer);
    if (const char* e = std::getenv("STRELKA_AMD_FEED"))
        if (std::atoi(e) == 0) return false;
    BamFile& bf = bam_file(name);
    if (!bf.usable) return false;
    init();
    Feed feed;
    const int32_t n_chunks = sk_bai_query(bf.bai.data(), int64_t(bf.bai.size()), tid, begin, end, nullptr, 0);
    if (n_chunks == -2) return false; // (a reference the index does not know: htslib's iterator decides what that means)
    if (n_chunks < 0) fail("malformed .bai", name);
    feed.chunks.resize(size_t(n_chunks) + 1);
    if (n_chunks > 0 && sk_bai_query(bf.bai.data(), int64_t(bf.bai.size()), tid, begin, end, feed.chunks.data(), n_chunks) != n_chunks) fail("index query", name);
    feed.chunks.resize(size_t(n_chunks));
    feed.file = &bf;
    feed.name = name;
    feed.tid = tid;
    feed.begin = begin;
    feed.end = end;
    fs().regions++;
    fs().feeds[streamer] = std::move(feed);
    return true;
}

bool feed_active(const void* streamer) { return fs().feeds.count(streamer) != 0; }

void feed_drop(const void* streamer) { fs().feeds.erase(streamer); }

// bam_read1 (htslib sam.c:435-500) from the record's bytes; >= 0: a record, -1: none left, -4: a record htslib refuses
int feed_next(const void* streamer, void* bam1)
{
    Feed& f = fs().feeds[streamer];
    if (f.next >= f.rec_at.size() && (f.finished || !feed_refill(f))) return -1;
    const uint8_t* r = f.bytes.data() + f.rec_at[f.next++];
    bam1_t* b = static_cast<bam1_t*>(bam1);
    bam1_core_t* c = &b->core;
    const int32_t block_len = int32_t(le32(r));
    if (block_len < 32) return -4;
    uint32_t x[8];
    for (int i = 0; i < 8; ++i) x[i] = le32(r + 4 + 4 * i);
    c->tid = int32_t(x[0]);
    c->pos = int32_t(x[1]);
    c->bin = uint16_t(x[2] >> 16);
    c->qual = uint8_t(x[2] >> 8 & 0xff);
    c->l_qname = uint8_t(x[2] & 0xff);
    c->l_extranul = (c->l_qname % 4 != 0) ? uint8_t(4 - c->l_qname % 4) : 0;
    if (uint32_t(c->l_qname) + c->l_extranul > 255) return -4;
    c->flag = uint16_t(x[3] >> 16);
    c->n_cigar = x[3] & 0xffff;
    c->l_qseq = int32_t(x[4]);
    c->mtid = int32_t(x[5]);
    c->mpos = int32_t(x[6]);
    c->isize = int32_t(x[7]);
    b->l_data = block_len - 32 + c->l_extranul;
    if (b->l_data < 0 || c->l_qseq < 0 || c->l_qname < 1) return -4;
    if ((uint64_t(c->n_cigar) << 2) + c->l_qname + c->l_extranul + ((uint64_t(c->l_qseq) + 1) >> 1) + uint64_t(c->l_qseq) > uint64_t(b->l_data)) return -4;
    if (b->m_data < b->l_data) {
        uint32_t new_m = uint32_t(b->l_data);
        kroundup32(new_m);
        uint8_t* new_data = static_cast<uint8_t*>(std::realloc(b->data, new_m));
        if (!new_data) return -4;
        b->data = new_data;
        b->m_data = new_m;
    }
    const uint8_t* body = r + 36;
    std::memcpy(b->data, body, c->l_qname);
    for (int i = 0; i < c->l_extranul; ++i) b->data[c->l_qname + i] = '\0';
    const int l_qname_file = c->l_qname;
    c->l_qname = uint8_t(c->l_qname + c->l_extranul);
    if (b->l_data < c->l_qname) return -4;
    std::memcpy(b->data + c->l_qname, body + l_qname_file, size_t(b->l_data - c->l_qname));
    // bam_tag2cigar (:367-432): a placeholder CIGAR with the real one in a CG:B,I tag (more than 65535 operations) is not handled here
    if (c->n_cigar > 0 && c->tid >= 0 && c->pos >= 0) {
        const uint32_t* cigar0 = bam_get_cigar(b);
        if (bam_cigar_op(cigar0[0]) == BAM_CSOFT_CLIP && int32_t(bam_cigar_oplen(cigar0[0])) == c->l_qseq && bam_aux_get(b, "CG") != nullptr) {
            std::cerr << "strelka_amd feed: record " << bam_get_qname(b) << " keeps its CIGAR in a CG tag; run with STRELKA_AMD_FEED=0\n";
            return -4;
        }
    }
    if (c->n_cigar > 0) { // :484-495: "bin" recomputed, CIGAR against the query length
        int rlen = bam_cigar2rlen(int(c->n_cigar), bam_get_cigar(b));
        const int qlen = bam_cigar2qlen(int(c->n_cigar), bam_get_cigar(b));
        if (c->flag & BAM_FUNMAP) rlen = 1;
        c->bin = uint16_t(hts_reg2bin(c->pos, c->pos + rlen, 14, 5));
        if (c->l_qseq > 0 && !(c->flag & BAM_FUNMAP) && qlen != c->l_qseq) {
            std::cerr << "strelka_amd feed: CIGAR and query sequence lengths differ for " << bam_get_qname(b) << "\n";
            return -4;
        }
    }
    return 4 + block_len;
}


// normalizeAlignment (L/starling_common/normalizeAlignment.cpp:647-703) at its call site in processInputReadAlignment
// (starling_pos_processor_util.cpp:432), for the record the streamer is at: the first call normalises ALL records of the region in one
// sk_normalize_alignments call (kernel B4) against the region's reference segment -- from the decoded bases and CIGARs the feed kept,
// each path cleaned by the reference's own apath_cleaner first, as the call site does -- and every call then hands its record's result
// over.  Declines (the caller runs the reference's function) when the streamer has no feed, with STRELKA_AMD_FEED_NORMALIZE=0, and
// when `al` is not the alignment the feed holds for the record.
bool feed_normalize_current(const void* streamer, const reference_contig_segment& ref, alignment& al)
{
    auto it = fs().feeds.find(streamer);
    if (it == fs().feeds.end()) return false;
    Feed& f = it->second;
    if (f.next == 0 || f.next > f.rec_at.size()) return false;
    static const bool enabled = !(std::getenv("STRELKA_AMD_FEED_NORMALIZE") && std::atoi(std::getenv("STRELKA_AMD_FEED_NORMALIZE")) == 0);
    if (!enabled) return false;
    const size_t n_rec = f.rec_at.size();
    if (!f.normalized) {
        f.normalized = true;
        f.in_batch.assign(n_rec, -1);
        std::vector<int64_t> b_read_off{ 0 };
        static Pinned<uint8_t> b_code; // (page-locked: the largest array sk_normalize_alignments uploads)
        b_code.resize(f.code.size() + 1);
        size_t b_code_n(0);
        f.n_path_off.assign(1, 0);
        ALIGNPATH::path_t apath;
        for (size_t r = 0; r < n_rec; ++r) {
            if (!f.mapped[r] || f.path_off[r + 1] == f.path_off[r]) continue;
            apath.clear();
            for (int64_t k = f.path_off[r]; k < f.path_off[r + 1]; ++k)
                apath.push_back(ALIGNPATH::path_segment(static_cast<ALIGNPATH::align_t>(f.path[size_t(k)].type), f.path[size_t(k)].length));
            ALIGNPATH::apath_cleaner(apath);
            if (apath.empty()) continue;
            f.in_batch[r] = int32_t(f.n_pos_in.size());
            for (const auto& ps : apath) f.n_path_in.push_back(sk_path_seg{ uint32_t(ps.type), ps.length });
            f.n_path_off.push_back(int64_t(f.n_path_in.size()));
            f.n_seg_in.push_back(int32_t(apath.size()));
            f.n_pos_in.push_back(f.pos[r]);
            std::memcpy(b_code.data() + b_code_n, f.code.data() + f.read_off[r], size_t(f.read_off[r + 1] - f.read_off[r]));
            b_code_n += size_t(f.read_off[r + 1] - f.read_off[r]);
            b_read_off.push_back(int64_t(b_code_n));
        }
        const int32_t nb = int32_t(f.n_pos_in.size());
        f.n_path = f.n_path_in;
        f.n_seg = f.n_seg_in;
        f.n_pos = f.n_pos_in;
        f.n_changed.assign(size_t(nb) + 1, 0);
        if (nb > 0) {
            const std::string& seq = ref.seq();
            FeedTimer whole(fs().t_refill);
            FeedTimer abi(fs().t_abi);
            if (sk_normalize_alignments(seq.data(), int32_t(ref.get_offset()), int32_t(seq.size()), nb, b_read_off.data(), b_code.data(), f.n_path_off.data(),
                                        f.n_seg.data(), f.n_path.data(), f.n_pos.data(), f.n_changed.data())) {
                std::ostringstream oss;
                oss << "strelka_amd feed: sk_normalize_alignments: " << sk_last_error();
                throw blt_exception(oss.str().c_str());
            }
        }
        fs().norm_batches++;
        fs().norm_reads += (unsigned long)nb;
    }
    const size_t r = f.next - 1;
    const int32_t b = f.in_batch[r];
    bool same = (b >= 0) && al.pos == f.n_pos_in[size_t(b)] && int64_t(al.path.size()) == f.n_path_off[size_t(b) + 1] - f.n_path_off[size_t(b)];
    for (size_t k = 0; same && k < al.path.size(); ++k) {
        const sk_path_seg& q = f.n_path_in[size_t(f.n_path_off[size_t(b)]) + k];
        same = (uint32_t(al.path[k].type) == q.type && al.path[k].length == q.length);
    }
    if (!same) {
        fs().norm_declined++;
        return false;
    }
    if (f.n_changed[size_t(b)]) {
        fs().norm_changed++;
        al.pos = f.n_pos[size_t(b)];
        al.path.clear();
        const sk_path_seg* q = f.n_path.data() + f.n_path_off[size_t(b)];
        for (int32_t k = 0; k < f.n_seg[size_t(b)]; ++k) al.path.push_back(ALIGNPATH::path_segment(static_cast<ALIGNPATH::align_t>(q[k].type), q[k].length));
    }
    return true;
}

} // namespace sk_adapter
