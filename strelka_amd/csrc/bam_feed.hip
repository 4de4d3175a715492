// bam_feed.hip -- SURVEY.md section 8f rank 4, the feed: BGZF block inflation and BAM record decoding on the device.
//
// In the reference this work is a third-party dependency: L/htsapi/bam_streamer.cpp:268 calls htslib's sam_itr_next
// (redist/htslib-1.7-6-g6d2bfb7), which inflates BGZF blocks with zlib (bgzf.c:441-493: raw DEFLATE, window 15, then CRC-32 and
// ISIZE of the block trailer) and parses BAM records (sam.c bam_read1).  The algorithms restated here are the published ones:
// DEFLATE (RFC 1951), the BGZF container and the BAM record layout (SAM specification v1, sections 4.1 and 4.2).  Parity is
// anchored on zlib / samtools output for the reference's own demo BAMs and synthetic ones (tests/test_bam_feed.py).
//
//   B1  bgzf_inflate_kernel   one THREAD per BGZF block (blocks are independent DEFLATE streams of at most 64 KiB of output, a BAM
//                             of a few GB is 10^5 blocks): canonical-Huffman decoding (per-length counts in registers, symbols in a
//                             private array), LZ77 copies inside the thread's own output range in unaligned 8/4/2/1-byte pieces.
//                             A serial bit stream per block is what the format is; the parallelism is across blocks.
//   B1s bgzf_inflate_scalar_kernel  one WAVE per block, output assembled in LDS, the symbol loop on the scalar unit with the input and
//                             the first-level tables in vector registers: a twelfth of B1's latency per block, for the few hundred
//                             blocks a caller process inflates at a time
//   B2  bgzf_crc32_kernel     one thread per block: CRC-32 (IEEE 802.3, table in LDS) of the inflated bytes against the trailer
//   B4  normalize_kernel      one thread per read: normalizeAlignment (L/starling_common/normalizeAlignment.cpp:647-703; the
//                             reference applies it to every read as it comes off the BAM stream), csrc/normalize_core.h, in place
//   B3  bam_decode_kernel     one thread per BAM record: fixed fields, CIGAR -> path segments (ALIGNPATH::align_t = BAM op + 1),
//                             4-bit packed bases -> one BAM code per byte (what sk_read_input.read_code takes), qualities
//
// Byte work, bound by the serial decode per block rather than by HBM; algorithmic bytes = compressed in + inflated out.
//
// PROVENANCE AND LICENCE.  Nothing here comes from /root/reference's own sources.  The canonical-Huffman construction and decode
// of the thread-per-block kernel (huff_construct, huff_decode, the LEN_* / DIST_* base and extra-bit tables, the order of the
// code-length code lengths) follow zlib's contrib/puff/puff.c by Mark Adler (construct(), decode(), lens / lext / dists / dext,
// order[]), down to its variable names; zlib licence: "Copyright (C) 2002-2013 Mark Adler ... This software is provided 'as-is',
// without any express or implied warranty ... Permission is granted to anyone to use this software for any purpose, including
// commercial applications, and to alter it and redistribute it freely, subject to the following restrictions: 1. The origin of this
// software must not be misrepresented ... 2. Altered source versions must be plainly marked as such ... 3. This notice may not be
// removed or altered from any source distribution."  This is an altered version (HIP, per-block status codes, no setjmp).  The tables
// themselves are RFC 1951's.  is_multmodp / the x^(8n) exponentiation of the wave kernel follow zlib's crc32.c (multmodp, x2nmodp;
// same licence).  The wave-per-block decoder (first-level tables in registers, scalar bit buffer, lane-parallel table construction and
// copies) is original.

#include "sk_common.h"

#include "normalize_core.h"

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace
{

enum { INF_OK = 0, INF_BAD_BLOCK_TYPE = 1, INF_BAD_STORED = 2, INF_BAD_CODE = 3, INF_BAD_DISTANCE = 4, INF_OUT_OVERFLOW = 5,
       INF_IN_OVERRUN = 6, INF_BAD_LENGTHS = 7, INF_SIZE_MISMATCH = 8, INF_BAD_HEADER = 9, INF_CRC_MISMATCH = 10 };

// byte-addressed data at any alignment in one memory instruction (the global address space takes unaligned dword accesses; a packed
// struct tells the compiler so).  What a lane of the thread-per-block kernel costs is the NUMBER of its memory instructions -- its
// 64 lanes are 64 different streams, every access is 64 separate cache lines -- so bytes are moved 8 / 4 / 2 / 1 at a time.
struct __attribute__((packed)) U128u { uint32_t a, b, c, d; };
struct __attribute__((packed)) U64u { uint64_t v; };
struct __attribute__((packed)) U32u { uint32_t v; };
struct __attribute__((packed)) U16u { uint16_t v; };
__device__ __forceinline__ uint64_t ld_u64(const uint8_t* p) { return reinterpret_cast<const U64u*>(p)->v; }
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { return reinterpret_cast<const U32u*>(p)->v; }
__device__ __forceinline__ uint16_t ld_u16(const uint8_t* p) { return reinterpret_cast<const U16u*>(p)->v; }
__device__ __forceinline__ void st_u64(uint8_t* p, const uint64_t v) { reinterpret_cast<U64u*>(p)->v = v; }
__device__ __forceinline__ void st_u32(uint8_t* p, const uint32_t v) { reinterpret_cast<U32u*>(p)->v = v; }
__device__ __forceinline__ void st_u16(uint8_t* p, const uint16_t v) { reinterpret_cast<U16u*>(p)->v = v; }

// up to 16 bytes from src into two registers / out of them to dst: 8 + 4 + 2 + 1 pieces, at most four instructions each way
__device__ __forceinline__ void ld_upto16(const uint8_t* src, const int k, uint64_t& lo, uint64_t& hi)
{
    lo = 0;
    hi = 0;
    if (k >= 16) {
        lo = ld_u64(src);
        hi = ld_u64(src + 8);
        return;
    }
    int at = 0;
    if (k & 8) {
        lo = ld_u64(src);
        at = 8;
    }
    uint64_t rest = 0;
    int sh = 0;
    if (k & 4) {
        rest |= uint64_t(ld_u32(src + at)) << sh;
        at += 4;
        sh += 32;
    }
    if (k & 2) {
        rest |= uint64_t(ld_u16(src + at)) << sh;
        at += 2;
        sh += 16;
    }
    if (k & 1) rest |= uint64_t(src[at]) << sh;
    if (k & 8) hi = rest;
    else lo = rest;
}
__device__ __forceinline__ void st_upto16(uint8_t* dst, const int k, const uint64_t lo, const uint64_t hi)
{
    if (k >= 16) {
        st_u64(dst, lo);
        st_u64(dst + 8, hi);
        return;
    }
    int at = 0;
    uint64_t rest = lo;
    if (k & 8) {
        st_u64(dst, lo);
        at = 8;
        rest = hi;
    }
    if (k & 4) {
        st_u32(dst + at, uint32_t(rest));
        at += 4;
        rest >>= 32;
    }
    if (k & 2) {
        st_u16(dst + at, uint16_t(rest));
        at += 2;
        rest >>= 16;
    }
    if (k & 1) dst[at] = uint8_t(rest);
}

struct BitReader
{
    const uint8_t* p;
    const uint8_t* end;
    uint64_t buf;
    int cnt;
    bool overrun;
    // Sixteen input bytes at a time in registers.  A lane's input line does not survive in the L2 between two of its refills (1.3e5
    // lanes each walk their own line), so every 4-byte refill from memory fetched a whole line again: 24 GB of fetches per 1.3e5
    // blocks whose compressed bytes are 1.4 GB (profiles/r03_v24_pmc_traffic.json).  One unaligned 16-byte load feeds four refills.
    uint32_t res[4];
    const uint8_t* res_at; // the address res[] was loaded from, or nullptr
    __device__ uint32_t next_dword()
    {
        if (!(res_at && p >= res_at && p + 4 <= res_at + 16 && ((p - res_at) & 3) == 0)) {
            if (p + 16 <= end) {
                const U128u v = *reinterpret_cast<const U128u*>(p);
                res[0] = v.a; res[1] = v.b; res[2] = v.c; res[3] = v.d;
                res_at = p;
            } else {
                res_at = nullptr;
                return ld_u32(p);
            }
        }
        const int j = int(p - res_at) >> 2;
        return (j == 0) ? res[0] : (j == 1) ? res[1] : (j == 2) ? res[2] : res[3];
    }
    // top the buffer up to at least 25 bits where the input has them (four independent byte loads and one wait when there is room
    // for four bytes); bits past the end of the input read as 0 and are only an error when a code or a field consumes them
    __device__ void refill()
    {
        if (cnt <= 32 && p + 4 <= end) {
            buf |= uint64_t(next_dword()) << cnt;
            cnt += 32;
            p += 4;
            return;
        }
        while (cnt <= 24 && p < end) {
            buf |= uint64_t(*p++) << cnt;
            cnt += 8;
        }
    }
    __device__ void consume(const int n)
    {
        if (n > cnt) overrun = true;
        buf >>= n;
        cnt = (n > cnt) ? 0 : cnt - n;
    }
    __device__ unsigned bits(const int n) // n <= 16
    {
        if (n == 0) return 0;
        if (cnt < n) refill();
        const unsigned v = unsigned(buf & ((1ull << n) - 1ull));
        consume(n);
        return v;
    }
    // the bytes of a stored block follow the bit buffer's byte boundary: hand the whole bytes still buffered back
    __device__ void align_to_byte()
    {
        cnt -= (cnt & 7);
        p -= cnt / 8;
        buf = 0;
        cnt = 0;
    }
};

// canonical Huffman code (RFC 1951 3.2.2): count[len] codes of each length, symbols ordered by (length, value).
//
// What a lane of the thread-per-block kernel costs (round 3, profiles/r03_v16_inflate_kernels.txt; 510 blocks = one block's
// latency, 1.3e5 blocks = the bench's launch).  Its 64 lanes decode 64 unrelated streams, so EVERY memory instruction is a
// wave-wide scatter of 64 cache lines, and the lane's time is the number of such instructions it issues:
//   * the code tables in LDS instead of private arrays: same latency per block, a third of the blocks in flight (156 ms vs 89 ms);
//   * 4 byte loads in flight per refill, copies as 16 byte loads then 16 byte stores, literals stored 8 at a time: 89 -> 76 ms
//     (fewer waits, the same number of instructions);
//   * the per-length counts in registers and the decode loop unrolled over a 16-bit peek: no change (the nine scratch loads per
//     symbol it removed were hits in the lane's own line);
//   * the same bytes moved with ONE unaligned 8 / 4 / 2 / 1 byte instruction per piece (refill = one dword load, a 15-byte match
//     = 4 loads + 4 stores instead of 15 + 15, pending literals = at most 4 stores): 76 -> 63 ms, 46 -> 33 ms per block;
//   * length / distance bases in closed form instead of four table loads per match: 63 -> 61 ms.
// The sixteen per-length counts (and the running offsets while a table is built) are packed 16 bits each into four 64-bit
// registers, and a code is decoded from a 16-bit peek of the bit buffer by a fully unrolled loop over the lengths.
struct Packed16
{
    uint64_t w[4];
    __device__ void clear() { w[0] = w[1] = w[2] = w[3] = 0; }
    __device__ unsigned get(const int i) const // dynamic index
    {
        const uint64_t v = (i < 8) ? ((i < 4) ? w[0] : w[1]) : ((i < 12) ? w[2] : w[3]);
        return unsigned(v >> (16 * (i & 3))) & 0xffffu;
    }
    template <int I>
    __device__ unsigned at() const { return unsigned(w[I >> 2] >> (16 * (I & 3))) & 0xffffu; } // constant index
    __device__ void add(const int i, const unsigned d)
    {
        const uint64_t v = uint64_t(d) << (16 * (i & 3));
        w[0] += (i < 4) ? v : 0;
        w[1] += (i >= 4 && i < 8) ? v : 0;
        w[2] += (i >= 8 && i < 12) ? v : 0;
        w[3] += (i >= 12) ? v : 0;
    }
};

// The symbols of a code, ordered by (length, value), in LDS: the 64 lanes' tables interleaved byte by byte (element i of lane l at
// byte i * 64 + l: whatever the lanes' indices, consecutive lanes hit consecutive bytes), the low eight bits of a symbol in `lo`,
// the ninth (literal/length symbols reach 285) as a bit of `hi`.  In private arrays these tables were 1.3 KB of scratch memory per
// lane, 170 MB for a launch of 1.3e5 blocks -- far more than the L2 holds -- and every symbol look-up went to HBM for a whole line:
// 138 GB of traffic for 6.3 GB of algorithmic bytes (profiles/r03_v23_pmc_traffic.json).
struct SymTab
{
    uint8_t* lo;     // literal/length code: &lo_base[lane], element i at lo[i * 64]; nullptr = the distance code (registers only)
    uint64_t hi[5];  // literal/length code: the symbols' ninth bits (LDS room for them would cost the eighth resident wave of a CU:
                     // 288 x 64 bytes is 18 KB per wave, 8 waves = 144 of the CU's 160 KB); distance code: its 30 symbols, five
                     // bits each, twelve to a register
    __device__ void set(const int i, const int sym)
    {
        if (lo) {
            lo[i * 64] = uint8_t(sym);
            if (sym & 0x100) {
                const uint64_t bit = 1ull << (i & 63);
                const int j = i >> 6;
                hi[0] |= (j == 0) ? bit : 0;
                hi[1] |= (j == 1) ? bit : 0;
                hi[2] |= (j == 2) ? bit : 0;
                hi[3] |= (j == 3) ? bit : 0;
                hi[4] |= (j == 4) ? bit : 0;
            }
        } else {
            const int j = (i >= 24) ? 2 : (i >= 12) ? 1 : 0;
            const uint64_t v = uint64_t(unsigned(sym) & 31u) << (5 * (i - 12 * j));
            hi[0] |= (j == 0) ? v : 0;
            hi[1] |= (j == 1) ? v : 0;
            hi[2] |= (j == 2) ? v : 0;
        }
    }
    __device__ int get(const int i) const
    {
        if (lo) {
            const int v = lo[i * 64];
            const int j = i >> 6;
            const uint64_t word = (j == 0) ? hi[0] : (j == 1) ? hi[1] : (j == 2) ? hi[2] : (j == 3) ? hi[3] : hi[4];
            return ((word >> (i & 63)) & 1ull) ? (v | 0x100) : v;
        }
        const int j = (i >= 24) ? 2 : (i >= 12) ? 1 : 0;
        const uint64_t word = (j == 0) ? hi[0] : (j == 1) ? hi[1] : hi[2];
        return int((word >> (5 * (i - 12 * j))) & 31ull);
    }
    __device__ void clear_hi(const int /*n*/) { hi[0] = hi[1] = hi[2] = hi[3] = hi[4] = 0; }
};

struct Huffman
{
    Packed16 count; // [16] codes of each length
    SymTab symbol;  // [n] symbols ordered by (length, value)
};

// returns the number of codes left unused (0 = complete, < 0 = over-subscribed), as zlib's / puff's construct
__device__ int huff_construct(Huffman& h, const short* length, const int n)
{
    h.count.clear();
    h.symbol.clear_hi(n);
    for (int s = 0; s < n; ++s) h.count.add(length[s], 1u);
    if (int(h.count.get(0)) == n) return 0;
    int left = 1;
    for (int len = 1; len <= 15; ++len) {
        left <<= 1;
        left -= int(h.count.get(len));
        if (left < 0) return left;
    }
    Packed16 offs;
    offs.clear();
    {
        unsigned o = 0;
        for (int len = 1; len < 15; ++len) {
            o += h.count.get(len);
            offs.add(len + 1, o);
        }
    }
    for (int s = 0; s < n; ++s) {
        const int len = length[s];
        if (len != 0) {
            h.symbol.set(int(offs.get(len)), s);
            offs.add(len, 1u);
        }
    }
    return left;
}

template <int LEN>
__device__ __forceinline__ bool huff_step(const Huffman& h, unsigned& w, int& code, int& first, int& index, int& sym_index)
{
    code |= int(w & 1u);
    w >>= 1;
    const int count = int(h.count.at<LEN>());
    if (code - count < first) {
        sym_index = index + (code - first);
        return true;
    }
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
    return false;
}

__device__ int huff_decode(BitReader& br, const Huffman& h)
{
    if (br.cnt < 15) br.refill();
    unsigned w = unsigned(br.buf) & 0xffffu;
    int code = 0, first = 0, index = 0, si = 0, len = 0;
#define SK_HUFF_STEP(L) if (len == 0 && huff_step<L>(h, w, code, first, index, si)) len = L;
    SK_HUFF_STEP(1) SK_HUFF_STEP(2) SK_HUFF_STEP(3) SK_HUFF_STEP(4) SK_HUFF_STEP(5) SK_HUFF_STEP(6) SK_HUFF_STEP(7) SK_HUFF_STEP(8)
    SK_HUFF_STEP(9) SK_HUFF_STEP(10) SK_HUFF_STEP(11) SK_HUFF_STEP(12) SK_HUFF_STEP(13) SK_HUFF_STEP(14) SK_HUFF_STEP(15)
#undef SK_HUFF_STEP
    if (len == 0) return -1;
    br.consume(len);
    return h.symbol.get(si);
}

__device__ const short LEN_BASE[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
__device__ const short LEN_EXTRA[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
__device__ const short DIST_BASE[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
__device__ const short DIST_EXTRA[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
__device__ const unsigned char CLEN_ORDER[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

struct InflateArgs
{
    const uint8_t* data;      // the compressed file bytes (device)
    const int64_t* block_off; // [n_blocks+1] start of each BGZF block (its 18-byte header)
    const int64_t* out_off;   // [n_blocks+1] where each block's inflated bytes go
    uint8_t* out;
    int32_t* status;          // [n_blocks]
    int32_t n_blocks;
};

__device__ int inflate_codes(BitReader& br, uint8_t* out, int64_t& pos, const int64_t cap, const Huffman& lencode, const Huffman& distcode)
{
    // On this hardware a wave's loads and stores retire through ONE in-order counter: a load issued after a store cannot be waited
    // for without waiting for the store's acknowledgement too.  A literal stored straight away therefore puts a full write round
    // trip into the lane's dependent chain (the next table / input load waits for it).  Literals are collected in a register and
    // written eight at a time -- before a match (which may read them), when the register is full, and at the end of the block.
    uint64_t lit = 0;
    int nlit = 0; // pending literals: out[pos - nlit, pos)
    auto flush = [&]() {
        if (nlit) st_upto16(out + (pos - nlit), nlit, lit, 0);
        lit = 0;
        nlit = 0;
    };
    for (;;) {
        int sym = huff_decode(br, lencode);
        if (sym < 0 || br.overrun) {
            flush();
            return (sym < 0) ? INF_BAD_CODE : INF_IN_OVERRUN;
        }
        if (sym < 256) {
            if (pos >= cap) {
                flush();
                return INF_OUT_OVERFLOW;
            }
            lit |= uint64_t(unsigned(sym)) << (8 * nlit);
            ++nlit;
            ++pos;
            if (nlit == 8) flush();
        } else if (sym == 256) {
            flush();
            return INF_OK;
        } else {
            flush();
            sym -= 257;
            if (sym >= 29) return INF_BAD_CODE;
            // RFC 1951 3.2.5 in closed form (the four 30-entry tables were four more scattered loads per match): length codes 8..27
            // and distance codes 4..29 come in groups of four / two with one more extra bit per group
            const int le = (sym < 8 || sym == 28) ? 0 : ((sym - 4) >> 2);
            const int lbase = (sym < 8) ? (3 + sym) : (sym == 28) ? 258 : (3 + ((4 + (sym & 3)) << le));
            const int len = lbase + int(br.bits(le));
            const int ds = huff_decode(br, distcode);
            if (ds < 0 || ds >= 30) return INF_BAD_CODE;
            const int de = (ds < 4) ? 0 : ((ds - 2) >> 1);
            const int64_t dist = ((ds < 4) ? (1 + ds) : (1 + ((2 + (ds & 1)) << de))) + int(br.bits(de));
            if (dist > pos) return INF_BAD_DISTANCE; // (a BGZF block has no preset dictionary)
            if (pos + len > cap) return INF_OUT_OVERFLOW;
            // The copy reads this lane's own earlier output.  In pieces of k = min(left, dist, 16) bytes the loads of a piece do not
            // depend on its stores, and a piece moves as 8 + 4 + 2 + 1 byte accesses: at most four loads and four stores for 15
            // bytes where the byte loop issued fifteen of each, every one a wave-wide scatter.
            int left = len;
            while (left > 0) {
                const int k = min(left, int(min(dist, int64_t(16))));
                uint64_t lo, hi;
                ld_upto16(out + (pos - dist), k, lo, hi);
                st_upto16(out + pos, k, lo, hi);
                pos += k;
                left -= k;
            }
        }
    }
}

__global__ __launch_bounds__(64) void bgzf_inflate_kernel(const InflateArgs a)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.n_blocks) return;
    const uint8_t* blk = a.data + a.block_off[b];
    const int64_t blen = a.block_off[b + 1] - a.block_off[b];
    // gzip member header with the BGZF extra field (SAM spec 4.1): ID1 ID2 CM FLG(=4) MTIME XFL OS XLEN(=6) 'B' 'C' 2 BSIZE
    if (blen < 28 || blk[0] != 31 || blk[1] != 139 || blk[2] != 8 || !(blk[3] & 4)) {
        a.status[b] = INF_BAD_HEADER;
        return;
    }
    const int xlen = int(blk[10]) | (int(blk[11]) << 8);
    const uint8_t* cdata = blk + 12 + xlen;
    const uint8_t* cend = blk + blen - 8;
    if (cdata > cend) {
        a.status[b] = INF_BAD_HEADER;
        return;
    }
    const uint32_t isize = uint32_t(cend[4]) | (uint32_t(cend[5]) << 8) | (uint32_t(cend[6]) << 16) | (uint32_t(cend[7]) << 24);
    uint8_t* out = a.out + a.out_off[b];
    const int64_t cap = a.out_off[b + 1] - a.out_off[b];
    int64_t pos = 0;

    BitReader br;
    br.p = cdata;
    br.end = cend;
    br.buf = 0;
    br.cnt = 0;
    br.overrun = false;
    br.res_at = nullptr;
    __shared__ uint8_t s_len_lo[288 * 64];
    short lengths[320];
    Huffman lencode, distcode;
    lencode.symbol.lo = s_len_lo + threadIdx.x;
    lencode.symbol.clear_hi(0);
    distcode.symbol.lo = nullptr;
    distcode.symbol.clear_hi(0);
    int err = INF_OK, last = 0;
    do {
        last = int(br.bits(1));
        const int type = int(br.bits(2));
        if (type == 0) { // stored
            br.align_to_byte(); // (discard the rest of the current byte; whole bytes read ahead go back to the input)
            if (br.p + 4 > br.end) { err = INF_BAD_STORED; break; }
            const unsigned len = unsigned(br.p[0]) | (unsigned(br.p[1]) << 8);
            const unsigned nlen = unsigned(br.p[2]) | (unsigned(br.p[3]) << 8);
            br.p += 4;
            if (len != (~nlen & 0xffffu) || br.p + len > br.end) { err = INF_BAD_STORED; break; }
            if (pos + int64_t(len) > cap) { err = INF_OUT_OVERFLOW; break; }
            for (unsigned i = 0; i < len; i += 16) {
                const int k = int(min(16u, len - i));
                uint64_t lo, hi;
                ld_upto16(br.p + i, k, lo, hi);
                st_upto16(out + pos + i, k, lo, hi);
            }
            pos += len;
            br.p += len;
        } else if (type == 1) { // fixed codes (3.2.6)
            int s = 0;
            for (; s < 144; ++s) lengths[s] = 8;
            for (; s < 256; ++s) lengths[s] = 9;
            for (; s < 280; ++s) lengths[s] = 7;
            for (; s < 288; ++s) lengths[s] = 8;
            huff_construct(lencode, lengths, 288);
            for (s = 0; s < 30; ++s) lengths[s] = 5;
            huff_construct(distcode, lengths, 30);
            err = inflate_codes(br, out, pos, cap, lencode, distcode);
        } else if (type == 2) { // dynamic codes (3.2.7)
            const int nlen = int(br.bits(5)) + 257, ndist = int(br.bits(5)) + 1, ncode = int(br.bits(4)) + 4;
            if (nlen > 286 || ndist > 30) { err = INF_BAD_LENGTHS; break; }
            int idx = 0;
            for (; idx < ncode; ++idx) lengths[CLEN_ORDER[idx]] = short(br.bits(3));
            for (; idx < 19; ++idx) lengths[CLEN_ORDER[idx]] = 0;
            if (huff_construct(lencode, lengths, 19) != 0) { err = INF_BAD_LENGTHS; break; } // (the code-length code must be complete)
            idx = 0;
            while (idx < nlen + ndist) {
                int sym = huff_decode(br, lencode);
                if (sym < 0) { err = INF_BAD_CODE; break; }
                if (sym < 16) {
                    lengths[idx++] = short(sym);
                } else {
                    int len = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) { err = INF_BAD_LENGTHS; break; }
                        len = lengths[idx - 1];
                        rep = 3 + int(br.bits(2));
                    } else if (sym == 17) {
                        rep = 3 + int(br.bits(3));
                    } else {
                        rep = 11 + int(br.bits(7));
                    }
                    if (idx + rep > nlen + ndist) { err = INF_BAD_LENGTHS; break; }
                    while (rep--) lengths[idx++] = short(len);
                }
            }
            if (err != INF_OK) break;
            if (lengths[256] == 0) { err = INF_BAD_LENGTHS; break; }
            int left = huff_construct(lencode, lengths, nlen);
            if (left != 0 && (left < 0 || nlen != int(lencode.count.get(0) + lencode.count.get(1)))) { err = INF_BAD_LENGTHS; break; } // (incomplete only as one 1-bit code)
            left = huff_construct(distcode, lengths + nlen, ndist);
            if (left != 0 && (left < 0 || ndist != int(distcode.count.get(0) + distcode.count.get(1)))) { err = INF_BAD_LENGTHS; break; }
            err = inflate_codes(br, out, pos, cap, lencode, distcode);
        } else {
            err = INF_BAD_BLOCK_TYPE;
        }
        if (br.overrun && err == INF_OK) err = INF_IN_OVERRUN;
    } while (err == INF_OK && !last);
    if (err == INF_OK && (pos != cap || uint32_t(pos) != isize)) err = INF_SIZE_MISMATCH;
    a.status[b] = err;
}

__global__ __launch_bounds__(64) void bgzf_crc32_kernel(const InflateArgs a)
{
    __shared__ uint32_t table[256];
    for (int i = threadIdx.x; i < 256; i += 64) {
        uint32_t c = uint32_t(i);
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
        table[i] = c;
    }
    __syncthreads();
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.n_blocks || a.status[b] != INF_OK) return;
    const uint8_t* out = a.out + a.out_off[b];
    const int64_t n = a.out_off[b + 1] - a.out_off[b];
    uint32_t c = 0xffffffffu;
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) { // (one unaligned 8-byte load per eight table steps: the loads are wave-wide scatters)
        const uint64_t v = ld_u64(out + i);
#pragma unroll
        for (int k = 0; k < 8; ++k) c = table[(c ^ uint32_t(v >> (8 * k))) & 0xffu] ^ (c >> 8);
    }
    for (; i < n; ++i) c = table[(c ^ out[i]) & 0xffu] ^ (c >> 8);
    c ^= 0xffffffffu;
    const uint8_t* t = a.data + a.block_off[b + 1] - 8;
    const uint32_t want = uint32_t(t[0]) | (uint32_t(t[1]) << 8) | (uint32_t(t[2]) << 16) | (uint32_t(t[3]) << 24);
    if (c != want) a.status[b] = INF_CRC_MISMATCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// B1s  bgzf_inflate_scalar_kernel: the wave-per-block decoder with its serial part on the SCALAR unit.
//
// What a slice of a few hundred blocks costs is the latency of ONE block: ~1e4 symbols (BAM data: two thirds of them matches of ~9
// bytes), one after the other.  Round 3's wave-per-block kernel ran every symbol on all lanes in lockstep through LDS: two to six
// dependent LDS round trips of ~100 cycles each (the input ring word, the table entry, the base / extra tables, the copy) plus the
// waits the fences put around every copy -- 11.4 ms per slice of 510 blocks (profiles/r04_inflate_history.txt).  A wave alone on its
// SIMD issues one instruction every ~6 cycles, so what counts is the number of instructions and waits on the symbol-to-symbol chain.
// Here nothing on that chain goes to memory:
//   * the compressed bytes sit in two VGPRs (lane l holds dword l of the current 256 bytes, and of the next 256, loaded ahead straight
//     from HBM); the next word enters the bit buffer through v_readlane with a scalar index;
//   * the first-level tables sit in VGPRs too -- 2 048 literal/length entries in 32 registers, 1 024 distance entries in 16, an entry
//     32 bits with everything the symbol needs (code length, extra-bit count, base value, kind): an entry is a register picked by the
//     high index bits (s_set_gpr_idx) and v_readlane by the low ones, five instructions and no wait;
//   * the bit buffer, its count, the output position, match length and distance are wave-uniform values the compiler keeps in SGPRs:
//     shifts and masks issue on the scalar unit;
//   * LDS is written, never waited for, by a literal (one byte store; the other lanes store into a sink of their own, an address select
//     instead of a branch); a match's copy is the one wait: read, then store, in rounds of 64 bytes (LDS executes a wave's accesses in
//     order, so the read sees every byte stored before);
//   * the tables are built by the lanes: per-length counts and every symbol's rank among its length by ballots (no serial pass over the
//     code lengths), the canonical code of a symbol = first code of its length + rank.
// Same checks and status codes as B1.
// ---------------------------------------------------------------------------------------------------------------------

constexpr int IS_FAST = 11, IS_DFAST = 10, IS_CFAST = 7;
typedef uint32_t is_v32u __attribute__((ext_vector_type(32)));
typedef uint32_t is_v16u __attribute__((ext_vector_type(16)));

// A first-level entry is 32 bits, everything a symbol needs without a further look-up:
//   bits 0-3  the code's length (0: no code this short ends here -- longer than the table, or not a code at all)
//   bits 4-7  the number of extra bits that follow
//   bits 8-23 a literal's byte / a match length's base (3..258) / a distance's base (1..24 577) / a code-length symbol (0..18)
//   bit 31    set on everything of the literal/length alphabet that is not a literal with a valid length (so that ONE sign test picks
//             the literal path), bit 30: end of block, bit 29: a symbol that may not occur (286, 287; distances 30, 31)
constexpr uint32_t IS_NOT_LITERAL = 1u << 31, IS_END = 1u << 30, IS_BAD = 1u << 29;

__device__ __forceinline__ uint32_t is_pack_litlen(const int sym, const int len)
{
    if (sym < 256) return uint32_t(len) | (uint32_t(sym) << 8);
    if (sym == 256) return uint32_t(len) | IS_NOT_LITERAL | IS_END;
    const int s = sym - 257;
    if (s >= 29) return uint32_t(len) | IS_NOT_LITERAL | IS_BAD;
    // RFC 1951 3.2.5 in closed form: length codes 8..27 come in groups of four with one more extra bit per group
    const int le = (s < 8 || s == 28) ? 0 : ((s - 4) >> 2);
    const int base = (s < 8) ? (3 + s) : (s == 28) ? 258 : (3 + ((4 + (s & 3)) << le));
    return uint32_t(len) | (uint32_t(le) << 4) | (uint32_t(base) << 8) | IS_NOT_LITERAL;
}
__device__ __forceinline__ uint32_t is_pack_dist(const int ds, const int len)
{
    if (ds >= 30) return uint32_t(len) | IS_BAD;
    const int de = (ds < 4) ? 0 : ((ds - 2) >> 1); // distance codes 4..29: groups of two
    const int base = (ds < 4) ? (1 + ds) : (1 + ((2 + (ds & 1)) << de));
    return uint32_t(len) | (uint32_t(de) << 4) | (uint32_t(base) << 8);
}
__device__ __forceinline__ uint32_t is_pack_plain(const int sym, const int len) { return uint32_t(len) | (uint32_t(sym) << 8); }
enum { IS_KIND_LITLEN = 0, IS_KIND_DIST = 1, IS_KIND_PLAIN = 2 };
__device__ __forceinline__ uint32_t is_pack(const int kind, const int sym, const int len)
{
    return kind == IS_KIND_LITLEN ? is_pack_litlen(sym, len) : kind == IS_KIND_DIST ? is_pack_dist(sym, len) : is_pack_plain(sym, len);
}

struct InflateScalarLds
{
    uint8_t out[65536];
    uint32_t crc_table[256];
    uint32_t fast[1 << IS_FAST];           // the first-level table being built (literal/length codes; code-length codes): copied to registers
    uint32_t fast_dist[1 << IS_DFAST];     // ... of the distance codes
    uint16_t lit_sym[288], dist_sym[32];   // symbols sorted by code (the walk for codes longer than the first-level table)
    uint16_t lit_count[16], dist_count[16];
    uint16_t offs[16], first[16];
    uint8_t lengths[320], lengths2[320];
    uint32_t sink[64];                     // where a lane with nothing to store stores (a bank each)
};

struct ScalarBits
{
    uint64_t buf;        // wave-uniform
    int cnt;             // bits in buf
    int widx;            // next dword of `cur` to enter buf, 0..64 (64: the next refill moves on to `nxt` first)
    int base;            // offset of `cur`'s 256 bytes from g0
    uint32_t cur, nxt;   // per lane: dword `lane` of the 256 bytes at base / base + 256 (nxt: as loaded, see is_chunk_seen)
    const uint8_t* g0;   // 4-byte aligned origin
    int limit;           // readable bytes from g0 (the block's end, trailer included)
};

// dword `lane` of the 256 bytes at `off`, as loaded / as the reader may see it (zero past the end).  Two steps so that the load can be
// issued one chunk ahead and waited for only when its chunk becomes the current one; no per-lane branch: a lane past the end loads the
// last whole dword and drops it.
__device__ __forceinline__ uint32_t is_load_chunk_raw(const ScalarBits& b, const int off, const int lane)
{
    // (said to be global memory: a flat load -- the pointer came through the by-reference argument of is_codes -- is waited for on the
    // LDS counter too, and then every symbol waits for the byte it stored)
    typedef const __attribute__((address_space(1))) uint32_t* global_u32;
    return *(global_u32)(b.g0 + min(off + lane * 4, (b.limit - 4) & ~3));
}
__device__ __forceinline__ uint32_t is_chunk_seen(const ScalarBits& b, const uint32_t raw, const int off, const int lane)
{
    return (off + lane * 4 + 4 <= b.limit) ? raw : 0u;
}

__device__ __forceinline__ void is_roll(ScalarBits& b, const int lane)
{
    b.base += 256;
    b.cur = is_chunk_seen(b, b.nxt, b.base, lane);
    b.nxt = is_load_chunk_raw(b, b.base + 256, lane);
    b.widx = 0;
}

__device__ __forceinline__ uint64_t is_uniform64(const uint64_t v)
{
    return uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v))))) |
           (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))))) << 32);
}
// The bit reader's state is the same in every lane; said to the compiler after code whose per-lane branches (table fills, run fills)
// make it keep the state in vector registers -- the symbol loops run on the scalar unit only if it knows.
__device__ __forceinline__ void is_pin(ScalarBits& b)
{
    b.buf = is_uniform64(b.buf);
    b.cnt = __builtin_amdgcn_readfirstlane(b.cnt);
    b.widx = __builtin_amdgcn_readfirstlane(b.widx);
    b.base = __builtin_amdgcn_readfirstlane(b.base);
    b.limit = __builtin_amdgcn_readfirstlane(b.limit);
    b.g0 = reinterpret_cast<const uint8_t*>(is_uniform64(reinterpret_cast<uint64_t>(b.g0)));
}

// at least 32 bits in the buffer afterwards
__device__ __forceinline__ void is_refill(ScalarBits& b, const int lane)
{
    if (b.cnt < 32) {
        if (b.widx == 64) is_roll(b, lane);
        const uint32_t w = uint32_t(__builtin_amdgcn_readlane(int(b.cur), b.widx));
        b.buf |= uint64_t(w) << b.cnt;
        b.cnt += 32;
        ++b.widx;
    }
}

__device__ __forceinline__ unsigned is_take(ScalarBits& b, const int n) // n <= 16, the buffer holds them
{
    const unsigned v = unsigned(b.buf) & ((1u << n) - 1u);
    b.buf >>= n;
    b.cnt -= n;
    return v;
}

__device__ __forceinline__ void is_seek(ScalarBits& b, const int off, const int lane)
{
    b.base = off & ~3;
    b.cur = is_chunk_seen(b, is_load_chunk_raw(b, b.base, lane), b.base, lane);
    b.nxt = is_load_chunk_raw(b, b.base + 256, lane);
    b.widx = 0;
    b.buf = 0;
    b.cnt = 0;
    is_refill(b, lane);
    (void)is_take(b, (off & 3) * 8);
}

__device__ __forceinline__ int is_byte_offset(const ScalarBits& b) { return b.base + b.widx * 4 - (b.cnt >> 3); }

// the canonical walk over the buffered bits for a code longer than the first-level table: symbol | length << 9, -1 if no code matches
// (One exit, no early return: with an exit per length in the caller's loop the compiler threads flags through all of that loop.)
template <typename CountPtr>
__device__ __forceinline__ int is_slow_decode(const uint64_t buf, const CountPtr count, const CountPtr symbol)
{
    int code = 0, first = 0, index = 0, at = -1, len_hit = 0;
#pragma nounroll
    for (int len = 1; len <= 15; ++len) {
        code |= int((buf >> (len - 1)) & 1ull);
        const int c = __builtin_amdgcn_readfirstlane(int(count[len])); // (uniform to the compiler too: no per-lane branch in the caller's loop)
        const bool hit = at < 0 && code - c < first;
        at = hit ? index + (code - first) : at;
        len_hit = hit ? len : len_hit;
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    const int sym = __builtin_amdgcn_readfirstlane(int(symbol[max(at, 0)]));
    return at < 0 ? -1 : (sym | (len_hit << 9));
}

// Per-length counts, the symbols sorted by code and the first-level table of a canonical code, all by the lanes; returns puff's `left`
// (0: complete, > 0: incomplete, < 0: over-subscribed; nothing is built then).  `length` is an LDS array of n <= 320 code lengths.
__device__ __noinline__ int is_construct(InflateScalarLds& L, uint16_t* count, uint16_t* symbol, uint32_t* fast, const int fast_bits, const int kind,
                                         const uint8_t* length, const int n, const int lane)
{
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
    int cnt[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) cnt[l] = 0;
    int my_len[5], my_rank[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const int s = c * 64 + lane;
        const int l = (s < n) ? int(length[s]) : 16;
        my_len[c] = l;
        my_rank[c] = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const uint64_t m = __ballot(l == v);
            if (l == v) my_rank[c] = cnt[v] + __popcll(m & below);
            cnt[v] += __popcll(m);
        }
    }
    for (int i = lane; i < (1 << fast_bits); i += 64) fast[i] = 0;
    if (lane == 0) {
#pragma unroll
        for (int l = 0; l < 16; ++l) count[l] = uint16_t(cnt[l]);
    }
    int left = 1;
    bool over = false;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= cnt[l];
        over = over || (left < 0);
        if (over) left = left < 0 ? left : -1;
    }
    if (cnt[0] == n) left = 0;
    if (cnt[0] == n || over) {
        __syncthreads();
        return left;
    }
    if (lane == 0) {
        int start = 0, code0 = 0;
#pragma unroll
        for (int l = 1; l <= 15; ++l) {
            L.offs[l] = uint16_t(start);
            L.first[l] = uint16_t(code0);
            start += cnt[l];
            code0 = (code0 + cnt[l]) << 1;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const int l = my_len[c];
        if (l >= 1 && l <= 15) {
            const int s = c * 64 + lane;
            symbol[int(L.offs[l]) + my_rank[c]] = uint16_t(s);
            if (l <= fast_bits) {
                const unsigned code = unsigned(L.first[l]) + unsigned(my_rank[c]);
                const unsigned rev = __brev(code) >> (32 - l); // DEFLATE packs Huffman codes most significant bit first
                const uint32_t entry = is_pack(kind, s, l);
                for (unsigned e = rev; e < (1u << fast_bits); e += (1u << l)) fast[e] = entry;
            }
        }
    }
    __syncthreads();
    return left;
}

// entry i of a table in registers: lane (i & 63) of register (i >> 6).  (The registers are passed by value: a table inside a struct
// taken by reference stays in scratch memory.)
__device__ __forceinline__ uint32_t is_lit_entry(const is_v32u lit, const unsigned idx)
{
    return uint32_t(__builtin_amdgcn_readlane(int(lit[idx >> 6]), int(idx & 63u)));
}
__device__ __forceinline__ uint32_t is_dist_entry(const is_v16u dist, const unsigned idx)
{
    return uint32_t(__builtin_amdgcn_readlane(int(dist[idx >> 6]), int(idx & 63u)));
}

// The symbols of one deflate block.  Nothing in this loop branches on a per-lane value: a lane with nothing to store stores into a
// sink of its own (an address select, not a branch); a match of up to 64 bytes is read by lane min(lane, length - 1), longer ones in
// uniform rounds of 64.  A function of its own, not inlined: inside the kernel the compiler merges this loop with the loop over deflate blocks
// around it, whose per-lane branches (table fills) put the whole region through the structuriser -- every exit of the symbol loop
// becomes a flag tested on the way round -- and whose hoisted lane masks take the scalar registers this loop lives in.
extern __shared__ __align__(16) unsigned char is_lds_raw[];

#ifdef SK_IS_TIMING  // experiments: where a block's time goes (build with SK_EXTRA_HIPCC_FLAGS=-DSK_IS_TIMING, run with $SK_INFLATE_TIMING=1)
__device__ unsigned long long is_dbg[16384 * 8];
#define IS_DBG_ADD(i, v) do { if (lane == 0) is_dbg[size_t(blockIdx.x) * 8 + (i)] += (unsigned long long)(v); } while (0)
#else
#define IS_DBG_ADD(i, v) do { } while (0)
#endif

__device__ __noinline__ int is_codes(ScalarBits* bits, int* pos_io, const int cap_arg, const unsigned lds_base_arg, const int lane)
{
    // (arguments arrive in vector registers; the LDS base as a number: a function that names the dynamic LDS array itself loads its
    // address from a table at every use -- a scalar load and a wait for it per symbol)
    const int cap = __builtin_amdgcn_readfirstlane(cap_arg);
    typedef __attribute__((address_space(3))) uint8_t* lds_u8;
    typedef __attribute__((address_space(3))) uint16_t* lds_u16;
    typedef __attribute__((address_space(3))) uint32_t* lds_u32;
    const lds_u8 lds = (lds_u8)(uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane(int(lds_base_arg));
    const lds_u32 fast = (lds_u32)(lds + offsetof(InflateScalarLds, fast)), fast_dist = (lds_u32)(lds + offsetof(InflateScalarLds, fast_dist));
    const lds_u16 lit_count = (lds_u16)(lds + offsetof(InflateScalarLds, lit_count)), lit_sym = (lds_u16)(lds + offsetof(InflateScalarLds, lit_sym));
    const lds_u16 dist_count = (lds_u16)(lds + offsetof(InflateScalarLds, dist_count)), dist_sym = (lds_u16)(lds + offsetof(InflateScalarLds, dist_sym));
    ScalarBits b = *bits;
    is_pin(b);
    is_v32u lit;
#pragma unroll
    for (int r = 0; r < 32; ++r) lit[r] = fast[r * 64 + lane];
    is_v16u dist;
#pragma unroll
    for (int r = 0; r < 16; ++r) dist[r] = fast_dist[r * 64 + lane];
    int pos = __builtin_amdgcn_readfirstlane(*pos_io);
    const uint32_t my_sink = uint32_t(offsetof(InflateScalarLds, sink)) + 4u * lane; // (offsets into the LDS struct; `out` is at 0)
    int err = INF_OK;
#ifdef SK_IS_TIMING
    const long long t_in = clock64();
    int n_lit = 0, n_match = 0;
    long long t_lit = 0, t_match = 0, t_mdec = 0;
#endif
    for (;;) {
#ifdef SK_IS_TIMING
        const long long t_it = clock64();
#endif
        is_refill(b, lane);
        uint32_t e = is_lit_entry(lit, unsigned(b.buf) & ((1u << IS_FAST) - 1u));
        if (__builtin_expect((e & 15u) == 0, 0)) {
            const int r = is_slow_decode(b.buf, lit_count, lit_sym);
            e = r < 0 ? (IS_NOT_LITERAL | IS_BAD) : is_pack_litlen(r & 511, r >> 9);
        }
        (void)is_take(b, int(e & 15u));
        if (int32_t(e) >= 0) { // a literal
            // lane 0 stores the byte, the others into their sink: an address select, not a branch (and not 64 stores to one address).  No
            // wait: nothing on the way to the next symbol reads LDS.
            // (pos <= cap <= 65 536 here: at pos == cap the store lands on byte 0 of a block that fails)
            lds[lane == 0 ? uint32_t(pos & 0xffff) : my_sink] = uint8_t(e >> 8);
            ++pos;
#ifdef SK_IS_TIMING
            ++n_lit;
            t_lit += clock64() - t_it;
#endif
            if (__builtin_expect(pos > cap, 0)) { err = INF_OUT_OVERFLOW; break; }
            continue;
        }
        if (__builtin_expect((e & (IS_END | IS_BAD)) != 0, 0)) {
            if (e & IS_BAD) err = INF_BAD_CODE;
            break;
        }
        const int mlen = int((e >> 8) & 0xffffu) + int(is_take(b, int((e >> 4) & 15u)));
        is_refill(b, lane);
        uint32_t d = is_dist_entry(dist, unsigned(b.buf) & ((1u << IS_DFAST) - 1u));
        if (__builtin_expect((d & 15u) == 0, 0)) {
            const int r = is_slow_decode(b.buf, dist_count, dist_sym);
            d = r < 0 ? IS_BAD : is_pack_dist(r & 511, r >> 9);
        }
        if (d & IS_BAD) { err = INF_BAD_CODE; break; }
        (void)is_take(b, int(d & 15u));
        const int dst = int((d >> 8) & 0xffffu) + int(is_take(b, int((d >> 4) & 15u)));
        if (dst > pos) { err = INF_BAD_DISTANCE; break; } // (a BGZF block has no preset dictionary)
        if (pos + mlen > cap) { err = INF_OUT_OVERFLOW; break; }
        // The copy, in uniform rounds of 64 bytes (one round for all but the longest matches): LDS executes a wave's accesses in order, so
        // the reads see every byte stored before, and byte k of an overlapping match (distance < length) is byte (k mod distance) of the
        // `distance` bytes before it -- never a byte of this match.  The read's round trip is the one wait of the symbol loop.
#ifdef SK_IS_TIMING
        t_mdec += clock64() - t_it;
#endif
        const int src0 = pos - dst;
        if (__builtin_expect(dst >= mlen && mlen <= 64, 1)) {
            lds[lane < mlen ? uint32_t(pos + lane) : my_sink] = lds[src0 + min(lane, mlen - 1)];
        } else {
            const float inv = 1.0f / float(dst);
            for (int base = 0; base < mlen; base += 64) {
                const int k = min(base + lane, mlen - 1);
                int r = k;
                if (dst < mlen) { // k < 258 and distance < 258: the quotient from a float reciprocal is off by at most one
                    const int q = int(float(k) * inv);
                    r = k - q * dst;
                    r += (r < 0) ? dst : 0;
                    r -= (r >= dst) ? dst : 0;
                }
                lds[base + lane < mlen ? uint32_t(pos + k) : my_sink] = lds[src0 + r];
            }
        }
        pos += mlen;
#ifdef SK_IS_TIMING
        ++n_match;
        t_match += clock64() - t_it;
#endif
    }
#ifdef SK_IS_TIMING
    IS_DBG_ADD(2, clock64() - t_in);
    IS_DBG_ADD(3, n_lit);
    IS_DBG_ADD(4, n_match);
    IS_DBG_ADD(1, t_lit);
    IS_DBG_ADD(7, t_match);
    IS_DBG_ADD(6, t_mdec);
#endif
    *pos_io = pos;
    *bits = b;
    return err;
}

// a * b mod P over GF(2), reflected (zlib crc32.c multmodp)
__device__ __forceinline__ uint32_t is_multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1u)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? ((b >> 1) ^ 0xedb88320u) : (b >> 1);
    }
    return p;
}

__global__ __launch_bounds__(64) void bgzf_inflate_scalar_kernel(const InflateArgs a)
{
    InflateScalarLds& L = *reinterpret_cast<InflateScalarLds*>(is_lds_raw);
    const int lane = threadIdx.x;
    const int blk_i = blockIdx.x;
    if (blk_i >= a.n_blocks) return;
    const uint8_t* blk = a.data + a.block_off[blk_i];
    const int64_t blen = a.block_off[blk_i + 1] - a.block_off[blk_i];
    if (blen < 28 || blen > (1 << 20) || blk[0] != 31 || blk[1] != 139 || blk[2] != 8 || !(blk[3] & 4)) {
        if (lane == 0) a.status[blk_i] = INF_BAD_HEADER;
        return;
    }
    const int xlen = int(blk[10]) | (int(blk[11]) << 8);
    const uint8_t* cdata = blk + 12 + xlen;
    const uint8_t* cend = blk + blen - 8;
    if (cdata > cend) {
        if (lane == 0) a.status[blk_i] = INF_BAD_HEADER;
        return;
    }
    const uint32_t isize = uint32_t(cend[4]) | (uint32_t(cend[5]) << 8) | (uint32_t(cend[6]) << 16) | (uint32_t(cend[7]) << 24);
    const uint32_t want_crc = uint32_t(cend[0]) | (uint32_t(cend[1]) << 8) | (uint32_t(cend[2]) << 16) | (uint32_t(cend[3]) << 24);
    const int64_t cap64 = a.out_off[blk_i + 1] - a.out_off[blk_i];
    if (cap64 < 0 || cap64 > 65536) {
        if (lane == 0) a.status[blk_i] = INF_OUT_OVERFLOW;
        return;
    }
    const int cap = __builtin_amdgcn_readfirstlane(int(cap64));

    for (int i = lane; i < 256; i += 64) {
        uint32_t c = uint32_t(i);
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
        L.crc_table[i] = c;
    }
#ifdef SK_IS_TIMING
    const long long t_start = clock64();
    if (lane < 8) is_dbg[size_t(blockIdx.x) * 8 + lane] = 0;
#endif

    ScalarBits b;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(cdata);
    b.g0 = cdata - (addr & 3u);
    b.limit = int((blk + blen) - b.g0);
    const int origin = int(addr & 3u);
    is_seek(b, origin, lane);
    const int total_bits = int(cend - cdata) * 8;
    auto consumed_bits = [&]() { return (b.base + b.widx * 4 - origin) * 8 - b.cnt; };

    int pos = 0, err = INF_OK, last = 0;
    do {
        is_pin(b);
        pos = __builtin_amdgcn_readfirstlane(pos);
        is_refill(b, lane);
        last = int(is_take(b, 1));
        const int type = int(is_take(b, 2));
        if (type == 0) { // stored: the rest of the byte is dropped, LEN NLEN, then LEN bytes straight from the input
            (void)is_take(b, b.cnt & 7);
            is_refill(b, lane);
            const unsigned len = is_take(b, 16);
            is_refill(b, lane);
            const unsigned nlen = is_take(b, 16);
            if (len != (~nlen & 0xffffu) || consumed_bits() + int(len) * 8 > total_bits) { err = INF_BAD_STORED; break; }
            if (pos + int(len) > cap) { err = INF_OUT_OVERFLOW; break; }
            const int from = is_byte_offset(b);
            for (int k = lane; k < int(len); k += 64) L.out[pos + k] = b.g0[from + k];
            pos += int(len);
            is_seek(b, from + int(len), lane);
        } else if (type == 1 || type == 2) {
            int nlen = 288, ndist = 30;
            if (type == 1) { // fixed codes (RFC 1951 3.2.6)
                __syncthreads();
                for (int s = lane; s < 288; s += 64) L.lengths[s] = uint8_t(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                if (lane < 30) L.lengths[288 + lane] = 5;
                __syncthreads();
            } else { // dynamic codes (3.2.7)
                nlen = int(is_take(b, 5)) + 257;
                ndist = int(is_take(b, 5)) + 1;
                const int ncode = int(is_take(b, 4)) + 4;
                if (nlen > 286 || ndist > 30) { err = INF_BAD_LENGTHS; break; }
                __syncthreads();
                if (lane < 19) L.lengths2[lane] = 0;
                __syncthreads();
                for (int idx = 0; idx < ncode; ++idx) {
                    is_refill(b, lane);
                    const unsigned v = is_take(b, 3);
                    L.lengths2[CLEN_ORDER[idx]] = uint8_t(v);
                }
                // the code-length code: at most 7 bits, one register of entries
                if (is_construct(L, L.dist_count, L.dist_sym, L.fast, IS_CFAST, IS_KIND_PLAIN, L.lengths2, 19, lane) != 0) { err = INF_BAD_LENGTHS; break; }
                const uint32_t ct0 = L.fast[lane], ct1 = L.fast[64 + lane];
                int idx = 0, prev = 0;
                const int want = nlen + ndist;
                __syncthreads();
                is_pin(b);
                while (idx < want) {
                    is_refill(b, lane);
                    const unsigned at = unsigned(b.buf) & 127u;
                    const uint32_t e = uint32_t(__builtin_amdgcn_readlane(int((at & 64u) ? ct1 : ct0), int(at & 63u)));
                    const int len = int(e & 15u), sym = int(e >> 8);
                    if (len == 0) { err = INF_BAD_CODE; break; }
                    (void)is_take(b, len);
                    if (sym < 16) {
                        L.lengths[idx] = uint8_t(sym);
                        prev = sym;
                        ++idx;
                    } else {
                        int fill = 0, rep;
                        if (sym == 16) {
                            if (idx == 0) { err = INF_BAD_LENGTHS; break; }
                            fill = prev;
                            rep = 3 + int(is_take(b, 2));
                        } else if (sym == 17) {
                            rep = 3 + int(is_take(b, 3));
                        } else {
                            rep = 11 + int(is_take(b, 7));
                        }
                        if (idx + rep > want) { err = INF_BAD_LENGTHS; break; }
                        for (int k = lane; k < rep; k += 64) L.lengths[idx + k] = uint8_t(fill);
                        idx += rep;
                        prev = fill;
                    }
                }
                if (err != INF_OK) break;
                __syncthreads();
                if (__builtin_amdgcn_readfirstlane(int(L.lengths[256])) == 0) { err = INF_BAD_LENGTHS; break; }
            }
            int left = is_construct(L, L.lit_count, L.lit_sym, L.fast, IS_FAST, IS_KIND_LITLEN, L.lengths, nlen, lane);
            if (type == 2 && left != 0 && (left < 0 || nlen != __builtin_amdgcn_readfirstlane(int(L.lit_count[0]) + int(L.lit_count[1])))) { err = INF_BAD_LENGTHS; break; }
            left = is_construct(L, L.dist_count, L.dist_sym, L.fast_dist, IS_DFAST, IS_KIND_DIST, L.lengths + nlen, ndist, lane);
            if (type == 2 && left != 0 && (left < 0 || ndist != __builtin_amdgcn_readfirstlane(int(L.dist_count[0]) + int(L.dist_count[1])))) { err = INF_BAD_LENGTHS; break; }
            // (the base through an empty asm: handed over as a constant expression the compiler propagates it into is_codes and the table
            // loads are back)
            unsigned lds_base = unsigned(uintptr_t((__attribute__((address_space(3))) unsigned char*)is_lds_raw));
            asm volatile("" : "+v"(lds_base));
            err = __builtin_amdgcn_readfirstlane(is_codes(&b, &pos, cap, lds_base, lane));
        } else {
            err = INF_BAD_BLOCK_TYPE;
        }
        if (err == INF_OK && consumed_bits() > total_bits) err = INF_IN_OVERRUN;
    } while (err == INF_OK && !last);
    if (err == INF_OK && (pos != cap || uint32_t(pos) != isize)) err = INF_SIZE_MISMATCH;
    __syncthreads();
#ifdef SK_IS_TIMING
    const long long t_decoded = clock64();
#endif
    if (err == INF_OK) {
        // CRC-32 of the block: 64 slices of whole dwords, slice i shifted by the bytes after it (crc(A || B) = crc(A) * x^(8 |B|) + crc(B))
        const int n = pos;
        const int slice = (((n + 63) / 64) + 3) & ~3;
        const int s0 = min(n, lane * slice), s1 = min(n, s0 + slice);
        uint32_t c = 0;
        if (s1 > s0) {
            c = 0xffffffffu;
            int i = s0;
            for (; i + 4 <= s1; i += 4) {
                const uint32_t w = *reinterpret_cast<const uint32_t*>(&L.out[i]);
                c = L.crc_table[(c ^ w) & 0xffu] ^ (c >> 8);
                c = L.crc_table[(c ^ (w >> 8)) & 0xffu] ^ (c >> 8);
                c = L.crc_table[(c ^ (w >> 16)) & 0xffu] ^ (c >> 8);
                c = L.crc_table[(c ^ (w >> 24)) & 0xffu] ^ (c >> 8);
            }
            for (; i < s1; ++i) c = L.crc_table[(c ^ L.out[i]) & 0xffu] ^ (c >> 8);
            c ^= 0xffffffffu;
            // x^(8 (n - s1)) mod P by square and multiply (zlib x2nmodp)
            uint32_t sq = 1u << 30; // x^1
            uint32_t pw = 1u << 31; // x^0
            uint32_t e = uint32_t(n - s1) * 8u;
            while (e) {
                if (e & 1u) pw = is_multmodp(sq, pw);
                sq = is_multmodp(sq, sq);
                e >>= 1;
            }
            c = is_multmodp(pw, c);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c ^= __shfl_xor(c, d, 64);
        if (c != want_crc) err = INF_CRC_MISMATCH;
        // the block to HBM: consecutive lanes, consecutive bytes (4 per lane where the destination allows)
        uint8_t* dst = a.out + a.out_off[blk_i];
        const int head = min(n, int((4u - (reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u));
        if (lane < head) dst[lane] = L.out[lane];
        const int words = (n - head) / 4;
        for (int w = lane; w < words; w += 64) {
            const int o = head + 4 * w;
            const uint32_t v = uint32_t(L.out[o]) | (uint32_t(L.out[o + 1]) << 8) | (uint32_t(L.out[o + 2]) << 16) | (uint32_t(L.out[o + 3]) << 24);
            *reinterpret_cast<uint32_t*>(dst + o) = v;
        }
        for (int o = head + 4 * words + lane; o < n; o += 64) dst[o] = L.out[o];
    }
#ifdef SK_IS_TIMING
    __syncthreads();
    IS_DBG_ADD(0, clock64() - t_start);
    IS_DBG_ADD(5, clock64() - t_decoded);
#endif
    if (lane == 0) a.status[blk_i] = err;
}

// ---------------------------------------------------------------------------------------------------------------------
// BAM records (SAM spec 4.2): block_size, refID, pos, l_read_name, mapq, bin, n_cigar_op, flag, l_seq, next_refID, next_pos, tlen,
// read_name, cigar[n_cigar_op] (len << 4 | op), seq[(l_seq+1)/2] (4 bits per base, high nibble first), qual[l_seq]

struct DecodeArgs
{
    const uint8_t* stream;   // inflated BAM bytes (device)
    const int64_t* rec_off;  // [n_records] offset of each record's block_size field
    int32_t n_records;
    const int64_t* read_off; // [n_records+1] into read_code / read_qual
    const int64_t* path_off; // [n_records+1] into path
    sk_bam_record* rec;      // [n_records]
    uint8_t* read_code;
    uint8_t* read_qual;
    sk_path_seg* path;
};

__device__ __forceinline__ int32_t le32(const uint8_t* p) { return int32_t(ld_u32(p)); } // (one unaligned load; the device is little-endian)

__global__ __launch_bounds__(64) void bam_decode_kernel(const DecodeArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_records) return;
    const uint8_t* p = a.stream + a.rec_off[r];
    sk_bam_record o;
    o.ref_id = le32(p + 4);
    o.pos = le32(p + 8);
    const int l_read_name = p[12];
    o.mapq = p[13];
    const int n_cigar = int(p[16]) | (int(p[17]) << 8);
    o.flag = uint16_t(unsigned(p[18]) | (unsigned(p[19]) << 8));
    const int32_t l_seq = le32(p + 20);
    o.mate_ref_id = le32(p + 24);
    o.mate_pos = le32(p + 28);
    o.template_size = le32(p + 32);
    o.l_seq = l_seq;
    o.n_cigar = n_cigar;
    o.is_fwd_strand = (o.flag & 0x10u) ? 0 : 1;
    o.pad = 0;
    a.rec[r] = o;
    const uint8_t* cig = p + 36 + l_read_name;
    sk_path_seg* path = a.path + a.path_off[r];
    for (int i = 0; i < n_cigar; ++i) {
        const uint32_t c = uint32_t(le32(cig + 4 * i));
        sk_path_seg s;
        s.type = (c & 15u) + 1u; // BAM_CMATCH.. = 0..8 -> ALIGNPATH::MATCH.. = 1..9 (L/htsapi/align_path_bam_util.cpp)
        s.length = c >> 4;
        path[i] = s;
    }
    const uint8_t* seq = cig + 4 * n_cigar;
    const uint8_t* qual = seq + (l_seq + 1) / 2;
    uint8_t* code = a.read_code + a.read_off[r];
    uint8_t* q = a.read_qual + a.read_off[r];
    // eight bases per trip: 4 packed bytes in, 8 codes out, 8 qualities through -- four memory instructions where the byte loop had
    // twenty-eight (a lane's accesses are a wave-wide scatter each: the count of instructions is the cost)
    int32_t i = 0;
    for (; i + 8 <= l_seq; i += 8) {
        const uint32_t packed = ld_u32(seq + (i >> 1));
        uint64_t codes = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t byte = (packed >> (8 * k)) & 0xffu;
            codes |= uint64_t(byte >> 4) << (16 * k);
            codes |= uint64_t(byte & 15u) << (16 * k + 8);
        }
        st_u64(code + i, codes);
        st_u64(q + i, ld_u64(qual + i));
    }
    for (; i < l_seq; ++i) {
        const uint8_t byte = seq[i >> 1];
        code[i] = (i & 1) ? uint8_t(byte & 15u) : uint8_t(byte >> 4);
        q[i] = qual[i];
    }
}

struct NormArgs
{
    const char* ref;
    int32_t ref_offset, ref_len;
    int32_t n_reads;
    const int64_t* read_off;
    const uint8_t* read_code;
    const int64_t* path_off;
    int32_t* n_seg;
    sk_path_seg* path;
    int32_t* pos;
    uint8_t* changed;
};

__global__ __launch_bounds__(64) void normalize_kernel(const NormArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_reads) return;
    sknorm::Seqs s;
    s.ref = a.ref;
    s.ref_offset = a.ref_offset;
    s.ref_len = a.ref_len;
    s.read_code = a.read_code + a.read_off[r];
    s.read_len = int32_t(a.read_off[r + 1] - a.read_off[r]);
    sknorm::Aln al;
    al.pos = a.pos[r];
    al.path = a.path + a.path_off[r];
    al.n_seg = a.n_seg[r];
    const bool changed = sknorm::normalize_alignment(s, al);
    a.pos[r] = al.pos;
    a.n_seg[r] = al.n_seg;
    a.changed[r] = changed ? 1 : 0;
}

struct FeedBuffers
{
    void* p[10] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    size_t cap[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    int64_t kept_len = -1; // p[8] holds a stream of this many bytes left by sk_bgzf_inflate_prefixed (-1: nothing kept)
    int reserve(const int i, const size_t bytes)
    {
        if (bytes <= cap[i]) return 0;
        if (p[i]) (void)skrt::free_(p[i]);
        p[i] = nullptr;
        cap[i] = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        SK_HIP(skrt::malloc_(&p[i], want));
        cap[i] = want;
        return 0;
    }
};
FeedBuffers& feed_bufs()
{
    static FeedBuffers b;
    return b;
}

} // namespace

extern "C" {

int sk_bgzf_inflate_dev(const uint8_t* dev_data, const int64_t* dev_block_off, const int64_t* dev_out_off, int32_t n_blocks, uint8_t* dev_out,
                        int32_t* dev_status, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (n_blocks < 0) return sk_fail("sk_bgzf_inflate_dev: negative block count");
    if (n_blocks == 0) return 0;
    if (!dev_data || !dev_block_off || !dev_out_off || !dev_out || !dev_status) return sk_fail("sk_bgzf_inflate_dev: null argument");
    InflateArgs a;
    a.data = dev_data;
    a.block_off = dev_block_off;
    a.out_off = dev_out_off;
    a.out = dev_out;
    a.status = dev_status;
    a.n_blocks = n_blocks;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    // a wave per block up to the launch size where blocks in flight beat latency per block: B1s holds two blocks per CU (LDS), 512 at a
    // time at ~2.9 ms a round; B1 takes ~35 ms for anything up to 1.6e4 blocks (a lane each) and 59 ms for 1.3e5 -- the curves cross near
    // 6 000 blocks (profiles/r04_inflate_history.txt).  $SK_INFLATE_KERNEL = thread | wave pins one (tests run every input through both)
    bool wave = n_blocks <= 6144;
    if (const char* e = std::getenv("SK_INFLATE_KERNEL")) wave = (std::strcmp(e, "wave") == 0) ? true : (std::strncmp(e, "thread", 6) == 0) ? false : wave;
    if (wave) {
        bool& attr_set = sk_ctx().inflate_scalar_lds_allowed;
        if (!attr_set) {
            SK_HIP(skrt::funcSetAttribute(reinterpret_cast<const void*>(bgzf_inflate_scalar_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       int(sizeof(InflateScalarLds))));
            attr_set = true;
        }
        SK_LAUNCH(bgzf_inflate_scalar_kernel, dim3(n_blocks), dim3(64), sizeof(InflateScalarLds), st, a);
#ifdef SK_IS_TIMING
        if (std::getenv("SK_INFLATE_TIMING") && n_blocks <= 16384 && !skrt::remote()) {
            SK_HIP(skrt::streamSynchronize(st));
            std::vector<unsigned long long> d(size_t(n_blocks) * 8);
            SK_HIP(hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(is_dbg), d.size() * 8));
            size_t worst = 0;
            unsigned long long sum[8] = {0};
            for (size_t i = 0; i < size_t(n_blocks); ++i) {
                if (d[i * 8] > d[worst * 8]) worst = i;
                for (int k = 0; k < 8; ++k) sum[k] += d[i * 8 + k];
            }
            std::fprintf(stderr, "[inflate timing] slowest block %zu: in literal iterations %llu ticks, in match iterations %llu (up to the copy: %llu)\n", worst,
                         d[worst * 8 + 1], d[worst * 8 + 7], d[worst * 8 + 6]);
            std::fprintf(stderr, "[inflate timing] slowest block %zu: total %llu ticks, symbol loops %llu, crc+store %llu; %llu literals %llu matches -> %llu bytes; "
                         "all %d blocks: total %llu, symbol loops %llu, crc+store %llu, %llu literals %llu matches %llu bytes\n", worst, d[worst * 8],
                         d[worst * 8 + 2], d[worst * 8 + 5], d[worst * 8 + 3], d[worst * 8 + 4], d[worst * 8 + 6], n_blocks, sum[0], sum[2], sum[5], sum[3],
                         sum[4], sum[6]);
        }
#endif
    } else {
        SK_LAUNCH(bgzf_inflate_kernel, dim3((n_blocks + 63) / 64), dim3(64), 0, st, a);
        SK_LAUNCH(bgzf_crc32_kernel, dim3((n_blocks + 63) / 64), dim3(64), 0, st, a);
    }
    SK_HIP(skrt::getLastError());
    return 0;
}

int sk_bgzf_inflate_prefixed(const uint8_t* data, const int64_t* block_off, const int64_t* out_off, int32_t n_blocks,
                             const uint8_t* prefix, int64_t prefix_len, uint8_t* out)
{
    feed_bufs().kept_len = -1;
    if (prefix_len < 0 || (prefix_len > 0 && !prefix)) return sk_fail("sk_bgzf_inflate: bad prefix");
    SK_REQUIRE_INIT();
    skrt::wakeHint();
    if (n_blocks < 0) return sk_fail("sk_bgzf_inflate: negative block count");
    if (n_blocks == 0 && prefix_len == 0) return 0;
    if ((n_blocks > 0 && (!data || !block_off || !out_off)) || !out) return sk_fail("sk_bgzf_inflate: null argument");
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    hipStream_t st = ctx.stream;
    if (n_blocks == 0) {
        // a slice that is only the carried record: `out` and the kept stream are the prefix (the header's contract: out receives
        // prefix_len bytes, then the inflated blocks, and sk_bam_decode_kept decodes from the kept stream)
        FeedBuffers& B0 = feed_bufs();
        if (B0.reserve(8, size_t(prefix_len) + 16)) return 1;
        SK_HIP(skrt::memcpyAsync(B0.p[8], prefix, size_t(prefix_len), hipMemcpyHostToDevice, st));
        if (out != prefix) std::memcpy(out, prefix, size_t(prefix_len));
        SK_HIP(skrt::streamSynchronize(st));
        B0.kept_len = prefix_len;
        return 0;
    }
    const int64_t in_bytes = block_off[n_blocks] - block_off[0], out_bytes = out_off[n_blocks];
    if (in_bytes < 0 || out_bytes < 0 || out_off[0] != 0) return sk_fail("sk_bgzf_inflate: bad offsets");
    FeedBuffers& B = feed_bufs();
    if (B.reserve(0, size_t(in_bytes) + 16) || B.reserve(1, 8 * size_t(n_blocks + 1)) || B.reserve(2, 8 * size_t(n_blocks + 1)) ||
        B.reserve(8, size_t(prefix_len) + size_t(out_bytes) + 16) || B.reserve(4, 4 * size_t(n_blocks)))
        return 1;
    std::vector<int64_t> rel(size_t(n_blocks) + 1);
    for (int i = 0; i <= n_blocks; ++i) rel[size_t(i)] = block_off[i] - block_off[0];
    SK_HIP(skrt::memcpyAsync(B.p[0], data + block_off[0], size_t(in_bytes), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(B.p[1], rel.data(), 8 * size_t(n_blocks + 1), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(B.p[2], out_off, 8 * size_t(n_blocks + 1), hipMemcpyHostToDevice, st));
    if (prefix_len > 0) SK_HIP(skrt::memcpyAsync(B.p[8], prefix, size_t(prefix_len), hipMemcpyHostToDevice, st));
    if (sk_bgzf_inflate_dev(static_cast<uint8_t*>(B.p[0]), static_cast<int64_t*>(B.p[1]), static_cast<int64_t*>(B.p[2]), n_blocks,
                            static_cast<uint8_t*>(B.p[8]) + prefix_len, static_cast<int32_t*>(B.p[4]), st))
        return 1;
    std::vector<int32_t> status(static_cast<size_t>(n_blocks));
    if (prefix_len > 0 && out != prefix) std::memcpy(out, prefix, size_t(prefix_len));
    SK_HIP(skrt::memcpyAsync(out + prefix_len, static_cast<uint8_t*>(B.p[8]) + prefix_len, size_t(out_bytes), hipMemcpyDeviceToHost, st));
    SK_HIP(skrt::memcpyAsync(status.data(), B.p[4], 4 * size_t(n_blocks), hipMemcpyDeviceToHost, st));
    SK_HIP(skrt::streamSynchronize(st));
    for (int i = 0; i < n_blocks; ++i)
        if (status[size_t(i)] != INF_OK) {
            static const char* const what[] = { "", "invalid block type", "invalid stored block", "invalid code", "invalid distance", "output exceeds ISIZE",
                                                "compressed data ends early", "invalid code lengths", "inflated size differs from ISIZE", "not a BGZF block",
                                                "CRC-32 mismatch" };
            return sk_fail(std::string("sk_bgzf_inflate: block ") + std::to_string(i) + ": " + what[status[size_t(i)]]);
        }
    B.kept_len = prefix_len + out_bytes;
    return 0;
}

int sk_bgzf_inflate(const uint8_t* data, const int64_t* block_off, const int64_t* out_off, int32_t n_blocks, uint8_t* out)
{
    const int rc = sk_bgzf_inflate_prefixed(data, block_off, out_off, n_blocks, nullptr, 0, out);
    feed_bufs().kept_len = -1; // (only the prefixed form promises a kept stream)
    return rc;
}


int sk_bam_decode_dev(const uint8_t* dev_stream, const int64_t* dev_rec_off, int32_t n_records, const int64_t* dev_read_off,
                      const int64_t* dev_path_off, sk_bam_record* dev_rec, uint8_t* dev_read_code, uint8_t* dev_read_qual, sk_path_seg* dev_path,
                      void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (n_records < 0) return sk_fail("sk_bam_decode_dev: negative record count");
    if (n_records == 0) return 0;
    if (!dev_stream || !dev_rec_off || !dev_read_off || !dev_path_off || !dev_rec || !dev_read_code || !dev_read_qual || !dev_path)
        return sk_fail("sk_bam_decode_dev: null argument");
    DecodeArgs a;
    a.stream = dev_stream;
    a.rec_off = dev_rec_off;
    a.n_records = n_records;
    a.read_off = dev_read_off;
    a.path_off = dev_path_off;
    a.rec = dev_rec;
    a.read_code = dev_read_code;
    a.read_qual = dev_read_qual;
    a.path = dev_path;
    SK_LAUNCH(bam_decode_kernel, dim3((n_records + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(skrt::getLastError());
    return 0;
}

static int bam_decode_host(const uint8_t* stream, int64_t stream_len, const int64_t* rec_off, int32_t n_records, const int64_t* read_off,
                           const int64_t* path_off, sk_bam_record* rec, uint8_t* read_code, uint8_t* read_qual, sk_path_seg* path, const bool kept)
{
    SK_REQUIRE_INIT();
    if (n_records < 0 || stream_len < 0) return sk_fail("sk_bam_decode: negative count");
    if (n_records == 0) return 0;
    if (!stream || !rec_off || !read_off || !path_off || !rec || !read_code || !read_qual || !path) return sk_fail("sk_bam_decode: null argument");
    // the offsets are the caller's: a record must lie inside the stream with the size its own fields state (sk_bam_scan_records
    // delivers such offsets; anything else would send the kernel outside the buffer)
    for (int32_t i = 0; i < n_records; ++i) {
        const int64_t at = rec_off[i];
        if (at < 0 || at + 36 > stream_len) return sk_fail("sk_bam_decode: record offset outside the stream");
        const uint8_t* p = stream + at;
        const int64_t block_size = int64_t(uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24));
        const int64_t l_seq = int64_t(int32_t(uint32_t(p[20]) | (uint32_t(p[21]) << 8) | (uint32_t(p[22]) << 16) | (uint32_t(p[23]) << 24)));
        const int64_t n_cigar = int64_t(uint32_t(p[16]) | (uint32_t(p[17]) << 8));
        if (block_size < 32 || at + 4 + block_size > stream_len || l_seq < 0 ||
            32 + int64_t(p[12]) + 4 * n_cigar + (l_seq + 1) / 2 + l_seq > block_size || read_off[i + 1] - read_off[i] != l_seq ||
            path_off[i + 1] - path_off[i] != n_cigar)
            return sk_fail("sk_bam_decode: record does not fit its offsets");
    }
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    hipStream_t st = ctx.stream;
    const int64_t n_bases = read_off[n_records], n_segs = path_off[n_records];
    FeedBuffers& B = feed_bufs();
    if (kept && B.kept_len != stream_len) return sk_fail("sk_bam_decode_kept: no stream of this length was kept by sk_bgzf_inflate_prefixed");
    if ((!kept && B.reserve(0, size_t(stream_len) + 16)) || B.reserve(1, 8 * size_t(n_records)) || B.reserve(2, 8 * size_t(n_records + 1)) ||
        B.reserve(5, 8 * size_t(n_records + 1)) || B.reserve(3, sizeof(sk_bam_record) * size_t(n_records)) || B.reserve(4, size_t(n_bases) + 16) ||
        B.reserve(6, size_t(n_bases) + 16) || B.reserve(7, sizeof(sk_path_seg) * size_t(n_segs) + 16))
        return 1;
    if (!kept) SK_HIP(skrt::memcpyAsync(B.p[0], stream, size_t(stream_len), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(B.p[1], rec_off, 8 * size_t(n_records), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(B.p[2], read_off, 8 * size_t(n_records + 1), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(B.p[5], path_off, 8 * size_t(n_records + 1), hipMemcpyHostToDevice, st));
    if (sk_bam_decode_dev(static_cast<uint8_t*>(kept ? B.p[8] : B.p[0]), static_cast<int64_t*>(B.p[1]), n_records, static_cast<int64_t*>(B.p[2]),
                          static_cast<int64_t*>(B.p[5]), static_cast<sk_bam_record*>(B.p[3]), static_cast<uint8_t*>(B.p[4]),
                          static_cast<uint8_t*>(B.p[6]), static_cast<sk_path_seg*>(B.p[7]), st))
        return 1;
    SK_HIP(skrt::memcpyAsync(rec, B.p[3], sizeof(sk_bam_record) * size_t(n_records), hipMemcpyDeviceToHost, st));
    if (n_bases) {
        SK_HIP(skrt::memcpyAsync(read_code, B.p[4], size_t(n_bases), hipMemcpyDeviceToHost, st));
        SK_HIP(skrt::memcpyAsync(read_qual, B.p[6], size_t(n_bases), hipMemcpyDeviceToHost, st));
    }
    if (n_segs) SK_HIP(skrt::memcpyAsync(path, B.p[7], sizeof(sk_path_seg) * size_t(n_segs), hipMemcpyDeviceToHost, st));
    SK_HIP(skrt::streamSynchronize(st));
    return 0;
}

int sk_bam_decode(const uint8_t* stream, int64_t stream_len, const int64_t* rec_off, int32_t n_records, const int64_t* read_off,
                  const int64_t* path_off, sk_bam_record* rec, uint8_t* read_code, uint8_t* read_qual, sk_path_seg* path)
{
    return bam_decode_host(stream, stream_len, rec_off, n_records, read_off, path_off, rec, read_code, read_qual, path, false);
}

int sk_bam_decode_kept(const uint8_t* stream, int64_t stream_len, const int64_t* rec_off, int32_t n_records, const int64_t* read_off,
                       const int64_t* path_off, sk_bam_record* rec, uint8_t* read_code, uint8_t* read_qual, sk_path_seg* path)
{
    return bam_decode_host(stream, stream_len, rec_off, n_records, read_off, path_off, rec, read_code, read_qual, path, true);
}

int sk_normalize_alignments_dev(const char* dev_ref_seq, int32_t ref_offset, int32_t ref_len, int32_t n_reads, const int64_t* dev_read_off,
                                const uint8_t* dev_read_code, const int64_t* dev_path_off, int32_t* dev_n_seg, sk_path_seg* dev_path,
                                int32_t* dev_pos, uint8_t* dev_changed, void* hip_stream)
{
    SK_REQUIRE_INIT();
    skrt::wakeHint();
    if (n_reads < 0 || ref_len < 0) return sk_fail("sk_normalize_alignments_dev: negative count");
    if (n_reads == 0) return 0;
    if (!dev_ref_seq || !dev_read_off || !dev_read_code || !dev_path_off || !dev_n_seg || !dev_path || !dev_pos || !dev_changed)
        return sk_fail("sk_normalize_alignments_dev: null argument");
    NormArgs a;
    a.ref = dev_ref_seq;
    a.ref_offset = ref_offset;
    a.ref_len = ref_len;
    a.n_reads = n_reads;
    a.read_off = dev_read_off;
    a.read_code = dev_read_code;
    a.path_off = dev_path_off;
    a.n_seg = dev_n_seg;
    a.path = dev_path;
    a.pos = dev_pos;
    a.changed = dev_changed;
    SK_LAUNCH(normalize_kernel, dim3((n_reads + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(hip_stream), a);
    SK_HIP(skrt::getLastError());
    return 0;
}

int sk_normalize_alignments(const char* ref_seq, int32_t ref_offset, int32_t ref_len, int32_t n_reads, const int64_t* read_off,
                            const uint8_t* read_code, const int64_t* path_off, int32_t* n_seg, sk_path_seg* path, int32_t* pos, uint8_t* changed)
{
    SK_REQUIRE_INIT();
    if (n_reads < 0 || ref_len < 0) return sk_fail("sk_normalize_alignments: negative count");
    if (n_reads == 0) return 0;
    if (!ref_seq || !read_off || !read_code || !path_off || !n_seg || !path || !pos || !changed) return sk_fail("sk_normalize_alignments: null argument");
    for (int32_t r = 0; r < n_reads; ++r)
        if (n_seg[r] < 0 || int64_t(n_seg[r]) > path_off[r + 1] - path_off[r]) return sk_fail("sk_normalize_alignments: n_seg beyond the read's path slots");
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    hipStream_t st = ctx.stream;
    const int64_t n_bases = read_off[n_reads], n_segs = path_off[n_reads];
    FeedBuffers& B = feed_bufs();
    if (B.reserve(0, size_t(ref_len) + 16) || B.reserve(1, 8 * size_t(n_reads + 1)) || B.reserve(2, size_t(n_bases) + 16) || B.reserve(3, 8 * size_t(n_reads + 1)) ||
        B.reserve(4, 4 * size_t(n_reads)) || B.reserve(5, sizeof(sk_path_seg) * size_t(n_segs) + 16) || B.reserve(6, 4 * size_t(n_reads)) ||
        B.reserve(7, size_t(n_reads) + 16))
        return 1;
    SK_HIP(skrt::memcpyAsync(B.p[0], ref_seq, size_t(ref_len), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(B.p[1], read_off, 8 * size_t(n_reads + 1), hipMemcpyHostToDevice, st));
    if (n_bases) SK_HIP(skrt::memcpyAsync(B.p[2], read_code, size_t(n_bases), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(B.p[3], path_off, 8 * size_t(n_reads + 1), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(B.p[4], n_seg, 4 * size_t(n_reads), hipMemcpyHostToDevice, st));
    if (n_segs) SK_HIP(skrt::memcpyAsync(B.p[5], path, sizeof(sk_path_seg) * size_t(n_segs), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memcpyAsync(B.p[6], pos, 4 * size_t(n_reads), hipMemcpyHostToDevice, st));
    if (sk_normalize_alignments_dev(static_cast<char*>(B.p[0]), ref_offset, ref_len, n_reads, static_cast<int64_t*>(B.p[1]), static_cast<uint8_t*>(B.p[2]),
                                    static_cast<int64_t*>(B.p[3]), static_cast<int32_t*>(B.p[4]), static_cast<sk_path_seg*>(B.p[5]),
                                    static_cast<int32_t*>(B.p[6]), static_cast<uint8_t*>(B.p[7]), st))
        return 1;
    SK_HIP(skrt::memcpyAsync(n_seg, B.p[4], 4 * size_t(n_reads), hipMemcpyDeviceToHost, st));
    if (n_segs) SK_HIP(skrt::memcpyAsync(path, B.p[5], sizeof(sk_path_seg) * size_t(n_segs), hipMemcpyDeviceToHost, st));
    SK_HIP(skrt::memcpyAsync(pos, B.p[6], 4 * size_t(n_reads), hipMemcpyDeviceToHost, st));
    SK_HIP(skrt::memcpyAsync(changed, B.p[7], size_t(n_reads), hipMemcpyDeviceToHost, st));
    SK_HIP(skrt::streamSynchronize(st));
    return 0;
}

} // extern "C"
