// pileup.hip -- row a8: project aligned reads into per-locus basecall columns.
//
// Reference: starling_pos_processor_base::pileup_read_segment (L/starling_common/starling_pos_processor_base.cpp:1127-1421)
// with create_mismatch_filter_map (L/starling_common/starling_read_util.cpp:40-213), qphred_to_mapped_qphred
// (L/blt_util/qscore.hh:104-121), getReadAmbiguousEndLength (L/htsapi/bam_seq_read_util.cpp:29-54), base_call
// (L/blt_common/snp_pos_info.hh:53-123) and, for the cleaned modes, PileupCleaner::CleanPileupFilter
// (L/starling_common/PileupCleaner.cpp:28-66).  Integer/byte work: results are identical to the reference's, including the
// order of the calls inside a column (read order).
//
// The reference appends one basecall at a time to a position-keyed map as each read passes the pileup stage.  Here the
// same columns are built without any ordered scatter:
//   P1 `pileup_read_kernel`   one wave per read, lanes = read positions.  Mismatch-density counts come from the
//        reference's difference array (LDS atomics + a wave scan); every read base gets a 16-bit RECORD in read-major
//        order: the packed base_call, bit 14 = "tier2 stream", bit 15 = "emitted".  Spanning-deletion / submapped
//        counters are order-free and use global atomics.
//   P2 `pileup_column_kernel` one wave per 64 consecutive loci, lanes = loci.  The wave walks the reads overlapping its
//        loci IN READ ORDER (read geometry is wave-uniform, in SGPRs); a lane whose locus falls in a match segment loads
//        that base's record (adjacent lanes read adjacent bases: coalesced) and, if the record belongs to the requested
//        mode, counts it (first launch) or stores it at call_off[locus] + its running count (second launch).  Read order
//        inside a column is therefore exact by construction.
//   Between the two P2 launches an exclusive scan turns counts into CSR offsets; two more scans (prefix max of read
//   ends, suffix min of read begins) bound the read range a wave has to look at.  The scans are rocPRIM's.
//
// Roofline: HBM-bound, ~4 B per read base in P1 (code, quality in; 2-byte record out) and ~2 x 2 B per call in P2.

#include <cstring>

#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>

#include "sk_common.h"

#include <algorithm>
#include <climits>
#include <iterator>
#include <vector>

namespace
{

constexpr int WAVE = 64;
constexpr int P1_WAVES = 4;
constexpr int MAX_READ_LEN = 1024;  // LDS: 4 B delta + 1 B flag per read base
constexpr unsigned REC_TIER2 = 1u << 14, REC_EMIT = 1u << 15;
constexpr int COL_STAGE = 3072;     // calls of one wave's 64 columns staged in LDS (6 KiB); deeper spans store directly

struct PileupArgs
{
    sk_read_batch b;
    sk_pileup_options o;
    const SkTables* tab;
    uint16_t* rec;        // [n_bases]
    int2* span;           // [n_reads] {begin, end} of reads that pile up, {INT_MAX, INT_MIN} otherwise
    const int* maxend;    // [n_reads] prefix max of span.y
    const int* minbegin;  // [n_reads] suffix min of span.x
    uint32_t* count;      // [n_loci + 1]
    const int64_t* call_off;
    uint16_t* calls;
    uint32_t* spandel;
    uint32_t* submapped;
    int n_loci;
    int mode;
    int store; // 0: count, 1: store
};

__device__ __forceinline__ bool seg_match(const uint32_t t) { return t == SK_SEG_MATCH || t == SK_SEG_SEQ_MATCH || t == SK_SEG_SEQ_MISMATCH; }
__device__ __forceinline__ bool seg_read_len(const uint32_t t) { return seg_match(t) || t == SK_SEG_INSERT || t == SK_SEG_SOFT_CLIP; }
__device__ __forceinline__ bool seg_ref_len(const uint32_t t) { return seg_match(t) || t == SK_SEG_DELETE || t == SK_SEG_SKIP; }

__device__ __forceinline__ unsigned ref_code_at(const sk_read_batch& b, const int p)
{
    if (p < b.ref_offset || p >= b.ref_offset + b.ref_len) return SK_BAM_ANY; // reference_contig_segment::get_base -> 'N'
    switch (b.ref_seq[p - b.ref_offset]) {
    case 'A': return SK_BAM_A;
    case 'C': return SK_BAM_C;
    case 'G': return SK_BAM_G;
    case 'T': return SK_BAM_T;
    default: return SK_BAM_ANY;
    }
}

// the same with an unconditional load (clamped index) and selects: straight-line code for the per-position loops
__device__ __forceinline__ unsigned ref_code_at_nobranch(const sk_read_batch& b, const int p)
{
    const int i = p - b.ref_offset;
    const bool inside = (i >= 0) && (i < b.ref_len);
    const int ci = min(max(i, 0), max(b.ref_len - 1, 0));
    const unsigned c = (b.ref_len > 0) ? unsigned(b.ref_seq[ci]) : unsigned('N');
    unsigned code = SK_BAM_ANY;
    code = (c == 'A') ? unsigned(SK_BAM_A) : code;
    code = (c == 'C') ? unsigned(SK_BAM_C) : code;
    code = (c == 'G') ? unsigned(SK_BAM_G) : code;
    code = (c == 'T') ? unsigned(SK_BAM_T) : code;
    return inside ? code : unsigned(SK_BAM_ANY);
}

constexpr int FAST_K = 4; // positions per lane on the short-read path (reads up to 256 bases)

__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, WAVE));
    return v;
}
__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, WAVE));
    return v;
}

// P1, reads of up to 256 bases: every lane keeps its (at most four) read positions in registers -- base code, quality,
// reference position -- so the read is loaded once, the alignment path is walked once, and every record is written once.
// Same arithmetic as the general path below.
__device__ void pileup_read_short(const PileupArgs& a, const int r, const int lane, int* delta)
{
    const int64_t ro = a.b.read_off[r];
    const int L = int(a.b.read_off[r + 1] - ro);
    const int64_t so = a.b.path_off[r];
    const int nseg = int(a.b.path_off[r + 1] - so);
    const sk_path_seg* __restrict__ path = a.b.path + so;
    const int pos = a.b.pos[r];
    const bool fwd = a.b.is_fwd[r] != 0;
    const unsigned mapq = a.b.mapq[r];
    const unsigned level = a.b.map_level[r];
    const sk_pileup_options& o = a.o;

    unsigned code[FAST_K], qual[FAST_K];
#pragma unroll
    for (int k = 0; k < FAST_K; ++k) {
        const int p = lane + WAVE * k;
        code[k] = (p < L) ? a.b.read_code[ro + p] : unsigned(SK_BAM_ANY);
        qual[k] = (p < L) ? a.b.read_qual[ro + p] : 0u;
    }
    if (lane == 0) a.span[r] = make_int2(INT_MAX, INT_MIN);
    auto store_none = [&]() {
#pragma unroll
        for (int k = 0; k < FAST_K; ++k) {
            const int p = lane + WAVE * k;
            if (p < L) a.rec[ro + p] = 0;
        }
    };

    // one walk over the path: totals, edge segments, and for each of the lane's positions its segment
    int ref_len = 0, read_len_path = 0, first_match = nseg, last_match = nseg;
    int refpos[FAST_K];
    bool in_match[FAST_K];
#pragma unroll
    for (int k = 0; k < FAST_K; ++k) {
        refpos[k] = 0;
        in_match[k] = false;
    }
    for (int i = 0; i < nseg; ++i) {
        const uint32_t t = path[i].type;
        const int len = int(path[i].length);
        if (seg_match(t)) {
            if (first_match == nseg) first_match = i;
            last_match = i;
#pragma unroll
            for (int k = 0; k < FAST_K; ++k) {
                const int p = lane + WAVE * k;
                if (p >= read_len_path && p < read_len_path + len) {
                    in_match[k] = true;
                    refpos[k] = pos + ref_len + (p - read_len_path);
                }
            }
        }
        if (seg_ref_len(t)) ref_len += len;
        if (seg_read_len(t)) read_len_path += len;
    }
    if (nseg == 0 || read_len_path != L || ref_len > L + o.largest_total_indel_ref_span_per_read || pos >= o.report_end ||
        pos + ref_len <= o.report_begin) {
        store_none();
        return;
    }
    // ambiguous read end: trailing Ns of a forward read, leading Ns of a reverse read
    int amb;
    {
        int hi_non_n = -1, lo_non_n = L;
#pragma unroll
        for (int k = 0; k < FAST_K; ++k) {
            const int p = lane + WAVE * k;
            if (p < L && code[k] != SK_BAM_ANY) {
                hi_non_n = max(hi_non_n, p);
                lo_non_n = min(lo_non_n, p);
            }
        }
        amb = fwd ? (L - 1 - wave_max(hi_non_n)) : wave_min(lo_non_n);
    }
    int read_begin = 0, read_end = L;
    if (amb > 0) {
        if (fwd) read_end -= amb;
        else read_begin += amb;
    }
    if (o.min_distance_from_read_edge > 0) {
        read_begin += o.min_distance_from_read_edge;
        if (o.min_distance_from_read_edge <= read_end) read_end -= o.min_distance_from_read_edge;
        else read_end = 0;
        if (read_end <= read_begin) {
            store_none();
            return;
        }
    }
    const bool is_submapped = !(level == SK_MAPLEVEL_TIER1 || level == SK_MAPLEVEL_TIER2);
    const bool is_tier1 = (level == SK_MAPLEVEL_TIER1);
    const bool mdf = (o.mismatch_density_flank_size > 0);
    const int fs = o.mismatch_density_flank_size, fs2 = 2 * fs;
    const int delta_size = max(1 + fs2, L) - fs2;

    bool mmk[FAST_K];
#pragma unroll
    for (int k = 0; k < FAST_K; ++k) mmk[k] = false;
    if (!is_submapped && mdf) {
        for (int i = lane; i < delta_size; i += WAVE) delta[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto inc = [&](const int start, const int length) {
            atomicAdd(&delta[max(fs2, start) - fs2], 1);
            if (start + length < delta_size) atomicAdd(&delta[start + length], -1);
        };
        if (lane == 0) { // internal indels
            int read_head = 0;
            for (int i = 0; i < nseg; ++i) {
                const uint32_t t = path[i].type;
                const int len = int(path[i].length);
                const bool edge = (i < first_match) || (i > last_match);
                if (t == SK_SEG_INSERT && !edge) inc(read_head, len);
                if (t == SK_SEG_DELETE && !edge) inc(read_head, 0);
                if (seg_read_len(t)) read_head += len;
            }
        }
#pragma unroll
        for (int k = 0; k < FAST_K; ++k) {
            const int p = lane + WAVE * k;
            const unsigned fc = ref_code_at_nobranch(a.b, refpos[k]);
            if (in_match[k] && p >= read_begin && p < read_end) {
                if (code[k] != fc) {
                    bool cand = false;
                    if (a.b.cand_snv_mask && refpos[k] >= a.b.ref_offset && refpos[k] < a.b.ref_offset + a.b.ref_len) {
                        const unsigned id = code[k] == SK_BAM_A ? 0u : code[k] == SK_BAM_C ? 1u : code[k] == SK_BAM_G ? 2u : code[k] == SK_BAM_T ? 3u : 4u;
                        cand = (id < 4u) && ((a.b.cand_snv_mask[refpos[k] - a.b.ref_offset] >> id) & 1u);
                    }
                    if (!cand) {
                        mmk[k] = true;
                        inc(p, 1);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int carry = 0;
        for (int base = 0; base < delta_size; base += WAVE) {
            const int i = base + lane;
            int v = (i < delta_size) ? delta[i] : 0;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                const int up = __shfl_up(v, d, WAVE);
                if (lane >= d) v += up;
            }
            v += carry;
            if (i < delta_size) delta[i] = v;
            carry = __shfl(v, WAVE - 1, WAVE);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane == 0) a.span[r] = make_int2(pos, pos + ref_len);

    const unsigned adj_mapq = mapq < 5u ? 5u : mapq;
    const bool mapq_adjust = o.is_mapq_adjust && (adj_mapq <= 80u);
    if (is_submapped) { // (per read, hence wave-uniform) no basecalls, only the counter
#pragma unroll
        for (int k = 0; k < FAST_K; ++k) {
            const int p = lane + WAVE * k;
            const bool live = in_match[k] && p >= read_begin && p < read_end && refpos[k] >= o.report_begin && refpos[k] < o.report_end;
            if (live && a.submapped) atomicAdd(&a.submapped[refpos[k] - o.report_begin], 1u);
            if (p < L) a.rec[ro + p] = 0;
        }
    } else {
        // straight-line per position: table and LDS reads use clamped indices, every flag is a select, and the record of a
        // position that does not emit a call is 0
        const uint8_t* __restrict__ mq = a.tab->mappedq[mapq_adjust ? adj_mapq : 0u];
#pragma unroll
        for (int k = 0; k < FAST_K; ++k) {
            const int p = lane + WAVE * k;
            const bool live = in_match[k] && p >= read_begin && p < read_end && refpos[k] >= o.report_begin && refpos[k] < o.report_end;
            const unsigned c = code[k];
            const unsigned id = c == SK_BAM_A ? 0u : c == SK_BAM_C ? 1u : c == SK_BAM_G ? 2u : c == SK_BAM_T ? 3u : 4u;
            const unsigned q0 = qual[k];
            const unsigned qm = mq[q0 > 70u ? 70u : q0];
            const unsigned q = mapq_adjust ? qm : q0;
            const bool base_filter = (c == SK_BAM_ANY) || (int(q) < o.min_basecall_qscore);
            const int del = mdf ? delta[min(delta_size - 1, max(fs, p) - fs)] : 0; // (delta_size >= 1)
            const bool is_call_filter = base_filter || (mdf && o.mismatch_density_max_count < del);
            const bool is_tier2_call_filter =
                base_filter || (mdf && (o.use_tier2_evidence ? (o.tier2_mismatch_density_max_count < del) : (o.mismatch_density_max_count < del)));
            const bool nmm = mdf && (del - int(mmk[k])) > 0;
            const bool current = is_tier1 ? is_call_filter : is_tier2_call_filter;
            const bool tscf = is_tier1 && is_call_filter && !is_tier2_call_filter;
            const unsigned qb = q > 63u ? 63u : q;
            const unsigned rec = qb | (id << 6) | (fwd ? 1u << 10 : 0u) | (nmm ? 1u << 11 : 0u) | (current ? 1u << 12 : 0u) |
                                 (tscf ? 1u << 13 : 0u) | (is_tier1 ? 0u : REC_TIER2) | REC_EMIT;
            if (p < L) a.rec[ro + p] = uint16_t(live ? rec : 0u);
        }
    }
    // spanning deletions (order-free counters)
    {
        int ref_head = pos;
        for (int i = 0; i < nseg; ++i) {
            const uint32_t t = path[i].type;
            const int len = int(path[i].length);
            if (t == SK_SEG_DELETE && !((i < first_match) || (i > last_match))) {
                for (int j = lane; j < len; j += WAVE) {
                    const int refp = ref_head + j;
                    if (refp < o.report_begin || refp >= o.report_end) continue;
                    uint32_t* ctr = is_submapped ? a.submapped : a.spandel;
                    if (ctr) atomicAdd(&ctr[refp - o.report_begin], 1u);
                }
            }
            if (seg_ref_len(t)) ref_head += len;
        }
    }
}

// P1: one wave per read
__global__ __launch_bounds__(P1_WAVES* WAVE) void pileup_read_kernel(const PileupArgs a)
{
    __shared__ int s_delta[P1_WAVES][MAX_READ_LEN + 1];
    __shared__ unsigned char s_mm[P1_WAVES][MAX_READ_LEN];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / WAVE);
    const int r = blockIdx.x * P1_WAVES + wave;
    if (r >= a.b.n_reads) return;
    int* delta = s_delta[wave];
    unsigned char* mm = s_mm[wave];

    const int64_t ro = a.b.read_off[r];
    const int L = int(a.b.read_off[r + 1] - ro);
    if (L <= FAST_K * WAVE) {
        pileup_read_short(a, r, lane, delta);
        return;
    }
    const int64_t so = a.b.path_off[r];
    const int nseg = int(a.b.path_off[r + 1] - so);
    const sk_path_seg* __restrict__ path = a.b.path + so;
    const int pos = a.b.pos[r];
    const bool fwd = a.b.is_fwd[r] != 0;
    const unsigned mapq = a.b.mapq[r];
    const unsigned level = a.b.map_level[r];
    const sk_pileup_options& o = a.o;

    // every base starts as "not emitted"
    for (int p = lane; p < L; p += WAVE) a.rec[ro + p] = 0;
    if (lane == 0) a.span[r] = make_int2(INT_MAX, INT_MIN);

    // ---- read-level gates (:1144-1196)
    int ref_len = 0, read_len_path = 0;
    int first_match = nseg, last_match = nseg; // get_match_edge_segments
    for (int i = 0; i < nseg; ++i) {
        const uint32_t t = path[i].type;
        if (seg_ref_len(t)) ref_len += int(path[i].length);
        if (seg_read_len(t)) read_len_path += int(path[i].length);
        if (seg_match(t)) {
            if (first_match == nseg) first_match = i;
            last_match = i;
        }
    }
    if (nseg == 0 || read_len_path != L || L > MAX_READ_LEN) return; // empty alignment / malformed (host validates)
    if (ref_len > L + o.largest_total_indel_ref_span_per_read) return;
    if (pos >= o.report_end) return;
    if (pos + ref_len <= o.report_begin) return;

    // ambiguous read end (getReadAmbiguousEndLength): trailing Ns of a forward read, leading Ns of a reverse read
    int amb = 0;
    {
        const uint8_t* __restrict__ code = a.b.read_code + ro;
        if (fwd) {
            int e = L;
            while (e > 0 && code[e - 1] == SK_BAM_ANY) --e;
            amb = L - e;
        } else {
            int s = 0;
            while (s < L && code[s] == SK_BAM_ANY) ++s;
            amb = s;
        }
    }
    int read_begin = 0, read_end = L;
    if (amb > 0) {
        if (fwd) read_end -= amb;
        else read_begin += amb;
    }
    if (o.min_distance_from_read_edge > 0) {
        read_begin += o.min_distance_from_read_edge;
        if (o.min_distance_from_read_edge <= read_end) read_end -= o.min_distance_from_read_edge;
        else read_end = 0;
        if (read_end <= read_begin) return;
    }

    const bool is_submapped = !(level == SK_MAPLEVEL_TIER1 || level == SK_MAPLEVEL_TIER2);
    const bool is_tier1 = (level == SK_MAPLEVEL_TIER1);
    const bool mdf = (o.mismatch_density_flank_size > 0);
    const bool do_mdf = (!is_submapped) && mdf;
    const int fs = o.mismatch_density_flank_size, fs2 = 2 * fs;
    const int delta_size = max(1 + fs2, L) - fs2; // ddata, starling_read_util.cpp:43-56

    // ---- mismatch-density difference array (create_mismatch_filter_map :121-213)
    if (do_mdf) {
        for (int i = lane; i < delta_size; i += WAVE) delta[i] = 0;
        for (int i = lane; i < L; i += WAVE) mm[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto inc = [&](const int start, const int length) { // ddata::inc :58-68
            atomicAdd(&delta[max(fs2, start) - fs2], 1);
            if (start + length < delta_size) atomicAdd(&delta[start + length], -1);
        };
        int read_head = 0, ref_head = pos;
        for (int i = 0; i < nseg; ++i) {
            const uint32_t t = path[i].type;
            const int len = int(path[i].length);
            const bool edge = (i < first_match) || (i > last_match);
            if (t == SK_SEG_INSERT) {
                if (!edge && lane == 0) inc(read_head, len);
                read_head += len;
            } else if (t == SK_SEG_DELETE) {
                if (!edge && lane == 0) inc(read_head, 0);
                ref_head += len;
            } else if (seg_match(t)) {
                for (int j = lane; j < len; j += WAVE) {
                    const int rp = read_head + j;
                    if (rp < read_begin || rp >= read_end) continue;
                    const int refp = ref_head + j;
                    const unsigned rc = a.b.read_code[ro + rp];
                    const unsigned fc = ref_code_at(a.b, refp);
                    if (rc != fc) {
                        bool cand = false; // a candidate SNV of an active region does not count (:180-185)
                        if (a.b.cand_snv_mask && refp >= a.b.ref_offset && refp < a.b.ref_offset + a.b.ref_len) {
                            const unsigned id = rc == SK_BAM_A ? 0u : rc == SK_BAM_C ? 1u : rc == SK_BAM_G ? 2u : rc == SK_BAM_T ? 3u : 4u;
                            cand = (id < 4u) && ((a.b.cand_snv_mask[refp - a.b.ref_offset] >> id) & 1u);
                        }
                        if (!cand) {
                            mm[rp] = 1;
                            inc(rp, 1);
                        }
                    }
                }
                read_head += len;
                ref_head += len;
            } else if (t == SK_SEG_SOFT_CLIP) {
                read_head += len;
            } else if (t == SK_SEG_SKIP) {
                ref_head += len; // (the reference throws on N in this function; spliced reads are outside the DNA path)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ddata::total: inclusive prefix sum, 64 entries per step
        int carry = 0;
        for (int base = 0; base < delta_size; base += WAVE) {
            const int i = base + lane;
            int v = (i < delta_size) ? delta[i] : 0;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) {
                const int up = __shfl_up(v, d, WAVE);
                if (lane >= d) v += up;
            }
            v += carry;
            if (i < delta_size) delta[i] = v;
            carry = __shfl(v, WAVE - 1, WAVE);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    if (lane == 0) a.span[r] = make_int2(pos, pos + ref_len);

    // ---- records and order-free counters
    const unsigned adj_mapq = mapq < 5u ? 5u : mapq;
    const bool mapq_adjust = o.is_mapq_adjust && (adj_mapq <= 80u);
    int read_head = 0, ref_head = pos;
    for (int i = 0; i < nseg; ++i) {
        const uint32_t t = path[i].type;
        const int len = int(path[i].length);
        if (seg_match(t)) {
            for (int j = lane; j < len; j += WAVE) {
                const int rp = read_head + j;
                if (rp < read_begin || rp >= read_end) continue;
                const int refp = ref_head + j;
                if (refp < o.report_begin || refp >= o.report_end) continue;
                if (is_submapped) {
                    if (a.submapped) atomicAdd(&a.submapped[refp - o.report_begin], 1u);
                    continue;
                }
                const unsigned code = a.b.read_code[ro + rp];
                const unsigned id = code == SK_BAM_A ? 0u : code == SK_BAM_C ? 1u : code == SK_BAM_G ? 2u : code == SK_BAM_T ? 3u : 4u;
                unsigned q = a.b.read_qual[ro + rp];
                if (mapq_adjust) q = a.tab->mappedq[adj_mapq][q > 70u ? 70u : q];
                bool is_call_filter = (code == SK_BAM_ANY) || (int(q) < o.min_basecall_qscore);
                bool is_tier2_call_filter = is_call_filter;
                bool nmm = false;
                if (mdf) {
                    const int del = delta[min(delta_size - 1, max(fs, rp) - fs)]; // ddata::get :70-79
                    if (!is_call_filter) {
                        is_call_filter = (o.mismatch_density_max_count < del);
                        is_tier2_call_filter = o.use_tier2_evidence ? (o.tier2_mismatch_density_max_count < del) : is_call_filter;
                    }
                    nmm = (del - int(mm[rp])) > 0;
                }
                const bool current = is_tier1 ? is_call_filter : is_tier2_call_filter;
                const bool tscf = is_tier1 && is_call_filter && !is_tier2_call_filter;
                const unsigned qb = q > 63u ? 63u : q;
                const unsigned bc = qb | (id << 6) | (fwd ? 1u << 10 : 0u) | (nmm ? 1u << 11 : 0u) | (current ? 1u << 12 : 0u) |
                                    (tscf ? 1u << 13 : 0u);
                a.rec[ro + rp] = uint16_t(bc | (is_tier1 ? 0u : REC_TIER2) | REC_EMIT);
            }
        } else if (t == SK_SEG_DELETE) {
            const bool edge = (i < first_match) || (i > last_match); // DNA reads carry no exon pins: edge deletions are dropped
            if (!edge) {
                for (int j = lane; j < len; j += WAVE) {
                    const int refp = ref_head + j;
                    if (refp < o.report_begin || refp >= o.report_end) continue;
                    uint32_t* ctr = is_submapped ? a.submapped : a.spandel;
                    if (ctr) atomicAdd(&ctr[refp - o.report_begin], 1u);
                }
            }
        }
        if (seg_read_len(t)) read_head += len;
        if (seg_ref_len(t)) ref_head += len;
    }
}

// does a record belong to the requested column?  `part` = 1 selects the second half of SK_PILEUP_CLEAN_TIER2
__device__ __forceinline__ bool rec_selected(const unsigned rec, const int mode, const int part)
{
    if (!(rec & REC_EMIT)) return false;
    const bool t2 = rec & REC_TIER2, filt = rec & (1u << 12), tscf = rec & (1u << 13);
    switch (mode) {
    case SK_PILEUP_RAW_TIER1: return !t2;
    case SK_PILEUP_RAW_TIER2: return t2;
    case SK_PILEUP_CLEAN_TIER1: return !t2 && !filt;
    default: return part == 0 ? (!t2 && (!filt || tscf)) : (t2 && !filt); // PileupCleaner.cpp:43-64
    }
}

// P2: one wave per 64 loci
__global__ __launch_bounds__(WAVE) void pileup_column_kernel(const PileupArgs a)
{
    const int lane = threadIdx.x;
    const int l0 = blockIdx.x * WAVE;
    const int l = l0 + lane;
    const int p0 = a.o.report_begin + l0;
    const int p = p0 + lane;
    const int n = a.b.n_reads;
    // reads [lo, hi): lo = first read whose prefix-max end exceeds p0; hi = first read from which every begin >= p0+64
    // 64-ary searches: the lanes probe 64 evenly spaced elements per step, so a range of 2^20 reads takes 4 dependent
    // loads instead of the 20 of a binary search (those 2 x 20 round trips were most of this kernel's time)
    auto first_true = [&](int x, int y, auto&& pred) { // first index in [x, y) with pred, y if none; pred is monotone
        while (x < y) {
            const int span = y - x;
            const int step = (span + WAVE - 1) / WAVE;
            const int m = x + lane * step;
            const bool valid = (m < y);
            const bool v = valid ? pred(m) : true;
            const unsigned long long hit = __ballot(v);                 // lanes past the range report true
            const int f = hit ? __ffsll((long long)hit) - 1 : WAVE;    // first true probe; WAVE: every probe was false
            const int mf = x + f * step;
            if (step == 1) return (mf < y) ? mf : y;
            const int nx = (f == 0) ? x : x + (f - 1) * step + 1;
            y = (mf < y) ? mf : y;
            x = nx;
        }
        return y;
    };
    const int lo = first_true(0, n, [&](const int m) { return a.maxend[m] > p0; });
    const int hi = first_true(lo, n, [&](const int m) { return a.minbegin[m] >= p0 + WAVE; });
    unsigned cnt = 0;
    const int64_t base = (a.store && l < a.n_loci) ? a.call_off[l] : 0;
    const int parts = (a.mode == SK_PILEUP_CLEAN_TIER2) ? 2 : 1;
    const bool live = (l < a.n_loci);
    // store pass: the wave's 64 columns are one contiguous span of `calls`; it is assembled in LDS and written out with
    // consecutive stores (a lane storing 2 bytes every ~80 bytes costs a 32-byte memory transaction per call)
    __shared__ uint16_t s_col[COL_STAGE];
    int64_t span0 = 0;
    int span_n = 0;
    bool staged = false;
    if (a.store) {
        span0 = a.call_off[min(l0, a.n_loci)];
        const int64_t tot = a.call_off[min(l0 + WAVE, a.n_loci)] - span0;
        staged = (tot <= COL_STAGE);
        span_n = staged ? int(tot) : 0;
    }
    const unsigned lbase = unsigned(base - span0);
    // Index of the record read k of the batch contributes to this lane's locus (-1: none).  A locus lies in at most one
    // match segment of a read.  The segments of reads with up to three segments were decoded by the lane that fetched the
    // read (mb/me/mr: reference begin / end and read begin of its match segments, empty ranges otherwise), so per read
    // the wave only broadcasts nine integers and does three range tests; longer paths are walked segment by segment.
    auto locate = [&](const int2 sp, const int64_t ro_k, const int64_t so_k, const int nseg_k, const int (&mb)[3], const int (&me)[3],
                      const int (&mr)[3], const int k) -> int64_t {
        const int sx = __builtin_amdgcn_readlane(sp.x, k), sy = __builtin_amdgcn_readlane(sp.y, k);
        if (sy <= p0 || sx >= p0 + WAVE) return -1;
        const int64_t ro = (int64_t(__builtin_amdgcn_readlane(int(ro_k >> 32), k)) << 32) |
                           uint32_t(__builtin_amdgcn_readlane(int(ro_k & 0xffffffff), k));
        const int nseg = __builtin_amdgcn_readlane(nseg_k, k);
        int off = -1;
        if (nseg <= 3) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int b = __builtin_amdgcn_readlane(mb[i], k), e = __builtin_amdgcn_readlane(me[i], k);
                const int r = __builtin_amdgcn_readlane(mr[i], k);
                off = (p >= b && p < e) ? r + (p - b) : off;
            }
        } else {
            const int64_t so = (int64_t(__builtin_amdgcn_readlane(int(so_k >> 32), k)) << 32) |
                               uint32_t(__builtin_amdgcn_readlane(int(so_k & 0xffffffff), k));
            int read_head = 0, ref_head = sx;
            for (int i = 0; i < nseg; ++i) {
                const uint32_t t = a.b.path[so + i].type;
                const int len = int(a.b.path[so + i].length);
                if (seg_match(t) && p >= ref_head && p < ref_head + len) off = read_head + (p - ref_head);
                if (seg_read_len(t)) read_head += len;
                if (seg_ref_len(t)) ref_head += len;
                if (ref_head >= p0 + WAVE) break;
            }
        }
        return (live && off >= 0) ? ro + off : int64_t(-1);
    };
    constexpr int RU = 8; // reads whose record loads are in flight together
    for (int part = 0; part < parts; ++part) {
        for (int rb = lo; rb < hi; rb += WAVE) {
            // the lanes fetch the geometry of 64 reads at once (one memory round trip instead of one per read); the
            // loops below read it back lane by lane into scalar registers
            const int rk = rb + lane;
            const bool have = (rk < hi);
            const int2 sp = have ? a.span[rk] : make_int2(INT_MAX, INT_MIN);
            const int64_t ro_k = have ? a.b.read_off[rk] : 0;
            const int64_t so_k = have ? a.b.path_off[rk] : 0;
            const int nseg_k = have ? int(a.b.path_off[rk + 1] - so_k) : 0;
            sk_path_seg g0 = { 0u, 0u }, g1 = { 0u, 0u }, g2 = { 0u, 0u };
            if (nseg_k >= 1) g0 = a.b.path[so_k];
            if (nseg_k >= 2) g1 = a.b.path[so_k + 1];
            if (nseg_k >= 3) g2 = a.b.path[so_k + 2];
            int mb[3], me[3], mr[3];
            {
                int read_head = 0, ref_head = sp.x;
                const sk_path_seg gs[3] = { g0, g1, g2 };
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const uint32_t t = gs[i].type;
                    const int len = int(gs[i].length);
                    const bool is_m = (i < nseg_k) && seg_match(t);
                    mb[i] = is_m ? ref_head : INT_MAX;
                    me[i] = is_m ? ref_head + len : INT_MIN;
                    mr[i] = read_head;
                    if (i < nseg_k && seg_read_len(t)) read_head += len;
                    if (i < nseg_k && seg_ref_len(t)) ref_head += len;
                }
            }
            const int nk = min(WAVE, hi - rb);
            // RU reads at a time: all their record loads are issued before the first is consumed (one dependent load per
            // read was the whole cost of this kernel); the calls are still appended in read order
            for (int k0 = 0; k0 < nk; k0 += RU) {
                int64_t idx[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) idx[u] = (k0 + u < nk) ? locate(sp, ro_k, so_k, nseg_k, mb, me, mr, k0 + u) : -1;
                unsigned rec[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) rec[u] = (idx[u] >= 0) ? unsigned(a.rec[idx[u]]) : 0u;
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    if (idx[u] >= 0 && rec_selected(rec[u], a.mode, part)) {
                        if (a.store) {
                            if (staged) s_col[lbase + cnt] = uint16_t(rec[u] & 0x3fffu);
                            else a.calls[base + cnt] = uint16_t(rec[u] & 0x3fffu);
                        }
                        ++cnt;
                    }
                }
            }
        }
    }
    if (!a.store && l <= a.n_loci) a.count[l] = (l < a.n_loci) ? cnt : 0u;
    if (a.store && staged) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint16_t* __restrict__ dst = a.calls + span0;
        for (int i = lane; i < span_n; i += WAVE) dst[i] = s_col[i];
    }
}

__global__ void span_split_kernel(const int2* span, int* begin, int* end, const int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        begin[i] = span[i].x;
        end[i] = span[i].y;
    }
}

struct MaxOp
{
    __host__ __device__ int operator()(const int a, const int b) const { return a > b ? a : b; }
};
struct MinOp
{
    __host__ __device__ int operator()(const int a, const int b) const { return a < b ? a : b; }
};

inline int64_t align256(const int64_t x) { return (x + 255) & ~int64_t(255); }

struct ScratchLayout
{
    int64_t rec, span, begin, end, maxend, minbegin, count, tmp, tmp_bytes, total;
};

ScratchLayout layout(const int32_t n_reads, const int64_t n_bases, const int32_t n_loci)
{
    ScratchLayout s;
    int64_t o = 0;
    s.rec = o; o += align256(2 * std::max<int64_t>(n_bases, 1));
    s.span = o; o += align256(8 * int64_t(std::max(n_reads, 1)));
    s.begin = o; o += align256(4 * int64_t(std::max(n_reads, 1)));
    s.end = o; o += align256(4 * int64_t(std::max(n_reads, 1)));
    s.maxend = o; o += align256(4 * int64_t(std::max(n_reads, 1)));
    s.minbegin = o; o += align256(4 * int64_t(std::max(n_reads, 1)));
    s.count = o; o += align256(4 * (int64_t(n_loci) + 1));
    size_t t1 = 0, t2 = 0, t3 = 0;
    int* ip = nullptr;
    uint32_t* up = nullptr;
    int64_t* lp = nullptr;
    (void)rocprim::inclusive_scan(nullptr, t1, ip, ip, size_t(std::max(n_reads, 1)), MaxOp());
    (void)rocprim::inclusive_scan(nullptr, t2, std::make_reverse_iterator(ip), std::make_reverse_iterator(ip), size_t(std::max(n_reads, 1)), MinOp());
    (void)rocprim::exclusive_scan(nullptr, t3, up, lp, int64_t(0), size_t(n_loci) + 1, rocprim::plus<int64_t>());
    s.tmp = o;
    s.tmp_bytes = int64_t(std::max(t1, std::max(t2, t3))) + 256;
    o += align256(s.tmp_bytes);
    s.total = o;
    return s;
}

} // namespace

extern "C" {

void sk_pileup_options_default(sk_pileup_options* o)
{
    o->min_basecall_qscore = 17;
    o->mismatch_density_flank_size = 20;
    o->mismatch_density_max_count = 2;
    o->use_tier2_evidence = 0;
    o->tier2_mismatch_density_max_count = 10;
    o->is_mapq_adjust = 1;
    o->min_distance_from_read_edge = 0;
    o->largest_total_indel_ref_span_per_read = 49;
    o->report_begin = 0;
    o->report_end = 0;
}

int64_t sk_pileup_scratch_bytes(const int32_t n_reads, const int64_t n_bases, const int32_t n_loci)
{
    if (n_reads < 0 || n_bases < 0 || n_loci < 0) return -1;
    return layout(n_reads, n_bases, n_loci).total;
}

int sk_pileup_reads_dev(const sk_read_batch* b, const int64_t n_bases, const sk_pileup_options* opt, const int mode,
                        sk_pileup_columns* out, void* dev_scratch, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (!b || !opt || !out || !dev_scratch) return sk_fail("sk_pileup_reads_dev: null argument");
    if (mode < SK_PILEUP_RAW_TIER1 || mode > SK_PILEUP_CLEAN_TIER2) return sk_fail("sk_pileup_reads_dev: unknown mode");
    if (opt->report_end < opt->report_begin || out->n_loci != opt->report_end - opt->report_begin)
        return sk_fail("sk_pileup_reads_dev: n_loci must equal report_end - report_begin");
    if (b->n_reads < 0 || n_bases < 0) return sk_fail("sk_pileup_reads_dev: negative count");
    if (!out->call_off || !out->calls) return sk_fail("sk_pileup_reads_dev: null output");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const int n_loci = out->n_loci;
    const ScratchLayout L = layout(b->n_reads, n_bases, n_loci);
    char* base = static_cast<char*>(dev_scratch);
    PileupArgs a;
    a.b = *b;
    a.o = *opt;
    a.tab = sk_ctx().dev_tables;
    a.rec = reinterpret_cast<uint16_t*>(base + L.rec);
    a.span = reinterpret_cast<int2*>(base + L.span);
    int* d_begin = reinterpret_cast<int*>(base + L.begin);
    int* d_end = reinterpret_cast<int*>(base + L.end);
    int* d_maxend = reinterpret_cast<int*>(base + L.maxend);
    int* d_minbegin = reinterpret_cast<int*>(base + L.minbegin);
    a.maxend = d_maxend;
    a.minbegin = d_minbegin;
    a.count = reinterpret_cast<uint32_t*>(base + L.count);
    a.call_off = out->call_off;
    a.calls = out->calls;
    a.spandel = out->spandel_count;
    a.submapped = out->submapped_count;
    a.n_loci = n_loci;
    a.mode = mode;
    a.store = 0;
    void* tmp = base + L.tmp;
    size_t tmp_bytes = size_t(L.tmp_bytes);

    if (a.spandel && n_loci) SK_HIP(hipMemsetAsync(a.spandel, 0, 4 * size_t(n_loci), st));
    if (a.submapped && n_loci) SK_HIP(hipMemsetAsync(a.submapped, 0, 4 * size_t(n_loci), st));
    if (b->n_reads > 0) {
        hipLaunchKernelGGL(pileup_read_kernel, dim3((b->n_reads + P1_WAVES - 1) / P1_WAVES), dim3(P1_WAVES * WAVE), 0, st, a);
        hipLaunchKernelGGL(span_split_kernel, dim3((b->n_reads + 255) / 256), dim3(256), 0, st, a.span, d_begin, d_end, b->n_reads);
        SK_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, d_end, d_maxend, size_t(b->n_reads), MaxOp(), st));
        tmp_bytes = size_t(L.tmp_bytes);
        SK_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, std::make_reverse_iterator(d_begin + b->n_reads),
                                       std::make_reverse_iterator(d_minbegin + b->n_reads), size_t(b->n_reads), MinOp(), st));
    }
    const int blocks = (n_loci + 1 + WAVE - 1) / WAVE; // the extra locus carries the total through the scan
    hipLaunchKernelGGL(pileup_column_kernel, dim3(blocks), dim3(WAVE), 0, st, a);
    tmp_bytes = size_t(L.tmp_bytes);
    SK_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, a.count, out->call_off, int64_t(0), size_t(n_loci) + 1, rocprim::plus<int64_t>(), st));
    // capacity check needs the total on the host
    int64_t total = 0;
    SK_HIP(hipMemcpyAsync(&total, out->call_off + n_loci, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    SK_HIP(hipStreamSynchronize(st));
    if (total > out->capacity) return sk_fail("sk_pileup_reads_dev: calls capacity too small");
    a.store = 1;
    if (total > 0) hipLaunchKernelGGL(pileup_column_kernel, dim3(blocks), dim3(WAVE), 0, st, a);
    SK_HIP(hipGetLastError());
    return 0;
}

int sk_pileup_reads(const sk_read_batch* hb, const sk_pileup_options* opt, const int mode, sk_pileup_columns* out)
{
    SK_REQUIRE_INIT();
    if (!hb || !opt || !out) return sk_fail("sk_pileup_reads: null argument");
    if (hb->n_reads < 0) return sk_fail("sk_pileup_reads: negative n_reads");
    if (opt->report_end < opt->report_begin || out->n_loci != opt->report_end - opt->report_begin)
        return sk_fail("sk_pileup_reads: n_loci must equal report_end - report_begin");
    const int n = hb->n_reads;
    const int n_loci = out->n_loci;
    if (n > 0 && (hb->read_off[0] != 0 || hb->path_off[0] != 0)) return sk_fail("sk_pileup_reads: CSR offsets must start at 0");
    const int64_t n_bases = n ? hb->read_off[n] : 0, n_segs = n ? hb->path_off[n] : 0;
    // what the reference rejects by throwing / exiting
    for (int r = 0; r < n; ++r) {
        const int64_t L = hb->read_off[r + 1] - hb->read_off[r];
        if (L < 0 || hb->path_off[r + 1] < hb->path_off[r]) return sk_fail("sk_pileup_reads: bad CSR offsets");
        if (L > MAX_READ_LEN) return sk_fail("sk_pileup_reads: read longer than 1024 bases");
        int64_t plen = 0;
        for (int64_t i = hb->path_off[r]; i < hb->path_off[r + 1]; ++i) {
            const uint32_t t = hb->path[i].type;
            if (t == SK_SEG_MATCH || t == SK_SEG_SEQ_MATCH || t == SK_SEG_SEQ_MISMATCH || t == SK_SEG_INSERT || t == SK_SEG_SOFT_CLIP)
                plen += hb->path[i].length;
            else if (!(t == SK_SEG_DELETE || t == SK_SEG_HARD_CLIP || t == SK_SEG_SKIP))
                return sk_fail("sk_pileup_reads: Can't handle cigar code"); // starling_read_util.cpp:198-203
        }
        if (hb->path_off[r + 1] > hb->path_off[r] && plen != L)
            return sk_fail("sk_pileup_reads: alignment path does not span its read");
    }
    for (int64_t i = 0; i < n_bases; ++i) {
        const uint8_t c = hb->read_code[i];
        if (!(c == SK_BAM_REF || c == SK_BAM_A || c == SK_BAM_C || c == SK_BAM_G || c == SK_BAM_T || c == SK_BAM_ANY))
            return sk_fail("sk_pileup_reads: unsupported BAM base code"); // bam_seq_code_to_id base_error, bam_seq.hh:145-147
        if (opt->is_mapq_adjust && hb->read_qual[i] > 70)
            return sk_fail("Attempting to lookup basecall quality score " + std::to_string(int(hb->read_qual[i])) +
                           " which exceeds the maximum cached basecall quality score of 70");
    }

    SkContext& ctx = sk_ctx();
    SK_HIP(hipSetDevice(ctx.device));
    const int64_t scratch = sk_pileup_scratch_bytes(n, n_bases, n_loci);
    const int64_t cap = out->capacity;
    struct Item { const void* src; int64_t bytes; int64_t off; };
    std::vector<Item> items = {
        { hb->read_off, 8 * (int64_t(n) + 1), 0 }, { hb->read_code, n_bases, 0 }, { hb->read_qual, n_bases, 0 },
        { hb->path_off, 8 * (int64_t(n) + 1), 0 }, { hb->path, 8 * n_segs, 0 }, { hb->pos, 4 * int64_t(n), 0 },
        { hb->is_fwd, n, 0 }, { hb->mapq, n, 0 }, { hb->map_level, n, 0 }, { hb->ref_seq, hb->ref_len, 0 },
        { hb->cand_snv_mask, hb->cand_snv_mask ? hb->ref_len : 0, 0 },
    };
    int64_t bytes = 0;
    for (Item& it : items) {
        it.off = bytes;
        bytes += align256(std::max<int64_t>(it.bytes, 1));
    }
    const int64_t o_off = bytes; bytes += align256(8 * (int64_t(n_loci) + 1));
    const int64_t o_calls = bytes; bytes += align256(2 * std::max<int64_t>(cap, 1));
    const int64_t o_sd = bytes; bytes += align256(4 * int64_t(std::max(n_loci, 1)));
    const int64_t o_sm = bytes; bytes += align256(4 * int64_t(std::max(n_loci, 1)));
    const int64_t o_scr = bytes; bytes += align256(scratch);
    SkArena ar;
    if (ar.reserve(size_t(bytes) + 256)) return 1;
    char* d = ar.take<char>(size_t(bytes));
    for (const Item& it : items)
        if (it.src && it.bytes > 0) SK_HIP(hipMemcpyAsync(d + it.off, it.src, size_t(it.bytes), hipMemcpyHostToDevice, ctx.stream));
    sk_read_batch db = *hb;
    db.read_off = reinterpret_cast<const int64_t*>(d + items[0].off);
    db.read_code = reinterpret_cast<const uint8_t*>(d + items[1].off);
    db.read_qual = reinterpret_cast<const uint8_t*>(d + items[2].off);
    db.path_off = reinterpret_cast<const int64_t*>(d + items[3].off);
    db.path = reinterpret_cast<const sk_path_seg*>(d + items[4].off);
    db.pos = reinterpret_cast<const int32_t*>(d + items[5].off);
    db.is_fwd = reinterpret_cast<const uint8_t*>(d + items[6].off);
    db.mapq = reinterpret_cast<const uint8_t*>(d + items[7].off);
    db.map_level = reinterpret_cast<const uint8_t*>(d + items[8].off);
    db.ref_seq = d + items[9].off;
    db.cand_snv_mask = hb->cand_snv_mask ? reinterpret_cast<const uint8_t*>(d + items[10].off) : nullptr;
    sk_pileup_columns dc = *out;
    dc.call_off = reinterpret_cast<int64_t*>(d + o_off);
    dc.calls = reinterpret_cast<uint16_t*>(d + o_calls);
    dc.spandel_count = reinterpret_cast<uint32_t*>(d + o_sd);
    dc.submapped_count = reinterpret_cast<uint32_t*>(d + o_sm);
    if (sk_pileup_reads_dev(&db, n_bases, opt, mode, &dc, d + o_scr, ctx.stream)) return 1;
    SK_HIP(hipMemcpyAsync(out->call_off, dc.call_off, 8 * (size_t(n_loci) + 1), hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(hipStreamSynchronize(ctx.stream));
    const int64_t total = out->call_off[n_loci];
    if (total > 0) SK_HIP(hipMemcpyAsync(out->calls, dc.calls, 2 * size_t(total), hipMemcpyDeviceToHost, ctx.stream));
    if (out->spandel_count && n_loci) SK_HIP(hipMemcpyAsync(out->spandel_count, dc.spandel_count, 4 * size_t(n_loci), hipMemcpyDeviceToHost, ctx.stream));
    if (out->submapped_count && n_loci) SK_HIP(hipMemcpyAsync(out->submapped_count, dc.submapped_count, 4 * size_t(n_loci), hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(hipStreamSynchronize(ctx.stream));
    return 0;
}

} // extern "C"
