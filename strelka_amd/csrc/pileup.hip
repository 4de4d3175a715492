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
#include <chrono>
#include <climits>
#include <iterator>
#include <vector>

namespace
{

constexpr int WAVE = 64;
constexpr int P1_WAVES = 4;
constexpr int MAX_READ_LEN = SK_PILEUP_MAX_READ_LEN;  // LDS: 4 B delta + 1 B flag per read base
constexpr unsigned REC_TIER2 = 1u << 14, REC_EMIT = 1u << 15;
constexpr unsigned REC_SUBLIVE = 1u; // (REC_EMIT clear) a live position of a submapped read: no basecall, but it counts for the MAPQ tracker
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
    int r0;    // P1 handles reads [r0, n_reads): the reads before r0 kept their records from an earlier launch (pileup stream)
    // the three-column form of P2 (pileup_column3_kernel): raw tier1, raw tier2 and CleanPileupFilter'ed tier1 columns of the
    // same loci in one pass over the records, plus the MAPQ tracker (insert_mapq_count: every live match position of every read)
    uint32_t* count3[3];
    const int64_t* call_off3[3];
    uint16_t* calls3[3];
    uint32_t* mapq_count;
    uint32_t* mapq_zero;
    unsigned long long* mapq_sumsq;
    // the somatic extension of the three-column form (template flag SOM): a fourth column, CleanPileupFilter(pi, true)
    // (PileupCleaner.cpp:43-64: the tier1 calls that pass or fail only the tier1-specific filter, then the passing tier2 calls),
    // and, parallel to the raw tier1 column, each call's position in its read and the read's length
    // (updateSomaticScoringMetrics' readPos / readLength, starling_pos_processor_base.cpp:1360)
    uint32_t* count4;       // [n_loci + 1] size of the fourth column
    uint32_t* count4a;      // [n_loci + 1] ... of its tier1 part
    const int64_t* call_off4;
    uint16_t* calls4;
    uint32_t* read_pos;     // [tier1 calls] read_pos | read_size << 16, or nullptr
    // the germline EVS extension (template flag EVS): one word per live match position of every read -- submapped reads included --
    // with what updateGermlineScoringMetrics accumulates (starling_pos_processor_base.cpp:1346-1357): base id | mapq << 3 |
    // qscore << 11 | cycle << 18 | min(20, distance from the read edge) << 29 | is_submapped << 34; a locus has mapq_count of them
    const int64_t* evs_off;
    unsigned long long* evs_words;
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
    bool indel_since_match = false, has_internal_indel = false; // an insertion / deletion with match segments on both sides
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
            has_internal_indel = has_internal_indel || indel_since_match;
#pragma unroll
            for (int k = 0; k < FAST_K; ++k) {
                const int p = lane + WAVE * k;
                if (p >= read_len_path && p < read_len_path + len) {
                    in_match[k] = true;
                    refpos[k] = pos + ref_len + (p - read_len_path);
                }
            }
        } else if ((t == SK_SEG_INSERT || t == SK_SEG_DELETE) && first_match != nseg) {
            indel_since_match = true;
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
    // The mismatch-density counts: +1 / -1 marks per mismatch and internal indel in the difference array, then its running sum.  Most
    // reads have neither -- every count is 0 then, and the array, its atomics and the scan (three LDS round trips and 18 shuffles) are
    // skipped for the wave.  (No measurable change on the bench's reads, 2.35 ms per 2^20 either way: P1 spends its time in the ~500
    // vector and ~470 scalar instructions per read around this, profiles/r04_v46_pileup_sq_counters.txt.)
    bool have_delta = false;
    if (!is_submapped && mdf) {
        bool any_mm = false;
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
                    mmk[k] = !cand;
                    any_mm = any_mm || !cand;
                }
            }
        }
        have_delta = has_internal_indel || (__ballot(any_mm) != 0ull); // (wave-uniform)
    }
    if (have_delta) {
        for (int i = lane; i < delta_size; i += WAVE) delta[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto inc = [&](const int start, const int length) {
            atomicAdd(&delta[max(fs2, start) - fs2], 1);
            if (start + length < delta_size) atomicAdd(&delta[start + length], -1);
        };
        if (lane == 0 && has_internal_indel) { // internal indels
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
        for (int k = 0; k < FAST_K; ++k)
            if (mmk[k]) inc(lane + WAVE * k, 1);
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
            if (p < L) a.rec[ro + p] = uint16_t(live ? REC_SUBLIVE : 0u);
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
            const int del = have_delta ? delta[min(delta_size - 1, max(fs, p) - fs)] : 0; // (delta_size >= 1; no marks: every count is 0)
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
    const int r = a.r0 + blockIdx.x * P1_WAVES + wave;
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
                    a.rec[ro + rp] = uint16_t(REC_SUBLIVE);
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

// P2: 64 loci to a block, a lane a locus.  THREE = false: the column of a.mode; THREE = true: the raw tier1, raw tier2 and cleaned tier1
// columns (count3 / call_off3 / calls3, in that order) and the MAPQ tracker in the same walk over the records.
// A block is P2_WAVES waves that SHARE the walk: a wave takes one run of the reads that reach the block's loci (the reads [lo, hi) in
// P2_WAVES consecutive pieces), so a locus' column is its lanes' pieces one after the other -- the count pass adds the waves' counts up,
// the store pass counts its piece first (the same walk, nothing stored), starts its cursors behind the earlier waves' pieces and walks
// again.  With one wave a window's two passes took 2 x 64 us of a push's 310 us on the device: a wave per 64 loci is 128 waves for a
// window, each a chain of ~60 dependent record loads (profiles/r06_v49_stream_window_kernel_stats.csv).
// A launch that fills the device anyway (the one-shot entry point at bench size: 61 000 blocks) keeps one wave to a block: the pieces'
// extra quarter walk and the four searches are then work the device has no idle waves for (2.34 -> 2.90 ms with four).
#ifndef SK_P2_WAVES
#define SK_P2_WAVES 4 // (experiments: 2, 8)
#endif
constexpr int P2_WAVES_MAX = SK_P2_WAVES;
constexpr int P2_SPLIT_BELOW_BLOCKS = 2048; // (8 waves to each of 256 CUs)
template <bool THREE, bool SOM = false, bool EVS = false, int P2_WAVES = P2_WAVES_MAX>
__global__ __launch_bounds__(WAVE* P2_WAVES) void pileup_column_kernel_t(const PileupArgs a)
{
    static_assert(THREE || !(SOM || EVS), "the somatic / EVS columns extend the three-column form");
    const int lane = threadIdx.x & (WAVE - 1);
    const int sub = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / WAVE); // the wave's piece of the reads
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
    const int lo_all = first_true(0, n, [&](const int m) { return a.maxend[m] > p0; });
    const int hi_all = first_true(lo_all, n, [&](const int m) { return a.minbegin[m] >= p0 + WAVE; });
    unsigned cnt = 0, cnt2 = 0, cnt3 = 0, cnt4a = 0, cnt4b = 0;
    unsigned mq_n = 0, mq_zero = 0;
    unsigned long long mq_sq = 0;
    const int64_t* __restrict__ off_a = THREE ? a.call_off3[0] : a.call_off;
    uint16_t* __restrict__ calls_a = THREE ? a.calls3[0] : a.calls;
    const int64_t base = (a.store && l < a.n_loci) ? off_a[l] : 0;
    const int64_t base2 = (THREE && a.store && l < a.n_loci) ? a.call_off3[1][l] : 0;
    const int64_t base3 = (THREE && a.store && l < a.n_loci) ? a.call_off3[2][l] : 0;
    const int64_t base4a = (SOM && a.store && l < a.n_loci) ? a.call_off4[l] : 0;
    const int64_t base4b = (SOM && a.store && l < a.n_loci) ? base4a + int64_t(a.count4a[l]) : 0;
    const int64_t base_e = (EVS && a.store && l < a.n_loci) ? a.evs_off[l] : 0;
    unsigned cnt_e = 0;
    const int parts = (!THREE && a.mode == SK_PILEUP_CLEAN_TIER2) ? 2 : 1;
    // (the two-part column -- every read's first-part calls, then every read's second-part calls -- stays one wave's walk)
    const int piece = (parts == 2) ? (hi_all - lo_all) : (hi_all - lo_all + P2_WAVES - 1) / P2_WAVES;
    const int lo = (parts == 2) ? (sub == 0 ? lo_all : hi_all) : min(hi_all, lo_all + sub * piece);
    const int hi = (parts == 2) ? hi_all : min(hi_all, lo + piece);
    const bool live = (l < a.n_loci);
    // store pass: the wave's 64 columns are one contiguous span of `calls`; it is assembled in LDS and written out with
    // consecutive stores (a lane storing 2 bytes every ~80 bytes costs a 32-byte memory transaction per call)
    __shared__ uint16_t s_col[COL_STAGE];
    __shared__ uint16_t s_col3[THREE ? COL_STAGE : 1];
    int64_t span0 = 0, span0c = 0;
    int span_n = 0, span_nc = 0;
    bool staged = false;
    if (a.store) {
        span0 = off_a[min(l0, a.n_loci)];
        const int64_t tot = off_a[min(l0 + WAVE, a.n_loci)] - span0;
        staged = (tot <= COL_STAGE); // (the cleaned column is a subset of the raw tier1 one: it fits whenever that does)
        span_n = staged ? int(tot) : 0;
        if (THREE) {
            span0c = a.call_off3[2][min(l0, a.n_loci)];
            span_nc = staged ? int(a.call_off3[2][min(l0 + WAVE, a.n_loci)] - span0c) : 0;
        }
    }
    const unsigned lbase = unsigned(base - span0);
    const unsigned lbase3 = unsigned(base3 - span0c);
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
    // the wave's walk over its piece of the reads; store = false: only the counters move
    auto walk = [&](const bool store) {
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
            const int mapq_k = (THREE && have) ? int(a.b.mapq[rk]) : 0;
            const int len_k = ((SOM || EVS) && have) ? int(a.b.read_off[rk + 1] - ro_k) : 0;
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
                    if (THREE) {
                        if (idx[u] < 0) continue;
                        const unsigned rc = rec[u];
                        const uint16_t call = uint16_t(rc & 0x3fffu);
                        if (rc & REC_EMIT) {
                            if (rc & REC_TIER2) {
                                if (store) a.calls3[1][base2 + cnt2] = call;
                                ++cnt2;
                            } else {
                                if (store) {
                                    if (staged) s_col[lbase + cnt] = call;
                                    else calls_a[base + cnt] = call;
                                    if (SOM && a.read_pos) {
                                        const int64_t ro_u = (int64_t(__builtin_amdgcn_readlane(int(ro_k >> 32), k0 + u)) << 32) |
                                                             uint32_t(__builtin_amdgcn_readlane(int(ro_k & 0xffffffff), k0 + u));
                                        const unsigned len_u = unsigned(__builtin_amdgcn_readlane(len_k, k0 + u));
                                        a.read_pos[base + cnt] = unsigned(idx[u] - ro_u) | (len_u << 16);
                                    }
                                }
                                ++cnt;
                                if (!(rc & (1u << 12))) {
                                    if (store) {
                                        if (staged) s_col3[lbase3 + cnt3] = call;
                                        else a.calls3[2][base3 + cnt3] = call;
                                    }
                                    ++cnt3;
                                }
                            }
                            if (SOM) { // the fourth column's two parts
                                const bool t2 = rc & REC_TIER2, filt = rc & (1u << 12), tscf = rc & (1u << 13);
                                if (!t2 && (!filt || tscf)) {
                                    if (store) a.calls4[base4a + cnt4a] = call;
                                    ++cnt4a;
                                } else if (t2 && !filt) {
                                    if (store) a.calls4[base4b + cnt4b] = call;
                                    ++cnt4b;
                                }
                            }
                        }
                        if (EVS && store && ((rc & REC_EMIT) || rc == REC_SUBLIVE)) {
                            const int64_t ro_u = (int64_t(__builtin_amdgcn_readlane(int(ro_k >> 32), k0 + u)) << 32) |
                                                 uint32_t(__builtin_amdgcn_readlane(int(ro_k & 0xffffffff), k0 + u));
                            const unsigned len_u = unsigned(__builtin_amdgcn_readlane(len_k, k0 + u));
                            const unsigned mq = unsigned(__builtin_amdgcn_readlane(mapq_k, k0 + u));
                            const unsigned rp = unsigned(idx[u] - ro_u);
                            const unsigned code = a.b.read_code[idx[u]];
                            const unsigned id = code == SK_BAM_A ? 0u : code == SK_BAM_C ? 1u : code == SK_BAM_G ? 2u : code == SK_BAM_T ? 3u : 4u;
                            const unsigned q0 = a.b.read_qual[idx[u]];
                            const unsigned adj = mq < 5u ? 5u : mq; // pileup_read_segment :1181-1184
                            const bool mapq_adjust = a.o.is_mapq_adjust && (adj <= 80u);
                            const unsigned q = mapq_adjust ? unsigned(a.tab->mappedq[adj][q0 > 70u ? 70u : q0]) : q0;
                            const bool sub = (rc == REC_SUBLIVE);
                            const bool fwd = (rc >> 10) & 1u; // (a submapped position has no strand in its record; its cycle is never read)
                            const unsigned cycle = fwd ? rp : (len_u - (rp + 1u));
                            const unsigned edge = min(min(rp, len_u - (rp + 1u)), 20u);
                            // (a submapped position only feeds the MAPQ rank sum: its other fields are left 0)
                            a.evs_words[base_e + cnt_e] =
                                sub ? ((unsigned long long)id | ((unsigned long long)mq << 3) | (1ull << 34))
                                    : ((unsigned long long)id | ((unsigned long long)mq << 3) | ((unsigned long long)q << 11) |
                                       ((unsigned long long)cycle << 18) | ((unsigned long long)edge << 29));
                            ++cnt_e;
                        }
                        if (!store && ((rc & REC_EMIT) || rc == REC_SUBLIVE)) { // MapqTracker::add (L/blt_common/MapqTracker.hh:36-42)
                            const unsigned mq = unsigned(__builtin_amdgcn_readlane(mapq_k, k0 + u));
                            ++mq_n;
                            mq_sq += (unsigned long long)(mq * mq);
                            mq_zero += (mq == 0u) ? 1u : 0u;
                        }
                    } else if (idx[u] >= 0 && rec_selected(rec[u], a.mode, part)) {
                        if (store) {
                            if (staged) s_col[lbase + cnt] = uint16_t(rec[u] & 0x3fffu);
                            else calls_a[base + cnt] = uint16_t(rec[u] & 0x3fffu);
                        }
                        ++cnt;
                    }
                }
            }
        }
    }
    };
    // the waves' counts meet in LDS: [counter][wave][lane]
    __shared__ unsigned s_part[8][P2_WAVES][WAVE];
    __shared__ unsigned long long s_part_sq[P2_WAVES][WAVE];
    auto publish = [&]() {
        s_part[0][sub][lane] = cnt; s_part[1][sub][lane] = cnt2; s_part[2][sub][lane] = cnt3; s_part[3][sub][lane] = cnt4a;
        s_part[4][sub][lane] = cnt4b; s_part[5][sub][lane] = cnt_e; s_part[6][sub][lane] = mq_n; s_part[7][sub][lane] = mq_zero;
        s_part_sq[sub][lane] = mq_sq;
        __syncthreads();
    };
    if (P2_WAVES == 1) {
        walk(a.store != 0);
    } else if (!a.store) {
        walk(false);
        publish();
        if (sub != 0) return;
        cnt = cnt2 = cnt3 = cnt4a = cnt4b = cnt_e = mq_n = mq_zero = 0;
        mq_sq = 0;
#pragma unroll
        for (int w = 0; w < P2_WAVES; ++w) {
            cnt += s_part[0][w][lane]; cnt2 += s_part[1][w][lane]; cnt3 += s_part[2][w][lane]; cnt4a += s_part[3][w][lane];
            cnt4b += s_part[4][w][lane]; cnt_e += s_part[5][w][lane]; mq_n += s_part[6][w][lane]; mq_zero += s_part[7][w][lane];
            mq_sq += s_part_sq[w][lane];
        }
    } else {
        // where this wave's piece of every column starts: behind the pieces of the waves before it
        walk(false);
        publish();
        cnt = cnt2 = cnt3 = cnt4a = cnt4b = cnt_e = 0;
#pragma unroll
        for (int w = 0; w < P2_WAVES; ++w) {
            if (w < sub) {
                cnt += s_part[0][w][lane]; cnt2 += s_part[1][w][lane]; cnt3 += s_part[2][w][lane]; cnt4a += s_part[3][w][lane];
                cnt4b += s_part[4][w][lane];
                cnt_e += s_part[6][w][lane]; // (a locus has one EVS word per MAPQ-tracker entry: counted as mq_n when nothing is stored)
            }
        }
        walk(true);
    }
    if (!a.store && l <= a.n_loci) {
        if (THREE) {
            a.count3[0][l] = (l < a.n_loci) ? cnt : 0u;
            a.count3[1][l] = (l < a.n_loci) ? cnt2 : 0u;
            a.count3[2][l] = (l < a.n_loci) ? cnt3 : 0u;
            if (SOM) {
                a.count4[l] = (l < a.n_loci) ? cnt4a + cnt4b : 0u;
                a.count4a[l] = (l < a.n_loci) ? cnt4a : 0u;
            }
            if (l < a.n_loci) {
                a.mapq_count[l] = mq_n;
                a.mapq_zero[l] = mq_zero;
                a.mapq_sumsq[l] = mq_sq;
            } else {
                a.mapq_count[l] = 0; // (entry n_loci of the counts, as the columns' have)
            }
        } else {
            a.count[l] = (l < a.n_loci) ? cnt : 0u;
        }
    }
    if (a.store && staged) { // (`staged` is the block's: every wave comes here)
        __syncthreads();
        uint16_t* __restrict__ dst = calls_a + span0;
        for (int i = int(threadIdx.x); i < span_n; i += WAVE * P2_WAVES) dst[i] = s_col[i];
        if (THREE) {
            uint16_t* __restrict__ dst3 = a.calls3[2] + span0c;
            for (int i = int(threadIdx.x); i < span_nc; i += WAVE * P2_WAVES) dst3[i] = s_col3[i];
        }
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

// ---- the stream's small steps as three launches.  A window's push used to be ~30 submissions -- five device-to-device copies of the
// carried tail, the span split, two rocPRIM scans over the reads and three to five over the loci (two launches each), four copies of
// counters into the output block -- around the four kernels that do the work.  Here: one launch copies every carried piece, one makes
// both running bounds of the reads' spans, one makes every column's offsets and moves the counters: ~12 submissions per push.  (It
// is 4 % of a window's 0.45 ms, not more: what a push waits for is P1's latency on 2 200 reads -- 0.11 ms, a chain of dependent loads
// per read -- and the 2.5 MB of columns and records going back to the host.)
struct CopySeg
{
    const char* src;
    char* dst;
    int64_t bytes;
};
struct CopyArgs
{
    CopySeg seg[6];
    int n;
};
// blockIdx.y = the piece, grid-stride over it in 16-byte words where both ends allow
__global__ __launch_bounds__(256) void copy_segments_kernel(const CopyArgs a)
{
    const CopySeg s = a.seg[blockIdx.y];
    const int64_t tid = int64_t(blockIdx.x) * 256 + threadIdx.x, stride = int64_t(gridDim.x) * 256;
    if (((reinterpret_cast<uintptr_t>(s.src) | reinterpret_cast<uintptr_t>(s.dst)) & 15u) == 0) {
        const int64_t words = s.bytes / 16;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(s.src);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(s.dst);
        for (int64_t i = tid; i < words; i += stride) dst[i] = src[i];
        for (int64_t i = words * 16 + tid; i < s.bytes; i += stride) s.dst[i] = s.src[i];
    } else {
        for (int64_t i = tid; i < s.bytes; i += stride) s.dst[i] = s.src[i];
    }
}

// a block's inclusive scan of one value per thread (1 024 threads), `op` associative; `carry` joins from the chunk before
template <typename T, typename OP>
__device__ __forceinline__ T block_scan_1024(T v, const OP op, T* s_wave /*[16]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T up = __shfl_up(v, d, 64);
        if (lane >= d) v = op(up, v);
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    if (wave > 0) {
        T pre = s_wave[0];
        for (int w = 1; w < wave; ++w) pre = op(pre, s_wave[w]);
        v = op(pre, v);
    }
    __syncthreads();
    return v;
}

// block 0: maxend[i] = max of the spans' ends up to read i; block 1: minbegin[i] = min of the spans' begins from read i on
__global__ __launch_bounds__(1024) void span_bounds_kernel(const int2* __restrict__ span, int* __restrict__ maxend, int* __restrict__ minbegin, const int n)
{
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    const bool fwd = (blockIdx.x == 0);
    if (threadIdx.x == 0) s_carry = fwd ? INT_MIN : INT_MAX;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int k = base + int(threadIdx.x);           // position in scan order
        const int i = fwd ? k : n - 1 - k;               // the read
        int v = fwd ? INT_MIN : INT_MAX;
        if (k < n) v = fwd ? span[i].y : span[i].x;
        v = fwd ? block_scan_1024(v, MaxOp(), s_wave) : block_scan_1024(v, MinOp(), s_wave);
        const int carry = s_carry;
        v = fwd ? max(v, carry) : min(v, carry);
        if (k < n) (fwd ? maxend : minbegin)[i] = v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = v;
        __syncthreads();
    }
}

struct OffsetsArgs
{
    const uint32_t* count[5]; // per-locus counts, n + 1 entries each (the last is not read)
    int64_t* off[5];          // exclusive sums, n + 1 entries: off[n] = the total
    int n_scans;
    int n;                    // loci
    CopySeg copy[4];          // counters to move into the output block, a block each
    int n_copies;
};
struct PlusOp
{
    __device__ int64_t operator()(const int64_t a, const int64_t b) const { return a + b; }
};
__global__ __launch_bounds__(1024) void column_offsets_kernel(const OffsetsArgs a)
{
    __shared__ int64_t s_wave[16];
    __shared__ int64_t s_carry;
    if (int(blockIdx.x) >= a.n_scans) {
        const CopySeg s = a.copy[int(blockIdx.x) - a.n_scans];
        for (int64_t i = threadIdx.x; i < s.bytes; i += 1024) s.dst[i] = s.src[i];
        return;
    }
    const uint32_t* __restrict__ cnt = a.count[blockIdx.x];
    int64_t* __restrict__ off = a.off[blockIdx.x];
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base <= a.n; base += 1024) {
        const int i = base + int(threadIdx.x);
        const int64_t c = (i < a.n) ? int64_t(cnt[i]) : 0;
        const int64_t incl = block_scan_1024(c, PlusOp(), s_wave) + s_carry;
        if (i <= a.n) off[i] = incl - c;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = incl;
        __syncthreads();
    }
}

struct ScratchLayout
{
    int64_t rec, span, begin, end, maxend, minbegin, count, tmp, tmp_bytes, total;
};

// with_scans: the one-shot pileup's three device-library scans need temporary storage, sized by asking the library -- which asks the HIP
// runtime about the device.  The STREAM does not use those scans (column_offsets_kernel) and must not ask: in a client of the broker
// the question costs ~1 ms per push and wakes a runtime the process is not supposed to have (profiles/r06_v22: 3.8 of a farm's 5.4
// pileup ABI seconds were this line).
ScratchLayout layout(const int32_t n_reads, const int64_t n_bases, const int32_t n_loci, const bool with_scans = true)
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
    if (with_scans) {
        int* ip = nullptr;
        uint32_t* up = nullptr;
        int64_t* lp = nullptr;
        (void)rocprim::inclusive_scan(nullptr, t1, ip, ip, size_t(std::max(n_reads, 1)), MaxOp());
        (void)rocprim::inclusive_scan(nullptr, t2, std::make_reverse_iterator(ip), std::make_reverse_iterator(ip), size_t(std::max(n_reads, 1)), MinOp());
        (void)rocprim::exclusive_scan(nullptr, t3, up, lp, int64_t(0), size_t(n_loci) + 1, rocprim::plus<int64_t>());
    }
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
    return layout(n_reads, n_bases, n_loci, !skrt::remote()).total; // (a broker client never asks the device library: the one-shot pileup is not its to run)
}

int sk_pileup_reads_dev(const sk_read_batch* b, const int64_t n_bases, const sk_pileup_options* opt, const int mode,
                        sk_pileup_columns* out, void* dev_scratch, void* hip_stream)
{
    SK_REQUIRE_INIT();
    if (skrt::remote())
        return sk_fail("sk_pileup_reads: the one-shot pileup uses the device library's scans, which a broker client cannot call; a broker client piles up through "
                       "sk_pileup_stream_* (unset STRELKA_AMD_BROKER for this entry point)");
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
    a.r0 = 0;
    void* tmp = base + L.tmp;
    size_t tmp_bytes = size_t(L.tmp_bytes);

    if (a.spandel && n_loci) SK_HIP(skrt::memsetAsync(a.spandel, 0, 4 * size_t(n_loci), st));
    if (a.submapped && n_loci) SK_HIP(skrt::memsetAsync(a.submapped, 0, 4 * size_t(n_loci), st));
    if (b->n_reads > 0) {
        SK_LAUNCH(pileup_read_kernel, dim3((b->n_reads + P1_WAVES - 1) / P1_WAVES), dim3(P1_WAVES * WAVE), 0, st, a);
        SK_LAUNCH(span_split_kernel, dim3((b->n_reads + 255) / 256), dim3(256), 0, st, a.span, d_begin, d_end, b->n_reads);
        SK_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, d_end, d_maxend, size_t(b->n_reads), MaxOp(), st));
        tmp_bytes = size_t(L.tmp_bytes);
        SK_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, std::make_reverse_iterator(d_begin + b->n_reads),
                                       std::make_reverse_iterator(d_minbegin + b->n_reads), size_t(b->n_reads), MinOp(), st));
    }
    const int blocks = (n_loci + 1 + WAVE - 1) / WAVE; // the extra locus carries the total through the scan
    if (blocks >= P2_SPLIT_BELOW_BLOCKS) SK_LAUNCH((pileup_column_kernel_t<false, false, false, 1>), dim3(blocks), dim3(WAVE), 0, st, a);
    else SK_LAUNCH(pileup_column_kernel_t<false>, dim3(blocks), dim3(WAVE * P2_WAVES_MAX), 0, st, a);
    tmp_bytes = size_t(L.tmp_bytes);
    SK_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, a.count, out->call_off, int64_t(0), size_t(n_loci) + 1, rocprim::plus<int64_t>(), st));
    // capacity check needs the total on the host
    int64_t total = 0;
    SK_HIP(skrt::memcpyAsync(&total, out->call_off + n_loci, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    SK_HIP(skrt::streamSynchronize(st));
    if (total > out->capacity) return sk_fail("sk_pileup_reads_dev: calls capacity too small");
    a.store = 1;
    if (total > 0) {
        if (blocks >= P2_SPLIT_BELOW_BLOCKS) SK_LAUNCH((pileup_column_kernel_t<false, false, false, 1>), dim3(blocks), dim3(WAVE), 0, st, a);
        else SK_LAUNCH(pileup_column_kernel_t<false>, dim3(blocks), dim3(WAVE * P2_WAVES_MAX), 0, st, a);
    }
    SK_HIP(skrt::getLastError());
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
    SK_HIP(skrt::setDevice(ctx.device));
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
        if (it.src && it.bytes > 0) SK_HIP(skrt::memcpyAsync(d + it.off, it.src, size_t(it.bytes), hipMemcpyHostToDevice, ctx.stream));
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
    SK_HIP(skrt::memcpyAsync(out->call_off, dc.call_off, 8 * (size_t(n_loci) + 1), hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(skrt::streamSynchronize(ctx.stream));
    const int64_t total = out->call_off[n_loci];
    if (total > 0) SK_HIP(skrt::memcpyAsync(out->calls, dc.calls, 2 * size_t(total), hipMemcpyDeviceToHost, ctx.stream));
    if (out->spandel_count && n_loci) SK_HIP(skrt::memcpyAsync(out->spandel_count, dc.spandel_count, 4 * size_t(n_loci), hipMemcpyDeviceToHost, ctx.stream));
    if (out->submapped_count && n_loci) SK_HIP(skrt::memcpyAsync(out->submapped_count, dc.submapped_count, 4 * size_t(n_loci), hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(skrt::streamSynchronize(ctx.stream));
    return 0;
}

} // extern "C"

// =====================================================================================================================
// The pileup of one sample as a stream over a genome segment (sk_pileup_stream_*, include/strelka_amd.h).
//
// The reference piles reads up one at a time as its READ_BUFFER stage passes them and genotypes a position when the
// POST_ALIGN stage, largest_total_indel_ref_span_per_read positions behind, gets there
// (L/starling_common/starling_pos_processor_base.cpp:141-224, :810-890).  A caller of this stream hands over the reads of
// one stage window at a time, in read-buffer order, together with the position up to which no later read can add a
// basecall (`final_to`, the POST_ALIGN position of the window's end).  A push
//   * runs P1 on the new reads only: their records join, on the device, those of the earlier reads that still reach past
//     the previous `final_to` (the carried tail -- a few dozen reads: two record buffers used alternately, the tail is copied
//     across), so every read meets the candidate-SNV mask exactly once, as in the reference;
//   * runs the three-column P2 over [begin, end) = the not yet finalised positions below `final_to` that the batch covers:
//     raw tier1 / tier2 columns (snp_pos_info::calls / tier2_calls), the CleanPileupFilter'ed tier1 column, the MAPQ
//     tracker; spanning-deletion and submapped counters are P1's region-wide atomics;
//   * chains a9+a10 (germline_site_fused_kernel) on the cleaned columns where they lie in HBM;
//   * brings everything the host-side position processor keeps per position back in one copy.
// One host synchronisation per push.
// =====================================================================================================================

namespace
{

struct DevBuf
{
    void* p = nullptr;
    size_t cap = 0;
    int need(const size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)skrt::free_(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        if (skrt::malloc_(&p, want) != hipSuccess) return 1;
        cap = want;
        return 0;
    }
    void drop()
    {
        if (p) (void)skrt::free_(p);
        p = nullptr;
        cap = 0;
    }
};

struct PinBuf
{
    void* p = nullptr;
    size_t cap = 0;
    int need(const size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)skrt::hostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        if (skrt::hostMalloc(&p, want) != hipSuccess) return 1;
        cap = want;
        return 0;
    }
    void drop()
    {
        if (p) (void)skrt::hostFree(p);
        p = nullptr;
        cap = 0;
    }
};

__global__ void ref_base_id_kernel(const char* ref_seq, const int ref_offset, const int ref_len, const int begin, const int n,
                                   uint8_t* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = begin + i - ref_offset;
    uint8_t id = 4;
    if (p >= 0 && p < ref_len) {
        const char c = ref_seq[p];
        id = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
    }
    out[i] = id;
}

inline int path_ref_length(const sk_path_seg* path, const int nseg)
{
    int n = 0;
    for (int i = 0; i < nseg; ++i) {
        const uint32_t t = path[i].type;
        if (t == SK_SEG_MATCH || t == SK_SEG_SEQ_MATCH || t == SK_SEG_SEQ_MISMATCH || t == SK_SEG_DELETE || t == SK_SEG_SKIP) n += int(path[i].length);
    }
    return n;
}

} // namespace

namespace
{
// diagnostics ($SK_PILEUP_PUSH_SECONDS): where a push's wall time goes -- checks, packing + submissions, the wait, the carry bookkeeping
struct PushSeconds
{
    bool on = std::getenv("SK_PILEUP_PUSH_SECONDS") != nullptr;
    double check = 0, enqueue = 0, wait = 0, finish = 0, enq_buffers = 0, enq_pack = 0, enq_submit = 0;
    double lap[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    long pushes = 0;
    ~PushSeconds()
    {
        if (on && pushes)
            std::fprintf(stderr, "strelka_amd pileup push seconds: pushes=%ld check=%.4f enqueue=%.4f wait=%.4f finish=%.4f enq_buffers=%.4f enq_pack=%.4f enq_submit=%.4f\n", pushes, check, enqueue, wait, finish,
                         enq_buffers, enq_pack, enq_submit);
        if (on && pushes) {
            std::fprintf(stderr, "strelka_amd pileup push laps:");
            for (int i = 0; i < 10; ++i) std::fprintf(stderr, " %.4f", lap[i]);
            std::fprintf(stderr, "\n");
        }
    }
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
} g_push_seconds;
}

namespace
{

struct InLay { int64_t read_off, path_off, path, pos, is_fwd, mapq, level, code, qual, ploidy, mask, total; };
struct WorkLay { int64_t begin, end, maxend, minbegin, count0, count1, count2, count4, count4a, off2, off4, calls2, calls4, refbase, de, gscr, pods, tmp, total; };
struct OutLay { int64_t off0, off1, clean_n, clean4_n, mq_n, mq_zero, mq_sq, spandel, submapped, geno, summary, runs, calls0, calls1, read_pos, evs_off, evs, total; };

} // namespace

struct sk_pileup_stream
{
    sk_pileup_options opt;
    sk_germline_options gopt;
    bool genotype = false;
    bool somatic = false;      // also build the CleanPileupFilter(pi, true) column (kept on the device)
    bool want_read_pos = false; // ... and return each tier1 call's read position / read length
    bool want_evs = false;      // return the germline EVS words of every live call (sk_pileup_stream_enable_evs_words)
    bool want_runs = false;     // return the non-variant block that would start at every plain site (sk_pileup_stream_set_gvcf_block_options)
    sk_gvcf_block_options gvcf_opt;
    bool poisoned = false;      // a push failed after it had begun to change the stream's state: only begin_region is accepted
    bool in_flight = false;     // between sk_pileup_stream_push_begin and _finish: the device is working on the window
    // region
    bool has_region = false;
    int32_t ref_offset = 0, ref_len = 0;
    int32_t region_begin = 0, region_end = 0;
    DevBuf d_ref, d_mask, d_spandel, d_submapped;
    // carried reads: metadata on the host, records and spans on the device
    std::vector<int32_t> c_len, c_nseg, c_pos;
    std::vector<uint8_t> c_mapq, c_level;
    std::vector<sk_path_seg> c_path;
    int64_t c_bases = 0;
    int64_t c_tail_base = 0; // where the carried reads' records start in d_rec[cur]
    int64_t c_tail_read = 0; // ... and their spans in d_span[cur]
    int cur = 0;
    DevBuf d_rec[2], d_span[2];
    bool has_prev = false;
    int32_t next_begin = 0;
    DevBuf d_in2[2], d_work, d_out; // (two input blocks used alternately: with want_evs the carried reads' bases and qualities are copied across)
    int cur_in = 0;
    // the output block of a push: OUT_BLOCKS of them in rotation, so that a window's arrays outlive the next pushes (strelka_amd.h,
    // sk_pileup_window: the caller reads the bulk of a window -- the EVS words -- where the push left it, or not at all)
    PinBuf h_in, h_out_blocks[SK_PILEUP_WINDOW_LIFETIME + 1];
    int out_block = 0;
    PinBuf& h_out_cur() { return h_out_blocks[out_block]; }
    int64_t pushes = 0, reads_in = 0;
    // the push in flight (stream_enqueue -> stream_finish)
    InLay li;
    WorkLay wl;
    OutLay ol;
    int p_n = 0, p_new = 0, p_loci = 0;
    int64_t p_bases = 0, p_segs = 0;
    int32_t p_begin = 0, p_end = 0, p_F = 0;
};

struct sk_somatic_pileup_stream
{
    sk_pileup_stream* sample[2] = { nullptr, nullptr }; // normal, tumor
    sk_somatic_snv_options sopt;
    bool genotype = false;
    bool tier2 = false;
    DevBuf d_call; // forced flags, somatic records, the wrapper's scratch
    PinBuf h_forced, h_geno;
    bool in_flight = false; // between sk_somatic_pileup_stream_push_begin and _finish
    int p_loci = 0;         // ... the positions of the window in flight
};

namespace
{

// what the reference rejects by throwing / exiting (as sk_pileup_reads)
int stream_check_reads(const sk_pileup_stream* s, const sk_read_batch* reads, const int32_t mask_begin, const int32_t mask_len,
                       const uint8_t* cand_snv_mask)
{
    if (!s->has_region) return sk_fail("sk_pileup_stream_push: no region (sk_pileup_stream_begin_region)");
    if (reads->n_reads < 0) return sk_fail("sk_pileup_stream_push: negative n_reads");
    const int n_new = reads->n_reads;
    if (n_new > 0 && (reads->read_off[0] != 0 || reads->path_off[0] != 0)) return sk_fail("sk_pileup_stream_push: CSR offsets must start at 0");
    const int64_t new_bases = n_new ? reads->read_off[n_new] : 0;
    for (int r = 0; r < n_new; ++r) {
        const int64_t L = reads->read_off[r + 1] - reads->read_off[r];
        if (L < 0 || reads->path_off[r + 1] < reads->path_off[r]) return sk_fail("sk_pileup_stream_push: bad CSR offsets");
        if (L > MAX_READ_LEN) return sk_fail("sk_pileup_stream_push: read longer than 1024 bases");
        int64_t plen = 0;
        for (int64_t i = reads->path_off[r]; i < reads->path_off[r + 1]; ++i) {
            const uint32_t t = reads->path[i].type;
            if (t == SK_SEG_MATCH || t == SK_SEG_SEQ_MATCH || t == SK_SEG_SEQ_MISMATCH || t == SK_SEG_INSERT || t == SK_SEG_SOFT_CLIP)
                plen += reads->path[i].length;
            else if (!(t == SK_SEG_DELETE || t == SK_SEG_HARD_CLIP || t == SK_SEG_SKIP))
                return sk_fail("sk_pileup_stream_push: Can't handle cigar code"); // starling_read_util.cpp:198-203
        }
        if (reads->path_off[r + 1] > reads->path_off[r] && plen != L) return sk_fail("sk_pileup_stream_push: alignment path does not span its read");
    }
    {
        unsigned bad_code = 0, bad_q = 0;
        for (int64_t i = 0; i < new_bases; ++i) {
            const uint8_t c = reads->read_code[i];
            bad_code |= unsigned(!(c == SK_BAM_REF || c == SK_BAM_A || c == SK_BAM_C || c == SK_BAM_G || c == SK_BAM_T || c == SK_BAM_ANY));
            bad_q |= unsigned(reads->read_qual[i] > 70);
        }
        if (bad_code) return sk_fail("sk_pileup_stream_push: unsupported BAM base code"); // bam_seq_code_to_id base_error, bam_seq.hh:145-147
        if (bad_q && s->opt.is_mapq_adjust)
            return sk_fail("Attempting to lookup basecall quality score which exceeds the maximum cached basecall quality score of 70");
        // (without the MAPQ adjustment the raw quality goes into the calls as it is; the EVS word has seven bits for it, the
        // basecall record six: a quality the reference's own caches would throw on further down the path is refused here)
        if (bad_q && s->want_evs) return sk_fail("sk_pileup_stream_push: basecall quality above 70 with the EVS column on");
    }
    if (mask_len < 0 || (mask_len > 0 && (!cand_snv_mask || mask_begin < s->ref_offset || mask_begin + mask_len > s->ref_offset + s->ref_len)))
        return sk_fail("sk_pileup_stream_push: candidate-SNV mask window outside the reference segment");
    return 0;
}

// lowest start / highest end of the carried and the new reads (INT_MAX / INT_MIN when there are none); fails when a new read
// reaches positions an earlier push declared final
int stream_extent(const sk_pileup_stream* s, const sk_read_batch* reads, int32_t* lowest_out, int32_t* highest_out)
{
    int32_t lowest = INT_MAX, highest = INT_MIN;
    {
        int64_t so = 0;
        for (size_t i = 0; i < s->c_len.size(); ++i) {
            lowest = std::min(lowest, s->c_pos[i]);
            highest = std::max(highest, s->c_pos[i] + path_ref_length(s->c_path.data() + so, s->c_nseg[i]));
            so += s->c_nseg[i];
        }
    }
    for (int r = 0; r < reads->n_reads; ++r) {
        const int nseg = int(reads->path_off[r + 1] - reads->path_off[r]);
        if (nseg == 0) continue;
        const int e = reads->pos[r] + path_ref_length(reads->path + reads->path_off[r], nseg);
        if (s->has_prev && reads->pos[r] < s->next_begin && e > s->region_begin) {
            // (the reference's own guard is validate_new_pos_value, starling_pos_processor_base.cpp:1326: a basecall behind the
            // POST_ALIGN stage throws)
            bool any_match_before = false;
            int ref_head = reads->pos[r];
            for (int64_t i = reads->path_off[r]; i < reads->path_off[r + 1] && !any_match_before; ++i) {
                const uint32_t t = reads->path[i].type;
                const bool m = (t == SK_SEG_MATCH || t == SK_SEG_SEQ_MATCH || t == SK_SEG_SEQ_MISMATCH);
                if ((m || t == SK_SEG_DELETE) && ref_head < s->next_begin && ref_head >= s->region_begin) any_match_before = true;
                if (m || t == SK_SEG_DELETE || t == SK_SEG_SKIP) ref_head += int(reads->path[i].length);
            }
            if (any_match_before) return sk_fail("sk_pileup_stream_push: a read reaches positions that an earlier push declared final");
        }
        lowest = std::min(lowest, reads->pos[r]);
        highest = std::max(highest, e);
    }
    *lowest_out = lowest;
    *highest_out = highest;
    return 0;
}

// the range a push finalises: the not yet final positions below F that the reads (extent lowest..highest) cover
void stream_range(const sk_pileup_stream* s, const int32_t lowest, const int32_t highest, const int32_t F, int32_t* begin_out, int32_t* end_out)
{
    int32_t begin = s->next_begin, end = s->next_begin;
    if (lowest != INT_MAX) {
        begin = std::max(s->region_begin, s->has_prev ? std::max(s->next_begin, lowest) : lowest);
        end = std::max(begin, std::min(F, highest));
    }
    if (begin > F) begin = end = std::max(s->next_begin, std::min(begin, F));
    *begin_out = begin;
    *end_out = end;
}

// everything of a push up to and including the copy of its output block to the host, on the context's stream, no wait
int stream_enqueue(sk_pileup_stream* s, const sk_read_batch* reads, const int32_t largest_total_indel_ref_span_per_read,
                   const int32_t mask_begin, const int32_t mask_len, const uint8_t* cand_snv_mask, const int32_t F, const int32_t begin,
                   const int32_t end, const int32_t ploidy_begin, const int32_t ploidy_len, const uint8_t* ploidy)
{
    SkContext& ctx = sk_ctx();
    hipStream_t st = ctx.stream;
    s->opt.largest_total_indel_ref_span_per_read = largest_total_indel_ref_span_per_read;
    const int n_new = reads->n_reads;
    const int64_t new_bases = n_new ? reads->read_off[n_new] : 0, new_segs = n_new ? reads->path_off[n_new] : 0;
    const int n_c = int(s->c_len.size());
    const int n = n_c + n_new;
    const int64_t c_segs = int64_t(s->c_path.size());
    const int64_t n_bases = s->c_bases + new_bases, n_segs = c_segs + new_segs;
    const int n_loci = end - begin;
    s->p_n = n; s->p_new = n_new; s->p_loci = n_loci; s->p_bases = n_bases; s->p_segs = n_segs;
    s->p_begin = begin; s->p_end = end; s->p_F = F;

    // ---- device input block: carried metadata + the new reads
    const InLay prev_li = s->li;
    const int prev_in = s->cur_in;
    s->cur_in ^= 1;
    DevBuf& d_in = s->d_in2[s->cur_in];
    InLay& li = s->li;
    {
        int64_t o = 0;
        li.read_off = o; o += align256(8 * (int64_t(n) + 1));
        li.path_off = o; o += align256(8 * (int64_t(n) + 1));
        li.path = o; o += align256(8 * std::max<int64_t>(n_segs, 1));
        li.pos = o; o += align256(4 * int64_t(std::max(n, 1)));
        li.is_fwd = o; o += align256(std::max(n, 1));
        li.mapq = o; o += align256(std::max(n, 1));
        li.level = o; o += align256(std::max(n, 1));
        li.ploidy = o; o += align256(std::max(n_loci, 1));
        li.code = o; o += align256(std::max<int64_t>(n_bases, 1));
        li.qual = o; o += align256(std::max<int64_t>(n_bases, 1));
        li.mask = o; o += align256(std::max(mask_len, 1));
        li.total = o;
    }
    const double tq0 = g_push_seconds.on ? PushSeconds::now() : 0.0;
    if (s->h_in.need(size_t(li.total)) || d_in.need(size_t(li.total))) return sk_fail("sk_pileup_stream_push: out of memory (input block)");
    const double tq1 = g_push_seconds.on ? PushSeconds::now() : 0.0;
    char* hi = static_cast<char*>(s->h_in.p);
    {
        int64_t* ro = reinterpret_cast<int64_t*>(hi + li.read_off);
        int64_t* po = reinterpret_cast<int64_t*>(hi + li.path_off);
        sk_path_seg* pa = reinterpret_cast<sk_path_seg*>(hi + li.path);
        int32_t* ps = reinterpret_cast<int32_t*>(hi + li.pos);
        uint8_t* fw = reinterpret_cast<uint8_t*>(hi + li.is_fwd);
        uint8_t* mq = reinterpret_cast<uint8_t*>(hi + li.mapq);
        uint8_t* lv = reinterpret_cast<uint8_t*>(hi + li.level);
        int64_t b = 0, sg = 0;
        for (int i = 0; i < n_c; ++i) {
            ro[i] = b; po[i] = sg;
            b += s->c_len[i]; sg += s->c_nseg[i];
            ps[i] = s->c_pos[i]; fw[i] = 0; mq[i] = s->c_mapq[i]; lv[i] = s->c_level[i]; // (P1 runs over the new reads only: r0)
        }
        if (c_segs) std::memcpy(pa, s->c_path.data(), size_t(8 * c_segs));
        for (int r = 0; r < n_new; ++r) {
            ro[n_c + r] = s->c_bases + reads->read_off[r];
            po[n_c + r] = c_segs + reads->path_off[r];
            ps[n_c + r] = reads->pos[r]; fw[n_c + r] = reads->is_fwd[r]; mq[n_c + r] = reads->mapq[r]; lv[n_c + r] = reads->map_level[r];
        }
        ro[n] = n_bases; po[n] = n_segs;
        if (new_segs) std::memcpy(pa + c_segs, reads->path, size_t(8 * new_segs));
        if (new_bases) {
            std::memcpy(hi + li.code + s->c_bases, reads->read_code, size_t(new_bases));
            std::memcpy(hi + li.qual + s->c_bases, reads->read_qual, size_t(new_bases));
        }
        uint8_t* pl = reinterpret_cast<uint8_t*>(hi + li.ploidy);
        for (int i = 0; i < n_loci; ++i) {
            const int64_t k = int64_t(begin) + i - ploidy_begin;
            pl[i] = (ploidy && k >= 0 && k < ploidy_len) ? ploidy[k] : uint8_t(2);
        }
    }
    if (mask_len > 0) std::memcpy(hi + li.mask, cand_snv_mask, size_t(mask_len)); // (through the pinned block: a copy from the caller's pageable memory would wait for the device)
    char* di = static_cast<char*>(d_in.p);
    const double tq2 = g_push_seconds.on ? PushSeconds::now() : 0.0;
    SK_HIP(skrt::memcpyAsync(di, hi, size_t(li.total), hipMemcpyHostToDevice, st));
    struct SubmitLap
    {
        double t0;
        ~SubmitLap()
        {
            if (g_push_seconds.on) g_push_seconds.enq_submit += PushSeconds::now() - t0;
        }
    } submit_lap{ tq2 };
    if (g_push_seconds.on) g_push_seconds.enq_buffers += tq1 - tq0, g_push_seconds.enq_pack += tq2 - tq1;
    double lap_t = tq2;
    auto lap = [&](const int i) {
        if (!g_push_seconds.on) return;
        const double t = PushSeconds::now();
        g_push_seconds.lap[i] += t - lap_t;
        lap_t = t;
    };
    lap(0); // the H2D copy
    CopyArgs carry; // every device-to-device piece of this push, one launch
    carry.n = 0;
    int64_t carry_max = 0;
    auto add_copy = [&](const void* src, void* dst, const int64_t bytes) {
        if (bytes <= 0) return;
        carry.seg[carry.n].src = static_cast<const char*>(src);
        carry.seg[carry.n].dst = static_cast<char*>(dst);
        carry.seg[carry.n].bytes = bytes;
        ++carry.n;
        carry_max = std::max(carry_max, bytes);
    };
    if (s->want_evs && s->c_bases > 0) {
        // the EVS words are made in P2 from the reads' bases and qualities: the carried reads' come over from the previous block
        const char* dp = static_cast<const char*>(s->d_in2[prev_in].p);
        add_copy(dp + prev_li.code + s->c_tail_base, di + li.code, s->c_bases);
        add_copy(dp + prev_li.qual + s->c_tail_base, di + li.qual, s->c_bases);
    }
    if (mask_len > 0) add_copy(di + li.mask, static_cast<char*>(s->d_mask.p) + (mask_begin - s->ref_offset), mask_len);

    // ---- records and spans: the carried tail moves to the front of the other buffer
    const int nxt = s->cur ^ 1;
    if (s->d_rec[nxt].need(size_t(2 * std::max<int64_t>(n_bases, 1))) || s->d_span[nxt].need(size_t(8 * std::max(n, 1))))
        return sk_fail("sk_pileup_stream_push: out of device memory (records)");
    if (s->c_bases > 0) add_copy(static_cast<char*>(s->d_rec[s->cur].p) + 2 * s->c_tail_base, s->d_rec[nxt].p, 2 * s->c_bases);
    if (n_c > 0) add_copy(static_cast<char*>(s->d_span[s->cur].p) + 8 * s->c_tail_read, s->d_span[nxt].p, int64_t(8) * n_c);
    s->cur = nxt;
    if (carry.n > 0) {
        const int bx = int(std::min<int64_t>((carry_max / 16 + 255) / 256 + 1, 64));
        SK_LAUNCH(copy_segments_kernel, dim3(bx, carry.n), dim3(256), 0, st, carry);
    }

    lap(1); // record buffers + carry copies
    // ---- work and output blocks
    const ScratchLayout SL = layout(n, n_bases, n_loci, false); // (its rec / span / count parts and the scans' storage are unused here)
    const bool som = s->somatic;
    WorkLay& wl = s->wl;
    {
        int64_t o = 0;
        wl.begin = o; o += align256(4 * int64_t(std::max(n, 1)));
        wl.end = o; o += align256(4 * int64_t(std::max(n, 1)));
        wl.maxend = o; o += align256(4 * int64_t(std::max(n, 1)));
        wl.minbegin = o; o += align256(4 * int64_t(std::max(n, 1)));
        wl.count0 = o; o += align256(4 * (int64_t(n_loci) + 1));
        wl.count1 = o; o += align256(4 * (int64_t(n_loci) + 1));
        wl.count2 = o; o += align256(4 * (int64_t(n_loci) + 1));
        wl.count4 = o; o += align256(som ? 4 * (int64_t(n_loci) + 1) : 0);
        wl.count4a = o; o += align256(som ? 4 * (int64_t(n_loci) + 1) : 0);
        wl.off2 = o; o += align256(8 * (int64_t(n_loci) + 1));
        wl.off4 = o; o += align256(som ? 8 * (int64_t(n_loci) + 1) : 0);
        wl.calls2 = o; o += align256(2 * std::max<int64_t>(n_bases, 1));
        wl.calls4 = o; o += align256(som ? 2 * std::max<int64_t>(n_bases, 1) : 0);
        wl.refbase = o; o += align256(std::max(n_loci, 1));
        wl.de = o; o += align256(s->genotype ? 4 * std::max<int64_t>(n_bases, 1) : 0);
        wl.gscr = o; o += align256(s->genotype ? 4 * (n_bases + int64_t(n_loci) + 8) : 0);
        wl.pods = o; o += align256((s->genotype && s->want_runs) ? 17 * int64_t(std::max(n_loci, 1)) + 512 : 0); // (pods, then a tile per 32 of them)
        wl.tmp = o; o += align256(SL.tmp_bytes);
        wl.total = o;
    }
    // the tier2 columns' room: the bases of the reads that are not tier1-mapped (P1 marks a record REC_TIER2 when its read is not; a
    // germline run has few such reads: the block that goes back to the host is that much smaller)
    int64_t n_bases_tier2 = 0;
    for (int i = 0; i < n_c; ++i)
        if (s->c_level[i] != SK_MAPLEVEL_TIER1) n_bases_tier2 += s->c_len[i];
    for (int r = 0; r < n_new; ++r)
        if (reads->map_level[r] != SK_MAPLEVEL_TIER1) n_bases_tier2 += reads->read_off[r + 1] - reads->read_off[r];
    OutLay& ol = s->ol;
    {
        int64_t o = 0;
        ol.off0 = o; o += align256(8 * (int64_t(n_loci) + 1));
        ol.off1 = o; o += align256(8 * (int64_t(n_loci) + 1));
        ol.clean_n = o; o += align256(4 * (int64_t(n_loci) + 1));
        ol.clean4_n = o; o += align256(som ? 4 * (int64_t(n_loci) + 1) : 0);
        ol.mq_n = o; o += align256(4 * (int64_t(n_loci) + 1));
        ol.mq_zero = o; o += align256(4 * int64_t(std::max(n_loci, 1)));
        ol.mq_sq = o; o += align256(8 * int64_t(std::max(n_loci, 1)));
        ol.spandel = o; o += align256(4 * int64_t(std::max(n_loci, 1)));
        ol.submapped = o; o += align256(4 * int64_t(std::max(n_loci, 1)));
        ol.geno = o; o += align256(s->genotype ? int64_t(sizeof(sk_digt_call)) * std::max(n_loci, 1) : 0);
        ol.summary = o; o += align256(s->genotype ? int64_t(sizeof(sk_gvcf_site_summary)) * std::max(n_loci, 1) : 0);
        ol.runs = o; o += align256((s->genotype && s->want_runs) ? int64_t(sizeof(sk_gvcf_run)) * std::max(n_loci, 1) : 0);
        ol.calls0 = o; o += align256(2 * std::max<int64_t>(n_bases, 1));
        ol.calls1 = o; o += align256(2 * std::max<int64_t>(n_bases_tier2, 1)); // (the tier2 column holds the basecalls of tier2-mapped reads only)
        ol.read_pos = o; o += align256(s->want_read_pos ? 4 * std::max<int64_t>(n_bases, 1) : 0);
        ol.evs_off = o; o += align256(s->want_evs ? 8 * (int64_t(n_loci) + 1) : 0);
        ol.evs = o; o += align256(s->want_evs ? 8 * std::max<int64_t>(n_bases, 1) : 0);
        ol.total = o;
    }
    s->out_block = (s->out_block + 1) % (SK_PILEUP_WINDOW_LIFETIME + 1);
    if (s->d_work.need(size_t(wl.total)) || s->d_out.need(size_t(ol.total)) || s->h_out_cur().need(size_t(ol.total)))
        return sk_fail("sk_pileup_stream_push: out of memory (work / output blocks)");
    char* dw = static_cast<char*>(s->d_work.p);
    char* dout = static_cast<char*>(s->d_out.p);
    char* ho = static_cast<char*>(s->h_out_cur().p);
    lap(2); // layouts + work / out buffers

    PileupArgs a;
    std::memset(&a, 0, sizeof(a));
    a.b.n_reads = n;
    a.b.read_off = reinterpret_cast<const int64_t*>(di + li.read_off);
    a.b.read_code = reinterpret_cast<const uint8_t*>(di + li.code);
    a.b.read_qual = reinterpret_cast<const uint8_t*>(di + li.qual);
    a.b.path_off = reinterpret_cast<const int64_t*>(di + li.path_off);
    a.b.path = reinterpret_cast<const sk_path_seg*>(di + li.path);
    a.b.pos = reinterpret_cast<const int32_t*>(di + li.pos);
    a.b.is_fwd = reinterpret_cast<const uint8_t*>(di + li.is_fwd);
    a.b.mapq = reinterpret_cast<const uint8_t*>(di + li.mapq);
    a.b.map_level = reinterpret_cast<const uint8_t*>(di + li.level);
    a.b.ref_seq = static_cast<const char*>(s->d_ref.p);
    a.b.ref_offset = s->ref_offset;
    a.b.ref_len = s->ref_len;
    a.b.cand_snv_mask = static_cast<const uint8_t*>(s->d_mask.p);
    a.o = s->opt; // P1: the region's report range, region-wide counters
    a.tab = ctx.dev_tables;
    a.rec = static_cast<uint16_t*>(s->d_rec[s->cur].p);
    a.span = static_cast<int2*>(s->d_span[s->cur].p);
    a.spandel = static_cast<uint32_t*>(s->d_spandel.p);
    a.submapped = static_cast<uint32_t*>(s->d_submapped.p);
    a.n_loci = s->region_end - s->region_begin;
    a.r0 = n_c;
    if (n_new > 0) SK_LAUNCH(pileup_read_kernel, dim3((n_new + P1_WAVES - 1) / P1_WAVES), dim3(P1_WAVES * WAVE), 0, st, a);

    int* d_maxend = reinterpret_cast<int*>(dw + wl.maxend);
    int* d_minbegin = reinterpret_cast<int*>(dw + wl.minbegin);
    if (n > 0) SK_LAUNCH(span_bounds_kernel, dim3(2), dim3(1024), 0, st, a.span, d_maxend, d_minbegin, n);
    lap(3); // P1 + span bounds
    // P2: the window's columns
    PileupArgs c = a;
    c.o.report_begin = begin;
    c.o.report_end = end;
    c.n_loci = n_loci;
    c.maxend = d_maxend;
    c.minbegin = d_minbegin;
    c.count3[0] = reinterpret_cast<uint32_t*>(dw + wl.count0);
    c.count3[1] = reinterpret_cast<uint32_t*>(dw + wl.count1);
    c.count3[2] = reinterpret_cast<uint32_t*>(dw + wl.count2);
    int64_t* off0 = reinterpret_cast<int64_t*>(dout + ol.off0);
    int64_t* off1 = reinterpret_cast<int64_t*>(dout + ol.off1);
    int64_t* off2 = reinterpret_cast<int64_t*>(dw + wl.off2);
    c.call_off3[0] = off0; c.call_off3[1] = off1; c.call_off3[2] = off2;
    c.calls3[0] = reinterpret_cast<uint16_t*>(dout + ol.calls0);
    c.calls3[1] = reinterpret_cast<uint16_t*>(dout + ol.calls1);
    c.calls3[2] = reinterpret_cast<uint16_t*>(dw + wl.calls2);
    c.mapq_count = reinterpret_cast<uint32_t*>(dout + ol.mq_n);
    c.mapq_zero = reinterpret_cast<uint32_t*>(dout + ol.mq_zero);
    c.mapq_sumsq = reinterpret_cast<unsigned long long*>(dout + ol.mq_sq);
    if (som) {
        c.count4 = reinterpret_cast<uint32_t*>(dw + wl.count4);
        c.count4a = reinterpret_cast<uint32_t*>(dw + wl.count4a);
        c.call_off4 = reinterpret_cast<int64_t*>(dw + wl.off4);
        c.calls4 = reinterpret_cast<uint16_t*>(dw + wl.calls4);
        c.read_pos = s->want_read_pos ? reinterpret_cast<uint32_t*>(dout + ol.read_pos) : nullptr;
    }
    if (s->want_evs) {
        c.evs_off = reinterpret_cast<const int64_t*>(dout + ol.evs_off);
        c.evs_words = reinterpret_cast<unsigned long long*>(dout + ol.evs);
    }
    c.store = 0;
    const int blocks = (n_loci + 1 + WAVE - 1) / WAVE;
    // (a stream's windows are a few thousand loci: the split form; a caller that pushes a whole region at once gets the one-wave form)
    const bool split = blocks < P2_SPLIT_BELOW_BLOCKS;
    const int p2_threads = split ? WAVE * P2_WAVES_MAX : WAVE;
    void (*const p2)(const PileupArgs) =
        split ? (som ? pileup_column_kernel_t<true, true, false> : (s->want_evs ? pileup_column_kernel_t<true, false, true> : pileup_column_kernel_t<true, false, false>))
              : (som ? pileup_column_kernel_t<true, true, false, 1>
                     : (s->want_evs ? pileup_column_kernel_t<true, false, true, 1> : pileup_column_kernel_t<true, false, false, 1>));
    SK_LAUNCH(p2, dim3(blocks), dim3(p2_threads), 0, st, c);
    {
        // every column's offsets (exclusive sums of n_loci + 1 counts: the last entry is the total) and, in the same launch, the cleaned
        // columns' sizes for the caller's cache validation and the region-wide counters' slice
        OffsetsArgs oa;
        std::memset(&oa, 0, sizeof(oa));
        oa.n = n_loci;
        for (int m = 0; m < 3; ++m) {
            oa.count[oa.n_scans] = c.count3[m];
            oa.off[oa.n_scans++] = const_cast<int64_t*>(c.call_off3[m]);
        }
        if (som) {
            oa.count[oa.n_scans] = c.count4;
            oa.off[oa.n_scans++] = const_cast<int64_t*>(c.call_off4);
        }
        if (s->want_evs) { // a locus has mapq_count words
            oa.count[oa.n_scans] = c.mapq_count;
            oa.off[oa.n_scans++] = const_cast<int64_t*>(c.evs_off);
        }
        auto add = [&](const void* src, void* dst, const int64_t bytes) {
            oa.copy[oa.n_copies].src = static_cast<const char*>(src);
            oa.copy[oa.n_copies].dst = static_cast<char*>(dst);
            oa.copy[oa.n_copies].bytes = bytes;
            ++oa.n_copies;
        };
        add(c.count3[2], dout + ol.clean_n, 4 * (int64_t(n_loci) + 1));
        if (som) add(c.count4, dout + ol.clean4_n, 4 * (int64_t(n_loci) + 1));
        if (n_loci > 0) {
            add(static_cast<char*>(s->d_spandel.p) + 4 * size_t(begin - s->region_begin), dout + ol.spandel, 4 * int64_t(n_loci));
            add(static_cast<char*>(s->d_submapped.p) + 4 * size_t(begin - s->region_begin), dout + ol.submapped, 4 * int64_t(n_loci));
        }
        SK_LAUNCH(column_offsets_kernel, dim3(oa.n_scans + oa.n_copies), dim3(1024), 0, st, oa);
    }
    c.store = 1;
    if (n > 0 && n_loci > 0) SK_LAUNCH(p2, dim3(blocks), dim3(p2_threads), 0, st, c);
    lap(4); // P2 count, offsets, P2 store
    if (n_loci > 0 && (s->genotype || som)) {
        uint8_t* d_refbase = reinterpret_cast<uint8_t*>(dw + wl.refbase);
        SK_LAUNCH(ref_base_id_kernel, dim3((n_loci + 255) / 256), dim3(256), 0, st, static_cast<const char*>(s->d_ref.p), s->ref_offset,
                           s->ref_len, begin, n_loci, d_refbase);
    }
    // a9 + a10 on the cleaned columns, where they are
    if (s->genotype && n_loci > 0) {
        sk_pileup_batch pb;
        std::memset(&pb, 0, sizeof(pb));
        pb.n_loci = n_loci;
        pb.call_off = off2;
        pb.calls = c.calls3[2];
        pb.de = nullptr;
        pb.ref_base = reinterpret_cast<const uint8_t*>(dw + wl.refbase);
        pb.ploidy = reinterpret_cast<const uint8_t*>(di + li.ploidy);
        if (sk_site_digt_call_fused_dev(&pb, &s->gopt, reinterpret_cast<sk_digt_call*>(dout + ol.geno), reinterpret_cast<float*>(dw + wl.de), 0,
                                        dw + wl.gscr, n_bases, st))
            return 1;
        // ... and what the gVCF writer's block logic reads of each position, from the same column and the record just written
        if (sk_gvcf_site_summaries_dev(&pb, reinterpret_cast<const sk_digt_call*>(dout + ol.geno), reinterpret_cast<sk_gvcf_site_summary*>(dout + ol.summary), st))
            return 1;
        // ... and, from every plain site, the non-variant block the writer would start there
        if (s->want_runs &&
            sk_gvcf_plain_runs_dev(reinterpret_cast<const sk_gvcf_site_summary*>(dout + ol.summary), off2, off0, reinterpret_cast<const uint32_t*>(dout + ol.mq_n),
                                   &s->gvcf_opt, n_loci, dw + wl.pods, reinterpret_cast<sk_gvcf_run*>(dout + ol.runs), st))
            return 1;
    }
    lap(5); // genotypes, summaries, runs
    SK_HIP(skrt::getLastError());
    SK_HIP(skrt::memcpyAsync(ho, dout, size_t(ol.total), hipMemcpyDeviceToHost, st));
    lap(6); // D2H
    return 0;
}

// after the stream has been waited for: what the next push carries, and the window's pointers
void stream_finish(sk_pileup_stream* s, sk_pileup_window* out)
{
    const InLay& li = s->li;
    const OutLay& ol = s->ol;
    const char* hi = static_cast<const char*>(s->h_in.p);
    const char* ho = static_cast<const char*>(s->h_out_cur().p);
    const int n = s->p_n;
    const int32_t F = s->p_F;
    // the reads from the first one that ends past F on
    {
        const int64_t* ro = reinterpret_cast<const int64_t*>(hi + li.read_off);
        const int64_t* po = reinterpret_cast<const int64_t*>(hi + li.path_off);
        const sk_path_seg* pa = reinterpret_cast<const sk_path_seg*>(hi + li.path);
        const int32_t* ps = reinterpret_cast<const int32_t*>(hi + li.pos);
        const uint8_t* mq = reinterpret_cast<const uint8_t*>(hi + li.mapq);
        int i0 = n;
        if (F < s->region_end) {
            for (int i = 0; i < n; ++i) {
                const int nseg = int(po[i + 1] - po[i]);
                if (nseg > 0 && ps[i] + path_ref_length(pa + po[i], nseg) > F) {
                    i0 = i;
                    break;
                }
            }
        }
        std::vector<int32_t> nl, ns, np;
        std::vector<uint8_t> nm, nv;
        const uint8_t* lv = reinterpret_cast<const uint8_t*>(hi + li.level);
        for (int i = i0; i < n; ++i) {
            nl.push_back(int32_t(ro[i + 1] - ro[i]));
            ns.push_back(int32_t(po[i + 1] - po[i]));
            np.push_back(ps[i]);
            nm.push_back(mq[i]);
            nv.push_back(lv[i]);
        }
        std::vector<sk_path_seg> npath(pa + (i0 < n ? po[i0] : s->p_segs), pa + s->p_segs);
        s->c_tail_base = (i0 < n) ? ro[i0] : s->p_bases;
        s->c_tail_read = i0;
        s->c_bases = s->p_bases - s->c_tail_base;
        s->c_len.swap(nl); s->c_nseg.swap(ns); s->c_pos.swap(np); s->c_mapq.swap(nm); s->c_level.swap(nv); s->c_path.swap(npath);
    }
    s->has_prev = true;
    s->next_begin = std::max(s->next_begin, F);
    s->pushes++;
    s->reads_in += s->p_new;

    out->begin = s->p_begin;
    out->end = s->p_end;
    out->tier1_off = reinterpret_cast<const int64_t*>(ho + ol.off0);
    out->tier1_calls = reinterpret_cast<const uint16_t*>(ho + ol.calls0);
    out->tier2_off = reinterpret_cast<const int64_t*>(ho + ol.off1);
    out->tier2_calls = reinterpret_cast<const uint16_t*>(ho + ol.calls1);
    out->spandel_count = reinterpret_cast<const uint32_t*>(ho + ol.spandel);
    out->submapped_count = reinterpret_cast<const uint32_t*>(ho + ol.submapped);
    out->mapq_count = reinterpret_cast<const uint32_t*>(ho + ol.mq_n);
    out->mapq_zero_count = reinterpret_cast<const uint32_t*>(ho + ol.mq_zero);
    out->mapq_sum_square = reinterpret_cast<const uint64_t*>(ho + ol.mq_sq);
    out->clean_count = reinterpret_cast<const uint32_t*>(ho + ol.clean_n);
    out->genotype = s->genotype ? reinterpret_cast<const sk_digt_call*>(ho + ol.geno) : nullptr;
    out->site_summary = s->genotype ? reinterpret_cast<const sk_gvcf_site_summary*>(ho + ol.summary) : nullptr;
    out->gvcf_runs = (s->genotype && s->want_runs) ? reinterpret_cast<const sk_gvcf_run*>(ho + ol.runs) : nullptr;
    out->evs_off = s->want_evs ? reinterpret_cast<const int64_t*>(ho + ol.evs_off) : nullptr;
    out->evs_words = s->want_evs ? reinterpret_cast<const uint64_t*>(ho + ol.evs) : nullptr;
}

void stream_drop(sk_pileup_stream* s)
{
    s->d_ref.drop(); s->d_mask.drop(); s->d_spandel.drop(); s->d_submapped.drop();
    s->d_rec[0].drop(); s->d_rec[1].drop(); s->d_span[0].drop(); s->d_span[1].drop();
    s->d_in2[0].drop(); s->d_in2[1].drop(); s->d_work.drop(); s->d_out.drop();
    s->h_in.drop();
    for (PinBuf& b : s->h_out_blocks) b.drop();
}

} // namespace

extern "C" {

sk_pileup_stream* sk_pileup_stream_create(const sk_pileup_options* opt, const sk_germline_options* genotype_opt)
{
    if (!sk_ctx().ready) {
        sk_fail("strelka_amd: sk_init() has not succeeded");
        return nullptr;
    }
    if (!opt) {
        sk_fail("sk_pileup_stream_create: null options");
        return nullptr;
    }
    sk_pileup_stream* s = new sk_pileup_stream();
    s->opt = *opt;
    if (genotype_opt) {
        s->gopt = *genotype_opt;
        s->genotype = true;
    }
    return s;
}

int sk_pileup_stream_set_gvcf_block_options(sk_pileup_stream* s, const sk_gvcf_block_options* opt)
{
    if (!s) return sk_fail("sk_pileup_stream_set_gvcf_block_options: null stream");
    if (s->somatic) return sk_fail("sk_pileup_stream_set_gvcf_block_options: a germline stream's");
    s->want_runs = (opt != nullptr);
    if (opt) s->gvcf_opt = *opt;
    return 0;
}

int sk_pileup_stream_enable_evs_words(sk_pileup_stream* s, const int enable)
{
    if (!s) return sk_fail("sk_pileup_stream_enable_evs_words: null argument");
    if (s->somatic) return sk_fail("sk_pileup_stream_enable_evs_words: a sample of a somatic stream (it returns read positions instead)");
    s->want_evs = (enable != 0);
    return 0;
}

void sk_pileup_stream_destroy(sk_pileup_stream* s)
{
    if (!s) return;
    if (sk_ctx().ready) {
        (void)skrt::setDevice(sk_ctx().device);
        (void)skrt::streamSynchronize(sk_ctx().stream);
    }
    stream_drop(s);
    delete s;
}

int sk_pileup_stream_begin_region(sk_pileup_stream* s, const char* ref_seq, const int32_t ref_offset, const int32_t ref_len,
                                  const int32_t report_begin, const int32_t report_end,
                                  const int32_t largest_total_indel_ref_span_per_read)
{
    SK_REQUIRE_INIT();
    if (!s || (!ref_seq && ref_len > 0)) return sk_fail("sk_pileup_stream_begin_region: null argument");
    if (ref_len < 0 || report_end < report_begin) return sk_fail("sk_pileup_stream_begin_region: bad range");
    if (s->in_flight) return sk_fail("sk_pileup_stream_begin_region: the stream's last push has not been finished");
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    hipStream_t st = ctx.stream;
    const size_t n_region = size_t(report_end - report_begin);
    if (s->d_ref.need(size_t(ref_len) + 1) || s->d_mask.need(size_t(ref_len) + 1) || s->d_spandel.need(4 * n_region + 4) ||
        s->d_submapped.need(4 * n_region + 4))
        return sk_fail("sk_pileup_stream_begin_region: out of device memory");
    if (ref_len > 0) SK_HIP(skrt::memcpyAsync(s->d_ref.p, ref_seq, size_t(ref_len), hipMemcpyHostToDevice, st));
    SK_HIP(skrt::memsetAsync(s->d_mask.p, 0, size_t(ref_len) + 1, st));
    SK_HIP(skrt::memsetAsync(s->d_spandel.p, 0, 4 * n_region + 4, st));
    SK_HIP(skrt::memsetAsync(s->d_submapped.p, 0, 4 * n_region + 4, st));
    SK_HIP(skrt::streamSynchronize(st)); // ref_seq is the caller's
    s->poisoned = false;
    s->ref_offset = ref_offset;
    s->ref_len = ref_len;
    s->region_begin = report_begin;
    s->region_end = report_end;
    s->opt.report_begin = report_begin;
    s->opt.report_end = report_end;
    s->opt.largest_total_indel_ref_span_per_read = largest_total_indel_ref_span_per_read;
    s->c_len.clear(); s->c_nseg.clear(); s->c_pos.clear(); s->c_mapq.clear(); s->c_level.clear(); s->c_path.clear();
    s->c_bases = 0;
    s->c_tail_base = 0;
    s->c_tail_read = 0;
    s->has_prev = false;
    s->next_begin = report_begin;
    s->has_region = true;
    return 0;
}


// A push in two halves: _begin checks, packs and enqueues everything up to the copy of the window's output block and returns while the
// device works; _finish waits and hands out the window.  A caller with host work between the two (the adapter: the positions POST_ALIGN
// still has to go through before it needs this window) spends the device's ~0.4 ms there instead of asleep.  Between the two the stream
// object takes no other call.
int sk_pileup_stream_push_begin(sk_pileup_stream* s, const sk_read_batch* reads, const int32_t largest_total_indel_ref_span_per_read,
                                const int32_t mask_begin, const int32_t mask_len, const uint8_t* cand_snv_mask, const int32_t final_to,
                                const int32_t ploidy_begin, const int32_t ploidy_len, const uint8_t* ploidy)
{
    SK_REQUIRE_INIT();
    skrt::wakeHint();
    const bool tm = g_push_seconds.on;
    const double t0 = tm ? PushSeconds::now() : 0.0;
    if (!s || !reads) return sk_fail("sk_pileup_stream_push: null argument");
    if (s->poisoned) return sk_fail("sk_pileup_stream_push: an earlier push of this stream failed; begin the region again");
    if (s->in_flight) return sk_fail("sk_pileup_stream_push_begin: the stream's last push has not been finished");
    if (stream_check_reads(s, reads, mask_begin, mask_len, cand_snv_mask)) return 1;
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    const int32_t F = std::min(final_to, s->region_end);
    int32_t lowest, highest, begin, end;
    if (stream_extent(s, reads, &lowest, &highest)) return 1;
    stream_range(s, lowest, highest, F, &begin, &end);
    const double t1 = tm ? PushSeconds::now() : 0.0;
    // (stream_enqueue flips the record buffers and moves the carried tail as it goes: a failure from here on leaves work in flight and
    // the stream's bookkeeping half done -- the stream is drained and refuses further pushes until begin_region resets it)
    if (stream_enqueue(s, reads, largest_total_indel_ref_span_per_read, mask_begin, mask_len, cand_snv_mask, F, begin, end, ploidy_begin,
                       ploidy_len, ploidy)) {
        (void)skrt::streamSynchronize(ctx.stream);
        s->poisoned = true;
        return 1;
    }
    skrt::kick(); // (a broker client's records start running now, not at the wait)
    s->in_flight = true;
    if (tm) {
        const double t2 = PushSeconds::now();
        g_push_seconds.check += t1 - t0, g_push_seconds.enqueue += t2 - t1;
    }
    return 0;
}

int sk_pileup_stream_push_finish(sk_pileup_stream* s, sk_pileup_window* out)
{
    SK_REQUIRE_INIT();
    if (!s || !out) return sk_fail("sk_pileup_stream_push_finish: null argument");
    if (!s->in_flight) return sk_fail("sk_pileup_stream_push_finish: no push of this stream has been begun");
    const bool tm = g_push_seconds.on;
    const double t2 = tm ? PushSeconds::now() : 0.0;
    SkContext& ctx = sk_ctx();
    s->in_flight = false;
    if (skrt::streamSynchronize(ctx.stream) != hipSuccess) {
        s->poisoned = true;
        return sk_fail("sk_pileup_stream_push: the device reported an error");
    }
    const double t3 = tm ? PushSeconds::now() : 0.0;
    stream_finish(s, out);
    if (tm) {
        const double t4 = PushSeconds::now();
        g_push_seconds.wait += t3 - t2, g_push_seconds.finish += t4 - t3;
        ++g_push_seconds.pushes;
    }
    return 0;
}

int sk_pileup_stream_push(sk_pileup_stream* s, const sk_read_batch* reads, const int32_t largest_total_indel_ref_span_per_read,
                          const int32_t mask_begin, const int32_t mask_len, const uint8_t* cand_snv_mask, const int32_t final_to,
                          const int32_t ploidy_begin, const int32_t ploidy_len, const uint8_t* ploidy, sk_pileup_window* out)
{
    if (!out) return sk_fail("sk_pileup_stream_push: null argument");
    if (sk_pileup_stream_push_begin(s, reads, largest_total_indel_ref_span_per_read, mask_begin, mask_len, cand_snv_mask, final_to, ploidy_begin,
                                    ploidy_len, ploidy))
        return 1;
    return sk_pileup_stream_push_finish(s, out);
}

// ---- the two samples of a somatic run, pushed together and chained into a12+a13 ----------------------------------------------

sk_somatic_pileup_stream* sk_somatic_pileup_stream_create(const sk_pileup_options* opt, const sk_somatic_snv_options* genotype_opt,
                                                          const int with_read_pos)
{
    if (!sk_ctx().ready) {
        sk_fail("strelka_amd: sk_init() has not succeeded");
        return nullptr;
    }
    if (!opt) {
        sk_fail("sk_somatic_pileup_stream_create: null options");
        return nullptr;
    }
    sk_somatic_pileup_stream* p = new sk_somatic_pileup_stream();
    for (int i = 0; i < 2; ++i) {
        p->sample[i] = new sk_pileup_stream();
        p->sample[i]->opt = *opt;
        p->sample[i]->somatic = true;
    }
    p->sample[1]->want_read_pos = (with_read_pos != 0);
    p->tier2 = (opt->use_tier2_evidence != 0);
    if (genotype_opt) {
        p->sopt = *genotype_opt;
        p->genotype = true;
    }
    return p;
}

void sk_somatic_pileup_stream_destroy(sk_somatic_pileup_stream* p)
{
    if (!p) return;
    if (sk_ctx().ready) {
        (void)skrt::setDevice(sk_ctx().device);
        (void)skrt::streamSynchronize(sk_ctx().stream);
    }
    for (int i = 0; i < 2; ++i) {
        stream_drop(p->sample[i]);
        delete p->sample[i];
    }
    p->d_call.drop();
    p->h_forced.drop();
    p->h_geno.drop();
    delete p;
}

int sk_somatic_pileup_stream_begin_region(sk_somatic_pileup_stream* p, const char* ref_seq, const int32_t ref_offset, const int32_t ref_len,
                                          const int32_t report_begin, const int32_t report_end,
                                          const int32_t largest_total_indel_ref_span_per_read)
{
    if (!p) return sk_fail("sk_somatic_pileup_stream_begin_region: null argument");
    for (int i = 0; i < 2; ++i) {
        if (sk_pileup_stream_begin_region(p->sample[i], ref_seq, ref_offset, ref_len, report_begin, report_end, largest_total_indel_ref_span_per_read))
            return 1;
    }
    return 0;
}

int sk_somatic_pileup_stream_push_begin(sk_somatic_pileup_stream* p, const sk_read_batch* normal_reads, const sk_read_batch* tumor_reads,
                                        const int32_t largest_total_indel_ref_span_per_read, const int32_t mask_begin, const int32_t mask_len,
                                        const uint8_t* cand_snv_mask, const int32_t final_to, const int32_t forced_begin, const int32_t forced_len,
                                        const uint8_t* is_forced_output, const int is_compute_nonsomatic)
{
    SK_REQUIRE_INIT();
    skrt::wakeHint();
    if (!p || !normal_reads || !tumor_reads) return sk_fail("sk_somatic_pileup_stream_push: null argument");
    if (p->in_flight) return sk_fail("sk_somatic_pileup_stream_push_begin: the stream's last push has not been finished");
    const sk_read_batch* reads[2] = { normal_reads, tumor_reads };
    for (int i = 0; i < 2; ++i) {
        if (p->sample[i]->poisoned) return sk_fail("sk_somatic_pileup_stream_push: an earlier push of this stream failed; begin the region again");
        if (stream_check_reads(p->sample[i], reads[i], mask_begin, mask_len, cand_snv_mask)) return 1;
    }
    if (forced_len < 0 || (forced_len > 0 && !is_forced_output)) return sk_fail("sk_somatic_pileup_stream_push: bad forced-output window");
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    hipStream_t st = ctx.stream;
    sk_pileup_stream* sn = p->sample[0];
    sk_pileup_stream* stu = p->sample[1];
    const int32_t F = std::min(final_to, sn->region_end);
    // one range for both samples: the positions either sample's reads cover
    int32_t lowest = INT_MAX, highest = INT_MIN, begin, end;
    for (int i = 0; i < 2; ++i) {
        int32_t lo, hi;
        if (stream_extent(p->sample[i], reads[i], &lo, &hi)) return 1;
        lowest = std::min(lowest, lo);
        highest = std::max(highest, hi);
    }
    stream_range(sn, lowest, highest, F, &begin, &end);
    for (int i = 0; i < 2; ++i) {
        if (stream_enqueue(p->sample[i], reads[i], largest_total_indel_ref_span_per_read, mask_begin, mask_len, cand_snv_mask, F, begin, end, 0,
                           0, nullptr)) {
            // (the other sample may be enqueued already: drain, and both streams refuse pushes until begin_region)
            (void)skrt::streamSynchronize(st);
            p->sample[0]->poisoned = p->sample[1]->poisoned = true;
            return 1;
        }
    }
    const int n_loci = end - begin;
    // a12 + a13 on the four cleaned columns, where they are
    const size_t geno_bytes = sizeof(sk_somatic_snv_genotype) * size_t(std::max(n_loci, 1));
    if (p->genotype && n_loci > 0) {
        const size_t forced_bytes = size_t(align256(n_loci));
        const size_t scratch_bytes = sk_somatic_snv_tiers_scratch_bytes(n_loci);
        if (p->d_call.need(forced_bytes + size_t(align256(int64_t(geno_bytes))) + scratch_bytes) || p->h_forced.need(forced_bytes) ||
            p->h_geno.need(geno_bytes))
            return sk_fail("sk_somatic_pileup_stream_push: out of memory (somatic records)");
        uint8_t* hf = static_cast<uint8_t*>(p->h_forced.p);
        for (int i = 0; i < n_loci; ++i) {
            const int64_t k = int64_t(begin) + i - forced_begin;
            hf[i] = (is_forced_output && k >= 0 && k < forced_len) ? is_forced_output[k] : uint8_t(0);
        }
        char* dc = static_cast<char*>(p->d_call.p);
        uint8_t* d_forced = reinterpret_cast<uint8_t*>(dc);
        sk_somatic_snv_genotype* d_geno = reinterpret_cast<sk_somatic_snv_genotype*>(dc + forced_bytes);
        void* d_scratch = dc + forced_bytes + size_t(align256(int64_t(geno_bytes)));
        SK_HIP(skrt::memcpyAsync(d_forced, hf, size_t(n_loci), hipMemcpyHostToDevice, st));
        sk_pileup_batch b[4]; // normal t1, tumor t1, normal t2, tumor t2
        for (int i = 0; i < 2; ++i) {
            sk_pileup_stream* s = p->sample[i];
            char* dw = static_cast<char*>(s->d_work.p);
            std::memset(&b[i], 0, sizeof(sk_pileup_batch));
            b[i].n_loci = n_loci;
            b[i].call_off = reinterpret_cast<const int64_t*>(dw + s->wl.off2);
            b[i].calls = reinterpret_cast<const uint16_t*>(dw + s->wl.calls2);
            b[i].ref_base = reinterpret_cast<const uint8_t*>(static_cast<char*>(sn->d_work.p) + sn->wl.refbase);
            b[2 + i] = b[i];
            b[2 + i].call_off = reinterpret_cast<const int64_t*>(dw + s->wl.off4);
            b[2 + i].calls = reinterpret_cast<const uint16_t*>(dw + s->wl.calls4);
        }
        if (sk_somatic_snv_call_tiers_dev(&b[0], &b[1], p->tier2 ? &b[2] : nullptr, p->tier2 ? &b[3] : nullptr, &p->sopt, d_forced,
                                          is_compute_nonsomatic, d_geno, d_scratch, st))
            return 1;
        SK_HIP(skrt::memcpyAsync(p->h_geno.p, d_geno, sizeof(sk_somatic_snv_genotype) * size_t(n_loci), hipMemcpyDeviceToHost, st));
    }
    skrt::kick(); // (a broker client's records start running now, not at the wait)
    p->in_flight = true;
    p->p_loci = n_loci;
    return 0;
}

int sk_somatic_pileup_stream_push_finish(sk_somatic_pileup_stream* p, sk_somatic_pileup_window* out)
{
    SK_REQUIRE_INIT();
    if (!p || !out) return sk_fail("sk_somatic_pileup_stream_push_finish: null argument");
    if (!p->in_flight) return sk_fail("sk_somatic_pileup_stream_push_finish: no push of this stream has been begun");
    p->in_flight = false;
    sk_pileup_stream* sn = p->sample[0];
    sk_pileup_stream* stu = p->sample[1];
    const int n_loci = p->p_loci;
    if (skrt::streamSynchronize(sk_ctx().stream) != hipSuccess) {
        sn->poisoned = stu->poisoned = true;
        return sk_fail("sk_somatic_pileup_stream_push: the device reported an error");
    }
    stream_finish(sn, &out->normal);
    stream_finish(stu, &out->tumor);
    out->tumor_tier1_read_pos =
        stu->want_read_pos ? reinterpret_cast<const uint32_t*>(static_cast<const char*>(stu->h_out_cur().p) + stu->ol.read_pos) : nullptr;
    out->normal_clean_tier2_count = reinterpret_cast<const uint32_t*>(static_cast<const char*>(sn->h_out_cur().p) + sn->ol.clean4_n);
    out->tumor_clean_tier2_count = reinterpret_cast<const uint32_t*>(static_cast<const char*>(stu->h_out_cur().p) + stu->ol.clean4_n);
    out->genotype = (p->genotype && n_loci > 0) ? static_cast<const sk_somatic_snv_genotype*>(p->h_geno.p) : nullptr;
    return 0;
}


int sk_somatic_pileup_stream_push(sk_somatic_pileup_stream* p, const sk_read_batch* normal_reads, const sk_read_batch* tumor_reads,
                                  const int32_t largest_total_indel_ref_span_per_read, const int32_t mask_begin, const int32_t mask_len,
                                  const uint8_t* cand_snv_mask, const int32_t final_to, const int32_t forced_begin, const int32_t forced_len,
                                  const uint8_t* is_forced_output, const int is_compute_nonsomatic, sk_somatic_pileup_window* out)
{
    if (!out) return sk_fail("sk_somatic_pileup_stream_push: null argument");
    if (sk_somatic_pileup_stream_push_begin(p, normal_reads, tumor_reads, largest_total_indel_ref_span_per_read, mask_begin, mask_len, cand_snv_mask,
                                            final_to, forced_begin, forced_len, is_forced_output, is_compute_nonsomatic))
        return 1;
    return sk_somatic_pileup_stream_push_finish(p, out);
}

} // extern "C"
