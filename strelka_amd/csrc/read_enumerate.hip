// read_enumerate.hip -- candidate-alignment enumeration, flattening and scoring of a realignment job on the device
// (sk_realign_options.enumeration == 2).
//
//   E1  root_kernel, level_kernel x depth   candidate_alignment_search (csrc/realign_core.h, the statement of
//                          L/starling_common/starling_read_align.cpp:857-1277 shared with the host), level by level: one thread
//                          per call of the reference's recursion, all reads of the job at once
//   E2  group/dedupe/rank  the leaves of each read -> its std::set<CandidateAlignment>: duplicates dropped, set order
//   L1  pool_bounds/layout  reference window and insert sequences of each read's haplotype pool
//   L2  op_count_kernel    one thread per candidate alignment: ops it flattens to
//   F1  pool_fill_kernel   one wave per read: the pool's bytes
//   F2  flatten_kernel     one thread per candidate alignment: the walk of scoreCandidateAlignment
//                          (starling_read_align_score.cpp:286-493) emitting scoring ops (host/align_flatten.cpp holds the host form)
//   F3  entries_kernel     one thread per candidate alignment: the transition entries and event masks kernel A1 reads
//                          (csrc/align_entry.h) -- so neither sk_align_builder nor sk_align_prepare runs on the host for these reads
//   A1  score_wave_per_read (score_alignments.hip) on the batch E3 left in HBM
//
// All of this is integer/byte work; it has to reproduce the host stages exactly (tests/test_device_enumeration.py).  A read that
// exceeds a capacity of the core, or that the host code would have thrown on, is reported (status != ST_OK) and redone by the
// container-based host code, which then produces either the result or the reference's error text -- nothing is truncated.

#include "sk_common.h"

#include "align_entry.h"
#include "read_enumerate.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace skcore;

int sk_score_alignments_launch_hostleg(const sk_align_batch* b, double* dev_out_lnp, void* hip_stream); // score_alignments.hip

namespace
{

static_assert(sizeof(PFrame) % 4 == 0 && sizeof(PCal) % 4 == 0, "frames move through LDS as 32-bit words");

// ---------------------------------------------------------------------------------------------------------------------
// E1: the search, level by level.  The reference's recursion passes its whole state by value, so the calls at one depth are
// independent of each other (they share the set the leaves go into and two warning flags): level d of ALL reads of the job is
// one launch, one thread per call, the frames of the level in one array and the children appended to the next level's array.
// A thread's frame is staged into LDS first (the 64 frames of a workgroup are contiguous in HBM): expanding a call is a long
// chain of dependent accesses to its frame, and LDS is an order of magnitude closer than HBM.

struct EnumArgs
{
    PJob job; // device pointers
    const PRead* reads;
    int32_t n_reads;
    PFrame* level_in;
    PFrame* level_out;
    int32_t frame_cap;
    int32_t* level_count; // [Caps::K + 3]
    int32_t depth;
    PCal* pool; // leaves, as found (duplicates included)
    int32_t* leaf_read;
    uint32_t* leaf_hash;
    int32_t pool_cap;
    int32_t* n_leaves;
    int32_t* n_raw;  // [n_reads] leaves found
    int32_t* status; // [n_reads] max over the read's calls
    int32_t* warn;   // [n_reads] bit 0 origin, bit 1 toggle depth
    unsigned long long* n_nodes;
    // null, or per read cells that start at INT_MAX (win_begin, ins_lo) / INT_MIN (win_end, ins_hi): the root launch sets them when the
    // job runs as one fixed sequence (no fills of their own)
    int32_t* min_cells_a;
    int32_t* min_cells_b;
    int32_t* max_cells_a;
    int32_t* max_cells_b;
};

__device__ inline uint32_t cal_hash(const PCal& c)
{
    uint32_t h = 2166136261u;
    auto mix = [&](const uint32_t v) { h = (h ^ v) * 16777619u; };
    mix(uint32_t(c.pos));
    mix(uint32_t(c.fwd) | (uint32_t(c.n_seg) << 8) | (uint32_t(c.n_indels) << 16));
    mix(uint32_t(uint16_t(c.lead)) | (uint32_t(uint16_t(c.trail)) << 16));
    for (int i = 0; i < c.n_seg; ++i) mix(uint32_t(c.path[i].type) | (uint32_t(c.path[i].length) << 16));
    for (int i = 0; i < c.n_indels; ++i) mix(uint32_t(uint16_t(c.indels[i])));
    return h;
}

struct LeafSink // a leaf: the read's clips back on (clip_adder :508-542), dropped when outside the realign range (:1981-1993)
{
    const EnumArgs* a;
    const PRead* r;
    int32_t read_id;
    __device__ bool operator()(PCal& leaf)
    {
        if (r->clipped && !clip_adder(leaf, r->hc_lead, r->hc_trail, r->sc_lead, r->sc_trail)) return false;
        if (!superset_of(mk_range(r->realign_b, r->realign_e), strict_range(leaf))) return true;
        const int slot = atomicAdd(a->n_leaves, 1);
        if (slot >= a->pool_cap) return false;
        copy_cal(a->pool[slot], leaf);
        a->leaf_read[slot] = read_id;
        a->leaf_hash[slot] = cal_hash(leaf);
        atomicAdd(&a->n_raw[read_id], 1);
        return true;
    }
};

struct LevelAlloc
{
    const EnumArgs* a;
    __device__ PFrame* operator()()
    {
        const int slot = atomicAdd(&a->level_count[a->depth + 1], 1);
        if (slot >= a->frame_cap) return nullptr;
        return &a->level_out[slot];
    }
};

__global__ __launch_bounds__(64) void root_kernel(const EnumArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_reads) return;
    if (r == 0) a.level_count[0] = a.n_reads;
    if (a.min_cells_a) {
        a.min_cells_a[r] = a.min_cells_b[r] = INT_MAX;
        a.max_cells_a[r] = a.max_cells_b[r] = INT_MIN;
    }
    PFrame& f = a.level_out[r];
    root_frame(a.job, a.reads[r], f);
    f.read_id = r;
}

enum { FRAME_WORDS = sizeof(PFrame) / 4 };

__global__ __launch_bounds__(64) void level_kernel(const EnumArgs a)
{
    extern __shared__ __align__(16) uint32_t lds_words[];
    PFrame* frames = reinterpret_cast<PFrame*>(lds_words);
    const int n_in = min(a.level_count[a.depth], a.frame_cap);
    for (int base = blockIdx.x * 64; base < n_in; base += gridDim.x * 64) {
        const int n_here = min(64, n_in - base);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.level_in + base);
        for (int w = threadIdx.x; w < n_here * FRAME_WORDS; w += 64) lds_words[w] = src[w];
        __syncthreads();
        if (int(threadIdx.x) < n_here) {
            PFrame& f = frames[threadIdx.x];
            const int r = f.read_id;
            if (f.stage != 0xffff && a.status[r] == ST_OK) {
                SearchOut so;
                so.status = ST_OK;
                so.warn_origin = so.warn_toggle = 0;
                so.nodes = 0;
                LeafSink sink{ &a, &a.reads[r], r };
                LevelAlloc alloc{ &a };
                expand_node(a.job, a.reads[r], f, alloc, &f.cal, sink, so); // (a leaf is completed in place: the frame is done with)
                if (so.status != ST_OK) atomicMax(&a.status[r], so.status);
                if (so.warn_origin | so.warn_toggle) atomicOr(&a.warn[r], (so.warn_origin ? 1 : 0) | (so.warn_toggle ? 2 : 0));
            }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.n_nodes) atomicAdd(a.n_nodes, (unsigned long long)n_in);
}

// ---------------------------------------------------------------------------------------------------------------------
// E2: the leaves of each read -> a std::set<CandidateAlignment>: grouped by read, duplicates dropped, ordered

struct SetArgs
{
    const PCal* pool;
    const int32_t* leaf_read;
    const uint32_t* leaf_hash;
    const int32_t* status;
    int32_t n_reads, n_leaves;
    const int32_t* raw_off; // [n_reads+1] (reads that failed: empty)
    int32_t* fill;          // [n_reads] zeroed
    int32_t* grouped;       // [raw_off[n_reads]] leaf slots, read by read
    uint32_t* ghash;        // [raw_off[n_reads]] their hashes
    ulonglong2* gkey;       // [raw_off[n_reads]] their keys (leaf_key)
    uint8_t* dup;           // [raw_off[n_reads]]
    int32_t* n_uniq;        // [n_reads] zeroed
    const int32_t* cal_off; // [n_reads+1]
    int32_t* sorted;        // [cal_off[n_reads]] leaf slots in set order
    const int32_t* n_leaves_dev; // null, or where the search counted its leaves: the job runs as one fixed sequence, the host has not
                                 // seen the count (n_leaves is then the pool's capacity)
};

__device__ inline int group_of(const int32_t* off, const int n, const int g) // last r with off[r] <= g
{
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// The first fields cal_compare looks at (position, strand, number of segments, the first four segments), packed so that comparing
// the two words as numbers is comparing those fields in cal_compare's order: most pairs of leaves are told apart by the key alone,
// read from one contiguous array instead of from two 300-byte records
__device__ inline ulonglong2 leaf_key(const PCal& c)
{
    auto seg = [&](const int i) -> unsigned long long { // 20 bits: type, length
        return i < c.n_seg ? ((unsigned long long)(c.path[i].type & 0xfu) << 16) | c.path[i].length : 0ull;
    };
    ulonglong2 k;
    k.x = ((unsigned long long)(uint32_t(c.pos) ^ 0x80000000u) << 32) | ((unsigned long long)(c.fwd ? 1 : 0) << 31) |
          ((unsigned long long)(c.n_seg & 0x7fu) << 24) | (seg(0) << 4) | (seg(1) >> 16);
    k.y = ((seg(1) & 0xffffull) << 48) | (seg(2) << 28) | (seg(3) << 8);
    return k;
}
__global__ __launch_bounds__(256) void group_kernel(const SetArgs a)
{
    const int n_leaves = a.n_leaves_dev ? min(*a.n_leaves_dev, a.n_leaves) : a.n_leaves;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n_leaves; s += gridDim.x * blockDim.x) {
        const int r = a.leaf_read[s];
        if (a.status[r] != ST_OK) continue;
        const int g = a.raw_off[r] + atomicAdd(&a.fill[r], 1);
        a.grouped[g] = s;
        a.ghash[g] = a.leaf_hash[s];
        a.gkey[g] = leaf_key(a.pool[s]);
    }
}

// a leaf is a duplicate when an equal leaf stands before it in its read's group (which of the equal ones stands first differs
// from run to run; they are equal).  The scan over the group is a plain comparison of hashes with nothing in the loop that waits
// for a load (reads with thousands of leaves make it long); the records are compared only where a hash is met again.
__global__ __launch_bounds__(256) void dedupe_kernel(const SetArgs a)
{
    const int n_grouped = a.raw_off[a.n_reads];
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n_grouped; g += gridDim.x * blockDim.x) {
    const int r = group_of(a.raw_off, a.n_reads, g);
    const uint32_t h = a.ghash[g];
    const int q0 = a.raw_off[r];
    bool dup = false;
    for (int from = q0; from < g && !dup;) {
        int first = g; // the first leaf at or after `from` with this hash
        int q = from;
        for (; q + 8 <= g; q += 8) {
            uint32_t v[8];
            for (int z = 0; z < 8; ++z) v[z] = a.ghash[q + z];
            for (int z = 7; z >= 0; --z) first = (v[z] == h && q + z < first) ? q + z : first;
            if (first < g) break;
        }
        if (first == g)
            for (; q < g; ++q)
                if (a.ghash[q] == h) {
                    first = q;
                    break;
                }
        if (first >= g) break;
        if (cal_compare(a.pool[a.grouped[first]], a.pool[a.grouped[g]]) == 0) dup = true;
        from = first + 1;
    }
    a.dup[g] = dup ? 1 : 0;
    if (!dup) atomicAdd(&a.n_uniq[r], 1);
    }
}

// the rank of a kept leaf in the set order of its read (CandidateAlignment.hh:37-48, alignment.hh:72-90: cal_compare): the keys
// decide for almost every pair, without a branch; pairs with equal keys (their number is counted on the way) get the full comparison
__global__ __launch_bounds__(256) void rank_kernel(const SetArgs a)
{
    const int n_grouped = a.raw_off[a.n_reads];
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n_grouped; g += gridDim.x * blockDim.x) {
    if (a.dup[g]) continue;
    const int r = group_of(a.raw_off, a.n_reads, g);
    const ulonglong2 kmine = a.gkey[g];
    const int q0 = a.raw_off[r], q1 = a.raw_off[r + 1];
    int rank = 0, n_equal = 0;
    auto count = [&](const ulonglong2 k, const uint8_t dup) {
        const bool kept = dup == 0;
        const bool lower = (k.x < kmine.x) || (k.x == kmine.x && k.y < kmine.y);
        const bool equal = (k.x == kmine.x) && (k.y == kmine.y);
        rank += (kept && lower) ? 1 : 0;
        n_equal += (kept && equal) ? 1 : 0;
    };
    int q = q0;
    for (; q + 8 <= q1; q += 8) { // (eight keys in flight: one at a time the loop is a chain of load latencies, 120 us per 4 096 reads x 88)
        ulonglong2 k[8];
        uint8_t d[8];
#pragma unroll
        for (int z = 0; z < 8; ++z) {
            k[z] = a.gkey[q + z];
            d[z] = a.dup[q + z];
        }
#pragma unroll
        for (int z = 0; z < 8; ++z) count(k[z], d[z]);
    }
    for (; q < q1; ++q) count(a.gkey[q], a.dup[q]);
    if (n_equal > 1) { // (itself and others)
        const PCal& mine = a.pool[a.grouped[g]];
        for (int q = q0; q < q1; ++q) {
            if (q == g || a.dup[q]) continue;
            const ulonglong2 k = a.gkey[q];
            if (k.x == kmine.x && k.y == kmine.y && cal_compare(a.pool[a.grouped[q]], mine) < 0) ++rank;
        }
    }
    a.sorted[a.cal_off[r] + rank] = a.grouped[g];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The job as ONE fixed sequence of launches (no host decision between the upload and the results): the two prefix sums the host used to
// make between waits -- leaves per read -> raw_off, unique alignments per read -> cal_off -- are made here, by one workgroup (a job has at
// most a few ten thousand reads: a thread a run of consecutive reads, one block-wide scan of the runs' sums), together with what else
// the host derived from them: the totals, stage 3's two read lists, and the conditions under which the sequence's assumptions do not
// hold (then the host runs the job again the staged way: sk_enum_device_run).

enum { DYN_N_GROUPED = 0, DYN_N_CALS, DYN_N_LIGHT, DYN_N_HEAVY, DYN_FLAGS, DYN_MAX_CALS, DYN_REF_OUTSIDE, DYN_COUNT = 8 };
enum { DYNF_DEEPER = 1,    // calls were left at the first level the sequence did not launch
       DYNF_POOL_FULL = 2 }; // more leaves than the sequence's pool holds

__device__ inline int block_exclusive_scan(const int v, int* wave_sums /* LDS [blockDim.x / 64 + 1] */, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    int x = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads(); // (wave_sums may still be read from a previous scan)
    if (lane == 63) wave_sums[wave] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < n_waves; ++w) {
            const int t = wave_sums[w];
            wave_sums[w] = run;
            run += t;
        }
        wave_sums[n_waves] = run;
    }
    __syncthreads();
    *total = wave_sums[n_waves];
    return wave_sums[wave] + x - v;
}

struct ScanArgs
{
    int32_t n_reads;
    const int32_t* status;
    const int32_t* count;   // [n_reads] n_raw (scan 1) / n_uniq (scan 2)
    int32_t* off;           // [n_reads + 1] raw_off / cal_off
    int32_t* dyn;           // [DYN_COUNT]
    // scan 1
    const int32_t* level_count;
    int32_t first_level_not_launched; // (or -1: every level went out)
    const int32_t* n_leaves;
    int32_t pool_cap;
    // scan 2
    int32_t* list;          // [n_reads] stage 3: reads with at most light_cals alignments from the front, the others from the back
    int32_t light_cals;
};

// The job's way in and out as kernels: the packed input arena is read from the caller's page-locked mirror and the zero arena filled by
// one launch, the results (the zero arena, stage 3's records) are written to their page-locked mirrors by another.  The copy and fill
// calls they replace go through the runtime's copy engines, which eight caller processes sharing a GPU queue up for (a job's two copies
// in cost 0.16 ms of host time there against 0.015 ms alone: profiles/r05_enum_job_history.txt); a launch is a packet in the process's
// own queue.
__global__ __launch_bounds__(256) void job_stage_in_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const uint32_t n_in, uint4* __restrict__ zero,
                                                          const uint32_t n_zero)
{
    const uint32_t i0 = blockIdx.x * 256 + threadIdx.x, step = gridDim.x * 256;
    for (uint32_t i = i0; i < n_in; i += step) dst[i] = src[i];
    const uint4 z = { 0, 0, 0, 0 };
    for (uint32_t i = i0; i < n_zero; i += step) zero[i] = z;
}
__global__ __launch_bounds__(256) void job_stage_out_kernel(const uint4* __restrict__ a, uint4* __restrict__ host_a, const uint32_t n_a, const uint4* __restrict__ b,
                                                           uint4* __restrict__ host_b, const uint32_t n_b)
{
    const uint32_t i0 = blockIdx.x * 256 + threadIdx.x, step = gridDim.x * 256;
    for (uint32_t i = i0; i < n_a; i += step) host_a[i] = a[i];
    for (uint32_t i = i0; i < n_b; i += step) host_b[i] = b[i];
}

template <bool SECOND>
__global__ __launch_bounds__(1024) void job_scan_kernel(const ScanArgs a)
{
    __shared__ int wave_sums[17];
    const int n = a.n_reads, t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int r0 = min(n, t * per), r1 = min(n, r0 + per);
    int sum = 0, n_light = 0, n_heavy = 0, max_cals = 0;
    for (int r = r0; r < r1; ++r) {
        const int k = (a.status[r] == ST_OK) ? a.count[r] : 0;
        sum += k;
        if (SECOND) {
            if (k <= a.light_cals) ++n_light; else ++n_heavy;
            max_cals = max(max_cals, k);
        }
    }
    int total = 0;
    int at = block_exclusive_scan(sum, wave_sums, &total);
    for (int r = r0; r < r1; ++r) {
        a.off[r] = at;
        at += (a.status[r] == ST_OK) ? a.count[r] : 0;
    }
    if (t == 0) a.off[n] = total;
    if (!SECOND) {
        if (t == 0) {
            a.dyn[DYN_N_GROUPED] = total;
            int flags = 0;
            if (a.first_level_not_launched >= 0 && a.level_count[a.first_level_not_launched] != 0) flags |= DYNF_DEEPER;
            if (*a.n_leaves >= a.pool_cap) flags |= DYNF_POOL_FULL;
            a.dyn[DYN_FLAGS] = flags;
        }
    } else {
        int tot_light = 0, tot_heavy = 0;
        int lat = block_exclusive_scan(n_light, wave_sums, &tot_light);
        int hat = block_exclusive_scan(n_heavy, wave_sums, &tot_heavy);
        for (int r = r0; r < r1; ++r) {
            const int k = (a.status[r] == ST_OK) ? a.count[r] : 0;
            if (k <= a.light_cals) a.list[lat++] = r; else a.list[n - 1 - hat++] = r;
        }
        atomicMax(&a.dyn[DYN_MAX_CALS], max_cals);
        if (t == 0) {
            a.dyn[DYN_N_CALS] = total;
            a.dyn[DYN_N_LIGHT] = tot_light;
            a.dyn[DYN_N_HEAVY] = tot_heavy;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// flattening

__device__ inline uint8_t code_of(const char c) // get_bam_seq_code, L/htsapi/bam_seq.hh:73-92
{
    switch (c) {
    case '=': return SK_BAM_REF;
    case 'A': return SK_BAM_A;
    case 'C': return SK_BAM_C;
    case 'G': return SK_BAM_G;
    case 'T': return SK_BAM_T;
    default: return SK_BAM_ANY;
    }
}

struct FlatArgs
{
    PJob job;
    const char* ins_pool;
    const char* ref;
    int32_t ref_offset, ref_len;
    int32_t n_reads, n_cals;
    const PCal* pool;
    const int32_t* list; // [n_cals] leaf slots, each read's in set order
    int32_t* status;
    const int64_t* read_off;
    const uint8_t* read_code;
    const int32_t* cal_off;
    // per-read layout of the haplotype pool (L1 out)
    int32_t* win_begin; // (L1a: min over the read's candidate alignments, cells start at INT_MAX)
    int32_t* win_end;   // (cells start at INT_MIN)
    int32_t* ins_lo;    // lowest / highest table index of an indel with an insert sequence
    int32_t* ins_hi;
    uint8_t* n_seg8;    // [n_cals] path segments of each candidate alignment (capped at 255), set order: F5 orders a read's alignments by it
    int32_t* win_len;
    int32_t* hap_len;
    int32_t* n_ins;
    int16_t* ins_idx; // [n_reads][INS_CAP] table indices whose insert sequence is in the pool
    int32_t* ins_off; // [n_reads][INS_CAP] where
    // per candidate alignment
    int32_t* n_ops; // (L2 out)
    const int64_t* hap_off;
    PCal* cals;
    uint8_t* hap_code;
    const int64_t* op_off; // [n_cals + 1]
    sk_score_op* ops;
    uint32_t* entries;
    uint32_t* evmask;
    int32_t evmask_words, max_read_len;
    // the column form (strelka_amd.h, sk_align_batch::colmat)
    uint8_t* colmat; // (bytes of the words; pre-filled with the 0.0 column)
    const int64_t* colmat_off;
    uint32_t* addmask;
    int32_t* ref_outside;     // counts the window bytes that lie outside the job's reference segment (they read as N)
    int32_t n_cals_on_device; // the job runs as one fixed sequence: the number of candidate alignments is cal_off[n_reads], the host has not seen it
};
enum { INS_CAP = Caps::K + 2 };

__device__ inline int read_of_cal(const FlatArgs& a, const int c) // last r with cal_off[r] <= c
{
    int lo = 0, hi = a.n_reads;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.cal_off[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

// L1: the layout of each read's haplotype pool -- its reference window (the union of the reference spans of the match segments
// of all its candidate alignments), then the insert sequences of the table indels between the lowest and the highest one any of
// its candidate alignments holds, in table order.  (The host builder appends the insert sequences its ops use in order of first
// use and shares equal sequences; the layout is private to the read's batch entry and the scores do not depend on it.)
//   L1a  one thread per candidate alignment: window and indel-index bounds into the read's cells (atomicMin / atomicMax)
//   L1b  one thread per read: offsets
__global__ __launch_bounds__(256) void pool_bounds_kernel(const FlatArgs a)
{
    const int n_cals = a.n_cals_on_device ? a.cal_off[a.n_reads] : a.n_cals;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cals; c += gridDim.x * blockDim.x) {
    const int r = read_of_cal(a, c);
    const PCal& cal = a.pool[a.list[c]];
    a.n_seg8[c] = uint8_t(min(int(cal.n_seg), 255));
    int32_t wb = INT_MAX, we = INT_MIN, pos = cal.pos;
    for (int i = 0; i < cal.n_seg; ++i) {
        const PSeg s = cal.path[i];
        if (seg_align_match(s.type)) {
            wb = min(wb, pos);
            we = max(we, pos + int32_t(s.length));
            pos += int32_t(s.length);
        } else if (s.type == SK_SEG_DELETE || s.type == SK_SEG_SKIP) {
            pos += int32_t(s.length);
        }
    }
    if (wb <= we) {
        atomicMin(&a.win_begin[r], wb);
        atomicMax(&a.win_end[r], we);
    }
    int lo = INT_MAX, hi = INT_MIN;
    auto add = [&](const int t) {
        if (t < 0 || a.job.tab[t].ins_len == 0) return;
        lo = min(lo, t);
        hi = max(hi, t);
    };
    for (int i = 0; i < cal.n_indels; ++i) add(cal.indels[i]);
    add(cal.lead);
    add(cal.trail);
    if (lo <= hi) {
        atomicMin(&a.ins_lo[r], lo);
        atomicMax(&a.ins_hi[r], hi);
    }
    }
}

__global__ __launch_bounds__(64) void pool_layout_kernel(const FlatArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_reads) return;
    const int nc = a.cal_off[r + 1] - a.cal_off[r];
    a.n_ins[r] = 0;
    a.win_len[r] = 0;
    a.hap_len[r] = 0;
    if (nc == 0) return;
    int32_t wb = a.win_begin[r], we = a.win_end[r];
    if (wb > we) wb = we = 0;
    a.win_begin[r] = wb;
    int32_t len = we - wb;
    a.win_len[r] = len;
    int16_t* idx = a.ins_idx + size_t(r) * INS_CAP;
    int32_t* off = a.ins_off + size_t(r) * INS_CAP;
    int n = 0;
    for (int t = a.ins_lo[r]; t <= a.ins_hi[r]; ++t) {
        const uint32_t il = a.job.tab[t].ins_len;
        if (il == 0) continue;
        if (n >= INS_CAP) { // more insert sequences than this form holds: the host flattens this read
            a.status[r] = ST_OVERFLOW;
            n = 0;
            break;
        }
        idx[n] = int16_t(t);
        off[n] = len;
        len += int32_t(il);
        ++n;
    }
    a.n_ins[r] = n;
    a.hap_len[r] = max(len, 1);
}

// getMatchingIndelKey, starling_read_align_score.cpp:177-228: table index, -1 = no key, -2 = inconsistent
__device__ int matching_indel(const PJob& j, const PCal& c, const int32_t ref_head_pos, const unsigned del_len, const unsigned ins_len,
                              const int ends_first, const int ends_second, const int path_index)
{
    if (path_index < ends_first) return c.lead;
    if (path_index > ends_second) return c.trail;
    int found = -1;
    for (int k = 0; k < c.n_indels; ++k) {
        const PIndel& ci = j.tab[c.indels[k]];
        if (ci.pos == ref_head_pos && (ci.type == SK_INDEL_INDEL || ci.type == SK_INDEL_MISMATCH) && ci.del == del_len &&
            ci.ins_len == ins_len) {
            if (found >= 0) return -2;
            found = c.indels[k];
        } else if (ci.pos > ref_head_pos) {
            break;
        }
    }
    return found >= 0 ? found : -2;
}

// one candidate alignment -> ops (the walk of scoreCandidateAlignment :286-493 as host/align_flatten.cpp states it); returns the
// op count, -1 = leave this read to the host (it throws the reference's error, or handles what this form does not hold).
// WRITE = false only counts.
struct NoOpSink
{
    __device__ __forceinline__ void operator()(uint8_t, uint32_t, int32_t, bool) const {}
};

// SINK: called for every op in path order (kind, length, source offset, non-candidate penalty) -- the fused flatten + score kernel
// (F5) scores the op on the spot instead of storing it
template <bool WRITE, typename SINK = NoOpSink>
__device__ int flatten_cal(const FlatArgs& a, const int r, const PCal& c, const int32_t read_len, sk_score_op* ops, SINK&& sink = SINK())
{
    const int aps = c.n_seg;
    const int32_t win_begin = a.win_begin[r];
    const int n_ins = a.n_ins[r];
    const int16_t* ins_idx = a.ins_idx + size_t(r) * INS_CAP;
    const int32_t* ins_off = a.ins_off + size_t(r) * INS_CAP;
    unsigned read_offset = 0;
    int32_t ref_head_pos = c.pos;
    int ends_first = aps, ends_second = aps; // get_match_edge_segments, align_path.cpp:735-752
    {
        bool is_first_match = false;
        for (int i = 0; i < aps; ++i)
            if (seg_align_match(c.path[i].type)) {
                if (!is_first_match) ends_first = i;
                is_first_match = true;
                ends_second = i;
            }
    }
    int n = 0;
    auto emit = [&](const uint8_t kind, const uint32_t len, const int32_t src, const bool penalty) {
        if (kind == SK_OP_NOBASE && !penalty) return;
        sink(kind, len, src, penalty);
        if (WRITE) {
            sk_score_op op;
            op.length = uint16_t(len);
            op.kind = kind;
            op.flags = uint8_t(penalty ? SK_OPFLAG_NONCANDIDATE_PENALTY : 0);
            op.src = src;
            ops[n] = op;
        }
        ++n;
    };
    // offset of insert bases [head, head+len) of table indel `idx` in the pool, -1 = not representable here
    auto insert_src = [&](const int idx, const int32_t head, const uint32_t len) -> int32_t {
        const PIndel& k = a.job.tab[idx];
        if (head < 0 || uint32_t(head) + len > k.ins_len) return -1;
        for (int i = 0; i < n_ins; ++i)
            if (ins_idx[i] == idx) return ins_off[i] + head;
        return -1;
    };
    auto is_cand = [&](const int idx) -> bool { return job_cand(a.job, idx); };

    int path_index = 0;
    while (path_index < aps) {
        bool is_swap_start = false; // is_segment_swap_start, align_path.cpp:868-895
        {
            bool is_insert = false, is_delete = false;
            for (int i = path_index; i < aps; ++i) {
                if (c.path[i].type == SK_SEG_INSERT) is_insert = true;
                else if (c.path[i].type == SK_SEG_DELETE) is_delete = true;
                else break;
            }
            is_swap_start = is_insert && is_delete;
        }
        unsigned n_seg = 1;
        const PSeg ps = c.path[path_index];
        if (is_swap_start || ps.type == SK_SEG_SEQ_MISMATCH) {
            unsigned del_len, ins_len;
            if (ps.type == SK_SEG_SEQ_MISMATCH) {
                del_len = ins_len = ps.length;
            } else { // swap_info, align_path_util.hh:75-106
                int k = path_index;
                del_len = ins_len = 0;
                for (; k < aps && (c.path[k].type == SK_SEG_INSERT || c.path[k].type == SK_SEG_DELETE); ++k) {
                    if (c.path[k].type == SK_SEG_INSERT) ins_len += c.path[k].length;
                    else del_len += c.path[k].length;
                }
                n_seg = unsigned(k - path_index);
            }
            const int key = matching_indel(a.job, c, ref_head_pos, del_len, ins_len, ends_first, ends_second, path_index);
            if (key < 0) return -1;
            int32_t head = 0;
            if (path_index < ends_first) head = int32_t(a.job.tab[key].ins_len) - int32_t(ps.length);
            const bool pen = !is_cand(key);
            if (ins_len > 0) {
                if (ins_len > 0xffffu) return -1;
                const int32_t src = insert_src(key, head, ins_len);
                if (src < 0) return -1;
                emit(SK_OP_BASES, ins_len, src, pen);
            } else {
                emit(SK_OP_NOBASE, 0, 0, pen);
            }
        } else if (seg_align_match(ps.type)) {
            emit(SK_OP_BASES, ps.length, ref_head_pos - win_begin, false);
        } else if (ps.type == SK_SEG_INSERT) {
            const int key = matching_indel(a.job, c, ref_head_pos, 0, ps.length, ends_first, ends_second, path_index);
            if (key < 0) return -1;
            int32_t head = 0;
            if (path_index < ends_first) head = int32_t(a.job.tab[key].ins_len) - int32_t(ps.length);
            const int32_t src = insert_src(key, head, ps.length);
            if (src < 0) return -1;
            emit(SK_OP_BASES, ps.length, src, !is_cand(key));
        } else if (ps.type == SK_SEG_DELETE) {
            const int key = matching_indel(a.job, c, ref_head_pos, ps.length, 0, ends_first, ends_second, path_index);
            if (key < 0) return -1;
            emit(SK_OP_NOBASE, 0, 0, !is_cand(key));
        } else if (ps.type == SK_SEG_SKIP || ps.type == SK_SEG_HARD_CLIP) {
            // nothing
        } else if (ps.type == SK_SEG_SOFT_CLIP) {
            emit(SK_OP_SOFT_CLIP, ps.length, 0, false);
        } else {
            return -1;
        }
        for (unsigned i = 0; i < n_seg; ++i) { // increment_path, align_path_util.hh:38-68
            const PSeg s = c.path[path_index];
            if (seg_align_match(s.type)) {
                read_offset += s.length;
                ref_head_pos += int32_t(s.length);
            } else if (s.type == SK_SEG_DELETE || s.type == SK_SEG_SKIP) {
                ref_head_pos += int32_t(s.length);
            } else if (s.type == SK_SEG_INSERT || s.type == SK_SEG_SOFT_CLIP) {
                read_offset += s.length;
            }
            path_index++;
        }
    }
    if (int64_t(read_offset) != int64_t(read_len)) return -1;
    return n;
}

// L2, one thread per candidate alignment: how many ops it flattens to
__global__ __launch_bounds__(64) void op_count_kernel(const FlatArgs a)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.n_cals) return;
    const int r = read_of_cal(a, c);
    const PCal& cal = a.pool[a.list[c]];
    const int n = flatten_cal<false>(a, r, cal, int32_t(a.read_off[r + 1] - a.read_off[r]), nullptr);
    if (n < 0) a.status[r] = ST_FAIL;
    a.n_ops[c] = n < 0 ? 0 : n;
}

// F1, one wave per read: the bytes of its haplotype pool
__global__ __launch_bounds__(64) void pool_fill_kernel(const FlatArgs a)
{
    const int r = blockIdx.x;
    if (a.cal_off[r + 1] == a.cal_off[r]) return;
    uint8_t* hap = a.hap_code + a.hap_off[r];
    const int32_t P = int32_t(a.hap_off[r + 1] - a.hap_off[r]);
    const int32_t wb = a.win_begin[r];
    const int n_ins = a.n_ins[r];
    const int16_t* idx = a.ins_idx + size_t(r) * INS_CAP;
    const int32_t* off = a.ins_off + size_t(r) * INS_CAP;
    const int32_t win_len = a.win_len[r];
    for (int32_t i = threadIdx.x; i < P; i += 64) {
        uint8_t v = SK_BAM_ANY;
        if (i < win_len) {
            const int32_t p = wb + i; // reference_contig_segment::get_base :46-51
            const bool outside = (p < a.ref_offset || p >= a.ref_offset + a.ref_len);
            if (outside && a.ref_outside) atomicAdd(a.ref_outside, 1);
            v = outside ? uint8_t(SK_BAM_ANY) : code_of(a.ref[p - a.ref_offset]);
        }
        for (int k = 0; k < n_ins; ++k) {
            const PIndel& d = a.job.tab[idx[k]];
            if (i >= off[k] && i < off[k] + int32_t(d.ins_len)) v = code_of(a.ins_pool[d.ins_off + uint32_t(i - off[k])]);
        }
        hap[i] = v;
    }
}

// F2, one thread per candidate alignment: the alignment itself (for the host), its ops, and the candidate-status lookups the
// host form performs for every indel of the alignment (cal_to_c)
__global__ __launch_bounds__(64) void flatten_kernel(const FlatArgs a)
{
    // The wave's 64 records travel pool -> LDS -> cals as rows of consecutive dwords: a lane copying its own 288-byte record field
    // by field made every load and store a scatter over 64 records (1.4 GB of traffic for 360 000 candidate alignments,
    // profiles/r03_v20_pmc_traffic.json); the lanes then read their record from LDS (stride 73 dwords: odd, conflict-free).
    constexpr int REC_DW = int(sizeof(PCal) / 4), REC_STRIDE = REC_DW | 1;
    static_assert(sizeof(PCal) % 4 == 0 && REC_DW <= 128, "a record is at most two dwords per lane");
    __shared__ uint32_t s_cal[64 * REC_STRIDE];
    const int lane = threadIdx.x;
    const int c0 = blockIdx.x * blockDim.x;
    const int c = c0 + lane;
    const int nc = min(64, a.n_cals - c0);
    const int src_k = (lane < nc) ? a.list[c0 + lane] : 0;
    for (int k = 0; k < nc; ++k) {
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(a.pool + __shfl(src_k, k));
        uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(a.cals + (c0 + k));
        for (int j = lane; j < REC_DW; j += 64) {
            const uint32_t v = src[j];
            s_cal[k * REC_STRIDE + j] = v;
            dst[j] = v;
        }
    }
    __syncthreads();
    if (c >= a.n_cals) return;
    const int r = read_of_cal(a, c);
    const PCal& cal = *reinterpret_cast<const PCal*>(s_cal + lane * REC_STRIDE);
    if (a.status[r] != ST_OK) return; // (its op range is empty)
    (void)flatten_cal<true>(a, r, cal, int32_t(a.read_off[r + 1] - a.read_off[r]), a.ops + a.op_off[c]);
    for (int i = 0; i < cal.n_indels; ++i) (void)job_cand(a.job, cal.indels[i]);
    if (cal.lead >= 0) (void)job_cand(a.job, cal.lead);
    if (cal.trail >= 0) (void)job_cand(a.job, cal.trail);
}

// ---------------------------------------------------------------------------------------------------------------------
// F5, one wave per read: flattening AND scoring in one kernel -- scoreCandidateAlignment (starling_read_align_score.cpp:261-499) as the
// reference has it, one function from the alignment to its double.  F1-F3 + A1c stage the same work through HBM (the pool's bytes, 8-byte
// ops, transition entries, masks, column words: 6.8 x the algorithmic bytes, profiles/r03_v24_pmc_traffic.json) and wait on dependent
// global loads lane by lane (a wave of entries_wave_kernel took ~200 us per read); here the read (codes, the two terms of every position),
// its haplotype pool and the wave's 64 candidate-alignment records sit in LDS, every lane walks its alignment's path with flatten_cal and
// adds each op's terms as they come -- bases in read order, then the op's penalty, a soft clip's length x ln 0.25: the order of
// score_one_generic (score_alignments.hip), which A1 / A1c reproduce.  Nothing but the scores (and, while stage 3 reads them there, the
// records' copy in set order) goes back to HBM.
// A read this form does not hold (longer than F5_MAX_READ bases, pool over F5_MAX_POOL bytes) is counted in *n_unhandled: the caller
// then runs F1-F3 + A1c over the job.
constexpr int F5_MAX_READ = 256;  // (as F3's wave form; longer reads and larger pools take the staged chain)
constexpr int ONE_WAIT_MAX_READS = 16384; // reads of a job that runs as one fixed sequence (its buffers are sized by capacities: 256 leaves per read)
constexpr int F5_MAX_POOL = 768;
#ifndef F5_SORT
#define F5_SORT 1 // experiments: 0 = the set's order
#endif
constexpr int F5_TAB = 32;       // span of table indices the indels of one round's candidate alignments may cover

static_assert(F5_TAB <= 64, "the table copy is a lane an entry");

struct FusedScoreArgs
{
    FlatArgs f;
    const uint8_t* read_qual;
    const SkTables* tab;
    double* scores;
    unsigned* err;         // SkContext::dev_error_flags (a quality above 70)
    int32_t* n_unhandled;  // reads left to the staged kernels
    int32_t write_cals;    // the records' copy in set order (stage 3 and sk_enum_device_fetch_cals read it)
    unsigned long long* dbg; // diagnostics ($SK_F5_TIMING): per block 8 cycle stamps, or null
    unsigned* queue;       // [2] the launch's read queue: next read, waves that have left (both 0 between launches: the last wave out zeroes them)
};

// a candidate alignment's slot in LDS: the walk's output -- up to F5_SEGS + 1 transitions, one per op that covers read positions and
// one for the read's end (start position | penalties that precede the op's terms << 9 | soft clip << 15 | (pool offset - position + 256)
// << 16) -- then the record's indel indices.  The record's header and path are in the lane's registers (F5Rec).  21 words a lane
// instead of round 3's 39: with the table entries cut to the five words the walk reads and the rows of terms sized for the job's
// longest read, a wave's LDS goes from 17.8 KB to under 10 KB and a CU holds 16 of them instead of 9 (profiles/r04_a5_history.txt: a
// wave alone on a CU takes as long as nine sharing it; with sixteen the vector unit is ~85 % busy, profiles/r05_f5_history.txt).
constexpr int F5_SEGS = 16, F5_INDELS = 8;
constexpr int F5_IND0 = F5_SEGS + 1, F5_SLOT = (F5_IND0 + F5_INDELS / 2) | 1; // (odd stride: conflict-free)
__device__ __forceinline__ int f5_ent_word(const int k) { return k; }
constexpr int F5_ROW = 2; // doubles per read position: agree, differ (a position that adds nothing reads the shared 0.0 instead)

struct F5Tab // what the walk reads of a table entry
{
    int32_t pos;
    uint32_t del, ins_len;
    uint32_t type_cand; // type | cand << 8
    int32_t ins_at;     // where the entry's insert sequence starts in the read's pool, -1 = it is not there (ins_idx / ins_off by entry)
};

template <int MAXR>
struct F5Lds
{
    double row[(MAXR + 8) * F5_ROW];
    double zero; // 0.0
    uint64_t live[9]; // entry m: 0x01 in the m low bytes (the positions of an eight-base step that belong to the op)
    uint8_t read[MAXR + 8];
    uint8_t hap[F5_MAX_POOL + 8];
    uint32_t slot[64 * F5_SLOT];
    // what the walk looks up per path segment (a chain of dependent look-ups: from HBM / L2 they cost a wave ~100 us per round)
    F5Tab tab[F5_TAB];
};
// (the CU hands LDS out in 1 280-byte pieces: 10 240 bytes are sixteen waves to a CU, one byte more fourteen)
static_assert(sizeof(F5Lds<152>) <= 10240, "sixteen waves of the short-read form to a CU");
static_assert(INS_CAP <= 64, "a lane an insert");

// A compact record's path in the lane's registers: the walk looks at a segment's type up to a dozen times per turn (the edge
// segments, the look-ahead for swaps, the step to the next segment) and every look from LDS is a round trip on the lane's chain.  The
// sixteen types are 4 bits each in one 64-bit value, the sixteen lengths 16 bits each in four; segment i is a shift by i (the lengths:
// a select among the four values first).  The indel indices stay in the slot (read once, at the walk's start).
struct F5Rec
{
    uint32_t w0, w1, w2;  // pos; lead | trail << 16; fwd | n_seg << 8 | n_indels << 16
    uint64_t types;       // segment i: bits [4i, 4i + 4)
    uint64_t l0, l1, l2, l3; // segment i: bits [16 (i & 3), + 16) of l(i >> 2)  (four values, not an array: an array indexed per lane
                          // goes to scratch memory)
    const uint32_t* slot; // the lane's slot: indel indices at F5_IND0
    __device__ __forceinline__ int32_t pos() const { return int32_t(w0); }
    __device__ __forceinline__ int lead() const { return int(int16_t(w1 & 0xffffu)); }
    __device__ __forceinline__ int trail() const { return int(int16_t(w1 >> 16)); }
    __device__ __forceinline__ int n_seg() const { return int((w2 >> 8) & 0xffu); }
    __device__ __forceinline__ int n_indels() const { return int((w2 >> 16) & 0xffu); }
    __device__ __forceinline__ int indel(const int k) const { return int(int16_t((slot[F5_IND0 + (k >> 1)] >> (16 * (k & 1))) & 0xffffu)); }
};

// flatten_cal over a compact record, every look-up from the block's LDS copies: the walk of scoreCandidateAlignment :286-493 as
// host/align_flatten.cpp states it, as ONE loop of selects.  Written with the reference's four branches (swap, sequence mismatch, insert,
// delete; match; soft clip; skip / hard clip) a turn is ~1 000 instructions, two thirds of them the scalar unit's mask bookkeeping, and
// the lanes of a wave take the branches in turn (round 4's form, profiles/r05_f5_history.txt).  Here what the branches have in common is
// computed once for every lane and selected: the segment kinds as bit sets over the path (one bit per 4-bit type: a run of insert /
// delete segments, the first and last match segment are shifts and bit counts, not loops), getMatchingIndelKey over packed entries
// (position; del << 16 | ins) up to the largest count in the wave, the insert's place in the pool from the table copy (F5Tab::ins_at).
// Only a swap (insert + delete run: rare) keeps a branch of its own.  The walk writes the alignment's transitions into its slot:
// one word per op that covers read positions and one for the read's end; returns their number, -1 = leave the read to the host form.
template <typename LDS>
__device__ __forceinline__ int f5_walk_selects(LDS& S, const F5Rec c, const bool has, const int tab_lo, const int32_t win_begin, uint8_t* consulted,
                                               const int32_t L, const int32_t P, uint32_t* const myslot)
{
    constexpr uint64_t N1 = 0x1111111111111111ull;
    const uint64_t types = c.types, p_l0 = c.l0, p_l1 = c.l1, p_l2 = c.l2, p_l3 = c.l3;
    const int aps = has ? c.n_seg() : 0;
    auto eq4 = [=](const unsigned v) -> uint64_t { // bit 4 i: segment i is of type v
        uint64_t x = types ^ (N1 * v);
        x |= x >> 1;
        x |= x >> 2;
        return ~x & N1;
    };
    auto seg_len = [=](const int i) -> unsigned {
        const uint64_t lo = (i & 4) ? p_l1 : p_l0, hi = (i & 4) ? p_l3 : p_l2;
        return unsigned(((i & 8) ? hi : lo) >> (16 * (i & 3))) & 0xffffu;
    };
    const uint64_t in_path = (aps >= 16) ? N1 : (((1ull << (4 * aps)) - 1ull) & N1);
    const uint64_t M4 = (eq4(SK_SEG_MATCH) | eq4(SK_SEG_SEQ_MATCH) | eq4(SK_SEG_SEQ_MISMATCH)) & in_path;
    const uint64_t I4 = eq4(SK_SEG_INSERT) & in_path, D4 = eq4(SK_SEG_DELETE) & in_path;
    // get_match_edge_segments, align_path.cpp:735-752
    const int ends_first = M4 ? (__builtin_ctzll(M4) >> 2) : aps, ends_second = M4 ? ((63 - __builtin_clzll(M4)) >> 2) : aps;
    // the alignment's indels, read once: position and the two lengths as one word (del << 16 | ins; an entry that is a breakpoint, or no
    // entry, holds a word no segment asks for); lengths of 0xffff and more are left to the host form
    const int ni = c.n_indels();
    int ni_wave = 0; // (the most indels any of the wave's alignments holds: the entries past it are not looked at)
#pragma unroll
    for (int k = 1; k <= F5_INDELS; ++k) ni_wave = __any(ni >= k) ? k : ni_wave;
    int32_t k_pos[F5_INDELS];
    uint32_t k_pack[F5_INDELS];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < F5_INDELS; ++k) {
        k_pos[k] = INT_MIN;
        k_pack[k] = 0xffffffffu;
        if (k < ni_wave) {
            const F5Tab& ci = S.tab[((k < ni) ? c.indel(k) : tab_lo) - tab_lo];
            const unsigned ty = ci.type_cand & 0xffu;
            const bool kind = (ty == SK_INDEL_INDEL || ty == SK_INDEL_MISMATCH);
            if (k < ni) {
                k_pos[k] = ci.pos;
                if (kind) {
                    k_pack[k] = (ci.del << 16) | ci.ins_len;
                    bad = bad || ci.del >= 0xffffu || ci.ins_len >= 0xffffu;
                }
            }
        }
    }
    int pi = 0, n_ent = 0;
    int32_t pos = 0, ref_head_pos = c.pos(); // (read_offset of the reference's walk)
    unsigned npen = 0;
    for (;;) {
        const bool live = pi < aps && !bad;
        if (!__any(live)) break;
        const int sh = 4 * (pi & 15);
        const unsigned ty = unsigned(types >> sh) & 15u, ln = seg_len(pi & 15);
        // the run of insert / delete segments that starts here (is_segment_swap_start, align_path.cpp:868-895: a swap = both kinds in it)
        const uint64_t not_run = ~((I4 | D4) >> sh) & N1;
        const int run4 = not_run ? __builtin_ctzll(not_run) : 64;
        const uint64_t run_set = (run4 >= 64) ? ~0ull : ((1ull << run4) - 1ull);
        const bool is_swap = live && ((I4 >> sh) & run_set) != 0ull && ((D4 >> sh) & run_set) != 0ull;
        const bool is_match = (ty == SK_SEG_MATCH || ty == SK_SEG_SEQ_MATCH), is_clip = (ty == SK_SEG_SOFT_CLIP);
        const bool is_indel_seg = is_swap || ty == SK_SEG_SEQ_MISMATCH || ty == SK_SEG_INSERT || ty == SK_SEG_DELETE;
        unsigned del_len = (ty == SK_SEG_INSERT) ? 0u : ln, ins_len = (ty == SK_SEG_DELETE) ? 0u : ln;
        int n_seg = 1;
        unsigned read_step = (is_match || ty == SK_SEG_SEQ_MISMATCH || ty == SK_SEG_INSERT || is_clip) ? ln : 0u; // increment_path, align_path_util.hh:38-68
        unsigned ref_step = (is_match || ty == SK_SEG_SEQ_MISMATCH || ty == SK_SEG_DELETE || ty == SK_SEG_SKIP) ? ln : 0u;
        if (is_swap) { // swap_info, align_path_util.hh:75-106
            n_seg = run4 >> 2;
            del_len = ins_len = 0;
            for (int k = 0; k < n_seg; ++k) {
                const unsigned l = seg_len((pi + k) & 15);
                if ((I4 >> (4 * ((pi + k) & 15))) & 1ull) ins_len += l;
                else del_len += l;
            }
            read_step = ins_len;
            ref_step = del_len;
        }
        // getMatchingIndelKey, starling_read_align_score.cpp:177-228: the edge indels outside the match segments; between them the one
        // entry of the alignment's that is this indel -- the reference stops at a second match and at the first entry past the position
        const unsigned asked = (del_len << 16) | ins_len;
        const bool askable = del_len < 0xffffu && ins_len < 0xffffu;
        int k_found = 0, n_found = 0;
        bool open = true;
#pragma unroll
        for (int k = 0; k < F5_INDELS; ++k) {
            if (k < ni_wave) {
                const bool m = open && k_pos[k] == ref_head_pos && k_pack[k] == asked;
                k_found = m ? k : k_found;
                n_found += m ? 1 : 0;
                open = open && (m || k_pos[k] <= ref_head_pos);
            }
        }
        int key = (n_found == 1) ? c.indel(k_found) : -2;
        key = (pi < ends_first) ? c.lead() : (pi > ends_second) ? c.trail() : key;
        const bool key_ok = key >= 0;
        const F5Tab te = S.tab[(key_ok ? key : tab_lo) - tab_lo];
        const bool indel_ok = live && is_indel_seg && key_ok;
        if (indel_ok && consulted) consulted[key] = 1; // job_cand
        const bool pen = is_indel_seg && (te.type_cand >> 8) == 0u;
        const int32_t head = (pi < ends_first) ? int32_t(te.ins_len) - int32_t(ln) : 0;
        // the op: bases of the pool (a match segment's window bytes, an indel's insert sequence), a soft clip, or a penalty alone
        const bool ins_bases = is_indel_seg && ins_len > 0u;
        const bool ins_ok = ins_len <= 0xffffu && head >= 0 && uint32_t(head) + ins_len <= te.ins_len && te.ins_at >= 0;
        const bool bases = ins_bases || is_match;
        const int32_t src = ins_bases ? te.ins_at + head : ref_head_pos - win_begin;
        const unsigned len = ins_bases ? ins_len : ln;
        const bool known = is_indel_seg || is_match || is_clip || ty == SK_SEG_SKIP || ty == SK_SEG_HARD_CLIP;
        bool fail = !known || (is_indel_seg && (!key_ok || !askable)) || (ins_bases && !ins_ok);
        const bool entry = (bases || (is_clip && !is_indel_seg)) && len > 0u;
        if (entry) {
            fail = fail || pos + int32_t(len) > L || (bases && (src < 0 || int64_t(src) + int64_t(len) > int64_t(P)));
            const int hidx = bases ? int(src) - pos : 0;
            fail = fail || n_ent > F5_SEGS || npen > 4u || hidx < -256 || hidx > 3839;
            if (live && !fail) myslot[f5_ent_word(n_ent)] = unsigned(pos) | (npen << 9) | (bases ? 0u : 1u << 15) | (unsigned(hidx + 256) << 16);
        }
        if (live) {
            bad = fail;
            n_ent += entry ? 1 : 0;
            npen = entry ? (pen ? 1u : 0u) : npen + (pen ? 1u : 0u);
            pos += int32_t(read_step);
            ref_head_pos += int32_t(ref_step);
            pi += n_seg;
        }
    }
    if (!has) return 0;
    if (bad || pos != L || n_ent > F5_SEGS || npen > 4u) return -1;
    myslot[f5_ent_word(n_ent)] = unsigned(L) | (npen << 9) | (256u << 16); // trailing penalties
    return n_ent + 1;
}

// Phase B of F5, a lane its alignment, in path order (the order of score_one_generic): entering an op, the penalties that precede its
// terms, then a soft clip's length x ln 0.25; inside an op of bases, eight positions per turn -- eight read codes against the eight pool
// bytes they face (SWAR), each position's address: its row's agree or differ term, or the shared 0.0 (N, or past the op's end) -- eight
// reads, eight adds.  ONE loop per lane (eight bases and the step to the next op in the same turn), so that the turns a wave makes are the
// longest lane's, not the sum over ops of the longest op.  PLAIN: the read holds no '=' and no N (the block's lanes share the read: the
// kernel asks once), the two tests are left out.  What is rare for a whole wave -- a penalty, a soft clip -- sits behind a vote.
template <bool PLAIN, typename LDS>
__device__ __forceinline__ double f5_sum(LDS& S, const uint32_t* const myslot, const int n_ent, const double ln_noncand, const double ln_quarter)
{
    double lnp = 0.0;
    const unsigned zero_at = unsigned(reinterpret_cast<const unsigned char*>(&S.zero) - reinterpret_cast<const unsigned char*>(S.row));
    const unsigned char* rows = reinterpret_cast<const unsigned char*>(S.row);
    int e = 0, p = 0, stop = 0, hidx = 0;
    // the step to the next op, by selects (the loop has one back edge); true: the read's end
    auto next_op = [&]() -> bool {
        const bool adv = (p >= stop);
        const uint32_t cur = myslot[f5_ent_word(e)];
        const int e1 = (e + 1 < n_ent) ? e + 1 : e;
        const int next_start = int(myslot[f5_ent_word(e1)] & 0x1ffu);
        const unsigned np = adv ? ((cur >> 9) & 63u) : 0u; // (at most 4: the walk hands anything longer to the host form)
        if (__any(np != 0u)) {
            const double l1 = __dadd_rn(lnp, ln_noncand);
            lnp = (np >= 1u) ? l1 : lnp;
            const double l2 = __dadd_rn(lnp, ln_noncand);
            lnp = (np >= 2u) ? l2 : lnp;
            const double l3 = __dadd_rn(lnp, ln_noncand);
            lnp = (np >= 3u) ? l3 : lnp;
            const double l4 = __dadd_rn(lnp, ln_noncand);
            lnp = (np >= 4u) ? l4 : lnp;
        }
        if (adv && e + 1 >= n_ent) return true;
        const int start = int(cur & 0x1ffu);
        const bool clip = adv && (cur & (1u << 15)) != 0u;
        if (__any(clip)) {
            const double lc = __dadd_rn(lnp, __dmul_rn(double(unsigned(next_start - start)), ln_quarter));
            lnp = clip ? lc : lnp;
        }
        p = adv ? (clip ? next_start : start) : p;
        stop = adv ? next_start : stop;
        hidx = adv ? int(cur >> 16) - 256 : hidx;
        e = adv ? e1 : e;
        return false;
    };
    if (next_op()) return lnp; // (the first op: no bases before it)
    for (;;) {
        {
            const int m = stop - p; // bases of the current op still to add (0: a soft clip just stepped over)
            // (32-bit halves: positions 0-3, 4-7; every byte holds a 4-bit code, so a byte-wise add never carries across bytes)
            uint32_t R[2], H[2], live[2];
            __builtin_memcpy(R, S.read + p, 8);
            __builtin_memcpy(H, S.hap + (p + hidx), 8);
            __builtin_memcpy(live, &S.live[m > 8 ? 8 : m], 8);
            constexpr uint32_t B01 = 0x01010101u, B7F = 0x7f7f7f7fu, B71 = 0x71717171u;
            uint32_t none[2], dif8[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t ne = (((R[h] ^ H[h]) + B7F) >> 7) & B01; // 1: the bytes differ
                if (PLAIN) {
                    none[h] = live[h] ^ B01;
                    dif8[h] = ne << 3;                                  // per byte: 8 = the differ term, 0 = the agree term
                } else {
                    const uint32_t nz = ((R[h] + B7F) >> 7) & B01;      // 1: the read base is not '='
                    const uint32_t any = ((R[h] + B71) >> 7) & B01;     // 1: the read base is N (code 15)
                    none[h] = any | (live[h] ^ B01);
                    dif8[h] = (ne & nz) << 3;
                }
            }
            const unsigned base = unsigned(8 * F5_ROW * p);
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned nb = (none[u >> 2] >> (8 * (u & 3))) & 0xffu, db = (dif8[u >> 2] >> (8 * (u & 3))) & 0xffu;
                const unsigned at = nb ? zero_at : base + unsigned(8 * F5_ROW * u) + db;
                v[u] = *reinterpret_cast<const double*>(rows + at);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) lnp = __dadd_rn(lnp, v[u]);
            p += (m > 8) ? 8 : m;
        }
        if (next_op()) break;
    }
    return lnp;
}

// TIMING: the cycle stamps of $SK_F5_TIMING (fa.dbg); without it the stamps are constants and their s_memtime + waits are gone
// a block is F5_WAVES waves, each with a read and an LDS object of its own: what orders a wave's LDS accesses is the wave (its LDS
// instructions complete in order), so the points where the lanes exchange data through LDS need the compiler's and the counters'
// attention, not a workgroup barrier
__device__ __forceinline__ void f5_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int MAXR, bool TIMING>
__device__ __forceinline__ void f5_read(const FusedScoreArgs& fa, const int r, F5Lds<MAXR>& S)
{
    auto now = [&]() -> unsigned long long { return TIMING ? (unsigned long long)clock64() : 0ull; };
    const FlatArgs& a = fa.f;
    const int lane = threadIdx.x & 63;
    const int c0 = a.cal_off[r], c1 = a.cal_off[r + 1];
    const int ncr = c1 - c0;
    if (ncr == 0 || a.status[r] != ST_OK) return;
    unsigned long long stamp[8];
    stamp[0] = now();
    const unsigned long long wall0 = TIMING ? (unsigned long long)wall_clock64() : 0ull; // (the constant 100 MHz counter: the block's life in time)
    for (int i = 1; i < 8; ++i) stamp[i] = 0;
    const int64_t ro = a.read_off[r];
    const int32_t L = int32_t(a.read_off[r + 1] - ro);
    const int32_t P = a.hap_len[r];
    if (L > MAXR || P > F5_MAX_POOL) {
        if (lane == 0) atomicAdd(fa.n_unhandled, 1);
        return;
    }
    const int32_t win_begin = a.win_begin[r];
    const int n_ins = a.n_ins[r];
    int my_ins_idx = -1;    // lane k < n_ins: the table index of the read's insert k and where its sequence starts in the pool
    int32_t my_ins_off = 0;
    bool odd_code = false;  // the read holds a '=' or an N somewhere (phase B's plain form leaves those two tests out)
    // ---- the pool's bytes (as pool_fill_kernel), the read, its rows of terms, the pool's layout
    {
        const int16_t* idx = a.ins_idx + size_t(r) * INS_CAP;
        const int32_t* off = a.ins_off + size_t(r) * INS_CAP;
        // (insert k's length and source stay in lane k's registers -- the fill below takes them by readlane; its table index and
        // offset go into the round's table copy, F5Tab::ins_at)
        int32_t my_ins_len = 0;
        uint32_t my_ins_src = 0;
        if (lane < n_ins) {
            const int t_idx = idx[lane];
            my_ins_idx = t_idx;
            my_ins_off = off[lane];
            my_ins_len = int32_t(a.job.tab[t_idx].ins_len);
            my_ins_src = a.job.tab[t_idx].ins_off;
        }
        const int32_t win_len = a.win_len[r];
        // (the window's bytes first: they do not depend on the insert table just written)
        for (int32_t i = lane; i < P + 8; i += 64) {
            uint8_t v = SK_BAM_ANY;
            if (i < win_len) {
                const int32_t p = win_begin + i; // reference_contig_segment::get_base :46-51
                const bool outside = (p < a.ref_offset || p >= a.ref_offset + a.ref_len);
                if (outside && a.ref_outside) atomicAdd(a.ref_outside, 1);
                v = outside ? uint8_t(SK_BAM_ANY) : code_of(a.ref[p - a.ref_offset]);
            }
            S.hap[i] = v;
        }
        f5_wave_sync();
        for (int k = 0; k < n_ins; ++k) { // the insert sequences, in table order (a later one overwrites an earlier one, as pool_fill_kernel)
            const int32_t o = __builtin_amdgcn_readlane(my_ins_off, k), n = __builtin_amdgcn_readlane(my_ins_len, k);
            const uint32_t src = uint32_t(__builtin_amdgcn_readlane(int(my_ins_src), k));
            for (int32_t i = lane; i < n; i += 64)
                if (o + i < P) S.hap[o + i] = code_of(a.ins_pool[src + uint32_t(i)]);
        }
        const SkTables* __restrict__ T = fa.tab;
        if (lane == 0) S.zero = 0.0;
        if (lane < 9) S.live[lane] = (lane >= 8) ? 0x0101010101010101ull : (0x0101010101010101ull & ((1ull << (8 * lane)) - 1ull));
        for (int32_t i = lane; i < L + 8; i += 64) {
            unsigned q = 0;
            uint8_t code = SK_BAM_ANY;
            if (i < L) {
                code = a.read_code[ro + i];
                q = fa.read_qual[ro + i];
                if (q > 70u) { // the reference throws (qscore_cache.cpp:53-75): flagged, sk_check_device_errors reports it
                    atomicOr(fa.err, unsigned(SK_DEVERR_QSCORE));
                    q = 70u;
                }
            }
            odd_code = odd_code || (i < L && ((code & 15u) == 0u || (code & 15u) == 15u));
            S.read[i] = code & 15u;
            S.row[F5_ROW * i] = T->q2lncompe[q];
            S.row[F5_ROW * i + 1] = T->q2mis[q];
        }
    }
    const double ln_quarter = fa.tab->ln_quarter, ln_noncand = fa.tab->ln_noncand;
    const bool plain_codes = !__any(odd_code);

    // The read's candidate alignments in order of path length, longest first (a counting sort over the segment counts, the order in the
    // unused tail of the pool's bytes): a wave's walk lasts as long as its longest path, so a round of like paths wastes fewer turns
    // and the last, partly filled round gets the short ones.  Reads with more than 256 candidate alignments keep the set's order.
    uint8_t* const order = S.hap + ((P + 8 + 3) & ~3);
    const bool sorted_order = (F5_SORT != 0) && ncr > 64 && ncr <= 256 && ((P + 8 + 3) & ~3) + ncr <= int(sizeof(S.hap));
    if (sorted_order) {
        int seg_of[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = lane + 64 * k;
            seg_of[k] = (j < ncr) ? int(a.n_seg8[c0 + j]) : -1; // (left by pool_bounds_kernel, which reads every record anyway)
            seg_of[k] = min(seg_of[k], F5_SEGS + 1);
        }
        // (only the segment counts that occur: a read's alignments have a handful of distinct ones -- the loop over all eighteen was ~650
        // of a read's ~7 800 instructions)
        unsigned present = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) present |= (seg_of[k] >= 0) ? (1u << seg_of[k]) : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) present |= unsigned(__shfl_xor(int(present), d));
        present = unsigned(__builtin_amdgcn_readfirstlane(int(present)));
        int at = 0; // alignments placed so far
        while (present != 0u) { // descending, as the loop over every count was
            const int v = 31 - __builtin_clz(present);
            present &= ~(1u << v);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t m = __ballot(seg_of[k] == v);
                if (seg_of[k] == v) order[at + __popcll(m & ((1ull << lane) - 1ull))] = uint8_t(lane + 64 * k);
                at += __popcll(m);
            }
        }
    }

    for (int j0 = 0; j0 < ncr; j0 += 64) {
        const int nc = min(64, ncr - j0);
        f5_wave_sync(); // (pool and read complete; the previous round's slots read; the order written)
        const unsigned long long ta = now();
        if (j0 == 0) stamp[1] = ta; // prologue done
        // ---- the round's records, a lane its own: header + F5_SEGS path segments (five 16-byte loads) + F5_INDELS indel indices, all in
        // flight at once; a record with more of either is left to the staged chain
        const bool has = lane < nc;
        uint32_t* const myslot = S.slot + lane * F5_SLOT;
        bool fits = true;
        const int my_j = has ? (sorted_order ? int(order[j0 + lane]) : j0 + lane) : 0; // the lane's alignment among the read's
        F5Rec rec;
        rec.w0 = rec.w1 = rec.w2 = 0;
        rec.types = 0;
        rec.l0 = rec.l1 = rec.l2 = rec.l3 = 0;
        rec.slot = myslot;
        if (has) {
            const PCal* src = a.pool + a.list[c0 + my_j];
            const uint4* s4 = reinterpret_cast<const uint4*>(src);
            const uint4 q0 = s4[0], q1 = s4[1], q2 = s4[2], q3 = s4[3], q4 = s4[4];
            const uint32_t* si = reinterpret_cast<const uint32_t*>(src->indels);
            const uint32_t i0 = si[0], i1 = si[1], i2 = si[2], i3 = si[3];
            myslot[F5_IND0] = i0; myslot[F5_IND0 + 1] = i1; myslot[F5_IND0 + 2] = i2; myslot[F5_IND0 + 3] = i3;
            rec.w0 = q0.x;
            rec.w1 = q0.y;
            rec.w2 = q0.z;
            const unsigned nseg = (q0.z >> 8) & 0xffu, nind = (q0.z >> 16) & 0xffu;
            fits = (nseg <= unsigned(F5_SEGS) && nind <= unsigned(F5_INDELS));
            uint64_t types = 0;
            bool odd_type = false; // (not a segment type: the staged chain reports it)
            auto pack4 = [&](const uint32_t a0, const uint32_t a1, const uint32_t a2, const uint32_t a3, const int first) -> uint64_t {
                types |= (uint64_t(a0 & 15u) | (uint64_t(a1 & 15u) << 4) | (uint64_t(a2 & 15u) << 8) | (uint64_t(a3 & 15u) << 12)) << (4 * first);
                odd_type = odd_type || (unsigned(first) < nseg && (a0 & 0xffffu) > 15u) || (unsigned(first + 1) < nseg && (a1 & 0xffffu) > 15u) ||
                           (unsigned(first + 2) < nseg && (a2 & 0xffffu) > 15u) || (unsigned(first + 3) < nseg && (a3 & 0xffffu) > 15u);
                return uint64_t(a0 >> 16) | (uint64_t(a1 >> 16) << 16) | (uint64_t(a2 >> 16) << 32) | (uint64_t(a3 >> 16) << 48);
            };
            rec.l0 = pack4(q0.w, q1.x, q1.y, q1.z, 0);
            rec.l1 = pack4(q1.w, q2.x, q2.y, q2.z, 4);
            rec.l2 = pack4(q2.w, q3.x, q3.y, q3.z, 8);
            rec.l3 = pack4(q3.w, q4.x, q4.y, q4.z, 12);
            rec.types = types;
            fits = fits && !odd_type;
        }
        // the table entries this round's alignments name
        int tab_lo = 0;
        {
            int lo = INT_MAX, hi = INT_MIN;
            if (has && fits) {
                auto add = [&](const int i) {
                    lo = min(lo, i);
                    hi = max(hi, i);
                };
                const int ni = rec.n_indels();
                for (int i = 0; i < ni; ++i) add(rec.indel(i));
                if (rec.lead() >= 0) add(rec.lead());
                if (rec.trail() >= 0) add(rec.trail());
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                lo = min(lo, __shfl_xor(lo, d));
                hi = max(hi, __shfl_xor(hi, d));
            }
            const int n_tab = (lo <= hi) ? hi - lo + 1 : 0;
            // (a record beyond the compact form, indices further apart than the copy holds: the staged chain takes the job)
            if (__any(has && !fits) || (n_tab > 0 && lo < 0) || n_tab > F5_TAB) {
                if (lane == 0) atomicAdd(fa.n_unhandled, 1);
                return;
            }
            tab_lo = n_tab ? lo : 0;
            if (lane < n_tab) { // (n_tab <= F5_TAB <= 64: a lane an entry)
                const PIndel& g = a.job.tab[tab_lo + lane];
                F5Tab e;
                e.pos = g.pos;
                e.del = g.del;
                e.ins_len = g.ins_len;
                e.type_cand = unsigned(g.type) | (unsigned(g.cand) << 8);
                e.ins_at = -1;
                S.tab[lane] = e;
            }
            f5_wave_sync();
            // (the read's inserts are distinct table indices, pool_layout_kernel: a lane an insert, no two write the same entry)
            if (my_ins_idx >= tab_lo && my_ins_idx < tab_lo + n_tab) S.tab[my_ins_idx - tab_lo].ins_at = my_ins_off;
        }
        f5_wave_sync();
        const unsigned long long tc = now();
        stamp[2] += tc - ta; // staging + table copy
        // ---- phase A, a lane per candidate alignment: the walk of its path leaves the alignment's TRANSITIONS in its slot: one word per
        // op that covers read positions, and one for the read's end
        int n_ent = 0;
        bool bad = false;
        {
            const unsigned long long tA0 = now();
            n_ent = f5_walk_selects(S, rec, has, tab_lo, win_begin, a.job.consulted, L, P, myslot);
            const unsigned long long tA1 = now();
            stamp[7] += tA1 - tA0; // the walk
            bad = n_ent < 0;
            if (has) {
                // the candidate-status lookups the host form performs for every indel of the alignment (cal_to_c)
                if (a.job.consulted) {
                    const int ni = rec.n_indels();
                    for (int i = 0; i < ni; ++i) a.job.consulted[rec.indel(i)] = 1;
                    if (rec.lead() >= 0) a.job.consulted[rec.lead()] = 1;
                    if (rec.trail() >= 0) a.job.consulted[rec.trail()] = 1;
                }
                if (bad) a.status[r] = ST_FAIL;
            }
            (void)tA1;
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned long long td = now();
        stamp[4] += td - tc; // phase A
        // ---- phase B, a lane its alignment: the terms in path order (f5_sum)
        if (has && !bad) {
            const double lnp = plain_codes ? f5_sum<true>(S, myslot, n_ent, ln_noncand, ln_quarter) : f5_sum<false>(S, myslot, n_ent, ln_noncand, ln_quarter);
            fa.scores[c0 + my_j] = lnp;
        }
        stamp[5] += now() - td; // phase B
    }
    if (fa.dbg && lane == 0) {
        stamp[6] = now();
        stamp[3] = TIMING ? (unsigned long long)wall_clock64() - wall0 : 0ull;
        for (int i = 0; i < 8; ++i) fa.dbg[size_t(r) * 8 + i] = stamp[i];
    }
}

// A wave per read, WAVES waves to a block (each wave with an LDS object of its own; nothing is shared between them).  With a block per
// read the kernel was bound by the rate at which the dispatcher places workgroups (65 536 one-wave blocks went out at ~30 per us, a block
// lives ~58 us: ~7 of a CU's 16 slots filled); eight waves to a block: 2.23 -> 1.67 ms per 65 536 reads (profiles/r05_f5_history.txt).
// The waves TAKE their reads from a queue (one counter, one atomic per read; a grid of the blocks the device holds at a time).  With read r
// bound to wave r a block lived as long as its heaviest read -- a read's work goes with its candidate alignments, a round of the wave per
// 64 -- and its other seven slots stood empty meanwhile: 7.5 waves to a CU on average (SQ_WAVE_CYCLES, profiles/r06_v28_pmc_traffic.json)
// where two such blocks, 16 waves, are resident (tools/diag/lds_residency.hip).  1.67 -> 1.23 ms.  Round 5's persistent grid (reads r, r + 4 096, ... to wave r: slower) had the
// same binding, only longer.  The queue's order is the job's: taking the heavy reads first, or the light ones last (lists by rounds of 64
// candidate alignments, filled by pool_layout_kernel), was tried against the launch's tail and is SLOWER (1.33-1.37 ms: reads of one kind
// side by side keep their waves in step -- all in their prologues, all in their walks -- where the job's own order mixes them;
// profiles/r06_f5_history.txt).  The last wave to leave puts the queue back to zero, so that the next launch -- the same buffers, the same
// stream -- finds it as this one did.
template <int MAXR, bool TIMING, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void flatten_score_kernel(const FusedScoreArgs fa, const int n_reads)
{
    __shared__ __attribute__((aligned(16))) F5Lds<MAXR> S[WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (;;) {
        int r = 0;
        if (lane == 0) r = int(atomicAdd(&fa.queue[0], 1u));
        r = __builtin_amdgcn_readfirstlane(r);
        if (r >= n_reads) break;
        f5_read<MAXR, TIMING>(fa, r, S[wave]);
        f5_wave_sync(); // (the wave's LDS object is the next read's)
    }
    if (lane == 0) {
        __threadfence();
        if (atomicAdd(&fa.queue[1], 1u) == gridDim.x * unsigned(WAVES) - 1u) { // (every wave has made its last draw)
            fa.queue[0] = 0u;
            fa.queue[1] = 0u;
        }
    }
}

// the rows of terms sized for the job's longest read: with 150-base reads a wave's LDS is 10 KB, sixteen waves to a CU (the 256-base
// form: twelve)
// eight waves to a block where their LDS leaves room for two blocks to a CU (the short-read form: 76.7 KB), four with the long form
template <int MAXR, bool TIMING>
static void launch_flatten_score_t(const int n_reads, hipStream_t st, const FusedScoreArgs& fs)
{
    static const int waves = [] { const char* e = std::getenv("SK_F5_WAVES"); return e ? std::atoi(e) : 0; }(); // (experiments: 1, 2, 4, 8, 16; 0 = the default)
    constexpr int DEFAULT_WAVES = (sizeof(F5Lds<MAXR>) * 16 <= 160 * 1024) ? 8 : 4;
    const int w = waves > 0 ? waves : DEFAULT_WAVES;
    // the blocks the device holds at a time (256 CUs; by LDS: 16 waves of the short-read form to a CU, 14 of the long one; a grid larger
    // than that only queues blocks that find the read queue empty, a smaller one leaves slots unused): $SK_F5_GRID = blocks, experiments
    static const int grid_env = [] { const char* e = std::getenv("SK_F5_GRID"); return e ? std::atoi(e) : 0; }();
    constexpr int WAVES_PER_CU = int((160 * 1024) / sizeof(F5Lds<MAXR>)) >= 16 ? 16 : int((160 * 1024) / sizeof(F5Lds<MAXR>));
    auto grid = [&](const int wpb) {
        const int resident = grid_env > 0 ? grid_env : 256 * std::max(1, WAVES_PER_CU / wpb);
        return dim3(unsigned(std::max(1, std::min(resident, (n_reads + wpb - 1) / wpb))));
    };
    if (w == 16 && MAXR <= 152) SK_LAUNCH((flatten_score_kernel<152, TIMING, 16>), grid(16), dim3(1024), 0, st, fs, n_reads);
    else if (w >= 8) SK_LAUNCH((flatten_score_kernel<MAXR, TIMING, 8>), grid(8), dim3(512), 0, st, fs, n_reads);
    else if (w == 4) SK_LAUNCH((flatten_score_kernel<MAXR, TIMING, 4>), grid(4), dim3(256), 0, st, fs, n_reads);
    else if (w == 2) SK_LAUNCH((flatten_score_kernel<MAXR, TIMING, 2>), grid(2), dim3(128), 0, st, fs, n_reads);
    else SK_LAUNCH((flatten_score_kernel<MAXR, TIMING, 1>), grid(1), dim3(64), 0, st, fs, n_reads);
}

static void launch_flatten_score(const int n_reads, hipStream_t st, const FusedScoreArgs& fs)
{
    const bool short_reads = fs.f.max_read_len <= 152, timing = fs.dbg != nullptr;
    if (short_reads && !timing) launch_flatten_score_t<152, false>(n_reads, st, fs);
    else if (short_reads) launch_flatten_score_t<152, true>(n_reads, st, fs);
    else if (!timing) launch_flatten_score_t<F5_MAX_READ, false>(n_reads, st, fs);
    else launch_flatten_score_t<F5_MAX_READ, true>(n_reads, st, fs);
}

// the records in set order, for the host (sk_enum_device_fetch_cals; a job whose stage 3 runs on the host): pool[list[c]] -> cals[c], a wave
// per 64 records, rows of consecutive dwords
__global__ __launch_bounds__(64) void gather_cals_kernel(const PCal* __restrict__ pool, const int32_t* __restrict__ list, const int32_t n_cals,
                                                         PCal* __restrict__ cals)
{
    constexpr int REC_DW = int(sizeof(PCal) / 4);
    const int lane = threadIdx.x;
    const int c0 = blockIdx.x * 64;
    const int nc = min(64, n_cals - c0);
    const int src_k = (lane < nc) ? list[c0 + lane] : 0;
    for (int k0 = 0; k0 < nc; k0 += 16) {
        uint32_t v0[16], v1[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + u;
            const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(pool + __shfl(src_k, (k < nc) ? k : 0));
            v0[u] = (k < nc) ? src[lane] : 0u;
            v1[u] = (k < nc && lane + 64 < REC_DW) ? src[lane + 64] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + u;
            if (k < nc) {
                uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(cals + (c0 + k));
                dst[lane] = v0[u];
                if (lane + 64 < REC_DW) dst[lane + 64] = v1[u];
            }
        }
    }
}

// F3: ops -> transition entries + the read's event mask (host form: align_prepare in host/align_flatten.cpp; layout
// csrc/align_entry.h), then the column form (host form: sk_align_prepare_cols).  Two kernels produce the same bytes:
//   entries_kernel       one THREAD per candidate alignment walks its ops and then every read position (a dependent byte load of
//                        the haplotype pool per position);
//   entries_wave_kernel  one WAVE per read: the lanes emit the entries of their candidate alignments, the base comparisons are
//                        made once per distinct (haplotype offset, read position) -- a read's candidates face the pool at a
//                        handful of offsets, one per combination of indels before a position -- as packed 8-position words in
//                        LDS, and a lane assembles its alignment's column words from those with masks.

// the entries of candidate alignment c (slots ent[0 .. nslots)); returns true if the alignment is left to the generic routine
// ("complex"), in which case its column words stay as preset.  `hap`: the read's pool (P bytes).  n_ent: entries written.
__device__ inline bool f3_emit_entries(const FlatArgs& a, const int c, const int r, const uint8_t* hap, const int32_t L, const int32_t P,
                                       uint32_t* ent, int& n_ent)
{
    const int W = a.evmask_words;
    uint32_t* mask = a.evmask + int64_t(r) * W;
    const bool read_ok = (L >= 0 && L <= SK_ENT_MAX_READ_LEN && L <= a.max_read_len && P >= 0 && P <= SK_ENT_MAX_POOL);
    auto col_at = [&](const int idx) -> unsigned { return (idx >= 0 && idx < P) ? sk_ent_col_index(hap[idx]) : unsigned(SK_ENT_ZERO_COL); };
    const int64_t k0 = a.op_off[c], k1 = a.op_off[c + 1];
    const int nslots = int(k1 - k0) + 2;
    for (int i = 0; i < nslots; ++i) ent[i] = SK_ENT_END;
    bool complex_cal = !read_ok;
    int e = 0, pos = 0;
    unsigned npen = 0;
    auto emit = [&](const unsigned np, const bool clip, const int hidx) {
        if (np > 7u || hidx + SK_ENT_HIDX_BIAS < 0 || hidx + SK_ENT_HIDX_BIAS > 2047) complex_cal = true;
        if (complex_cal) return;
        ent[e++] = unsigned(pos) | (np << 10) | (clip ? 1u << 13 : 0u) | (col_at(hidx + pos) << 15) | (col_at(hidx + pos + 1) << 18) |
                   (unsigned(hidx + SK_ENT_HIDX_BIAS) << 21);
        if (pos > 0 && pos <= L) atomicOr(&mask[pos >> 5], 1u << (pos & 31));
    };
    for (int64_t kk = k0; kk < k1 && !complex_cal; ++kk) {
        const sk_score_op op = a.ops[kk];
        const int len = int(op.length);
        const unsigned pen = op.flags & SK_OPFLAG_NONCANDIDATE_PENALTY;
        if ((op.kind == SK_OP_BASES || op.kind == SK_OP_SOFT_CLIP) && len > 0) {
            if (pos >= int(SK_ENT_END)) {
                complex_cal = true;
                break;
            }
            const bool bases = (op.kind == SK_OP_BASES);
            if (bases && (op.src < 0 || int64_t(op.src) + len > P)) complex_cal = true;
            emit(npen, !bases, bases ? int(op.src) - pos : P - pos);
            pos += len;
            npen = pen;
        } else {
            npen += pen;
        }
    }
    if (!complex_cal) {
        if (pos != L || pos >= int(SK_ENT_END)) complex_cal = true;
        else emit(npen, false, P - pos);
    }
    if (complex_cal) {
        for (int i = 0; i < nslots; ++i) ent[i] = SK_ENT_END;
        ent[0] = SK_ENT_COMPLEX;
        atomicOr(&a.addmask[int64_t(r) * W + (W - 1)], 1u << 31);
        n_ent = 0;
        return true;
    }
    // where entries add terms
    uint32_t* am = a.addmask + int64_t(r) * W;
    for (int i = 0; i < e; ++i)
        if (ent[i] & SK_ENT_ADD_BITS) {
            const unsigned p = ent[i] & SK_ENT_POS_MASK;
            atomicOr(&am[p >> 5], 1u << (p & 31));
        }
    n_ent = e;
    return false;
}

// the column words of candidate alignment c, position by position (the serial form)
__device__ inline void f3_columns_serial(const FlatArgs& a, const int c, const int r, const uint8_t* hap, const int32_t L, const int32_t P,
                                         const uint32_t* ent, const int e)
{
    const int W = a.evmask_words;
    auto col_at = [&](const int idx) -> unsigned { return (idx >= 0 && idx < P) ? sk_ent_col_index(hap[idx]) : unsigned(SK_ENT_ZERO_COL); };
    const int64_t k0 = a.op_off[c], k1 = a.op_off[c + 1];
    // one word (eight read positions) at a time: assembled in a register, stored once
    const int ncr = a.cal_off[r + 1] - a.cal_off[r], j = c - a.cal_off[r];
    uint32_t* cm = reinterpret_cast<uint32_t*>(a.colmat) + a.colmat_off[r] + j;
    const uint8_t* read = a.read_code + a.read_off[r];
    constexpr uint32_t NONE_WORD = 0x11111111u * SK_SEL_NONE;
    uint32_t word = NONE_WORD;
    int wk = 0; // index of the word being assembled
    int pos = 0;
    const int nch = (L + 7) >> 3;
    int ei = 0; // next entry that adds terms (entries and positions both ascend)
    bool needs_entries = false;
    // an entry adding exactly one penalty at position i becomes bit 2 of that position's nibble; anything else leaves the read to
    // its entries (host form and rationale: sk_align_prepare_cols)
    auto flag_at = [&](const int i) -> unsigned {
        while (ei < e && int(ent[ei] & SK_ENT_POS_MASK) < i) ++ei;
        if (ei < e && int(ent[ei] & SK_ENT_POS_MASK) == i && (ent[ei] & SK_ENT_ADD_BITS)) {
            const bool simple = ((ent[ei] >> 10) & 7u) == 1u && !(ent[ei] & (1u << 13)) && i < 8 * nch;
            if (simple) return 4u;
            needs_entries = true;
        }
        return 0u;
    };
    auto put = [&](const int i, const unsigned sel) {
        const int k8 = i >> 3;
        if (k8 != wk) {
            cm[int64_t(wk) * ncr] = word;
            for (int q = wk + 1; q < k8; ++q) cm[int64_t(q) * ncr] = NONE_WORD; // (words a soft clip spans)
            word = NONE_WORD;
            wk = k8;
        }
        const unsigned shift = 8u * (unsigned(i) & 3u) + ((i & 4) ? 4u : 0u);
        word = (word & ~(0xfu << shift)) | (sel << shift);
    };
    for (int64_t kk = k0; kk < k1; ++kk) {
        const sk_score_op op = a.ops[kk];
        const int len = int(op.length);
        if (!((op.kind == SK_OP_BASES || op.kind == SK_OP_SOFT_CLIP) && len > 0)) continue;
        if (op.kind == SK_OP_BASES) {
            for (int t = 0; t < len && pos + t < L; ++t)
                put(pos + t, sk_col_selector(col_at(int(op.src) + t), read[pos + t]) | flag_at(pos + t));
        } else if (flag_at(pos)) { // (a soft clip's entry is never the simple kind; flag_at notes that the entries are needed)
        }
        pos += len;
    }
    {
        const unsigned f = flag_at(L); // trailing penalties: the nibble after the last base, where the last word has one
        if (f) put(L, SK_SEL_NONE | f);
    }
    for (; ei < e; ++ei) // (entries no position above stood for)
        if ((ent[ei] & SK_ENT_ADD_BITS) && int(ent[ei] & SK_ENT_POS_MASK) > L) needs_entries = true;
    if (needs_entries) atomicOr(&a.addmask[int64_t(r) * W + (W - 1)], 1u << 30);
    if (wk < nch) cm[int64_t(wk) * ncr] = word;
    for (int q = wk + 1; q < nch; ++q) cm[int64_t(q) * ncr] = NONE_WORD;
}

__global__ __launch_bounds__(64) void entries_kernel(const FlatArgs a)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.n_cals) return;
    const int r = read_of_cal(a, c);
    const int32_t L = int32_t(a.read_off[r + 1] - a.read_off[r]);
    const int32_t P = int32_t(a.hap_off[r + 1] - a.hap_off[r]);
    const uint8_t* hap = a.hap_code + a.hap_off[r];
    uint32_t* ent = a.entries + a.op_off[c] + 2 * int64_t(c);
    int e = 0;
    if (f3_emit_entries(a, c, r, hap, L, P, ent, e)) return;
    f3_columns_serial(a, c, r, hap, L, P, ent, e);
}

constexpr int F3W_MAX_D = 64;    // distinct haplotype offsets of a read's candidate alignments the wave form holds
constexpr int F3W_MAX_NCH = 32;  // reads of up to 256 bases
constexpr int F3W_MAX_POOL = 1024;

struct F3wLds
{
    uint8_t hap[F3W_MAX_POOL];
    uint8_t read[8 * F3W_MAX_NCH];
    uint8_t slot_of[2048];           // haplotype offset + SK_ENT_HIDX_BIAS -> its slot in M
    uint32_t delta_bits[64];         // which of the 2048 offsets occur
    int32_t n_slots;
    int16_t delta_of[F3W_MAX_D];     // slot -> haplotype offset
    uint32_t M[F3W_MAX_D][F3W_MAX_NCH];  // the column word of (offset, 8 read positions)
    uint32_t tile[F3W_MAX_NCH][64];      // the column words of 64 candidate alignments
};

// positions [0, x) of an 8-position column word (position i sits at bit 8 (i & 3) + 4 (i >> 2 & 1))
__device__ __forceinline__ uint32_t f3w_below(const int x)
{
    const uint32_t lo = (x >= 4) ? 0x0f0f0f0fu : (uint32_t((1ull << (8 * x)) - 1ull) & 0x0f0f0f0fu);
    const uint32_t hi = (x <= 4) ? 0u : (x >= 8 ? 0xf0f0f0f0u : (uint32_t((1ull << (8 * (x - 4))) - 1ull) & 0xf0f0f0f0u));
    return lo | hi;
}

__global__ __launch_bounds__(64) void entries_wave_kernel(const FlatArgs a)
{
    __shared__ F3wLds S;
    const int r = blockIdx.x;
    const int lane = threadIdx.x;
    const int c0 = a.cal_off[r], c1 = a.cal_off[r + 1];
    const int ncr = c1 - c0;
    if (ncr == 0) return;
    const int32_t L = int32_t(a.read_off[r + 1] - a.read_off[r]);
    const int32_t P = int32_t(a.hap_off[r + 1] - a.hap_off[r]);
    const uint8_t* ghap = a.hap_code + a.hap_off[r];
    const uint8_t* gread = a.read_code + a.read_off[r];
    const int nch = (L + 7) >> 3;
    const bool fits = (L >= 0 && nch <= F3W_MAX_NCH && P >= 0 && P <= F3W_MAX_POOL);
    if (fits) {
        for (int i = lane; i < P; i += 64) S.hap[i] = ghap[i];
        for (int i = lane; i < 8 * nch; i += 64) S.read[i] = (i < L) ? gread[i] : uint8_t(15);
    }
    S.delta_bits[lane] = 0;
    if (lane == 0) S.n_slots = 0;
    __syncthreads();
    const uint8_t* hap = fits ? S.hap : ghap;

    // ---- the entries of every candidate alignment; the haplotype offsets their base ops use
    for (int c = c0 + lane; c < c1; c += 64) {
        uint32_t* ent = a.entries + a.op_off[c] + 2 * int64_t(c);
        int e = 0;
        const bool complex_cal = f3_emit_entries(a, c, r, hap, L, P, ent, e);
        if (complex_cal) continue;
        if (!fits) { // (a read the LDS form does not hold: position by position, as the thread-per-alignment kernel)
            f3_columns_serial(a, c, r, hap, L, P, ent, e);
            continue;
        }
        int pos = 0;
        for (int64_t kk = a.op_off[c]; kk < a.op_off[c + 1]; ++kk) {
            const sk_score_op op = a.ops[kk];
            const int len = int(op.length);
            if (!((op.kind == SK_OP_BASES || op.kind == SK_OP_SOFT_CLIP) && len > 0)) continue;
            if (op.kind == SK_OP_BASES) {
                const int d = int(op.src) - pos + SK_ENT_HIDX_BIAS; // (0..2047: f3_emit_entries turned anything else down)
                atomicOr(&S.delta_bits[d >> 5], 1u << (d & 31));
            }
            pos += len;
        }
    }
    if (!fits) return;
    __syncthreads();
    // ---- a slot per distinct offset (ascending)
    {
        const uint32_t w = S.delta_bits[lane];
        int n = __popc(w), before = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(before, d, 64);
            if (lane >= d) before += up;
        }
        const int total = __shfl(before, 63, 64);
        before -= n;
        if (lane == 0) S.n_slots = total;
        if (total <= F3W_MAX_D) {
            uint32_t ww = w;
            int slot = before;
            while (ww) {
                const int b = __ffs(int(ww)) - 1;
                ww &= ww - 1u;
                S.delta_of[slot] = int16_t(32 * lane + b - SK_ENT_HIDX_BIAS);
                S.slot_of[32 * lane + b] = uint8_t(slot++);
            }
        }
    }
    __syncthreads();
    const int D = S.n_slots;
    if (D > F3W_MAX_D) { // (more offsets than the table holds: the serial form for this read)
        for (int c = c0 + lane; c < c1; c += 64) {
            const uint32_t* ent = a.entries + a.op_off[c] + 2 * int64_t(c);
            if (ent[0] == SK_ENT_COMPLEX) continue;
            int e = 0;
            while ((ent[e] & SK_ENT_POS_MASK) != SK_ENT_END) ++e;
            f3_columns_serial(a, c, r, hap, L, P, ent, e);
        }
        return;
    }
    // ---- M: every (offset, 8 positions) once
    constexpr uint32_t NONE_WORD = 0x11111111u * SK_SEL_NONE;
    for (int t = lane; t < D * nch; t += 64) {
        const int slot = t / nch, k = t - slot * nch;
        const int delta = S.delta_of[slot];
        uint32_t word = NONE_WORD;
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {
            const int p = 8 * k + t8;
            if (p < L) {
                const int idx = p + delta;
                const unsigned col = (idx >= 0 && idx < P) ? sk_ent_col_index(S.hap[idx]) : unsigned(SK_ENT_ZERO_COL);
                const unsigned sel = sk_col_selector(col, S.read[p]);
                const unsigned shift = 8u * (unsigned(t8) & 3u) + ((t8 & 4) ? 4u : 0u);
                word = (word & ~(0xfu << shift)) | (sel << shift);
            }
        }
        S.M[slot][k] = word;
    }
    __syncthreads();
    // ---- the column words of the candidate alignments, 64 at a time
    const int W = a.evmask_words;
    uint32_t* cm_base = reinterpret_cast<uint32_t*>(a.colmat) + a.colmat_off[r];
    for (int j0 = 0; j0 < ncr; j0 += 64) {
        const int j = j0 + lane;
        for (int k = 0; k < nch; ++k) S.tile[k][lane] = NONE_WORD;
        if (j < ncr) {
            const int c = c0 + j;
            const uint32_t* ent = a.entries + a.op_off[c] + 2 * int64_t(c);
            if (ent[0] != SK_ENT_COMPLEX) {
                int pos = 0;
                for (int64_t kk = a.op_off[c]; kk < a.op_off[c + 1]; ++kk) {
                    const sk_score_op op = a.ops[kk];
                    const int len = int(op.length);
                    if (!((op.kind == SK_OP_BASES || op.kind == SK_OP_SOFT_CLIP) && len > 0)) continue;
                    if (op.kind == SK_OP_BASES) {
                        const int slot = S.slot_of[int(op.src) - pos + SK_ENT_HIDX_BIAS];
                        const int p0 = pos, p1 = min(pos + len, int(L));
                        for (int k = p0 >> 3; p0 < p1 && k <= (p1 - 1) >> 3; ++k) {
                            const int lo = max(p0, 8 * k) - 8 * k, hi = min(p1, 8 * k + 8) - 8 * k;
                            const uint32_t m = f3w_below(hi) & ~f3w_below(lo);
                            S.tile[k][lane] = (S.tile[k][lane] & ~m) | (S.M[slot][k] & m);
                        }
                    }
                    pos += len;
                }
                // an entry adding exactly one penalty becomes bit 2 of its position's nibble; anything else leaves the read to its
                // entries (f3_columns_serial's flag_at: every entry sits at the start of a base op, of a soft clip, or at L)
                bool needs_entries = false;
                for (int ei = 0; (ent[ei] & SK_ENT_POS_MASK) != SK_ENT_END; ++ei) {
                    if (!(ent[ei] & SK_ENT_ADD_BITS)) continue;
                    const int i = int(ent[ei] & SK_ENT_POS_MASK);
                    const bool simple = ((ent[ei] >> 10) & 7u) == 1u && !(ent[ei] & (1u << 13)) && i < 8 * nch && i <= L;
                    if (simple) {
                        const unsigned shift = 8u * (unsigned(i) & 3u) + ((i & 4) ? 4u : 0u);
                        S.tile[i >> 3][lane] |= 4u << shift;
                    } else {
                        needs_entries = true;
                    }
                }
                if (needs_entries) atomicOr(&a.addmask[int64_t(r) * W + (W - 1)], 1u << 30);
            }
        }
        // (a lane reads and writes only its own column of the tile)
        if (j < ncr)
            for (int k = 0; k < nch; ++k) cm_base[int64_t(k) * ncr + j] = S.tile[k][lane];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// S3: stage 3 (csrc/stage3_core.h) -- the max / smooth candidate alignment, the realigned alignment clipped where the smooth
// pool disagrees, the late normalisation filter and score_indels -- one wavefront per read over the read's candidate alignments and
// scores where the scoring kernel left them: the loops over alignments are spread over the 64 lanes, lane 0 takes the decisions the
// reference defines by iteration order (see the header).  What leaves the device is a fixed-size record per read.
enum { S3_LIGHT_CALS = 256, S3_LDS_CALS = 1580 }; // (1580 alignments: 53 KB; with the shared state that stays under 64 KB per workgroup)

struct WaveLanes // the lanes of a workgroup: one wavefront for most reads, four for reads with many candidate alignments
{
    int id, width;
    __device__ void sync() const { __syncthreads(); }
    __device__ void max_i64(long long* p, const long long v) const { atomicMax(p, v); }
};

struct Stage3Args
{
    sk3::Tab tab;
    sk3::Opt opt;
    int32_t n_reads;
    const int32_t* status;
    const int32_t* cal_off;
    const PCal* cals;    // the records in set order -- or, with `slot`, the pool the search left its leaves in
    const int32_t* slot; // null, or [n_cals] the pool slot of every alignment in set order
    const double* scores;
    const int64_t* read_off;
    const uint8_t* read_code;
    const int32_t* map_level;
    int32_t* order;
    double* smooth;
    uint8_t* flag;
    uint32_t* key;
    double* sorted_score;
    uint32_t* sorted_hash;
    int32_t* next_same;
    int32_t* range_end;
    uint8_t* removed;
    uint8_t* rm_type;
    int32_t* rm_pos;
    sk3::Out* out;
    const int32_t* list; // the reads of this launch
    int32_t lds_cals;
    // the job as one fixed sequence: the launch has a block per read of the JOB, the number of reads on this launch's list is on the
    // device (n_list; the heavy list is filled from the back: list_step = -1), and nothing runs when F5 turned a read down (skip_if:
    // the scores are then incomplete -- the host runs the job again through the staged chain)
    const int32_t* n_list;
    int32_t list_step;
    const int32_t* skip_if;
};

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void stage3_kernel(const Stage3Args a)
{
    constexpr int T = 64 * WAVES;
    __shared__ sk3::Shared sh;
    extern __shared__ double s3_lds[]; // the per-alignment arrays of the selection and the late normalisation filter, for a read with at most a.lds_cals candidate alignments
    if (a.skip_if && *a.skip_if > 0) return;
    if (a.n_list && int(blockIdx.x) >= *a.n_list) return;
    const int r = a.list[int(blockIdx.x) * (a.n_list ? a.list_step : 1)];
    sk3::Out& o = a.out[r];
    const int32_t c0 = a.cal_off[r], c1 = a.cal_off[r + 1];
    if (a.status[r] != ST_OK || c1 == c0) {
        if (threadIdx.x == 0) o.status = sk3::S3_CAPACITY;
        return;
    }
    const int64_t b0 = a.read_off[r], b1 = a.read_off[r + 1];
    sk3::Read rd;
    rd.cals = a.slot ? sk3::CalView{ a.cals, a.slot + c0 } : sk3::CalView{ a.cals + c0, nullptr };
    rd.scores = a.scores + c0;
    rd.scores_select = rd.scores;
    rd.n_cals = c1 - c0;
    rd.map_level = a.map_level[r];
    rd.read_length = int32_t(b1 - b0);
    if (threadIdx.x == 0) sh.ne = 0; // (borrowed as the counter of bases that are not N)
    __syncthreads();
    {
        int na = 0;
        for (int64_t i = b0 + threadIdx.x; i < b1; i += T) na += (a.read_code[i] != SK_BAM_ANY) ? 1 : 0;
        for (int d = 32; d > 0; d >>= 1) na += __shfl_xor(na, d);
        if ((threadIdx.x & 63) == 0) atomicAdd(&sh.ne, na);
    }
    __syncthreads();
    rd.non_ambig = sh.ne;
    __syncthreads();
    sk3::Scratch w;
    w.order = a.order + c0;
    w.smooth = a.smooth + c0;
    w.flag = a.flag + c0;
    w.key = a.key + 4 * size_t(c0);
    w.sorted_score = a.sorted_score + c0;
    w.sorted_hash = a.sorted_hash + c0;
    w.next_same = a.next_same + c0;
    w.range_end = a.range_end + c0;
    w.removed = a.removed + c0;
    w.rm_type = a.rm_type + b0;
    w.rm_pos = a.rm_pos + b0;
    if (rd.n_cals <= a.lds_cals) {
        // what lane 0's scans and the ordering walk over and over sits in LDS (the accesses are chains of dependent loads)
        double* sc = s3_lds; // the scores for the selection, later the smoothed scores by place
        double* ssc = sc + a.lds_cals;
        int32_t* ord = reinterpret_cast<int32_t*>(ssc + a.lds_cals);
        uint32_t* shs = reinterpret_cast<uint32_t*>(ord + a.lds_cals);
        int32_t* nxt = reinterpret_cast<int32_t*>(shs + a.lds_cals);
        int32_t* rend = nxt + a.lds_cals;
        uint8_t* fl = reinterpret_cast<uint8_t*>(rend + a.lds_cals);
        uint8_t* rem = fl + a.lds_cals;
        for (int i = threadIdx.x; i < rd.n_cals; i += T) sc[i] = rd.scores[i];
        __syncthreads();
        rd.scores_select = sc;
        w.smooth = sc;
        w.sorted_score = ssc;
        w.sorted_hash = shs;
        w.next_same = nxt;
        w.range_end = rend;
        w.order = ord;
        w.flag = fl;
        w.removed = rem;
    }
    WaveLanes ln;
    ln.id = int(threadIdx.x);
    ln.width = T;
    sk3::finish_read(ln, a.tab, a.opt, rd, w, sh, o);
}

// ---------------------------------------------------------------------------------------------------------------------
// buffers: grown on demand, kept for the life of the process (one job after the other reuses them)

struct DevBuf
{
    void* p = nullptr;
    size_t cap = 0;
    int reserve(const size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)skrt::free_(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        SK_HIP(skrt::malloc_(&p, want));
        cap = want;
        return 0;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};
struct HostBuf // pinned
{
    void* p = nullptr;
    size_t cap = 0;
    int reserve(const size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)skrt::hostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        SK_HIP(skrt::hostMalloc(&p, want));
        cap = want;
        return 0;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct View // a piece of one of the arenas below
{
    void* p = nullptr;
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// The job's small arrays live in a few arenas so that a job is a handful of copies and fills, not dozens (each one costs 4-5 us on
// the stream whatever its size): everything that goes up before the search in `in_arena` (packed in a pinned mirror, one copy), every
// array that starts at zero in `zero_arena` (one fill; the parts the host reads come back in one copy per synchronisation point)
struct EnumBuffers
{
    DevBuf in_arena, zero_arena, minmax_arena, mask_arena;
    HostBuf h_in_arena, h_zero_arena;
    View tab, ins, toggle, ref, reads, read_off, read_code, read_qual, r2i, i2r, orig, map_level;
    View counters, status, warn, n_raw, hap_len, fill, n_uniq, consulted;
    View win_begin, ins_lo, win_end, ins_hi, evmask, addmask;
    View h_counters, h_status, h_warn, h_n_raw, h_hap_len, h_n_uniq, h_consulted;
    View cal_off_z, h_cal_off_z;
    DevBuf level_a, level_b, pool, leaf_read, leaf_hash;
    DevBuf raw_off, grouped, ghash, gkey, dup, sorted;
    DevBuf n_ops, win_len, n_ins, ins_idx, ins_off, n_seg8;
    DevBuf cal_off, hap_off, cals, hap_code, op_off, ops, entries, scores, colmat, colmat_off;
    DevBuf s3_order, s3_smooth, s3_flag, s3_rm_type, s3_rm_pos, s3_key, s3_sorted_score, s3_sorted_hash, s3_next_same, s3_range_end, s3_removed, s3_out, s3_list;
    HostBuf h_s3_out, h_s3_list;
    HostBuf h_raw_off, h_n_ops, h_cal_off, h_hap_off, h_op_off, h_cals, h_scores, h_colmat_off;
};
EnumBuffers& bufs()
{
    static EnumBuffers b;
    return b;
}

} // namespace

extern "C" int sk_enum_device_available(void) { return sk_ctx().ready ? 1 : 0; } // (host stages alone run without a device)

namespace
{
uint64_t g_generation = 0; // of the run whose candidate alignments the buffers hold
int32_t g_n_cals = 0;
bool g_cals_gathered = false; // bufs().cals holds the run's records in set order (else: pool + list, gathered when the host asks)
const PCal* g_cals_pool = nullptr;
const int32_t* g_cals_list = nullptr;
const double* g_scores = nullptr; // the run's scores, where the scoring kernel left them

// what sk_enum_device_rescore needs to run F1-F3 + the scoring kernel again on the candidate alignments of the last run
struct LastFlat
{
    bool valid = false;
    bool fused = false; // the last run scored with F5 (fs), else with the staged chain (fa, d)
    FusedScoreArgs fs;
    FlatArgs fa;
    sk_align_batch d;
    int32_t n = 0, n_cals = 0;
    int64_t cells = 0;
    size_t mask_bytes = 0, colmat_bytes = 0;
    double* scores = nullptr;
} g_last;
} // namespace

extern "C" int sk_enum_device_fetch_scores(const uint64_t generation, const int32_t first, const int32_t count, double* dst)
{
    if (generation != g_generation || first < 0 || count < 0 || first + count > g_n_cals || !g_scores) return 1;
    if (count == 0) return 0;
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    SK_HIP(skrt::memcpyAsync(dst, g_scores + first, 8 * size_t(count), hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(skrt::streamSynchronize(ctx.stream));
    return 0;
}

extern "C" int sk_enum_device_fetch_cals(const uint64_t generation, const int32_t first, const int32_t count, PCal* dst)
{
    if (generation != g_generation || first < 0 || count < 0 || first + count > g_n_cals) return 1;
    if (count == 0) return 0;
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    if (!g_cals_gathered) { // (F5 left the records where the search put them: in set order on demand, once per run)
        if (bufs().cals.reserve(sizeof(PCal) * size_t(g_n_cals))) return 1;
        SK_LAUNCH(gather_cals_kernel, dim3((g_n_cals + 63) / 64), dim3(64), 0, ctx.stream, g_cals_pool, g_cals_list, g_n_cals, bufs().cals.as<PCal>());
        SK_HIP(skrt::getLastError());
        g_cals_gathered = true;
    }
    SK_HIP(skrt::memcpyAsync(dst, bufs().cals.as<PCal>() + first, sizeof(PCal) * size_t(count), hipMemcpyDeviceToHost, ctx.stream));
    SK_HIP(skrt::streamSynchronize(ctx.stream));
    return 0;
}

namespace
{
// statistics of the process: jobs run as one fixed sequence (one wait), and those of them the host ran again the staged way because
// an assumption of the sequence did not hold (deeper levels than launched, a full leaf pool, a read outside F5's form)
int64_t g_jobs_one_wait = 0, g_jobs_one_wait_redone = 0, g_jobs_staged = 0;
// diagnostics ($SK_ENUM_JOB_SECONDS): where a one-sequence job's wall time goes -- buffers + packing, submissions, the wait -- on stderr at exit
struct JobSeconds
{
    double setup = 0, submit = 0, wait = 0, after = 0;
    ~JobSeconds()
    {
        if (std::getenv("SK_ENUM_JOB_SECONDS"))
            std::fprintf(stderr, "strelka_amd enum job seconds: jobs=%lld setup=%.4f submit=%.4f wait=%.4f after=%.4f\n", (long long)g_jobs_one_wait, setup, submit,
                         wait, after);
    }
} g_job_seconds;
}
extern "C" void sk_enum_device_job_counts(int64_t* one_wait, int64_t* one_wait_redone, int64_t* staged)
{
    if (one_wait) *one_wait = g_jobs_one_wait;
    if (one_wait_redone) *one_wait_redone = g_jobs_one_wait_redone;
    if (staged) *staged = g_jobs_staged;
}

// one_wait: the job as ONE fixed sequence of submissions and one host wait (see job_scan_kernel); *redo is set when the sequence's
// assumptions did not hold for this job and nothing of `out` is valid: the caller runs it again with one_wait = false
static int enum_device_run_impl(const SkEnumInput* in, SkEnumOutput* out, const bool one_wait, bool* redo)
{
    SK_REQUIRE_INIT();
    skrt::wakeHint();
    if (!in || !out) return sk_fail("sk_enum_device_run: null argument");
    std::memset(out, 0, sizeof(*out));
    out->generation = ++g_generation;
    g_n_cals = 0;
    g_scores = nullptr;
    g_cals_gathered = false;
    g_last.valid = false;
    const int n = in->n_reads;
    if (n < 0 || in->n_tab < 0) return sk_fail("sk_enum_device_run: negative count");
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    hipStream_t st = ctx.stream;
    EnumBuffers& B = bufs();
    const bool timing = std::getenv("SK_ENUM_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)skrt::streamSynchronize(st);
        std::fprintf(stderr, "[enum-dev] %-28s t=%.3f ms\n", what,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };

    const int64_t n_bases = n ? in->read_off[n] : 0;
    // capacities: calls in flight at one depth and leaves found (before duplicates are dropped), for the whole job; a job that
    // needs more has the reads that did not fit enumerated on the host
    // (sized by the job; the ceilings -- two level buffers of 2^23 frames, 2^25 leaves: ~21 GB in all -- are a fraction of 288 GB of HBM
    // and let a job of 2^16 reads with ~90 candidate alignments each run whole)
    int64_t frame_cap = std::min<int64_t>(std::max<int64_t>(int64_t(n) * 128, 1 << 18), 1 << 23);
    int64_t pool_cap = std::min<int64_t>(std::max<int64_t>(int64_t(n) * 512, 1 << 19), int64_t(32) << 20);
    if (const char* e = std::getenv("SK_ENUM_TEST_CAPS")) { // tests: small capacities, so that the overflow paths run
        long long fc = 0, pc = 0;
        if (std::sscanf(e, "%lld,%lld", &fc, &pc) == 2 && fc > 0 && pc > 0) {
            frame_cap = std::max<int64_t>(fc, n);
            pool_cap = pc;
        }
    }
    bool test_caps = false;
    if (const char* e = std::getenv("SK_ENUM_TEST_CAPS")) test_caps = (e[0] != 0);
    // one sequence: the buffers between the search and stage 3 are sized by the pool's capacity instead of by counts the host would have
    // to wait for -- 256 leaves per read (at least 2^17) cover every WGS-like job; one that needs more is run again the staged way
    if (one_wait && !test_caps) pool_cap = std::min<int64_t>(pool_cap, std::max<int64_t>(int64_t(n) * 256, 1 << 17));
    const int n_counters = Caps::K + 8 + DYN_COUNT + 2; // (+ 2: F5's read queue, FusedScoreArgs::queue)  // level counts [0, K+3), leaves, calls made (64 bit), F5's unhandled reads; the scans' results
#define RES(buf, bytes) \
    if (B.buf.reserve(std::max<size_t>(size_t(bytes), 256))) return 1
#define HRES(buf, bytes) \
    if (B.buf.reserve(size_t(bytes) + 64)) return 1
    RES(level_a, sizeof(PFrame) * size_t(frame_cap));
    RES(level_b, sizeof(PFrame) * size_t(frame_cap));
    RES(pool, sizeof(PCal) * size_t(pool_cap));
    RES(leaf_read, 4 * size_t(pool_cap));
    RES(leaf_hash, 4 * size_t(pool_cap));
    RES(raw_off, 4 * size_t(n + 1));
    RES(win_len, 4 * size_t(n));
    RES(n_ins, 4 * size_t(n));
    RES(ins_idx, 2 * size_t(n) * INS_CAP);
    RES(ins_off, 4 * size_t(n) * INS_CAP);
    RES(cal_off, 4 * size_t(n + 1));
    RES(hap_off, 8 * size_t(n + 1));
    HRES(h_raw_off, 4 * size_t(n + 1));
    HRES(h_cal_off, 4 * size_t(n + 1));
    HRES(h_hap_off, 8 * size_t(n + 1));
    const bool with_stage3 = in->want_scores && in->want_stage3;
    const int W = sk_ent_evmask_words(std::max(in->max_read_len, 0));
    // ---- the arenas: pieces at 256-byte offsets
    struct Piece
    {
        View* view;
        const void* src; // in_arena: where the bytes come from
        size_t bytes, off;
    };
    auto lay_out = [](Piece* pc, const int n_pieces) {
        size_t at = 0;
        for (int i = 0; i < n_pieces; ++i) {
            pc[i].off = at;
            at += (pc[i].bytes + 255) & ~size_t(255);
        }
        return at;
    };
    Piece ins_p[] = {
        { &B.tab, in->tab, sizeof(PIndel) * size_t(in->n_tab), 0 },
        { &B.ins, in->ins_pool, size_t(in->ins_pool_len), 0 },
        { &B.toggle, in->max_toggle, sizeof(uint32_t) * size_t(in->n_max_toggle), 0 },
        { &B.ref, in->ref, size_t(std::max(in->ref_len, 0)), 0 },
        { &B.reads, in->reads, sizeof(PRead) * size_t(n), 0 },
        { &B.read_off, in->read_off, sizeof(int64_t) * size_t(n + 1), 0 },
        { &B.read_code, in->read_code, size_t(n_bases), 0 },
        { &B.read_qual, in->read_qual, size_t(n_bases), 0 },
        { &B.r2i, in->r2i, with_stage3 ? 8 * size_t(in->n_tab) : 0, 0 },
        { &B.i2r, in->i2r, with_stage3 ? 8 * size_t(in->n_tab) : 0, 0 },
        { &B.orig, in->orig, with_stage3 ? 4 * size_t(in->n_tab) : 0, 0 },
        { &B.map_level, in->map_level, with_stage3 ? 4 * size_t(n) : 0, 0 },
    };
    const int n_ins_p = int(sizeof(ins_p) / sizeof(ins_p[0]));
    const size_t in_bytes = lay_out(ins_p, n_ins_p);
    // (the order matters: what the host reads after the search is one run, what it reads after the layout another)
    Piece zero_p[] = {
        { &B.counters, nullptr, 4 * size_t(n_counters), 0 }, { &B.warn, nullptr, 4 * size_t(n), 0 },    { &B.n_raw, nullptr, 4 * size_t(n), 0 },
        { &B.status, nullptr, 4 * size_t(n), 0 },            { &B.hap_len, nullptr, 4 * size_t(n), 0 }, { &B.fill, nullptr, 4 * size_t(n), 0 },
        { &B.n_uniq, nullptr, 4 * size_t(n), 0 },            { &B.consulted, nullptr, size_t(in->n_tab) + 1, 0 },
        { &B.cal_off_z, nullptr, one_wait ? 4 * size_t(n + 1) : 0, 0 }, // (one sequence: cal_off is made on the device and comes back with the arena)
    };
    View* zero_host[] = { &B.h_counters, &B.h_warn, &B.h_n_raw, &B.h_status, &B.h_hap_len, nullptr, &B.h_n_uniq, &B.h_consulted, &B.h_cal_off_z };
    const int n_zero_p = int(sizeof(zero_p) / sizeof(zero_p[0]));
    const size_t zero_bytes = lay_out(zero_p, n_zero_p);
    Piece minmax_p[] = { { &B.win_begin, nullptr, 4 * size_t(n), 0 }, { &B.ins_lo, nullptr, 4 * size_t(n), 0 },
                         { &B.win_end, nullptr, 4 * size_t(n), 0 },   { &B.ins_hi, nullptr, 4 * size_t(n), 0 } };
    const size_t minmax_bytes = lay_out(minmax_p, 4);
    Piece mask_p[] = { { &B.evmask, nullptr, 4 * (size_t(n) * size_t(W) + 1), 0 }, { &B.addmask, nullptr, 4 * (size_t(n) * size_t(W) + 1), 0 } };
    const size_t mask_bytes = lay_out(mask_p, 2);
    RES(in_arena, in_bytes);
    RES(zero_arena, zero_bytes);
    RES(minmax_arena, minmax_bytes);
    RES(mask_arena, mask_bytes);
    HRES(h_in_arena, in_bytes);
    HRES(h_zero_arena, zero_bytes);
    for (int i = 0; i < n_ins_p; ++i) {
        ins_p[i].view->p = static_cast<char*>(B.in_arena.p) + ins_p[i].off;
        if (ins_p[i].bytes) std::memcpy(static_cast<char*>(B.h_in_arena.p) + ins_p[i].off, ins_p[i].src, ins_p[i].bytes);
    }
    for (int i = 0; i < n_zero_p; ++i) {
        zero_p[i].view->p = static_cast<char*>(B.zero_arena.p) + zero_p[i].off;
        if (zero_host[i]) zero_host[i]->p = static_cast<char*>(B.h_zero_arena.p) + zero_p[i].off;
    }
    for (int i = 0; i < 4; ++i) minmax_p[i].view->p = static_cast<char*>(B.minmax_arena.p) + minmax_p[i].off;
    for (int i = 0; i < 2; ++i) mask_p[i].view->p = static_cast<char*>(B.mask_arena.p) + mask_p[i].off;
    // a run of the zero arena, device -> its pinned mirror
    auto fetch_zero = [&](const View& first, const View& last, const size_t last_bytes) -> int {
        const size_t a0 = size_t(static_cast<char*>(first.p) - static_cast<char*>(B.zero_arena.p));
        const size_t a1 = size_t(static_cast<char*>(last.p) - static_cast<char*>(B.zero_arena.p)) + last_bytes;
        SK_HIP(skrt::memcpyAsync(static_cast<char*>(B.h_zero_arena.p) + a0, static_cast<char*>(B.zero_arena.p) + a0, a1 - a0, hipMemcpyDeviceToHost, st));
        return 0;
    };

#define H2D(buf, src, bytes) \
    if ((bytes) > 0) SK_HIP(skrt::memcpyAsync(B.buf.p, src, size_t(bytes), hipMemcpyHostToDevice, st))
#define D2H(hbuf, buf, bytes) \
    if ((bytes) > 0) SK_HIP(skrt::memcpyAsync(B.hbuf.p, B.buf.p, size_t(bytes), hipMemcpyDeviceToHost, st))
    static const bool stage_by_kernel = !(std::getenv("SK_ENUM_STAGE_COPIES") != nullptr); // ($SK_ENUM_STAGE_COPIES: the copy / fill calls, for A-B runs)
    if (one_wait && stage_by_kernel) {
        const uint32_t n_in16 = uint32_t(in_bytes / 16), n_zero16 = uint32_t(zero_bytes / 16); // (pieces sit at 256-byte offsets)
        const int blocks = int(std::min<size_t>(std::max<size_t>((std::max(in_bytes, zero_bytes) / 16 + 255) / 256, 1), 256));
        SK_LAUNCH(job_stage_in_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const uint4*>(B.h_in_arena.p), static_cast<uint4*>(B.in_arena.p), n_in16,
                           static_cast<uint4*>(B.zero_arena.p), n_zero16);
    } else {
        if (in_bytes > 0) SK_HIP(skrt::memcpyAsync(B.in_arena.p, B.h_in_arena.p, in_bytes, hipMemcpyHostToDevice, st));
        SK_HIP(skrt::memsetAsync(B.zero_arena.p, 0, zero_bytes, st));
    }

    int32_t* h_cal_off = B.h_cal_off.as<int32_t>();
    h_cal_off[0] = 0;
    out->status = B.h_status.as<int32_t>();
    out->cal_off = h_cal_off;
    out->consulted = B.h_consulted.as<uint8_t>();
    if (n == 0) {
        std::memset(B.h_consulted.p, 0, size_t(in->n_tab));
        return 0;
    }
    if (n > frame_cap) return sk_fail("sk_enum_device_run: more reads in one job than the device pipeline holds");

    PJob dj;
    dj.tab = B.tab.as<PIndel>();
    dj.n_tab = in->n_tab;
    dj.max_toggle = B.toggle.as<uint32_t>();
    dj.n_max_toggle = in->n_max_toggle;
    dj.sample_count = in->sample_count;
    dj.max_read_indel_toggle = in->max_read_indel_toggle;
    dj.max_candidate_indel_density = in->max_candidate_indel_density;
    dj.is_haplotyping_enabled = in->is_haplotyping_enabled;
    dj.max_indel_size = in->max_indel_size;
    dj.consulted = B.consulted.as<uint8_t>();
    dj.max_nodes = 0;

    // ---- E1: the search, one launch per depth
    EnumArgs ea;
    ea.job = dj;
    ea.reads = B.reads.as<PRead>();
    ea.n_reads = n;
    ea.frame_cap = int32_t(frame_cap);
    ea.level_count = B.counters.as<int32_t>();
    ea.pool = B.pool.as<PCal>();
    ea.leaf_read = B.leaf_read.as<int32_t>();
    ea.leaf_hash = B.leaf_hash.as<uint32_t>();
    ea.pool_cap = int32_t(pool_cap);
    ea.n_leaves = B.counters.as<int32_t>() + (Caps::K + 3);
    ea.n_nodes = reinterpret_cast<unsigned long long*>(B.counters.as<int32_t>() + (Caps::K + 4) + ((Caps::K + 4) & 1));
    ea.n_raw = B.n_raw.as<int32_t>();
    ea.status = B.status.as<int32_t>();
    ea.warn = B.warn.as<int32_t>();
    ea.min_cells_a = ea.min_cells_b = ea.max_cells_a = ea.max_cells_b = nullptr;
    lap("H2D");
    const auto t_setup = std::chrono::steady_clock::now();
    if (one_wait) {
        // ---- the job as one fixed sequence: ~20 submissions, no host decision between them, ONE wait.  Everything the host used to
        // compute between waits is computed by job_scan_kernel; every buffer is sized by a capacity known now; launches whose size the
        // host does not know cover the capacity with a grid-stride loop (or a block per read of the job that leaves at once).
        const int32_t n_cap = int32_t(pool_cap); // leaves, grouped leaves and candidate alignments are all at most this many
        RES(grouped, 4 * size_t(n_cap));
        RES(ghash, 4 * size_t(n_cap));
        RES(gkey, 16 * size_t(n_cap));
        RES(dup, size_t(n_cap));
        RES(sorted, 4 * size_t(n_cap));
        RES(n_seg8, size_t(n_cap));
        RES(scores, 8 * size_t(n_cap));
        RES(s3_order, 4 * size_t(n_cap));
        RES(s3_smooth, 8 * size_t(n_cap));
        RES(s3_flag, size_t(n_cap));
        RES(s3_rm_type, size_t(n_bases));
        RES(s3_rm_pos, 4 * size_t(n_bases));
        RES(s3_key, 16 * size_t(n_cap));
        RES(s3_sorted_score, 8 * size_t(n_cap));
        RES(s3_sorted_hash, 4 * size_t(n_cap));
        RES(s3_next_same, 4 * size_t(n_cap));
        RES(s3_removed, size_t(n_cap));
        RES(s3_range_end, 4 * size_t(n_cap));
        RES(s3_out, sizeof(sk3::Out) * size_t(n));
        RES(s3_list, 4 * size_t(n));
        HRES(h_s3_out, sizeof(sk3::Out) * size_t(n));
        int32_t* const dyn = B.counters.as<int32_t>() + (Caps::K + 8);
        const int32_t* const h_dyn = B.h_counters.as<int32_t>() + (Caps::K + 8);
        int32_t* const n_unhandled = B.counters.as<int32_t>() + (Caps::K + 7);

        // E1: root + the levels up to the deepest order of the job (+ 2: indels that join an order on the way); whether calls were left
        // for a deeper level is looked at on the device (scan 1) and reported with the results
        PFrame* buf[2] = { B.level_a.as<PFrame>(), B.level_b.as<PFrame>() };
        ea.min_cells_a = B.win_begin.as<int32_t>();
        ea.min_cells_b = B.ins_lo.as<int32_t>();
        ea.max_cells_a = B.win_end.as<int32_t>();
        ea.max_cells_b = B.ins_hi.as<int32_t>();
        ea.level_in = nullptr;
        ea.level_out = buf[0];
        ea.depth = -1;
        SK_LAUNCH(root_kernel, dim3((n + 63) / 64), dim3(64), 0, st, ea);
        int max_order = 0;
        for (int r = 0; r < n; ++r) max_order = std::max(max_order, int(in->reads[r].n_order));
        int n_levels = std::min(max_order + 3, int(Caps::K) + 2);
        if (const char* e = std::getenv("SK_ENUM_TEST_LEVELS")) // tests: too few levels, so that the device reports it and the job runs again
            if (std::atoi(e) > 0) n_levels = std::min(n_levels, std::atoi(e));
        {
            const size_t lds = 64 * sizeof(PFrame);
            // (a level of this job has at most a few frames per read: blocks beyond that would only read the count and leave)
            const int blocks = int(std::min<int64_t>(std::min<int64_t>((frame_cap + 63) / 64, 1024), std::max<int64_t>(n, 16)));
            for (int d = 0; d < n_levels; ++d) {
                ea.level_in = buf[d & 1];
                ea.level_out = buf[(d + 1) & 1];
                ea.depth = d;
                SK_LAUNCH(level_kernel, dim3(blocks), dim3(64), lds, st, ea);
            }
        }
        // scan 1: raw_off
        ScanArgs sc;
        std::memset(&sc, 0, sizeof(sc));
        sc.n_reads = n;
        sc.status = ea.status;
        sc.count = ea.n_raw;
        sc.off = B.raw_off.as<int32_t>();
        sc.dyn = dyn;
        sc.level_count = ea.level_count;
        sc.first_level_not_launched = (n_levels >= int(Caps::K) + 2) ? -1 : n_levels;
        sc.n_leaves = ea.n_leaves;
        sc.pool_cap = n_cap;
        SK_LAUNCH(job_scan_kernel<false>, dim3(1), dim3(1024), 0, st, sc);
        // E2
        SetArgs sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.pool = ea.pool;
        sa.leaf_read = ea.leaf_read;
        sa.leaf_hash = ea.leaf_hash;
        sa.status = ea.status;
        sa.n_reads = n;
        sa.n_leaves = n_cap;
        sa.n_leaves_dev = ea.n_leaves;
        sa.raw_off = B.raw_off.as<int32_t>();
        sa.fill = B.fill.as<int32_t>();
        sa.grouped = B.grouped.as<int32_t>();
        sa.ghash = B.ghash.as<uint32_t>();
        sa.gkey = B.gkey.as<ulonglong2>();
        sa.dup = B.dup.as<uint8_t>();
        sa.n_uniq = B.n_uniq.as<int32_t>();
        sa.cal_off = B.cal_off_z.as<int32_t>();
        sa.sorted = B.sorted.as<int32_t>();
        // (grids for counts the host has not seen: enough blocks for ~64 leaves per read, the loops cover the rest)
        const int e2_blocks = int(std::min<int64_t>(std::max<int64_t>((int64_t(n) * 64 + 255) / 256, 8), 2048));
        SK_LAUNCH(group_kernel, dim3(e2_blocks), dim3(256), 0, st, sa);
        SK_LAUNCH(dedupe_kernel, dim3(e2_blocks), dim3(256), 0, st, sa);
        // scan 2: cal_off, stage 3's lists
        int light_cals = S3_LIGHT_CALS, lds_cals = S3_LDS_CALS;
        if (const char* e = std::getenv("SK_STAGE3_TEST_LDS_CALS")) {
            int a = 0, b = 0;
            if (std::sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b >= a && b <= S3_LDS_CALS) {
                light_cals = a;
                lds_cals = b;
            }
        }
        sc.count = B.n_uniq.as<int32_t>();
        sc.off = B.cal_off_z.as<int32_t>();
        sc.list = B.s3_list.as<int32_t>();
        sc.light_cals = light_cals;
        SK_LAUNCH(job_scan_kernel<true>, dim3(1), dim3(1024), 0, st, sc);
        SK_LAUNCH(rank_kernel, dim3(e2_blocks), dim3(256), 0, st, sa);
        // L1
        FlatArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.job = dj;
        fa.ins_pool = B.ins.as<char>();
        fa.ref = B.ref.as<char>();
        fa.ref_offset = in->ref_offset;
        fa.ref_len = in->ref_len;
        fa.n_reads = n;
        fa.n_cals = n_cap;
        fa.n_cals_on_device = 1;
        fa.pool = ea.pool;
        fa.list = B.sorted.as<int32_t>();
        fa.status = ea.status;
        fa.read_off = B.read_off.as<int64_t>();
        fa.read_code = B.read_code.as<uint8_t>();
        fa.cal_off = B.cal_off_z.as<int32_t>();
        fa.win_begin = B.win_begin.as<int32_t>();
        fa.win_len = B.win_len.as<int32_t>();
        fa.hap_len = B.hap_len.as<int32_t>();
        fa.n_ins = B.n_ins.as<int32_t>();
        fa.ins_idx = B.ins_idx.as<int16_t>();
        fa.ins_off = B.ins_off.as<int32_t>();
        fa.win_end = B.win_end.as<int32_t>();
        fa.ins_lo = B.ins_lo.as<int32_t>();
        fa.ins_hi = B.ins_hi.as<int32_t>();
        fa.n_seg8 = B.n_seg8.as<uint8_t>();
        fa.max_read_len = in->max_read_len;
        fa.ref_outside = dyn + DYN_REF_OUTSIDE;
        SK_LAUNCH(pool_bounds_kernel, dim3(e2_blocks), dim3(256), 0, st, fa);
        SK_LAUNCH(pool_layout_kernel, dim3((n + 63) / 64), dim3(64), 0, st, fa);
        // F5
        FusedScoreArgs fs;
        fs.f = fa;
        fs.read_qual = B.read_qual.as<uint8_t>();
        fs.tab = ctx.dev_tables;
        fs.scores = B.scores.as<double>();
        fs.err = ctx.dev_error_flags;
        fs.n_unhandled = n_unhandled;
        fs.queue = B.counters.as<unsigned>() + (Caps::K + 8 + DYN_COUNT);
        fs.write_cals = 0;
        fs.dbg = nullptr;
        launch_flatten_score(n, st, fs);
        // S3: a block per read of the job in either launch; the blocks past a list's length leave at once
        Stage3Args s3;
        s3.tab.tab = dj.tab;
        s3.tab.r2i = B.r2i.as<double>();
        s3.tab.i2r = B.i2r.as<double>();
        s3.tab.orig = B.orig.as<int32_t>();
        s3.tab.n_tab = in->n_tab;
        s3.tab.max_indel_size = in->max_indel_size;
        s3.tab.consulted = dj.consulted;
        s3.opt = in->stage3_opt;
        s3.n_reads = n;
        s3.status = ea.status;
        s3.cal_off = fa.cal_off;
        s3.cals = fa.pool;
        s3.slot = fa.list;
        s3.scores = B.scores.as<double>();
        s3.read_off = fa.read_off;
        s3.read_code = fa.read_code;
        s3.map_level = B.map_level.as<int32_t>();
        s3.order = B.s3_order.as<int32_t>();
        s3.smooth = B.s3_smooth.as<double>();
        s3.flag = B.s3_flag.as<uint8_t>();
        s3.rm_type = B.s3_rm_type.as<uint8_t>();
        s3.rm_pos = B.s3_rm_pos.as<int32_t>();
        s3.key = B.s3_key.as<uint32_t>();
        s3.sorted_score = B.s3_sorted_score.as<double>();
        s3.sorted_hash = B.s3_sorted_hash.as<uint32_t>();
        s3.next_same = B.s3_next_same.as<int32_t>();
        s3.removed = B.s3_removed.as<uint8_t>();
        s3.range_end = B.s3_range_end.as<int32_t>();
        s3.out = B.s3_out.as<sk3::Out>();
        s3.skip_if = n_unhandled;
        auto lds_bytes = [](const int cals) { return size_t(cals) * (8 + 8 + 4 + 4 + 4 + 4 + 1 + 1) + 8; };
        s3.list = B.s3_list.as<int32_t>();
        s3.n_list = dyn + DYN_N_LIGHT;
        s3.list_step = 1;
        s3.lds_cals = light_cals;
        SK_LAUNCH(stage3_kernel<1>, dim3(n), dim3(64), lds_bytes(light_cals), st, s3);
        s3.list = B.s3_list.as<int32_t>() + (n - 1);
        s3.n_list = dyn + DYN_N_HEAVY;
        s3.list_step = -1;
        s3.lds_cals = lds_cals; // (the largest read of the list is not known here: LDS for the most a block holds)
        SK_LAUNCH(stage3_kernel<4>, dim3(n), dim3(256), lds_bytes(lds_cals), st, s3);
        SK_HIP(skrt::getLastError());
        // the results: the zero arena whole (counters + the scans' numbers, warn, n_raw, status, ..., consulted, cal_off) and stage 3's records
        if (stage_by_kernel) {
            const size_t out_bytes = sizeof(sk3::Out) * size_t(n);
            const uint32_t n_a = uint32_t(zero_bytes / 16), n_b = uint32_t((out_bytes + 15) / 16); // (buffers are reserved with slack)
            const int blocks = int(std::min<size_t>(std::max<size_t>((std::max(zero_bytes, out_bytes) / 16 + 255) / 256, 1), 256));
            SK_LAUNCH(job_stage_out_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const uint4*>(B.zero_arena.p), static_cast<uint4*>(B.h_zero_arena.p), n_a,
                               static_cast<const uint4*>(B.s3_out.p), static_cast<uint4*>(B.h_s3_out.p), n_b);
            SK_HIP(skrt::getLastError());
        } else {
            SK_HIP(skrt::memcpyAsync(B.h_zero_arena.p, B.zero_arena.p, zero_bytes, hipMemcpyDeviceToHost, st));
            D2H(h_s3_out, s3_out, sizeof(sk3::Out) * size_t(n));
        }
        const auto t_submitted = std::chrono::steady_clock::now();
        SK_HIP(skrt::streamSynchronize(st));
        const auto t_waited = std::chrono::steady_clock::now();
        g_job_seconds.setup += std::chrono::duration<double>(t_setup - t_begin).count();
        g_job_seconds.submit += std::chrono::duration<double>(t_submitted - t_setup).count();
        g_job_seconds.wait += std::chrono::duration<double>(t_waited - t_submitted).count();
        lap("one sequence");
        const int32_t* h_counters = B.h_counters.as<int32_t>();
        if (h_dyn[DYN_FLAGS] != 0 || h_counters[Caps::K + 7] > 0) {
            if (timing) std::fprintf(stderr, "[enum-dev] one sequence: flags %d, reads outside F5's form %d -> the staged chain\n", h_dyn[DYN_FLAGS], h_counters[Caps::K + 7]);
            *redo = true;
            return 0;
        }
        ++g_jobs_one_wait;
        {
            uint8_t* w8 = reinterpret_cast<uint8_t*>(B.h_warn.p);
            const int32_t* w32 = B.h_warn.as<int32_t>();
            for (int r = 0; r < n; ++r) w8[r] = uint8_t(w32[r]);
            out->warn = w8;
        }
        std::memcpy(h_cal_off, B.h_cal_off_z.p, 4 * size_t(n + 1));
        const int32_t n_cals = h_cal_off[n];
        g_n_cals = n_cals;
        g_cals_pool = fa.pool;
        g_cals_list = fa.list;
        g_scores = B.scores.as<double>();
        out->cals = nullptr;   // (in the pool: sk_enum_device_fetch_cals)
        out->scores = nullptr; // (on the device: sk_enum_device_fetch_scores -- the host needs them only for a read stage 3 turned down)
        out->stage3 = B.h_s3_out.as<sk3::Out>();
        out->ref_reads_outside = h_dyn[DYN_REF_OUTSIDE];
        g_last.fused = true;
        g_last.fs = fs;
        g_last.fs.f.n_cals = n_cals;
        g_last.fs.f.ref_outside = nullptr; // (a repeat for the clock counts nothing)
        g_last.n = n;
        g_last.n_cals = n_cals;
        g_last.scores = B.scores.as<double>();
        g_last.cells = 0;
        for (int r = 0; r < n; ++r) g_last.cells += (in->read_off[r + 1] - in->read_off[r]) * int64_t(h_cal_off[r + 1] - h_cal_off[r]);
        g_last.valid = true;
        if (timing) std::fprintf(stderr, "[enum-dev] one sequence: %d reads, %d levels launched, %d leaves, %d candidate alignments, stage 3 lists %d + %d\n", n, n_levels,
                                 h_counters[Caps::K + 3], n_cals, h_dyn[DYN_N_LIGHT], h_dyn[DYN_N_HEAVY]);
        return 0;
    }
    ++g_jobs_staged;
    {
        PFrame* buf[2] = { B.level_a.as<PFrame>(), B.level_b.as<PFrame>() };
        ea.level_in = nullptr;
        ea.level_out = buf[0];
        ea.depth = -1;
        SK_LAUNCH(root_kernel, dim3((n + 63) / 64), dim3(64), 0, st, ea);
        const size_t lds = 64 * sizeof(PFrame);
        const int blocks = int(std::min<int64_t>((frame_cap + 63) / 64, 1024));
        // a call at depth d expands indel order[d]; a call at depth n_order is a leaf.  A read's order as it arrives says how deep its
        // search goes unless an alignment reaches indels beyond the read's first range (they join the order): the levels up to the
        // deepest order of the job (+1) go out at once, the level counts that come back with the results say whether more are needed
        int max_order = 0;
        for (int r = 0; r < n; ++r) max_order = std::max(max_order, int(in->reads[r].n_order));
        int done = 0, batch = std::min(std::max(max_order + 2, 3), int(Caps::K) + 2);
        for (;;) {
            const int upto = std::min(done + batch, int(Caps::K) + 2);
            for (int d = done; d < upto; ++d) {
                ea.level_in = buf[d & 1];
                ea.level_out = buf[(d + 1) & 1];
                ea.depth = d;
                SK_LAUNCH(level_kernel, dim3(blocks), dim3(64), lds, st, ea);
            }
            done = upto;
            SK_HIP(skrt::getLastError());
            if (fetch_zero(B.counters, B.status, 4 * size_t(n))) return 1; // counters, warn, n_raw, status
            SK_HIP(skrt::streamSynchronize(st));
            if (done >= Caps::K + 2 || B.h_counters.as<int32_t>()[done] == 0) break;
            batch = 4;
        }
    }
    lap("E1 search (all depths)");
    int32_t* h_status = B.h_status.as<int32_t>();
    const int32_t* h_counters = B.h_counters.as<int32_t>();
    const int32_t n_leaves = std::min<int32_t>(h_counters[Caps::K + 3], int32_t(pool_cap));
    if (timing) {
        long long calls = 0;
        int deepest = 0;
        for (int d = 0; d < Caps::K + 3; ++d) {
            calls += h_counters[d];
            if (h_counters[d]) deepest = d;
        }
        std::fprintf(stderr, "[enum-dev] search calls %lld, deepest level %d, leaves found %d\n", calls, deepest, n_leaves);
    }
    {
        uint8_t* w8 = reinterpret_cast<uint8_t*>(B.h_warn.p); // (in place: byte r is written after int r was read)
        const int32_t* w32 = B.h_warn.as<int32_t>();
        for (int r = 0; r < n; ++r) w8[r] = uint8_t(w32[r]);
        out->warn = w8;
    }

    // ---- E2: leaves -> each read's set
    const int32_t* h_n_raw = B.h_n_raw.as<int32_t>();
    int32_t* h_raw_off = B.h_raw_off.as<int32_t>();
    h_raw_off[0] = 0;
    for (int r = 0; r < n; ++r) h_raw_off[r + 1] = h_raw_off[r] + ((h_status[r] == ST_OK) ? h_n_raw[r] : 0);
    const int32_t n_grouped = h_raw_off[n];
    RES(grouped, 4 * size_t(n_grouped));
    RES(ghash, 4 * size_t(n_grouped));
    RES(gkey, 16 * size_t(n_grouped));
    RES(dup, size_t(n_grouped));
    RES(sorted, 4 * size_t(n_grouped));
    SetArgs sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.pool = ea.pool;
    sa.leaf_read = ea.leaf_read;
    sa.leaf_hash = ea.leaf_hash;
    sa.status = ea.status;
    sa.n_reads = n;
    sa.n_leaves = n_leaves;
    sa.raw_off = B.raw_off.as<int32_t>();
    sa.fill = B.fill.as<int32_t>();
    sa.grouped = B.grouped.as<int32_t>();
    sa.ghash = B.ghash.as<uint32_t>();
    sa.gkey = B.gkey.as<ulonglong2>();
    sa.dup = B.dup.as<uint8_t>();
    sa.n_uniq = B.n_uniq.as<int32_t>();
    sa.cal_off = B.cal_off.as<int32_t>();
    sa.sorted = B.sorted.as<int32_t>();
    sa.n_leaves_dev = nullptr;
    SK_HIP(skrt::memcpyAsync(B.raw_off.p, h_raw_off, 4 * size_t(n + 1), hipMemcpyHostToDevice, st));
    if (n_grouped > 0) {
        SK_LAUNCH(group_kernel, dim3((n_leaves + 255) / 256), dim3(256), 0, st, sa);
        SK_LAUNCH(dedupe_kernel, dim3((n_grouped + 255) / 256), dim3(256), 0, st, sa);
        SK_HIP(skrt::getLastError());
    }
    if (fetch_zero(B.n_uniq, B.n_uniq, 4 * size_t(n))) return 1;
    SK_HIP(skrt::streamSynchronize(st));
    const int32_t* h_n_uniq = B.h_n_uniq.as<int32_t>();
    for (int r = 0; r < n; ++r) h_cal_off[r + 1] = h_cal_off[r] + ((h_status[r] == ST_OK) ? h_n_uniq[r] : 0);
    const int32_t n_cals = h_cal_off[n];
    SK_HIP(skrt::memcpyAsync(B.cal_off.p, h_cal_off, 4 * size_t(n + 1), hipMemcpyHostToDevice, st));
    if (n_grouped > 0) {
        SK_LAUNCH(rank_kernel, dim3((n_grouped + 255) / 256), dim3(256), 0, st, sa);
        SK_HIP(skrt::getLastError());
    }
    lap("E2 sets");

    // ---- L1, L2: layout of the batch
    RES(n_ops, 4 * size_t(n_cals));
    RES(op_off, 8 * size_t(n_cals + 1));
    HRES(h_n_ops, 4 * size_t(n_cals));
    HRES(h_op_off, 8 * size_t(n_cals + 1));
    FlatArgs fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.job = dj;
    fa.ins_pool = B.ins.as<char>();
    fa.ref = B.ref.as<char>();
    fa.ref_offset = in->ref_offset;
    fa.ref_len = in->ref_len;
    fa.n_reads = n;
    fa.n_cals = n_cals;
    fa.pool = ea.pool;
    fa.list = B.sorted.as<int32_t>();
    fa.status = ea.status;
    fa.read_off = B.read_off.as<int64_t>();
    fa.read_code = B.read_code.as<uint8_t>();
    fa.cal_off = B.cal_off.as<int32_t>();
    fa.win_begin = B.win_begin.as<int32_t>();
    fa.win_len = B.win_len.as<int32_t>();
    fa.hap_len = B.hap_len.as<int32_t>();
    fa.n_ins = B.n_ins.as<int32_t>();
    fa.ins_idx = B.ins_idx.as<int16_t>();
    fa.ins_off = B.ins_off.as<int32_t>();
    fa.n_ops = B.n_ops.as<int32_t>();
    fa.win_end = B.win_end.as<int32_t>();
    fa.ins_lo = B.ins_lo.as<int32_t>();
    fa.ins_hi = B.ins_hi.as<int32_t>();
    RES(n_seg8, size_t(n_cals));
    fa.n_seg8 = B.n_seg8.as<uint8_t>();
    fa.ref_outside = B.counters.as<int32_t>() + (Caps::K + 8 + DYN_REF_OUTSIDE);
    // (byte patterns: 0x7f7f7f7f is large enough to stand for "no lower bound yet", 0x80808080 is below any position / index)
    SK_HIP(skrt::memsetAsync(B.minmax_arena.p, 0x7f, minmax_p[2].off, st));                               // win_begin, ins_lo
    SK_HIP(skrt::memsetAsync(static_cast<char*>(B.minmax_arena.p) + minmax_p[2].off, 0x80, minmax_bytes - minmax_p[2].off, st)); // win_end, ins_hi
    if (n_cals > 0) SK_LAUNCH(pool_bounds_kernel, dim3((n_cals + 255) / 256), dim3(256), 0, st, fa);
    SK_LAUNCH(pool_layout_kernel, dim3((n + 63) / 64), dim3(64), 0, st, fa);
    SK_HIP(skrt::getLastError());
    // ---- flattening + scoring.  The default is F5 (flatten_score_kernel): one launch from the records to the scores, no layout pass
    // and no host wait before stage 3.  F1-F3 + A1c (the staged chain: ops, entries, column words through HBM) run when the job
    // holds a read F5 turns down, when the host wants the alignments without scores, and with $SK_A5_FUSED=0 (tests: both chains).
    const bool fused_enabled = !(std::getenv("SK_A5_FUSED") != nullptr && std::strcmp(std::getenv("SK_A5_FUSED"), "0") == 0);
    const bool try_fused = fused_enabled && in->want_scores && n_cals > 0;
    RES(scores, 8 * size_t(n_cals));
    if (!(in->want_scores && in->want_stage3)) HRES(h_cals, sizeof(PCal) * size_t(n_cals));
    HRES(h_scores, 8 * size_t(n_cals));
    out->cals = B.h_cals.as<PCal>();
    g_n_cals = n_cals;
    fa.max_read_len = in->max_read_len;
    g_cals_pool = fa.pool;
    g_cals_list = fa.list;
    bool cals_in_set_order = false; // F5 reads the records where the search left them; the staged chain copies them (flatten_kernel)

    // what follows the scores in either chain: bookkeeping for sk_enum_device_rescore, the scores' way back, stage 3
    auto after_scores = [&]() -> int {
        D2H(h_scores, scores, 8 * size_t(n_cals));
        out->scores = B.h_scores.as<double>();
        if (in->want_stage3) {
            RES(s3_order, 4 * size_t(n_cals));
            RES(s3_smooth, 8 * size_t(n_cals));
            RES(s3_flag, size_t(n_cals));
            RES(s3_rm_type, size_t(n_bases));
            RES(s3_rm_pos, 4 * size_t(n_bases));
            RES(s3_key, 16 * size_t(n_cals));
            RES(s3_sorted_score, 8 * size_t(n_cals));
            RES(s3_sorted_hash, 4 * size_t(n_cals));
            RES(s3_next_same, 4 * size_t(n_cals));
            RES(s3_removed, size_t(n_cals));
            RES(s3_range_end, 4 * size_t(n_cals));
            RES(s3_out, sizeof(sk3::Out) * size_t(n));
            HRES(h_s3_out, sizeof(sk3::Out) * size_t(n));
            Stage3Args s3;
            s3.tab.tab = dj.tab;
            s3.tab.r2i = B.r2i.as<double>();
            s3.tab.i2r = B.i2r.as<double>();
            s3.tab.orig = B.orig.as<int32_t>();
            s3.tab.n_tab = in->n_tab;
            s3.tab.max_indel_size = in->max_indel_size;
            s3.tab.consulted = dj.consulted;
            s3.opt = in->stage3_opt;
            s3.n_reads = n;
            s3.status = ea.status;
            s3.cal_off = fa.cal_off;
            s3.cals = cals_in_set_order ? fa.cals : fa.pool;
        s3.slot = cals_in_set_order ? nullptr : fa.list;
            s3.scores = B.scores.as<double>();
            s3.read_off = fa.read_off;
            s3.read_code = fa.read_code;
            s3.map_level = B.map_level.as<int32_t>();
            s3.order = B.s3_order.as<int32_t>();
            s3.smooth = B.s3_smooth.as<double>();
            s3.flag = B.s3_flag.as<uint8_t>();
            s3.rm_type = B.s3_rm_type.as<uint8_t>();
            s3.rm_pos = B.s3_rm_pos.as<int32_t>();
            s3.key = B.s3_key.as<uint32_t>();
            s3.sorted_score = B.s3_sorted_score.as<double>();
            s3.sorted_hash = B.s3_sorted_hash.as<uint32_t>();
            s3.next_same = B.s3_next_same.as<int32_t>();
            s3.removed = B.s3_removed.as<uint8_t>();
            s3.range_end = B.s3_range_end.as<int32_t>();
            s3.out = B.s3_out.as<sk3::Out>();
            s3.n_list = nullptr;
            s3.list_step = 1;
            // (F5 counts the reads it turns down: stage 3 then has nothing to work on -- the staged chain scores the job and calls this again)
            s3.skip_if = cals_in_set_order ? nullptr : (B.counters.as<int32_t>() + (Caps::K + 7));
            // two launches: reads with few candidate alignments (small LDS, many wavefronts per CU) and the others
            RES(s3_list, 4 * size_t(n));
            HRES(h_s3_list, 4 * size_t(n));
            int32_t* h_list = B.h_s3_list.as<int32_t>();
            int light_cals = S3_LIGHT_CALS, lds_cals = S3_LDS_CALS;
            if (const char* e = std::getenv("SK_STAGE3_TEST_LDS_CALS")) { // tests: small capacities, so that the second launch and the HBM arrays run
                int a = 0, b = 0;
                if (std::sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b >= a && b <= S3_LDS_CALS) {
                    light_cals = a;
                    lds_cals = b;
                }
            }
            int n_light = 0, n_heavy = 0, max_cals = 0;
            for (int r = 0; r < n; ++r) {
                const int k = h_cal_off[r + 1] - h_cal_off[r];
                if (k <= light_cals) h_list[n_light++] = r;
                else {
                    h_list[n - 1 - n_heavy++] = r;
                    max_cals = std::max(max_cals, k);
                }
            }
            H2D(s3_list, h_list, 4 * size_t(n));
            auto lds_bytes = [](const int cals) { return size_t(cals) * (8 + 8 + 4 + 4 + 4 + 4 + 1 + 1) + 8; };
            if (n_light > 0) {
                s3.list = B.s3_list.as<int32_t>();
                s3.lds_cals = light_cals;
                SK_LAUNCH(stage3_kernel<1>, dim3(n_light), dim3(64), lds_bytes(light_cals), st, s3);
            }
            if (n_heavy > 0) {
                s3.list = B.s3_list.as<int32_t>() + (n - n_heavy);
                s3.lds_cals = std::min(max_cals, lds_cals); // (a read with more uses the arrays in HBM)
                // four wavefronts per read: the loops over the read's alignments go four times as wide, lane 0's share stays
                static const bool one_wave = std::getenv("SK_STAGE3_ONE_WAVE") != nullptr; // (diagnostics)
                if (one_wave) SK_LAUNCH(stage3_kernel<1>, dim3(n_heavy), dim3(64), lds_bytes(s3.lds_cals), st, s3);
                else SK_LAUNCH(stage3_kernel<4>, dim3(n_heavy), dim3(256), lds_bytes(s3.lds_cals), st, s3);
            }
            if (timing) std::fprintf(stderr, "[enum-dev] stage 3: %d reads with at most %d candidate alignments, %d with more (up to %d)\n", n_light, light_cals, n_heavy, max_cals);
            SK_HIP(skrt::getLastError());
            D2H(h_s3_out, s3_out, sizeof(sk3::Out) * size_t(n));
            out->stage3 = B.h_s3_out.as<sk3::Out>();
            lap("S3 stage 3");
        }
        return 0;
    };

    bool staged = !try_fused;
    if (try_fused) {
        FusedScoreArgs fs;
        fs.f = fa;
        fs.read_qual = B.read_qual.as<uint8_t>();
        fs.tab = ctx.dev_tables;
        fs.scores = B.scores.as<double>();
        fs.err = ctx.dev_error_flags;
        fs.n_unhandled = B.counters.as<int32_t>() + (Caps::K + 7);
        fs.queue = B.counters.as<unsigned>() + (Caps::K + 8 + DYN_COUNT);
        fs.write_cals = 0;
        fs.dbg = nullptr;
        launch_flatten_score(n, st, fs);
        SK_HIP(skrt::getLastError());
        lap("F5 flatten + score");
        if (in->want_stage3) {
            out->cals = nullptr; // (they stay here, in the pool: sk_enum_device_fetch_cals gathers them when the host asks)
        } else {
            RES(cals, sizeof(PCal) * size_t(n_cals));
            SK_LAUNCH(gather_cals_kernel, dim3((n_cals + 63) / 64), dim3(64), 0, st, fa.pool, fa.list, n_cals, B.cals.as<PCal>());
            SK_HIP(skrt::getLastError());
            g_cals_gathered = true;
            D2H(h_cals, cals, sizeof(PCal) * size_t(n_cals));
        }
        g_last.fused = true;
        g_last.fs = fs;
        g_last.fs.f.ref_outside = nullptr; // (a repeat for the clock counts nothing)
        g_last.n = n;
        g_last.n_cals = n_cals;
        g_last.scores = B.scores.as<double>();
        g_last.cells = 0;
        for (int r = 0; r < n; ++r) g_last.cells += (in->read_off[r + 1] - in->read_off[r]) * int64_t(h_cal_off[r + 1] - h_cal_off[r]);
        g_last.valid = true;
        if (after_scores()) return 1;
        if (fetch_zero(B.counters, B.counters, 4 * size_t(n_counters))) return 1; // (n_unhandled)
        if (fetch_zero(B.status, B.status, 4 * size_t(n))) return 1;               // (F5 may have turned reads down: ST_FAIL)
        SK_HIP(skrt::streamSynchronize(st));
        if (B.h_counters.as<int32_t>()[Caps::K + 7] > 0) staged = true; // a read outside F5's form: the staged chain over the job
    }
    if (staged) {
        RES(cals, sizeof(PCal) * size_t(n_cals));
        fa.cals = B.cals.as<PCal>();
        cals_in_set_order = true;
        g_cals_gathered = true; // (flatten_kernel below writes them)
        if (n_cals > 0) {
            SK_LAUNCH(op_count_kernel, dim3((n_cals + 63) / 64), dim3(64), 0, st, fa);
            SK_HIP(skrt::getLastError());
        }
        if (fetch_zero(B.status, B.hap_len, 4 * size_t(n))) return 1; // status, hap_len (the host has made bytes of its copy of warn)
        D2H(h_n_ops, n_ops, 4 * size_t(n_cals));
        SK_HIP(skrt::streamSynchronize(st));
        lap("L1+L2 layout");

        const int32_t* h_hap_len = B.h_hap_len.as<int32_t>();
        const int32_t* h_n_ops = B.h_n_ops.as<int32_t>();
        int64_t* h_hap_off = B.h_hap_off.as<int64_t>();
        int64_t* h_op_off = B.h_op_off.as<int64_t>();
        h_hap_off[0] = 0;
        h_op_off[0] = 0;
        int32_t max_hap = 1;
        for (int r = 0; r < n; ++r) {
            const int32_t c0 = h_cal_off[r], c1 = h_cal_off[r + 1];
            h_hap_off[r + 1] = h_hap_off[r] + ((c1 > c0) ? h_hap_len[r] : 0);
            if (c1 > c0) max_hap = std::max(max_hap, h_hap_len[r]);
            const bool ok = (h_status[r] == ST_OK); // (a read turned down by L1/L2 keeps its slots in the batch, with no ops)
            for (int32_t c = c0; c < c1; ++c) h_op_off[c + 1] = h_op_off[c] + (ok ? h_n_ops[c] : 0);
        }
        const int64_t n_hap = h_hap_off[n], ops_total = h_op_off[n_cals];

        RES(hap_code, size_t(n_hap) + 16);
        RES(ops, sizeof(sk_score_op) * size_t(ops_total));
        RES(entries, 4 * (size_t(ops_total) + 2 * size_t(n_cals) + 1));
        HRES(h_colmat_off, 8 * size_t(n + 1));
        int64_t* h_colmat_off = B.h_colmat_off.as<int64_t>();
        h_colmat_off[0] = 0;
        for (int r = 0; r < n; ++r)
            h_colmat_off[r + 1] = h_colmat_off[r] + ((in->read_off[r + 1] - in->read_off[r] + 7) / 8) * int64_t(h_cal_off[r + 1] - h_cal_off[r]);
        const int64_t colmat_words = h_colmat_off[n];
        RES(colmat, 4 * size_t(colmat_words) + 16);
        RES(colmat_off, 8 * size_t(n + 1));

        if (n_cals > 0) {
            SK_HIP(skrt::memcpyAsync(B.hap_off.p, h_hap_off, 8 * size_t(n + 1), hipMemcpyHostToDevice, st));
            SK_HIP(skrt::memcpyAsync(B.op_off.p, h_op_off, 8 * size_t(n_cals + 1), hipMemcpyHostToDevice, st));
            SK_HIP(skrt::memsetAsync(B.mask_arena.p, 0, mask_bytes, st)); // evmask, addmask
            SK_HIP(skrt::memsetAsync(B.colmat.p, SK_SEL_NONE | (SK_SEL_NONE << 4), 4 * size_t(colmat_words) + 16, st));
            SK_HIP(skrt::memcpyAsync(B.colmat_off.p, h_colmat_off, 8 * size_t(n + 1), hipMemcpyHostToDevice, st));
            fa.hap_off = B.hap_off.as<int64_t>();
            fa.op_off = B.op_off.as<int64_t>();
            fa.hap_code = B.hap_code.as<uint8_t>();
            fa.ops = B.ops.as<sk_score_op>();
            fa.entries = B.entries.as<uint32_t>();
            fa.evmask = B.evmask.as<uint32_t>();
            fa.evmask_words = W;
            fa.colmat = B.colmat.as<uint8_t>();
            fa.colmat_off = B.colmat_off.as<int64_t>();
            fa.addmask = B.addmask.as<uint32_t>();
            SK_LAUNCH(pool_fill_kernel, dim3(n), dim3(64), 0, st, fa);
            SK_LAUNCH(flatten_kernel, dim3((n_cals + 63) / 64), dim3(64), 0, st, fa);
            SK_HIP(skrt::getLastError());
            if (in->want_scores && in->want_stage3) out->cals = nullptr; // (they stay here: sk_enum_device_fetch_cals)
            else D2H(h_cals, cals, sizeof(PCal) * size_t(n_cals));
            if (in->want_scores) {
                // F3: a wave per read ($SK_F3_KERNEL = thread pins the thread-per-alignment form; the tests run both)
                const bool f3_thread = (std::getenv("SK_F3_KERNEL") != nullptr && std::strcmp(std::getenv("SK_F3_KERNEL"), "thread") == 0);
                if (f3_thread) SK_LAUNCH(entries_kernel, dim3((n_cals + 63) / 64), dim3(64), 0, st, fa);
                else SK_LAUNCH(entries_wave_kernel, dim3(n), dim3(64), 0, st, fa);
                SK_HIP(skrt::getLastError());
                lap("F1-F3 flatten");
                sk_align_batch d;
                std::memset(&d, 0, sizeof(d));
                d.n_reads = n;
                d.n_cals = n_cals;
                d.n_ops = ops_total;
                d.read_off = B.read_off.as<int64_t>();
                d.read_code = B.read_code.as<uint8_t>();
                d.read_qual = B.read_qual.as<uint8_t>();
                d.hap_off = fa.hap_off;
                d.hap_code = fa.hap_code;
                d.cal_off = fa.cal_off;
                d.op_off = fa.op_off;
                d.ops = fa.ops;
                d.max_read_len = in->max_read_len;
                d.max_hap_len = max_hap;
                d.entries = fa.entries;
                d.evmask = fa.evmask;
                d.evmask_words = W;
                d.colmat = B.colmat.as<uint32_t>();
                d.colmat_off = fa.colmat_off;
                d.addmask = fa.addmask;
                if (sk_score_alignments_launch_hostleg(&d, B.scores.as<double>(), st)) return 1;
                lap("A1 score");
                g_last.fused = false;
                g_last.fa = fa;
                g_last.fa.ref_outside = nullptr;
                g_last.d = d;
                g_last.n = n;
                g_last.n_cals = n_cals;
                g_last.mask_bytes = mask_bytes;
                g_last.colmat_bytes = 4 * size_t(colmat_words) + 16;
                g_last.scores = B.scores.as<double>();
                g_last.cells = 0;
                for (int r = 0; r < n; ++r) g_last.cells += (in->read_off[r + 1] - in->read_off[r]) * int64_t(h_cal_off[r + 1] - h_cal_off[r]);
                g_last.valid = true;
                if (after_scores()) return 1;
            }
        }
    }
    if (fetch_zero(B.consulted, B.consulted, size_t(in->n_tab))) return 1;
    if (fetch_zero(B.counters, B.counters, 4 * size_t(n_counters))) return 1;
    SK_HIP(skrt::streamSynchronize(st));
    // (when F5 turned a read down both chains filled the pools: the count may be double -- what matters to the caller is zero or not)
    out->ref_reads_outside = B.h_counters.as<int32_t>()[Caps::K + 8 + DYN_REF_OUTSIDE];
    lap("done");
#undef RES
#undef HRES
#undef H2D
#undef D2H
    return 0;
}

// A job with the whole read path on the device (scores + stage 3) and F5 enabled runs as one fixed sequence with one wait; a job for
// which the sequence's assumptions do not hold (reported by the device with the results), one too large for capacity-sized buffers, and
// a caller that wants the alignments or the scores on the host run the staged way (a wait after the search, one after the sets, one at
// the end).  $SK_ENUM_ONE_WAIT = 0 pins the staged way (tests run both).
extern "C" int sk_enum_device_run(const SkEnumInput* in, SkEnumOutput* out)
{
    const bool enabled = !(std::getenv("SK_ENUM_ONE_WAIT") != nullptr && std::strcmp(std::getenv("SK_ENUM_ONE_WAIT"), "0") == 0);
    const bool fused_enabled = !(std::getenv("SK_A5_FUSED") != nullptr && std::strcmp(std::getenv("SK_A5_FUSED"), "0") == 0);
    bool redo = false;
    if (enabled && fused_enabled && in && in->want_scores && in->want_stage3 && in->n_reads > 0 && in->n_reads <= ONE_WAIT_MAX_READS &&
        in->max_read_len <= F5_MAX_READ) {
        const int rc = enum_device_run_impl(in, out, true, &redo);
        if (rc != 0 || !redo) return rc;
        ++g_jobs_one_wait_redone;
    }
    return enum_device_run_impl(in, out, false, &redo);
}

// Measurement entry (bench.py's a5 leg): flattening AND scoring of the candidate alignments the last run left on the device, the way
// that run did it -- F5 (one launch from the records to the scores), or, for a job scored by the staged chain, pool bytes, ops,
// transition entries, masks and column words rebuilt from the PCal records (F1-F3), then the scoring kernel over them --
// `reps` times, timed with events on the library's stream.  This is what sk_enum_device_run does between the sets (E2) and stage 3,
// minus the layout pass (L1/L2: per-read pool bounds and op counts, whose results -- offsets -- a steady state already has).
extern "C" int sk_enum_device_rescore(const int32_t reps, float* out_ms, int32_t* out_n_reads, int32_t* out_n_cals, int64_t* out_cells)
{
    SK_REQUIRE_INIT();
    if (!g_last.valid) return sk_fail("sk_enum_device_rescore: no device run with scores to repeat");
    if (reps <= 0 || !out_ms) return sk_fail("sk_enum_device_rescore: bad argument");
    if (skrt::remote()) return sk_fail("sk_enum_device_rescore: a timing entry point (events); not carried by the broker -- unset STRELKA_AMD_BROKER");
    SkContext& ctx = sk_ctx();
    SK_HIP(skrt::setDevice(ctx.device));
    hipStream_t st = ctx.stream;
    EnumBuffers& B = bufs();
    const FlatArgs& fa = g_last.fa;
    hipEvent_t e0, e1;
    SK_HIP(hipEventCreate(&e0));
    SK_HIP(hipEventCreate(&e1));
    SK_HIP(hipEventRecord(e0, st));
    if (g_last.fused && std::getenv("SK_F5_TIMING")) { // diagnostics: where a wave of F5 spends its cycles
        FusedScoreArgs fs = g_last.fs;
        const size_t nb = size_t(g_last.n) * 8;
        unsigned long long* d = nullptr;
        SK_HIP(skrt::malloc_(reinterpret_cast<void**>(&d), nb * 8));
        SK_HIP(skrt::memsetAsync(d, 0, nb * 8, st));
        fs.dbg = d;
        launch_flatten_score(g_last.n, st, fs);
        std::vector<unsigned long long> h(nb);
        SK_HIP(skrt::memcpyAsync(h.data(), d, nb * 8, hipMemcpyDeviceToHost, st));
        SK_HIP(skrt::streamSynchronize(st));
        (void)skrt::free_(d);
        double sum[8] = { 0 };
        unsigned long long first = ~0ull, last = 0;
        int nblk = 0;
        for (int r = 0; r < g_last.n; ++r) {
            const unsigned long long* s = &h[size_t(r) * 8];
            if (s[6] == 0) continue;
            ++nblk;
            first = std::min(first, s[0]);
            last = std::max(last, s[6]);
            sum[0] += double(s[1] - s[0]);
            for (int i = 2; i <= 5; ++i) sum[i] += double(s[i]);
            sum[7] += double(s[7]);
            sum[6] += double(s[6] - s[0]);
        }
        for (const int grid : { 64, 256, 512, 1024, 1792, 2048, 4096 }) { // how the kernel's time grows with the blocks in flight
            if (grid > g_last.n) break;
            hipEvent_t a0, a1;
            SK_HIP(hipEventCreate(&a0));
            SK_HIP(hipEventCreate(&a1));
            FusedScoreArgs f2 = g_last.fs;
            launch_flatten_score(grid, st, f2);
            SK_HIP(hipEventRecord(a0, st));
            for (int k = 0; k < 5; ++k) launch_flatten_score(grid, st, f2);
            SK_HIP(hipEventRecord(a1, st));
            SK_HIP(hipEventSynchronize(a1));
            float ms = 0;
            SK_HIP(hipEventElapsedTime(&ms, a0, a1));
            std::fprintf(stderr, "[f5-timing] grid %d blocks: %.1f us per launch\n", grid, ms * 1000.f / 5.f);
            (void)hipEventDestroy(a0);
            (void)hipEventDestroy(a1);
        }
        if (nblk)
            std::fprintf(stderr, "[f5-timing] blocks %d: cycles per block: prologue %.0f staging %.0f phaseA %.0f (walk %.0f) phaseB %.0f total %.0f = %.1f us by the 100 MHz counter (%.0f MHz); kernel span %llu cycles => %.1f blocks in flight\n",
                         nblk, sum[0] / nblk, sum[2] / nblk, sum[4] / nblk, sum[7] / nblk, sum[5] / nblk, sum[6] / nblk, sum[3] / nblk / 100.0,
                         sum[6] / std::max(1.0, sum[3]) * 100.0, last - first,
                         sum[6] / double(last - first));
    }
    for (int i = 0; i < reps; ++i) {
        if (g_last.fused) { // F5: the records -> the scores in one launch
            launch_flatten_score(g_last.n, st, g_last.fs);
            continue;
        }
        SK_HIP(skrt::memsetAsync(B.mask_arena.p, 0, g_last.mask_bytes, st));
        SK_HIP(skrt::memsetAsync(B.colmat.p, SK_SEL_NONE | (SK_SEL_NONE << 4), g_last.colmat_bytes, st));
        SK_LAUNCH(pool_fill_kernel, dim3(g_last.n), dim3(64), 0, st, fa);
        SK_LAUNCH(flatten_kernel, dim3((g_last.n_cals + 63) / 64), dim3(64), 0, st, fa);
        const bool f3_thread = (std::getenv("SK_F3_KERNEL") != nullptr && std::strcmp(std::getenv("SK_F3_KERNEL"), "thread") == 0);
        if (f3_thread) SK_LAUNCH(entries_kernel, dim3((g_last.n_cals + 63) / 64), dim3(64), 0, st, fa);
        else SK_LAUNCH(entries_wave_kernel, dim3(g_last.n), dim3(64), 0, st, fa);
        if (sk_score_alignments_launch_hostleg(&g_last.d, g_last.scores, st)) return 1;
    }
    SK_HIP(hipEventRecord(e1, st));
    SK_HIP(hipEventSynchronize(e1));
    SK_HIP(skrt::getLastError());
    SK_HIP(hipEventElapsedTime(out_ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (out_n_reads) *out_n_reads = g_last.n;
    if (out_n_cals) *out_n_cals = g_last.n_cals;
    if (out_cells) *out_cells = g_last.cells;
    return 0;
}
