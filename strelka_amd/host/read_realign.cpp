// read_realign.cpp -- host side of hot path A: everything realignAndScoreRead does around the scoring loop.
//
//   stage 1  gate, input normalisation, candidate-alignment enumeration   (L/starling_common/starling_read_align.cpp)
//   stage 3  alignment selection with exact tie rules, ambiguity clipping, per-indel read support
//            (starling_read_align.cpp:1536-1741, starling_read_align_clipper.cpp, starling_read_align_score_indels.cpp)
//   stage 2 (the likelihood of every candidate alignment) runs on the GPU: score_alignments.hip.
//
// All of this is integer / ordering logic whose results (CIGARs, which alignment wins a tie, which indels a read
// scores) must equal the reference's exactly.  Data model: the indels visible to a job live in ONE table sorted in
// IndelKey order (IndelKey.hh:57-80); everything else refers to indels by table index, so "IndelKey order" is integer
// order and the reference's std::set<IndelKey>/std::map<IndelKey,...> iteration orders are reproduced by sorted
// integer containers.  Each function cites the reference lines it reproduces.

#include "strelka_amd.h"
#include "../csrc/realign_core.h"
#include "../csrc/read_enumerate.h"
#include "../csrc/stage3_core.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <thread>
#include <stdexcept>
#include <string>
#include <system_error>
#include <tuple>
#include <vector>


namespace
{

struct Fail : std::runtime_error
{
    using std::runtime_error::runtime_error;
};

// ---------------------------------------------------------------------------------------------------- paths
// ALIGNPATH (L/blt_util/align_path.hh, align_path.cpp)

struct Seg
{
    uint32_t type = SK_SEG_NONE;
    uint32_t length = 0;
    bool operator==(const Seg& o) const { return type == o.type && length == o.length; }
    bool operator<(const Seg& o) const { return (type != o.type) ? (type < o.type) : (length < o.length); }
};
typedef std::vector<Seg> Path;

inline bool seg_align_match(uint32_t t) { return t == SK_SEG_MATCH || t == SK_SEG_SEQ_MATCH || t == SK_SEG_SEQ_MISMATCH; }
inline bool seg_read_len(uint32_t t) { return seg_align_match(t) || t == SK_SEG_INSERT || t == SK_SEG_SOFT_CLIP; }
inline bool seg_ref_len(uint32_t t) { return seg_align_match(t) || t == SK_SEG_DELETE || t == SK_SEG_SKIP; }
inline bool seg_indel(uint32_t t) { return t == SK_SEG_INSERT || t == SK_SEG_DELETE; }
inline bool seg_unaligned_edge(uint32_t t) { return t == SK_SEG_INSERT || t == SK_SEG_HARD_CLIP || t == SK_SEG_SOFT_CLIP; }

unsigned path_ref_length(const Path& p)
{
    unsigned v = 0;
    for (const Seg& s : p) if (seg_ref_len(s.type)) v += s.length;
    return v;
}
unsigned path_read_length(const Path& p)
{
    unsigned v = 0;
    for (const Seg& s : p) if (seg_read_len(s.type)) v += s.length;
    return v;
}
unsigned unaligned_prefix(const Path& p) // unalignedPrefixSize, align_path.cpp:190-200
{
    unsigned v = 0;
    for (const Seg& s : p) {
        if (!seg_unaligned_edge(s.type)) return v;
        if (seg_read_len(s.type)) v += s.length;
    }
    return v;
}
unsigned unaligned_suffix(const Path& p) // :204-214
{
    unsigned v = 0;
    for (auto it = p.rbegin(); it != p.rend(); ++it) {
        if (!seg_unaligned_edge(it->type)) return v;
        if (seg_read_len(it->type)) v += it->length;
    }
    return v;
}
unsigned insert_lead(const Path& p) // apath_insert_lead_size :295-315
{
    unsigned v = 0;
    for (const Seg& s : p) {
        if (s.type == SK_SEG_HARD_CLIP || s.type == SK_SEG_SOFT_CLIP) continue;
        if (s.type == SK_SEG_INSERT) v += s.length;
        else break;
    }
    return v;
}
unsigned insert_trail(const Path& p) // :319-339
{
    unsigned v = 0;
    for (auto it = p.rbegin(); it != p.rend(); ++it) {
        if (it->type == SK_SEG_HARD_CLIP || it->type == SK_SEG_SOFT_CLIP) continue;
        if (it->type == SK_SEG_INSERT) v += it->length;
        else break;
    }
    return v;
}
std::pair<unsigned, unsigned> match_edge_segments(const Path& p) // get_match_edge_segments :735-752
{
    const unsigned n = unsigned(p.size());
    std::pair<unsigned, unsigned> r(n, n);
    bool first = false;
    for (unsigned i = 0; i < n; ++i)
        if (seg_align_match(p[i].type)) {
            if (!first) r.first = i;
            first = true;
            r.second = i;
        }
    return r;
}
bool path_is_clipped(const Path& p) // is_clipped :770-781
{
    const size_t n = p.size();
    if (n == 0) return false;
    if (p[0].type == SK_SEG_SOFT_CLIP || p[0].type == SK_SEG_HARD_CLIP) return true;
    if (n > 1 && (p[n - 1].type == SK_SEG_SOFT_CLIP || p[n - 1].type == SK_SEG_HARD_CLIP)) return true;
    return false;
}
bool path_is_soft_clipped(const Path& p)
{
    for (const Seg& s : p) if (s.type == SK_SEG_SOFT_CLIP) return true;
    return false;
}
bool is_edge_readref_len_segment(const Path& p) // :824-846
{
    const unsigned n = unsigned(p.size());
    if (n == 0) return false;
    const auto ends = match_edge_segments(p);
    for (unsigned i = 0; i < n; ++i) {
        const bool edge = (i < ends.first) || (i > ends.second);
        const uint32_t t = p[i].type;
        if (edge && (t == SK_SEG_INSERT || t == SK_SEG_DELETE || t == SK_SEG_SKIP || t == SK_SEG_SOFT_CLIP)) return true;
    }
    return false;
}
void clip_clipper(Path& p, unsigned& hc_lead, unsigned& hc_trail, unsigned& sc_lead, unsigned& sc_trail) // :460-504
{
    hc_lead = hc_trail = sc_lead = sc_trail = 0;
    bool lead = true;
    Path q;
    for (const Seg& s : p) {
        if (s.type == SK_SEG_HARD_CLIP) (lead ? hc_lead : hc_trail) += s.length;
        else if (s.type == SK_SEG_SOFT_CLIP) (lead ? sc_lead : sc_trail) += s.length;
        else {
            lead = false;
            q.push_back(s);
        }
    }
    p.swap(q);
}
void clip_adder(Path& p, unsigned hc_lead, unsigned hc_trail, unsigned sc_lead, unsigned sc_trail) // :508-542
{
    Path q;
    if (hc_lead) q.push_back(Seg{ SK_SEG_HARD_CLIP, hc_lead });
    if (sc_lead) q.push_back(Seg{ SK_SEG_SOFT_CLIP, sc_lead });
    q.insert(q.end(), p.begin(), p.end());
    if (sc_trail) q.push_back(Seg{ SK_SEG_SOFT_CLIP, sc_trail });
    if (hc_trail) q.push_back(Seg{ SK_SEG_HARD_CLIP, hc_trail });
    p.swap(q);
}
bool segment_swap_start(const Path& p, unsigned i) // is_segment_swap_start :868-895
{
    bool ins = false, del = false;
    for (; i < p.size(); ++i) {
        if (p[i].type == SK_SEG_INSERT) ins = true;
        else if (p[i].type == SK_SEG_DELETE) del = true;
        else break;
    }
    return ins && del;
}
struct SwapInfo // align_path_util.hh:75-106
{
    unsigned n_seg, insert_length = 0, delete_length = 0;
    SwapInfo(const Path& p, unsigned i) : n_seg(i)
    {
        for (; n_seg < p.size() && seg_indel(p[n_seg].type); ++n_seg) {
            if (p[n_seg].type == SK_SEG_INSERT) insert_length += p[n_seg].length;
            else delete_length += p[n_seg].length;
        }
        n_seg -= i;
    }
};

// ---------------------------------------------------------------------------------------------------- ranges

struct Range // known_pos_range / pos_range (L/blt_util/pos_range.hh) -- open ends only arise for breakpoints
{
    bool has_b = true, has_e = true;
    int32_t b = 0, e = 0;
    Range() {}
    Range(int32_t bb, int32_t ee) : b(bb), e(ee) {}
    bool pos_intersect(int32_t p) const { return (!has_b || p >= b) && (!has_e || p < e); }
    bool range_intersect(const Range& o) const
    {
        return (!o.has_e || !has_b || o.e > b) && (!o.has_b || !has_e || o.b < e);
    }
    bool superset_of(const Range& o) const
    {
        return (!has_e || (o.has_e && o.e <= e)) && (!has_b || (o.has_b && o.b >= b));
    }
};

// ---------------------------------------------------------------------------------------------------- alignments

struct Aln // alignment, L/starling_common/alignment.hh:38-101
{
    Path path;
    int32_t pos = 0;
    bool fwd = true;
    bool empty() const { return path.empty(); }
    bool operator==(const Aln& o) const { return pos == o.pos && path == o.path && fwd == o.fwd; }
    bool operator<(const Aln& o) const
    {
        if (pos != o.pos) return pos < o.pos;
        if (fwd != o.fwd) return fwd < o.fwd;
        if (path.size() != o.path.size()) return path.size() < o.path.size();
        for (size_t i = 0; i < path.size(); ++i) {
            if (path[i] < o.path[i]) return true;
            if (!(path[i] == o.path[i])) return false;
        }
        return false;
    }
};

Range strict_range(const Aln& a) { return Range(a.pos, a.pos + int32_t(path_ref_length(a.path))); } // alignment_util.cpp:35-41
Range soft_clip_range(const Aln& a)                                                                 // :45-56
{
    return Range(a.pos - int32_t(insert_lead(a.path)), a.pos + int32_t(path_ref_length(a.path)) + int32_t(insert_trail(a.path)));
}
Range alignment_zone(const Aln& a, unsigned seq_length) // get_alignment_range + get_alignment_zone :60-88
{
    const int32_t b = a.pos - int32_t(unaligned_prefix(a.path));
    const int32_t e = a.pos + int32_t(path_ref_length(a.path)) + int32_t(unaligned_suffix(a.path));
    Range r;
    r.b = std::max(0, std::min(b, e - int32_t(seq_length)));
    r.e = std::max(e, b + int32_t(seq_length));
    return r;
}
bool is_overmax(const Aln& a, unsigned max_indel_size) // alignment.cpp:32-50
{
    const size_t n = a.path.size();
    for (size_t i = 0; i < n; ++i) {
        if (i == 0 || i + 1 == n) continue;
        if (seg_indel(a.path[i].type) && a.path[i].length > max_indel_size) return true;
    }
    return false;
}
Aln remove_edge_deletions(const Aln& al, bool lead, bool trail) // alignment_util.cpp:92-124
{
    Aln r;
    r.fwd = al.fwd;
    r.pos = al.pos;
    const auto ends = match_edge_segments(al.path);
    for (unsigned i = 0; i < al.path.size(); ++i) {
        const Seg& s = al.path[i];
        const bool le = i < ends.first, te = i > ends.second;
        if (s.type == SK_SEG_DELETE && ((le && lead) || (te && trail))) {
            if (le) r.pos += int32_t(s.length);
        } else {
            r.path.push_back(s);
        }
    }
    return r;
}
Aln matchify_edge_segment_type(const Aln& al, uint32_t seg_type, bool lead = true, bool trail = true) // :130-174
{
    Aln r;
    r.fwd = al.fwd;
    r.pos = al.pos;
    const auto ends = match_edge_segments(al.path);
    for (unsigned i = 0; i < al.path.size(); ++i) {
        const Seg& s = al.path[i];
        const bool le = i < ends.first, te = i > ends.second;
        const bool target = (s.type == seg_type) && ((lead && le) || (trail && te));
        if (target && le) r.pos -= int32_t(s.length);
        if (target || seg_align_match(s.type)) {
            if (!r.path.empty() && seg_align_match(r.path.back().type)) r.path.back().length += s.length;
            else r.path.push_back(Seg{ SK_SEG_MATCH, s.length });
        } else {
            r.path.push_back(s);
        }
    }
    return r;
}
Aln matchify_edge_indels(const Aln& al, bool lead, bool trail) // :190-198
{
    return matchify_edge_segment_type(remove_edge_deletions(al, lead, trail), SK_SEG_INSERT, lead, trail);
}
int32_t translate_ref_offset_to_read_offset(int32_t target, const Path& p) // :226-263
{
    if (target < 0) return -1;
    int32_t ref_off = 0, read_off = 0;
    for (const Seg& s : p) {
        if (seg_read_len(s.type)) read_off += int32_t(s.length);
        if (!seg_ref_len(s.type)) continue;
        ref_off += int32_t(s.length);
        if (ref_off <= target) continue;
        if (!seg_read_len(s.type)) return -1;
        return read_off - (ref_off - target);
    }
    return -1;
}
int32_t lowest_fwd_read_pos_for_ref_range(const Aln& al, const Range& rr) // getLowestFwdReadPosForRefRange :267-300
{
    int32_t ref_off = (al.fwd ? rr.b : rr.e - 1) - al.pos;
    const int32_t ro = translate_ref_offset_to_read_offset(ref_off, al.path);
    if (ro < 0) return ro;
    if (al.fwd) return ro;
    return int32_t(path_read_length(al.path)) - (ro + 1);
}

// ---------------------------------------------------------------------------------------------------- indel keys

struct Key // IndelKey, L/starling_common/IndelKey.hh:39-199
{
    int32_t pos = 0;
    int type = SK_INDEL_NONE;
    uint32_t del = 0;
    std::string ins;
    uint32_t ins_len() const { return uint32_t(ins.size()); }
    int32_t right_pos() const { return pos + int32_t(del); }
    bool is_mismatch() const { return type == SK_INDEL_MISMATCH; }
    bool is_breakpoint() const { return type == SK_INDEL_BP_LEFT || type == SK_INDEL_BP_RIGHT; }
    bool primitive_del() const { return type == SK_INDEL_INDEL && ins.empty() && del > 0; }
    bool primitive_ins() const { return type == SK_INDEL_INDEL && !ins.empty() && del == 0; }
    bool operator==(const Key& o) const { return pos == o.pos && type == o.type && del == o.del && ins == o.ins; }
    bool operator<(const Key& o) const // :57-80
    {
        if (pos != o.pos) return pos < o.pos;
        if (type != o.type) return type < o.type;
        if (type == SK_INDEL_NONE || type == SK_INDEL_BP_LEFT || type == SK_INDEL_BP_RIGHT) return false;
        if (ins.size() != o.ins.size()) return ins.size() < o.ins.size();
        if (del != o.del) return del < o.del;
        return ins < o.ins;
    }
    Range open_pos_range() const // :119-135
    {
        Range r(pos, right_pos());
        if (type == SK_INDEL_BP_LEFT) { r.has_e = false; r.e = 0; }
        else if (type == SK_INDEL_BP_RIGHT) { r.has_b = false; r.b = 0; r.e = pos; }
        return r;
    }
};

bool is_indel_conflict(const Key& a, const Key& b) // indel_util.cpp:25-42
{
    const bool mm = a.is_mismatch() || b.is_mismatch();
    Range r1 = a.open_pos_range(), r2 = b.open_pos_range();
    if (!mm) { r1.e++; r2.e++; }
    return r1.range_intersect(r2);
}
bool range_intersect_indel_breakpoints(const Range& pr, const Key& k) // :45-60
{
    if (k.is_mismatch()) return pr.pos_intersect(k.pos);
    if (pr.range_intersect(Range(k.pos, k.pos))) return true;
    const int32_t rp = k.right_pos();
    if (k.pos == rp) return false;
    return pr.range_intersect(Range(rp, rp));
}
bool range_adjacent_indel_breakpoints(const Range& pr, const Key& k) // :64-73
{
    if (pr.range_intersect(Range(k.pos - 1, k.pos + 1))) return true;
    const int32_t rp = k.right_pos();
    if (k.pos == rp) return false;
    return pr.range_intersect(Range(rp - 1, rp + 1));
}

struct Indel
{
    Key key;
    bool cand = false;
    double r2i = 0, i2r = 0;
    int32_t arid = -1;
    int8_t hap[SK_MAX_SAMPLES] = {};
    bool bypass[SK_MAX_SAMPLES] = {};
    bool forced = false, ndfr = false;
    int32_t orig = -1;
};

typedef std::vector<int> ISet; // sorted table indices == an indel_set_t

inline bool iset_has(const ISet& s, int i) { return std::binary_search(s.begin(), s.end(), i); }
inline void iset_insert(ISet& s, int i)
{
    auto it = std::lower_bound(s.begin(), s.end(), i);
    if (it == s.end() || *it != i) s.insert(it, i);
}

struct Cal // CandidateAlignment, L/starling_common/CandidateAlignment.hh:35-78
{
    Aln al;
    int lead = -1, trail = -1; // table index of the leading / trailing edge indel
    ISet indels;
    bool operator<(const Cal& o) const
    {
        if (al < o.al) return true;
        if (!(al == o.al)) return false;
        if (indels != o.indels) return std::lexicographical_compare(indels.begin(), indels.end(), o.indels.begin(), o.indels.end());
        if (lead != o.lead) return lead < o.lead;
        return trail < o.trail;
    }
};

} // namespace

// =====================================================================================================================

extern std::atomic<int64_t> g_sk_ref_reads_outside; // host/align_flatten.cpp

struct sk_realign_job
{
    sk_realign_options opt;
    std::string ref;
    int32_t ref_offset = 0;
    std::vector<Indel> tab;       // IndelKey order
    std::vector<int> orig_to_tab; // index as given -> table index
    // which table entries had their candidate status consulted -- the places below that read it are the places where the
    // reference calls IndelBuffer::isCandidateIndel, whose first call per indel computes and caches the status
    // (IndelBuffer.hh:153-164): a caller that evaluated the status of more indels than the reference would have, earlier
    // than it would have, must only commit the ones the reference would have touched (sk_realign_job_indels_consulted)
    mutable std::unique_ptr<std::atomic<uint8_t>[]> consulted;
    // reads whose search ran in the container-free core / on the device / in the container-based code although 1 or 2 was asked for
    mutable std::atomic<int64_t> n_core_reads{ 0 }, n_device_reads{ 0 }, n_fallback_reads{ 0 };
    // reads whose stage 3 ran in the container-free core (host, enumeration == 1) / on the device (== 2)
    mutable std::atomic<int64_t> n_stage3_core{ 0 }, n_stage3_device{ 0 };
    bool cand(const int i) const
    {
        consulted[size_t(i)].store(1, std::memory_order_relaxed);
        return tab[size_t(i)].cand;
    }
    std::vector<unsigned> max_toggle; // starling_align_limit
    bool has_mismatch_keys = false;   // any SK_INDEL_MISMATCH entry in the table (haplotyping's mismatch "indels")
    std::string error;

    struct Read
    {
        std::vector<uint8_t> code, qual; // (empty for a read that did not pass the gate)
        uint32_t read_len = 0;
        Aln input;
        int map_level = 0, sample = 0;
        int32_t read_id = 0;
        std::vector<Cal> cals; // std::set iteration order
        bool incomplete_search = false;
        bool warn_origin = false, warn_toggle = false;
        // results
        bool realigned = false;
        Aln realignment;
        double max_score = 0;
        std::vector<sk_path_seg> out_path;
        std::vector<sk_read_path_scores> scores;
        std::vector<int32_t> suboverlap;
        int32_t cal_begin = 0; // first candidate alignment of this read in the job's batch
        // enumeration == 2: the search state handed to the device (set until the job resolves it), what a host search needs
        // should the device turn the read down, and where the read's scores are once the device has scored it
        std::shared_ptr<skcore::PRead> pending;
        std::vector<int> observed;
        int32_t realign_b = 0, realign_e = 0;
        int64_t dev_score_at = -1;
        bool stage3_done = false; // stage 3 of this read ran on the device: finish_reads leaves its results alone
        // the read's candidate alignments as the device listed them (job.dev_cals), until something on the host needs `cals`
        int64_t dev_cal_at = -1;
        int32_t n_dev_cals = 0;
    };
    std::vector<double> dev_scores;
    uint64_t dev_generation = 0; // the device run that holds the candidate alignments of the reads finished there
    std::vector<Read> reads;
    sk_align_builder* builder = nullptr;
    int32_t n_cals_total = 0;
    bool finished = false;

    // ---- table access
    const Key& key(int i) const { return tab[i].key; }
    char ref_base(int32_t p) const // reference_contig_segment::get_base :46-51
    {
        if (p < ref_offset || p >= ref_offset + int32_t(ref.size())) {
            g_sk_ref_reads_outside.fetch_add(1, std::memory_order_relaxed);
            return 'N';
        }
        return ref[size_t(p - ref_offset)];
    }
    // IndelBuffer::rangeIterator, IndelBuffer.cpp:76-92
    std::pair<int, int> range_iter(int32_t begin_pos, int32_t end_pos) const
    {
        Key ek;
        ek.pos = end_pos;
        const int end = int(std::lower_bound(tab.begin(), tab.end(), ek, [](const Indel& a, const Key& k) { return a.key < k; }) - tab.begin());
        Key bk;
        bk.pos = begin_pos - int32_t(opt.max_indel_size);
        int begin = int(std::lower_bound(tab.begin(), tab.end(), bk, [](const Indel& a, const Key& k) { return a.key < k; }) - tab.begin());
        for (; begin < end; ++begin)
            if (tab[begin].key.right_pos() >= begin_pos) break;
        if (begin > end) begin = end;
        return std::make_pair(begin, end);
    }
    int find_key(const Key& k) const
    {
        auto it = std::lower_bound(tab.begin(), tab.end(), k, [](const Indel& a, const Key& kk) { return a.key < kk; });
        if (it == tab.end() || !(it->key == k)) return -1;
        return int(it - tab.begin());
    }
    unsigned get_max_toggle(size_t n) const { return n >= max_toggle.size() ? 1u : max_toggle[n]; } // starling_align_limit.hh:40-51
};

namespace
{

typedef sk_realign_job Job;

// ---------------------------------------------------------------------------------------------- starling_align_limit
// max_candidate_alignment_toggle, starling_align_limit.cpp:59-73 (float arithmetic as in the reference)
unsigned max_candidate_alignment_toggle(unsigned n_indel, unsigned max_alignments)
{
    const float mx(max_alignments);
    float sum(1.);
    for (unsigned i = 0; i < n_indel; ++i) {
        const unsigned k = i + 1;
        // binomial_coefficient<float>(n,k): exact integer value rounded to float
        unsigned long long c = 1;
        unsigned kk = (k > n_indel - k) ? n_indel - k : k;
        for (unsigned j = 1; j <= kk; ++j) c = c * (unsigned long long)(n_indel - kk + j) / (unsigned long long)j;
        sum += std::pow(static_cast<float>(2), static_cast<float>(k)) * static_cast<float>(c);
        if (sum > mx) return i;
    }
    return n_indel;
}
void build_align_limit(Job& j) // starling_align_limit ctor :77-88
{
    j.max_toggle.clear();
    for (unsigned i = 0; i < 100; ++i) {
        const unsigned mt = max_candidate_alignment_toggle(i, j.opt.max_realignment_candidates);
        if (i > 1 && mt < 2) break;
        j.max_toggle.push_back(mt);
    }
}

// --------------------------------------------------------------------------------------------- make_start_pos_alignment
// starling_read_align.cpp:394-584.  `indels`: table indices in key order.
Cal make_start_pos_alignment(const std::vector<Indel>& tab, int32_t ref_start_pos, int32_t read_start_pos, bool fwd,
                             unsigned read_length, const ISet& indels)
{
    const bool is_leading_read = (read_start_pos != 0);
    Cal cal;
    cal.al.pos = ref_start_pos;
    cal.al.fwd = fwd;
    int32_t ref_head = ref_start_pos, read_head = read_start_pos;
    Path& ap = cal.al.path;
    bool prev_mm = false;
    for (const int ii : indels) {
        const Key& k = tab[ii].key;
        const bool mm = k.is_mismatch();
        if (k.right_pos() < ref_start_pos) continue;
        if (k.right_pos() == ref_start_pos) {
            if (mm) continue;
            if (!is_leading_read) continue;
        }
        const bool first = ap.empty();
        if (is_leading_read && first) {
            if (k.pos != ref_start_pos) throw Fail("Anomalous condition for indel candidate");
            ap.push_back(Seg{ SK_SEG_INSERT, uint32_t(read_start_pos) });
            if (k.del > 0) {
                ap.push_back(Seg{ SK_SEG_DELETE, k.del });
                ref_head += int32_t(k.del);
            }
            cal.lead = ii;
            prev_mm = mm;
            continue;
        }
        const bool edge_delete = k.primitive_del() && (k.pos == ref_start_pos);
        const int match_size = k.pos - ref_head;
        const int min_match = (prev_mm || mm) ? 0 : 1;
        if (match_size < min_match && !edge_delete) throw Fail("Indel candidate is not greater than ref_head_pos");
        const unsigned match_segment = unsigned(match_size);
        if ((read_head + int32_t(match_segment)) > int32_t(read_length) ||
            ((read_head + int32_t(match_segment)) == int32_t(read_length) && !k.primitive_del()))
            break;
        if (match_segment > 0) {
            ap.push_back(Seg{ SK_SEG_MATCH, match_segment });
            ref_head += int32_t(match_segment);
            read_head += int32_t(match_segment);
        }
        if (mm) {
            ap.push_back(Seg{ SK_SEG_SEQ_MISMATCH, k.del });
            ref_head += int32_t(k.del);
            read_head += int32_t(k.del);
            if (read_head >= int32_t(read_length)) break;
        } else if (k.type == SK_INDEL_INDEL) {
            if (k.del > 0) {
                ap.push_back(Seg{ SK_SEG_DELETE, k.del });
                ref_head += int32_t(k.del);
            }
            if (k.ins_len() > 0) {
                const unsigned max_ins = read_length - unsigned(read_head);
                const unsigned ins = std::min(k.ins_len(), max_ins);
                ap.push_back(Seg{ SK_SEG_INSERT, ins });
                read_head += int32_t(ins);
                if (k.ins_len() >= max_ins) {
                    cal.trail = ii;
                    break;
                }
            } else {
                if (match_segment == 0) cal.lead = ii;
                else if (read_head == int32_t(read_length)) cal.trail = ii;
            }
        } else if (k.type == SK_INDEL_BP_LEFT) {
            const unsigned overhang = read_length - unsigned(read_head);
            ap.push_back(Seg{ SK_SEG_INSERT, overhang });
            read_head += int32_t(overhang);
            cal.trail = ii;
            break;
        } else {
            throw Fail("Unexpected indel state");
        }
        prev_mm = mm;
    }
    if (read_head < int32_t(read_length)) ap.push_back(Seg{ SK_SEG_MATCH, read_length - unsigned(read_head) });
    return cal;
}

// get_end_pin_start_pos, starling_read_align.cpp:594-719
void get_end_pin_start_pos(const std::vector<Indel>& tab, const ISet& indels, unsigned read_length, int32_t ref_end_pos,
                           int32_t read_end_pos, int32_t& ref_start_pos, int32_t& read_start_pos)
{
    ref_start_pos = ref_end_pos;
    read_start_pos = read_end_pos;
    const bool is_trailing_read = (read_end_pos != int32_t(read_length));
    bool is_first = true, prev_mm = false;
    for (auto it = indels.rbegin(); it != indels.rend(); ++it) {
        const Key& k = tab[*it].key;
        const bool mm = k.is_mismatch();
        if (k.pos > ref_end_pos) continue;
        if (k.pos == ref_end_pos) {
            if (mm) continue;
            if (!is_trailing_read) continue;
        }
        const bool trailing_indel = (!mm) && (k.right_pos() == ref_end_pos);
        if (trailing_indel) {
            if (k.type == SK_INDEL_INDEL) ref_start_pos -= int32_t(k.del);
        } else {
            if (is_first && read_end_pos != int32_t(read_length)) throw Fail("Unexpected realignment state");
            const int match_size = int(ref_start_pos - k.right_pos());
            const int min_match = (prev_mm || mm) ? 0 : 1;
            if (match_size < min_match) throw Fail("Unexpected indel position");
            const unsigned match_segment = unsigned(std::min(match_size, int(read_start_pos)));
            ref_start_pos -= int32_t(match_segment);
            read_start_pos -= int32_t(match_segment);
            if (read_start_pos == 0) return;
            if (k.type == SK_INDEL_INDEL) {
                ref_start_pos -= int32_t(k.del);
                if (k.ins_len() > 0) {
                    if (int32_t(k.ins_len()) >= read_start_pos) return;
                    read_start_pos -= int32_t(k.ins_len());
                }
            } else if (mm) {
                ref_start_pos -= int32_t(k.del);
                read_start_pos -= int32_t(k.del);
                if (read_start_pos == 0) return;
            } else if (k.type == SK_INDEL_BP_RIGHT) {
                return;
            } else {
                throw Fail("Unexpected indel state");
            }
        }
        is_first = false;
        prev_mm = mm;
    }
    ref_start_pos -= read_start_pos;
    read_start_pos = 0;
}

// ------------------------------------------------------------------------------------------------ enumeration state

// the subset of std::map the search uses, over a sorted vector: the reference passes its status / haplotype maps by value at
// every recursion step, and copying a node-based map costs an allocation per entry
template <typename K, typename V>
struct FlatMap
{
    typedef std::pair<K, V> value_type;
    typedef typename std::vector<value_type>::iterator iterator;
    typedef typename std::vector<value_type>::const_iterator const_iterator;
    std::vector<value_type> v;
    iterator begin() { return v.begin(); }
    iterator end() { return v.end(); }
    const_iterator begin() const { return v.begin(); }
    const_iterator end() const { return v.end(); }
    size_t size() const { return v.size(); }
    iterator lower(const K& k)
    {
        return std::lower_bound(v.begin(), v.end(), k, [](const value_type& a, const K& kk) { return a.first < kk; });
    }
    const_iterator lower(const K& k) const
    {
        return std::lower_bound(v.begin(), v.end(), k, [](const value_type& a, const K& kk) { return a.first < kk; });
    }
    iterator find(const K& k)
    {
        iterator it = lower(k);
        return (it != v.end() && it->first == k) ? it : v.end();
    }
    const_iterator find(const K& k) const
    {
        const_iterator it = lower(k);
        return (it != v.end() && it->first == k) ? it : v.end();
    }
    V& operator[](const K& k)
    {
        iterator it = lower(k);
        if (it == v.end() || !(it->first == k)) it = v.insert(it, value_type(k, V()));
        return it->second;
    }
    V& at(const K& k)
    {
        iterator it = find(k);
        if (it == v.end()) throw std::out_of_range("FlatMap::at");
        return it->second;
    }
    void insert(const value_type& kv)
    {
        iterator it = lower(kv.first);
        if (it == v.end() || !(it->first == kv.first)) v.insert(it, kv);
    }
};

struct IndelStatus // starling_align_indel_info :48-53
{
    bool is_present = false, is_remove_only = false, in_original = false;
};
typedef FlatMap<int, IndelStatus> StatusMap; // ordered by table index == IndelKey order

// getUpdatedSampleHaplotypeConstraints :63-130
int updated_haplotype_constraints(int hc, int cur_hap_id, bool cur_on, bool any_on)
{
    if (hc < 0) return hc;
    if (cur_hap_id < 0 && cur_on) return -1;
    if (cur_hap_id <= 0) return hc;
    const int from_cur = cur_on ? cur_hap_id : (3 - cur_hap_id);
    switch (from_cur) {
    case 0: return any_on ? -1 : 0;
    case 1:
    case 2:
        if (hc == 3 || hc == from_cur) return from_cur;
        return any_on ? -1 : 0;
    case 3: return (hc > 0) ? hc : -1;
    default: return -1;
    }
}
struct HapStatus // HaplotypeStatus :134-179
{
    int hc[SK_MAX_SAMPLES];
    int n;
    bool any_on = false;
    explicit HapStatus(int nn = 1) : n(nn) { for (int i = 0; i < SK_MAX_SAMPLES; ++i) hc[i] = 3; }
    bool update(const int* cur_hap_ids, bool cur_on)
    {
        any_on = any_on || cur_on;
        bool valid = false;
        for (int s = 0; s < n; ++s) {
            const int u = updated_haplotype_constraints(hc[s], cur_hap_ids[s], cur_on, any_on);
            if (u >= 0) valid = true;
            hc[s] = u;
        }
        return valid;
    }
};
typedef FlatMap<int32_t, HapStatus> HapMap;

struct SearchCtx
{
    const Job& job;
    const std::vector<int>& observed; // table indices this read was observed to support (ascending, no repeats)
    int sample;
    unsigned read_length;
    Range realign_range;
    std::set<Cal>& cal_set;
    bool& warn_origin;
    bool& warn_toggle;
};

bool is_usable_indel(const SearchCtx& c, int i) // :289-305
{
    return c.job.cand(i) || std::binary_search(c.observed.begin(), c.observed.end(), i);
}

// add_indels_in_range :311-366
void add_indels_in_range(const SearchCtx& c, const Range& pr, StatusMap& sm, std::vector<int>& order)
{
    const auto it = c.job.range_iter(pr.b, pr.e);
    for (int i = it.first; i < it.second; ++i) {
        const Key& k = c.job.key(i);
        if (!range_adjacent_indel_breakpoints(pr, k)) continue;
        const bool remove_only = !range_intersect_indel_breakpoints(pr, k);
        auto f = sm.find(i);
        if (f != sm.end()) {
            if (!remove_only && f->second.is_remove_only) f->second.is_remove_only = false;
        } else if (is_usable_indel(c, i)) {
            IndelStatus st;
            st.is_present = false;
            st.is_remove_only = remove_only;
            sm[i] = st;
            order.push_back(i);
        }
    }
}

// sort_remove_only_indels_last :724-751
void sort_remove_only_indels_last(const StatusMap& sm, std::vector<int>& order, unsigned current_depth = 0)
{
    std::vector<int> o2(order.begin(), order.begin() + current_depth);
    for (size_t i = current_depth; i < order.size(); ++i) {
        const IndelStatus& s = sm.find(order[i])->second;
        if (s.is_present || !s.is_remove_only) o2.push_back(order[i]);
    }
    for (size_t i = current_depth; i < order.size(); ++i) {
        const IndelStatus& s = sm.find(order[i])->second;
        if (!(s.is_present || !s.is_remove_only)) o2.push_back(order[i]);
    }
    order.swap(o2);
}

// addKeysToCandidateAlignment :789-808
void add_keys_to_cal(const Job& job, const StatusMap& sm, Cal& cal)
{
    const Range pr = strict_range(cal.al);
    ISet s;
    for (const auto& kv : sm) {
        if (!kv.second.is_present) continue;
        if (!range_intersect_indel_breakpoints(pr, job.key(kv.first))) continue;
        s.push_back(kv.first); // map iteration is ascending
    }
    if (cal.lead >= 0) iset_insert(s, cal.lead);
    if (cal.trail >= 0) iset_insert(s, cal.trail);
    cal.indels.swap(s);
}

// getCurIndelHaplotypeIds :811-852
void cur_indel_haplotype_ids(const Job& job, int cur_sample, int idx, bool in_original, int* ids)
{
    const Indel& d = job.tab[idx];
    const int n = job.opt.sample_count;
    for (int s = 0; s < n; ++s) ids[s] = 0;
    if (d.arid < 0) return;
    for (int s = 0; s < n; ++s) {
        int h = d.hap[s];
        if (h == 0) {
            bool valid = (!job.opt.is_haplotyping_enabled) || d.bypass[s] || d.forced;
            if (!valid && s == cur_sample && in_original) valid = true;
            if (d.key.is_mismatch() && s != cur_sample) valid = false;
            h = valid ? 0 : -1;
        }
        ids[s] = h;
    }
}

// candidate_alignment_search :857-1277 (status map, haplotype map and order are passed by value, as in the reference)
void candidate_alignment_search(SearchCtx& c, StatusMap sm, HapMap hm, std::vector<int> order, const unsigned depth,
                                const unsigned indel_toggle_depth, const unsigned total_toggle_depth, Range read_range,
                                int max_read_indel_toggle, const Cal& cal)
{
    const Job& job = c.job;
    bool is_new_indels = (indel_toggle_depth == 0);
    {
        const size_t start_size = sm.size();
        const Range pr = soft_clip_range(cal.al);
        if (!c.realign_range.superset_of(pr)) return;
        if (pr.b < read_range.b) {
            add_indels_in_range(c, Range(pr.b, read_range.b + 1), sm, order);
            read_range.b = pr.b;
        }
        if (pr.e > read_range.e) {
            add_indels_in_range(c, Range(read_range.e - 1, pr.e), sm, order);
            read_range.e = pr.e;
        }
        if (!is_new_indels) is_new_indels = (start_size != sm.size());
        if (is_new_indels) sort_remove_only_indels_last(sm, order, unsigned(start_size));
    }
    if (depth == order.size()) {
        Cal with_keys(cal);
        add_keys_to_cal(job, sm, with_keys);
        c.cal_set.insert(std::move(with_keys));
        return;
    }
    if (is_new_indels) {
        const double max_indels = c.read_length * job.opt.max_candidate_indel_density;
        if (sm.size() > max_indels) max_read_indel_toggle = 1;
        else max_read_indel_toggle = job.opt.max_read_indel_toggle;
        const int max_toggle = int(job.get_max_toggle(sm.size()));
        max_read_indel_toggle = std::min(max_read_indel_toggle, max_toggle);
    }
    if (int(indel_toggle_depth) > max_read_indel_toggle) {
        c.warn_toggle = true;
        return;
    }

    const int cur = order[depth];
    const Key& cur_key = job.key(cur);
    bool cur_conflicting = false, contains_ndfr = false;
    for (unsigned i = 0; i < depth; ++i) {
        const int oi = order[i];
        if (!sm[oi].is_present) continue;
        if (is_indel_conflict(job.key(oi), cur_key)) cur_conflicting = true;
        if (!contains_ndfr && job.tab[oi].ndfr) contains_ndfr = true;
    }
    const bool cur_on = sm[cur].is_present;
    const Indel& cur_data = job.tab[cur];
    const int32_t arid = cur_data.arid;
    const bool in_ar = (arid >= 0);
    if (in_ar && hm.find(arid) == hm.end()) hm.insert(std::make_pair(arid, HapStatus(job.opt.sample_count)));
    const bool cur_ndfr = cur_data.ndfr;
    int hap_ids[SK_MAX_SAMPLES];
    cur_indel_haplotype_ids(job, c.sample, cur, sm[cur].in_original, hap_ids);

    { // alignment 1: unchanged
        bool valid = true;
        HapMap nhm(hm);
        if (!cur_conflicting && in_ar) valid = nhm.at(arid).update(hap_ids, cur_on);
        else valid = (!cur_key.is_mismatch()) || (!cur_on);
        if (cur_on && contains_ndfr && cur_ndfr) valid = false;
        if (!valid && total_toggle_depth == 0) valid = true;
        if (valid)
            candidate_alignment_search(c, sm, nhm, order, depth + 1, indel_toggle_depth, total_toggle_depth, read_range,
                                       max_read_indel_toggle, cal);
    }

    bool valid = true;
    HapMap nhm(hm);
    if (!cur_conflicting && in_ar) valid = nhm.at(arid).update(hap_ids, !cur_on);
    else valid = !cur_key.is_mismatch() || cur_on;
    if (!cur_on && contains_ndfr && cur_ndfr) valid = false;
    if (!valid) return;
    if (!cur_on) {
        if (sm[cur].is_remove_only) return;
        if (cur_conflicting) return;
    }
    const unsigned toggle_inc = cur_key.is_mismatch() ? 0 : 1;
    if (int(indel_toggle_depth + toggle_inc) > max_read_indel_toggle) {
        c.warn_toggle = true;
        return;
    }
    sm[cur].is_present = !cur_on;
    ISet current;
    for (const auto& kv : sm)
        if (kv.second.is_present) current.push_back(kv.first);

    { // alignment 2: start pin
        const int32_t ref_start = cal.al.pos;
        bool start_pin_valid = true;
        if (!cur_key.is_mismatch()) {
            const bool del_span = cur_key.open_pos_range().pos_intersect(ref_start);
            const bool indel_span = cur_on && (cur == cal.lead);
            start_pin_valid = !(del_span || indel_span);
        }
        if (start_pin_valid) {
            const int32_t read_start = int32_t(unaligned_prefix(cal.al.path));
            const Cal start_cal = make_start_pos_alignment(job.tab, ref_start, read_start, cal.al.fwd, c.read_length, current);
            candidate_alignment_search(c, sm, nhm, order, depth + 1, indel_toggle_depth + toggle_inc, total_toggle_depth + 1,
                                       read_range, max_read_indel_toggle, start_cal);
        }
    }
    if (cur_key.is_mismatch()) return;
    if (cur_key.type == SK_INDEL_INDEL && cur_key.del == cur_key.ins_len()) return;
    { // alignment 3: end pin
        const int32_t ref_end = cal.al.pos + int32_t(path_ref_length(cal.al.path));
        const bool del_span = cur_key.open_pos_range().pos_intersect(ref_end - 1);
        const bool indel_span = cur_on && (cur == cal.trail);
        if (!(del_span || indel_span)) {
            const int32_t read_end = int32_t(c.read_length) - int32_t(unaligned_suffix(cal.al.path));
            int32_t ref_start = 0, read_start = 0;
            get_end_pin_start_pos(job.tab, current, c.read_length, ref_end, read_end, ref_start, read_start);
            if (ref_start < 0) {
                c.warn_origin = true;
            } else {
                const Cal start_cal = make_start_pos_alignment(job.tab, ref_start, read_start, cal.al.fwd, c.read_length, current);
                candidate_alignment_search(c, sm, nhm, order, depth + 1, indel_toggle_depth + toggle_inc, total_toggle_depth + 1,
                                           read_range, max_read_indel_toggle, start_cal);
            }
        }
    }
}

// bam_seq::get_char of a read code (L/htsapi/bam_seq.hh:30-46)
char code_char(uint8_t c)
{
    switch (c) {
    case SK_BAM_REF: return '=';
    case SK_BAM_A: return 'A';
    case SK_BAM_C: return 'C';
    case SK_BAM_G: return 'G';
    case SK_BAM_T: return 'T';
    default: return 'N';
    }
}
uint8_t char_code(char c)
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
std::string read_substr(const std::vector<uint8_t>& code, unsigned b, unsigned e)
{
    std::string s;
    for (unsigned i = b; i < e; ++i) s.push_back(i < code.size() ? code_char(code[i]) : 'N');
    return s;
}

// getCandidateAlignment :1479-1522: edge indel keys of the input alignment
void input_edge_keys(const Aln& al, const std::vector<uint8_t>& code, bool& has_lead, Key& lead, bool& has_trail, Key& trail)
{
    has_lead = has_trail = false;
    int32_t read_pos = 0, ref_pos = al.pos;
    const auto ends = match_edge_segments(al.path);
    for (unsigned i = 0; i < al.path.size(); ++i) {
        const Seg& s = al.path[i];
        if (s.type == SK_SEG_INSERT || s.type == SK_SEG_DELETE) {
            Key k;
            k.pos = ref_pos;
            k.type = SK_INDEL_INDEL;
            if (s.type == SK_SEG_INSERT) k.ins = read_substr(code, unsigned(read_pos), unsigned(read_pos) + s.length);
            else k.del = s.length;
            if (i < ends.first) { has_lead = true; lead = k; }
            else if (i > ends.second) { has_trail = true; trail = k; }
        }
        if (seg_read_len(s.type)) read_pos += int32_t(s.length);
        if (seg_ref_len(s.type)) ref_pos += int32_t(s.length);
    }
}

// getAlignmentIndels, L/starling_common/CandidateAlignment.cpp:59-177 (keys, which may be absent from the table)
void alignment_indel_keys(const Job& job, const Aln& al, const Key* lead, const Key* trail, const std::vector<uint8_t>& code,
                          bool include_mismatches, std::set<Key>& out)
{
    out.clear();
    const Path& p = al.path;
    unsigned pi = 0, read_off = 0;
    int32_t ref_head = al.pos;
    const auto ends = match_edge_segments(p);
    while (pi < p.size()) {
        const bool edge = (pi < ends.first) || (pi > ends.second);
        const bool swap_start = segment_swap_start(p, pi);
        const Seg& s = p[pi];
        unsigned n_seg = 1;
        if (swap_start) n_seg = SwapInfo(p, pi).n_seg;
        if (edge) {
            if (s.type == SK_SEG_DELETE || s.type == SK_SEG_INSERT) {
                const Key* k = (pi < ends.first) ? lead : trail;
                if (!k) throw Fail("edge indel of the input alignment has no indel key");
                out.insert(*k);
            }
        } else if (swap_start) {
            const SwapInfo si(p, pi);
            if (std::max(si.insert_length, si.delete_length) <= job.opt.max_indel_size) {
                Key k;
                k.pos = ref_head;
                k.type = SK_INDEL_INDEL;
                k.del = si.delete_length;
                k.ins = read_substr(code, read_off, read_off + si.insert_length);
                out.insert(k);
            } else {
                Key l, r;
                l.pos = ref_head; l.type = SK_INDEL_BP_LEFT;
                r.pos = ref_head + int32_t(si.delete_length); r.type = SK_INDEL_BP_RIGHT;
                out.insert(l);
                out.insert(r);
            }
        } else if (seg_indel(s.type)) {
            if (s.length <= job.opt.max_indel_size) {
                Key k;
                k.pos = ref_head;
                k.type = SK_INDEL_INDEL;
                if (s.type == SK_SEG_INSERT) k.ins = read_substr(code, read_off, read_off + s.length);
                else k.del = s.length;
                out.insert(k);
            } else {
                Key l, r;
                l.pos = ref_head; l.type = SK_INDEL_BP_LEFT;
                r.pos = ref_head + ((s.type == SK_SEG_INSERT) ? 0 : int32_t(s.length)); r.type = SK_INDEL_BP_RIGHT;
                out.insert(l);
                out.insert(r);
            }
        } else if (include_mismatches && job.has_mismatch_keys && seg_align_match(s.type)) {
            // (a mismatch that is not in the table is dropped by the caller: without mismatch entries there is nothing to look for)
            for (unsigned i = 0; i < s.length; ++i) {
                const unsigned rp = read_off + i;
                const uint8_t sb = rp < code.size() ? code[rp] : uint8_t(SK_BAM_ANY);
                if (sb == SK_BAM_REF || sb == SK_BAM_ANY) continue;
                const int32_t refp = ref_head + int32_t(i);
                if (sb == char_code(job.ref_base(refp))) continue;
                Key k;
                k.pos = refp;
                k.type = SK_INDEL_MISMATCH;
                k.del = 1;
                k.ins = std::string(1, code_char(sb));
                out.insert(k);
            }
        }
        for (unsigned i = 0; i < n_seg; ++i) { // increment_path
            const Seg& q = p[pi];
            if (seg_align_match(q.type)) { read_off += q.length; ref_head += int32_t(q.length); }
            else if (q.type == SK_SEG_DELETE || q.type == SK_SEG_SKIP) ref_head += int32_t(q.length);
            else if (q.type == SK_SEG_INSERT || q.type == SK_SEG_SOFT_CLIP) read_off += q.length;
            ++pi;
        }
    }
}

// ---- the container-free enumeration core (csrc/realign_core.h) on the host: sk_realign_options.enumeration == 1 ----
struct CoreTables
{
    std::vector<skcore::PIndel> tab;
    std::string ins_pool;
    skcore::PJob pj;
};
void build_core_tables(const Job& job, CoreTables& t, uint8_t* consulted)
{
    t.tab.resize(job.tab.size());
    t.ins_pool.clear();
    std::map<std::tuple<int, uint32_t, std::string>, uint32_t> shapes;
    for (size_t i = 0; i < job.tab.size(); ++i) {
        const Indel& d = job.tab[i];
        skcore::PIndel& p = t.tab[i];
        p.pos = d.key.pos;
        p.del = d.key.del;
        p.ins_len = d.key.ins_len();
        p.arid = d.arid;
        p.type = uint8_t(d.key.type);
        p.cand = d.cand ? 1 : 0;
        p.forced = d.forced ? 1 : 0;
        p.ndfr = d.ndfr ? 1 : 0;
        for (int s = 0; s < SK_MAX_SAMPLES; ++s) {
            p.hap[s] = d.hap[s];
            p.bypass[s] = d.bypass[s] ? 1 : 0;
        }
        p.ins_off = uint32_t(t.ins_pool.size());
        t.ins_pool += d.key.ins;
        // what is_equiv_candidate compares, as a number: equal shapes <=> equal (type, deletion length, insert sequence)
        p.shape = shapes.emplace(std::make_tuple(int(d.key.type), d.key.del, d.key.ins), uint32_t(shapes.size())).first->second;
    }
    t.pj.tab = t.tab.data();
    t.pj.n_tab = int32_t(t.tab.size());
    t.pj.max_toggle = job.max_toggle.data();
    t.pj.n_max_toggle = int32_t(job.max_toggle.size());
    t.pj.sample_count = job.opt.sample_count;
    t.pj.max_read_indel_toggle = job.opt.max_read_indel_toggle;
    t.pj.max_candidate_indel_density = job.opt.max_candidate_indel_density;
    t.pj.is_haplotyping_enabled = job.opt.is_haplotyping_enabled;
    t.pj.max_indel_size = int32_t(job.opt.max_indel_size);
    t.pj.consulted = consulted;
    t.pj.max_nodes = 0;
}
bool to_core_cal(const Cal& c, skcore::PCal& p)
{
    if (c.al.path.size() > size_t(skcore::Caps::P) || c.indels.size() > size_t(skcore::Caps::K + 2)) return false;
    p.pos = c.al.pos;
    p.lead = int16_t(c.lead);
    p.trail = int16_t(c.trail);
    p.fwd = c.al.fwd ? 1 : 0;
    p.n_seg = uint8_t(c.al.path.size());
    p.n_indels = uint8_t(c.indels.size());
    p.pad = 0;
    for (size_t i = 0; i < c.al.path.size(); ++i) {
        if (c.al.path[i].length > 0xffffu) return false;
        p.path[i].type = uint16_t(c.al.path[i].type);
        p.path[i].length = uint16_t(c.al.path[i].length);
    }
    for (size_t i = 0; i < c.indels.size(); ++i) p.indels[i] = int16_t(c.indels[i]);
    return true;
}
Cal from_core_cal(const skcore::PCal& p)
{
    Cal c;
    c.al.pos = p.pos;
    c.al.fwd = p.fwd != 0;
    c.lead = p.lead;
    c.trail = p.trail;
    c.al.path.resize(p.n_seg);
    for (int i = 0; i < p.n_seg; ++i) c.al.path[size_t(i)] = Seg{ p.path[i].type, p.path[i].length };
    c.indels.assign(p.indels, p.indels + p.n_indels);
    return c;
}
// the state getCandidateAlignments hands to candidate_alignment_search, in the core's form; false = beyond a cap
bool to_core_read(const Job& job, const std::vector<int>& observed, int sample, unsigned read_length, const Range& realign_range,
                  const Range& exemplar_pr, const StatusMap& sm, const std::vector<int>& order, const Cal& cal, skcore::PRead& r)
{
    if (job.tab.size() > 32000 || sm.size() > size_t(skcore::Caps::K) || order.size() > size_t(skcore::Caps::K) ||
        observed.size() > size_t(skcore::Caps::OBS) || read_length > 0xffffu)
        return false;
    std::memset(&r, 0, sizeof(r));
    r.realign_b = realign_range.b;
    r.realign_e = realign_range.e;
    r.read_length = int32_t(read_length);
    r.sample = sample;
    r.exemplar_range = skcore::mk_range(exemplar_pr.b, exemplar_pr.e);
    if (!to_core_cal(cal, r.cal)) return false;
    r.cal.n_indels = 0;
    int n = 0;
    for (const auto& kv : sm) {
        skcore::PStatus& s = r.sm[n++];
        s.idx = int16_t(kv.first);
        s.is_present = kv.second.is_present;
        s.is_remove_only = kv.second.is_remove_only;
        s.in_original = kv.second.in_original;
        s.pad = 0;
    }
    r.n_sm = uint8_t(n);
    for (size_t i = 0; i < order.size(); ++i) r.order[i] = int16_t(order[i]);
    r.n_order = uint8_t(order.size());
    n = 0;
    for (const int o : observed) r.observed[n++] = int16_t(o);
    r.n_observed = uint8_t(n);
    return true;
}

// getCandidateAlignments :1816-1994
void get_candidate_alignments(const Job& job, sk_realign_job::Read& rd, const std::vector<int>& observed, const Aln& input,
                              const Range& realign_range, std::set<Cal>& cal_set, const bool force_host = false)
{
    const unsigned read_length = unsigned(rd.code.size());
    StatusMap sm;
    HapMap hm;
    std::vector<int> order;

    Cal cal;
    cal.al = input;
    bool has_lead, has_trail;
    Key lead_k, trail_k;
    input_edge_keys(input, rd.code, has_lead, lead_k, has_trail, trail_k);
    if (has_lead) {
        cal.lead = job.find_key(lead_k);
        if (cal.lead < 0) throw Fail("leading edge indel of the input alignment is not in the indel table");
    }
    if (has_trail) {
        cal.trail = job.find_key(trail_k);
        if (cal.trail < 0) throw Fail("trailing edge indel of the input alignment is not in the indel table");
    }

    SearchCtx ctx{ job, observed, rd.sample, read_length, realign_range, cal_set, rd.warn_origin, rd.warn_toggle };
    const Range exemplar_pr = soft_clip_range(cal.al);
    add_indels_in_range(ctx, exemplar_pr, sm, order);
    {
        std::set<Key> keys;
        alignment_indel_keys(job, cal.al, has_lead ? &lead_k : nullptr, has_trail ? &trail_k : nullptr, rd.code, true, keys);
        ISet valid;
        bool recompute = false;
        for (const Key& k : keys) {
            const int idx = job.find_key(k);
            if (idx < 0 || sm.find(idx) == sm.end()) {
                if (k.is_mismatch()) continue;
                throw Fail("Exemplar alignment contains indel not found in the overlap indel set");
            }
            if (k.is_mismatch()) recompute = true;
            sm[idx].is_present = true;
            sm[idx].in_original = true;
            iset_insert(valid, idx);
        }
        if (recompute) {
            const int32_t read_start = int32_t(unaligned_prefix(cal.al.path));
            cal = make_start_pos_alignment(job.tab, cal.al.pos, read_start, cal.al.fwd, read_length, valid);
        }
    }
    order.clear();
    for (const auto& kv : sm) if (kv.second.is_present) order.push_back(kv.first);
    for (const auto& kv : sm) if (!kv.second.is_present) order.push_back(kv.first);
    sort_remove_only_indels_last(sm, order);

    unsigned cal_read_length = read_length, hc_lead = 0, hc_trail = 0, sc_lead = 0, sc_trail = 0;
    const bool clipped = path_is_clipped(cal.al.path);
    if (clipped) {
        clip_clipper(cal.al.path, hc_lead, hc_trail, sc_lead, sc_trail);
        cal_read_length -= (sc_lead + sc_trail);
    }
    ctx.read_length = cal_read_length;
    bool searched = false;
    if (job.opt.enumeration == 2 && !force_host) {
        std::shared_ptr<skcore::PRead> pr(new skcore::PRead);
        if (to_core_read(job, observed, rd.sample, cal_read_length, realign_range, exemplar_pr, sm, order, cal, *pr)) {
            pr->clipped = clipped ? 1 : 0;
            pr->hc_lead = hc_lead;
            pr->hc_trail = hc_trail;
            pr->sc_lead = sc_lead;
            pr->sc_trail = sc_trail;
            rd.pending = pr;
            return;
        }
        job.n_fallback_reads.fetch_add(1, std::memory_order_relaxed);
    }
    if (job.opt.enumeration == 1) {
        // the container-free core, on this thread (enumeration == 2 runs the same code on the device, read_enumerate.hip)
        std::unique_ptr<skcore::PRead> pr(new skcore::PRead);
        if (to_core_read(job, observed, rd.sample, cal_read_length, realign_range, exemplar_pr, sm, order, cal, *pr)) {
            CoreTables tables;
            std::vector<uint8_t> consulted(job.tab.size() + 1, 0);
            build_core_tables(job, tables, consulted.data());
            std::vector<skcore::PFrame> stack(2 * skcore::Caps::K + 5);
            std::set<Cal> leaves;
            bool too_many = false;
            skcore::PCal tmp_cal;
            auto sink = [&](skcore::PCal& leaf) -> bool {
                leaves.insert(from_core_cal(leaf));
                if (leaves.size() > 100000) too_many = true;
                return !too_many;
            };
            const skcore::SearchOut so =
                skcore::candidate_alignment_search(tables.pj, *pr, stack.data(), int(stack.size()) - 1, &stack.back(), &tmp_cal, sink);
            // (anything else: the container-based code below redoes the read and reports what the reference would)
            if (so.status == skcore::ST_OK) {
                for (size_t i = 0; i < job.tab.size(); ++i)
                    if (consulted[i]) (void)job.cand(int(i));
                if (so.warn_origin) rd.warn_origin = true;
                if (so.warn_toggle) rd.warn_toggle = true;
                cal_set.swap(leaves);
                searched = true;
                job.n_core_reads.fetch_add(1, std::memory_order_relaxed);
            }
        }
        if (!searched) job.n_fallback_reads.fetch_add(1, std::memory_order_relaxed);
    }
    if (!searched) candidate_alignment_search(ctx, sm, hm, order, 0, 0, 0, exemplar_pr, job.opt.max_read_indel_toggle, cal);

    if (clipped) {
        std::set<Cal> s2;
        s2.swap(cal_set);
        for (Cal c : s2) {
            clip_adder(c.al.path, hc_lead, hc_trail, sc_lead, sc_trail);
            cal_set.insert(c);
        }
    }
    { // drop out-of-range candidates (:1981-1993)
        std::set<Cal> s2;
        s2.swap(cal_set);
        for (const Cal& c : s2)
            if (realign_range.superset_of(strict_range(c.al))) cal_set.insert(c);
    }
}

// ------------------------------------------------------------------------------------------------------- stage 3

struct PathInfo // extra_path_info :1281-1321
{
    unsigned indel_count = 0, del_size = 0, ins_size = 0, sum_seg_pos = 0;
};
PathInfo path_info(const Path& p)
{
    PathInfo e;
    unsigned read_pos = 0;
    for (const Seg& s : p) {
        if (!seg_align_match(s.type)) e.indel_count++;
        if (s.type == SK_SEG_DELETE) { e.del_size += s.length; e.sum_seg_pos += read_pos; }
        if (s.type == SK_SEG_INSERT) { e.ins_size += s.length; e.sum_seg_pos += read_pos; }
        if (seg_read_len(s.type)) read_pos += s.length;
    }
    return e;
}
unsigned candidate_indel_count(const Job& job, const Cal& c) // :1325-1337
{
    unsigned v = 0;
    for (int i : c.indels) if (job.cand(i)) ++v;
    return v;
}
bool first_cal_preferred(const Job& job, const Cal& c1, const Cal& c2) // isFirstCandidateAlignmentPreferred :1351-1376
{
    const PathInfo e1 = path_info(c1.al.path), e2 = path_info(c2.al.path);
    if (e2.indel_count < e1.indel_count) return false;
    if (e2.indel_count > e1.indel_count) return true;
    const unsigned k1 = candidate_indel_count(job, c1), k2 = candidate_indel_count(job, c2);
    if (k2 > k1) return false;
    if (k2 < k1) return true;
    if (e2.ins_size < e1.ins_size) return false;
    if (e2.ins_size > e1.ins_size) return true;
    if (e2.del_size < e1.del_size) return false;
    if (e2.del_size > e1.del_size) return true;
    return e2.sum_seg_pos >= e1.sum_seg_pos;
}

// ---- clipper, starling_read_align_clipper.cpp
struct RefMap
{
    enum T { NONE, MATCH, INSERT, SOFT_CLIP, CONFLICT } type = NONE;
    int32_t pos = 0;
};
void alignment_ref_map(const Aln& al, std::vector<RefMap>& m) // get_alignment_ref_map :109-157
{
    m.clear();
    int32_t ref_head = al.pos;
    for (const Seg& s : al.path) {
        if (seg_align_match(s.type)) {
            for (unsigned j = 0; j < s.length; ++j) { RefMap r; r.type = RefMap::MATCH; r.pos = ref_head + int32_t(j); m.push_back(r); }
            ref_head += int32_t(s.length);
        } else if (s.type == SK_SEG_INSERT) {
            for (unsigned j = 0; j < s.length; ++j) { RefMap r; r.type = RefMap::INSERT; m.push_back(r); }
        } else if (s.type == SK_SEG_DELETE || s.type == SK_SEG_SKIP) {
            ref_head += int32_t(s.length);
        } else if (s.type == SK_SEG_SOFT_CLIP) {
            for (unsigned j = 0; j < s.length; ++j) { RefMap r; r.type = RefMap::SOFT_CLIP; m.push_back(r); }
        } else if (s.type != SK_SEG_HARD_CLIP) {
            throw Fail("Can't handle cigar code");
        }
    }
}
void mark_ref_map_conflicts(const Aln& al, std::vector<RefMap>& m) // :161-231
{
    int32_t ref_head = al.pos, read_head = 0;
    for (const Seg& s : al.path) {
        if (seg_align_match(s.type) || s.type == SK_SEG_INSERT || s.type == SK_SEG_SOFT_CLIP) {
            for (unsigned j = 0; j < s.length; ++j) {
                RefMap& r = m[size_t(read_head) + j];
                if (r.type == RefMap::CONFLICT) continue;
                bool ok;
                if (seg_align_match(s.type)) ok = (r.type == RefMap::MATCH) && (r.pos == ref_head + int32_t(j));
                else if (s.type == SK_SEG_INSERT) ok = (r.type == RefMap::INSERT);
                else ok = (r.type == RefMap::SOFT_CLIP);
                if (!ok) r.type = RefMap::CONFLICT;
            }
            read_head += int32_t(s.length);
            if (seg_align_match(s.type)) ref_head += int32_t(s.length);
        } else if (s.type == SK_SEG_DELETE || s.type == SK_SEG_SKIP) {
            ref_head += int32_t(s.length);
        } else if (s.type != SK_SEG_HARD_CLIP) {
            throw Fail("Can't handle cigar code");
        }
    }
}
void extend_or_add_sc(Aln& al, unsigned len)
{
    if (!al.path.empty() && al.path.back().type == SK_SEG_SOFT_CLIP) al.path.back().length += len;
    else al.path.push_back(Seg{ SK_SEG_SOFT_CLIP, len });
}
void soft_clip_alignment(Aln& al, unsigned leading_clip, unsigned trailing_clip) // :250-339
{
    unsigned read_head = 0;
    Aln n;
    n.pos = al.pos;
    n.fwd = al.fwd;
    for (const Seg& s : al.path) {
        if (seg_align_match(s.type) || s.type == SK_SEG_INSERT) {
            if (leading_clip > read_head) {
                const unsigned clip = std::min(s.length, leading_clip - read_head);
                extend_or_add_sc(n, clip);
                if (seg_align_match(s.type)) n.pos += int32_t(clip);
                if (clip < s.length) n.path.push_back(Seg{ s.type, s.length - clip });
            } else if (trailing_clip < read_head + s.length) {
                const unsigned clip = std::min(s.length, (read_head + s.length) - trailing_clip);
                if (clip < s.length) n.path.push_back(Seg{ s.type, s.length - clip });
                extend_or_add_sc(n, clip);
            } else {
                n.path.push_back(s);
            }
            read_head += s.length;
        } else if (s.type == SK_SEG_DELETE || s.type == SK_SEG_SKIP) {
            if (leading_clip >= read_head) n.pos += int32_t(s.length);
            else if (trailing_clip <= read_head) { }
            else n.path.push_back(s);
        } else if (s.type == SK_SEG_SOFT_CLIP) {
            extend_or_add_sc(n, s.length);
            read_head += s.length;
        } else if (s.type == SK_SEG_HARD_CLIP) {
            n.path.push_back(s);
        } else {
            throw Fail("Can't handle cigar code");
        }
    }
    al = n;
}
void clipped_alignment_from_pool(const std::vector<const Cal*>& pool, unsigned best, Aln& out) // :343-424
{
    out = pool[best]->al;
    if (pool.size() == 1) return;
    std::vector<RefMap> m;
    alignment_ref_map(out, m);
    for (unsigned i = 0; i < pool.size(); ++i)
        if (i != best) mark_ref_map_conflicts(pool[i]->al, m);
    const unsigned n = unsigned(m.size());
    unsigned lead = 0;
    for (; lead < n; ++lead) if (m[lead].type == RefMap::MATCH) break;
    for (; lead > 0; --lead) if (m[lead - 1].type == RefMap::CONFLICT || m[lead - 1].type == RefMap::SOFT_CLIP) break;
    unsigned trail = n;
    for (; trail > 0; --trail) if (m[trail - 1].type == RefMap::MATCH) break;
    for (; trail < n; ++trail) if (m[trail].type == RefMap::CONFLICT || m[trail].type == RefMap::SOFT_CLIP) break;
    if (lead >= trail) {
        out = Aln();
        return;
    }
    if (lead != 0 || trail != n) soft_clip_alignment(out, lead, trail);
}

// scoreCandidateAlignments :1534-1741 (unpinned read segments; isTestSoftClippedInputAligned == false)
void select_alignments(const Job& job, sk_realign_job::Read& rd, const double* scores, double& max_score, const Cal*& max_cal)
{
    max_cal = nullptr;
    max_score = 0;
    const size_t n = rd.cals.size();
    for (size_t i = 0; i < n; ++i) {
        const Cal& c = rd.cals[i];
        const double lnp = scores[i];
        if (max_cal != nullptr) {
            if (lnp < max_score) continue;
            if (lnp <= max_score && first_cal_preferred(job, *max_cal, c)) continue;
        }
        max_score = lnp;
        max_cal = &c;
    }
    const double max_allowed = max_score;
    const double allowed_range = job.opt.is_smoothed_alignments ? job.opt.smoothed_lnp_range : 0.;
    const Cal* smooth_cal = nullptr;
    std::vector<const Cal*> pool;
    for (size_t i = 0; i < n; ++i) {
        if ((scores[i] + allowed_range) < max_allowed) continue;
        const Cal& c = rd.cals[i];
        pool.push_back(&c);
        if (smooth_cal == nullptr || !first_cal_preferred(job, *smooth_cal, c)) smooth_cal = &c;
    }
    if (!smooth_cal) throw Fail("no alignment in the smooth pool");
    rd.realigned = true;
    // finishRealignment :1409-1449
    if (pool.size() > 1) {
        unsigned best = unsigned(pool.size());
        for (unsigned i = 0; i < pool.size(); ++i) if (pool[i] == smooth_cal) { best = i; break; }
        clipped_alignment_from_pool(pool, best, rd.realignment);
        if (rd.realignment.empty()) rd.realignment = smooth_cal->al;
    } else {
        rd.realignment = smooth_cal->al;
    }
}

// get_alignment_indel_bp_overlap, starling_read_align_score_indels.cpp:134-228
std::pair<int, int> alignment_indel_bp_overlap(unsigned upstream_oligo, const Aln& al, const Key& k)
{
    int32_t read_head = 0, ref_head = al.pos;
    bool has_l = false, has_r = false;
    int32_t lpos = 0, rpos = 0;
    for (const Seg& s : al.path) {
        int32_t nread = read_head, nref = ref_head;
        if (seg_align_match(s.type)) { nread += int32_t(s.length); nref += int32_t(s.length); }
        else if (s.type == SK_SEG_INSERT) nread += int32_t(s.length);
        else if (s.type == SK_SEG_DELETE) nref += int32_t(s.length);
        else if (s.type == SK_SEG_SOFT_CLIP || s.type == SK_SEG_HARD_CLIP) { }
        else throw Fail("unexpected CIGAR type in breakpoint overlap");
        if (!has_l && k.pos <= nref) { lpos = read_head + (k.pos - ref_head); has_l = true; }
        if (!has_r && k.right_pos() < nref) { rpos = read_head + (k.right_pos() - ref_head); has_r = true; }
        read_head = nread;
        ref_head = nref;
    }
    int lext = 0, rext = 0;
    if (al.fwd) { if (lpos > 0) lext = int(upstream_oligo); }
    else { if ((read_head - rpos) > 0) rext = int(upstream_oligo); }
    int lo = 0, ro = 0;
    if (has_l) lo = std::max(0, std::min(lpos + lext, read_head - lpos));
    if (has_r) ro = std::max(0, std::min(rpos, (read_head - rpos) + rext));
    return std::make_pair(lo, ro);
}

// late_indel_normalization_filter :303-450.  NOTE: the reference passes its `nonnorm_indels` set BY VALUE (:310), so
// the set the caller later consults stays empty; only the alignment filter flags and the max alignment come back.
void late_indel_normalization_filter(const Job& job, const sk_realign_job::Read& rd, const double* scores,
                                     std::vector<bool>& is_filtered, double& max_score, const Cal*& max_cal)
{
    const unsigned n = unsigned(rd.cals.size());
    const double equiv_range = job.opt.is_smoothed_alignments ? job.opt.smoothed_lnp_range : 0.;
    std::vector<std::pair<double, unsigned>> sorted;
    sorted.reserve(n);
    for (unsigned i = 0; i < n; ++i) sorted.push_back(std::make_pair(scores[i], i));
    std::sort(sorted.rbegin(), sorted.rend());
    std::vector<double> smooth(scores, scores + n);
    std::vector<std::pair<int, int>> pairs;
    bool any_excluded = false;
    for (unsigned i1 = 0; i1 < n; ++i1) {
        const unsigned s1 = sorted[i1].second;
        if (is_filtered[s1]) continue;
        for (unsigned i2 = i1 + 1; i2 < n; ++i2) {
            const unsigned s2 = sorted[i2].second;
            if (is_filtered[s2]) continue;
            if (smooth[s2] + equiv_range < smooth[s1]) break;
            // is_equiv_candidate :240-269
            const ISet& a = rd.cals[s1].indels;
            const ISet& b = rd.cals[s2].indels;
            if (a.size() != b.size()) continue;
            // (the reference's std::set of key pairs: a is ascending without repeats, so the pairs are in set order as found)
            pairs.clear();
            bool equiv = true;
            for (size_t q = 0; q < a.size(); ++q) {
                if (a[q] == b[q]) continue;
                const Key& k1 = job.key(a[q]);
                const Key& k2 = job.key(b[q]);
                if (k1.type != k2.type || k1.del != k2.del || k1.ins != k2.ins) { equiv = false; break; }
                pairs.push_back(std::make_pair(a[q], b[q]));
            }
            if (!equiv || pairs.empty()) continue;
            bool s1_removed = false, removed = false;
            for (const auto& pr : pairs) {
                // is_first_indel_dominant :276-292
                const bool c1 = job.cand(pr.first), c2 = job.cand(pr.second);
                bool first_dom;
                if (c2 && !c1) first_dom = false;
                else if (c2 == c1) first_dom = (job.key(pr.first).pos <= job.key(pr.second).pos);
                else first_dom = true;
                if (first_dom) {
                    if (!removed) {
                        is_filtered[s2] = true;
                        any_excluded = true;
                        smooth[s1] = std::max(smooth[s1], smooth[s2]);
                    }
                } else {
                    if (!removed) {
                        is_filtered[s1] = true;
                        any_excluded = true;
                        smooth[s2] = std::max(smooth[s1], smooth[s2]);
                        s1_removed = true;
                    }
                }
                removed = true;
            }
            if (s1_removed) break;
        }
    }
    if (any_excluded) {
        for (unsigned i = 0; i < n; ++i) {
            const unsigned s = sorted[i].second;
            if (is_filtered[s]) continue;
            max_score = scores[s];
            max_cal = &rd.cals[s];
            break;
        }
    }
}

// The reference's iks_map_t (:54): best score per (evaluated indel, (is present, which indel)).  Both indels always come from
// the read's evaluated set, so the map is a dense [evaluated][present][evaluated] table over positions in that (sorted) set.
struct ScoringInfo // indexed by POSITION in the evaluated-indel list (call, present?, which)
{
    int n;
    std::vector<double> val;
    std::vector<char> has;
    explicit ScoringInfo(const size_t n_eval) : n(int(n_eval)), val(size_t(2) * n_eval * n_eval), has(val.size(), 0) {}
    size_t slot(const int call, const bool present, const int which) const { return (size_t(call) * 2 + (present ? 1 : 0)) * size_t(n) + size_t(which); }
    void update(const int call, const bool present, const int which, const double lnp) // updateIndelScoringInfo :61-77
    {
        const size_t k = slot(call, present, which);
        if (has[k] && val[k] >= lnp) return;
        has[k] = 1;
        val[k] = lnp;
    }
    bool find(const int call, const bool present, const int which, double& out) const
    {
        const size_t k = slot(call, present, which);
        if (!has[k]) return false;
        out = val[k];
        return true;
    }
};

// score_indels :454-1079
void score_indels(const Job& job, sk_realign_job::Read& rd, const double* scores, double max_score, const Cal* max_cal)
{
    const unsigned n = unsigned(rd.cals.size());
    std::vector<bool> is_filtered(n, false);
    late_indel_normalization_filter(job, rd, scores, is_filtered, max_score, max_cal);

    const Cal& mc = *max_cal;
    ISet to_eval;
    {
        const Range mr = soft_clip_range(mc.al);
        const auto it = job.range_iter(mr.b, mr.e);
        for (int e = it.first; e < it.second; ++e) {
            const Key& k = job.key(e);
            if (k.is_mismatch()) continue;
            if (!job.cand(e)) continue;
            const bool in_max = iset_has(mc.indels, e);
            const Cal* best = nullptr;
            if (in_max) {
                best = &mc;
            } else {
                double best_score = 0;
                for (unsigned i = 0; i < n; ++i) {
                    const Cal& c = rd.cals[i];
                    if (&c == &mc) continue;
                    if (is_filtered[i]) continue;
                    if (!iset_has(c.indels, e)) continue;
                    if (best == nullptr || scores[i] > best_score) {
                        best_score = scores[i];
                        best = &c;
                    }
                }
            }
            if (best == nullptr) continue;
            const auto bpo2 = alignment_indel_bp_overlap(job.opt.upstream_oligo_size, best->al, k);
            const int bpo = std::max(bpo2.first, bpo2.second);
            if (bpo < job.opt.min_read_bp_flank) {
                if (bpo > 0) rd.suboverlap.push_back(job.tab[e].orig);
                continue;
            }
            to_eval.push_back(e);
        }
    }
    // orthogonal (conflicting) evaluated indels of each evaluated indel, as positions in to_eval, ascending -- the reference's
    // std::map<IndelKey, std::set<IndelKey>>; (its operator[] creating empty entries has no observable effect)
    const size_t ne = to_eval.size();
    std::vector<std::vector<int>> ortho(ne);
    for (size_t i = 0; i < ne; ++i)
        for (size_t j = i + 1; j < ne; ++j)
            if (is_indel_conflict(job.key(to_eval[i]), job.key(to_eval[j]))) {
                ortho[i].push_back(int(j));
                ortho[j].push_back(int(i));
            }
    for (auto& o : ortho) std::sort(o.begin(), o.end());

    ScoringInfo info(ne);
    std::vector<char> in_cal(ne);
    ISet noncand_ortho;
    for (unsigned ci = 0; ci < n; ++ci) {
        if (is_filtered[ci]) continue;
        const Cal& c = rd.cals[ci];
        const double score = scores[ci];
        { // which evaluated indels the alignment holds: one merge over two ascending lists
            size_t a = 0;
            for (size_t q = 0; q < ne; ++q) {
                while (a < c.indels.size() && c.indels[a] < to_eval[q]) ++a;
                in_cal[q] = (a < c.indels.size() && c.indels[a] == to_eval[q]) ? 1 : 0;
            }
        }
        noncand_ortho.clear();
        for (size_t q = 0; q < ne; ++q) {
            const int e = to_eval[q];
            const Indel& ed = job.tab[e];
            if (in_cal[q]) {
                info.update(int(q), true, int(q), score);
                info.update(int(q), false, int(q), score + ed.r2i);
                for (const int o : ortho[q]) {
                    info.update(o, false, o, score + ed.r2i);
                    info.update(o, true, int(q), score);
                }
            } else {
                // which_interfering_indel :100-119
                int interfering = -1;
                for (const int cur : c.indels) {
                    if (job.key(cur).is_mismatch()) continue;
                    if (is_indel_conflict(job.key(cur), job.key(e))) { interfering = cur; break; }
                }
                if (interfering >= 0 && !iset_has(to_eval, interfering)) iset_insert(noncand_ortho, interfering);
                if (interfering < 0) {
                    info.update(int(q), false, int(q), score);
                    info.update(int(q), true, int(q), score + ed.i2r);
                } else {
                    info.update(int(q), true, int(q), score + ed.i2r);
                }
            }
        }
        for (const int nc : noncand_ortho) {
            for (size_t q = 0; q < ne; ++q) {
                if (!is_indel_conflict(job.key(nc), job.key(to_eval[q]))) continue;
                info.update(int(q), false, int(q), score + job.tab[nc].r2i);
            }
        }
    }

    const unsigned read_length = unsigned(rd.code.size());
    uint16_t non_ambig = 0;
    for (uint8_t c : rd.code) if (c != SK_BAM_ANY) ++non_ambig;
    const bool tier1 = (rd.map_level == SK_MAPLEVEL_TIER1);
    for (size_t q = 0; q < ne; ++q) {
        const int e = to_eval[q];
        const Key& k = job.key(e);
        const bool in_max = iset_has(mc.indels, e);
        double indel_score = max_score;
        if (!in_max && !info.find(int(q), true, int(q), indel_score)) continue; // incomplete search or "safe mode" warning: skipped either way
        double ref_score = 0;
        if (!info.find(int(q), false, int(q), ref_score)) continue;
        const Range rr(k.pos - 1, k.right_pos() + 1);
        const int32_t read_pos = lowest_fwd_read_pos_for_ref_range(mc.al, rr);
        int32_t edge_dist = int32_t(read_length);
        {
            Aln rev = mc.al;
            rev.fwd = !rev.fwd;
            const int32_t rev_pos = lowest_fwd_read_pos_for_ref_range(rev, rr);
            if (read_pos >= 0) edge_dist = read_pos;
            if (rev_pos >= 0 && rev_pos < edge_dist) edge_dist = rev_pos;
        }
        sk_read_path_scores s;
        std::memset(&s, 0, sizeof(s));
        s.indel = job.tab[e].orig;
        s.ref_lnp = static_cast<float>(ref_score);
        s.indel_lnp = static_cast<float>(indel_score);
        s.non_ambig = non_ambig;
        s.read_length = uint16_t(read_length);
        s.is_tier1_read = tier1;
        s.is_fwd_strand = mc.al.fwd;
        s.read_pos = int16_t(read_pos);
        s.distance_from_closest_read_edge = int16_t(edge_dist);
        for (const int oq : ortho[q]) {
            const int o = to_eval[size_t(oq)];
            double alt_score;
            if (!info.find(int(q), true, oq, alt_score)) continue;
            // ReadPathScores::insertAlt, IndelData.cpp:40-68: keep the two best
            const float a = static_cast<float>(alt_score);
            if (s.n_alt < 2) {
                s.alt_indel[s.n_alt] = job.tab[o].orig;
                s.alt_lnp[s.n_alt] = a;
                s.n_alt++;
            } else {
                int min_index = 2;
                float mn = a;
                for (int qq = 0; qq < 2; ++qq)
                    if (s.alt_lnp[qq] < mn) { mn = s.alt_lnp[qq]; min_index = qq; }
                if (min_index < 2) {
                    s.alt_indel[min_index] = job.tab[o].orig;
                    s.alt_lnp[min_index] = a;
                }
            }
        }
        rd.scores.push_back(s);
    }
}

void cal_to_c(const Job& job, const Cal& c, std::vector<sk_path_seg>& segs, std::vector<sk_indel_key>& keys,
              sk_candidate_alignment& out)
{
    segs.clear();
    for (const Seg& s : c.al.path) segs.push_back(sk_path_seg{ s.type, s.length });
    keys.clear();
    auto mk = [&](int i) {
        sk_indel_key k;
        std::memset(&k, 0, sizeof(k));
        if (i < 0) return k;
        const Indel& d = job.tab[i];
        k.pos = d.key.pos;
        k.type = d.key.type;
        k.del_len = d.key.del;
        k.ins_len = d.key.ins_len();
        k.ins_seq = d.key.ins.c_str();
        k.is_candidate = job.cand(i) ? 1 : 0; // scoreCandidateAlignment asks for every indel of the alignment (:471-487)
        return k;
    };
    for (int i : c.indels) keys.push_back(mk(i));
    out.pos = c.al.pos;
    out.n_seg = int32_t(segs.size());
    out.path = segs.data();
    out.n_indels = int32_t(keys.size());
    out.indels = keys.data();
    out.leading = mk(c.lead);
    out.trailing = mk(c.trail);
}

Key key_from_c(const sk_indel_key& k)
{
    Key r;
    r.pos = k.pos;
    r.type = k.type;
    r.del = k.del_len;
    if (k.ins_seq && k.ins_len) r.ins.assign(k.ins_seq, k.ins_len);
    return r;
}

} // namespace

// =====================================================================================================================

extern "C" {

void sk_realign_options_default(sk_realign_options* o)
{
    o->max_read_indel_toggle = 5;
    o->max_candidate_indel_density = 0.15;
    o->max_realignment_candidates = 5000;
    o->max_indel_size = 49;
    o->is_smoothed_alignments = 1;
    volatile double ten = 10.;
    o->smoothed_lnp_range = std::log(ten);
    o->upstream_oligo_size = 0;
    o->is_haplotyping_enabled = 0;
    o->min_read_bp_flank = 5;
    o->sample_count = 1;
    o->host_threads = 1; // the reference runs one process per core; an adapter that owns more cores raises this
    // the device pipeline where there is a device (the CPU double of the ABI, oracle/abi_double.cpp, has none);
    // SK_ENUMERATION overrides the default for runs of whole suites / of the adapter in one mode
    o->enumeration = sk_enum_device_available() ? 2 : 0;
    if (const char* e = std::getenv("SK_ENUMERATION")) o->enumeration = std::atoi(e);
}

sk_realign_job* sk_realign_job_create(const sk_realign_options* opt)
{
    if (!opt || opt->sample_count < 1 || opt->sample_count > SK_MAX_SAMPLES) return nullptr;
    sk_realign_job* j = new sk_realign_job();
    j->opt = *opt;
    build_align_limit(*j);
    j->builder = sk_align_builder_create();
    return j;
}

void sk_realign_job_destroy(sk_realign_job* j)
{
    if (!j) return;
    sk_align_builder_destroy(j->builder);
    delete j;
}

const char* sk_realign_job_error(const sk_realign_job* j) { return j ? j->error.c_str() : "null job"; }

int sk_realign_job_set_reference(sk_realign_job* j, const char* seq, int32_t off, int32_t len)
{
    if (!j || (len > 0 && !seq) || len < 0) return 1;
    j->ref.assign(seq ? seq : "", size_t(len));
    j->ref_offset = off;
    return 0;
}

int sk_realign_job_set_indels(sk_realign_job* j, const sk_indel_info* indels, int32_t n)
{
    if (!j || n < 0 || (n > 0 && !indels)) return 1;
    if (!j->reads.empty()) {
        j->error = "sk_realign_job_set_indels: clear the job's reads first";
        return 1;
    }
    std::vector<Indel> t(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        const sk_indel_info& s = indels[i];
        Indel& d = t[size_t(i)];
        d.key = key_from_c(s.key);
        if (d.key.type == SK_INDEL_NONE) {
            j->error = "sk_realign_job_set_indels: indel of type NONE";
            return 1;
        }
        d.cand = s.key.is_candidate != 0;
        d.r2i = s.ref_to_indel_log_prob;
        d.i2r = s.indel_to_ref_log_prob;
        d.arid = s.active_region_id;
        for (int q = 0; q < SK_MAX_SAMPLES; ++q) {
            d.hap[q] = s.haplotype_id[q];
            d.bypass[q] = s.is_haplotyping_bypassed[q] != 0;
        }
        d.forced = s.is_forced_output != 0;
        d.ndfr = s.not_discovered_from_reads != 0;
        d.orig = i;
    }
    std::stable_sort(t.begin(), t.end(), [](const Indel& a, const Indel& b) { return a.key < b.key; });
    for (size_t i = 1; i < t.size(); ++i)
        if (t[i - 1].key == t[i].key) {
            j->error = "sk_realign_job_set_indels: duplicate indel key";
            return 1;
        }
    j->tab.swap(t);
    j->consulted.reset(new std::atomic<uint8_t>[j->tab.size() + 1]);
    for (size_t i = 0; i <= j->tab.size(); ++i) j->consulted[i].store(0, std::memory_order_relaxed);
    j->orig_to_tab.assign(size_t(n), -1);
    for (size_t i = 0; i < j->tab.size(); ++i) j->orig_to_tab[size_t(j->tab[i].orig)] = int(i);
    j->has_mismatch_keys = false;
    for (const Indel& d : j->tab)
        if (d.key.is_mismatch()) j->has_mismatch_keys = true;
    return 0;
}

int sk_realign_job_rescore(const int32_t reps, float* out_ms, int32_t* out_n_reads, int32_t* out_n_cals, int64_t* out_cells)
{
    return sk_enum_device_rescore(reps, out_ms, out_n_reads, out_n_cals, out_cells);
}

int64_t sk_realign_reference_reads_outside(void) { return g_sk_ref_reads_outside.load(std::memory_order_relaxed); }

void sk_realign_device_job_counts(int64_t* n_one_wait, int64_t* n_one_wait_redone, int64_t* n_staged)
{
    sk_enum_device_job_counts(n_one_wait, n_one_wait_redone, n_staged);
}

int sk_realign_job_enumeration_counts(const sk_realign_job* j, int64_t* n_core, int64_t* n_device, int64_t* n_fallback)
{
    if (!j) return 1;
    if (n_core) *n_core = j->n_core_reads.load();
    if (n_device) *n_device = j->n_device_reads.load();
    if (n_fallback) *n_fallback = j->n_fallback_reads.load();
    return 0;
}

int sk_realign_job_stage3_counts(const sk_realign_job* j, int64_t* n_core, int64_t* n_device)
{
    if (!j) return 1;
    if (n_core) *n_core = j->n_stage3_core.load();
    if (n_device) *n_device = j->n_stage3_device.load();
    return 0;
}

int sk_realign_job_indels_consulted(const sk_realign_job* j, uint8_t* out, int32_t n_indels)
{
    if (!j || !out || n_indels != int32_t(j->orig_to_tab.size())) return 1;
    for (int32_t i = 0; i < n_indels; ++i)
        out[i] = j->consulted ? j->consulted[size_t(j->orig_to_tab[size_t(i)])].load(std::memory_order_relaxed) : 0;
    return 0;
}

void sk_realign_job_clear_reads(sk_realign_job* j)
{
    if (!j) return;
    j->reads.clear();
    j->dev_scores.clear();
    j->n_cals_total = 0;
    sk_align_builder_clear(j->builder);
    j->finished = false;
    j->error.clear();
}

int sk_realign_job_n_reads(const sk_realign_job* j) { return j ? int(j->reads.size()) : 0; }

} // extern "C"

namespace
{

// the gate, input normalisation and enumeration of a validated read (rd.code/input/observed/realign range are set); with
// enumeration == 2 and `force_host` false the search itself is left pending for the device
// is_realignable :2045 and check_for_candidate_indel_overlap :217-270: does the read go on to the search at all?
bool passes_gate(const Job& job, const Aln& input, const unsigned read_len, const Range& realign_range)
{
    if (is_overmax(input, job.opt.max_indel_size)) return false;
    const Range rr = alignment_zone(input, read_len);
    if (!realign_range.superset_of(rr)) return false;
    const auto it = job.range_iter(rr.b, rr.e);
    for (int i = it.first; i < it.second; ++i) {
        if (!range_intersect_indel_breakpoints(rr, job.key(i))) continue;
        if (job.cand(i)) return true;
    }
    return false;
}

void enumerate_read(const Job& job, sk_realign_job::Read& rd, const bool force_host)
{
    const Range realign_range(rd.realign_b, rd.realign_e);
    const unsigned read_len = rd.read_len;
    rd.pending.reset();
    rd.warn_origin = rd.warn_toggle = false;

    std::set<Cal> cal_set;
    const bool gate = passes_gate(job, rd.input, read_len, realign_range);
    if (gate) {
        // normalizeInputAlignmentIndels :2001-2021 (no pinned edges on DNA reads)
        Aln norm = rd.input;
        if (is_edge_readref_len_segment(norm.path)) norm = matchify_edge_indels(norm, true, true);
        if (path_is_soft_clipped(norm.path)) norm = matchify_edge_segment_type(norm, SK_SEG_SOFT_CLIP); // :2051-2057
        if (norm.pos >= 0) {
            get_candidate_alignments(job, rd, rd.observed, norm, realign_range, cal_set, force_host);
            if (cal_set.empty() && !rd.pending) throw Fail("Empty candidate alignment set while realigning normed input alignment");
        }
    }
    rd.incomplete_search = rd.warn_origin || rd.warn_toggle;
    rd.cals.assign(cal_set.begin(), cal_set.end());
}

// stage 1 for one read: validation, gate, normalisation, enumeration.  Touches nothing but `rd` (the job is read-only here),
// so reads can be prepared concurrently.  Throws Fail.
void prepare_read(const Job& job, const sk_read_input* in, sk_realign_job::Read& rd)
{
    if (in->read_len < 0 || in->n_seg < 0 || in->n_observed < 0) throw Fail("negative read, path or observed-list length");
    if ((in->read_len > 0 && (in->read_code == nullptr || in->read_qual == nullptr)) || (in->n_seg > 0 && in->path == nullptr) ||
        (in->n_observed > 0 && in->observed == nullptr))
        throw Fail("null read_code / read_qual / path / observed pointer with a positive length");
    if (in->sample_index < 0 || in->sample_index >= job.opt.sample_count) throw Fail("sample_index out of range");
    rd.read_len = uint32_t(in->read_len);
    rd.map_level = in->map_level;
    rd.sample = in->sample_index;
    rd.input.pos = in->pos;
    rd.input.fwd = in->is_fwd_strand != 0;
    rd.input.path.reserve(size_t(in->n_seg));
    for (int i = 0; i < in->n_seg; ++i) {
        // the reference cuts a read with SKIP segments into exon segments realigned with pinned edges
        // (get_segment_edge_pin, starling_read_align.cpp:1711-1737): that RNA path is not built, say so instead of
        // realigning the read as if it were one DNA segment
        if (in->path[i].type == SK_SEG_SKIP) throw Fail("spliced (RNA) read: SKIP segments / exon pins are not supported on this path");
        rd.input.path.push_back(Seg{ in->path[i].type, in->path[i].length });
    }
    if (rd.input.empty() || path_read_length(rd.input.path) != unsigned(in->read_len))
        throw Fail("invalid alignment path associated with read segment"); // realignAndScoreRead :2036-2040
    for (int i = 0; i < in->n_observed; ++i)
        if (in->observed[i] < 0 || size_t(in->observed[i]) >= job.orig_to_tab.size()) throw Fail("observed indel index out of range");
    rd.realign_b = in->realign_begin;
    rd.realign_e = in->realign_end;
    // a read that does not pass the gate (most reads of a sample, away from candidate indels) is done: its bases, qualities and
    // observed indels are never looked at
    if (!passes_gate(job, rd.input, rd.read_len, Range(rd.realign_b, rd.realign_e))) {
        rd.pending.reset();
        rd.warn_origin = rd.warn_toggle = rd.incomplete_search = false;
        rd.cals.clear();
        return;
    }
    rd.code.assign(in->read_code, in->read_code + in->read_len);
    rd.qual.assign(in->read_qual, in->read_qual + in->read_len);
    rd.observed.resize(size_t(in->n_observed));
    for (int i = 0; i < in->n_observed; ++i) rd.observed[size_t(i)] = job.orig_to_tab[size_t(in->observed[i])];
    std::sort(rd.observed.begin(), rd.observed.end());
    rd.observed.erase(std::unique(rd.observed.begin(), rd.observed.end()), rd.observed.end());
    enumerate_read(job, rd, false);
    if (rd.pending) // the device scores these reads without passing the builder, whose check this is
        for (const uint8_t q : rd.qual)
            if (q > 70) // qphred_cache::high_qscore_error, L/blt_util/qscore_cache.cpp:66-75
                throw Fail("flatten: Attempting to lookup basecall quality score " + std::to_string(int(q)) +
                           " which exceeds the maximum cached basecall quality score of 70");
}

// flatten a prepared read's candidate alignments into `builder`; returns the number of candidate alignments added
int32_t flatten_read(const Job& j, sk_align_builder* builder, const sk_realign_job::Read& rd)
{
    std::vector<std::vector<sk_path_seg>> segs(rd.cals.size());
    std::vector<std::vector<sk_indel_key>> keys(rd.cals.size());
    std::vector<sk_candidate_alignment> cc(rd.cals.size());
    for (size_t i = 0; i < rd.cals.size(); ++i) cal_to_c(j, rd.cals[i], segs[i], keys[i], cc[i]);
    if (!rd.cals.empty() &&
        sk_align_builder_add_read(builder, rd.code.data(), rd.qual.data(), int32_t(rd.code.size()), j.ref.data(), j.ref_offset,
                                  int32_t(j.ref.size()), cc.data(), int32_t(cc.size())) != 0)
        throw Fail(std::string("flatten: ") + sk_align_builder_error(builder));
    return int32_t(rd.cals.size());
}

// flatten into the job's batch and keep the read (sequential: batch order = read order)
void append_read(Job& j, sk_realign_job::Read&& rd)
{
    rd.cal_begin = j.n_cals_total;
    if (j.opt.enumeration != 2) j.n_cals_total += flatten_read(j, j.builder, rd); // (2: the batch is assembled by the job's run)
    j.reads.push_back(std::move(rd));
}

int host_threads(const Job& j, const size_t n_items)
{
    int t = j.opt.host_threads;
    if (t <= 0) t = int(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
    return int(std::max<size_t>(1, std::min<size_t>(size_t(t), n_items / 64))); // not worth a thread below ~64 reads each
}

// run f(i) for i in [0, n) on up to `threads` threads (contiguous slices); returns the first exception's message, if any
template <typename F>
std::string parallel_for(const size_t n, const int threads, F&& f)
{
    const size_t nt = static_cast<size_t>(threads);
    std::vector<std::string> err(nt);
    std::vector<size_t> err_at(nt, n);
    auto work = [&](const int t) {
        const size_t b = n * size_t(t) / size_t(threads), e = n * size_t(t + 1) / size_t(threads);
        for (size_t i = b; i < e; ++i) {
            try {
                f(i);
            } catch (const std::exception& ex) {
                err[size_t(t)] = ex.what();
                err_at[size_t(t)] = i;
                return;
            }
        }
    };
    if (threads <= 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        std::vector<int> inline_slices{ 0 };
        for (int t = 1; t < threads; ++t) {
            try {
                pool.emplace_back(work, t);
            } catch (const std::system_error&) { // no more threads to be had: this slice runs here
                inline_slices.push_back(t);
            }
        }
        for (const int t : inline_slices) work(t);
        for (auto& th : pool) th.join();
    }
    size_t best = n;
    std::string msg;
    for (int t = 0; t < threads; ++t)
        if (err_at[size_t(t)] < best) {
            best = err_at[size_t(t)];
            msg = "read " + std::to_string(best) + ": " + err[size_t(t)];
        }
    return msg;
}

} // namespace

extern "C" {

int sk_realign_job_add_read(sk_realign_job* j, const sk_read_input* in)
{
    if (!j || !in || in->read_len < 0 || in->n_seg < 0) return -1;
    try {
        if (j->finished) throw Fail("job already finished: clear the reads first");
        sk_realign_job::Read rd;
        prepare_read(*j, in, rd);
        append_read(*j, std::move(rd));
        return int(j->reads.size()) - 1;
    } catch (const std::exception& e) {
        j->error = e.what();
        return -1;
    }
}

int sk_realign_job_add_reads(sk_realign_job* j, const sk_read_input* in, int32_t n)
{
    if (!j || (!in && n > 0) || n < 0) return -1;
    try {
        if (j->finished) throw Fail("job already finished: clear the reads first");
        // every thread prepares AND flattens its contiguous slice of the reads into a builder of its own; the slices are then
        // appended to the job's batch in order, so the batch is byte for byte what read-by-read calls produce
        // (the reads are prepared where they will stay; a rejected call takes them away again)
        const size_t first_read = j->reads.size();
        j->reads.resize(first_read + size_t(n));
        sk_realign_job::Read* const prepared = j->reads.data() + first_read;
        const int threads = host_threads(*j, size_t(n));
        struct Slice
        {
            sk_align_builder* builder = nullptr;
            ~Slice() { if (builder) sk_align_builder_destroy(builder); }
        };
        std::vector<Slice> slices(static_cast<size_t>(threads));
        const bool flatten_here = (j->opt.enumeration != 2); // (2: the batch is assembled by the job's run)
        if (flatten_here)
            for (auto& sl : slices) sl.builder = sk_align_builder_create();
        const std::string err = parallel_for(size_t(threads), threads, [&](const size_t t) {
            const size_t b = size_t(n) * t / size_t(threads), e = size_t(n) * (t + 1) / size_t(threads);
            for (size_t i = b; i < e; ++i) {
                try {
                    prepare_read(*j, in + i, prepared[i]);
                    prepared[i].cal_begin = flatten_here ? flatten_read(*j, slices[t].builder, prepared[i]) : 0; // (count for now)
                } catch (const std::exception& ex) {
                    throw Fail("read " + std::to_string(i) + ": " + ex.what());
                }
            }
        });
        if (!err.empty()) {
            j->reads.resize(first_read); // nothing was added
            throw Fail(err.substr(err.find(": ") + 2));
        }
        const int first = int(first_read);
        if (flatten_here)
            for (int t = 0; t < threads; ++t)
                if (sk_align_builder_append(j->builder, slices[size_t(t)].builder)) {
                    j->reads.resize(first_read);
                    throw Fail("append");
                }
        for (int32_t i = 0; i < n; ++i) {
            const int32_t n_cals = prepared[i].cal_begin;
            prepared[i].cal_begin = j->n_cals_total;
            j->n_cals_total += n_cals;
        }
        return first;
    } catch (const std::exception& e) {
        j->error = e.what();
        return -1;
    }
}

// flatten the job's batch; the prepare step inside gets the job's thread budget and its error text reaches the job
static int finish_builder(sk_realign_job* j, sk_align_batch* out)
{
    sk_align_builder_set_host_threads(j->builder, j->opt.host_threads);
    if (sk_align_builder_finish(j->builder, out)) {
        const char* e = sk_align_builder_error(j->builder);
        j->error = (e && *e) ? e : "sk_align_builder_finish failed";
        return 1;
    }
    return 0;
}

} // extern "C"

// ---- stage 3 in its container-free form (csrc/stage3_core.h; the device runs the same code, read_enumerate.hip) ----
struct Stage3Tables
{
    CoreTables core;
    std::vector<double> r2i, i2r;
    std::vector<int32_t> orig;
    std::vector<uint8_t> consulted;
    sk3::Tab tab;
    sk3::Opt opt;
};
static void build_stage3_tables(const sk_realign_job& j, Stage3Tables& t)
{
    build_core_tables(j, t.core, nullptr);
    const size_t n = j.tab.size();
    t.r2i.resize(n);
    t.i2r.resize(n);
    t.orig.resize(n);
    t.consulted.assign(n, 0);
    for (size_t i = 0; i < n; ++i) {
        t.r2i[i] = j.tab[i].r2i;
        t.i2r[i] = j.tab[i].i2r;
        t.orig[i] = j.tab[i].orig;
    }
    t.tab.tab = t.core.tab.data();
    t.tab.r2i = t.r2i.data();
    t.tab.i2r = t.i2r.data();
    t.tab.orig = t.orig.data();
    t.tab.n_tab = int32_t(n);
    t.tab.max_indel_size = int32_t(j.opt.max_indel_size);
    t.tab.consulted = nullptr;
    t.opt.is_smoothed_alignments = j.opt.is_smoothed_alignments;
    t.opt.smoothed_lnp_range = j.opt.smoothed_lnp_range;
    t.opt.upstream_oligo_size = j.opt.upstream_oligo_size;
    t.opt.min_read_bp_flank = j.opt.min_read_bp_flank;
    t.opt.no_tables = std::getenv("SK_STAGE3_NO_TABLES") ? 1 : 0;
}
// a read's stage-3 results as the core (host or device) produced them -> the read's structures
static void take_stage3_result(sk_realign_job::Read& rd, const bool fwd, const sk3::Out& o)
{
    rd.realigned = o.is_realigned != 0;
    rd.realignment = Aln();
    rd.realignment.pos = o.realign_pos;
    rd.realignment.fwd = fwd;
    for (int q = 0; q < o.n_seg; ++q) {
        rd.realignment.path.push_back(Seg{ o.path[q].type, o.path[q].length });
        rd.out_path.push_back(sk_path_seg{ o.path[q].type, o.path[q].length });
    }
    rd.max_score = o.max_score;
    rd.scores.assign(o.scores, o.scores + o.n_scores);
    rd.suboverlap.assign(o.sub, o.sub + o.n_sub);
}
// false: beyond a capacity of the core, or input the container-based code throws on -- that code decides
static bool finish_read_core(const sk_realign_job& j, const Stage3Tables& t, sk_realign_job::Read& rd, const double* scores)
{
    const size_t n = rd.cals.size();
    if (rd.code.size() > size_t(sk3::RL_MAX)) return false;
    std::vector<skcore::PCal> cals(n);
    for (size_t i = 0; i < n; ++i)
        if (!to_core_cal(rd.cals[i], cals[i])) return false;
    std::vector<int32_t> order(n), next_same(n), range_end(n), rm_pos(rd.code.size());
    std::vector<double> smooth(n), sorted_score(n);
    std::vector<uint32_t> key(4 * n), sorted_hash(n);
    std::vector<uint8_t> flag(n), removed(n), rm_type(rd.code.size()), consulted(j.tab.size(), 0);
    sk3::Tab tab = t.tab;
    tab.consulted = consulted.data();
    sk3::Read r;
    r.cals = sk3::CalView{ cals.data(), nullptr };
    r.scores = scores;
    r.scores_select = scores;
    r.n_cals = int32_t(n);
    r.map_level = rd.map_level;
    r.read_length = int32_t(rd.code.size());
    r.non_ambig = 0;
    for (const uint8_t c : rd.code)
        if (c != SK_BAM_ANY) ++r.non_ambig;
    sk3::Scratch w{ flag.data(), key.data(), order.data(), sorted_score.data(), smooth.data(), sorted_hash.data(), next_same.data(), range_end.data(), removed.data(), rm_type.data(), rm_pos.data() };
    sk3::Out o;
    sk3::Shared sh;
    sk3::finish_read(sk3::HostLanes(), tab, t.opt, r, w, sh, o);
    if (o.status != sk3::S3_OK) return false;
    for (size_t i = 0; i < consulted.size(); ++i)
        if (consulted[i]) (void)j.cand(int(i));
    take_stage3_result(rd, rd.cals[0].al.fwd, o);
    return true;
}

// enumeration == 2: hand every pending read to the device pipeline (csrc/read_enumerate.hip); afterwards every read of the job
// has its candidate alignments (and, with `want_scores`, the device reads their scores in j.dev_scores).  Throws Fail.
static void resolve_pending(sk_realign_job& j, const bool want_scores)
{
    std::vector<size_t> idx;
    for (size_t i = 0; i < j.reads.size(); ++i)
        if (j.reads[i].pending) idx.push_back(i);
    if (idx.empty()) return;
    CoreTables tables;
    build_core_tables(j, tables, nullptr);
    std::vector<skcore::PRead> preads(idx.size());
    std::vector<int64_t> read_off(idx.size() + 1, 0);
    std::vector<uint8_t> code, qual;
    int32_t max_read_len = 0;
    for (size_t k = 0; k < idx.size(); ++k) {
        const auto& rd = j.reads[idx[k]];
        preads[k] = *rd.pending;
        code.insert(code.end(), rd.code.begin(), rd.code.end());
        qual.insert(qual.end(), rd.qual.begin(), rd.qual.end());
        read_off[k + 1] = int64_t(code.size());
        max_read_len = std::max(max_read_len, int32_t(rd.code.size()));
    }
    SkEnumInput in;
    std::memset(&in, 0, sizeof(in));
    in.tab = tables.tab.data();
    in.n_tab = int32_t(tables.tab.size());
    in.ins_pool = tables.ins_pool.data();
    in.ins_pool_len = int64_t(tables.ins_pool.size());
    in.max_toggle = j.max_toggle.data();
    in.n_max_toggle = int32_t(j.max_toggle.size());
    in.sample_count = j.opt.sample_count;
    in.max_read_indel_toggle = j.opt.max_read_indel_toggle;
    in.is_haplotyping_enabled = j.opt.is_haplotyping_enabled;
    in.max_indel_size = int32_t(j.opt.max_indel_size);
    in.max_candidate_indel_density = j.opt.max_candidate_indel_density;
    in.ref = j.ref.data();
    in.ref_offset = j.ref_offset;
    in.ref_len = int32_t(j.ref.size());
    in.reads = preads.data();
    in.n_reads = int32_t(preads.size());
    in.read_off = read_off.data();
    in.read_code = code.data();
    in.read_qual = qual.data();
    in.max_read_len = max_read_len;
    in.want_scores = want_scores ? 1 : 0;
    Stage3Tables s3t;
    std::vector<int32_t> map_level(idx.size());
    if (want_scores) { // the whole read path: stage 3 on the device as well
        build_stage3_tables(j, s3t);
        for (size_t k = 0; k < idx.size(); ++k) map_level[k] = j.reads[idx[k]].map_level;
        in.want_stage3 = 1;
        in.r2i = s3t.r2i.data();
        in.i2r = s3t.i2r.data();
        in.orig = s3t.orig.data();
        in.map_level = map_level.data();
        in.stage3_opt = s3t.opt;
        if (const char* e = std::getenv("SK_STAGE3_DEVICE")) in.want_stage3 = std::atoi(e) ? 1 : 0;
    }
    SkEnumOutput out;
    const bool timing = std::getenv("SK_ENUM_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (sk_enum_device_run(&in, &out)) throw Fail(std::string("device enumeration: ") + sk_last_error());
    if (out.ref_reads_outside > 0) g_sk_ref_reads_outside.fetch_add(out.ref_reads_outside, std::memory_order_relaxed);
    const auto t1 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < j.tab.size(); ++i)
        if (out.consulted[i]) (void)j.cand(int(i));
    const int32_t n_cals = out.cal_off[idx.size()];
    j.dev_scores.clear();
    if (want_scores && n_cals > 0) {
        if (out.scores) j.dev_scores.assign(out.scores, out.scores + n_cals);
        else j.dev_scores.assign(size_t(n_cals), 0.0); // (the job ran as one sequence: a read's scores are fetched below if the host needs them)
    }
    j.dev_generation = out.generation;
    // reads the device enumerated but did not finish (a capacity of stage 3): their candidate alignments come over now
    std::vector<std::vector<skcore::PCal>> unfinished(idx.size());
    if (out.stage3 && !out.cals)
        for (size_t k = 0; k < idx.size(); ++k)
            if (out.status[k] == skcore::ST_OK && out.stage3[k].status != sk3::S3_OK) {
                unfinished[k].resize(size_t(out.cal_off[k + 1] - out.cal_off[k]));
                if (sk_enum_device_fetch_cals(out.generation, out.cal_off[k], int32_t(unfinished[k].size()), unfinished[k].data()))
                    throw Fail(std::string("device enumeration: candidate alignments not available: ") + sk_last_error());
                if (want_scores && !out.scores &&
                    sk_enum_device_fetch_scores(out.generation, out.cal_off[k], int32_t(unfinished[k].size()), j.dev_scores.data() + out.cal_off[k]))
                    throw Fail(std::string("device enumeration: scores not available: ") + sk_last_error());
            }
    // device results -> the reads' structures (reads are independent: each slice of the loop touches only its own reads)
    const std::string err = parallel_for(idx.size(), host_threads(j, idx.size()), [&](const size_t k) {
        auto& rd = j.reads[idx[k]];
        rd.pending.reset();
        rd.dev_score_at = -1;
        rd.dev_cal_at = -1;
        rd.stage3_done = false;
        if (out.status[k] == skcore::ST_OK) {
            const int32_t b = out.cal_off[k], e = out.cal_off[k + 1];
            if (b == e) throw Fail("Empty candidate alignment set while realigning normed input alignment");
            rd.cals.clear();
            rd.warn_origin = (out.warn[k] & 1) != 0;
            rd.warn_toggle = (out.warn[k] & 2) != 0;
            rd.incomplete_search = rd.warn_origin || rd.warn_toggle;
            if (want_scores) rd.dev_score_at = b;
            j.n_device_reads.fetch_add(1, std::memory_order_relaxed);
            if (out.stage3 && out.stage3[k].status == sk3::S3_OK) {
                // finished on the device: the candidate alignments stay in the device's form unless the host asks for them
                rd.dev_cal_at = b;
                rd.n_dev_cals = e - b;
                rd.scores.clear();
                rd.suboverlap.clear();
                rd.out_path.clear();
                take_stage3_result(rd, rd.input.fwd, out.stage3[k]); // (candidate alignments keep the read's strand)
                rd.stage3_done = true;
                j.n_stage3_device.fetch_add(1, std::memory_order_relaxed);
            } else {
                rd.cals.reserve(size_t(e - b));
                for (int32_t c = b; c < e; ++c) rd.cals.push_back(from_core_cal(out.cals ? out.cals[c] : unfinished[k][size_t(c - b)]));
            }
        } else {
            // beyond a capacity of the device form, or input the host code throws on: the container-based code decides
            const auto tf = std::chrono::steady_clock::now();
            enumerate_read(j, rd, true);
            j.n_fallback_reads.fetch_add(1, std::memory_order_relaxed);
            if (timing && std::getenv("SK_ENUM_TIMING_READS"))
                std::fprintf(stderr, "[enum] read %zu: device status %d, host search %.3f ms -> %zu candidate alignments\n", idx[k], out.status[k],
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf).count(), rd.cals.size());
        }
    });
    if (!err.empty()) throw Fail(err);
    if (timing)
        std::fprintf(stderr, "[enum] %zu reads %d cals: device pipeline %.2f ms, results -> host structures %.2f ms\n", idx.size(), n_cals,
                     std::chrono::duration<double, std::milli>(t1 - t0).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
}

// a read finished on the device: its candidate alignments into the host's structures
static void materialize_cals(const sk_realign_job& j, sk_realign_job::Read& rd)
{
    if (rd.dev_cal_at < 0 || !rd.cals.empty()) return;
    std::vector<skcore::PCal> cals(size_t(rd.n_dev_cals));
    if (sk_enum_device_fetch_cals(j.dev_generation, int32_t(rd.dev_cal_at), rd.n_dev_cals, cals.data()) == 0) {
        rd.cals.reserve(cals.size());
        for (const skcore::PCal& c : cals) rd.cals.push_back(from_core_cal(c));
    } else {
        enumerate_read(j, rd, true); // (the device has run another job since: the host lists them again, the same ones)
    }
}

// enumeration == 2: (re)build the job's host batch from the reads' candidate alignments
static void rebuild_host_batch(sk_realign_job* j)
{
    sk_align_builder_clear(j->builder);
    j->n_cals_total = 0;
    for (auto& rd : j->reads) {
        materialize_cals(*j, rd);
        rd.cal_begin = j->n_cals_total;
        j->n_cals_total += flatten_read(*j, j->builder, rd);
    }
}

extern "C" {

int sk_realign_job_get_batch(sk_realign_job* j, sk_align_batch* out)
{
    if (!j || !out) return 1;
    if (j->opt.enumeration == 2) {
        try {
            resolve_pending(*j, false);
            rebuild_host_batch(j);
        } catch (const std::exception& e) {
            j->error = e.what();
            return 1;
        }
    }
    return finish_builder(j, out);
}

} // extern "C"

// stage 3 for every read; score_of(read) -> the scores of its candidate alignments
template <typename ScoreOf>
static int finish_reads(sk_realign_job* j, ScoreOf&& score_of)
{
    try {
        std::unique_ptr<Stage3Tables> core_tables;
        if (j->opt.enumeration == 1) { // the container-free statement of stage 3, on this host
            core_tables.reset(new Stage3Tables);
            build_stage3_tables(*j, *core_tables);
        }
        // reads are independent in stage 3 as well: each writes only its own results
        const std::string err = parallel_for(j->reads.size(), host_threads(*j, j->reads.size()), [&](const size_t ri) {
            auto& rd = j->reads[ri];
            if (rd.stage3_done) return; // (the device finished this read)
            rd.scores.clear();
            rd.suboverlap.clear();
            rd.realigned = false;
            rd.out_path.clear();
            if (rd.cals.empty()) return;
            const double* s = score_of(rd);
            if (!s) throw Fail("sk_realign_job_finish: null scores");
            if (core_tables) {
                if (finish_read_core(*j, *core_tables, rd, s)) {
                    j->n_stage3_core.fetch_add(1, std::memory_order_relaxed);
                    return;
                }
                rd.out_path.clear();
            }
            const Cal* max_cal = nullptr;
            select_alignments(*j, rd, s, rd.max_score, max_cal);
            if (rd.map_level == SK_MAPLEVEL_TIER1 || rd.map_level == SK_MAPLEVEL_TIER2) // is_tier1or2_mapping :1800
                score_indels(*j, rd, s, rd.max_score, max_cal);
            for (const Seg& q : rd.realignment.path) rd.out_path.push_back(sk_path_seg{ q.type, q.length });
        });
        if (!err.empty()) throw Fail(err);
        j->finished = true;
        return 0;
    } catch (const std::exception& e) {
        j->error = e.what();
        return 1;
    }
}

extern "C" int sk_realign_job_finish(sk_realign_job* j, const double* scores)
{
    if (!j) return 1;
    for (auto& rd : j->reads) { // scores from the caller decide, whatever an earlier sk_realign_job_run left
        materialize_cals(*j, rd);
        rd.stage3_done = false;
    }
    return finish_reads(j, [&](const sk_realign_job::Read& rd) -> const double* { return scores ? scores + rd.cal_begin : nullptr; });
}

// enumeration == 2: search, flattening and scoring on the device; reads the device turned down go through the host stages and
// a host batch of their own
static int run_with_device_enumeration(sk_realign_job* j)
{
    std::vector<double> host_scores;
    const bool timing = std::getenv("SK_ENUM_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [](const std::chrono::steady_clock::time_point& t) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    };
    try {
        resolve_pending(*j, true);
        if (timing) std::fprintf(stderr, "[enum] resolve_pending %.2f ms\n", ms_since(t0));
        struct Tmp
        {
            sk_align_builder* b = sk_align_builder_create();
            ~Tmp() { sk_align_builder_destroy(b); }
        } tmp;
        int32_t n = 0;
        for (auto& rd : j->reads) {
            if (rd.dev_score_at >= 0 || rd.cals.empty()) continue;
            rd.cal_begin = n;
            n += flatten_read(*j, tmp.b, rd);
        }
        if (n > 0) {
            sk_align_batch b;
            sk_align_builder_set_host_threads(tmp.b, j->opt.host_threads);
            if (sk_align_builder_finish(tmp.b, &b)) throw Fail(std::string("flatten: ") + sk_align_builder_error(tmp.b));
            host_scores.resize(size_t(b.n_cals));
            if (sk_score_alignments(&b, host_scores.data()) != 0) throw Fail(sk_last_error());
        }
    } catch (const std::exception& e) {
        j->error = e.what();
        return 1;
    }
    const auto t1 = std::chrono::steady_clock::now();
    const int rc = finish_reads(j, [&](const sk_realign_job::Read& rd) -> const double* {
        return rd.dev_score_at >= 0 ? j->dev_scores.data() + rd.dev_score_at : host_scores.data() + rd.cal_begin;
    });
    if (timing) std::fprintf(stderr, "[enum] host-enumerated reads %.2f ms, stage 3 %.2f ms\n", ms_since(t0) - ms_since(t1), ms_since(t1));
    return rc;
}

extern "C" {

int sk_realign_job_run(sk_realign_job* j)
{
    if (!j) return 1;
    if (j->opt.enumeration == 2) return run_with_device_enumeration(j);
    sk_align_batch b;
    if (finish_builder(j, &b)) return 1;
    std::vector<double> scores(size_t(b.n_cals));
    if (b.n_cals > 0 && sk_score_alignments(&b, scores.data()) != 0) {
        j->error = sk_last_error();
        return 1;
    }
    return sk_realign_job_finish(j, scores.data());
}

int sk_realign_job_read_result(const sk_realign_job* j, int32_t i, sk_read_result* out)
{
    if (!j || !out || i < 0 || size_t(i) >= j->reads.size()) return 1;
    const auto& rd = j->reads[size_t(i)];
    std::memset(out, 0, sizeof(*out));
    out->n_candidate_alignments = rd.cals.empty() ? (rd.dev_cal_at >= 0 ? rd.n_dev_cals : 0) : int32_t(rd.cals.size());
    out->is_realigned = rd.realigned ? 1 : 0;
    out->realign_pos = rd.realignment.pos;
    out->realign_n_seg = int32_t(rd.out_path.size());
    out->realign_path = rd.out_path.data();
    out->max_score = rd.max_score;
    out->n_scores = int32_t(rd.scores.size());
    out->scores = rd.scores.data();
    out->n_suboverlap = int32_t(rd.suboverlap.size());
    out->suboverlap = rd.suboverlap.data();
    out->warn_origin_skip = rd.warn_origin ? 1 : 0;
    out->warn_max_toggle_depth = rd.warn_toggle ? 1 : 0;
    return 0;
}

int sk_make_start_pos_alignment(int32_t ref_start_pos, int32_t read_start_pos, int32_t is_fwd, uint32_t read_length,
                                const sk_indel_key* indels, int32_t n, int32_t* out_pos, sk_path_seg* out_path,
                                int32_t path_cap, int32_t* out_lead, int32_t* out_trail)
{
    try {
        std::vector<Indel> tab(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) { tab[size_t(i)].key = key_from_c(indels[i]); tab[size_t(i)].orig = i; }
        std::stable_sort(tab.begin(), tab.end(), [](const Indel& a, const Indel& b) { return a.key < b.key; });
        ISet all;
        for (int i = 0; i < n; ++i) all.push_back(i);
        const Cal c = make_start_pos_alignment(tab, ref_start_pos, read_start_pos, is_fwd != 0, read_length, all);
        if (int32_t(c.al.path.size()) > path_cap) return -1;
        *out_pos = c.al.pos;
        for (size_t i = 0; i < c.al.path.size(); ++i) out_path[i] = sk_path_seg{ c.al.path[i].type, c.al.path[i].length };
        *out_lead = c.lead < 0 ? -1 : tab[size_t(c.lead)].orig;
        *out_trail = c.trail < 0 ? -1 : tab[size_t(c.trail)].orig;
        return int(c.al.path.size());
    } catch (...) {
        return -1;
    }
}

int sk_get_end_pin_start_pos(const sk_indel_key* indels, int32_t n, uint32_t read_length, int32_t ref_end_pos,
                             int32_t read_end_pos, int32_t* out_ref_start, int32_t* out_read_start)
{
    try {
        std::vector<Indel> tab(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) tab[size_t(i)].key = key_from_c(indels[i]);
        std::stable_sort(tab.begin(), tab.end(), [](const Indel& a, const Indel& b) { return a.key < b.key; });
        ISet all;
        for (int i = 0; i < n; ++i) all.push_back(i);
        get_end_pin_start_pos(tab, all, read_length, ref_end_pos, read_end_pos, *out_ref_start, *out_read_start);
        return 0;
    } catch (...) {
        return 1;
    }
}

} // extern "C"
