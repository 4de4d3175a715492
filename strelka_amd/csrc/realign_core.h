// realign_core.h -- candidate-alignment enumeration as container-free code that compiles for the
// device and for the host alike.
//
//   candidate_alignment_search      L/starling_common/starling_read_align.cpp:857-1277
//   make_start_pos_alignment        :394-584         get_end_pin_start_pos       :594-719
//   add_indels_in_range             :311-366         sort_remove_only_indels_last :724-751
//   addKeysToCandidateAlignment     :789-808         getCurIndelHaplotypeIds     :811-852
//   HaplotypeStatus / getUpdatedSampleHaplotypeConstraints  :63-179
//
// The reference recurses, passing its indel-status map, haplotype map and indel order BY VALUE at every level, and collects
// the leaves in a std::set<CandidateAlignment>.  Here the recursion is an explicit stack of fixed-size frames (one per level,
// a frame is the by-value state of that call), every container is a small array bounded by `Caps`, and leaves are handed to
// a sink; ordering and de-duplication of the leaves (the std::set) is the caller's sort by `cal_less`.  The host stages in
// host/read_realign.cpp hold the container-based statement of the same functions, which is what is pinned to the reference
// (golden fixtures, live fuzzing); tests/test_device_enumeration.py requires this core to produce the same candidate
// alignments, on the host and on the device, for every read within the caps.  A read beyond a cap is reported (status
// OVERFLOW) and enumerated by the container-based code instead -- never truncated.
#pragma once

#include "strelka_amd.h"

#include <stdint.h>

#if defined(__HIP__)
#define SKC_HD __host__ __device__
#else
#define SKC_HD
#endif

namespace skcore
{

struct Caps
{
    enum {
        K = 40,   // indels in one read's status map
        P = 48,   // path segments of one alignment
        H = 6,    // active regions touched by one read
        OBS = 64  // observed (non-candidate but usable) indels of one read
    };
};

enum { ST_OK = 0, ST_OVERFLOW = 1, ST_FAIL = 2 };

struct PSeg
{
    uint16_t type, length;
};

struct PIndel // the fields of an IndelBuffer entry the search reads
{
    int32_t pos;
    uint32_t del;
    uint32_t ins_len;
    int32_t arid;
    uint8_t type, cand, forced, ndfr;
    int8_t hap[SK_MAX_SAMPLES];
    uint8_t bypass[SK_MAX_SAMPLES];
    uint32_t ins_off; // insert sequence in the job's character pool (flattening only)
    uint32_t shape;   // number of the indel's (type, deletion length, insert sequence) among the table's: stage 3's equivalence test
};

SKC_HD inline int32_t right_pos(const PIndel& k) { return k.pos + int32_t(k.del); }
SKC_HD inline bool is_mismatch(const PIndel& k) { return k.type == SK_INDEL_MISMATCH; }
SKC_HD inline bool primitive_del(const PIndel& k) { return k.type == SK_INDEL_INDEL && k.ins_len == 0 && k.del > 0; }

struct PRange
{
    int32_t b, e;
    uint8_t has_b, has_e;
};
SKC_HD inline PRange mk_range(int32_t b, int32_t e)
{
    PRange r;
    r.b = b;
    r.e = e;
    r.has_b = r.has_e = 1;
    return r;
}
SKC_HD inline bool pos_intersect(const PRange& r, int32_t p) { return (!r.has_b || p >= r.b) && (!r.has_e || p < r.e); }
SKC_HD inline bool range_intersect(const PRange& a, const PRange& o)
{
    return (!o.has_e || !a.has_b || o.e > a.b) && (!o.has_b || !a.has_e || o.b < a.e);
}
SKC_HD inline bool superset_of(const PRange& a, const PRange& o)
{
    return (!a.has_e || (o.has_e && o.e <= a.e)) && (!a.has_b || (o.has_b && o.b >= a.b));
}
SKC_HD inline PRange open_pos_range(const PIndel& k) // IndelKey.hh:119-135
{
    PRange r = mk_range(k.pos, right_pos(k));
    if (k.type == SK_INDEL_BP_LEFT) {
        r.has_e = 0;
        r.e = 0;
    } else if (k.type == SK_INDEL_BP_RIGHT) {
        r.has_b = 0;
        r.b = 0;
        r.e = k.pos;
    }
    return r;
}
SKC_HD inline bool is_indel_conflict(const PIndel& a, const PIndel& b) // indel_util.cpp:25-42
{
    const bool mm = is_mismatch(a) || is_mismatch(b);
    PRange r1 = open_pos_range(a), r2 = open_pos_range(b);
    if (!mm) {
        r1.e++;
        r2.e++;
    }
    return range_intersect(r1, r2);
}
SKC_HD inline bool range_intersect_indel_breakpoints(const PRange& pr, const PIndel& k) // :45-60
{
    if (is_mismatch(k)) return pos_intersect(pr, k.pos);
    if (range_intersect(pr, mk_range(k.pos, k.pos))) return true;
    const int32_t rp = right_pos(k);
    if (k.pos == rp) return false;
    return range_intersect(pr, mk_range(rp, rp));
}
SKC_HD inline bool range_adjacent_indel_breakpoints(const PRange& pr, const PIndel& k) // :64-73
{
    if (range_intersect(pr, mk_range(k.pos - 1, k.pos + 1))) return true;
    const int32_t rp = right_pos(k);
    if (k.pos == rp) return false;
    return range_intersect(pr, mk_range(rp - 1, rp + 1));
}

SKC_HD inline bool seg_align_match(unsigned t) { return t == SK_SEG_MATCH || t == SK_SEG_SEQ_MATCH || t == SK_SEG_SEQ_MISMATCH; }
SKC_HD inline bool seg_read_len(unsigned t) { return seg_align_match(t) || t == SK_SEG_INSERT || t == SK_SEG_SOFT_CLIP; }
SKC_HD inline bool seg_ref_len(unsigned t) { return seg_align_match(t) || t == SK_SEG_DELETE || t == SK_SEG_SKIP; }
SKC_HD inline bool seg_unaligned_edge(unsigned t) { return t == SK_SEG_INSERT || t == SK_SEG_HARD_CLIP || t == SK_SEG_SOFT_CLIP; }

struct PCal // CandidateAlignment: alignment + edge indels (+ the indel set once it is a leaf)
{
    int32_t pos;
    int16_t lead, trail; // table index of the leading / trailing edge indel, -1 = none
    uint8_t fwd, n_seg, n_indels, pad;
    PSeg path[Caps::P];
    int16_t indels[Caps::K + 2]; // ascending table indices (leaves only)
};

SKC_HD inline unsigned path_ref_length(const PCal& c)
{
    unsigned v = 0;
    for (int i = 0; i < c.n_seg; ++i)
        if (seg_ref_len(c.path[i].type)) v += c.path[i].length;
    return v;
}
SKC_HD inline unsigned unaligned_prefix(const PCal& c) // align_path.cpp:190-200
{
    unsigned v = 0;
    for (int i = 0; i < c.n_seg; ++i) {
        if (!seg_unaligned_edge(c.path[i].type)) return v;
        if (seg_read_len(c.path[i].type)) v += c.path[i].length;
    }
    return v;
}
SKC_HD inline unsigned unaligned_suffix(const PCal& c) // :204-214
{
    unsigned v = 0;
    for (int i = int(c.n_seg) - 1; i >= 0; --i) {
        if (!seg_unaligned_edge(c.path[i].type)) return v;
        if (seg_read_len(c.path[i].type)) v += c.path[i].length;
    }
    return v;
}
SKC_HD inline unsigned insert_lead(const PCal& c) // apath_insert_lead_size :295-315
{
    unsigned v = 0;
    for (int i = 0; i < c.n_seg; ++i) {
        const unsigned t = c.path[i].type;
        if (t == SK_SEG_HARD_CLIP || t == SK_SEG_SOFT_CLIP) continue;
        if (t == SK_SEG_INSERT) v += c.path[i].length;
        else break;
    }
    return v;
}
SKC_HD inline unsigned insert_trail(const PCal& c) // :319-339
{
    unsigned v = 0;
    for (int i = int(c.n_seg) - 1; i >= 0; --i) {
        const unsigned t = c.path[i].type;
        if (t == SK_SEG_HARD_CLIP || t == SK_SEG_SOFT_CLIP) continue;
        if (t == SK_SEG_INSERT) v += c.path[i].length;
        else break;
    }
    return v;
}
SKC_HD inline PRange strict_range(const PCal& c) { return mk_range(c.pos, c.pos + int32_t(path_ref_length(c))); }
SKC_HD inline PRange soft_clip_range(const PCal& c)
{
    return mk_range(c.pos - int32_t(insert_lead(c)), c.pos + int32_t(path_ref_length(c)) + int32_t(insert_trail(c)));
}

// the std::set<CandidateAlignment> order: alignment::operator< (alignment.hh:72-90: pos, strand, path size, segments), then the
// indel set, then the edge keys (CandidateAlignment.hh:37-48); table indices stand for IndelKeys (the table is in key order)
SKC_HD inline int cal_compare(const PCal& a, const PCal& b)
{
    if (a.pos != b.pos) return a.pos < b.pos ? -1 : 1;
    if (a.fwd != b.fwd) return a.fwd < b.fwd ? -1 : 1;
    if (a.n_seg != b.n_seg) return a.n_seg < b.n_seg ? -1 : 1;
    for (int i = 0; i < a.n_seg; ++i) {
        if (a.path[i].type != b.path[i].type) return a.path[i].type < b.path[i].type ? -1 : 1;
        if (a.path[i].length != b.path[i].length) return a.path[i].length < b.path[i].length ? -1 : 1;
    }
    const int n = a.n_indels < b.n_indels ? a.n_indels : b.n_indels;
    for (int i = 0; i < n; ++i)
        if (a.indels[i] != b.indels[i]) return a.indels[i] < b.indels[i] ? -1 : 1;
    if (a.n_indels != b.n_indels) return a.n_indels < b.n_indels ? -1 : 1;
    if (a.lead != b.lead) return a.lead < b.lead ? -1 : 1;
    if (a.trail != b.trail) return a.trail < b.trail ? -1 : 1;
    return 0;
}

struct PStatus // one entry of starling_align_indel_status (:48-53, :189)
{
    int16_t idx; // table index; entries are kept in ascending order (= IndelKey order, the map's order)
    uint8_t is_present, is_remove_only, in_original, pad;
};

struct PHap // HaplotypeStatus :134-179
{
    int32_t arid;
    int8_t hc[SK_MAX_SAMPLES];
    uint8_t any_on, pad[3];
};

struct PFrame // the by-value state of one call of candidate_alignment_search
{
    PStatus sm[Caps::K];
    int16_t order[Caps::K]; // table indices
    PHap hm[Caps::H];
    PHap nhm[Caps::H];      // the haplotype map handed to the toggled children (alignments 2 and 3)
    uint8_t n_sm, n_order, n_hm, n_nhm;
    uint16_t depth, indel_toggle_depth, total_toggle_depth, stage; // stage: which child comes next
    int32_t max_read_indel_toggle;
    int32_t read_id; // (level-synchronous expansion on the device: which read the frame belongs to)
    PRange read_range;
    PCal cal;
    int16_t cur; // order[depth] of this call
    uint8_t cur_on, toggle_inc;
    int16_t current[Caps::K]; // the present indels after the toggle (ascending), for the two pinned children
    uint8_t n_current, pad[3];
};

struct PRead // what the host preamble (getCandidateAlignments :1816-1957) hands to the search
{
    int32_t realign_b, realign_e;
    int32_t read_length; // of the clipped read the search works on
    int32_t sample;
    PRange exemplar_range;
    PCal cal;
    PStatus sm[Caps::K];
    int16_t order[Caps::K];
    int16_t observed[Caps::OBS]; // ascending table indices
    uint8_t n_sm, n_order, n_observed, clipped;
    uint32_t hc_lead, hc_trail, sc_lead, sc_trail; // clip_adder (:508-542) arguments
};

struct PJob
{
    const PIndel* tab;
    int32_t n_tab;
    const uint32_t* max_toggle; // starling_align_limit (:77-88)
    int32_t n_max_toggle;
    int32_t sample_count;
    int32_t max_read_indel_toggle;
    double max_candidate_indel_density;
    int32_t is_haplotyping_enabled;
    int32_t max_indel_size;
    uint8_t* consulted; // [n_tab] or null: candidate status consulted (see sk_realign_job_indels_consulted)
    int32_t max_nodes;  // 0 = no limit; else a search longer than this many calls ends with ST_OVERFLOW (the device hands such
                        // reads to the host instead of keeping one lane busy for long)
};

SKC_HD inline bool job_cand(const PJob& j, int i)
{
    if (j.consulted) j.consulted[i] = 1;
    return j.tab[i].cand != 0;
}

// IndelBuffer::rangeIterator (IndelBuffer.cpp:76-92): from the first key at or after begin_pos - max_indel_size whose right
// end reaches begin_pos, up to the first key at or after end_pos
SKC_HD inline void range_iter(const PJob& j, int32_t begin_pos, int32_t end_pos, int& lo, int& hi)
{
    int a = 0, b = j.n_tab;
    while (a < b) {
        const int m = (a + b) >> 1;
        if (j.tab[m].pos < end_pos) a = m + 1; else b = m;
    }
    hi = a;
    const int32_t from = begin_pos - j.max_indel_size;
    a = 0;
    b = j.n_tab;
    while (a < b) {
        const int m = (a + b) >> 1;
        if (j.tab[m].pos < from) a = m + 1; else b = m;
    }
    for (; a < hi; ++a)
        if (right_pos(j.tab[a]) >= begin_pos) break;
    if (a > hi) a = hi;
    lo = a;
}

struct SearchOut
{
    int status;
    int warn_origin, warn_toggle;
    int nodes; // calls of candidate_alignment_search made
};

// ---- status-map helpers (entries ascending by table index) ----
template <typename F>
SKC_HD inline int sm_find(const F& f, int idx)
{
    int a = 0, b = f.n_sm;
    while (a < b) {
        const int m = (a + b) >> 1;
        if (f.sm[m].idx < idx) a = m + 1; else b = m;
    }
    return (a < f.n_sm && f.sm[a].idx == idx) ? a : -1;
}
template <typename F>
SKC_HD inline bool sm_insert(F& f, int idx, bool remove_only)
{
    if (f.n_sm >= Caps::K) return false;
    int a = 0;
    while (a < f.n_sm && f.sm[a].idx < idx) ++a;
    for (int i = f.n_sm; i > a; --i) f.sm[i] = f.sm[i - 1];
    PStatus s;
    s.idx = int16_t(idx);
    s.is_present = 0;
    s.is_remove_only = remove_only ? 1 : 0;
    s.in_original = 0;
    s.pad = 0;
    f.sm[a] = s;
    f.n_sm++;
    return true;
}

SKC_HD inline bool is_observed(const PRead& r, int idx)
{
    int a = 0, b = r.n_observed;
    while (a < b) {
        const int m = (a + b) >> 1;
        if (r.observed[m] < idx) a = m + 1; else b = m;
    }
    return a < r.n_observed && r.observed[a] == idx;
}

// add_indels_in_range :311-366; false = a cap was hit
template <typename F>
SKC_HD inline bool add_indels_in_range(const PJob& j, const PRead& r, const PRange& pr, F& f)
{
    int lo, hi;
    range_iter(j, pr.b, pr.e, lo, hi);
    for (int i = lo; i < hi; ++i) {
        const PIndel& k = j.tab[i];
        if (!range_adjacent_indel_breakpoints(pr, k)) continue;
        const bool remove_only = !range_intersect_indel_breakpoints(pr, k);
        const int at = sm_find(f, i);
        if (at >= 0) {
            if (!remove_only && f.sm[at].is_remove_only) f.sm[at].is_remove_only = 0;
        } else if (job_cand(j, i) || is_observed(r, i)) { // is_usable_indel :289-305
            if (!sm_insert(f, i, remove_only) || f.n_order >= Caps::K) return false;
            f.order[f.n_order++] = int16_t(i);
        }
    }
    return true;
}

// sort_remove_only_indels_last :724-751
template <typename F>
SKC_HD inline void sort_remove_only_indels_last(F& f, unsigned current_depth)
{
    int16_t* o2 = f.current; // free at this point of the call: `current` is only live between the frame's stages 1 and 2
    int n = 0;
    for (unsigned i = 0; i < current_depth; ++i) o2[n++] = f.order[i];
    for (unsigned i = current_depth; i < f.n_order; ++i) {
        const PStatus& s = f.sm[sm_find(f, f.order[i])];
        if (s.is_present || !s.is_remove_only) o2[n++] = f.order[i];
    }
    for (unsigned i = current_depth; i < f.n_order; ++i) {
        const PStatus& s = f.sm[sm_find(f, f.order[i])];
        if (!(s.is_present || !s.is_remove_only)) o2[n++] = f.order[i];
    }
    for (int i = 0; i < n; ++i) f.order[i] = o2[i];
}

// getUpdatedSampleHaplotypeConstraints :63-130
SKC_HD inline int updated_haplotype_constraints(int hc, int cur_hap_id, bool cur_on, bool any_on)
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
SKC_HD inline bool hap_update(PHap& h, int n_samples, const int* cur_hap_ids, bool cur_on)
{
    h.any_on = (h.any_on || cur_on) ? 1 : 0;
    bool valid = false;
    for (int s = 0; s < n_samples; ++s) {
        const int u = updated_haplotype_constraints(h.hc[s], cur_hap_ids[s], cur_on, h.any_on != 0);
        if (u >= 0) valid = true;
        h.hc[s] = int8_t(u);
    }
    return valid;
}
SKC_HD inline int hm_find(const PHap* hm, int n, int32_t arid)
{
    for (int i = 0; i < n; ++i)
        if (hm[i].arid == arid) return i;
    return -1;
}

// getCurIndelHaplotypeIds :811-852
SKC_HD inline void cur_indel_haplotype_ids(const PJob& j, int cur_sample, int idx, bool in_original, int* ids)
{
    const PIndel& d = j.tab[idx];
    const int n = j.sample_count;
    for (int s = 0; s < n; ++s) ids[s] = 0;
    if (d.arid < 0) return;
    for (int s = 0; s < n; ++s) {
        int h = d.hap[s];
        if (h == 0) {
            bool valid = (!j.is_haplotyping_enabled) || d.bypass[s] || d.forced;
            if (!valid && s == cur_sample && in_original) valid = true;
            if (is_mismatch(d) && s != cur_sample) valid = false;
            h = valid ? 0 : -1;
        }
        ids[s] = h;
    }
}

SKC_HD inline bool push_seg(PCal& c, unsigned type, unsigned length)
{
    if (c.n_seg >= Caps::P || length > 0xffffu) return false;
    c.path[c.n_seg].type = uint16_t(type);
    c.path[c.n_seg].length = uint16_t(length);
    c.n_seg++;
    return true;
}

// make_start_pos_alignment :394-584.  `indels`: ascending table indices.  Returns ST_OK / ST_OVERFLOW / ST_FAIL.
SKC_HD inline int make_start_pos_alignment(const PJob& j, int32_t ref_start_pos, int32_t read_start_pos, bool fwd,
                                           unsigned read_length, const int16_t* indels, int n_indels, PCal& cal)
{
    const bool is_leading_read = (read_start_pos != 0);
    cal.pos = ref_start_pos;
    cal.fwd = fwd ? 1 : 0;
    cal.n_seg = 0;
    cal.n_indels = 0;
    cal.lead = cal.trail = -1;
    cal.pad = 0;
    int32_t ref_head = ref_start_pos, read_head = read_start_pos;
    bool prev_mm = false;
    for (int q = 0; q < n_indels; ++q) {
        const int ii = indels[q];
        const PIndel& k = j.tab[ii];
        const bool mm = is_mismatch(k);
        if (right_pos(k) < ref_start_pos) continue;
        if (right_pos(k) == ref_start_pos) {
            if (mm) continue;
            if (!is_leading_read) continue;
        }
        const bool first = (cal.n_seg == 0);
        if (is_leading_read && first) {
            if (k.pos != ref_start_pos) return ST_FAIL;
            if (!push_seg(cal, SK_SEG_INSERT, unsigned(read_start_pos))) return ST_OVERFLOW;
            if (k.del > 0) {
                if (!push_seg(cal, SK_SEG_DELETE, k.del)) return ST_OVERFLOW;
                ref_head += int32_t(k.del);
            }
            cal.lead = int16_t(ii);
            prev_mm = mm;
            continue;
        }
        const bool edge_delete = primitive_del(k) && (k.pos == ref_start_pos);
        const int match_size = k.pos - ref_head;
        const int min_match = (prev_mm || mm) ? 0 : 1;
        if (match_size < min_match && !edge_delete) return ST_FAIL;
        const unsigned match_segment = unsigned(match_size);
        if ((read_head + int32_t(match_segment)) > int32_t(read_length) ||
            ((read_head + int32_t(match_segment)) == int32_t(read_length) && !primitive_del(k)))
            break;
        if (match_segment > 0) {
            if (!push_seg(cal, SK_SEG_MATCH, match_segment)) return ST_OVERFLOW;
            ref_head += int32_t(match_segment);
            read_head += int32_t(match_segment);
        }
        if (mm) {
            if (!push_seg(cal, SK_SEG_SEQ_MISMATCH, k.del)) return ST_OVERFLOW;
            ref_head += int32_t(k.del);
            read_head += int32_t(k.del);
            if (read_head >= int32_t(read_length)) break;
        } else if (k.type == SK_INDEL_INDEL) {
            if (k.del > 0) {
                if (!push_seg(cal, SK_SEG_DELETE, k.del)) return ST_OVERFLOW;
                ref_head += int32_t(k.del);
            }
            if (k.ins_len > 0) {
                const unsigned max_ins = read_length - unsigned(read_head);
                const unsigned ins = k.ins_len < max_ins ? k.ins_len : max_ins;
                if (!push_seg(cal, SK_SEG_INSERT, ins)) return ST_OVERFLOW;
                read_head += int32_t(ins);
                if (k.ins_len >= max_ins) {
                    cal.trail = int16_t(ii);
                    break;
                }
            } else {
                if (match_segment == 0) cal.lead = int16_t(ii);
                else if (read_head == int32_t(read_length)) cal.trail = int16_t(ii);
            }
        } else if (k.type == SK_INDEL_BP_LEFT) {
            const unsigned overhang = read_length - unsigned(read_head);
            if (!push_seg(cal, SK_SEG_INSERT, overhang)) return ST_OVERFLOW;
            read_head += int32_t(overhang);
            cal.trail = int16_t(ii);
            break;
        } else {
            return ST_FAIL;
        }
        prev_mm = mm;
    }
    if (read_head < int32_t(read_length))
        if (!push_seg(cal, SK_SEG_MATCH, read_length - unsigned(read_head))) return ST_OVERFLOW;
    return ST_OK;
}

// get_end_pin_start_pos :594-719
SKC_HD inline int get_end_pin_start_pos(const PJob& j, const int16_t* indels, int n_indels, unsigned read_length,
                                        int32_t ref_end_pos, int32_t read_end_pos, int32_t& ref_start_pos, int32_t& read_start_pos)
{
    ref_start_pos = ref_end_pos;
    read_start_pos = read_end_pos;
    const bool is_trailing_read = (read_end_pos != int32_t(read_length));
    bool is_first = true, prev_mm = false;
    for (int q = n_indels - 1; q >= 0; --q) {
        const PIndel& k = j.tab[indels[q]];
        const bool mm = is_mismatch(k);
        if (k.pos > ref_end_pos) continue;
        if (k.pos == ref_end_pos) {
            if (mm) continue;
            if (!is_trailing_read) continue;
        }
        const bool trailing_indel = (!mm) && (right_pos(k) == ref_end_pos);
        if (trailing_indel) {
            if (k.type == SK_INDEL_INDEL) ref_start_pos -= int32_t(k.del);
        } else {
            if (is_first && read_end_pos != int32_t(read_length)) return ST_FAIL;
            const int match_size = int(ref_start_pos - right_pos(k));
            const int min_match = (prev_mm || mm) ? 0 : 1;
            if (match_size < min_match) return ST_FAIL;
            const unsigned match_segment = unsigned(match_size < int(read_start_pos) ? match_size : int(read_start_pos));
            ref_start_pos -= int32_t(match_segment);
            read_start_pos -= int32_t(match_segment);
            if (read_start_pos == 0) return ST_OK;
            if (k.type == SK_INDEL_INDEL) {
                ref_start_pos -= int32_t(k.del);
                if (k.ins_len > 0) {
                    if (int32_t(k.ins_len) >= read_start_pos) return ST_OK;
                    read_start_pos -= int32_t(k.ins_len);
                }
            } else if (mm) {
                ref_start_pos -= int32_t(k.del);
                read_start_pos -= int32_t(k.del);
                if (read_start_pos == 0) return ST_OK;
            } else if (k.type == SK_INDEL_BP_RIGHT) {
                return ST_OK;
            } else {
                return ST_FAIL;
            }
        }
        is_first = false;
        prev_mm = mm;
    }
    ref_start_pos -= read_start_pos;
    read_start_pos = 0;
    return ST_OK;
}

// addKeysToCandidateAlignment :789-808: the present indels that intersect the alignment (ascending), then the two edge keys
// inserted in order
SKC_HD inline void iset_insert(int16_t* s, int& n, int16_t v)
{
    int a = 0;
    while (a < n && s[a] < v) ++a;
    if (a < n && s[a] == v) return;
    for (int i = n; i > a; --i) s[i] = s[i - 1];
    s[a] = v;
    ++n;
}
template <typename F>
SKC_HD inline void add_keys_to_cal(const PJob& j, const F& f, PCal& cal)
{
    const PRange pr = strict_range(cal);
    int n = 0;
    for (int i = 0; i < f.n_sm; ++i) {
        if (!f.sm[i].is_present) continue;
        if (!range_intersect_indel_breakpoints(pr, j.tab[f.sm[i].idx])) continue;
        cal.indels[n++] = f.sm[i].idx;
    }
    if (cal.lead >= 0) iset_insert(cal.indels, n, cal.lead);
    if (cal.trail >= 0) iset_insert(cal.indels, n, cal.trail);
    cal.n_indels = uint8_t(n);
}

SKC_HD inline unsigned get_max_toggle(const PJob& j, unsigned n_indels) // starling_align_limit.hh:40-51
{
    return (n_indels >= unsigned(j.n_max_toggle)) ? 1u : j.max_toggle[n_indels];
}

// the live part of a candidate alignment (the arrays' tails are not read)
SKC_HD inline void copy_cal(PCal& d, const PCal& q)
{
    d.pos = q.pos;
    d.lead = q.lead;
    d.trail = q.trail;
    d.fwd = q.fwd;
    d.n_seg = q.n_seg;
    d.n_indels = q.n_indels;
    d.pad = 0;
    for (int i = 0; i < q.n_seg; ++i) d.path[i] = q.path[i];
    for (int i = 0; i < q.n_indels; ++i) d.indels[i] = q.indels[i];
}

// The frame of the first call (getCandidateAlignments :1957)
SKC_HD inline void root_frame(const PJob& j, const PRead& r, PFrame& f)
{
    for (int i = 0; i < r.n_sm; ++i) f.sm[i] = r.sm[i];
    f.n_sm = r.n_sm;
    for (int i = 0; i < r.n_order; ++i) f.order[i] = r.order[i];
    f.n_order = r.n_order;
    f.n_hm = 0;
    f.n_nhm = 0;
    f.depth = f.indel_toggle_depth = f.total_toggle_depth = 0;
    f.stage = 0;
    f.max_read_indel_toggle = j.max_read_indel_toggle;
    f.read_range = r.exemplar_range;
    f.read_id = 0;
    copy_cal(f.cal, r.cal);
}

// ONE call of candidate_alignment_search (:857-1277) up to its recursive calls: `f` is the call's by-value state (it is modified,
// as the reference modifies its by-value arguments); a leaf goes to `sink(PCal&)` (it may modify the leaf, and returns false when
// it cannot take more), and every recursive call the reference would make becomes a child frame -- `alloc()` hands out the slot
// (null = none left) -- that the caller expands in turn, in any order: the calls share nothing but the set the leaves go into
// and the two warning flags.  `tmp`: one PCal of scratch.  Anything but ST_OK in out.status ends the read's search.
template <typename Alloc, typename Sink>
SKC_HD inline void expand_node(const PJob& j, const PRead& r, PFrame& f, Alloc& alloc, PCal* tmp, Sink& sink, SearchOut& out)
{
    const unsigned read_length = unsigned(r.read_length);
    const PRange realign_range = mk_range(r.realign_b, r.realign_e);
    out.nodes++;

    // the child's by-value copy of this call's state; its haplotype map and alignment are written by the caller of `spawn`
    auto spawn = [&](PFrame& c, int n_hm, unsigned depth, unsigned itd, unsigned ttd) {
        for (int i = 0; i < f.n_sm; ++i) c.sm[i] = f.sm[i];
        c.n_sm = f.n_sm;
        for (int i = 0; i < f.n_order; ++i) c.order[i] = f.order[i];
        c.n_order = f.n_order;
        c.n_hm = uint8_t(n_hm);
        c.n_nhm = 0;
        c.depth = uint16_t(depth);
        c.indel_toggle_depth = uint16_t(itd);
        c.total_toggle_depth = uint16_t(ttd);
        c.stage = 0;
        c.max_read_indel_toggle = f.max_read_indel_toggle;
        c.read_range = f.read_range;
        c.read_id = f.read_id;
    };

    // ---- entry of the call (:873-1005)
    bool is_new_indels = (f.indel_toggle_depth == 0);
    {
        const unsigned start_size = f.n_sm;
        const PRange pr = soft_clip_range(f.cal);
        if (!superset_of(realign_range, pr)) return;
        if (pr.b < f.read_range.b) {
            if (!add_indels_in_range(j, r, mk_range(pr.b, f.read_range.b + 1), f)) { out.status = ST_OVERFLOW; return; }
            f.read_range.b = pr.b;
        }
        if (pr.e > f.read_range.e) {
            if (!add_indels_in_range(j, r, mk_range(f.read_range.e - 1, pr.e), f)) { out.status = ST_OVERFLOW; return; }
            f.read_range.e = pr.e;
        }
        if (!is_new_indels) is_new_indels = (start_size != f.n_sm);
        if (is_new_indels) sort_remove_only_indels_last(f, start_size);
    }
    if (f.depth == f.n_order) {
        PCal& leaf = *tmp;
        copy_cal(leaf, f.cal);
        add_keys_to_cal(j, f, leaf);
        if (!sink(leaf)) out.status = ST_OVERFLOW;
        return;
    }
    if (is_new_indels) {
        const double max_indels = double(read_length) * j.max_candidate_indel_density;
        if (double(f.n_sm) > max_indels) f.max_read_indel_toggle = 1;
        else f.max_read_indel_toggle = j.max_read_indel_toggle;
        const int mt = int(get_max_toggle(j, f.n_sm));
        if (mt < f.max_read_indel_toggle) f.max_read_indel_toggle = mt;
    }
    if (int(f.indel_toggle_depth) > f.max_read_indel_toggle) {
        out.warn_toggle = 1;
        return;
    }

    const int cur = f.order[f.depth];
    const PIndel& cur_key = j.tab[cur];
    bool cur_conflicting = false, contains_ndfr = false;
    for (unsigned i = 0; i < f.depth; ++i) {
        const int oi = f.order[i];
        if (!f.sm[sm_find(f, oi)].is_present) continue;
        if (is_indel_conflict(j.tab[oi], cur_key)) cur_conflicting = true;
        if (!contains_ndfr && j.tab[oi].ndfr) contains_ndfr = true;
    }
    const int cur_at = sm_find(f, cur);
    const bool cur_on = f.sm[cur_at].is_present != 0;
    const int32_t arid = cur_key.arid;
    const bool in_ar = (arid >= 0);
    if (in_ar && hm_find(f.hm, f.n_hm, arid) < 0) {
        if (f.n_hm >= Caps::H) { out.status = ST_OVERFLOW; return; }
        PHap h;
        h.arid = arid;
        for (int s = 0; s < SK_MAX_SAMPLES; ++s) h.hc[s] = 3;
        h.any_on = 0;
        h.pad[0] = h.pad[1] = h.pad[2] = 0;
        f.hm[f.n_hm++] = h;
    }
    const bool cur_ndfr = cur_key.ndfr != 0;
    int hap_ids[SK_MAX_SAMPLES];
    cur_indel_haplotype_ids(j, r.sample, cur, f.sm[cur_at].in_original != 0, hap_ids);

    // the toggled children's haplotype map and validity are decided from the state as it is now (:1090-1124)
    bool valid2 = true;
    for (int i = 0; i < f.n_hm; ++i) f.nhm[i] = f.hm[i];
    f.n_nhm = f.n_hm;
    if (!cur_conflicting && in_ar) valid2 = hap_update(f.nhm[hm_find(f.nhm, f.n_nhm, arid)], j.sample_count, hap_ids, !cur_on);
    else valid2 = !is_mismatch(cur_key) || cur_on;
    if (!cur_on && contains_ndfr && cur_ndfr) valid2 = false;
    if (valid2 && !cur_on) {
        if (f.sm[cur_at].is_remove_only) valid2 = false;
        if (cur_conflicting) valid2 = false;
    }
    const unsigned toggle_inc = is_mismatch(cur_key) ? 0 : 1;
    if (valid2 && int(f.indel_toggle_depth + toggle_inc) > f.max_read_indel_toggle) {
        out.warn_toggle = 1;
        valid2 = false;
    }

    { // alignment 1: unchanged (:1051-1088)
        bool valid = true;
        PHap one;
        int at = -1;
        if (!cur_conflicting && in_ar) {
            at = hm_find(f.hm, f.n_hm, arid);
            one = f.hm[at];
            valid = hap_update(one, j.sample_count, hap_ids, cur_on);
        } else {
            valid = (!is_mismatch(cur_key)) || (!cur_on);
        }
        if (cur_on && contains_ndfr && cur_ndfr) valid = false;
        if (!valid && f.total_toggle_depth == 0) valid = true;
        if (valid) {
            PFrame* c = alloc();
            if (!c) { out.status = ST_OVERFLOW; return; }
            for (int i = 0; i < f.n_hm; ++i) c->hm[i] = f.hm[i];
            if (at >= 0) c->hm[at] = one;
            copy_cal(c->cal, f.cal);
            spawn(*c, f.n_hm, f.depth + 1, f.indel_toggle_depth, f.total_toggle_depth);
        }
    }
    if (!valid2) return;

    // toggle the indel, collect the present set (:1126-1140)
    f.sm[cur_at].is_present = cur_on ? 0 : 1;
    {
        int n = 0;
        for (int i = 0; i < f.n_sm; ++i)
            if (f.sm[i].is_present) f.current[n++] = f.sm[i].idx;
        f.n_current = uint8_t(n);
    }
    { // alignment 2: start pin (:1142-1190)
        const int32_t ref_start = f.cal.pos;
        bool start_pin_valid = true;
        if (!is_mismatch(cur_key)) {
            const bool del_span = pos_intersect(open_pos_range(cur_key), ref_start);
            const bool indel_span = cur_on && (cur == f.cal.lead);
            start_pin_valid = !(del_span || indel_span);
        }
        if (start_pin_valid) {
            const int32_t read_start = int32_t(unaligned_prefix(f.cal));
            PFrame* c = alloc();
            if (!c) { out.status = ST_OVERFLOW; return; }
            c->stage = 0xffff; // (dead unless completed below)
            const int rc = make_start_pos_alignment(j, ref_start, read_start, f.cal.fwd != 0, read_length, f.current, f.n_current, c->cal);
            if (rc != ST_OK) { out.status = rc; return; }
            for (int i = 0; i < f.n_nhm; ++i) c->hm[i] = f.nhm[i];
            spawn(*c, f.n_nhm, f.depth + 1, f.indel_toggle_depth + toggle_inc, f.total_toggle_depth + 1);
        }
    }
    if (is_mismatch(cur_key)) return;
    if (cur_key.type == SK_INDEL_INDEL && cur_key.del == cur_key.ins_len) return;
    { // alignment 3: end pin (:1198-1270)
        const int32_t ref_end = f.cal.pos + int32_t(path_ref_length(f.cal));
        const bool del_span = pos_intersect(open_pos_range(cur_key), ref_end - 1);
        const bool indel_span = cur_on && (cur == f.cal.trail);
        if (del_span || indel_span) return;
        const int32_t read_end = int32_t(read_length) - int32_t(unaligned_suffix(f.cal));
        int32_t ref_start = 0, read_start = 0;
        const int rc0 = get_end_pin_start_pos(j, f.current, f.n_current, read_length, ref_end, read_end, ref_start, read_start);
        if (rc0 != ST_OK) { out.status = rc0; return; }
        if (ref_start < 0) {
            out.warn_origin = 1;
            return;
        }
        PFrame* c = alloc();
        if (!c) { out.status = ST_OVERFLOW; return; }
        c->stage = 0xffff;
        const int rc = make_start_pos_alignment(j, ref_start, read_start, f.cal.fwd != 0, read_length, f.current, f.n_current, c->cal);
        if (rc != ST_OK) { out.status = rc; return; }
        for (int i = 0; i < f.n_nhm; ++i) c->hm[i] = f.nhm[i];
        spawn(*c, f.n_nhm, f.depth + 1, f.indel_toggle_depth + toggle_inc, f.total_toggle_depth + 1);
    }
}

// The whole search of one read, depth first on a stack of frames (`stack`: up to `max_frames`, 2 * Caps::K + 4 always suffice:
// expanding a frame takes it off the stack and puts at most three on).  Leaves reach the sink in an order that differs from the
// reference's (children are expanded last-made-first); the set they form is the same.
template <typename Sink>
SKC_HD inline SearchOut candidate_alignment_search(const PJob& j, const PRead& r, PFrame* stack, const int max_frames, PFrame* cur,
                                                   PCal* tmp, Sink& sink)
{
    SearchOut out;
    out.status = ST_OK;
    out.warn_origin = out.warn_toggle = 0;
    out.nodes = 0;
    int sp = 0;
    root_frame(j, r, stack[0]);
    sp = 1;
    struct StackAlloc
    {
        PFrame* stack;
        int* sp;
        int max_frames;
        SKC_HD PFrame* operator()()
        {
            if (*sp >= max_frames) return nullptr;
            return &stack[(*sp)++];
        }
    } alloc{ stack, &sp, max_frames };
    while (sp > 0) {
        // by value: the frame leaves the stack, its children may take its slot
        {
            const PFrame& top = stack[sp - 1];
            for (int i = 0; i < top.n_sm; ++i) cur->sm[i] = top.sm[i];
            cur->n_sm = top.n_sm;
            for (int i = 0; i < top.n_order; ++i) cur->order[i] = top.order[i];
            cur->n_order = top.n_order;
            for (int i = 0; i < top.n_hm; ++i) cur->hm[i] = top.hm[i];
            cur->n_hm = top.n_hm;
            cur->n_nhm = 0;
            cur->depth = top.depth;
            cur->indel_toggle_depth = top.indel_toggle_depth;
            cur->total_toggle_depth = top.total_toggle_depth;
            cur->stage = top.stage;
            cur->max_read_indel_toggle = top.max_read_indel_toggle;
            cur->read_range = top.read_range;
            cur->read_id = top.read_id;
            copy_cal(cur->cal, top.cal);
        }
        --sp;
        if (cur->stage == 0xffff) continue;
        if (j.max_nodes > 0 && out.nodes >= j.max_nodes) {
            out.status = ST_OVERFLOW;
            return out;
        }
        expand_node(j, r, *cur, alloc, tmp, sink, out);
        if (out.status != ST_OK) return out;
    }
    return out;
}

// clip_adder :508-542 on a leaf, in place
SKC_HD inline bool clip_adder(PCal& c, unsigned hc_lead, unsigned hc_trail, unsigned sc_lead, unsigned sc_trail)
{
    const int n_lead = (hc_lead ? 1 : 0) + (sc_lead ? 1 : 0), n_trail = (hc_trail ? 1 : 0) + (sc_trail ? 1 : 0);
    if (int(c.n_seg) + n_lead + n_trail > Caps::P) return false;
    if (hc_lead > 0xffffu || hc_trail > 0xffffu || sc_lead > 0xffffu || sc_trail > 0xffffu) return false;
    if (n_lead)
        for (int i = int(c.n_seg) - 1; i >= 0; --i) c.path[i + n_lead] = c.path[i];
    int n = 0;
    if (hc_lead) {
        c.path[n].type = SK_SEG_HARD_CLIP;
        c.path[n++].length = uint16_t(hc_lead);
    }
    if (sc_lead) {
        c.path[n].type = SK_SEG_SOFT_CLIP;
        c.path[n++].length = uint16_t(sc_lead);
    }
    n += c.n_seg;
    if (sc_trail) {
        c.path[n].type = SK_SEG_SOFT_CLIP;
        c.path[n++].length = uint16_t(sc_trail);
    }
    if (hc_trail) {
        c.path[n].type = SK_SEG_HARD_CLIP;
        c.path[n++].length = uint16_t(hc_trail);
    }
    c.n_seg = uint8_t(n);
    return true;
}

} // namespace skcore
