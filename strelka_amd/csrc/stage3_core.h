// stage3_core.h -- stage 3 of the read path (rows a6, a7 of SURVEY.md section 8) as container-free code for the device and the host:
//
//   scoreCandidateAlignments   L/starling_common/starling_read_align.cpp:1534-1741  max / smooth pool, tie rules :1351-1376
//   finishRealignment          :1409-1449 with the clipper, starling_read_align_clipper.cpp:109-424
//   score_indels               starling_read_align_score_indels.cpp:455-1079 (late_indel_normalization_filter :303-450,
//                              get_alignment_indel_bp_overlap :134-228, updateIndelScoringInfo :61-77, which_interfering_indel :100-119)
//
// host/read_realign.cpp holds the container-based statement of the same functions (pinned to the reference: golden fixtures, live
// comparison); this one works on the device's own data -- the read's candidate alignments as PCal records in std::set order, their
// scores, the indel table -- so that after the scoring kernel only results leave the device.  A read beyond a capacity below is
// reported (status) and finished by the host code; nothing is truncated.
#pragma once

#include "realign_core.h"

namespace sk3
{

using namespace skcore;

enum {
    E_MAX = 16,    // indels evaluated for one read (score_indels' `indelsToEvaluate`)
    SUB_MAX = 32,  // suboverlap entries of one read
    RL_MAX = 1024, // read length the clipper's per-base map holds
    OUT_SEGS = Caps::P + 4,
    RAISED_WORDS = 64, // 32-bit words of the late normalisation filter's set of raised shape hashes
    MAX_LANES = 256, // of a lane group
    CONF_SPAN = 256 // table indices one read's alignments span, for the conflict table of score_indels
};
enum { S3_OK = 0, S3_CAPACITY = 1, S3_FAIL = 2 };

struct Tab // the job's indel table as stage 3 reads it
{
    const PIndel* tab;
    const double* r2i; // refToIndel / indelToRef log error rates of the read's sample
    const double* i2r;
    const int32_t* orig; // index as given to sk_realign_job_set_indels
    int32_t n_tab;
    int32_t max_indel_size;
    uint8_t* consulted;
};
SKC_HD inline bool tab_cand(const Tab& t, const int i)
{
    if (t.consulted) t.consulted[i] = 1;
    return t.tab[i].cand != 0;
}

struct Opt
{
    int32_t is_smoothed_alignments;
    double smoothed_lnp_range;
    uint32_t upstream_oligo_size;
    int32_t min_read_bp_flank;
    int32_t no_tables; // (tests: score_indels without its per-read conflict tables, as for reads whose indels span too much of the table)
};

// A read's candidate alignments in std::set order: an array of records, or -- on the device, where the search leaves its leaves in
// one pool and the set order is a list of slots -- the pool and the read's part of that list (no copy of the records in set order)
struct CalView
{
    const PCal* base;
    const int32_t* slot; // null: base[i] is alignment i
    SKC_HD const PCal& operator[](const int i) const { return slot ? base[slot[i]] : base[i]; }
};

struct Read
{
    CalView cals; // std::set order
    const double* scores;        // of the candidate alignments
    const double* scores_select; // the same values, maybe in closer memory; read until the late normalisation filter has sorted
    int32_t n_cals;
    int32_t map_level;
    int32_t read_length; // of the whole read
    int32_t non_ambig;   // bases that are not 'N'
};

struct Scratch // per read, sized by the caller; [n_cals] each unless noted
{
    uint8_t* flag;    // by alignment: bit 0: removed by the late normalisation filter, bit 1: in the smooth pool
    uint32_t* key;    // [4 * n_cals] extra_path_info + candidate indel count of each alignment
    // the late normalisation filter's, by PLACE in the order by score:
    int32_t* order;        // the alignment at the place
    double* sorted_score;  // its score
    double* smooth;        // its smoothed score (may be the memory of Read::scores_select)
    uint32_t* sorted_hash; // cal_shape_hash
    int32_t* next_same;    // see late_indel_normalization_filter
    int32_t* range_end;    // the first place whose score is out of range of this place's score
    uint8_t* removed;
    uint8_t* rm_type; // [read_length] clipper's per-base map
    int32_t* rm_pos;  // [read_length]
};

struct Out
{
    int32_t status;
    int32_t is_realigned;
    int32_t realign_pos;
    int32_t n_seg;
    PSeg path[OUT_SEGS];
    double max_score;
    int32_t n_scores, n_sub;
    sk_read_path_scores scores[E_MAX];
    int32_t sub[SUB_MAX];
};

// ---- extra_path_info :1281-1321 and the tie rule :1351-1376
struct PathInfo
{
    unsigned indel_count, del_size, ins_size, sum_seg_pos;
};
SKC_HD inline PathInfo path_info(const PCal& c)
{
    PathInfo e;
    e.indel_count = e.del_size = e.ins_size = e.sum_seg_pos = 0;
    unsigned read_pos = 0;
    for (int i = 0; i < c.n_seg; ++i) {
        const PSeg s = c.path[i];
        if (!seg_align_match(s.type)) e.indel_count++;
        if (s.type == SK_SEG_DELETE) {
            e.del_size += s.length;
            e.sum_seg_pos += read_pos;
        }
        if (s.type == SK_SEG_INSERT) {
            e.ins_size += s.length;
            e.sum_seg_pos += read_pos;
        }
        if (seg_read_len(s.type)) read_pos += s.length;
    }
    return e;
}
SKC_HD inline unsigned path_read_length(const PCal& c)
{
    unsigned v = 0;
    for (int i = 0; i < c.n_seg; ++i)
        if (seg_read_len(c.path[i].type)) v += c.path[i].length;
    return v;
}

// ---- the clipper (starling_read_align_clipper.cpp): per read base what it is aligned to, NONE/MATCH/INSERT/SOFT_CLIP/CONFLICT
enum { RM_NONE = 0, RM_MATCH = 1, RM_INSERT = 2, RM_SOFT_CLIP = 3, RM_CONFLICT = 4 };

SKC_HD inline int alignment_ref_map(const PCal& al, const Scratch& w, const int cap, int& n) // get_alignment_ref_map :109-157
{
    n = 0;
    int32_t ref_head = al.pos;
    for (int i = 0; i < al.n_seg; ++i) {
        const PSeg s = al.path[i];
        if (seg_align_match(s.type) || s.type == SK_SEG_INSERT || s.type == SK_SEG_SOFT_CLIP) {
            if (n + int(s.length) > cap) return S3_CAPACITY;
            for (unsigned j = 0; j < s.length; ++j) {
                w.rm_type[n] = seg_align_match(s.type) ? uint8_t(RM_MATCH) : (s.type == SK_SEG_INSERT ? uint8_t(RM_INSERT) : uint8_t(RM_SOFT_CLIP));
                w.rm_pos[n] = seg_align_match(s.type) ? ref_head + int32_t(j) : 0;
                ++n;
            }
            if (seg_align_match(s.type)) ref_head += int32_t(s.length);
        } else if (s.type == SK_SEG_DELETE || s.type == SK_SEG_SKIP) {
            ref_head += int32_t(s.length);
        } else if (s.type != SK_SEG_HARD_CLIP) {
            return S3_FAIL;
        }
    }
    return S3_OK;
}
SKC_HD inline int mark_ref_map_conflicts(const PCal& al, const Scratch& w, const int n) // :161-231
{
    int32_t ref_head = al.pos, read_head = 0;
    for (int i = 0; i < al.n_seg; ++i) {
        const PSeg s = al.path[i];
        if (seg_align_match(s.type) || s.type == SK_SEG_INSERT || s.type == SK_SEG_SOFT_CLIP) {
            if (read_head + int32_t(s.length) > n) return S3_FAIL; // (the host code would index past its map)
            for (unsigned j = 0; j < s.length; ++j) {
                const int at = read_head + int32_t(j);
                if (w.rm_type[at] == RM_CONFLICT) continue;
                bool ok;
                if (seg_align_match(s.type)) ok = (w.rm_type[at] == RM_MATCH) && (w.rm_pos[at] == ref_head + int32_t(j));
                else if (s.type == SK_SEG_INSERT) ok = (w.rm_type[at] == RM_INSERT);
                else ok = (w.rm_type[at] == RM_SOFT_CLIP);
                if (!ok) w.rm_type[at] = RM_CONFLICT;
            }
            read_head += int32_t(s.length);
            if (seg_align_match(s.type)) ref_head += int32_t(s.length);
        } else if (s.type == SK_SEG_DELETE || s.type == SK_SEG_SKIP) {
            ref_head += int32_t(s.length);
        } else if (s.type != SK_SEG_HARD_CLIP) {
            return S3_FAIL;
        }
    }
    return S3_OK;
}

struct OutPath
{
    int32_t pos;
    int n;
    PSeg* seg;
    bool overflow;
    SKC_HD void push(const unsigned type, const unsigned length)
    {
        if (n >= OUT_SEGS || length > 0xffffu) {
            overflow = true;
            return;
        }
        seg[n].type = uint16_t(type);
        seg[n].length = uint16_t(length);
        ++n;
    }
    SKC_HD void extend_or_add_sc(const unsigned len)
    {
        if (n > 0 && seg[n - 1].type == SK_SEG_SOFT_CLIP) {
            if (unsigned(seg[n - 1].length) + len > 0xffffu) overflow = true;
            else seg[n - 1].length = uint16_t(seg[n - 1].length + len);
        } else {
            push(SK_SEG_SOFT_CLIP, len);
        }
    }
};

SKC_HD inline int soft_clip_alignment(const PCal& al, const unsigned leading_clip, const unsigned trailing_clip, OutPath& o) // :250-339
{
    unsigned read_head = 0;
    o.pos = al.pos;
    o.n = 0;
    for (int i = 0; i < al.n_seg; ++i) {
        const PSeg s = al.path[i];
        if (seg_align_match(s.type) || s.type == SK_SEG_INSERT) {
            if (leading_clip > read_head) {
                const unsigned clip = (unsigned(s.length) < leading_clip - read_head) ? unsigned(s.length) : leading_clip - read_head;
                o.extend_or_add_sc(clip);
                if (seg_align_match(s.type)) o.pos += int32_t(clip);
                if (clip < s.length) o.push(s.type, s.length - clip);
            } else if (trailing_clip < read_head + s.length) {
                const unsigned over = (read_head + s.length) - trailing_clip;
                const unsigned clip = (unsigned(s.length) < over) ? unsigned(s.length) : over;
                if (clip < s.length) o.push(s.type, s.length - clip);
                o.extend_or_add_sc(clip);
            } else {
                o.push(s.type, s.length);
            }
            read_head += s.length;
        } else if (s.type == SK_SEG_DELETE || s.type == SK_SEG_SKIP) {
            if (leading_clip >= read_head) o.pos += int32_t(s.length);
            else if (trailing_clip <= read_head) {
            } else o.push(s.type, s.length);
        } else if (s.type == SK_SEG_SOFT_CLIP) {
            o.extend_or_add_sc(s.length);
            read_head += s.length;
        } else if (s.type == SK_SEG_HARD_CLIP) {
            o.push(s.type, s.length);
        } else {
            return S3_FAIL;
        }
    }
    return o.overflow ? S3_CAPACITY : S3_OK;
}

SKC_HD inline void copy_path(const PCal& al, OutPath& o)
{
    o.pos = al.pos;
    o.n = 0;
    for (int i = 0; i < al.n_seg; ++i) o.push(al.path[i].type, al.path[i].length);
}

// get_alignment_indel_bp_overlap :134-228 -> max(left, right)
SKC_HD inline int indel_bp_overlap(const unsigned upstream_oligo, const PCal& al, const PIndel& k, int& status)
{
    int32_t read_head = 0, ref_head = al.pos;
    bool has_l = false, has_r = false;
    int32_t lpos = 0, rpos = 0;
    const int32_t kpos = k.pos, krp = right_pos(k);
    for (int i = 0; i < al.n_seg; ++i) {
        const PSeg s = al.path[i];
        int32_t nread = read_head, nref = ref_head;
        if (seg_align_match(s.type)) {
            nread += int32_t(s.length);
            nref += int32_t(s.length);
        } else if (s.type == SK_SEG_INSERT) nread += int32_t(s.length);
        else if (s.type == SK_SEG_DELETE) nref += int32_t(s.length);
        else if (s.type == SK_SEG_SOFT_CLIP || s.type == SK_SEG_HARD_CLIP) {
        } else {
            status = S3_FAIL;
            return 0;
        }
        if (!has_l && kpos <= nref) {
            lpos = read_head + (kpos - ref_head);
            has_l = true;
        }
        if (!has_r && krp < nref) {
            rpos = read_head + (krp - ref_head);
            has_r = true;
        }
        read_head = nread;
        ref_head = nref;
    }
    int lext = 0, rext = 0;
    if (al.fwd) {
        if (lpos > 0) lext = int(upstream_oligo);
    } else {
        if ((read_head - rpos) > 0) rext = int(upstream_oligo);
    }
    int lo = 0, ro = 0;
    if (has_l) {
        const int a = lpos + lext, b = read_head - lpos;
        lo = (a < b ? a : b);
        if (lo < 0) lo = 0;
    }
    if (has_r) {
        const int a = rpos, b = (read_head - rpos) + rext;
        ro = (a < b ? a : b);
        if (ro < 0) ro = 0;
    }
    return lo > ro ? lo : ro;
}

SKC_HD inline int32_t translate_ref_offset_to_read_offset(const int32_t target, const PSeg* p, const int n) // alignment_util.cpp:226-263
{
    if (target < 0) return -1;
    int32_t ref_off = 0, read_off = 0;
    for (int i = 0; i < n; ++i) {
        const PSeg s = p[i];
        if (seg_read_len(s.type)) read_off += int32_t(s.length);
        if (!seg_ref_len(s.type)) continue;
        ref_off += int32_t(s.length);
        if (ref_off <= target) continue;
        if (!seg_read_len(s.type)) return -1;
        return read_off - (ref_off - target);
    }
    return -1;
}
SKC_HD inline int32_t lowest_fwd_read_pos(const PCal& al, const bool fwd, const int32_t rb, const int32_t re) // :267-300
{
    const int32_t ref_off = (fwd ? rb : re - 1) - al.pos;
    const int32_t ro = translate_ref_offset_to_read_offset(ref_off, al.path, al.n_seg);
    if (ro < 0) return ro;
    if (fwd) return ro;
    return int32_t(path_read_length(al)) - (ro + 1);
}

SKC_HD inline bool cal_has(const PCal& c, const int idx)
{
    int a = 0, b = c.n_indels;
    while (a < b) {
        const int m = (a + b) >> 1;
        if (c.indels[m] < idx) a = m + 1; else b = m;
    }
    return a < c.n_indels && c.indels[a] == idx;
}


// ---- cooperative form ------------------------------------------------------------------------------------------------------------
// Stage 3 of one read is run by a group of lanes: one on the host (HostLanes below), the 64 lanes of a wavefront on the device
// (stage3_kernel).  The reference's loops over the read's candidate alignments are spread over the lanes where their iterations are
// independent (per-alignment features, pool membership, the clipper's conflict marking, the ordering by score, the equivalence tests
// of the late normalisation filter, the best alignment holding an indel, the (evaluated indel, alignment) score table); what the
// reference defines by iteration order -- the tie rules of the max / smooth alignment, the filter's removals, the evaluated-indel
// list, the output records -- is decided by lane 0 from what the lanes prepared.  All state the lanes share lives in `Shared` and
// the scratch arrays; `sync()` separates a phase that writes it from one that reads it and is reached by every lane.
// The places that consult an indel's candidate status (tab_cand) are lane 0's, in the reference's order: the set of indels marked
// consulted is the reference's.
//
// A lane group type provides: id, width, sync(), max_i64(p, v) (atomic max).
struct HostLanes
{
    int id = 0, width = 1;
    void sync() const {}
    void max_i64(long long* p, const long long v) const
    {
        if (*p < v) *p = v;
    }
};

// a double as an integer with the same order (the score table keeps maxima; INT64_MIN = no entry)
SKC_HD inline long long f64_key(const double d)
{
    long long b;
    __builtin_memcpy(&b, &d, 8);
    return b >= 0 ? b : (b ^ 0x7fffffffffffffffLL);
}
SKC_HD inline double key_f64(const long long k)
{
    const long long b = k >= 0 ? k : (k ^ 0x7fffffffffffffffLL);
    double d;
    __builtin_memcpy(&d, &b, 8);
    return d;
}
#define SK3_NO_ENTRY (-0x7fffffffffffffffLL - 1)

struct Shared
{
    int32_t status;
    int32_t max_i, smooth_i, n_pool, nmap;
    double max_score;
    int32_t any_excluded, found;
    int32_t n_raised;
    uint32_t raised_bits[RAISED_WORDS]; // a set of (shape hash mod size) of the places whose smoothed score is no longer their score
    long long idx_hi, idx_nlo; // the largest table index in the read's alignments, the largest negated one
    uint16_t conf_mask[CONF_SPAN]; // [i - idx_min]: the evaluated indels table indel i (not a mismatch) conflicts with
    uint8_t eval_slot[CONF_SPAN];  // [i - idx_min]: which evaluated indel table indel i is, 0xff: none
    int32_t lo, hi, ne;
    int16_t to_eval[E_MAX];
    uint32_t ortho[E_MAX];
    double red_score[MAX_LANES]; // per lane: the best alignment holding the indel at hand
    int32_t red_i[MAX_LANES];
    long long info[2 * E_MAX * E_MAX]; // [evaluated indel][is present][which indel] -> f64_key of the best score
};

SKC_HD inline void mark_cal(const Tab& t, const PCal& c)
{
    if (!t.consulted) return;
    for (int i = 0; i < c.n_indels; ++i) t.consulted[c.indels[i]] = 1;
}
// is_first_cal_preferred :1351-1376 on the precomputed features; consults the candidate status of both alignments' indels exactly
// where the reference counts candidate indels (equal indel counts)
SKC_HD inline bool first_cal_preferred(const Tab& t, const Read& rd, const Scratch& w, const int a, const int b)
{
    const uint32_t* ka = w.key + 4 * a;
    const uint32_t* kb = w.key + 4 * b;
    const unsigned ic1 = ka[0] >> 16, ic2 = kb[0] >> 16;
    if (ic2 < ic1) return false;
    if (ic2 > ic1) return true;
    mark_cal(t, rd.cals[a]);
    mark_cal(t, rd.cals[b]);
    const unsigned k1 = ka[0] & 0xffffu, k2 = kb[0] & 0xffffu;
    if (k2 > k1) return false;
    if (k2 < k1) return true;
    if (kb[1] < ka[1]) return false;
    if (kb[1] > ka[1]) return true;
    if (kb[2] < ka[2]) return false;
    if (kb[2] > ka[2]) return true;
    return kb[3] >= ka[3];
}

// scoreCandidateAlignments :1534-1741 + finishRealignment :1409-1449
template <typename L>
SKC_HD inline void select_alignments(const L& ln, const Tab& t, const Opt& opt, const Read& rd, const Scratch& w, Shared& sh, Out& out)
{
    const int n = rd.n_cals;
    for (int i = ln.id; i < n; i += ln.width) {
        const PCal& c = rd.cals[i];
        const PathInfo e = path_info(c);
        unsigned nc = 0;
        for (int q = 0; q < c.n_indels; ++q)
            if (t.tab[c.indels[q]].cand) ++nc;
        w.key[4 * i + 0] = (e.indel_count << 16) | nc;
        w.key[4 * i + 1] = e.ins_size;
        w.key[4 * i + 2] = e.del_size;
        w.key[4 * i + 3] = e.sum_seg_pos;
        if (c.n_indels > 0) { // (ascending)
            ln.max_i64(&sh.idx_hi, c.indels[c.n_indels - 1]);
            ln.max_i64(&sh.idx_nlo, -(long long)c.indels[0]);
        }
    }
    ln.sync();
    if (ln.id == 0) {
        int max_i = -1;
        double max_score = 0;
        for (int i = 0; i < n; ++i) {
            const double lnp = rd.scores_select[i];
            if (max_i >= 0) {
                if (lnp < max_score) continue;
                if (lnp <= max_score && first_cal_preferred(t, rd, w, max_i, i)) continue;
            }
            max_score = lnp;
            max_i = i;
        }
        sh.max_i = max_i;
        sh.max_score = max_score;
        out.max_score = max_score;
    }
    ln.sync();
    {
        const double max_score = sh.max_score;
        const double allowed_range = opt.is_smoothed_alignments ? opt.smoothed_lnp_range : 0.;
        for (int i = ln.id; i < n; i += ln.width) w.flag[i] = ((rd.scores_select[i] + allowed_range) < max_score) ? 0 : 2;
    }
    ln.sync();
    if (ln.id == 0) {
        int smooth_i = -1, n_pool = 0;
        for (int i = 0; i < n; ++i) {
            if (!(w.flag[i] & 2)) continue;
            ++n_pool;
            if (smooth_i < 0 || !first_cal_preferred(t, rd, w, smooth_i, i)) smooth_i = i;
        }
        sh.smooth_i = smooth_i;
        sh.n_pool = n_pool;
        sh.nmap = 0;
        if (smooth_i < 0) {
            sh.status = S3_FAIL;
        } else {
            out.is_realigned = 1;
            if (n_pool > 1) { // get_clipped_alignment_from_cal_pool :343-424
                int nmap = 0;
                const int rc = alignment_ref_map(rd.cals[smooth_i], w, (rd.read_length < RL_MAX ? rd.read_length : RL_MAX), nmap);
                if (rc != S3_OK) sh.status = rc;
                sh.nmap = nmap;
            }
        }
    }
    ln.sync();
    if (sh.status != S3_OK) return;
    if (sh.n_pool > 1) {
        // (a base once in conflict stays so and the tests read only what get_alignment_ref_map wrote: the pool's order is immaterial)
        for (int i = ln.id; i < n; i += ln.width) {
            if (!(w.flag[i] & 2) || i == sh.smooth_i) continue;
            const int rc = mark_ref_map_conflicts(rd.cals[i], w, sh.nmap);
            if (rc != S3_OK) sh.status = rc;
        }
        ln.sync();
    }
    if (ln.id == 0 && sh.status == S3_OK) {
        OutPath o;
        o.seg = out.path;
        o.n = 0;
        o.pos = 0;
        o.overflow = false;
        const PCal& best = rd.cals[sh.smooth_i];
        bool have = false;
        if (sh.n_pool > 1) {
            const int nmap = sh.nmap;
            int lead = 0;
            for (; lead < nmap; ++lead)
                if (w.rm_type[lead] == RM_MATCH) break;
            for (; lead > 0; --lead)
                if (w.rm_type[lead - 1] == RM_CONFLICT || w.rm_type[lead - 1] == RM_SOFT_CLIP) break;
            int trail = nmap;
            for (; trail > 0; --trail)
                if (w.rm_type[trail - 1] == RM_MATCH) break;
            for (; trail < nmap; ++trail)
                if (w.rm_type[trail] == RM_CONFLICT || w.rm_type[trail] == RM_SOFT_CLIP) break;
            if (lead < trail) {
                if (lead != 0 || trail != nmap) {
                    const int rc = soft_clip_alignment(best, unsigned(lead), unsigned(trail), o);
                    if (rc != S3_OK) sh.status = rc;
                } else {
                    copy_path(best, o);
                }
                have = (o.n > 0);
            }
        }
        if (!have) copy_path(best, o);
        if (o.overflow) sh.status = S3_CAPACITY;
        out.realign_pos = o.pos;
        out.n_seg = o.n;
    }
    ln.sync();
}

// Equivalent alignments (below) have indels of the same shapes in the same order (PIndel.shape numbers the distinct (type, deletion
// length, insert sequence) of the table): a hash of the shapes tells most non-equivalent pairs apart without reading the alignments
SKC_HD inline uint32_t cal_shape_hash(const Tab& t, const PCal& c)
{
    uint32_t h = 0x9e3779b9u + uint32_t(c.n_indels);
    for (int q = 0; q < c.n_indels; ++q) h = (h ^ t.tab[c.indels[q]].shape) * 0x01000193u + uint32_t(q);
    return h;
}

// which of the two alignments' indels (same count, ascending) differ: bit q = pair q.  Both lists are read whole, with loads that
// do not wait for each other, instead of entry by entry behind the comparison of the previous entries
SKC_HD inline uint64_t differing_pairs(const PCal& a, const PCal& b)
{
    enum { WORDS = (Caps::K + 2) / 2 };
    static_assert((Caps::K + 2) % 2 == 0 && Caps::K + 2 <= 64, "indel lists are compared as 32-bit words, one bit per entry");
    uint32_t wa[WORDS], wb[WORDS];
    __builtin_memcpy(wa, a.indels, sizeof(wa));
    __builtin_memcpy(wb, b.indels, sizeof(wb));
    uint64_t d = 0;
    for (int w = 0; w < WORDS; ++w) {
        const uint32_t x = wa[w] ^ wb[w];
        if (x & 0xffffu) d |= uint64_t(1) << (2 * w);
        if (x >> 16) d |= uint64_t(1) << (2 * w + 1);
    }
    const int n = a.n_indels;
    return n >= 64 ? d : (d & ((uint64_t(1) << n) - 1)); // (entries past the count are not part of the lists)
}

// is_equiv_candidate :240-269: the same indels up to position, with at least one pair that differs
SKC_HD inline bool equiv_with_pairs(const Tab& t, const PCal& a, const PCal& b)
{
    if (a.n_indels != b.n_indels) return false;
    uint64_t d = differing_pairs(a, b);
    if (d == 0) return false;
    for (; d != 0; d &= d - 1) {
        const int q = __builtin_ctzll(d);
        // same type, deletion length and insert SEQUENCE (equal insert lengths with different sequences are different keys)
        if (t.tab[a.indels[q]].shape != t.tab[b.indels[q]].shape) return false;
    }
    return true;
}

// ---- late_indel_normalization_filter :303-450 ----
// The reference sorts the alignments by (score, index) descending and, for the alignment at each place i1 of that order, walks the
// later places: removed ones are skipped, the walk ends at the first one whose smoothed score is out of range, an equivalent one
// (is_equiv_candidate) leads to the removal of one of the two and hands its smoothed score on (`apply_equivalent`).
// Here the order, the smoothed scores and the removals are kept by PLACE; the lanes sort and link, lane 0 makes the walks.

// one equivalent pair found by the walk of place i1 at place i2: :386-431
struct Applied
{
    bool s1_removed, x1_raised;
};
SKC_HD inline void note_raised(const Scratch& w, Shared& sh, const int place)
{
    const uint32_t h = w.sorted_hash[place] % uint32_t(32 * RAISED_WORDS);
    sh.raised_bits[h >> 5] |= 1u << (h & 31);
    sh.n_raised++;
}
SKC_HD inline bool maybe_raised(const Shared& sh, const uint32_t hash) // false: no place with this hash has a raised smoothed score
{
    if (sh.n_raised == 0) return false;
    const uint32_t h = hash % uint32_t(32 * RAISED_WORDS);
    return ((sh.raised_bits[h >> 5] >> (h & 31)) & 1u) != 0;
}
SKC_HD inline Applied apply_equivalent(const Tab& t, const Read& rd, const Scratch& w, Shared& sh, const int i1, const int i2)
{
    const PCal& a = rd.cals[w.order[i1]];
    const PCal& b = rd.cals[w.order[i2]];
    Applied r;
    r.s1_removed = r.x1_raised = false;
    bool removed = false;
    for (uint64_t d = differing_pairs(a, b); d != 0; d &= d - 1) { // the pairs in set order (a is ascending)
        const int q = __builtin_ctzll(d);
        const int p1 = a.indels[q], p2 = b.indels[q];
        const bool c1 = tab_cand(t, p1), c2 = tab_cand(t, p2); // is_first_indel_dominant :276-292
        bool first_dom;
        if (c2 && !c1) first_dom = false;
        else if (c2 == c1) first_dom = (t.tab[p1].pos <= t.tab[p2].pos);
        else first_dom = true;
        if (!removed) {
            sh.any_excluded = 1;
            if (first_dom) {
                w.removed[i2] = 1;
                if (w.smooth[i1] < w.smooth[i2]) {
                    w.smooth[i1] = w.smooth[i2];
                    note_raised(w, sh, i1);
                    r.x1_raised = true;
                }
            } else {
                w.removed[i1] = 1;
                if (w.smooth[i2] < w.smooth[i1]) {
                    w.smooth[i2] = w.smooth[i1];
                    note_raised(w, sh, i2);
                }
                r.s1_removed = true;
            }
        }
        removed = true;
    }
    return r;
}

// leaves the (possibly new) max alignment in sh.max_i / sh.max_score and bit 0 of flag[] set for the alignments removed
template <typename L>
SKC_HD inline void late_indel_normalization_filter(const L& ln, const Tab& t, const Opt& opt, const Read& rd, const Scratch& w, Shared& sh)
{
    const int n = rd.n_cals;
    const double equiv_range = opt.is_smoothed_alignments ? opt.smoothed_lnp_range : 0.;
    // (score, index) pairs in DESCENDING pair order -- what std::sort(rbegin, rend) of the reference leaves.  A bitonic network whose
    // comparators all point the same way, so that places past n (which would hold the pairs that sort last) need not exist
    for (int i = ln.id; i < n; i += ln.width) {
        w.order[i] = i;
        w.sorted_score[i] = rd.scores_select[i];
        w.flag[i] &= uint8_t(~1u);
    }
    if (ln.id == 0) {
        sh.any_excluded = 0;
        sh.found = 0;
        sh.n_raised = 0;
    }
    for (int z = ln.id; z < RAISED_WORDS; z += ln.width) sh.raised_bits[z] = 0;
    ln.sync();
    for (int k = 2; (k >> 1) < n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool flip = (j == (k >> 1));
            for (int c = ln.id; c < n; c += ln.width) { // comparator c of this step (its low place is at or after c)
                const int lo = ((c / j) * 2 * j) + (c % j);
                const int hi = flip ? (lo ^ (k - 1)) : (lo + j);
                if (lo >= n) break;
                if (hi >= n) continue;
                const double xl = w.sorted_score[lo], xh = w.sorted_score[hi];
                const int il = w.order[lo], ih = w.order[hi];
                if (xl < xh || (xl == xh && il < ih)) { // the pair at the low place sorts after the other one: swap
                    w.sorted_score[lo] = xh;
                    w.sorted_score[hi] = xl;
                    w.order[lo] = ih;
                    w.order[hi] = il;
                }
            }
            ln.sync();
        }
    }
    // (rd.scores_select is not read from here on: `smooth` may be the same memory)
    for (int i = ln.id; i < n; i += ln.width) {
        w.sorted_hash[i] = cal_shape_hash(t, rd.cals[w.order[i]]);
        w.smooth[i] = w.sorted_score[i];
        w.removed[i] = 0;
    }
    ln.sync();
    // Equivalent alignments have the same shape hash: every lane links its places to the next place down the order with the same hash
    // whose score is within the range (next_same; -1: none).  Same-hash places whose scores are within the range of each other are all
    // on one chain of such links.
    for (int i1 = ln.id; i1 < n; i1 += ln.width) {
        const double x1 = w.sorted_score[i1];
        const uint32_t h1 = w.sorted_hash[i1];
        int nxt = -1;
        for (int i2 = i1 + 1; i2 < n; ++i2) {
            if (w.sorted_score[i2] + equiv_range < x1) break;
            if (w.sorted_hash[i2] == h1) {
                nxt = i2;
                break;
            }
        }
        w.next_same[i1] = nxt;
        if (nxt >= 0) sh.found = 1;
        int lo = i1 + 1, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (w.sorted_score[mid] + equiv_range < x1) hi = mid; else lo = mid + 1;
        }
        w.range_end[i1] = lo;
    }
    ln.sync();
    if (!sh.found) return; // no two alignments can be equivalent: the reference's loop changes nothing
    // The walks.  With x1 the smoothed score of place i1: every place whose SCORE is in range of x1 is in range (a smoothed score is
    // never below the score) and those places come first; among them only the linked ones can be equivalent to i1.  After them the walk
    // goes on over removed places and places whose smoothed score was raised into the range, and ends at the first other place: that
    // tail matters only when a raised place has the hash of i1.
    if (ln.id == 0) {
        for (int i1 = 0; i1 < n; ++i1) {
            int m = w.next_same[i1];
            const bool tail = maybe_raised(sh, w.sorted_hash[i1]);
            if (m < 0 && !tail) continue;
            if (w.removed[i1]) continue;
            double x1 = w.smooth[i1];
            int last = i1;
            bool s1_removed = false;
            for (int prev = i1; m >= 0; m = w.next_same[m]) {
                if (w.sorted_score[m] + equiv_range < x1) break; // past the places in range by score
                last = m;
                if (w.removed[m]) { // (removed for good: later walks need not pass here again)
                    w.next_same[prev] = w.next_same[m];
                    continue;
                }
                prev = m;
                if (!equiv_with_pairs(t, rd.cals[w.order[i1]], rd.cals[w.order[m]])) continue;
                const Applied ap = apply_equivalent(t, rd, w, sh, i1, m);
                if (ap.s1_removed) {
                    s1_removed = true;
                    break;
                }
                if (ap.x1_raised) x1 = w.smooth[i1];
            }
            if (s1_removed || !tail) continue;
            int p = w.range_end[i1]; // the first place out of range by score
            if (x1 != w.sorted_score[i1]) { // (x1 was raised: the range is narrower)
                int lo = i1 + 1, hi = p;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (w.sorted_score[mid] + equiv_range < x1) hi = mid; else lo = mid + 1;
                }
                p = lo;
            }
            if (p <= last) p = last + 1;
            const uint32_t h1 = w.sorted_hash[i1];
            for (; p < n; ++p) {
                if (w.removed[p]) continue;
                if (w.smooth[p] + equiv_range < x1) break;
                if (w.sorted_hash[p] != h1 || !equiv_with_pairs(t, rd.cals[w.order[i1]], rd.cals[w.order[p]])) continue;
                const Applied ap = apply_equivalent(t, rd, w, sh, i1, p);
                if (ap.s1_removed) break;
                if (ap.x1_raised) x1 = w.smooth[i1];
            }
        }
    }
    ln.sync();
    if (sh.any_excluded) {
        for (int i = ln.id; i < n; i += ln.width)
            if (w.removed[i]) w.flag[w.order[i]] |= 1;
        if (ln.id == 0)
            for (int i = 0; i < n; ++i) {
                if (w.removed[i]) continue;
                const int s = w.order[i];
                sh.max_score = rd.scores[s];
                sh.max_i = s;
                break;
            }
    }
    ln.sync();
}

SKC_HD inline int info_slot(const int ne, const int call, const bool present, const int which) { return (call * 2 + (present ? 1 : 0)) * ne + which; }
template <typename L>
SKC_HD inline void info_update(const L& ln, Shared& sh, const int call, const bool present, const int which, const double lnp) // :61-77
{
    long long* const p = &sh.info[info_slot(sh.ne, call, present, which)];
    const long long k = f64_key(lnp);
    if (*p >= k) return; // (entries only grow: a stale value read here costs an update that changes nothing, never one that is missed)
    ln.max_i64(p, k);
}
SKC_HD inline bool info_find(const Shared& sh, const int call, const bool present, const int which, double& out)
{
    const long long k = sh.info[info_slot(sh.ne, call, present, which)];
    if (k == SK3_NO_ENTRY) return false;
    out = key_f64(k);
    return true;
}

// score_indels :455-1079
template <typename L>
SKC_HD inline void score_indels(const L& ln, const Tab& t, const Opt& opt, const Read& rd, const Scratch& w, Shared& sh, Out& out)
{
    const int n = rd.n_cals;
    late_indel_normalization_filter(ln, t, opt, rd, w, sh);
    const int max_i = sh.max_i;
    const PCal& mc = rd.cals[max_i];

    // the indels to evaluate: candidates in the max alignment's range that some alignment holds with enough flank
    if (ln.id == 0) {
        const PRange mr = soft_clip_range(mc);
        PJob rj = {};
        rj.tab = t.tab;
        rj.n_tab = t.n_tab;
        rj.max_indel_size = t.max_indel_size;
        int lo, hi;
        range_iter(rj, mr.b, mr.e, lo, hi);
        sh.lo = lo;
        sh.hi = hi;
        sh.ne = 0;
    }
    ln.sync();
    const int lo = sh.lo, hi = sh.hi;
    for (int e = lo; e < hi; ++e) {
        const PIndel& k = t.tab[e];
        if (is_mismatch(k)) continue;
        if (ln.id == 0 && t.consulted) t.consulted[e] = 1;
        if (!k.cand) continue;
        const bool in_max = cal_has(mc, e);
        if (!in_max) { // the best-scoring other alignment with the indel (the first of equals)
            int best = -1;
            double best_score = 0;
            for (int i = ln.id; i < n; i += ln.width) {
                if (i == max_i) continue;
                if (w.flag[i] & 1) continue;
                if (!cal_has(rd.cals[i], e)) continue;
                if (best < 0 || rd.scores[i] > best_score) {
                    best_score = rd.scores[i];
                    best = i;
                }
            }
            sh.red_i[ln.id] = best;
            sh.red_score[ln.id] = best_score;
            ln.sync();
        }
        if (ln.id == 0 && sh.status == S3_OK) {
            int best = max_i;
            if (!in_max) {
                best = -1;
                double best_score = 0;
                for (int l = 0; l < ln.width; ++l) {
                    const int bi = sh.red_i[l];
                    if (bi < 0) continue;
                    const double bs = sh.red_score[l];
                    if (best < 0 || bs > best_score || (bs == best_score && bi < best)) {
                        best_score = bs;
                        best = bi;
                    }
                }
            }
            if (best >= 0) {
                int st = S3_OK;
                const int bpo = indel_bp_overlap(opt.upstream_oligo_size, rd.cals[best], k, st);
                if (st != S3_OK) {
                    sh.status = st;
                } else if (bpo < opt.min_read_bp_flank) {
                    if (bpo > 0) {
                        if (out.n_sub >= SUB_MAX) sh.status = S3_CAPACITY;
                        else out.sub[out.n_sub++] = t.orig[e];
                    }
                } else if (sh.ne >= E_MAX) {
                    sh.status = S3_CAPACITY;
                } else {
                    sh.to_eval[sh.ne++] = int16_t(e);
                }
            }
        }
        ln.sync();
    }
    if (sh.status != S3_OK) return;
    const int ne = sh.ne;
    // which evaluated indels conflict with which (positions in to_eval, ascending)
    if (ln.id == 0) {
        for (int i = 0; i < ne; ++i) sh.ortho[i] = 0;
        for (int i = 0; i < ne; ++i)
            for (int j = i + 1; j < ne; ++j)
                if (is_indel_conflict(t.tab[sh.to_eval[i]], t.tab[sh.to_eval[j]])) {
                    sh.ortho[i] |= 1u << j;
                    sh.ortho[j] |= 1u << i;
                }
    }
    for (int k = ln.id; k < 2 * ne * ne; k += ln.width) sh.info[k] = SK3_NO_ENTRY;
    // Once per read, for every table indel the alignments can hold: which evaluated indels it conflicts with (which_interfering_indel
    // skips mismatches: they get no bits) and which evaluated indel it is -- the per-alignment loop below reads these instead of
    // comparing table entries
    const int idx_min = int(-sh.idx_nlo), idx_span = int(sh.idx_hi) - idx_min + 1;
    const bool use_tables = (idx_span > 0 && idx_span <= CONF_SPAN && ne > 0 && !opt.no_tables);
    if (use_tables)
        for (int d = ln.id; d < idx_span; d += ln.width) {
            const PIndel& cur = t.tab[idx_min + d];
            uint32_t mask = 0;
            uint8_t slot = 0xff;
            for (int q = 0; q < ne; ++q) {
                if (sh.to_eval[q] == idx_min + d) slot = uint8_t(q);
                if (!is_mismatch(cur) && is_indel_conflict(cur, t.tab[sh.to_eval[q]])) mask |= 1u << q;
            }
            sh.conf_mask[d] = uint16_t(mask);
            sh.eval_slot[d] = slot;
        }
    ln.sync();
    auto interferes = [&](const int cur, const int q) -> bool { // a non-mismatch indel of the table against evaluated indel q
        return !is_mismatch(t.tab[cur]) && is_indel_conflict(t.tab[cur], t.tab[sh.to_eval[q]]);
    };

    // every alignment's contribution to the table (maxima: the alignments' order is immaterial)
    for (int ci = ln.id; ci < n; ci += ln.width) {
        if (w.flag[ci] & 1) continue;
        const PCal& c = rd.cals[ci];
        const double score = rd.scores[ci];
        if (use_tables) {
            uint32_t in_cal = 0, seen = 0; // evaluated indels the alignment holds / does not hold but interferes with
            for (int a = 0; a < c.n_indels; ++a) {
                const uint8_t slot = sh.eval_slot[c.indels[a] - idx_min];
                if (slot != 0xff) in_cal |= 1u << slot;
            }
            // the alignment's non-evaluated indels that are the first to interfere with an evaluated one (ascending, no repeats)
            int16_t noncand_ortho[E_MAX];
            int n_nco = 0;
            for (int a = 0; a < c.n_indels; ++a) {
                const int d = c.indels[a] - idx_min;
                const uint32_t first_for = sh.conf_mask[d] & ~in_cal & ~seen;
                if (first_for == 0) continue;
                seen |= first_for;
                if (sh.eval_slot[d] != 0xff) continue;
                if (n_nco >= E_MAX) {
                    sh.status = S3_CAPACITY;
                    break;
                }
                noncand_ortho[n_nco++] = c.indels[a];
            }
            for (int q = 0; q < ne; ++q) {
                const int e = sh.to_eval[q];
                if ((in_cal >> q) & 1u) {
                    info_update(ln, sh, q, true, q, score);
                    info_update(ln, sh, q, false, q, score + t.r2i[e]);
                    for (uint32_t om = sh.ortho[q]; om != 0; om &= om - 1) {
                        const int o = __builtin_ctz(om);
                        info_update(ln, sh, o, false, o, score + t.r2i[e]);
                        info_update(ln, sh, o, true, q, score);
                    }
                } else if ((seen >> q) & 1u) {
                    info_update(ln, sh, q, true, q, score + t.i2r[e]);
                } else {
                    info_update(ln, sh, q, false, q, score);
                    info_update(ln, sh, q, true, q, score + t.i2r[e]);
                }
            }
            for (int z = 0; z < n_nco; ++z) {
                const int nc = noncand_ortho[z];
                for (uint32_t qm = sh.conf_mask[nc - idx_min]; qm != 0; qm &= qm - 1)
                    info_update(ln, sh, __builtin_ctz(qm), false, __builtin_ctz(qm), score + t.r2i[nc]);
            }
            continue;
        }
        // (alignments whose indels span more of the table than the tables hold: the same from the table entries)
        uint32_t in_cal = 0;
        {
            int a = 0;
            for (int q = 0; q < ne; ++q) {
                while (a < c.n_indels && c.indels[a] < sh.to_eval[q]) ++a;
                if (a < c.n_indels && c.indels[a] == sh.to_eval[q]) in_cal |= 1u << q;
            }
        }
        // the alignment's non-evaluated indels that interfere with an evaluated one (ascending table order, no repeats)
        int16_t noncand_ortho[E_MAX];
        int n_nco = 0;
        bool overflow = false;
        for (int q = 0; q < ne; ++q) {
            const int e = sh.to_eval[q];
            if ((in_cal >> q) & 1u) {
                info_update(ln, sh, q, true, q, score);
                info_update(ln, sh, q, false, q, score + t.r2i[e]);
                for (int o = 0; o < ne; ++o)
                    if ((sh.ortho[q] >> o) & 1u) {
                        info_update(ln, sh, o, false, o, score + t.r2i[e]);
                        info_update(ln, sh, o, true, q, score);
                    }
            } else {
                int interfering = -1; // which_interfering_indel :100-119
                for (int a = 0; a < c.n_indels; ++a) {
                    const int cur = c.indels[a];
                    if (interferes(cur, q)) {
                        interfering = cur;
                        break;
                    }
                }
                if (interfering >= 0) {
                    bool evaluated = false;
                    for (int o = 0; o < ne; ++o)
                        if (sh.to_eval[o] == interfering) evaluated = true;
                    if (!evaluated) { // sorted insert without repeats
                        int at = 0;
                        while (at < n_nco && noncand_ortho[at] < interfering) ++at;
                        if (!(at < n_nco && noncand_ortho[at] == interfering)) {
                            if (n_nco >= E_MAX) {
                                overflow = true;
                            } else {
                                for (int z = n_nco; z > at; --z) noncand_ortho[z] = noncand_ortho[z - 1];
                                noncand_ortho[at] = int16_t(interfering);
                                ++n_nco;
                            }
                        }
                    }
                    info_update(ln, sh, q, true, q, score + t.i2r[e]);
                } else {
                    info_update(ln, sh, q, false, q, score);
                    info_update(ln, sh, q, true, q, score + t.i2r[e]);
                }
            }
        }
        if (overflow) sh.status = S3_CAPACITY;
        for (int z = 0; z < n_nco; ++z) {
            const int nc = noncand_ortho[z];
            for (int q = 0; q < ne; ++q) {
                if (!interferes(nc, q)) continue;
                info_update(ln, sh, q, false, q, score + t.r2i[nc]);
            }
        }
    }
    ln.sync();
    if (ln.id != 0 || sh.status != S3_OK) return;

    const double max_score = sh.max_score;
    const bool tier1 = (rd.map_level == SK_MAPLEVEL_TIER1);
    for (int q = 0; q < ne; ++q) {
        const int e = sh.to_eval[q];
        const PIndel& k = t.tab[e];
        const bool in_max = cal_has(mc, e);
        double indel_score = max_score;
        if (!in_max && !info_find(sh, q, true, q, indel_score)) continue;
        double ref_score = 0;
        if (!info_find(sh, q, false, q, ref_score)) continue;
        const int32_t rb = k.pos - 1, re = right_pos(k) + 1;
        const int32_t read_pos = lowest_fwd_read_pos(mc, mc.fwd != 0, rb, re);
        int32_t edge_dist = rd.read_length;
        {
            const int32_t rev_pos = lowest_fwd_read_pos(mc, !(mc.fwd != 0), rb, re);
            if (read_pos >= 0) edge_dist = read_pos;
            if (rev_pos >= 0 && rev_pos < edge_dist) edge_dist = rev_pos;
        }
        sk_read_path_scores s;
        for (unsigned z = 0; z < sizeof(s); ++z) reinterpret_cast<unsigned char*>(&s)[z] = 0;
        s.indel = t.orig[e];
        s.ref_lnp = static_cast<float>(ref_score);
        s.indel_lnp = static_cast<float>(indel_score);
        s.non_ambig = uint16_t(rd.non_ambig);
        s.read_length = uint16_t(rd.read_length);
        s.is_tier1_read = tier1 ? 1 : 0;
        s.is_fwd_strand = mc.fwd ? 1 : 0;
        s.read_pos = int16_t(read_pos);
        s.distance_from_closest_read_edge = int16_t(edge_dist);
        for (int oq = 0; oq < ne; ++oq) {
            if (!((sh.ortho[q] >> oq) & 1u)) continue;
            double alt_score;
            if (!info_find(sh, q, true, oq, alt_score)) continue;
            const float a = static_cast<float>(alt_score); // ReadPathScores::insertAlt, IndelData.cpp:40-68: keep the two best
            if (s.n_alt < 2) {
                s.alt_indel[s.n_alt] = t.orig[sh.to_eval[oq]];
                s.alt_lnp[s.n_alt] = a;
                s.n_alt++;
            } else {
                int min_index = 2;
                float mn = a;
                for (int qq = 0; qq < 2; ++qq)
                    if (s.alt_lnp[qq] < mn) {
                        mn = s.alt_lnp[qq];
                        min_index = qq;
                    }
                if (min_index < 2) {
                    s.alt_indel[min_index] = t.orig[sh.to_eval[oq]];
                    s.alt_lnp[min_index] = a;
                }
            }
        }
        out.scores[out.n_scores++] = s;
    }
}

// stage 3 of one read, by the lanes of `ln` (every lane makes this call with the same arguments)
template <typename L>
SKC_HD inline void finish_read(const L& ln, const Tab& t, const Opt& opt, const Read& rd, const Scratch& w, Shared& sh, Out& out)
{
    if (ln.id == 0) {
        sh.status = S3_OK;
        sh.idx_hi = -1;
        sh.idx_nlo = -0x7fffffff;
        out.status = S3_OK;
        out.is_realigned = 0;
        out.realign_pos = 0;
        out.n_seg = 0;
        out.max_score = 0;
        out.n_scores = 0;
        out.n_sub = 0;
    }
    ln.sync();
    if (rd.n_cals <= 0) return;
    select_alignments(ln, t, opt, rd, w, sh, out);
    if (sh.status == S3_OK && (rd.map_level == SK_MAPLEVEL_TIER1 || rd.map_level == SK_MAPLEVEL_TIER2)) // is_tier1or2_mapping :1800
        score_indels(ln, t, opt, rd, w, sh, out);
    ln.sync();
    if (ln.id == 0) out.status = sh.status;
}

} // namespace sk3
