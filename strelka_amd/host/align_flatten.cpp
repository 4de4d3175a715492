// align_flatten.cpp -- host adapter of hot path A: (read, reference segment, CandidateAlignment set) -> sk_align_batch.
//
// Mirrors the control flow of scoreCandidateAlignment (L/starling_common/starling_read_align_score.cpp:261-499) but
// emits scoring ops instead of adding likelihood terms; the arithmetic itself happens in score_alignments.hip.
// Everything here is integer/byte work and must reproduce the reference's indexing exactly (which reference base or
// insert base each read base is compared with).

#include "strelka_amd.h"

#include "../csrc/align_entry.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstring>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

namespace
{

inline bool is_align_match(const uint32_t t) { return t == SK_SEG_MATCH || t == SK_SEG_SEQ_MATCH || t == SK_SEG_SEQ_MISMATCH; }
inline bool is_type_indel(const uint32_t t) { return t == SK_SEG_INSERT || t == SK_SEG_DELETE; }

// get_bam_seq_code, L/htsapi/bam_seq.hh:73-92
inline uint8_t code_of(const char c)
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

struct Error
{
    std::string msg;
};

} // namespace

// reference reads of the read path that fell outside the segment the caller gave (they read as 'N', reference_contig_segment::get_base):
// a caller that hands a job a WINDOW of its contig segment learns from this whether the window was wide enough
// (sk_realign_reference_reads_outside, strelka_amd.h)
std::atomic<int64_t> g_sk_ref_reads_outside{ 0 };

struct sk_align_builder
{
    std::vector<int64_t> read_off{ 0 }, hap_off{ 0 }, op_off{ 0 };
    std::vector<int32_t> cal_off{ 0 };
    std::vector<uint8_t> read_code, read_qual, hap_code;
    std::vector<sk_score_op> ops;
    std::vector<uint32_t> entries, evmask; // prepared form, filled by finish
    std::vector<uint32_t> colmat, addmask;
    std::vector<int64_t> colmat_off;
    int32_t max_read_len = 0, max_hap_len = 0;
    int32_t host_threads = 1; // budget of finish()'s prepare step (sk_align_builder_set_host_threads)
    std::string error;

    void clear()
    {
        read_off.assign(1, 0);
        hap_off.assign(1, 0);
        op_off.assign(1, 0);
        cal_off.assign(1, 0);
        read_code.clear();
        read_qual.clear();
        hap_code.clear();
        ops.clear();
        max_read_len = max_hap_len = 0;
        error.clear();
    }
};

namespace
{

// is_segment_swap_start, L/blt_util/align_path.cpp:868-895
bool is_segment_swap_start(const sk_path_seg* path, const int n, int i)
{
    bool is_insert = false, is_delete = false;
    for (; i < n; ++i) {
        if (path[i].type == SK_SEG_INSERT) is_insert = true;
        else if (path[i].type == SK_SEG_DELETE) is_delete = true;
        else break;
    }
    return is_insert && is_delete;
}

// getMatchingIndelKey, starling_read_align_score.cpp:177-228
const sk_indel_key* matching_indel(const sk_candidate_alignment& cal, const int32_t ref_head_pos, const unsigned del_len,
                                   const unsigned ins_len, const int ends_first, const int ends_second,
                                   const int path_index)
{
    if (path_index < ends_first) return &cal.leading;
    if (path_index > ends_second) return &cal.trailing;
    const sk_indel_key* found = nullptr;
    for (int k = 0; k < cal.n_indels; ++k) {
        const sk_indel_key& ci = cal.indels[k];
        if (ci.pos == ref_head_pos && (ci.type == SK_INDEL_INDEL || ci.type == SK_INDEL_MISMATCH) &&
            ci.del_len == del_len && ci.ins_len == ins_len) {
            if (found) throw Error{ "candidate alignment holds two indels matching one CIGAR gap" };
            found = &ci;
        } else if (ci.pos > ref_head_pos) {
            break;
        }
    }
    if (!found) throw Error{ "candidate alignment CIGAR gap has no matching indel key" };
    return found;
}

struct ReadPool
{
    // the read's reference window [win_begin, win_end) followed by insert sequences
    int32_t win_begin = 0, win_end = 0;
    std::vector<uint8_t> bytes;
    struct InsEntry
    {
        const char* seq;
        uint32_t len;
        int32_t off;
    };
    std::vector<InsEntry> ins;

    // offset of insert bases [head, head+len) of `key` (positions outside the sequence read as 'N',
    // string_bam_seq::get_char, L/htsapi/bam_seq.hh:268-273)
    int32_t insert_src(const sk_indel_key& key, const int32_t head, const uint32_t len)
    {
        if (head >= 0 && uint32_t(head) + len <= key.ins_len) {
            for (const InsEntry& e : ins)
                if (e.len == key.ins_len && (e.seq == key.ins_seq || std::memcmp(e.seq, key.ins_seq, e.len) == 0)) return e.off + head;
            const int32_t off = int32_t(bytes.size());
            for (uint32_t i = 0; i < key.ins_len; ++i) bytes.push_back(code_of(key.ins_seq[i]));
            ins.push_back(InsEntry{ key.ins_seq, key.ins_len, off });
            return off + head;
        }
        const int32_t off = int32_t(bytes.size());
        for (uint32_t i = 0; i < len; ++i) {
            const int64_t p = int64_t(head) + i;
            bytes.push_back((p >= 0 && p < int64_t(key.ins_len)) ? code_of(key.ins_seq[p]) : uint8_t(SK_BAM_ANY));
        }
        return off;
    }
};

void flatten_one(const sk_candidate_alignment& cal, const int32_t read_len, ReadPool& pool, std::vector<sk_score_op>& ops)
{
    const sk_path_seg* path = cal.path;
    const int aps = cal.n_seg;
    unsigned read_offset = 0;
    int32_t ref_head_pos = cal.pos;

    int ends_first = aps, ends_second = aps; // get_match_edge_segments, align_path.cpp:735-752
    {
        bool is_first_match = false;
        for (int i = 0; i < aps; ++i)
            if (is_align_match(path[i].type)) {
                if (!is_first_match) ends_first = i;
                is_first_match = true;
                ends_second = i;
            }
    }
    auto emit = [&](const uint8_t kind, const uint32_t len, const int32_t src, const bool penalty) {
        if (len > 0xffffu) throw Error{ "path segment longer than 65535" };
        if (kind == SK_OP_NOBASE && !penalty) return;
        ops.push_back(sk_score_op{ uint16_t(len), kind, uint8_t(penalty ? SK_OPFLAG_NONCANDIDATE_PENALTY : 0), src });
    };

    int path_index = 0;
    while (path_index < aps) {
        const bool is_swap_start = is_segment_swap_start(path, aps, path_index);
        unsigned n_seg = 1;
        const sk_path_seg& ps = path[path_index];

        if (is_swap_start || ps.type == SK_SEG_SEQ_MISMATCH) {
            unsigned del_len, ins_len;
            if (ps.type == SK_SEG_SEQ_MISMATCH) {
                del_len = ins_len = ps.length;
            } else { // swap_info, align_path_util.hh:75-106
                int k = path_index;
                del_len = ins_len = 0;
                for (; k < aps && is_type_indel(path[k].type); ++k) {
                    if (path[k].type == SK_SEG_INSERT) ins_len += path[k].length;
                    else del_len += path[k].length;
                }
                n_seg = unsigned(k - path_index);
            }
            const sk_indel_key* key = matching_indel(cal, ref_head_pos, del_len, ins_len, ends_first, ends_second, path_index);
            if (key->type == SK_INDEL_NONE) throw Error{ "edge swap without leading/trailing indel key" };
            int32_t head = 0;
            if (path_index < ends_first) head = int32_t(key->ins_len) - int32_t(ps.length);
            const bool pen = !key->is_candidate;
            if (ins_len > 0) emit(SK_OP_BASES, ins_len, pool.insert_src(*key, head, ins_len), pen);
            else emit(SK_OP_NOBASE, 0, 0, pen);
        } else if (is_align_match(ps.type)) {
            emit(SK_OP_BASES, ps.length, ref_head_pos - pool.win_begin, false);
        } else if (ps.type == SK_SEG_INSERT) {
            const sk_indel_key* key = matching_indel(cal, ref_head_pos, 0, ps.length, ends_first, ends_second, path_index);
            if (key->type == SK_INDEL_NONE) throw Error{ "edge insertion without leading/trailing indel key" };
            int32_t head = 0;
            if (path_index < ends_first) head = int32_t(key->ins_len) - int32_t(ps.length);
            emit(SK_OP_BASES, ps.length, pool.insert_src(*key, head, ps.length), !key->is_candidate);
        } else if (ps.type == SK_SEG_DELETE) {
            const sk_indel_key* key = matching_indel(cal, ref_head_pos, ps.length, 0, ends_first, ends_second, path_index);
            if (key->type == SK_INDEL_NONE) throw Error{ "edge deletion without leading/trailing indel key" };
            emit(SK_OP_NOBASE, 0, 0, !key->is_candidate);
        } else if (ps.type == SK_SEG_SKIP || ps.type == SK_SEG_HARD_CLIP) {
            // nothing
        } else if (ps.type == SK_SEG_SOFT_CLIP) {
            emit(SK_OP_SOFT_CLIP, ps.length, 0, false);
        } else {
            throw Error{ "Can't handle cigar code" };
        }

        for (unsigned i = 0; i < n_seg; ++i) { // increment_path, align_path_util.hh:38-68
            const sk_path_seg& s = path[path_index];
            if (is_align_match(s.type)) {
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
    if (int64_t(read_offset) != int64_t(read_len)) throw Error{ "candidate alignment path does not span the read" };
}

} // namespace

extern "C" {

sk_align_builder* sk_align_builder_create(void) { return new sk_align_builder(); }
void sk_align_builder_destroy(sk_align_builder* b) { delete b; }
void sk_align_builder_clear(sk_align_builder* b)
{
    if (b) b->clear();
}

// append every read of `src` after the reads of `dst` (batches built on several threads are joined this way: op sources are
// relative to their read's haplotype pool, so only the CSR offsets shift)
int sk_align_builder_append(sk_align_builder* dst, const sk_align_builder* src)
{
    if (!dst || !src) return 1;
    const int64_t rb = dst->read_off.back(), hb = dst->hap_off.back(), ob = dst->op_off.back();
    const int32_t cb = dst->cal_off.back();
    for (size_t i = 1; i < src->read_off.size(); ++i) dst->read_off.push_back(rb + src->read_off[i]);
    for (size_t i = 1; i < src->hap_off.size(); ++i) dst->hap_off.push_back(hb + src->hap_off[i]);
    for (size_t i = 1; i < src->cal_off.size(); ++i) dst->cal_off.push_back(cb + src->cal_off[i]);
    for (size_t i = 1; i < src->op_off.size(); ++i) dst->op_off.push_back(ob + src->op_off[i]);
    dst->read_code.insert(dst->read_code.end(), src->read_code.begin(), src->read_code.end());
    dst->read_qual.insert(dst->read_qual.end(), src->read_qual.begin(), src->read_qual.end());
    dst->hap_code.insert(dst->hap_code.end(), src->hap_code.begin(), src->hap_code.end());
    dst->ops.insert(dst->ops.end(), src->ops.begin(), src->ops.end());
    dst->max_read_len = std::max(dst->max_read_len, src->max_read_len);
    dst->max_hap_len = std::max(dst->max_hap_len, src->max_hap_len);
    return 0;
}

// error text of the last failing builder call (the GPU library's sk_last_error covers the device entry points)
const char* sk_align_builder_error(const sk_align_builder* b) { return b ? b->error.c_str() : "null builder"; }

int sk_align_builder_add_read(sk_align_builder* b, const uint8_t* read_code, const uint8_t* read_qual,
                              const int32_t read_len, const char* ref_seq, const int32_t ref_offset,
                              const int32_t ref_len, const sk_candidate_alignment* cals, const int32_t n_cals)
{
    if (!b || read_len < 0 || n_cals < 0 || (read_len && (!read_code || !read_qual))) return 1;
    try {
        for (int32_t i = 0; i < read_len; ++i)
            if (read_qual[i] > 70) // qphred_cache::high_qscore_error, L/blt_util/qscore_cache.cpp:66-75
                throw Error{ "Attempting to lookup basecall quality score " + std::to_string(int(read_qual[i])) +
                             " which exceeds the maximum cached basecall quality score of 70" };
        // reference window: union of the reference spans of all match segments of all candidates
        int32_t wb = INT_MAX, we = INT_MIN;
        for (int32_t c = 0; c < n_cals; ++c) {
            int32_t pos = cals[c].pos;
            for (int i = 0; i < cals[c].n_seg; ++i) {
                const sk_path_seg& s = cals[c].path[i];
                if (is_align_match(s.type)) {
                    wb = std::min(wb, pos);
                    we = std::max(we, pos + int32_t(s.length));
                    pos += int32_t(s.length);
                } else if (s.type == SK_SEG_DELETE || s.type == SK_SEG_SKIP) {
                    pos += int32_t(s.length);
                }
            }
        }
        ReadPool pool;
        if (wb > we) wb = we = 0;
        pool.win_begin = wb;
        pool.win_end = we;
        pool.bytes.resize(size_t(we - wb));
        {
            int64_t outside = 0;
            for (int32_t p = wb; p < we; ++p) { // reference_contig_segment::get_base, :46-51
                const bool out = (p < ref_offset || p >= ref_offset + ref_len);
                outside += out ? 1 : 0;
                pool.bytes[size_t(p - wb)] = out ? uint8_t(SK_BAM_ANY) : code_of(ref_seq[p - ref_offset]);
            }
            if (outside) g_sk_ref_reads_outside.fetch_add(outside, std::memory_order_relaxed);
        }

        const size_t ops_mark = b->ops.size(), off_mark = b->op_off.size();
        try {
            for (int32_t c = 0; c < n_cals; ++c) {
                flatten_one(cals[c], read_len, pool, b->ops);
                b->op_off.push_back(int64_t(b->ops.size()));
            }
        } catch (...) {
            b->ops.resize(ops_mark);
            b->op_off.resize(off_mark);
            throw;
        }
        b->read_code.insert(b->read_code.end(), read_code, read_code + read_len);
        b->read_qual.insert(b->read_qual.end(), read_qual, read_qual + read_len);
        b->read_off.push_back(int64_t(b->read_code.size()));
        if (pool.bytes.empty()) pool.bytes.push_back(SK_BAM_ANY);
        b->hap_code.insert(b->hap_code.end(), pool.bytes.begin(), pool.bytes.end());
        b->hap_off.push_back(int64_t(b->hap_code.size()));
        b->cal_off.push_back(b->cal_off.back() + n_cals);
        b->max_read_len = std::max(b->max_read_len, read_len);
        b->max_hap_len = std::max<int32_t>(b->max_hap_len, int32_t(pool.bytes.size()));
        return 0;
    } catch (const Error& e) {
        b->error = e.msg;
        return 1;
    } catch (const std::exception& e) {
        b->error = e.what();
        return 1;
    }
}

int32_t sk_align_evmask_words(const int32_t max_read_len) { return sk_ent_evmask_words(max_read_len < 0 ? 0 : max_read_len); }

// ops -> transition entries + per-read event masks (layout: csrc/align_entry.h).  `host_threads`: 1 = this thread only
// (the default everywhere: the reference runs one process per core), n = up to n threads, 0 = up to 16 hardware threads;
// reads are independent (each owns its entry slots and its mask words), so large batches split by read.
static int align_prepare(const sk_align_batch* b, uint32_t* entries, uint32_t* evmask, const int host_threads)
{
    if (!b || !entries || !evmask || b->n_reads < 0) return 1;
    const int W = sk_ent_evmask_words(b->max_read_len);
    const int64_t budget = (host_threads == 0) ? int64_t(std::min(16u, std::max(1u, std::thread::hardware_concurrency())))
                                               : int64_t(std::max(1, host_threads));
    const int threads = int(std::max<int64_t>(1, std::min<int64_t>(budget, int64_t(b->n_reads) / 4096)));
    auto slice = [&](const int r_begin, const int r_end) {
    for (int r = r_begin; r < r_end; ++r) {
        const int64_t L64 = b->read_off[r + 1] - b->read_off[r], P64 = b->hap_off[r + 1] - b->hap_off[r];
        uint32_t* mask = evmask + int64_t(r) * W;
        std::memset(mask, 0, sizeof(uint32_t) * size_t(W));
        const bool read_ok = (L64 >= 0 && L64 <= SK_ENT_MAX_READ_LEN && L64 <= b->max_read_len && P64 >= 0 && P64 <= SK_ENT_MAX_POOL);
        const int L = int(L64), P = int(P64);
        const uint8_t* hap = b->hap_code + b->hap_off[r];
        auto col_at = [&](const int idx) -> unsigned {
            return (idx >= 0 && idx < P) ? sk_ent_col_index(hap[idx]) : unsigned(SK_ENT_ZERO_COL);
        };
        for (int c = b->cal_off[r]; c < b->cal_off[r + 1]; ++c) {
            const int64_t k0 = b->op_off[c], k1 = b->op_off[c + 1];
            uint32_t* ent = entries + k0 + 2 * int64_t(c);
            const int nslots = int(k1 - k0) + 2;
            for (int i = 0; i < nslots; ++i) ent[i] = SK_ENT_END;
            bool complex_cal = !read_ok;
            int e = 0, pos = 0;
            unsigned npen = 0;
            auto emit = [&](const unsigned np, const bool clip, const int hidx) {
                if (np > 7u || hidx + SK_ENT_HIDX_BIAS < 0 || hidx + SK_ENT_HIDX_BIAS > 2047) complex_cal = true;
                if (complex_cal) return;
                ent[e++] = unsigned(pos) | (np << 10) | (clip ? 1u << 13 : 0u) | (col_at(hidx + pos) << 15) |
                           (col_at(hidx + pos + 1) << 18) | (unsigned(hidx + SK_ENT_HIDX_BIAS) << 21);
                if (pos > 0 && pos <= L) mask[pos >> 5] |= 1u << (pos & 31);
            };
            for (int64_t k = k0; k < k1 && !complex_cal; ++k) {
                const sk_score_op& op = b->ops[k];
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
                    npen += pen; // NOBASE / zero-length ops carry only their penalty (0 * ln 1/4 adds exactly nothing)
                }
            }
            if (!complex_cal) {
                if (pos != L || pos >= int(SK_ENT_END)) complex_cal = true; // does not span the read
                else emit(npen, false, P - pos);
            }
            if (complex_cal) {
                for (int i = 0; i < nslots; ++i) ent[i] = SK_ENT_END;
                ent[0] = SK_ENT_COMPLEX;
            }
        }
    }
    };
    if (threads <= 1) {
        slice(0, b->n_reads);
    } else {
        std::vector<std::thread> pool;
        std::vector<int> inline_slices{ 0 };
        auto lo = [&](const int t) { return int(int64_t(b->n_reads) * t / threads); };
        for (int t = 1; t < threads; ++t) {
            try {
                pool.emplace_back(slice, lo(t), lo(t + 1));
            } catch (const std::system_error&) { // no more threads to be had: this slice runs here
                inline_slices.push_back(t);
            }
        }
        for (const int t : inline_slices) slice(lo(t), lo(t + 1));
        for (auto& th : pool) th.join();
    }
    return 0;
}

int sk_align_prepare(const sk_align_batch* b, uint32_t* entries, uint32_t* evmask) { return align_prepare(b, entries, evmask, 1); }

int64_t sk_align_colmat_words(const sk_align_batch* b)
{
    if (!b || b->n_reads < 0) return -1;
    int64_t n = 0;
    for (int r = 0; r < b->n_reads; ++r)
        n += ((b->read_off[r + 1] - b->read_off[r] + 7) / 8) * int64_t(b->cal_off[r + 1] - b->cal_off[r]);
    return n;
}

// the column form (layout: strelka_amd.h, sk_align_batch::colmat): which table column each read position of each candidate
// alignment faces -- the walk of align_prepare above with every position written out
int sk_align_prepare_cols(const sk_align_batch* b, uint32_t* colmat, int64_t* colmat_off, uint32_t* addmask)
{
    if (!b || !colmat || !colmat_off || !addmask || !b->entries || b->n_reads < 0) return 1;
    const int W = sk_ent_evmask_words(b->max_read_len);
    colmat_off[0] = 0;
    for (int r = 0; r < b->n_reads; ++r)
        colmat_off[r + 1] = colmat_off[r] + ((b->read_off[r + 1] - b->read_off[r] + 7) / 8) * int64_t(b->cal_off[r + 1] - b->cal_off[r]);
    const uint8_t zero_byte = uint8_t(SK_SEL_NONE | (SK_SEL_NONE << 4)); // two positions that add nothing
    for (int r = 0; r < b->n_reads; ++r) {
        const int L = int(b->read_off[r + 1] - b->read_off[r]), P = int(b->hap_off[r + 1] - b->hap_off[r]);
        const int ncr = b->cal_off[r + 1] - b->cal_off[r], nch = (L + 7) / 8;
        uint32_t* am = addmask + int64_t(r) * W;
        std::memset(am, 0, sizeof(uint32_t) * size_t(W));
        uint8_t* cm = reinterpret_cast<uint8_t*>(colmat + colmat_off[r]);
        std::memset(cm, zero_byte, size_t(4) * size_t(nch) * size_t(ncr));
        const uint8_t* hap = b->hap_code + b->hap_off[r];
        const uint8_t* read = b->read_code + b->read_off[r];
        for (int j = 0; j < ncr; ++j) {
            const int c = b->cal_off[r] + j;
            const uint32_t* ent = b->entries + b->op_off[c] + 2 * int64_t(c);
            if (ent[0] == SK_ENT_COMPLEX) { // scored by the generic routine; the mask's last bit tells the kernel the read has one
                am[W - 1] |= 1u << 31;
                continue;
            }
            for (int e = 0; (ent[e] & SK_ENT_POS_MASK) != SK_ENT_END; ++e)
                if (ent[e] & SK_ENT_ADD_BITS) {
                    const unsigned p = ent[e] & SK_ENT_POS_MASK;
                    am[p >> 5] |= 1u << (p & 31);
                }
            int pos = 0;
            // (selectors first, then the penalty flags on top of them)
            for (int64_t k = b->op_off[c]; k < b->op_off[c + 1]; ++k) {
                const sk_score_op& op = b->ops[k];
                const int len = int(op.length);
                if (!((op.kind == SK_OP_BASES || op.kind == SK_OP_SOFT_CLIP) && len > 0)) continue;
                for (int t = 0; t < len && pos + t < L; ++t) {
                    unsigned col = SK_ENT_ZERO_COL;
                    if (op.kind == SK_OP_BASES) {
                        const int idx = int(op.src) + t;
                        col = (idx >= 0 && idx < P) ? sk_ent_col_index(hap[idx]) : unsigned(SK_ENT_ZERO_COL);
                    }
                    const int i = pos + t;
                    const unsigned sel = sk_col_selector(col, read[i]);
                    uint8_t& byte = cm[(size_t(i >> 3) * size_t(ncr) + size_t(j)) * 4 + size_t(i & 3)];
                    byte = (i & 4) ? uint8_t((byte & 0x0fu) | (sel << 4)) : uint8_t((byte & 0xf0u) | sel);
                }
                pos += len;
            }
            // one non-candidate-indel penalty and nothing else is the common case of an entry that adds terms: it is flagged
            // in the position's own nibble (bit 2), and a read whose candidates have nothing but such entries is swept without
            // its entry lists; anything else (several penalties, a soft clip, a penalty after the last word) sets bit
            // 32 * W - 2 of the read's add mask: that read's entries are consulted, its flags ignored
            for (int e = 0; (ent[e] & SK_ENT_POS_MASK) != SK_ENT_END; ++e) {
                if (!(ent[e] & SK_ENT_ADD_BITS)) continue;
                const int i = int(ent[e] & SK_ENT_POS_MASK);
                const bool simple = ((ent[e] >> 10) & 7u) == 1u && !(ent[e] & (1u << 13)) && i < 8 * nch;
                if (!simple) {
                    am[W - 1] |= 1u << 30;
                    continue;
                }
                uint8_t& byte = cm[(size_t(i >> 3) * size_t(ncr) + size_t(j)) * 4 + size_t(i & 3)];
                byte |= (i & 4) ? 0x40u : 0x04u;
            }
        }
    }
    return 0;
}

int sk_align_builder_set_host_threads(sk_align_builder* b, const int32_t host_threads)
{
    if (!b || host_threads < 0) return 1;
    b->host_threads = host_threads;
    return 0;
}

int sk_align_builder_finish(sk_align_builder* b, sk_align_batch* out)
{
    if (!b || !out) return 1;
    out->n_reads = int32_t(b->read_off.size() - 1);
    out->n_cals = b->cal_off.back();
    out->n_ops = int64_t(b->ops.size());
    out->read_off = b->read_off.data();
    out->read_code = b->read_code.data();
    out->read_qual = b->read_qual.data();
    out->hap_off = b->hap_off.data();
    out->hap_code = b->hap_code.data();
    out->cal_off = b->cal_off.data();
    out->op_off = b->op_off.data();
    out->ops = b->ops.data();
    out->max_read_len = b->max_read_len;
    out->max_hap_len = b->max_hap_len;
    // the device-ready form of the ops
    out->evmask_words = sk_ent_evmask_words(b->max_read_len);
    b->entries.assign(b->ops.size() + 2 * size_t(out->n_cals) + 1, SK_ENT_END);
    b->evmask.assign(size_t(out->n_reads) * size_t(out->evmask_words) + 1, 0u);
    out->entries = nullptr;
    out->evmask = nullptr;
    out->colmat = nullptr;
    out->colmat_off = nullptr;
    out->addmask = nullptr;
    if (align_prepare(out, b->entries.data(), b->evmask.data(), b->host_threads)) return 1;
    out->entries = b->entries.data();
    out->evmask = b->evmask.data();
    // the column form
    out->colmat = nullptr;
    out->colmat_off = nullptr;
    out->addmask = nullptr;
    b->colmat.assign(size_t(sk_align_colmat_words(out)) + 1, 0u);
    b->colmat_off.assign(size_t(out->n_reads) + 1, 0);
    b->addmask.assign(size_t(out->n_reads) * size_t(out->evmask_words) + 1, 0u);
    if (sk_align_prepare_cols(out, b->colmat.data(), b->colmat_off.data(), b->addmask.data())) return 1;
    out->colmat = b->colmat.data();
    out->colmat_off = b->colmat_off.data();
    out->addmask = b->addmask.data();
    return 0;
}

} // extern "C"
