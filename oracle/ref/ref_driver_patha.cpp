// ref_driver_patha.cpp -- C entry points over the REFERENCE's own read-realignment code (hot path A).
//
// TEST INFRASTRUCTURE ONLY; contains no reference code.  Like the reference's own unit test
// (L/starling_common/test/starling_read_align_test.cpp:22) this TU #includes starling_read_align.cpp where it lies to
// reach its file-static functions (make_start_pos_alignment, get_end_pin_start_pos, getCandidateAlignments,
// scoreCandidateAlignments ...).

#include "starling_common/starling_read_align.cpp"

#include "htsapi/align_path_bam_util.hh"
#include "starling_common/starling_read.hh"
#include "starling_common/starling_read_align_score.hh"
#include "test/starling_base_options_test.hh"

#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace
{

struct RefIndel
{
    int32_t pos;
    int32_t type;
    uint32_t del_len;
    uint32_t ins_len;
    const char* ins_seq;
    int32_t is_candidate;
};

IndelKey to_key(const RefIndel& r)
{
    return IndelKey(r.pos, static_cast<INDEL::index_t>(r.type), r.del_len, r.ins_seq ? std::string(r.ins_seq, r.ins_len).c_str() : "");
}

void from_key(const IndelKey& k, int32_t* pos, int32_t* type, uint32_t* del_len, char* ins, int ins_cap)
{
    *pos = k.pos;
    *type = k.type;
    *del_len = k.deletionLength;
    std::strncpy(ins, k.insertSequence.c_str(), ins_cap - 1);
    ins[ins_cap - 1] = 0;
}

// options + derived options are built once per mode (their construction parses the built-in indel error / theta models)
struct Mode
{
    starling_base_options_test opt;
    std::unique_ptr<starling_base_deriv_options> dopt;
    explicit Mode(const bool is_somatic)
    {
        if (is_somatic) opt.randomBaseMatchProb = 0.5;
        opt.is_candidate_indel_signal_test = false;
        opt.isHaplotypingEnabled = false;
        dopt.reset(new starling_base_deriv_options(opt));
    }
};

Mode& mode(const bool is_somatic)
{
    static Mode germline(false), somatic(true);
    return is_somatic ? somatic : germline;
}

struct PreparedScoreRead
{
    std::unique_ptr<starling_read> sread;
    std::vector<CandidateAlignment> cals;
};
struct PreparedRealignRead
{
    std::unique_ptr<starling_read> sread;
    known_pos_range range;
    PreparedRealignRead() : range(0, 0) {}
};

struct Session
{
    std::vector<PreparedScoreRead> scoreReads;     // ref_session_prepare_score_read / ref_session_time_score
    std::vector<PreparedRealignRead> realignReads; // ref_session_prepare_realign_read / ref_session_time_realign
    Mode& m;
    starling_base_options_test& opt;
    std::unique_ptr<starling_base_deriv_options>& dopt;
    std::unique_ptr<starling_sample_options> sopt;
    explicit Session(const bool is_somatic) : m(mode(is_somatic)), opt(m.opt), dopt(m.dopt) {}
    reference_contig_segment ref;
    std::unique_ptr<IndelBuffer> buffer;
    depth_buffer db, db2;
    unsigned next_read_id = 1;
};

} // namespace

extern "C" {

/// make_start_pos_alignment (starling_read_align.cpp:394-584).  Returns 0 on success, 1 when the reference throws.
int ref_make_start_pos_alignment(int ref_start_pos, int read_start_pos, int is_fwd, unsigned read_length, int n_indels,
                                 const RefIndel* indels, int* out_pos, char* out_cigar, int cigar_cap, int32_t* lead,
                                 char* lead_ins, int32_t* trail, char* trail_ins, int ins_cap)
{
    try {
        indel_set_t iset;
        for (int i = 0; i < n_indels; ++i) iset.insert(to_key(indels[i]));
        const CandidateAlignment cal(make_start_pos_alignment(ref_start_pos, read_start_pos, is_fwd != 0, read_length, iset));
        *out_pos = cal.al.pos;
        const std::string cigar(ALIGNPATH::apath_to_cigar(cal.al.path));
        std::strncpy(out_cigar, cigar.c_str(), cigar_cap - 1);
        out_cigar[cigar_cap - 1] = 0;
        uint32_t dl;
        from_key(cal.leading_indel_key, &lead[0], &lead[1], &dl, lead_ins, ins_cap);
        lead[2] = int32_t(dl);
        from_key(cal.trailing_indel_key, &trail[0], &trail[1], &dl, trail_ins, ins_cap);
        trail[2] = int32_t(dl);
        return 0;
    } catch (...) {
        return 1;
    }
}

/// get_end_pin_start_pos (starling_read_align.cpp:594-719)
int ref_get_end_pin_start_pos(int n_indels, const RefIndel* indels, unsigned read_length, int ref_end_pos,
                              int read_end_pos, int* ref_start_pos, int* read_start_pos)
{
    try {
        indel_set_t iset;
        for (int i = 0; i < n_indels; ++i) iset.insert(to_key(indels[i]));
        pos_t rsp(0), rdp(0);
        get_end_pin_start_pos(iset, read_length, ref_end_pos, read_end_pos, rsp, rdp);
        *ref_start_pos = rsp;
        *read_start_pos = rdp;
        return 0;
    } catch (...) {
        return 1;
    }
}

/// a reference segment + an IndelBuffer with one sample, as in starling_read_align_test.cpp:345-372
void* ref_session_create(const char* ref_seq, int ref_offset, int is_somatic)
{
    Session* s = new Session(is_somatic != 0);
    s->sopt.reset(new starling_sample_options(s->opt));
    s->ref.seq() = ref_seq;
    s->ref.set_offset(ref_offset);
    s->buffer.reset(new IndelBuffer(s->opt, *s->dopt, s->ref));
    s->buffer->registerSample(s->db, s->db2, false);
    s->buffer->finalizeSamples();
    return s;
}

void ref_session_destroy(void* p) { delete static_cast<Session*>(p); }

/// insert an indel and force its candidate status (IndelData::status is the cache isCandidateIndel consults,
/// IndelBuffer.hh:153-164)
int ref_session_add_indel(void* p, const RefIndel* ind)
{
    Session* s = static_cast<Session*>(p);
    try {
        IndelObservation obs;
        obs.key = to_key(*ind);
        obs.data.is_external_candidate = (ind->is_candidate != 0);
        obs.data.iat = INDEL_ALIGN_TYPE::GENOME_TIER1_READ;
        obs.data.id = 1000000 + s->next_read_id++;
        s->buffer->addIndelObservation(0, obs);
        const IndelData* idp(s->buffer->getIndelDataPtr(obs.key));
        if (!idp) return 1;
        idp->status.is_candidate_indel = (ind->is_candidate != 0);
        idp->status.is_candidate_indel_cached = true;
        return 0;
    } catch (...) {
        return 1;
    }
}

/// the error rates the reference attached to an indel (IndelData::initializeAuxInfo): refToIndel, indelToRef
int ref_session_indel_error_rates(void* p, const RefIndel* ind, double* ref_to_indel, double* indel_to_ref)
{
    Session* s = static_cast<Session*>(p);
    const IndelData* idp(s->buffer->getIndelDataPtr(to_key(*ind)));
    if (!idp) return 1;
    const auto& er(idp->getSampleData(0).getErrorRates());
    *ref_to_indel = er.refToIndelErrorProb.getValue();
    *indel_to_ref = er.indelToRefErrorProb.getValue();
    return 0;
}

struct RefPathSeg
{
    uint32_t type, length;
};
struct RefCal
{
    int32_t pos;
    int32_t n_seg;
    const RefPathSeg* path;
    int32_t n_indels;
    const RefIndel* indels;
    RefIndel leading, trailing;
};

/// scoreCandidateAlignment (starling_read_align_score.cpp:261-499) for one candidate alignment of one read.
/// read_seq: ACGTN= characters.  Every indel of the alignment must have been added to the session.
int ref_session_score_cal(void* p, const char* read_seq, const uint8_t* qual, int read_len, const RefCal* c, double* out)
{
    Session* s = static_cast<Session*>(p);
    try {
        bam_record bamRead;
        bamRead.set_qname("R");
        bamRead.set_readqual(read_seq, qual);
        alignment al;
        al.pos = c->pos;
        for (int i = 0; i < c->n_seg; ++i)
            al.path.push_back(ALIGNPATH::path_segment(static_cast<ALIGNPATH::align_t>(c->path[i].type), c->path[i].length));
        // the bam record only supplies sequence/qualities here; give it a trivially valid alignment
        alignment bal;
        bal.pos = c->pos;
        bal.path.push_back(ALIGNPATH::path_segment(ALIGNPATH::MATCH, unsigned(read_len)));
        bam1_t& br(*(bamRead.get_data()));
        br.core.pos = bal.pos;
        edit_bam_cigar(bal.path, br);
        starling_read sread(bamRead, bal, MAPLEVEL::TIER1_MAPPED, 0);
        read_segment& rseg(sread.get_full_segment());

        CandidateAlignment cal;
        cal.al = al;
        indel_set_t iset;
        for (int i = 0; i < c->n_indels; ++i) iset.insert(to_key(c->indels[i]));
        cal.setIndels(iset);
        if (c->leading.type != INDEL::NONE) cal.leading_indel_key = to_key(c->leading);
        if (c->trailing.type != INDEL::NONE) cal.trailing_indel_key = to_key(c->trailing);
        *out = scoreCandidateAlignment(s->opt, *s->buffer, rseg, cal, s->ref);
        return 0;
    } catch (...) {
        return 1;
    }
}


/// as ref_session_add_indel, but the observation is attributed to read `read_id` (tier1_map_read_ids of sample 0), so
/// that is_usable_indel (starling_read_align.cpp:289-305) accepts it for that read even when it is not a candidate
int ref_session_add_indel_observed(void* p, const RefIndel* ind, unsigned read_id)
{
    Session* s = static_cast<Session*>(p);
    try {
        IndelObservation obs;
        obs.key = to_key(*ind);
        obs.data.is_external_candidate = false;
        obs.data.iat = INDEL_ALIGN_TYPE::GENOME_TIER1_READ;
        obs.data.id = read_id;
        s->buffer->addIndelObservation(0, obs);
        const IndelData* idp(s->buffer->getIndelDataPtr(obs.key));
        if (!idp) return 1;
        idp->status.is_candidate_indel = (ind->is_candidate != 0);
        idp->status.is_candidate_indel_cached = true;
        return 0;
    } catch (...) {
        return 1;
    }
}

/// haplotyping annotations of an indel already in the session (sample 0)
int ref_session_set_indel_haplotype(void* p, const RefIndel* ind, int active_region_id, int haplotype_id,
                                    int is_haplotyping_bypassed, int is_forced_output, int not_discovered_from_reads)
{
    Session* s = static_cast<Session*>(p);
    IndelData* idp(s->buffer->getIndelDataPtr(to_key(*ind)));
    if (!idp) return 1;
    idp->activeRegionId = active_region_id;
    idp->getSampleData(0).haplotypeId = uint8_t(haplotype_id);
    idp->getSampleData(0).isHaplotypingBypassed = (is_haplotyping_bypassed != 0);
    idp->isForcedOutput = (is_forced_output != 0);
    idp->status.notDiscoveredFromReads = (not_discovered_from_reads != 0);
    return 0;
}

/// log error rates as score_indels reads them (getLogValue)
int ref_session_indel_log_error_rates(void* p, const RefIndel* ind, double* ref_to_indel, double* indel_to_ref)
{
    Session* s = static_cast<Session*>(p);
    const IndelData* idp(s->buffer->getIndelDataPtr(to_key(*ind)));
    if (!idp) return 1;
    const auto& er(idp->getSampleData(0).getErrorRates());
    *ref_to_indel = er.refToIndelErrorProb.getLogValue();
    *indel_to_ref = er.indelToRefErrorProb.getLogValue();
    return 0;
}

/// realignAndScoreRead (starling_read_align.cpp:2026-2126) on one read of sample 0.
/// Returns 0 ok, 1 when the reference throws.  Scores are left in the session's IndelBuffer (ref_session_read_scores).
int ref_session_realign(void* p, const char* read_seq, const uint8_t* qual, int read_len, int pos, int n_seg,
                        const RefPathSeg* path, int is_fwd, int map_level, int realign_begin, int realign_end,
                        unsigned read_id, int is_haplotyping_enabled, int min_read_bp_flank, int* is_realigned,
                        int* out_pos, char* out_cigar, int cigar_cap)
{
    Session* s = static_cast<Session*>(p);
    try {
        bam_record bamRead;
        bamRead.set_qname("R");
        bamRead.set_readqual(read_seq, qual);
        alignment al;
        al.pos = pos;
        al.is_fwd_strand = (is_fwd != 0);
        for (int i = 0; i < n_seg; ++i)
            al.path.push_back(ALIGNPATH::path_segment(static_cast<ALIGNPATH::align_t>(path[i].type), path[i].length));
        bam1_t& br(*(bamRead.get_data()));
        br.core.pos = al.pos;
        if (!is_fwd) br.core.flag |= BAM_FLAG::STRAND;
        edit_bam_cigar(al.path, br);
        (void)read_len;
        starling_read sread(bamRead, al, static_cast<MAPLEVEL::index_t>(map_level), read_id);
        read_segment& rseg(sread.get_full_segment());

        s->opt.isHaplotypingEnabled = (is_haplotyping_enabled != 0);
        s->sopt->min_read_bp_flank = min_read_bp_flank;
        const known_pos_range realign_range(realign_begin, realign_end);
        realignAndScoreRead(s->opt, *s->dopt, *s->sopt, s->ref, realign_range, 0, rseg, *s->buffer);
        *is_realigned = rseg.is_realigned ? 1 : 0;
        *out_pos = rseg.is_realigned ? rseg.realignment.pos : 0;
        const std::string cigar(rseg.is_realigned ? ALIGNPATH::apath_to_cigar(rseg.realignment.path) : std::string());
        std::strncpy(out_cigar, cigar.c_str(), cigar_cap - 1);
        out_cigar[cigar_cap - 1] = 0;
        return 0;
    } catch (...) {
        return 1;
    }
}

struct RefReadScore
{
    int32_t pos, type;
    uint32_t del_len, ins_len;
    char ins[64];
    float ref_lnp, indel_lnp;
    uint16_t non_ambig, read_length;
    int32_t is_tier1_read, is_fwd_strand;
    int32_t read_pos, edge_dist;
    int32_t n_alt;
    int32_t alt_pos[2], alt_type[2];
    uint32_t alt_del_len[2];
    char alt_ins[2][64];
    float alt_lnp[2];
    int32_t is_suboverlap; // 1: the read is only in suboverlap_tier{1,2}_read_ids of this indel
};

/// everything score_indels stored for read `read_id`: read_path_lnp entries and suboverlap memberships, in IndelKey order
int ref_session_read_scores(void* p, unsigned read_id, RefReadScore* out, int cap)
{
    Session* s = static_cast<Session*>(p);
    int n = 0;
    const auto range(s->buffer->rangeIterator(-1000000, 1000000000));
    for (auto it(range.first); it != range.second; ++it) {
        const IndelKey& k(it->first);
        const IndelSampleData& sd(it->second.getSampleData(0));
        const auto f(sd.read_path_lnp.find(read_id));
        const bool sub = (sd.suboverlap_tier1_read_ids.count(read_id) > 0) || (sd.suboverlap_tier2_read_ids.count(read_id) > 0);
        if (f == sd.read_path_lnp.end() && !sub) continue;
        if (n >= cap) return -1;
        RefReadScore& o(out[n++]);
        std::memset(&o, 0, sizeof(o));
        o.pos = k.pos;
        o.type = k.type;
        o.del_len = k.deletionLength;
        o.ins_len = k.insertSequence.size();
        std::strncpy(o.ins, k.insertSequence.c_str(), 63);
        if (f == sd.read_path_lnp.end()) {
            o.is_suboverlap = 1;
            continue;
        }
        const ReadPathScores& r(f->second);
        o.ref_lnp = r.ref;
        o.indel_lnp = r.indel;
        o.non_ambig = r.nonAmbiguousBasesInRead;
        o.read_length = r.read_length;
        o.is_tier1_read = r.is_tier1_read;
        o.is_fwd_strand = r.is_fwd_strand;
        o.read_pos = r.read_pos;
        o.edge_dist = r.distanceFromClosestReadEdge;
        o.n_alt = int32_t(r.alt_indel.size());
        for (int a = 0; a < o.n_alt && a < 2; ++a) {
            o.alt_pos[a] = r.alt_indel[a].first.pos;
            o.alt_type[a] = r.alt_indel[a].first.type;
            o.alt_del_len[a] = r.alt_indel[a].first.deletionLength;
            std::strncpy(o.alt_ins[a], r.alt_indel[a].first.insertSequence.c_str(), 63);
            o.alt_lnp[a] = r.alt_indel[a].second;
        }
    }
    return n;
}


// ---- timing entry points for bench.py's "reference" CPU baseline: inputs are materialised in the reference's own structs
// first, only the reference's compute calls are inside the clock ----

/// a read and its candidate alignments, ready for scoreCandidateAlignment
int ref_session_prepare_score_read(void* p, const char* read_seq, const uint8_t* qual, int read_len, const RefCal* cals, int n_cals)
{
    Session* s = static_cast<Session*>(p);
    try {
        PreparedScoreRead pr;
        bam_record bamRead;
        bamRead.set_qname("R");
        bamRead.set_readqual(read_seq, qual);
        alignment bal;
        bal.pos = n_cals ? cals[0].pos : 0;
        bal.path.push_back(ALIGNPATH::path_segment(ALIGNPATH::MATCH, unsigned(read_len)));
        bam1_t& br(*(bamRead.get_data()));
        br.core.pos = bal.pos;
        edit_bam_cigar(bal.path, br);
        pr.sread.reset(new starling_read(bamRead, bal, MAPLEVEL::TIER1_MAPPED, 0));
        for (int k = 0; k < n_cals; ++k) {
            const RefCal& c(cals[k]);
            CandidateAlignment cal;
            cal.al.pos = c.pos;
            for (int i = 0; i < c.n_seg; ++i)
                cal.al.path.push_back(ALIGNPATH::path_segment(static_cast<ALIGNPATH::align_t>(c.path[i].type), c.path[i].length));
            indel_set_t iset;
            for (int i = 0; i < c.n_indels; ++i) iset.insert(to_key(c.indels[i]));
            cal.setIndels(iset);
            if (c.leading.type != INDEL::NONE) cal.leading_indel_key = to_key(c.leading);
            if (c.trailing.type != INDEL::NONE) cal.trailing_indel_key = to_key(c.trailing);
            pr.cals.push_back(cal);
        }
        s->scoreReads.push_back(std::move(pr));
        return 0;
    } catch (...) {
        return 1;
    }
}

/// scoreCandidateAlignment over the prepared reads, pass after pass, for about `seconds`; returns the seconds spent and the
/// number of (read base x candidate alignment) cells scored
double ref_session_time_score(void* p, double seconds, double* cells, double* checksum)
{
    Session* s = static_cast<Session*>(p);
    double acc = 0, n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    double dt = 0;
    do {
        for (auto& pr : s->scoreReads) {
            read_segment& rseg(pr.sread->get_full_segment());
            for (const auto& cal : pr.cals) acc += scoreCandidateAlignment(s->opt, *s->buffer, rseg, cal, s->ref);
            n += double(rseg.read_size()) * double(pr.cals.size());
        }
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (dt < seconds);
    *cells = n;
    *checksum = acc;
    return dt;
}

/// a read ready for realignAndScoreRead (as ref_session_realign builds it)
int ref_session_prepare_realign_read(void* p, const char* read_seq, const uint8_t* qual, int pos, int n_seg, const RefPathSeg* path,
                                     int is_fwd, int map_level, int realign_begin, int realign_end, unsigned read_id)
{
    Session* s = static_cast<Session*>(p);
    try {
        bam_record bamRead;
        bamRead.set_qname("R");
        bamRead.set_readqual(read_seq, qual);
        alignment al;
        al.pos = pos;
        al.is_fwd_strand = (is_fwd != 0);
        for (int i = 0; i < n_seg; ++i)
            al.path.push_back(ALIGNPATH::path_segment(static_cast<ALIGNPATH::align_t>(path[i].type), path[i].length));
        bam1_t& br(*(bamRead.get_data()));
        br.core.pos = al.pos;
        if (!is_fwd) br.core.flag |= BAM_FLAG::STRAND;
        edit_bam_cigar(al.path, br);
        PreparedRealignRead pr;
        pr.sread.reset(new starling_read(bamRead, al, static_cast<MAPLEVEL::index_t>(map_level), read_id));
        pr.range = known_pos_range(realign_begin, realign_end);
        s->realignReads.push_back(std::move(pr));
        return 0;
    } catch (...) {
        return 1;
    }
}

/// realignAndScoreRead over the prepared reads, pass after pass, for about `seconds`; returns seconds, *reads = reads done
double ref_session_time_realign(void* p, double seconds, int is_haplotyping_enabled, int min_read_bp_flank, double* reads)
{
    Session* s = static_cast<Session*>(p);
    s->opt.isHaplotypingEnabled = (is_haplotyping_enabled != 0);
    s->sopt->min_read_bp_flank = min_read_bp_flank;
    double n = 0, dt = 0;
    const auto t0 = std::chrono::steady_clock::now();
    do {
        for (auto& pr : s->realignReads) {
            read_segment& rseg(pr.sread->get_full_segment());
            try {
                realignAndScoreRead(s->opt, *s->dopt, *s->sopt, s->ref, pr.range, 0, rseg, *s->buffer);
            } catch (...) {
            }
            n += 1;
        }
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (dt < seconds && !s->realignReads.empty());
    *reads = n;
    return dt;
}

} // extern "C"
