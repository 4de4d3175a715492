// ref_driver_pileup.cpp -- the REFERENCE's own position processor, end to end, for pinning rows a1-a8:
// reads -> read buffer -> realignAndScoreRead (stage READ_BUFFER) -> pileup_read_segment -> per-position pileup columns.
//
// TEST INFRASTRUCTURE ONLY; contains no reference code.  A minimal subclass of starling_pos_processor_base
// (L/starling_common/starling_pos_processor_base.hh:86) supplies the one pure virtual (process_pos_variants_impl, called
// at stage POST_ALIGN for every reportable position) and uses it to snapshot what the germline/somatic callers would
// see: the position's snp_pos_info (calls, tier2 calls, spanning-deletion and submapped counts) and the final alignment
// of every read buffered at that position.

#include "appstats/RunStats.hh"
#include "appstats/RunStatsManager.hh"
#include "htsapi/align_path_bam_util.hh"
#include "starling_common/normalizeAlignment.hh"
#include "starling_common/starling_pos_processor_base.hh"
#include "starling_common/starling_pos_processor_base_stages.hh"
#include "starling_common/starling_streams_base.hh"
#include "options/AlignmentFileOptions.hh"
#include "starling_common/starling_base_shared.hh"

#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>


namespace
{

struct Column
{
    int32_t pos;
    std::vector<uint16_t> calls, tier2_calls;
    uint32_t spandel, submapped;
    // with the germline EVS accumulators switched on (updateGermlineScoringMetrics): what the reference derives from them
    // (snp_pos_info::get_mq_ranksum / get_baseq_ranksum / get_read_pos_ranksum / get_raw_pos / get_raw_baseQ, distanceFromReadEdge)
    double evs[6];
    uint32_t mapq_count;
};

struct FinalAlignment
{
    uint32_t read_id;
    int32_t is_realigned, pos, is_fwd, skipped;
    std::string cigar;
    int32_t input_pos, realign_begin, realign_end; // what realignAndScoreRead was given for this read
    std::string input_cigar;
};

struct DumpIndel
{
    int32_t pos, type;
    uint32_t del_len;
    std::string ins;
    int32_t is_candidate, not_discovered_from_reads, is_forced_output;
    double ref_to_indel_lnp, indel_to_ref_lnp;
    std::vector<uint32_t> read_ids; // tier1/tier2/submapped/noise observations of sample 0 (is_usable_indel)
    // read_path_lnp of sample 0 (score_indels output): read id, {ref, indel} lnp, then the alternate indels
    struct Score
    {
        uint32_t read_id;
        float ref_lnp, indel_lnp;
        int32_t non_ambig, read_length, is_tier1, is_fwd, read_pos, edge_dist, n_alt;
        int32_t alt_pos[2];
        uint32_t alt_del[2];
        std::string alt_ins[2];
        float alt_lnp[2];
    };
    std::vector<Score> scores;
};

struct Streams : public starling_streams_base
{
    explicit Streams(const unsigned n) : starling_streams_base(n) {}
};

struct PP : public starling_pos_processor_base
{
    PP(const starling_base_options& opt, const starling_base_deriv_options& dopt, const reference_contig_segment& ref,
       const Streams& streams, RunStatsManager& stats)
        : starling_pos_processor_base(opt, dopt, ref, streams, 1, stats)
    {
        // as the application processors do (L/applications/starling/starling_pos_processor.cpp:62-68)
        sample_info& sif(sample(0));
        getIndelBuffer().registerSample(sif.estdepth_buff, sif.estdepth_buff_tier2, true);
        getIndelBuffer().finalizeSamples();
    }

    void resetRegion(const std::string& chrom, const known_pos_range2& range) { resetRegionBase(chrom, range); }

    void process_pos_variants_impl(const pos_t pos, const bool /*isPosPrecedingReportableRange*/) override
    {
        static_assert(sizeof(base_call) == 2, "base_call is a 16-bit bitfield");
        const snp_pos_info& pi(sample(0).basecallBuffer.get_pos(pos));
        Column c;
        c.pos = pos;
        for (const base_call& bc : pi.calls) {
            uint16_t v;
            std::memcpy(&v, &bc, 2);
            c.calls.push_back(v);
        }
        for (const base_call& bc : pi.tier2_calls) {
            uint16_t v;
            std::memcpy(&v, &bc, 2);
            c.tier2_calls.push_back(v);
        }
        c.spandel = pi.spanningDeletionReadCount;
        c.submapped = pi.submappedReadCount;
        c.evs[0] = pi.get_mq_ranksum();
        c.evs[1] = pi.get_baseq_ranksum();
        c.evs[2] = pi.get_read_pos_ranksum();
        c.evs[3] = pi.get_raw_pos();
        c.evs[4] = pi.get_raw_baseQ();
        c.evs[5] = pi.distanceFromReadEdge.mean();
        c.mapq_count = pi.mapqTracker.count;
        if (!(c.calls.empty() && c.tier2_calls.empty() && c.spandel == 0 && c.submapped == 0)) columns.push_back(c);

    }

    // called at stage POST_ALIGN for EVERY position (reportable or not), after the reads buffered at `pos` were realigned
    // and piled up (stage READ_BUFFER) and before the read buffer is cleared: record their final alignments, in the
    // order pileup_pos_reads visited them
    void post_align_clear_pos(const pos_t pos) override
    {
        // the realignment range align_pos(pos) used (get_realignment_range, starling_pos_processor_base.cpp:705-727);
        // valid as long as the stage sizes did not change between the two stages (the test reads keep them fixed)
        int32_t rb, re;
        {
            const stage_data& sd(_stagemanPtr->get_stage_data());
            const unsigned head_offset(sd.get_stage_id_shift(STAGE::HEAD));
            const unsigned buffer_offset(sd.get_stage_id_shift(STAGE::READ_BUFFER));
            const unsigned post_offset(sd.get_stage_id_shift(STAGE::POST_ALIGN));
            pos_t min_offset(static_cast<pos_t>(post_offset - buffer_offset));
            pos_t max_offset(static_cast<pos_t>(buffer_offset - head_offset));
            min_offset = std::max(0, min_offset - 1);
            max_offset = std::max(0, max_offset - 1);
            rb = std::max(static_cast<pos_t>(0), pos - min_offset);
            re = pos + 1 + max_offset;
        }
        // the indels keyed at this position, as the realigner saw them
        {
            const auto range(getIndelBuffer().rangeIterator(pos, pos + 1));
            for (auto it(range.first); it != range.second; ++it) {
                const IndelKey& k(it->first);
                if (k.pos != pos) continue;
                const IndelData& d(it->second);
                DumpIndel di;
                di.pos = k.pos;
                di.type = k.type;
                di.del_len = k.deletionLength;
                di.ins = k.insertSequence;
                di.is_candidate = getIndelBuffer().isCandidateIndel(k, d) ? 1 : 0;
                di.not_discovered_from_reads = d.status.notDiscoveredFromReads ? 1 : 0;
                di.is_forced_output = d.isForcedOutput ? 1 : 0;
                const IndelSampleData& sdat(d.getSampleData(0));
                di.ref_to_indel_lnp = sdat.getErrorRates().refToIndelErrorProb.getLogValue();
                di.indel_to_ref_lnp = sdat.getErrorRates().indelToRefErrorProb.getLogValue();
                for (const auto id : sdat.tier1_map_read_ids) di.read_ids.push_back(id);
                for (const auto id : sdat.tier2_map_read_ids) di.read_ids.push_back(id);
                for (const auto id : sdat.submap_read_ids) di.read_ids.push_back(id);
                for (const auto id : sdat.noise_read_ids) di.read_ids.push_back(id);
                for (const auto& kv : sdat.read_path_lnp) {
                    const ReadPathScores& r(kv.second);
                    DumpIndel::Score sc;
                    sc.read_id = kv.first;
                    sc.ref_lnp = r.ref;
                    sc.indel_lnp = r.indel;
                    sc.non_ambig = r.nonAmbiguousBasesInRead;
                    sc.read_length = r.read_length;
                    sc.is_tier1 = r.is_tier1_read;
                    sc.is_fwd = r.is_fwd_strand;
                    sc.read_pos = r.read_pos;
                    sc.edge_dist = r.distanceFromClosestReadEdge;
                    sc.n_alt = int32_t(r.alt_indel.size());
                    for (int a = 0; a < 2 && a < sc.n_alt; ++a) {
                        sc.alt_pos[a] = r.alt_indel[a].first.pos;
                        sc.alt_del[a] = r.alt_indel[a].first.deletionLength;
                        sc.alt_ins[a] = r.alt_indel[a].first.insertSequence;
                        sc.alt_lnp[a] = r.alt_indel[a].second;
                    }
                    di.scores.push_back(sc);
                }
                indels.push_back(di);
            }
        }
        read_segment_iter ri(sample(0).readBuffer.get_pos_read_segment_iter(pos));
        for (read_segment_iter::ret_val r; true; ri.next()) {
            r = ri.get_ptr();
            if (nullptr == r.first) break;
            const read_segment& rseg(r.first->get_segment(r.second));
            FinalAlignment fa;
            fa.input_pos = rseg.getInputAlignment().pos;
            fa.input_cigar = ALIGNPATH::apath_to_cigar(rseg.getInputAlignment().path);
            fa.realign_begin = rb;
            fa.realign_end = re;
            fa.read_id = rseg.getReadIndex();
            fa.is_realigned = rseg.is_realigned ? 1 : 0;
            const alignment& al(rseg.is_realigned ? rseg.realignment : rseg.getInputAlignment());
            fa.pos = al.pos;
            fa.is_fwd = al.is_fwd_strand;
            fa.cigar = ALIGNPATH::apath_to_cigar(al.path);
            fa.skipped = (!rseg.is_realigned && !rseg.is_any_nonovermax(_opt.maxIndelSize)) ||
                         (rseg.is_realigned && rseg.is_invalid_realignment);
            finals.push_back(fa);
        }
    }

    std::vector<Column> columns;
    std::vector<FinalAlignment> finals;
    std::vector<DumpIndel> indels;
};

/// the test options with the germline EVS accumulators on demand (starling_options::is_compute_germline_scoring_metrics is true when
/// scoring models are loaded or --report-evs-features is given, L/applications/starling/starling_shared.hh:70)
struct DriverOptions : public starling_base_options
{
    bool isGermlineMetrics = false;
    bool is_compute_germline_scoring_metrics() const override { return isGermlineMetrics; }
    const AlignmentFileOptions& getAlignmentFileOptions() const override // (as L/test/starling_base_options_test.hh, which is final)
    {
        static AlignmentFileOptions alignFileOpt;
        if (alignFileOpt.alignmentFilenames.empty()) alignFileOpt.alignmentFilenames.push_back("sample.bam");
        return alignFileOpt;
    }
};

bool g_germline_metrics = false;

struct Session
{
    DriverOptions opt;
    std::unique_ptr<starling_base_deriv_options> dopt;
    reference_contig_segment ref;
    std::unique_ptr<Streams> streams;
    std::unique_ptr<RunStatsManager> stats;
    std::unique_ptr<PP> pp;
};

struct RefPathSeg
{
    uint32_t type, length;
};

} // namespace

extern "C" {

/// one position processor over [report_begin, report_end) of a reference segment
void* refpp_create(const char* ref_seq, int ref_offset, int report_begin, int report_end, int min_basecall_qscore,
                   int mdf_flank, int mdf_max_count, int use_tier2, int tier2_mdf_max_count, int is_mapq_adjust,
                   int min_dist_from_read_edge)
{
    try {
        Session* s = new Session();
        s->opt.isGermlineMetrics = g_germline_metrics;
        s->opt.isHaplotypingEnabled = false;
        s->opt.minBasecallErrorPhredProb = min_basecall_qscore;
        s->opt.mismatchDensityFilterFlankSize = mdf_flank;
        s->opt.mismatchDensityFilterMaxMismatchCount = mdf_max_count;
        s->opt.useTier2Evidence = (use_tier2 != 0);
        s->opt.tier2.mismatchDensityFilterMaxMismatchCount = tier2_mdf_max_count;
        s->opt.isBasecallQualAdjustedForMapq = (is_mapq_adjust != 0);
        s->opt.minDistanceFromReadEdge = min_dist_from_read_edge;
        s->dopt.reset(new starling_base_deriv_options(s->opt));
        s->ref.seq() = ref_seq;
        s->ref.set_offset(ref_offset);
        s->streams.reset(new Streams(1));
        s->stats.reset(new RunStatsManager(""));
        s->pp.reset(new PP(s->opt, *s->dopt, s->ref, *s->streams, *s->stats));
        s->pp->resetRegion("chrT", known_pos_range2(report_begin, report_end));
        return s;
    } catch (...) {
        return nullptr;
    }
}

void refpp_destroy(void* p) { delete static_cast<Session*>(p); }

/// processInputReadAlignment's tail (L/starling_common/starling_pos_processor_util.cpp:395-440): left-normalise, then
/// insert_read.  Reads must arrive in position order (set_head_pos is advanced as starling_run.cpp:127 does).
/// Returns the read id (>= 0), -1 if the processor declined the read, -2 on exception.
int refpp_add_read(void* p, const char* read_seq, const uint8_t* qual, int pos, int n_seg, const RefPathSeg* path,
                   int is_fwd, int mapq, int map_level)
{
    Session* s = static_cast<Session*>(p);
    try {
        s->pp->set_head_pos(pos - 1);
        bam_record br;
        br.set_qname("R");
        br.set_readqual(read_seq, qual);
        alignment al;
        al.pos = pos;
        al.is_fwd_strand = (is_fwd != 0);
        for (int i = 0; i < n_seg; ++i)
            al.path.push_back(ALIGNPATH::path_segment(static_cast<ALIGNPATH::align_t>(path[i].type), path[i].length));
        bam1_t& b(*(br.get_data()));
        b.core.pos = al.pos;
        b.core.qual = uint8_t(mapq);
        if (!is_fwd) b.core.flag |= BAM_FLAG::STRAND;
        edit_bam_cigar(al.path, b);
        {
            const rc_segment_bam_seq refBamSeq(s->ref);
            const bam_seq readBamSeq(br.get_bam_read());
            normalizeAlignment(refBamSeq, readBamSeq, al);
        }
        const boost::optional<align_id_t> id(s->pp->insert_read(br, al, "chrT", static_cast<MAPLEVEL::index_t>(map_level), 0));
        return id ? int(*id) : -1;
    } catch (...) {
        return -2;
    }
}

/// an externally supplied candidate indel (as a --candidate-indel-input-vcf record would be)
int refpp_add_candidate_indel(void* p, int pos, int del_len, const char* ins_seq)
{
    Session* s = static_cast<Session*>(p);
    try {
        s->pp->set_head_pos(pos - 1);
        IndelObservation obs;
        obs.key = IndelKey(pos, INDEL::INDEL, del_len, ins_seq ? ins_seq : "");
        obs.data.is_external_candidate = true;
        s->pp->insert_indel(obs, 0);
        return 0;
    } catch (...) {
        return 1;
    }
}

/// flush every stage (starling_pos_processor_base::reset)
int refpp_finish(void* p)
{
    Session* s = static_cast<Session*>(p);
    try {
        s->pp->reset();
        return 0;
    } catch (...) {
        return 1;
    }
}

int refpp_n_columns(void* p) { return int(static_cast<Session*>(p)->pp->columns.size()); }

int refpp_column_info(void* p, int i, int32_t* pos, int32_t* n_calls, int32_t* n_tier2, uint32_t* spandel, uint32_t* submapped)
{
    const Column& c(static_cast<Session*>(p)->pp->columns[i]);
    *pos = c.pos;
    *n_calls = int32_t(c.calls.size());
    *n_tier2 = int32_t(c.tier2_calls.size());
    *spandel = c.spandel;
    *submapped = c.submapped;
    return 0;
}

int refpp_column_calls(void* p, int i, uint16_t* calls, uint16_t* tier2_calls)
{
    const Column& c(static_cast<Session*>(p)->pp->columns[i]);
    if (!c.calls.empty()) std::memcpy(calls, c.calls.data(), 2 * c.calls.size());
    if (!c.tier2_calls.empty()) std::memcpy(tier2_calls, c.tier2_calls.data(), 2 * c.tier2_calls.size());
    return 0;
}

/// sessions created from now on accumulate the germline EVS metrics in their pileups (0 / 1)
void refpp_set_germline_metrics(int on) { g_germline_metrics = (on != 0); }

/// column i: {MQRankSum, BaseQRankSum, ReadPosRankSum, rawPos, avgBaseQ, meanDistanceFromReadEdge} as the reference derives them, and
/// the MapqTracker's count
int refpp_column_evs(void* p, int i, double* evs6, uint32_t* mapq_count)
{
    const Column& c(static_cast<Session*>(p)->pp->columns[i]);
    for (int k = 0; k < 6; ++k) evs6[k] = c.evs[k];
    *mapq_count = c.mapq_count;
    return 0;
}

/// the reference's own accumulators fed from a list of observations: the same six numbers from what updateGermlineScoringMetrics got
/// per basecall (is_reference, mapq, qscore, cycle, distance from read edge (already capped), is_submapped)
int ref_germline_metrics_from_observations(int n, const uint8_t* is_reference, const uint8_t* mapq, const uint8_t* qscore,
                                           const uint16_t* cycle, const uint8_t* edge, const uint8_t* is_submapped, double* evs6)
{
    snp_pos_info pi;
    for (int i = 0; i < n; ++i) { // pos_basecall_buffer::updateGermlineScoringMetrics, pos_basecall_buffer.cpp:43-70
        const bool isRef(is_reference[i] != 0);
        pi.mq_ranksum.add_observation(isRef, static_cast<unsigned>(mapq[i]));
        if (!is_submapped[i]) {
            pi.baseq_ranksum.add_observation(isRef, static_cast<unsigned>(qscore[i]));
            pi.readPositionRankSum.add_observation(isRef, cycle[i]);
            if (!isRef) pi.distanceFromReadEdge.addObservation(edge[i]);
        }
    }
    evs6[0] = pi.get_mq_ranksum();
    evs6[1] = pi.get_baseq_ranksum();
    evs6[2] = pi.get_read_pos_ranksum();
    evs6[3] = pi.get_raw_pos();
    evs6[4] = pi.get_raw_baseQ();
    evs6[5] = pi.distanceFromReadEdge.mean();
    return 0;
}

int refpp_n_finals(void* p) { return int(static_cast<Session*>(p)->pp->finals.size()); }

int refpp_final(void* p, int i, uint32_t* read_id, int32_t* is_realigned, int32_t* pos, int32_t* is_fwd, int32_t* skipped,
                char* cigar, int cap)
{
    const FinalAlignment& f(static_cast<Session*>(p)->pp->finals[i]);
    *read_id = f.read_id;
    *is_realigned = f.is_realigned;
    *pos = f.pos;
    *is_fwd = f.is_fwd;
    *skipped = f.skipped;
    std::strncpy(cigar, f.cigar.c_str(), cap - 1);
    cigar[cap - 1] = 0;
    return 0;
}

/// what realignAndScoreRead was given for final i: the (normalised) input alignment and the realignment range
int refpp_final_input(void* p, int i, int32_t* input_pos, int32_t* realign_begin, int32_t* realign_end, char* cigar, int cap)
{
    const FinalAlignment& f(static_cast<Session*>(p)->pp->finals[i]);
    *input_pos = f.input_pos;
    *realign_begin = f.realign_begin;
    *realign_end = f.realign_end;
    std::strncpy(cigar, f.input_cigar.c_str(), cap - 1);
    cigar[cap - 1] = 0;
    return 0;
}

int refpp_n_indels(void* p) { return int(static_cast<Session*>(p)->pp->indels.size()); }

/// indel i of the buffer as the realigner saw it; returns the number of observing read ids (copied up to id_cap)
int refpp_indel(void* p, int i, int32_t* pos, int32_t* type, uint32_t* del_len, char* ins, int ins_cap, int32_t* is_candidate,
                int32_t* not_discovered_from_reads, int32_t* is_forced_output, double* ref_to_indel_lnp, double* indel_to_ref_lnp, uint32_t* read_ids, int id_cap)
{
    const DumpIndel& d(static_cast<Session*>(p)->pp->indels[i]);
    *pos = d.pos;
    *type = d.type;
    *del_len = d.del_len;
    std::strncpy(ins, d.ins.c_str(), ins_cap - 1);
    ins[ins_cap - 1] = 0;
    *is_candidate = d.is_candidate;
    *not_discovered_from_reads = d.not_discovered_from_reads;
    *is_forced_output = d.is_forced_output;
    *ref_to_indel_lnp = d.ref_to_indel_lnp;
    *indel_to_ref_lnp = d.indel_to_ref_lnp;
    const int n(int(d.read_ids.size()));
    for (int k = 0; k < n && k < id_cap; ++k) read_ids[k] = d.read_ids[k];
    return n;
}

int refpp_indel_n_scores(void* p, int i) { return int(static_cast<Session*>(p)->pp->indels[i].scores.size()); }

/// score k of indel i: ints = {read_id, non_ambig, read_length, is_tier1, is_fwd, read_pos, edge_dist, n_alt, alt_pos[2],
/// alt_del[2]}; floats = {ref, indel, alt_lnp[2]}; alt insert sequences concatenated with '|'
int refpp_indel_score(void* p, int i, int k, int32_t* ints, float* floats, char* alt_ins, int cap)
{
    const DumpIndel::Score& s(static_cast<Session*>(p)->pp->indels[i].scores[k]);
    ints[0] = int32_t(s.read_id);
    ints[1] = s.non_ambig;
    ints[2] = s.read_length;
    ints[3] = s.is_tier1;
    ints[4] = s.is_fwd;
    ints[5] = s.read_pos;
    ints[6] = s.edge_dist;
    ints[7] = s.n_alt;
    for (int a = 0; a < 2; ++a) {
        ints[8 + a] = (a < s.n_alt) ? s.alt_pos[a] : 0;
        ints[10 + a] = (a < s.n_alt) ? int32_t(s.alt_del[a]) : 0;
        floats[2 + a] = (a < s.n_alt) ? s.alt_lnp[a] : 0.f;
    }
    floats[0] = s.ref_lnp;
    floats[1] = s.indel_lnp;
    const std::string joined((s.n_alt > 0 ? s.alt_ins[0] : std::string()) + "|" + (s.n_alt > 1 ? s.alt_ins[1] : std::string()));
    std::strncpy(alt_ins, joined.c_str(), cap - 1);
    alt_ins[cap - 1] = 0;
    return 0;
}

} // extern "C"
