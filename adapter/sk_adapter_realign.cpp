// sk_adapter_realign.cpp -- site 1: realignAndScoreRead (L/starling_common/starling_pos_processor_base.cpp:732-773,
// align_pos) as one sk_realign_job per sample and stage window.
//
// The reference realigns the reads buffered at position P when its READ_BUFFER stage reaches P.  Here that stage runs
// read_buffer_defer() positions later (sk_adapter.hh), so when it reaches P every read buffered at [P, P+W) is in the
// read buffer and every indel such a read can reach is as final as it was for the reference: the window's reads go
// through one job (gate, normalisation, enumeration and flattening on the host, scoring of all their candidate alignments
// in one kernel launch, selection and score_indels on the host) and the results are written where the reference writes
// them: rseg.realignment / is_realigned (starling_read_align.cpp:1738-1739), IndelSampleData::read_path_lnp
// (starling_read_align_score_indels.cpp:1071) and the suboverlap read-id sets (:616-626).
#include "sk_adapter_access.hh"

#include <cstdlib>

#include "blt_util/log.hh"
#include "starling_common/alignment_util.hh"
#include "starling_common/starling_read_segment.hh"

#include <algorithm>
#include <iostream>
#include <unordered_map>

namespace sk_adapter
{

namespace
{

struct WindowRead
{
    read_segment* rseg;
    pos_t bufferPos;
    GeometryShadow::Params geometry;
    pos_t rangeBegin, rangeEnd;
    pos_t zoneBegin, zoneEnd;
    size_t codeOffset, pathOffset, observedOffset;
    unsigned observedCount;
};

void toIndelKey(const IndelKey& k, const bool isCandidate, sk_indel_key& out)
{
    out.pos = k.pos;
    out.type = static_cast<int32_t>(k.type);
    out.del_len = k.deletionLength;
    out.ins_len = static_cast<uint32_t>(k.insertSequence.size());
    out.ins_seq = k.insertSequence.c_str();
    out.is_candidate = isCandidate ? 1 : 0;
}

void realign_sample_window(starling_pos_processor_base& pp, const unsigned sampleIndex, const pos_t begin, const pos_t end)
{
    State& s(state());
    AccumTimer hookTimer(s.tRealignHook);
    const starling_base_options& opt(Access::opt(pp));
    const unsigned sampleCount(Access::sampleCount(pp));
    starling_pos_processor_base::sample_info& sif(pp.sample(sampleIndex));
    IndelBuffer& indelBuffer(Access::indelBuffer(pp));
    const reference_contig_segment& ref(Access::ref(pp));

    if (sampleCount > SK_MAX_SAMPLES) throw blt_exception("strelka_amd adapter: more samples than SK_MAX_SAMPLES");

    // ---- the window's reads, in read-buffer order ----
    std::vector<WindowRead> reads;
    std::vector<uint8_t> codes;
    std::vector<sk_path_seg> paths;
    pos_t tableBegin(0), tableEnd(0);
    // how far past its alignment zone the candidate-alignment search of a read can reach: every toggled indel moves the
    // far end by at most maxIndelSize (starling_read_align.cpp:859-1277, max_read_indel_toggle)
    const pos_t reach(static_cast<pos_t>(opt.max_read_indel_toggle * opt.maxIndelSize + 5));
    for (const WindowSegment& ws : s.windowSegments[sampleIndex])
    {
        read_segment& rseg(*ws.rseg);
        const pos_t pos(ws.bufferPos);
        if (not (opt.is_realign_submapped_reads || rseg.is_tier1or2_mapping())) continue;
        if (! rseg.is_valid())
        {
            log_os << "ERROR: invalid alignment path associated with read segment:\n" << rseg;
            exit(EXIT_FAILURE);
        }
        WindowRead wr;
        wr.rseg = &rseg;
        wr.bufferPos = pos;
        wr.geometry = s.geometry.query(pos);
        wr.rangeBegin = std::max(static_cast<pos_t>(0), pos - wr.geometry.rangeMinOffset);
        wr.rangeEnd = pos + 1 + wr.geometry.rangeMaxOffset;
        const alignment& al(rseg.getInputAlignment());
        const known_pos_range zone(get_alignment_zone(al, rseg.read_size()));
        wr.zoneBegin = zone.begin_pos;
        wr.zoneEnd = zone.end_pos;
        const pos_t lo(std::max(wr.rangeBegin, std::min(zone.begin_pos, pos) - reach));
        const pos_t hi(std::min(wr.rangeEnd, zone.end_pos + reach));
        if (reads.empty())
        {
            tableBegin = lo;
            tableEnd = hi;
        }
        else
        {
            tableBegin = std::min(tableBegin, lo);
            tableEnd = std::max(tableEnd, hi);
        }
        wr.codeOffset = 0;
        wr.pathOffset = 0;
        wr.observedOffset = 0;
        wr.observedCount = 0;
        reads.push_back(wr);
    }
    s.realignReads += reads.size();
    if (reads.empty()) return;

    // ---- the IndelBuffer entries those reads can see ----
    std::vector<sk_indel_info> table;
    std::vector<const IndelKey*> keys;
    std::vector<IndelData*> data;
    std::unordered_map<align_id_t, std::vector<int32_t>> observedBy;
    {
        const auto range(indelBuffer.rangeIterator(tableBegin, tableEnd));
        for (auto it(range.first); it != range.second; ++it)
        {
            const IndelKey& k(it->first);
            IndelData& d(getIndelData(it));
            if (k.is_breakpoint()) throw blt_exception("strelka_amd adapter: open-ended breakpoint alleles are not supported on this path");
            sk_indel_info e;
            std::memset(&e, 0, sizeof(e));
            // candidate status as of now, WITHOUT committing it to the IndelBuffer's cache: the reference computes and caches
            // an indel's status when a read first asks for it (IndelBuffer.hh:153-164), and most indels of this table are
            // never asked about by this window's reads -- those keep their status open, as in the reference, until the
            // window (or the caller stage) that does ask (the commit is after the job, below)
            const IndelData::status_t savedStatus(d.status);
            const bool isCandidate(indelBuffer.isCandidateIndel(k, d));
            // (isCandidateIndelImpl also notes, in the same cached status, that an externally specified indel is a candidate WITHOUT read
            // support -- IndelBuffer.cpp:279-289; the search reads that note, starling_read_align.cpp:1003, :1026, of indels whose status a
            // read of the reference has always just computed.  It is taken from the evaluation made here, not from the restored cache:
            // found by tools/fuzz/e2e_seeds.py adversarial -- a forced-output indel no read carries, first asked about by a later window)
            const bool isNotDiscoveredFromReads(d.status.notDiscoveredFromReads);
            d.status = savedStatus;
            toIndelKey(k, isCandidate, e.key);
            const IndelSampleData& isd(d.getSampleData(sampleIndex));
            e.ref_to_indel_log_prob = isd.getErrorRates().refToIndelErrorProb.getLogValue();
            e.indel_to_ref_log_prob = isd.getErrorRates().indelToRefErrorProb.getLogValue();
            e.active_region_id = static_cast<int32_t>(d.activeRegionId);
            for (unsigned si(0); si < sampleCount; ++si)
            {
                e.haplotype_id[si] = static_cast<int8_t>(d.getSampleData(si).haplotypeId);
                e.is_haplotyping_bypassed[si] = d.getSampleData(si).isHaplotypingBypassed ? 1 : 0;
            }
            e.is_forced_output = d.isForcedOutput ? 1 : 0;
            e.not_discovered_from_reads = isNotDiscoveredFromReads ? 1 : 0;
            const int32_t index(static_cast<int32_t>(table.size()));
            // is_usable_indel (starling_read_align.cpp:289-305): the reads observed to carry this indel in this sample
            for (const auto id : isd.tier1_map_read_ids) observedBy[id].push_back(index);
            for (const auto id : isd.tier2_map_read_ids) observedBy[id].push_back(index);
            for (const auto id : isd.submap_read_ids) observedBy[id].push_back(index);
            for (const auto id : isd.noise_read_ids) observedBy[id].push_back(index);
            table.push_back(e);
            keys.push_back(&k);
            data.push_back(&d);
        }
    }

    // Most reads of a window meet no indel at all: realignAndScoreRead returns at check_for_candidate_indel_overlap
    // (starling_read_align.cpp:2047, :219-270), whose range query (IndelBuffer.cpp:76-92) finds nothing -- no candidate status is
    // asked for, nothing is cached.  Those reads need not travel: a read stays in the job iff some table entry starts inside
    // [zone begin - maxIndelSize - 1, zone end + 1], a superset of what that query can return for it.
    {
        std::vector<pos_t> keyPos;
        keyPos.reserve(keys.size());
        for (const IndelKey* k : keys) keyPos.push_back(k->pos); // the IndelBuffer's order: ascending position
        std::vector<WindowRead> kept;
        for (const WindowRead& wr : reads)
        {
            const pos_t lo(wr.zoneBegin - static_cast<pos_t>(opt.maxIndelSize) - 1), hi(wr.zoneEnd + 1);
            const auto it(std::lower_bound(keyPos.begin(), keyPos.end(), lo));
            if (it != keyPos.end() && *it <= hi) kept.push_back(wr);
        }
        reads.swap(kept);
    }
    if (reads.empty()) return;
    for (WindowRead& wr : reads)
    {
        const read_segment& rseg(*wr.rseg);
        wr.codeOffset = codes.size();
        const bam_seq bseq(rseg.get_bam_read());
        const unsigned readSize(rseg.read_size());
        for (unsigned i(0); i < readSize; ++i) codes.push_back(bseq.get_code(static_cast<pos_t>(i)));
        wr.pathOffset = paths.size();
        for (const auto& ps : rseg.getInputAlignment().path)
        {
            sk_path_seg seg;
            seg.type = static_cast<uint32_t>(ps.type);
            seg.length = ps.length;
            paths.push_back(seg);
        }
    }

    std::vector<int32_t> observed;
    for (WindowRead& wr : reads)
    {
        const auto it(observedBy.find(wr.rseg->getReadIndex()));
        if (it == observedBy.end()) continue;
        wr.observedOffset = observed.size();
        std::vector<int32_t> ids(it->second);
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        wr.observedCount = static_cast<unsigned>(ids.size());
        observed.insert(observed.end(), ids.begin(), ids.end());
    }

    // ---- the job ----
    sk_realign_options ro;
    sk_realign_options_default(&ro);
    ro.max_read_indel_toggle = opt.max_read_indel_toggle;
    ro.max_candidate_indel_density = opt.max_candidate_indel_density;
    ro.max_realignment_candidates = opt.max_realignment_candidates;
    ro.max_indel_size = opt.maxIndelSize;
    ro.is_smoothed_alignments = opt.is_smoothed_alignments ? 1 : 0;
    ro.smoothed_lnp_range = opt.smoothed_lnp_range;
    ro.upstream_oligo_size = opt.upstream_oligo_size;
    ro.is_haplotyping_enabled = opt.isHaplotypingEnabled ? 1 : 0;
    ro.min_read_bp_flank = sif.sampleOptions.min_read_bp_flank;
    ro.sample_count = static_cast<int32_t>(sampleCount);
    // Every job's search, flattening, scoring and stage 3 run on the device (enumeration 2): as ONE fixed sequence of launches with one
    // host wait (csrc/read_enumerate.hip, job_scan_kernel), staged in and out by kernels, with the window of the reference its reads can
    // reach.  In that form a job costs less inside the C-ABI than the host statement of the search + one scoring launch did, alone on a
    // GPU (0.25 against 0.33 ms) and with eight caller processes sharing it (0.35 against 0.39 ms; profiles/r05_enum_job_history.txt) --
    // round 4's three-wait form, with the whole contig segment copied and uploaded per job, cost 0.56 ms and the adapter kept jobs under
    // 512 reads on the host.  $STRELKA_AMD_DEVICE_ENUM_MIN_READS brings a threshold back; an explicit $SK_ENUMERATION decides for every
    // job (the tests run whole suites in one mode).
    {
        static const bool isModePinned(std::getenv("SK_ENUMERATION") != nullptr);
        static const size_t minDeviceReads([]() { const char* v(std::getenv("STRELKA_AMD_DEVICE_ENUM_MIN_READS")); return (v && *v) ? static_cast<size_t>(std::strtoul(v, nullptr, 10)) : size_t(1); }());
        if ((! isModePinned) && ro.enumeration == 2 && reads.size() < minDeviceReads) ro.enumeration = 0;
        if (ro.enumeration != 2) s.realignHostJobs++;
    }
    if (opt.isRetainOptimalSoftClipping) throw blt_exception("strelka_amd adapter: --retain-optimal-soft-clipping (RNA) is not supported on this path");

    struct JobHolder
    {
        sk_realign_job* job;
        ~JobHolder() { if (job) sk_realign_job_destroy(job); }
        void reset(sk_realign_job* j) { if (job) sk_realign_job_destroy(job); job = j; }
    } holder{nullptr};
    sk_realign_job* job(nullptr);

    auto jobCheck = [&](const int rc, const char* what)
    {
        if (rc == 0) return;
        throw blt_exception((std::string("strelka_amd: ") + what + ": " + sk_realign_job_error(job)).c_str());
    };
    // The job keeps a copy of its reference (and uploads it): of the contig segment -- megabases -- it gets the WINDOW its reads can
    // reach, the union of their realignment ranges and alignment zones plus a margin.  Whether that was enough is not argued but
    // counted: the library counts every reference base a job reads outside what it was given (such a base reads as 'N'); if the count
    // moved with a window narrower than the segment, the job runs again with the whole segment (realign_ref_window_misses).
    const pos_t segBegin(static_cast<pos_t>(ref.get_offset())), segEnd(segBegin + static_cast<pos_t>(ref.seq().size()));
    pos_t winBegin(segBegin), winEnd(segEnd);
    {
        static const bool isWholeSegment([]() { const char* v(std::getenv("STRELKA_AMD_REALIGN_WHOLE_REFERENCE")); return v && *v && *v != '0'; }());
        if (! isWholeSegment)
        {
            pos_t lo(reads.front().rangeBegin), hi(reads.front().rangeEnd);
            for (const WindowRead& wr : reads)
            {
                lo = std::min(lo, std::min(wr.rangeBegin, wr.zoneBegin));
                hi = std::max(hi, std::max(wr.rangeEnd, wr.zoneEnd));
            }
            const pos_t margin(static_cast<pos_t>(opt.maxIndelSize) + 64);
            winBegin = std::max(segBegin, lo - margin);
            winEnd = std::min(segEnd, hi + margin);
            if (winEnd < winBegin) winEnd = winBegin;
        }
    }
    std::vector<sk_read_input> inputs(reads.size());
    for (int attempt(0); attempt < 2; ++attempt)
    {
    holder.reset(sk_realign_job_create(&ro));
    job = holder.job;
    if (job == nullptr) throw blt_exception("strelka_amd adapter: sk_realign_job_create failed");
    const int64_t outsideBefore(sk_realign_reference_reads_outside());
    jobCheck(sk_realign_job_set_reference(job, ref.seq().data() + (winBegin - segBegin), static_cast<int32_t>(winBegin),
                                          static_cast<int32_t>(winEnd - winBegin)), "sk_realign_job_set_reference");
    jobCheck(sk_realign_job_set_indels(job, table.data(), static_cast<int32_t>(table.size())), "sk_realign_job_set_indels");

    for (size_t i(0); i < reads.size(); ++i)
    {
        const WindowRead& wr(reads[i]);
        const read_segment& rseg(*wr.rseg);
        const alignment& al(rseg.getInputAlignment());
        sk_read_input& in(inputs[i]);
        std::memset(&in, 0, sizeof(in));
        in.read_code = codes.data() + wr.codeOffset;
        in.read_qual = rseg.qual();
        in.read_len = static_cast<int32_t>(rseg.read_size());
        in.pos = al.pos;
        in.n_seg = static_cast<int32_t>(al.path.size());
        in.path = paths.data() + wr.pathOffset;
        in.is_fwd_strand = al.is_fwd_strand ? 1 : 0;
        in.map_level = static_cast<int32_t>(rseg.getInputAlignmentMapLevel());
        in.sample_index = static_cast<int32_t>(sampleIndex);
        in.realign_begin = wr.rangeBegin;
        in.realign_end = wr.rangeEnd;
        in.n_observed = static_cast<int32_t>(wr.observedCount);
        in.observed = wr.observedCount ? observed.data() + wr.observedOffset : nullptr;
    }
    {
        AccumTimer abiTimer(s.tRealignAbi);
        if (sk_realign_job_add_reads(job, inputs.data(), static_cast<int32_t>(inputs.size())) < 0)
        {
            jobCheck(1, "sk_realign_job_add_reads");
        }
        jobCheck(sk_realign_job_run(job), "sk_realign_job_run");
    }
    if (sk_realign_reference_reads_outside() == outsideBefore || (winBegin == segBegin && winEnd == segEnd)) break;
    // a read reached past the window (or past the segment's own end, where the reference reads 'N' as well): once more, whole segment
    s.realignRefWindowMisses++;
    winBegin = segBegin;
    winEnd = segEnd;
    }
    s.realignBatches++;
    s.realignJobReads += reads.size();
    {
        int64_t onCore(0), onDevice(0), onHostInstead(0);
        (void)sk_realign_job_enumeration_counts(job, &onCore, &onDevice, &onHostInstead);
        s.realignDeviceEnumerated += static_cast<unsigned long>(onDevice);
        s.realignHostEnumerated += static_cast<unsigned long>(onHostInstead);
    }

    // commit the candidate status of the indels the job's reads asked about
    {
        std::vector<uint8_t> consulted(table.size() + 1, 0);
        jobCheck(sk_realign_job_indels_consulted(job, consulted.data(), static_cast<int32_t>(table.size())), "sk_realign_job_indels_consulted");
        for (size_t i(0); i < table.size(); ++i)
        {
            if (consulted[i]) (void)indelBuffer.isCandidateIndel(*keys[i], *data[i]);
        }
    }

    // ---- results, written where the reference writes them ----
    const bool isMaxToggleWarnEnabled(opt.verbosity >= LOG_LEVEL::ALLWARN);
    for (size_t i(0); i < reads.size(); ++i)
    {
        const WindowRead& wr(reads[i]);
        read_segment& rseg(*wr.rseg);
        sk_read_result res;
        jobCheck(sk_realign_job_read_result(job, static_cast<int32_t>(i), &res), "sk_realign_job_read_result");
        if (res.n_candidate_alignments == 0) continue;

        if (res.warn_origin_skip || (res.warn_max_toggle_depth && isMaxToggleWarnEnabled))
        {
            auto writeSkipWarning = [&rseg](const char* reason)
            {
                log_os << "WARNING: re-alignment skipped some alternate alignments for read: " << rseg.key() << "\n"
                       << "\treason: " << reason << "\n";
            };
            if (res.warn_origin_skip) writeSkipWarning("alignments crossed chromosome origin");
            if (res.warn_max_toggle_depth && isMaxToggleWarnEnabled) writeSkipWarning("exceeded max number of indel switches");
        }

        if (res.is_realigned)
        {
            rseg.is_realigned = true;
            rseg.realignment.pos = res.realign_pos;
            rseg.realignment.is_fwd_strand = rseg.getInputAlignment().is_fwd_strand;
            rseg.realignment.path.clear();
            for (int32_t k(0); k < res.realign_n_seg; ++k)
            {
                rseg.realignment.path.push_back(ALIGNPATH::path_segment(static_cast<ALIGNPATH::align_t>(res.realign_path[k].type),
                                                                         res.realign_path[k].length));
            }
            // align_pos's check that the read was not moved behind the POST_ALIGN stage (:761-770), against the
            // reference's own geometry
            if (! (rseg.realignment.pos > wr.geometry.validThreshold))
            {
                log_os << "WARNING: read realigned outside bounds of realignment stage buffer. Skipping...\n"
                       << "\tread: " << rseg.key() << "\n";
                rseg.is_invalid_realignment = true;
            }
        }

        const align_id_t readId(rseg.getReadIndex());
        for (int32_t k(0); k < res.n_scores; ++k)
        {
            const sk_read_path_scores& sc(res.scores[k]);
            ReadPathScores rps(sc.ref_lnp, sc.indel_lnp, sc.non_ambig, sc.read_length, sc.is_tier1_read != 0,
                               sc.is_fwd_strand != 0, sc.read_pos, sc.distance_from_closest_read_edge);
            for (int32_t a(0); a < sc.n_alt; ++a)
            {
                rps.alt_indel.push_back(std::make_pair(*keys[sc.alt_indel[a]], sc.alt_lnp[a]));
            }
            data[sc.indel]->getSampleData(sampleIndex).read_path_lnp[readId] = rps;
        }
        const bool isTier1(rseg.is_tier1_mapping());
        for (int32_t k(0); k < res.n_suboverlap; ++k)
        {
            IndelSampleData& isd(data[res.suboverlap[k]]->getSampleData(sampleIndex));
            if (isTier1) isd.suboverlap_tier1_read_ids.insert(readId);
            else isd.suboverlap_tier2_read_ids.insert(readId);
        }
    }
}

}

bool align_pos(starling_pos_processor_base& pp, const pos_t pos)
{
    State& s(state());
    if (s.isAnyRealigned && pos < s.realignedTo) return true;
    const pos_t end(pos + static_cast<pos_t>(std::max(1u, read_buffer_defer() + 1)));
    const unsigned sampleCount(Access::sampleCount(pp));
    // the window's read segments in read-buffer order, once: the realignment job and the pileup push (site 9) both walk them
    s.windowSegments.resize(sampleCount);
    for (unsigned sampleIndex(0); sampleIndex < sampleCount; ++sampleIndex)
    {
        std::vector<WindowSegment>& segs(s.windowSegments[sampleIndex]);
        segs.clear();
        starling_pos_processor_base::sample_info& sif(pp.sample(sampleIndex));
        for (pos_t p(pos); p < end; ++p)
        {
            read_segment_iter ri(sif.readBuffer.get_pos_read_segment_iter(p));
            for (read_segment_iter::ret_val r; true; ri.next())
            {
                r = ri.get_ptr();
                if (nullptr == r.first) break;
                if (r.second != 0) throw blt_exception("strelka_amd adapter: spliced (RNA) read segments are not supported on this path");
                WindowSegment ws;
                ws.rseg = &(r.first->get_segment(r.second));
                ws.bufferPos = p;
                segs.push_back(ws);
            }
        }
    }
    for (unsigned sampleIndex(0); sampleIndex < sampleCount; ++sampleIndex)
    {
        try
        {
            realign_sample_window(pp, sampleIndex, pos, end);
        }
        catch (...)
        {
            log_os << "Exception caught in align_pos() while realigning the reads buffered at positions [" << (pos + 1) << ","
                   << end << "] of sample " << sampleIndex << "\n";
            throw;
        }
    }
    s.isAnyRealigned = true;
    s.realignedTo = end;
    return true;
}

}
