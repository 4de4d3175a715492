// sk_adapter_pileup.cpp -- site 9: pileup_pos_reads (L/starling_common/starling_pos_processor_base.cpp:1107-1123, called at :813)
// -> pileup_read_segment (:1127-1421), chained on the device into sites 2+3 (adjust_joint_eprob + position_snp_call_pprob_digt).
//
// The reference piles the reads buffered at position P up when its READ_BUFFER stage reaches P, one basecall at a time into the
// position-keyed pos_basecall_buffer, and genotypes P when POST_ALIGN gets there.  Here, when the (deferred) READ_BUFFER stage
// reaches the first position of a stage window -- right after the window's realignment job (sk_adapter_realign.cpp) -- the
// window's reads go, per sample, through one sk_pileup_stream_push with their best alignments: the kernels build the columns,
// genotype the positions no later read can reach (everything below window end - largest_total_indel_ref_span_per_read, the
// POST_ALIGN position of the window's end) and hand back, for exactly those positions, what the reference keeps per position:
// calls, tier2_calls, spanningDeletionReadCount, submappedReadCount, the MapqTracker.  They are written into the reference's own
// pos_basecall_buffer in bulk, so everything downstream (CleanPileupFilter, site / indel locus info, gVCF) runs unchanged; the
// genotypes wait in a per-sample cache for process_pos_snp_digt (sk_adapter_germline.cpp).
//
// The somatic caller's two samples go through ONE stream (sk_somatic_pileup_stream_*): both are pushed together, share the finalised
// range, and their four cleaned columns go on the device straight into position_somatic_snv_call's kernels (site 5); the records
// wait for process_pos_snp_somatic (sk_adapter_somatic.cpp).  With the somatic EVS models loaded the reference also feeds, for
// every tumor basecall, a rank-sum and a list of alternate-allele read positions (updateSomaticScoringMetrics) that only the
// writer of a somatic record ever reads: the stream returns each tumor call's read position and read length, and
// somatic_fill_scoring_metrics() rebuilds the two accumulators for a position just before its record is written.
//
// The germline EVS accumulators (updateGermlineScoringMetrics: three rank sums and a mean over every basecall) are only read where a
// variant record is scored: the stream returns the per-call arguments (P2's EVS column) and germline_fill_scoring_metrics() rebuilds
// a position's accumulators just before they are read.  Not routed (the reference's own pileup runs): $STRELKA_AMD_PILEUP=0.
#include "sk_adapter_access.hh"

#include "blt_util/log.hh"
#include "blt_util/seq_util.hh"
#include "starling_common/starling_read_segment.hh"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace sk_adapter
{

namespace
{

bool env_flag(const char* name, const bool def)
{
    const char* v(std::getenv(name));
    if (v == nullptr || *v == 0) return def;
    return std::strtol(v, nullptr, 10) != 0;
}

void pileupOptions(const starling_base_options& opt, sk_pileup_options& po)
{
    sk_pileup_options_default(&po);
    po.min_basecall_qscore = opt.minBasecallErrorPhredProb;
    po.mismatch_density_flank_size = opt.isMismatchDensityFilter() ? static_cast<int32_t>(opt.mismatchDensityFilterFlankSize) : 0;
    po.mismatch_density_max_count = static_cast<int32_t>(opt.mismatchDensityFilterMaxMismatchCount);
    po.use_tier2_evidence = opt.useTier2Evidence ? 1 : 0;
    po.tier2_mismatch_density_max_count = static_cast<int32_t>(opt.tier2.mismatchDensityFilterMaxMismatchCount);
    po.is_mapq_adjust = opt.isBasecallQualAdjustedForMapq ? 1 : 0;
    po.min_distance_from_read_edge = static_cast<int32_t>(opt.minDistanceFromReadEdge);
    // the read-level test against get_largest_total_indel_ref_span_per_read() (:1186-1191) is made here, per read, with the value
    // the reference had when IT piled the read up (geometry shadow); the kernel's copy of the test never fires
    po.largest_total_indel_ref_span_per_read = INT_MAX / 4;
}

void germlineOptionsForStream(const starling_base_options& opt, sk_germline_options& go)
{
    sk_germline_options_default(&go);
    go.bsnp_diploid_theta = opt.bsnp_diploid_theta;
    go.bsnp_ssd_no_mismatch = opt.bsnp_ssd_no_mismatch;
    go.bsnp_ssd_one_mismatch = opt.bsnp_ssd_one_mismatch;
    go.is_min_vexp = opt.is_min_vexp ? 1 : 0;
    go.min_vexp = opt.min_vexp;
}

struct WindowBatch
{
    std::vector<int64_t> readOff, pathOff;
    std::vector<uint8_t> code, qual, isFwd, mapq, level;
    std::vector<sk_path_seg> path;
    std::vector<int32_t> pos;
    pos_t lo = INT_MAX, hi = INT_MIN;
    void clear()
    {
        readOff.assign(1, 0);
        pathOff.assign(1, 0);
        code.clear(); qual.clear(); isFwd.clear(); mapq.clear(); level.clear(); path.clear(); pos.clear();
        lo = INT_MAX;
        hi = INT_MIN;
    }
};

/// the reads buffered in [begin, end) of one sample with their best alignments, as pileup_read_segment would take them
void gatherWindow(starling_pos_processor_base& pp, const unsigned sampleIndex, const pos_t begin, const pos_t end, WindowBatch& wb)
{
    State& s(state());
    const starling_base_options& opt(Access::opt(pp));
    wb.clear();
    static const std::vector<WindowSegment> noSegments;
    for (const WindowSegment& ws : (begin < end ? s.windowSegments[sampleIndex] : noSegments))
    {
        const read_segment& rseg(*ws.rseg);
        const pos_t pos(ws.bufferPos);

        // pileup_read_segment's read-level exits (:1132-1165), messages included
        const alignment* best(&(rseg.getInputAlignment()));
        if (rseg.is_realigned) best = &(rseg.realignment);
        else if (! rseg.is_any_nonovermax(opt.maxIndelSize)) continue;
        if (best->empty())
        {
            if (! rseg.is_realigned)
            {
                if (opt.verbosity >= LOG_LEVEL::ALLWARN)
                {
                    log_os << "WARNING: skipping read_segment with no genomic alignment and contig alignment outside of indel.\n";
                    log_os << "\tread_name: " << rseg.key() << "\n";
                }
            }
            else
            {
                log_os << "WARNING: skipping read_segment which has multiple equally likely but incompatible alignments: " << rseg.key() << "\n";
            }
            continue;
        }
        if (rseg.is_realigned && rseg.is_invalid_realignment) continue;

        const unsigned readSize(rseg.read_size());
        const unsigned refSpan(ALIGNPATH::apath_ref_length(best->path));
        // :1186-1191 with the reference's value at the time its READ_BUFFER stage was at this read
        const unsigned spanThen(static_cast<unsigned>(s.geometry.query(pos).rangeMinOffset) + 1);
        if (refSpan > (readSize + spanThen)) continue;

        const size_t c0(wb.code.size());
        wb.code.resize(c0 + readSize);
        const uint8_t* q(rseg.qual());
        if (rseg.full_read_offset() == 0)
        {
            // the BAM record's packed bases sit right before its qualities (bam_get_qual = bam_get_seq + (l_qseq + 1) / 2): two codes per
            // byte, unpacked a byte at a time (bam_seq::get_code: high nibble first)
            static uint16_t pairOf[256];
            static bool isPairTable(false);
            if (! isPairTable)
            {
                for (unsigned b(0); b < 256; ++b) pairOf[b] = static_cast<uint16_t>((b >> 4) | ((b & 15u) << 8));
                isPairTable = true;
            }
            const uint8_t* packed(q - ((rseg.full_read_size() + 1) >> 1));
            uint8_t* dst(wb.code.data() + c0);
            unsigned i(0);
#if defined(__SSE2__)
            // sixteen packed bytes -> thirty-two codes per step (high nibble first): a read's 75 bytes are five steps instead of 75 look-ups
            {
                const __m128i low4(_mm_set1_epi8(0x0f));
                for (; i + 32 <= readSize; i += 32)
                {
                    const __m128i v(_mm_loadu_si128(reinterpret_cast<const __m128i*>(packed + (i >> 1))));
                    const __m128i hi(_mm_and_si128(_mm_srli_epi16(v, 4), low4)), lo(_mm_and_si128(v, low4));
                    _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), _mm_unpacklo_epi8(hi, lo));
                    _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i + 16), _mm_unpackhi_epi8(hi, lo));
                }
            }
#endif
            for (; i + 2 <= readSize; i += 2) std::memcpy(dst + i, &pairOf[packed[i >> 1]], 2);
            if (i < readSize) dst[i] = static_cast<uint8_t>(packed[i >> 1] >> 4);
        }
        else
        {
            const bam_seq bseq(rseg.get_bam_read());
            for (unsigned i(0); i < readSize; ++i) wb.code[c0 + i] = bseq.bam_seq::get_code(static_cast<pos_t>(i)); // (qualified: no virtual dispatch)
        }
        wb.qual.insert(wb.qual.end(), q, q + readSize);
        for (const auto& seg : best->path)
        {
            sk_path_seg out;
            out.type = static_cast<uint32_t>(seg.type);
            out.length = seg.length;
            wb.path.push_back(out);
        }
        wb.readOff.push_back(static_cast<int64_t>(wb.code.size()));
        wb.pathOff.push_back(static_cast<int64_t>(wb.path.size()));
        wb.pos.push_back(best->pos);
        wb.isFwd.push_back(best->is_fwd_strand ? 1 : 0);
        wb.mapq.push_back(rseg.map_qual());
        wb.level.push_back(static_cast<uint8_t>(rseg.getInputAlignmentMapLevel()));
        wb.lo = std::min(wb.lo, best->pos);
        wb.hi = std::max(wb.hi, best->pos + static_cast<pos_t>(refSpan));
    }
}

void fillReadBatch(const WindowBatch& wb, sk_read_batch& rb)
{
    std::memset(&rb, 0, sizeof(rb));
    rb.n_reads = static_cast<int32_t>(wb.pos.size());
    static const uint8_t none8(0);
    static const sk_path_seg noneSeg = {0u, 0u};
    static const int32_t none32(0);
    rb.read_off = wb.readOff.data();
    rb.read_code = wb.code.empty() ? &none8 : wb.code.data();
    rb.read_qual = wb.qual.empty() ? &none8 : wb.qual.data();
    rb.path_off = wb.pathOff.data();
    rb.path = wb.path.empty() ? &noneSeg : wb.path.data();
    rb.pos = wb.pos.empty() ? &none32 : wb.pos.data();
    rb.is_fwd = wb.isFwd.empty() ? &none8 : wb.isFwd.data();
    rb.mapq = wb.mapq.empty() ? &none8 : wb.mapq.data();
    rb.map_level = wb.level.empty() ? &none8 : wb.level.data();
}

/// CandidateSnvBuffer::isCandidateSnvAnySample over [lo, hi) clipped to the reference segment, as of now
void candidateMask(starling_pos_processor_base& pp, const pos_t lo, const pos_t hi, std::vector<uint8_t>& mask, pos_t& maskBegin,
                   pos_t& maskEnd)
{
    const reference_contig_segment& ref(Access::ref(pp));
    maskBegin = std::max(lo, static_cast<pos_t>(ref.get_offset()));
    maskEnd = std::min(hi, static_cast<pos_t>(ref.get_offset()) + static_cast<pos_t>(ref.seq().size()));
    if (maskEnd < maskBegin) maskEnd = maskBegin;
    mask.assign(static_cast<size_t>(maskEnd - maskBegin), 0);
    const CandidateSnvBuffer& csb(Access::candidateSnvBuffer(pp));
    if (! csb.empty())
    {
        const unsigned sampleCount(Access::sampleCount(pp));
        for (pos_t p(maskBegin); p < maskEnd; ++p)
        {
            uint8_t m(0);
            for (unsigned si(0); si < sampleCount; ++si)
            {
                for (unsigned b(0); b < 4; ++b)
                {
                    if (csb.getHaplotypeId(si, p, static_cast<BASE_ID::index_t>(b)) != 0) m |= static_cast<uint8_t>(1u << b);
                }
            }
            mask[static_cast<size_t>(p - maskBegin)] = m;
        }
    }
}

/// the finalised positions of a window into the reference's pos_basecall_buffer (what insert_pos_basecall / insert_mapq_count /
/// insert_pos_spandel_count / insert_pos_submap_count would have left there)
void assignWindow(starling_pos_processor_base::sample_info& sif, const sk_pileup_window& w)
{
    const size_t n(static_cast<size_t>(w.end - w.begin));
    static_assert(sizeof(base_call) == 2, "base_call is the 16-bit record the kernels write");
    for (size_t i(0); i < n; ++i)
    {
        const uint32_t mq(w.mapq_count[i]), sd(w.spandel_count[i]), sm(w.submapped_count[i]);
        if ((mq | sd | sm) == 0) continue; // untouched: the reference has no entry either
        snp_pos_info& pi(Access::pileupRef(sif.basecallBuffer, w.begin + static_cast<pos_t>(i)));
        const size_t n1(static_cast<size_t>(w.tier1_off[i + 1] - w.tier1_off[i]));
        const size_t n2(static_cast<size_t>(w.tier2_off[i + 1] - w.tier2_off[i]));
        // (base_call is the 16-bit record itself: the column is a run of them)
        const base_call* const t1(reinterpret_cast<const base_call*>(w.tier1_calls + w.tier1_off[i]));
        const base_call* const t2(reinterpret_cast<const base_call*>(w.tier2_calls + w.tier2_off[i]));
        pi.calls.assign(t1, t1 + n1);
        if (n2 != 0 || (! pi.tier2_calls.empty())) pi.tier2_calls.assign(t2, t2 + n2); // (a germline column has next to no tier2 calls: no call for nothing)
        pi.spanningDeletionReadCount = sd;
        pi.submappedReadCount = sm;
        pi.mapqTracker.count = mq;
        pi.mapqTracker.zeroCount = w.mapq_zero_count[i];
        pi.mapqTracker.sumSquare = static_cast<double>(w.mapq_sum_square[i]);
    }
}

void pileup_complete_push(starling_pos_processor_base& pp, const unsigned sampleIndex);
void pileup_complete_somatic_push(starling_pos_processor_base& pp);

static bool isPushAsync()
{
    static const bool on([] { const char* v(std::getenv("STRELKA_AMD_PUSH_ASYNC")); return v == nullptr || *v == 0 || std::atoi(v) != 0; }());
    return on;
}

/// one sample's window into its stream (the push is begun here and finished by pileup_complete_push)
void pileup_sample_window(starling_pos_processor_base& pp, const unsigned sampleIndex, const pos_t begin, const pos_t end,
                          const bool isFinal)
{
    State& s(state());
    PileupState& ps(s.pileup);
    if (sampleIndex >= ps.pending.size()) ps.pending.resize(sampleIndex + 1);
    pileup_complete_push(pp, sampleIndex); // (the sample's last window, if POST_ALIGN has not asked for it yet)
    const reference_contig_segment& ref(Access::ref(pp));
    sk_pileup_stream* stream(ps.streams[sampleIndex]);

    if (! ps.isRegionOpen[sampleIndex])
    {
        check(sk_pileup_stream_begin_region(stream, ref.seq().data(), static_cast<int32_t>(ref.get_offset()),
                                            static_cast<int32_t>(ref.seq().size()), ps.regionBegin, ps.regionEnd,
                                            static_cast<int32_t>(Access::largestTotalIndelRefSpanPerRead(pp))), "sk_pileup_stream_begin_region");
        gvcf_configure_stream(pp, sampleIndex, stream); // (site 10: the chromosome's depth ceiling with the other block options)
        ps.isRegionOpen[sampleIndex] = 1;
    }

    static WindowBatch wb;
    {
        AccumTimer gatherTimer(s.tPileupGather);
        gatherWindow(pp, sampleIndex, begin, end, wb);
    }

    const pos_t span(static_cast<pos_t>(Access::largestTotalIndelRefSpanPerRead(pp)));
    const int32_t finalTo(isFinal ? INT32_MAX : static_cast<int32_t>(end - span));

    static std::vector<uint8_t> mask;
    pos_t maskBegin(0), maskEnd(0);
    mask.clear();
    if (! wb.pos.empty()) candidateMask(pp, wb.lo, wb.hi, mask, maskBegin, maskEnd);

    // caller ploidy of the positions this push can finalise (process_pos_snp_digt "prep step 2", starling_pos_processor.cpp:637-651,
    // without the indel calls' adjustment, which is not known yet: site_diploid_genotype() checks it when the locus is called)
    static std::vector<uint8_t> ploidy;
    pos_t ploidyBegin(0);
    const uint8_t* ploidyPtr(nullptr);
    int32_t ploidyLen(0);
    if (ps.isGenotyping && Access::hasPloidyRegions(pp, sampleIndex))
    {
        ploidyBegin = std::max(ps.regionBegin, std::min(ps.nextFinal[sampleIndex], ps.regionEnd));
        // one past the last position this push can finalise
        pos_t reach(isFinal ? ps.pendingEnd[sampleIndex] : static_cast<pos_t>(finalTo));
        if (isFinal && (! wb.pos.empty())) reach = std::max(reach, wb.hi);
        const pos_t ploidyEnd(std::max(ploidyBegin, std::min(reach, ps.regionEnd)));
        ploidy.resize(static_cast<size_t>(ploidyEnd - ploidyBegin));
        for (pos_t p(ploidyBegin); p < ploidyEnd; ++p)
        {
            const unsigned pl(Access::ploidy(pp, p, sampleIndex));
            ploidy[static_cast<size_t>(p - ploidyBegin)] = static_cast<uint8_t>(pl == 0 ? 2 : pl);
        }
        ploidyPtr = ploidy.data();
        ploidyLen = static_cast<int32_t>(ploidy.size());
    }

    sk_read_batch rb;
    fillReadBatch(wb, rb);

    // this push writes the output block that the push SK_PILEUP_WINDOW_LIFETIME + 1 before it wrote: a chunk that still reads its EVS
    // words there takes its copy now -- unless POST_ALIGN is through with its positions
    if (sampleIndex >= ps.pushCount.size()) ps.pushCount.resize(sampleIndex + 1, 0);
    ps.pushCount[sampleIndex]++;
    if (ps.isGermlineMetrics && sampleIndex < ps.chunks.size())
    {
        for (SiteChunk& c : ps.chunks[sampleIndex])
        {
            if (c.evsLive == nullptr || c.evsPush + SK_PILEUP_WINDOW_LIFETIME >= ps.pushCount[sampleIndex]) continue;
            if (c.end - 1 <= ps.lastVariantsPos)
            {
                c.evsLive = nullptr; // (POST_ALIGN is through with every position of the chunk)
                c.isEvsGone = true;
                ps.evsWordsLeft++;
            }
            else
            {
                c.materializeEvsWords();
                ps.evsWordsCopied++;
            }
        }
    }
    {
        AccumTimer abiTimer(s.tPileupAbi);
        check(sk_pileup_stream_push_begin(stream, &rb, static_cast<int32_t>(span), static_cast<int32_t>(maskBegin),
                                          static_cast<int32_t>(maskEnd - maskBegin), mask.empty() ? nullptr : mask.data(), finalTo,
                                          static_cast<int32_t>(ploidyBegin), ploidyLen, ploidyPtr), "sk_pileup_stream_push_begin");
    }
    s.pileupBatches++;
    s.pileupReads += wb.pos.size();
    if (! wb.pos.empty()) ps.pendingEnd[sampleIndex] = std::max(ps.pendingEnd[sampleIndex], wb.hi);
    PendingPush& pend(ps.pending[sampleIndex]);
    pend.active = true;
    pend.isFinal = isFinal;
    pend.finalTo = finalTo;
    pend.ploidyBegin = ploidyBegin;
    pend.hasPloidy = (ploidyPtr != nullptr);
    if (pend.hasPloidy) pend.ploidy.assign(ploidyPtr, ploidyPtr + ploidyLen);
    else pend.ploidy.clear();
    if (! isPushAsync()) pileup_complete_push(pp, sampleIndex);
}

/// the window of a sample's push in flight, once the device is through with it: the finalised positions into the reference's buffers,
/// the records the later sites read into the sample's chunks
void pileup_complete_push(starling_pos_processor_base& pp, const unsigned sampleIndex)
{
    State& s(state());
    PileupState& ps(s.pileup);
    if (sampleIndex >= ps.pending.size() || ! ps.pending[sampleIndex].active) return;
    PendingPush& pend(ps.pending[sampleIndex]);
    pend.active = false;
    starling_pos_processor_base::sample_info& sif(pp.sample(sampleIndex));
    sk_pileup_stream* stream(ps.streams[sampleIndex]);
    const bool isFinal(pend.isFinal);
    const int32_t finalTo(pend.finalTo);
    const pos_t ploidyBegin(pend.ploidyBegin);
    const uint8_t* const ploidyPtr(pend.hasPloidy ? pend.ploidy.data() : nullptr);
    const int32_t ploidyLen(static_cast<int32_t>(pend.ploidy.size()));
    sk_pileup_window w;
    std::memset(&w, 0, sizeof(w));
    {
        AccumTimer abiTimer(s.tPileupAbi);
        check(sk_pileup_stream_push_finish(stream, &w), "sk_pileup_stream_push_finish");
    }

    {
        AccumTimer assignTimer(s.tPileupAssign);
        assignWindow(sif, w);
    }
    const size_t n(static_cast<size_t>(w.end - w.begin));
    s.pileupLoci += n;

    AccumTimer chunkTimer(s.tPileupChunk);
    if ((ps.isGenotyping || ps.isGermlineMetrics) && n > 0)
    {
        SiteChunk chunk;
        chunk.begin = w.begin;
        chunk.end = w.end;
        if (ps.isGenotyping)
        {
            chunk.calls.assign(w.genotype, w.genotype + n);
            chunk.cleanCount.assign(w.clean_count, w.clean_count + n);
            if (w.site_summary != nullptr)
            {
                chunk.summary.assign(w.site_summary, w.site_summary + n);
                if (w.gvcf_runs != nullptr) chunk.runs.assign(w.gvcf_runs, w.gvcf_runs + n);
                chunk.rawCount.resize(n);
                for (size_t i(0); i < n; ++i) chunk.rawCount[i] = static_cast<uint32_t>(w.tier1_off[i + 1] - w.tier1_off[i]);
                // test switch: every position's site summary as the window brought it (the kernel's, or over the CPU double the shared
                // statement's), one text line each -- tests/test_gvcf_site_reference.py sets them beside what the REFERENCE printed for
                // the same positions of the same sample (tests/golden/gvcf_site_reference.npz)
                static const char* const dumpPath(std::getenv("STRELKA_AMD_GVCF_SITE_DUMP"));
                if (dumpPath != nullptr && sampleIndex == 0)
                {
                    static FILE* dump(std::fopen(dumpPath, "w"));
                    if (dump != nullptr)
                    {
                        for (size_t i(0); i < n; ++i)
                        {
                            std::fprintf(dump, "%d %u %d %u %u %u %u\n", static_cast<int>(w.begin + static_cast<pos_t>(i)), w.site_summary[i].flags, w.site_summary[i].gqx,
                                         w.site_summary[i].ref_fwd, w.site_summary[i].ref_rev, w.clean_count[i], chunk.rawCount[i]);
                        }
                        std::fflush(dump);
                    }
                }
            }
            chunk.ploidy.resize(n);
            for (size_t i(0); i < n; ++i)
            {
                const int64_t k(static_cast<int64_t>(w.begin) + static_cast<int64_t>(i) - ploidyBegin);
                chunk.ploidy[i] = (ploidyPtr && k >= 0 && k < ploidyLen) ? ploidyPtr[k] : 2;
            }
            s.siteLoci += n;
            s.siteBatches++;
        }
        if (ps.isGermlineMetrics)
        {
            chunk.evsOff.assign(w.evs_off, w.evs_off + n + 1);
            chunk.evsLive = w.evs_words; // (no copy: SiteChunk::evsLive)
            chunk.evsPush = ps.pushCount[sampleIndex];
            chunk.isMetricsFilled.assign(n, 0);
        }
        ps.chunks[sampleIndex].push_back(std::move(chunk));
    }
    ps.nextFinal[sampleIndex] = isFinal ? INT_MAX : std::max(ps.nextFinal[sampleIndex], static_cast<pos_t>(finalTo));
}

/// the somatic caller: the normal and the tumor sample's windows into the one stream that holds both, chained into site 5
void pileup_somatic_window(starling_pos_processor_base& pp, const pos_t begin, const pos_t end, const bool isFinal)
{
    State& s(state());
    PileupState& ps(s.pileup);
    pileup_complete_somatic_push(pp); // (the last window, if POST_ALIGN has not asked for it yet)
    const reference_contig_segment& ref(Access::ref(pp));
    if (! ps.isRegionOpen[0])
    {
        check(sk_somatic_pileup_stream_begin_region(ps.somaticStream, ref.seq().data(), static_cast<int32_t>(ref.get_offset()),
                                                    static_cast<int32_t>(ref.seq().size()), ps.regionBegin, ps.regionEnd,
                                                    static_cast<int32_t>(Access::largestTotalIndelRefSpanPerRead(pp))),
              "sk_somatic_pileup_stream_begin_region");
        ps.isRegionOpen[0] = ps.isRegionOpen[1] = 1;
    }
    static WindowBatch wb[2];
    sk_read_batch rb[2];
    pos_t lo(INT_MAX), hi(INT_MIN);
    for (unsigned si(0); si < 2; ++si)
    {
        gatherWindow(pp, si, begin, end, wb[si]);
        fillReadBatch(wb[si], rb[si]);
        if (! wb[si].pos.empty())
        {
            lo = std::min(lo, wb[si].lo);
            hi = std::max(hi, wb[si].hi);
            ps.pendingEnd[si] = std::max(ps.pendingEnd[si], wb[si].hi);
        }
        s.pileupReads += wb[si].pos.size();
    }
    const pos_t span(static_cast<pos_t>(Access::largestTotalIndelRefSpanPerRead(pp)));
    const int32_t finalTo(isFinal ? INT32_MAX : static_cast<int32_t>(end - span));

    static std::vector<uint8_t> mask;
    pos_t maskBegin(0), maskEnd(0);
    mask.clear();
    if (lo != INT_MAX) candidateMask(pp, lo, hi, mask, maskBegin, maskEnd);

    // is_forced_output_pos of the positions this push can finalise
    static std::vector<uint8_t> forced;
    forced.clear();
    pos_t forcedBegin(0);
    if (ps.isGenotyping)
    {
        forcedBegin = std::max(ps.regionBegin, std::min(ps.nextFinal[0], ps.regionEnd));
        pos_t reach(isFinal ? std::max(ps.pendingEnd[0], ps.pendingEnd[1]) : static_cast<pos_t>(finalTo));
        const pos_t forcedEnd(std::max(forcedBegin, std::min(reach, ps.regionEnd)));
        forced.resize(static_cast<size_t>(forcedEnd - forcedBegin));
        for (pos_t p(forcedBegin); p < forcedEnd; ++p)
        {
            forced[static_cast<size_t>(p - forcedBegin)] = Access::isForcedOutputPos(pp, p) ? 1 : 0;
        }
    }

    sk_somatic_snv_options so;
    bool isComputeNonSomatic(false);
    (void)somatic_stream_options(pp, so, isComputeNonSomatic);

    {
        AccumTimer abiTimer(s.tPileupAbi);
        check(sk_somatic_pileup_stream_push_begin(ps.somaticStream, &rb[0], &rb[1], static_cast<int32_t>(span), static_cast<int32_t>(maskBegin),
                                                  static_cast<int32_t>(maskEnd - maskBegin), mask.empty() ? nullptr : mask.data(), finalTo,
                                                  static_cast<int32_t>(forcedBegin), static_cast<int32_t>(forced.size()),
                                                  forced.empty() ? nullptr : forced.data(), isComputeNonSomatic ? 1 : 0),
              "sk_somatic_pileup_stream_push_begin");
    }
    s.pileupBatches++;
    PendingPush& pend(ps.somaticPending);
    pend.active = true;
    pend.isFinal = isFinal;
    pend.finalTo = finalTo;
    pend.forcedBegin = forcedBegin;
    pend.forced = forced;
    if (! isPushAsync()) pileup_complete_somatic_push(pp);
}

/// the somatic stream's window in flight, once the device is through with it
void pileup_complete_somatic_push(starling_pos_processor_base& pp)
{
    State& s(state());
    PileupState& ps(s.pileup);
    if (! ps.somaticPending.active) return;
    PendingPush& pend(ps.somaticPending);
    pend.active = false;
    const bool isFinal(pend.isFinal);
    const int32_t finalTo(pend.finalTo);
    const pos_t forcedBegin(pend.forcedBegin);
    const std::vector<uint8_t>& forced(pend.forced);
    sk_somatic_pileup_window w;
    std::memset(&w, 0, sizeof(w));
    {
        AccumTimer abiTimer(s.tPileupAbi);
        check(sk_somatic_pileup_stream_push_finish(ps.somaticStream, &w), "sk_somatic_pileup_stream_push_finish");
    }
    assignWindow(pp.sample(0), w.normal);
    assignWindow(pp.sample(1), w.tumor);
    const size_t n(static_cast<size_t>(w.normal.end - w.normal.begin));
    s.pileupLoci += n;
    if (n > 0 && (ps.isGenotyping || ps.isSomaticMetrics))
    {
        SomaticChunk chunk;
        chunk.begin = w.normal.begin;
        chunk.end = w.normal.end;
        if (ps.isGenotyping)
        {
            chunk.genotypes.assign(w.genotype, w.genotype + n);
            chunk.count[0].assign(w.normal.clean_count, w.normal.clean_count + n);
            chunk.count[1].assign(w.tumor.clean_count, w.tumor.clean_count + n);
            chunk.count[2].assign(w.normal_clean_tier2_count, w.normal_clean_tier2_count + n);
            chunk.count[3].assign(w.tumor_clean_tier2_count, w.tumor_clean_tier2_count + n);
            const sk_pileup_window* ws[2] = {&w.normal, &w.tumor};
            for (unsigned si(0); si < 2; ++si)
            {
                chunk.rawCount[si].resize(n);
                chunk.rawCount[2 + si].resize(n);
                for (size_t i(0); i < n; ++i)
                {
                    chunk.rawCount[si][i] = static_cast<uint32_t>(ws[si]->tier1_off[i + 1] - ws[si]->tier1_off[i]);
                    chunk.rawCount[2 + si][i] = static_cast<uint32_t>(ws[si]->tier2_off[i + 1] - ws[si]->tier2_off[i]);
                }
            }
            chunk.forced.resize(n);
            for (size_t i(0); i < n; ++i)
            {
                const int64_t k(static_cast<int64_t>(chunk.begin) + static_cast<int64_t>(i) - forcedBegin);
                chunk.forced[i] = (k >= 0 && k < static_cast<int64_t>(forced.size())) ? forced[static_cast<size_t>(k)] : 0;
            }
            s.siteLoci += n;
            s.siteBatches++;
        }
        if (ps.isSomaticMetrics)
        {
            chunk.tumorTier1Off.assign(w.tumor.tier1_off, w.tumor.tier1_off + n + 1);
            chunk.tumorReadPos.assign(w.tumor_tier1_read_pos, w.tumor_tier1_read_pos + w.tumor.tier1_off[n]);
            chunk.isMetricsFilled.assign(n, 0);
        }
        ps.somaticChunks.push_back(std::move(chunk));
    }
    const pos_t next(isFinal ? INT_MAX : static_cast<pos_t>(finalTo));
    for (unsigned si(0); si < 2; ++si) ps.nextFinal[si] = isFinal ? INT_MAX : std::max(ps.nextFinal[si], next);
}

}

static bool pileup_routed(const starling_base_options& /*opt*/)
{
    return env_flag("STRELKA_AMD_PILEUP", true);
}

bool pileup_genotypes_with_stream(const starling_base_options& opt)
{
    // genotypes straight from the device columns: the diploid germline model, the somatic SNV model
    return pileup_routed(opt) && (opt.isSomaticCallingMode || opt.is_bsnp_diploid()) && env_flag("STRELKA_AMD_PILEUP_GENOTYPE", true);
}

bool pileup_enabled(starling_pos_processor_base& pp)
{
    PileupState& ps(state().pileup);
    if (ps.decided) return ps.enabled;
    const starling_base_options& opt(Access::opt(pp));
    ps.decided = true;
    ps.enabled = pileup_routed(opt);
    ps.isSomatic = ps.enabled && opt.isSomaticCallingMode;
    if (ps.isSomatic && Access::sampleCount(pp) != 2) throw blt_exception("strelka_amd adapter: the somatic pileup stream takes a normal and a tumor sample");
    ps.isSomaticMetrics = ps.isSomatic && opt.is_compute_somatic_scoring_metrics;
    ps.isGermlineMetrics = ps.enabled && (! ps.isSomatic) && opt.is_compute_germline_scoring_metrics();
    ps.isGenotyping = pileup_genotypes_with_stream(opt);
    if (ps.isSomatic && ps.isGenotyping)
    {
        sk_somatic_snv_options so;
        bool isComputeNonSomatic(false);
        ps.isGenotyping = somatic_stream_options(pp, so, isComputeNonSomatic);
    }
    return ps.enabled;
}

void pileup_reset_region(starling_pos_processor_base& pp)
{
    State& s(state());
    PileupState& ps(s.pileup);
    if (! pileup_enabled(pp)) return;
    const starling_base_options& opt(Access::opt(pp));
    const unsigned sampleCount(Access::sampleCount(pp));
    if (ps.isSomatic && ps.somaticStream == nullptr)
    {
        sk_pileup_options po;
        pileupOptions(opt, po);
        sk_somatic_snv_options so;
        bool isComputeNonSomatic(false);
        const bool isCalling(somatic_stream_options(pp, so, isComputeNonSomatic) && ps.isGenotyping);
        ps.somaticStream = sk_somatic_pileup_stream_create(&po, isCalling ? &so : nullptr, ps.isSomaticMetrics ? 1 : 0);
        if (ps.somaticStream == nullptr) check(1, "sk_somatic_pileup_stream_create");
    }
    if ((! ps.isSomatic) && ps.streams.empty())
    {
        sk_pileup_options po;
        pileupOptions(opt, po);
        sk_germline_options go;
        germlineOptionsForStream(opt, go);
        if (ps.isGenotyping && opt.isHetVariantFrequencyExtensionDefined())
        {
            throw blt_exception("strelka_amd adapter: --het-variant-frequency-extension (RNA) is not supported on this path");
        }
        for (unsigned i(0); i < sampleCount; ++i)
        {
            sk_pileup_stream* st(sk_pileup_stream_create(&po, ps.isGenotyping ? &go : nullptr));
            if (st == nullptr) check(1, "sk_pileup_stream_create");
            if (ps.isGermlineMetrics) check(sk_pileup_stream_enable_evs_words(st, 1), "sk_pileup_stream_enable_evs_words");
            ps.streams.push_back(st);
        }
    }
    // (a window still in flight -- POST_ALIGN never asked for a position it covers -- is finished before its stream starts over)
    for (unsigned i(0); i < ps.pending.size(); ++i) pileup_complete_push(pp, i);
    pileup_complete_somatic_push(pp);
    // (the reference segment of the region is loaded after resetRegion, starling_run.cpp:117-119: the streams get it with the
    // region's first push)
    const known_pos_range2& rr(Access::reportRange(pp));
    ps.regionBegin = rr.begin_pos();
    ps.regionEnd = rr.end_pos();
    ps.isRegionOpen.assign(sampleCount, 0);
    ps.chunks.assign(sampleCount, std::deque<SiteChunk>());
    ps.somaticChunks.clear();
    ps.nextFinal.assign(sampleCount, rr.begin_pos());
    ps.pendingEnd.assign(sampleCount, INT_MIN);
    ps.maxBufferPos.assign(sampleCount, INT_MIN);
    ps.isAnyPiled = false;
    ps.piledTo = 0;
    ps.isFlushing = false;
    ps.lastVariantsPos = INT_MIN;
}

bool pileup_pos_reads(starling_pos_processor_base& pp, const pos_t pos)
{
    if (! pileup_enabled(pp)) return false;
    State& s(state());
    PileupState& ps(s.pileup);
    if (ps.isAnyPiled && pos < ps.piledTo) return true;
    AccumTimer hookTimer(s.tPileupHook);
    // align_pos(pos) has just realigned the window [pos, realignedTo) (process_pos :812-813 calls them back to back)
    const pos_t end(s.realignedTo);
    if (! (s.isAnyRealigned && end > pos)) throw blt_exception("strelka_amd adapter: pileup window without its realignment job");
    const unsigned sampleCount(Access::sampleCount(pp));
    if (ps.isSomatic)
    {
        // the last window of a region: at the final flush, when no read of either sample is buffered beyond it
        const bool isFinal(ps.isFlushing && ps.maxBufferPos[0] < end && ps.maxBufferPos[1] < end);
        try
        {
            pileup_somatic_window(pp, pos, end, isFinal);
        }
        catch (...)
        {
            log_os << "Exception caught in pileup_pos_reads() while piling up the reads buffered at positions [" << (pos + 1) << "," << end
                   << "] of the normal and the tumor sample\n";
            throw;
        }
    }
    else
    {
        for (unsigned sampleIndex(0); sampleIndex < sampleCount; ++sampleIndex)
        {
            // the last window of a region: at the final flush, when no read is buffered beyond it
            const bool isFinal(ps.isFlushing && ps.maxBufferPos[sampleIndex] < end);
            try
            {
                pileup_sample_window(pp, sampleIndex, pos, end, isFinal);
            }
            catch (...)
            {
                log_os << "Exception caught in pileup_pos_reads() while piling up the reads buffered at positions [" << (pos + 1) << "," << end
                       << "] of sample " << sampleIndex << "\n";
                throw;
            }
        }
    }
    ps.isAnyPiled = true;
    ps.piledTo = end;
    return true;
}

void pileup_before_variants(starling_pos_processor_base& pp, const pos_t pos)
{
    // POST_ALIGN asks for a position that no push has finalised: only at the end of a region, when the READ_BUFFER stage has run
    // out of positions before the last window's tail was final
    State& s(state());
    PileupState& ps(s.pileup);
    if (! ps.enabled) return;
    ps.lastVariantsPos = std::max(ps.lastVariantsPos, pos - 1); // (POST_ALIGN is through with the positions below `pos`)
    if (ps.isSomatic)
    {
        // (chunks kept only for the EVS read positions -- the records are computed per site window -- are dropped here; when the
        // records come with the stream, site 5 drops them as it passes)
        if (! ps.isGenotyping)
        {
            while ((! ps.somaticChunks.empty()) && ps.somaticChunks.front().end <= pos) ps.somaticChunks.pop_front();
        }
        if (pos < ps.nextFinal[0]) return;
        if (ps.somaticPending.active)
        {
            AccumTimer hookTimer(s.tPileupHook);
            pileup_complete_somatic_push(pp); // (the window in flight is the one that covers it)
            if (pos < ps.nextFinal[0]) return;
        }
        if (! ps.isFlushing) throw blt_exception("strelka_amd adapter: the POST_ALIGN stage reached a position whose pileup is not final");
        AccumTimer hookTimer(s.tPileupHook);
        pileup_somatic_window(pp, 0, 0, true);
        pileup_complete_somatic_push(pp);
        return;
    }
    const unsigned sampleCount(Access::sampleCount(pp));
    if (! ps.isGenotyping)
    {
        // (chunks kept for the EVS accumulators only: POST_ALIGN only moves forward)
        for (unsigned sampleIndex(0); sampleIndex < sampleCount && sampleIndex < ps.chunks.size(); ++sampleIndex)
        {
            std::deque<SiteChunk>& chunks(ps.chunks[sampleIndex]);
            while ((! chunks.empty()) && chunks.front().end <= pos) chunks.pop_front();
        }
    }
    for (unsigned sampleIndex(0); sampleIndex < sampleCount; ++sampleIndex)
    {
        if (pos < ps.nextFinal[sampleIndex]) continue;
        if (sampleIndex < ps.pending.size() && ps.pending[sampleIndex].active)
        {
            // the window in flight is the one that covers it
            AccumTimer hookTimer(s.tPileupHook);
            pileup_complete_push(pp, sampleIndex);
            if (pos < ps.nextFinal[sampleIndex]) continue;
        }
        if (! ps.isFlushing) throw blt_exception("strelka_amd adapter: the POST_ALIGN stage reached a position whose pileup is not final");
        AccumTimer hookTimer(s.tPileupHook);
        // nothing new to pile up: an empty window that finalises everything
        pileup_sample_window(pp, sampleIndex, 0, 0, true);
        pileup_complete_push(pp, sampleIndex);
    }
}

void germline_fill_scoring_metrics(const unsigned sampleIndex, const pos_t pos, const snp_pos_info& pi)
{
    PileupState& ps(state().pileup);
    if (! (ps.enabled && ps.isGermlineMetrics) || sampleIndex >= ps.chunks.size()) return;
    for (SiteChunk& c : ps.chunks[sampleIndex])
    {
        if (pos < c.begin || pos >= c.end) continue;
        const size_t k(static_cast<size_t>(pos - c.begin));
        if (c.isMetricsFilled[k]) return;
        c.isMetricsFilled[k] = 1;
        if (c.isEvsGone) throw blt_exception("strelka_amd adapter: a position is scored after the POST_ALIGN stage has been through its window");
        const int64_t o(c.evsOff[k]);
        const size_t n(static_cast<size_t>(c.evsOff[k + 1] - o));
        if (n != pi.mapqTracker.count) throw blt_exception("strelka_amd adapter: the pileup of a scored position is not the stream's");
        snp_pos_info& acc(const_cast<snp_pos_info&>(pi));
        const char refBase(pi.get_ref_base());
        for (size_t i(0); i < n; ++i)
        {
            // pos_basecall_buffer::updateGermlineScoringMetrics (pos_basecall_buffer.cpp:43-70), in pileup order
            const uint64_t w(c.evsWordsPtr()[static_cast<size_t>(o) + i]);
            const uint8_t callId(static_cast<uint8_t>(w & 7u));
            const unsigned mapq(static_cast<unsigned>((w >> 3) & 0xffu)), qscore(static_cast<unsigned>((w >> 11) & 0x7fu));
            const unsigned cycle(static_cast<unsigned>((w >> 18) & 0x7ffu)), edge(static_cast<unsigned>((w >> 29) & 0x1fu));
            const bool isSubmapped((w >> 34) & 1u);
            const bool isReference(refBase == id_to_base(callId));
            acc.mq_ranksum.add_observation(isReference, mapq);
            if (! isSubmapped)
            {
                acc.baseq_ranksum.add_observation(isReference, qscore);
                acc.readPositionRankSum.add_observation(isReference, cycle);
                if (! isReference) acc.distanceFromReadEdge.addObservation(edge); // (already min(20, .))
            }
        }
        return;
    }
}

void somatic_fill_scoring_metrics(starling_pos_processor_base& pp, const pos_t pos)
{
    PileupState& ps(state().pileup);
    if (! (ps.enabled && ps.isSomaticMetrics)) return;
    for (SomaticChunk& c : ps.somaticChunks)
    {
        if (pos < c.begin || pos >= c.end) continue;
        const size_t k(static_cast<size_t>(pos - c.begin));
        if (c.isMetricsFilled[k]) return;
        c.isMetricsFilled[k] = 1;
        snp_pos_info& pi(Access::pileupRef(pp.sample(1).basecallBuffer, pos));
        const int64_t o(c.tumorTier1Off[k]);
        const size_t n(static_cast<size_t>(c.tumorTier1Off[k + 1] - o));
        if (n != pi.calls.size()) throw blt_exception("strelka_amd adapter: the tumor pileup of a written position is not the stream's");
        const char refBase(pi.get_ref_base());
        for (size_t i(0); i < n; ++i)
        {
            // updateSomaticScoringMetrics (:984-1000): tier1 reads of sample != 0, calls that pass the tier1 filter
            const base_call& bc(pi.calls[i]);
            if (bc.is_call_filter) continue;
            const uint32_t v(c.tumorReadPos[static_cast<size_t>(o) + i]);
            const uint16_t readPos(static_cast<uint16_t>(v & 0xffffu)), readLength(static_cast<uint16_t>(v >> 16));
            const bool isReference(refBase == id_to_base(bc.base_id));
            pi.readPositionRankSum.add_observation(isReference, readPos); // update_read_pos_ranksum, pos_basecall_buffer.cpp:75-85
            if (! isReference) pi.altAlleleReadPositionInfo.push_back({readPos, readLength}); // insert_alt_read_pos, .hh:84-95
        }
        return;
    }
}

void on_flush_begin(starling_pos_processor_base& /*pp*/)
{
    state().pileup.isFlushing = true;
}

void on_flush_end(starling_pos_processor_base& /*pp*/)
{
    state().pileup.isFlushing = false;
}

void pileup_note_read(const unsigned sampleIndex, const pos_t bufferPos)
{
    PileupState& ps(state().pileup);
    if (sampleIndex < ps.maxBufferPos.size()) ps.maxBufferPos[sampleIndex] = std::max(ps.maxBufferPos[sampleIndex], bufferPos);
}

}
