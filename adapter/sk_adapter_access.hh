// sk_adapter_access.hh -- the friend through which the adapter reaches the private state of the reference's position
// processors, and the per-process adapter state.  Included by the adapter's own translation units only.
#pragma once

#include <chrono>

#include "sk_adapter.hh"
#include "strelka_amd.h"

#include "blt_util/blt_exception.hh"
#include "starling_common/starling_pos_processor_base.hh"

#include <climits>
#include <deque>
#include <functional>
#include <queue>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <vector>

struct read_segment;

namespace sk_adapter
{

struct Access
{
    typedef starling_pos_processor_base base_t;
    static const starling_base_options& opt(const base_t& pp) { return pp._opt; }
    static const starling_base_deriv_options& dopt(const base_t& pp) { return pp._dopt; }
    static const reference_contig_segment& ref(const base_t& pp) { return pp._ref; }
    static unsigned sampleCount(const base_t& pp) { return pp.getSampleCount(); }
    static IndelBuffer& indelBuffer(base_t& pp) { return pp.getIndelBuffer(); }
    static const PileupCleaner& pileupCleaner(const base_t& pp) { return pp._pileupCleaner; }
    static unsigned largestReadSize(const base_t& pp) { return pp.get_largest_read_size(); }
    static unsigned largestTotalIndelRefSpanPerRead(const base_t& pp) { return pp.get_largest_total_indel_ref_span_per_read(); }
    static bool isPosReportable(const base_t& pp, const pos_t pos) { return pp.is_pos_reportable(pos); }
    static unsigned ploidy(const base_t& pp, const pos_t pos, const unsigned sampleIndex) { return pp.get_ploidy(pos, sampleIndex); }
    static bool isForcedOutputPos(const base_t& pp, const pos_t pos) { return pp.is_forced_output_pos(pos); }
    static bool isAnyForcedOutputPos(const base_t& pp, const pos_t begin, const pos_t end)
    {
        const auto it(pp._forced_output_pos.lower_bound(begin));
        return it != pp._forced_output_pos.end() && *it < end;
    }
    static void clearActiveRegionReadBuffer(base_t& pp, const pos_t pos) { pp._getActiveRegionDetector().clearReadBuffer(pos); }
    static const CandidateSnvBuffer& candidateSnvBuffer(const base_t& pp) { return pp._candidateSnvBuffer; }
    static bool hasPloidyRegions(const base_t& pp, const unsigned sampleIndex) { return ! pp.sample(sampleIndex).ploidyRegions.empty(); }
    static const known_pos_range2& reportRange(const base_t& pp) { return pp._reportRange; }
    static ActiveRegionId activeRegionId(const base_t& pp, const pos_t pos) { return pp.getActiveRegionDetector().getActiveRegionId(pos); }
    /// pos_basecall_buffer::_pdata.getRef(pos): the position's pileup record, created (with its reference base) if absent
    static snp_pos_info& pileupRef(pos_basecall_buffer& buffer, const pos_t pos) { return buffer._pdata.getRef(pos); }
};

/// C-ABI status -> the reference's exception type (the context chain of starling_pos_processor_base.cpp:755-760 prints it)
inline void check(const int rc, const char* what)
{
    if (rc == 0) return;
    throw blt_exception((std::string("strelka_amd: ") + what + ": " + sk_last_error()).c_str());
}

/// Shadow of the stage geometry the UNHOOKED reference would have at each moment (see sk_adapter.hh).
struct GeometryShadow
{
    struct Params
    {
        pos_t upto;           ///< last READ_BUFFER position these values apply to
        pos_t rangeMinOffset; ///< get_realignment_range (starling_pos_processor_base.cpp:705-727)
        pos_t rangeMaxOffset;
        pos_t validThreshold; ///< a realignment is valid iff realignment.pos > validThreshold (:764, stage_manager.cpp:190-198)
    };

    void reset(const unsigned sampleCount);
    void onSetHeadPos(const pos_t pos, const unsigned readBufferShift, const unsigned indelSpan);
    /// values the reference's align_pos(pos) would have used
    Params query(const pos_t pos) const;

    bool isFirstPosSet = false;
    pos_t maxPos = 0;          ///< stage_manager::_max_pos
    pos_t lastReadBufferPos = 0;
    bool isAnyReadBufferPos = false;
    unsigned curReadBufferShift = 0, curIndelSpan = 0;
    std::deque<Params> segments;
    pos_t activeRegionClearedTo = 0; ///< ActiveRegionDetector::clearReadBuffer has been called up to here (undeferred)
    bool isAnyActiveRegionCleared = false;
    pos_t clearedToPos = 0;    ///< reads at buffer positions <= this have left the reference's read buffer (CLEAR_READ_BUFFER)
    bool isAnyCleared = false;
    /// per sample: buffer positions of the reads the reference would still hold (a min-heap: inserted per read, dropped from the low end
    /// as the reference's CLEAR_READ_BUFFER stage would pass them, counted -- no node per read)
    typedef std::priority_queue<pos_t, std::vector<pos_t>, std::greater<pos_t>> PosHeap;
    std::vector<PosHeap> bufferedReadPos;
};

/// germline SNV genotypes of one stage window, computed ahead of the per-position calls of process_pos_snp_digt
struct SiteCache
{
    pos_t begin = 0, end = 0;      ///< positions [begin, end) are cached
    std::vector<uint8_t> isValid;  ///< [(pos-begin)*sampleCount + sample]
    std::vector<uint8_t> ploidy;   ///< the ploidy the entry was computed with
    std::vector<uint32_t> callCount;
    std::vector<sk_digt_call> calls;
    void clear() { begin = end = 0; isValid.clear(); ploidy.clear(); callCount.clear(); calls.clear(); }
};

/// somatic SNV records of one stage window
struct SomaticSiteCache
{
    pos_t begin = 0, end = 0;
    std::vector<uint8_t> isValid, forced;
    std::vector<uint32_t> callCount; ///< [(pos-begin)*4 + {normal t1, tumor t1, normal t2, tumor t2}]
    std::vector<sk_somatic_snv_genotype> genotypes;
    void clear() { begin = end = 0; isValid.clear(); forced.clear(); callCount.clear(); genotypes.clear(); }
};

/// genotypes of a run of positions one pileup push finalised (site 9 chained into sites 2+3)
struct SiteChunk
{
    pos_t begin = 0, end = 0;
    std::vector<sk_digt_call> calls;
    std::vector<uint32_t> cleanCount; ///< calls of the cleaned column each genotype was computed from
    std::vector<uint8_t> ploidy;      ///< ... and the ploidy
    std::vector<uint32_t> rawCount;   ///< calls of the raw tier1 column the window wrote into the reference's buffer
    std::vector<sk_gvcf_site_summary> summary; ///< what the gVCF writer's block logic reads of each position (site 10)
    std::vector<sk_gvcf_run> runs;    ///< ... and the non-variant block that would start at each plain position
    // germline EVS: the per-call arguments of updateGermlineScoringMetrics, kept until POST_ALIGN has passed the chunk
    std::vector<int64_t> evsOff;      ///< [n+1]
    std::vector<uint64_t> evsWords;
    /// The words where the push left them -- one of the stream's output blocks, valid through its next SK_PILEUP_WINDOW_LIFETIME pushes
    /// (strelka_amd.h) -- instead of a copy: 2.6 MB a window (0.3 ms), read at one position in a few hundred (updateSiteSampleInfo asks at
    /// forced and variant sites only, starling_pos_processor.cpp:232-241).  By the time the block comes round again POST_ALIGN has as a
    /// rule passed the chunk's positions and nobody will ask; a chunk it has not passed -- the stage machine caught up over a coverage
    /// gap, pushes in quick succession -- takes its copy then (pileup_push, evsWordsCopied counts them).
    const uint64_t* evsLive = nullptr;
    unsigned long evsPush = 0;        ///< the sample stream's push that brought the chunk
    bool isEvsGone = false;           ///< the words were left in the stream's block when it came round again (nobody asks: germline_fill_scoring_metrics checks)
    std::vector<uint8_t> isMetricsFilled;
    const uint64_t* evsWordsPtr() const { return evsLive != nullptr ? evsLive : evsWords.data(); }
    void materializeEvsWords()
    {
        if (evsLive == nullptr) return;
        evsWords.assign(evsLive, evsLive + (evsOff.empty() ? 0 : evsOff.back()));
        evsLive = nullptr;
    }
};

/// somatic SNV records of a run of positions one push of the two samples' pileups finalised (site 9 chained into site 5)
struct SomaticChunk
{
    pos_t begin = 0, end = 0;
    std::vector<sk_somatic_snv_genotype> genotypes; ///< empty when the stream does not genotype
    std::vector<uint32_t> count[4];                 ///< cleaned column sizes: normal t1, tumor t1, normal t2, tumor t2
    std::vector<uint32_t> rawCount[4];              ///< raw column sizes: normal tier1, tumor tier1, normal tier2, tumor tier2
    std::vector<uint8_t> forced;
    // what updateSomaticScoringMetrics would have accumulated for the tumor sample, kept until a record is written
    std::vector<int64_t> tumorTier1Off;             ///< [n+1]
    std::vector<uint32_t> tumorReadPos;             ///< parallel to the tumor's tier1 calls: read_pos | read_size << 16
    std::vector<uint8_t> isMetricsFilled;
};

/// a window whose push has been begun and not finished (sk_pileup_stream_push_begin / _finish): what its completion still needs
struct PendingPush
{
    bool active = false;
    bool isFinal = false;
    int32_t finalTo = 0;
    pos_t ploidyBegin = 0;
    bool hasPloidy = false;
    std::vector<uint8_t> ploidy;
    pos_t forcedBegin = 0;            ///< (the somatic stream's window: is_forced_output_pos of the positions it can finalise)
    std::vector<uint8_t> forced;
};

/// site 9: one pileup stream per sample (sk_adapter_pileup.cpp); the somatic caller's two samples share one
struct PileupState
{
    /// The device works on a sample's window while the stage machine goes through the positions POST_ALIGN still has before it:
    /// the push is finished when POST_ALIGN reaches the first position no finished window covers (pileup_before_variants), before the
    /// sample's next push, and at a region's end.  $STRELKA_AMD_PUSH_ASYNC=0: begun and finished in one call.
    std::vector<PendingPush> pending;
    PendingPush somaticPending;                ///< the two samples of the somatic caller are one push
    bool decided = false, enabled = false, isGenotyping = false;
    bool isSomatic = false, isSomaticMetrics = false, isGermlineMetrics = false;
    std::vector<sk_pileup_stream*> streams;
    sk_somatic_pileup_stream* somaticStream = nullptr;
    std::deque<SomaticChunk> somaticChunks;
    std::vector<std::deque<SiteChunk>> chunks; ///< per sample, ascending
    std::vector<uint8_t> isRegionOpen;         ///< per sample: sk_pileup_stream_begin_region done for the current region
    std::vector<pos_t> nextFinal;              ///< per sample: positions below are final (filled into the reference's buffers)
    std::vector<pos_t> pendingEnd;             ///< per sample: one past the highest position any pushed read reaches
    std::vector<pos_t> maxBufferPos;           ///< per sample: highest read-buffer position of any read inserted in this region
    bool isAnyPiled = false;
    pos_t piledTo = 0;                         ///< reads buffered at positions < piledTo have been pushed
    bool isFlushing = false;                   ///< inside starling_pos_processor_base::reset(): no more reads will arrive
    pos_t regionBegin = 0, regionEnd = 0;
    pos_t lastVariantsPos = INT_MIN;           ///< the last position POST_ALIGN has been through (pileup_before_variants)
    std::vector<unsigned long> pushCount;      ///< per sample: pushes of its stream so far
    unsigned long evsWordsCopied = 0, evsWordsLeft = 0; ///< chunks whose EVS words were copied after all / never were (SiteChunk::evsLive)
};

/// a read segment of the current stage window (collected by align_pos, used by the realignment job and the pileup push)
struct WindowSegment
{
    read_segment* rseg;
    pos_t bufferPos;
};

struct State
{
    GeometryShadow geometry;
    bool isAnyRealigned = false;
    pos_t realignedTo = 0;         ///< reads buffered at positions < realignedTo went through a realign job already
    SiteCache sites;
    SomaticSiteCache somaticSites;
    PileupState pileup;
    std::vector<std::vector<WindowSegment>> windowSegments; ///< per sample: the read segments buffered in [window begin, realignedTo)
    // counters reported at exit with $STRELKA_AMD_VERBOSE=1
    // wall seconds inside the hooks (whole hook) and inside the C-ABI calls they make; reported with STRELKA_AMD_VERBOSE=1
    double tRealignHook = 0, tRealignAbi = 0, tSiteHook = 0, tSiteAbi = 0, tPileupHook = 0, tPileupAbi = 0, tPileupGather = 0, tPileupAssign = 0, tPileupChunk = 0, tInit = 0, tIndelAbi = 0, tHaplotypeAbi = 0;
    unsigned long pileupBatches = 0, pileupReads = 0, pileupLoci = 0;
    unsigned long indelGroupsWide = 0; ///< allele groups with more alternate alleles than SK_MAX_ALT (sk_allele_group_genotype_lhoods_wide)
    unsigned long indelGroupsXWide = 0; ///< ... of those, with more than SK_MAX_ALT_WIDE (sk_allele_group_genotype_lhoods_xwide: runs of five to eight samples)
    unsigned long realignJobReads = 0; ///< reads that went into a realignment job (realignReads counts every read a window looked at)
    unsigned long realignDeviceEnumerated = 0, realignHostEnumerated = 0; // reads whose candidate alignments the device / the host listed
    unsigned long realignRefWindowMisses = 0; // jobs run a second time with the whole contig segment as their reference
    unsigned long realignHostJobs = 0; // jobs whose search ran as the host statement (below the device threshold, or $SK_ENUMERATION)
    unsigned long realignBatches = 0, realignReads = 0, siteBatches = 0, siteLoci = 0, siteRecomputed = 0, siteRecomputeCalls = 0, indelGroups = 0, haplotypes = 0, haplotypeBatches = 0;
};
/// the adapter's state (one per process); the hooks ask for it at every position and every read, so after the first call it is a load
State& make_state();
extern State* g_state;
inline State& state()
{
    return (g_state != nullptr) ? *g_state : make_state();
}

// site 9 internals (sk_adapter_pileup.cpp)
bool pileup_enabled(starling_pos_processor_base& pp);
bool pileup_genotypes_with_stream(const starling_base_options& opt);
void pileup_reset_region(starling_pos_processor_base& pp);
void pileup_before_variants(starling_pos_processor_base& pp, const pos_t pos);
void pileup_note_read(const unsigned sampleIndex, const pos_t bufferPos);
/// (sk_adapter_somatic.cpp) the somatic SNV model's options for the chained stream; false: the run does not call somatic SNVs
bool somatic_stream_options(const starling_pos_processor_base& pp, sk_somatic_snv_options& so, bool& isComputeNonSomatic);

struct AccumTimer // adds its lifetime to `acc`
{
    double& acc;
    std::chrono::steady_clock::time_point t0;
    explicit AccumTimer(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~AccumTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

}
