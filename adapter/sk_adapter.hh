// sk_adapter.hh -- the hook functions the hooked copies of the reference's position-processor sources call
// (adapter/apply_hooks.py inserts the calls; the hooked copies live under oracle/_ref/adapter_src/, never in the repo).
//
// This is the compiled form of INTEGRATION.md: C++ written against the REFERENCE's headers (it only builds where
// /root/reference is present) that re-routes the hot-path call sites of starling_pos_processor_base / starling_pos_processor /
// strelka_pos_processor through the C-ABI of include/strelka_amd.h, with stage-window batching.  It is the integration
// layer a Strelka2 maintainer would add, and the test vehicle for the end-to-end byte-identity tests
// (tests/test_e2e_adapter.py); the product is libstrelka_amd.so.
//
// Batching.  The reference runs every call site once per position as its stage manager
// advances (L/blt_util/stage_manager.cpp:206-243).  The adapter pushes the READ_BUFFER stage `read_buffer_defer()`
// positions and the POST_ALIGN stage a further `post_align_defer()` positions behind their reference distances
// (L/starling_common/starling_pos_processor_base.cpp:141-224), so that when a stage reaches position P every input of
// positions [P, P+W) is already final: it then runs the kernel-backed work of the whole window in one C-ABI call and serves
// the per-position calls of the reference's unchanged control flow from the cached results.  Everything the reference
// derives from the stage geometry itself (realignment range, realignment-validity test, read-buffer occupancy) is
// reproduced from a shadow of the ORIGINAL geometry (GeometryShadow, sk_adapter_common.cpp).
#pragma once

#include "sk_adapter_fwd.hh"

#include "blt_util/blt_types.hh"

#include <string>
#include <vector>

struct starling_base_options;
struct starling_pos_processor_base;
struct starling_pos_processor;
struct strelka_pos_processor;
struct diploid_genotype;
struct starling_read;
struct CleanedPileup;
struct snp_pos_info;
struct somatic_snv_genotype_grid;
struct somatic_indel_call;
struct strelka_options;
struct starling_sample_options;
struct IndelData;
struct reference_contig_segment;
struct IndelKey;
struct alignment;
class ActiveRegionReadBuffer;
struct bam_seq_base;

struct sk_pileup_stream; // (include/strelka_amd.h)

namespace sk_adapter
{

/// extra distance of the READ_BUFFER stage behind HEAD / of the POST_ALIGN stage behind READ_BUFFER
/// ($STRELKA_AMD_READ_WINDOW / $STRELKA_AMD_SITE_WINDOW, default 8192 positions / 1024 with the pileup stream, else 4096)
unsigned read_buffer_defer();
unsigned post_align_defer();
/// the same, decided on first use from the options: a run whose genotypes come with the pileup stream (site 9) has nothing to batch at
/// POST_ALIGN and keeps the reference's distance (0) -- every deferred position is a position's worth of pileup and read buffer kept
/// alive, and the host code runs out of cache long before the device runs out of work
unsigned post_align_defer(const starling_base_options& opt);

/// select the device ($STRELKA_AMD_DEVICE, default 0) and sk_init(); throws blt_exception on failure.  Idempotent.
void init();

// ---- geometry shadow (starling_pos_processor_base.cpp: set_head_pos, insert_read, resetRegionBase, reset) ----
void on_reset_region(starling_pos_processor_base& pp);
void on_set_head_pos(starling_pos_processor_base& pp, const pos_t pos, const unsigned readBufferShift, const unsigned indelSpan);
unsigned buffered_read_count(const starling_pos_processor_base& pp, const unsigned sampleIndex, const unsigned actualCount);
void on_read_inserted(starling_pos_processor_base& pp, const unsigned sampleIndex, const starling_read& sread);

// ---- site 1: realignAndScoreRead at starling_pos_processor_base.cpp:752 (align_pos) ----
/// replaces the body of align_pos(pos); always returns true
bool align_pos(starling_pos_processor_base& pp, const pos_t pos);

// ---- sites 2+3: adjust_joint_eprob (PileupCleaner.cpp:73) + position_snp_call_pprob_digt (starling_pos_processor.cpp:265) ----
void before_process_pos_variants(starling_pos_processor_base& pp, const pos_t pos);
void site_diploid_genotype(starling_pos_processor& pp, const pos_t pos, const unsigned sampleIndex, const unsigned ploidy,
                           diploid_genotype& dgt);
/// the four zero-depth genotypes of the constructor (starling_pos_processor_base.cpp:259-274)
void empty_site_genotype(const starling_pos_processor_base& pp, const unsigned refBaseId, diploid_genotype& dgt);

// ---- site 9: pileup_pos_reads at starling_pos_processor_base.cpp:813 (pileup_read_segment :1127-1421), chained into sites 2+3 ----
/// replaces the body of pileup_pos_reads(pos); false: the reference's own pileup runs (somatic / EVS-metric runs, STRELKA_AMD_PILEUP=0)
bool pileup_pos_reads(starling_pos_processor_base& pp, const pos_t pos);
/// starling_pos_processor_base::reset() (:347-356) is about to flush / has flushed its stages: no more reads in this region
void on_flush_begin(starling_pos_processor_base& pp);
void on_flush_end(starling_pos_processor_base& pp);

// ---- site 5: position_somatic_snv_call at strelka_pos_processor.cpp:213-219 ----
void somatic_window(starling_pos_processor_base& pp, const pos_t pos);
/// the position's record; normalCpi / tumorCpi: the processor's cleaned pileups {tier1, tier2}; isCleanDeferred: they have not been
/// built for this position (somatic_defer_clean) -- cleared when this call had to build them
void somatic_snv_genotype(starling_pos_processor_base& pp, const pos_t pos, CleanedPileup* const* normalCpi, CleanedPileup* const* tumorCpi,
                          bool& isCleanDeferred, const bool isComputeNonSomatic, somatic_snv_genotype_grid& sgt);
/// true: process_pos_snp_somatic skips its four CleanPileup calls for this position (the stream's records serve site 5)
bool somatic_defer_clean(starling_pos_processor_base& pp, const pos_t pos);
/// the skipped CleanPileup calls (strelka_pos_processor.cpp:180-186), now
void somatic_clean_now(starling_pos_processor_base& pp, const pos_t pos, CleanedPileup* const* normalCpi, CleanedPileup* const* tumorCpi);
/// process_pos_sample_stats (starling_pos_processor_base.cpp:1473-1495): used / unused basecall counts of the position from the
/// stream's column sizes; false: the caller cleans the pileup itself (always, outside the somatic stream)
bool sample_stats_counts(starling_pos_processor_base& pp, const pos_t pos, const unsigned sampleIndex, unsigned& used, unsigned& unused);

// ---- site 10 (sk_adapter_gvcf.cpp): the gVCF writer's non-variant blocks fed from the pileup stream's window
/// the germline half of sample_stats_counts: the counts of a position the window calls a plain site (its pileup is then not cleaned)
bool germline_sample_stats_counts(starling_pos_processor_base& pp, const pos_t pos, const unsigned sampleIndex, unsigned& used, unsigned& unused);
/// process_pos_snp (L/applications/starling/starling_pos_processor.cpp:143-197), first thing: true = the position was a plain
/// homozygous-reference site and has gone into the writer's open block (gvcf_writer::skip_to_pos + add_site_internal on a kept
/// locus); false = the reference's process_pos_snp runs (its cleaned pileup is made first if process_pos_sample_stats left it out)
bool gvcf_plain_site(starling_pos_processor& pp, const pos_t pos);
/// a new region begins (starling_pos_processor_base::resetRegionBase): no block installed by site 10 reaches into it
void gvcf_reset_region();
/// at the start of a region: the options that decide a plain site's filters and block membership (gvcf_options, the chromosome's depth
/// ceiling) to the sample's pileup stream, which then returns the block that would start at every plain site (sk_gvcf_run)
void gvcf_configure_stream(starling_pos_processor_base& pp, const unsigned sampleIndex, ::sk_pileup_stream* stream);

/// the tumor sample's readPositionRankSum / altAlleleReadPositionInfo of `pos` (updateSomaticScoringMetrics,
/// starling_pos_processor_base.cpp:984-1000), rebuilt from the pileup stream's window before the position's record is written
/// (strelka_pos_processor.cpp:255); nothing to do when the reference's own pileup ran
void somatic_fill_scoring_metrics(starling_pos_processor_base& pp, const pos_t pos);

/// the germline EVS accumulators of one sample's pileup at `pos` (mq_ranksum, baseq_ranksum, readPositionRankSum, distanceFromReadEdge:
/// updateGermlineScoringMetrics, starling_pos_processor_base.cpp:1346-1357), rebuilt from the pileup stream's window just before
/// updateSiteSampleInfo reads them (starling_pos_processor.cpp:235-246); nothing to do when the reference's own pileup ran
void germline_fill_scoring_metrics(const unsigned sampleIndex, const pos_t pos, const snp_pos_info& pi);

// ---- site 6: get_somatic_indel at strelka_pos_processor.cpp:343-349 ----
void somatic_indel(const strelka_options& opt, const starling_sample_options& normalOpt, const starling_sample_options& tumorOpt,
                   const IndelKey& indelKey, const IndelData& indelData, const unsigned normalSampleIndex,
                   const unsigned tumorSampleIndex, const bool isUseAltIndel, somatic_indel_call& sindel);

/// ActiveRegionDetector::clearReadBuffer at the position the UNDEFERRED READ_BUFFER stage would be at while HEAD is at `headStagePos`
void clear_active_region_read_buffer_undeferred(starling_pos_processor_base& pp, const pos_t headStagePos, const unsigned readBufferShift,
                                                const pos_t minPos);

// ---- site 7: ActiveRegionProcessor::discoverIndelsAndMismatches (ActiveRegionProcessor.cpp:572-705) ----
/// `selectedHaplotypes` / `selectedHaplotypeIndex`: the region's selected haplotypes and the one asked for -- the first call of a
/// region aligns all of its alternate haplotypes in ONE sk_global_align batch, the later calls of processSelectedHaplotypes' loop
/// (:528-551) take their path from it
bool discover_indels_and_mismatches(const std::vector<std::string>& selectedHaplotypes, const unsigned selectedHaplotypeIndex,
                                    const std::string& refSegment, const reference_contig_segment& ref, const pos_t regionBegin, const pos_t regionEnd,
                                    const pos_t prevActiveRegionEnd, const unsigned maxIndelSize,
                                    std::vector<IndelKey>& discovered, int& numIndels);

// ---- site 8: the reads of a region, bam_streamer::resetRegion / next (L/htsapi/bam_streamer.cpp:211-287) ----
/// index lookup, block inflation and record selection for [begin, end) of reference `tid` through the feed entry points of the C-ABI;
/// false: this streamer keeps the reference's own iterator (not a BAM with a .bai, or STRELKA_AMD_FEED=0)
bool feed_reset_region(const void* streamer, const char* name, const int tid, const int begin, const int end);
bool feed_active(const void* streamer);
/// the next record of the region into the streamer's bam1_t, as sam_itr_next returns it (>= 0, -1 at the end, < -1 on a bad record)
int feed_next(const void* streamer, void* bam1);
void feed_drop(const void* streamer);
/// normalizeAlignment at starling_pos_processor_util.cpp:432 for the streamer's current record, from one batched sk_normalize_alignments
/// call per region; false: the caller runs the reference's function
bool feed_normalize_current(const void* streamer, const reference_contig_segment& ref, alignment& al);

}
