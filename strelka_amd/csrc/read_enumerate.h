// read_enumerate.h -- internal interface between the host stages of the realignment job (host/read_realign.cpp) and the device
// enumeration pipeline (csrc/read_enumerate.hip): sk_realign_options.enumeration == 2.
//
// Only what the reference's getCandidateAlignments hands to candidate_alignment_search crosses PCIe on the way in (PRead: start
// alignment, indel status map, indel order, observed indels) plus the read bases/qualities, the indel table and the reference
// segment; on the way out the candidate alignments of every read in std::set<CandidateAlignment> order and their scores.
#pragma once

#include "realign_core.h"
#include "stage3_core.h"

struct SkEnumInput
{
    const skcore::PIndel* tab;
    int32_t n_tab;
    const char* ins_pool; // insert sequences (ACGTN characters), PIndel.ins_off / ins_len index it
    int64_t ins_pool_len;
    const uint32_t* max_toggle;
    int32_t n_max_toggle;
    int32_t sample_count, max_read_indel_toggle, is_haplotyping_enabled, max_indel_size;
    double max_candidate_indel_density;
    const char* ref; // the job's reference segment
    int32_t ref_offset, ref_len;
    const skcore::PRead* reads; // reads to enumerate, in job order
    int32_t n_reads;
    const int64_t* read_off; // [n_reads+1] the full (unclipped) reads, as scored
    const uint8_t* read_code;
    const uint8_t* read_qual;
    int32_t max_read_len;
    int32_t want_scores; // 0 = enumeration only (sk_realign_job_get_batch flattens on the host)
    // stage 3 on the device as well (needs want_scores): per-indel error rates and given order, per-read mapping level
    int32_t want_stage3;
    const double* r2i;
    const double* i2r;
    const int32_t* orig;
    const int32_t* map_level;
    sk3::Opt stage3_opt;
};

struct SkEnumOutput // host arrays owned by the pipeline, valid until its next run
{
    const int32_t* status;     // [n_reads] skcore::ST_*; anything but ST_OK: the read has no entries below, enumerate it on the host
    const uint8_t* warn;       // [n_reads] bit 0: origin warning, bit 1: toggle-depth warning
    const int32_t* cal_off;    // [n_reads+1]
    const skcore::PCal* cals;  // [cal_off[n_reads]] each read's candidate alignments in set order; null with want_stage3: they stay on
                               // the device (sk_enum_device_fetch_cals) -- 288 bytes per alignment that the host rarely needs
    uint64_t generation;       // of this run, for sk_enum_device_fetch_cals
    const double* scores;      // [cal_off[n_reads]]; null without want_scores, and null when the job ran as one fixed sequence: the scores stay on
                               // the device (sk_enum_device_fetch_scores) -- the host reads them only for a read stage 3 turned down
    const uint8_t* consulted;  // [n_tab] candidate status consulted by the search, the flattening or stage 3
    int64_t ref_reads_outside; // reference reads of the flattening that fell outside [ref_offset, ref_offset + ref_len) (read as 'N')
    const sk3::Out* stage3;    // [n_reads] or null; stage3[r].status != S3_OK (or status[r] != ST_OK): stage 3 of the read is the host's
};

extern "C" int sk_enum_device_run(const SkEnumInput* in, SkEnumOutput* out);

/** candidate alignments [first, first + count) of run `generation`, from the device's buffers; 1 when another run has replaced them */
extern "C" int sk_enum_device_fetch_cals(uint64_t generation, int32_t first, int32_t count, skcore::PCal* out);

/** scores [first, first + count) of run `generation`; 1 when another run has replaced them */
extern "C" int sk_enum_device_fetch_scores(uint64_t generation, int32_t first, int32_t count, double* out);
/** jobs of this process: run as one fixed sequence with one wait, of those run again the staged way, run the staged way from the start */
extern "C" void sk_enum_device_job_counts(int64_t* one_wait, int64_t* one_wait_redone, int64_t* staged);

/** whether enumeration == 2 can run (it is the default where it can): 1 in the GPU library once sk_init has succeeded, 0 before
 *  that and in the CPU double of the ABI */
extern "C" int sk_enum_device_available(void);
// bench: F1-F3 + scoring over the last run's resident candidate alignments, `reps` times; elapsed milliseconds (stream events)
extern "C" int sk_enum_device_rescore(int32_t reps, float* out_ms, int32_t* out_n_reads, int32_t* out_n_cals, int64_t* out_cells);
