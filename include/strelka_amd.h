/*
 * strelka_amd.h -- C-ABI of the MI355X-native Strelka2 hot path.
 *
 * Every entry point replaces one call site of the reference (Illumina/strelka v2.9.x); the reference has no FFI
 * seam of its own, so the seam is introduced directly below the `starling_pos_processor_base`
 * class surface, at the call sites cited on each function.  `L/` = `src/c++/lib/` of the reference tree.
 *
 * Conventions
 *  - plain C: POD structs, pointers and sizes only.  No C++/torch types cross this boundary.
 *  - every function returns 0 on success, non-zero on failure; `sk_last_error()` gives the message.  Nothing throws or
 *    aborts across the ABI (the reference throws `blt_exception`; the host adapter in INTEGRATION.md converts).
 *  - batches are flat SoA with CSR offsets for ragged dimensions (reads->candidate alignments->ops, loci->calls,
 *    indels->reads).  The caller owns every buffer; the library retains no pointer after a call returns.
 *  - `*_dev` variants take DEVICE pointers inside the same batch structs plus a `hipStream_t` (as `void*`) and only
 *    enqueue work; the plain variants take HOST pointers, stage through library-owned device buffers, and block.
 *  - one host thread per process (as in the reference); many processes may share one GPU.
 *
 * The product path has no CPU fallback: if the HIP runtime or a gfx950 device is missing `sk_init` fails.
 */
#ifndef STRELKA_AMD_H
#define STRELKA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SK_VERSION 100

/* ------------------------------------------------------------------------------------------------------------------
 * lifecycle
 * ---------------------------------------------------------------------------------------------------------------- */

/** Select `device`, build the host-side q-score/likelihood tables with the host libm (exactly the expressions of
 *  L/blt_util/qscore_cache.cpp:34-50 and the memoised tables of
 *  L/applications/strelka/position_somatic_snv_strand_grid_lhood_cached.cpp:41-234) and upload them.
 *  Idempotent for the same device. */
/** number of gfx950 devices this process can see (0 when there is none); callable before sk_init.  A launcher that hands
 *  segment process i the device `i mod N` may name more devices than a node has: the adapter takes the index modulo this. */
int sk_device_count(void);
/** Page-locked host memory for buffers that cross this ABI again and again (a region's compressed and inflated bytes, decoded
 *  reads): the device copies straight from / into it, whereas a copy from ordinary memory has the driver pin and unpin the pages
 *  every time -- system time that sixteen caller processes sharing a GPU spend contending for the same locks.  Any host pointer is
 *  accepted everywhere; these are for speed.  NULL on failure (sk_last_error).  Needs sk_init. */
void* sk_host_alloc(size_t bytes);
void sk_host_free(void* p);
int sk_init(int device);
/** sk_init, but fails when the host C library is not the one the kernels restate (sk_libm_restated() would be 0): for
 *  callers that must guarantee results bit-identical to the reference (the adapter, smoke(), bench.py). */
int sk_init_strict(int device);
void sk_shutdown(void);
const char* sk_last_error(void);
int sk_version(void);
/** 1 when sk_init succeeded on a gfx950 device. */
int sk_is_initialized(void);
/** How this process waits for its device: 1 = blocking (it sleeps: the default, many caller processes share a GPU and a CPU quota),
 *  0 = spinning ($STRELKA_AMD_SPIN_WAIT, or the runtime refused the flag), -1 = not initialised.  The flag is per device: it is set
 *  after the device is chosen (a farm process on device 3 gets it as one on device 0 does). */
int sk_sync_mode(void);
/** 1 when sk_init found the host libm's powf/logf to be the routines the kernels restate (glibc >= 2.28, see
 *  strelka_amd/csrc/libm_flt32.h): the dependent error probabilities and germline likelihoods are then bit-identical to
 *  the reference's; 0: the device library's pow/log stand in (agreement to 1e-5 relative). */
int sk_libm_restated(void);

/* ---- many caller processes per GPU: the broker ------------------------------------------------------------------------------
 * SURVEY.md 8(b) "Threading": the ABI is called by one thread per process and by MANY processes per GPU -- the workflow starts one
 * caller process per core (PY/strelkaSharedOptions.py:153-161, PY/strelkaGermlineWorkflow.py:81-150).  The device runs eight
 * processes' compute work side by side; a ninth is time-sliced.  With $STRELKA_AMD_BROKER=1 in a caller process's environment this
 * library creates no GPU context there: every allocation, copy, launch and wait behind the SAME entry points is carried (records in a
 * shared-memory ring; page-locked buffers mapped at one address in both processes) to `sk_broker`, the one process per device that
 * holds its context, started on demand by the first client and gone 20 s after the last
 * (strelka_amd/csrc/sk_rt.h; $STRELKA_AMD_BROKER_IDLE_S, _QUEUES, _LOG, _SOCKET, _NO_SPAWN).  Host stages (a realignment job's gate and
 * packing, a pileup stream's carry bookkeeping) stay in the caller.  Not available to a client: the `*_dev` entry points on a stream
 * of the caller's, and the event-timed diagnostics. */
/** 1 when this process is a broker client ($STRELKA_AMD_BROKER, sk_broker_enable). */
int sk_broker_client(void);
/** Turn the broker mode of this process on or off without the environment variable; before sk_init (fails afterwards).  The adapter
 *  turns it on by default -- a workflow's caller processes are one per core -- and leaves it off with $STRELKA_AMD_BROKER=0. */
int sk_broker_enable(int on);
/** The server's main loop (what `sk_broker --device D [--socket NAME] [--idle-exit S]` runs): serves `device` on the abstract unix
 *  socket `socket_name` (NULL / "": the default name -- user, library build, device) until it has had no client for `idle_seconds`.
 *  Returns 0 after an idle exit, 2 when the device is unusable, 3 when another broker already serves the name. */
int sk_broker_serve(int device, const char* socket_name, int idle_seconds);
/** Round trip of every kind of call a client makes (device memory, copies from / to pageable and page-locked memory, a launch, the
 *  wait) over n 32-bit words; 0 when the results are right.  Works in a process with its own context too.  Needs sk_init. */
int sk_broker_selftest(int n, uint32_t mul);

/** The `*_dev` entry points only enqueue work and cannot validate device-resident input the way the host-buffer entry
 *  points do.  Where the reference would throw on such input (a basecall quality above 70, qscore_cache.cpp:53-75), the
 *  kernel raises a sticky device flag instead of returning a plausible number; this call synchronises the device, returns
 *  non-zero with the reference's message in sk_last_error() if a flag is up, and clears it. */
int sk_check_device_errors(void);

/** Test hook: which statement of the fused germline site kernel sk_site_digt_call_fused[_dev] launches (0 = round 1's, 1 = the second one:
 *  csrc/germline_fused.hip); < 0 = $SK_G3_VARIANT or the
 *  default.  Every variant writes the same records (tests/test_gpu_parity.py runs them all against the oracle) WHEN
 *  sk_libm_restated() == 1: variant 1 takes the ranked-call terms of groups without a neighbouring mismatch from a table built with
 *  the host libm (csrc/germline_common.h, v0r0), its pending terms and all of variant 0's from the device routines; on a host whose
 *  libm is not the restated one (STRELKA_AMD_ALLOW_INEXACT_LIBM) the two agree to the documented 1e-5, not bit for bit. */
int sk_debug_set_g3_variant(int variant);
/** Test hook: on != 0 makes every kernel take the path it takes on a host whose libm is NOT the restated one (the device
 *  math library's double routines stand in; sk_libm_restated() then reports 0); on == 0 restores the sk_init outcome.
 *  Lets the test suite measure the documented 1e-5 bar of that path on a box where the exact path is available. */
int sk_debug_force_device_libm(int on);

/** Host copies of the q-score tables the kernels use (71 entries each, Q0..Q70; L/blt_util/qscore_cache.hh:66-68). */
int sk_get_qscore_tables(double* q2p, double* q2lncompe, double* q2lne);

/* ------------------------------------------------------------------------------------------------------------------
 * Hot path A: candidate-alignment scoring
 *   replaces the loop at L/starling_common/starling_read_align.cpp:1568-1569 calling
 *   scoreCandidateAlignment (L/starling_common/starling_read_align_score.cpp:261-499).
 * ---------------------------------------------------------------------------------------------------------------- */

/** BAM 4-bit base codes, one per byte (L/htsapi/bam_seq.hh:30-50). */
enum { SK_BAM_REF = 0, SK_BAM_A = 1, SK_BAM_C = 2, SK_BAM_G = 4, SK_BAM_T = 8, SK_BAM_ANY = 15 };

/** Scoring op kinds: a candidate alignment's CIGAR + indel keys flattened in path order by
 *  sk_flatten_candidate_alignment (host adapter).  Read offset advances by `length` for BASES/SOFT_CLIP ops. */
enum {
    SK_OP_BASES = 0,     /* score read[ro+i] against hap pool [src+i]: a MATCH segment against the reference window
                            (scoreMatchSegment :144-170) or an INSERT segment against the indel's insert sequence
                            (scoreInsertSegment :110-137) -- the two reference loops are identical                 */
    SK_OP_SOFT_CLIP = 1, /* lnp += length * ln(0.25)                                        (:453-454)          */
    SK_OP_NOBASE = 2     /* DELETE/SKIP/HARD_CLIP: no base term, only the optional penalty  (:419-460)          */
};
/** flags bit 0: add ln(1e-5) after this op (its indel is not a candidate, :471-487). */
enum { SK_OPFLAG_NONCANDIDATE_PENALTY = 1 };

typedef struct sk_score_op {
    uint16_t length;
    uint8_t kind;
    uint8_t flags;
    int32_t src; /* BASES: offset into this read's hap pool (>= 0, src+length <= pool size) */
} sk_score_op;

typedef struct sk_align_batch {
    int32_t n_reads;
    int32_t n_cals;           /* total candidate alignments = cal_off[n_reads] */
    int64_t n_ops;            /* = op_off[n_cals] */
    const int64_t* read_off;  /* [n_reads+1] into read_code/read_qual */
    const uint8_t* read_code; /* BAM 4-bit codes, 1/byte */
    const uint8_t* read_qual; /* phred, must be <= 70 (the reference throws above, qscore_cache.cpp:53-75) */
    const int64_t* hap_off;   /* [n_reads+1] per-read haplotype source pool into hap_code */
    const uint8_t* hap_code;  /* BAM 4-bit codes: the read's reference window ('N'/outside the contig segment = 15)
                                 followed by the insert sequences of the read's candidate indels */
    const int32_t* cal_off;   /* [n_reads+1] candidate-alignment range of each read */
    const int64_t* op_off;    /* [n_cals+1] op range of each candidate alignment; the BASES+SOFT_CLIP lengths of one
                                 candidate sum to its read's length */
    const sk_score_op* ops;   /* [n_ops] */
    int32_t max_read_len;     /* upper bound of any read length in the batch, 0 = unknown (selects the generic kernel) */
    int32_t max_hap_len;      /* upper bound of any per-read hap pool size, 0 = unknown */
    /* Device-ready form of `ops` (sk_align_prepare; sk_align_builder_finish fills it): transition entries and per-read
     * event masks, layout in strelka_amd/csrc/align_entry.h.  NULL = not prepared: sk_score_alignments prepares on the
     * fly, sk_score_alignments_dev falls back to the generic thread-per-alignment kernel. */
    const uint32_t* entries;  /* [n_ops + 2*n_cals] */
    const uint32_t* evmask;   /* [n_reads * evmask_words] */
    int32_t evmask_words;     /* sk_align_evmask_words(max_read_len) */
    /* Column form (sk_align_prepare_cols; sk_align_builder_finish fills it; needs `entries`): for every candidate alignment,
     * per read position, WHICH term its haplotype base selects -- 0: the bases agree, ln(1-e_q); 1: they differ, ln(e_q/3);
     * 2: nothing is added (read base N, soft-clipped position, position past the read's end) -- so that the kernel streams the
     * candidate haplotypes instead of following transitions: the bytes SURVEY.md 8d counts (H * L_h per read, here four bits
     * per base).  Read r owns colmat[colmat_off[r] ..): word [k * n_cals(r) + j] holds read positions 8k..8k+7 of the read's
     * j-th candidate alignment: position 8k+q in the low nibble of byte q, position 8k+4+q in its high nibble (q = 0..3).
     * Bit 2 of a nibble: ONE non-candidate-indel penalty (and nothing else) is added before this position's term.
     * addmask has bit p of read r set when an entry of ANY of its candidates adds penalty / soft-clip terms at read position p
     * (0 <= p <= read length); its last bit (32 * evmask_words - 1) is set when one of the read's candidates does not fit the
     * entry format, the bit before it when some entry adds more than the one penalty a nibble can flag (the kernel then takes
     * this read's added terms from its entries and ignores the flags).  NULL = absent: the kernel follows the entries. */
    const uint32_t* colmat;
    const int64_t* colmat_off; /* [n_reads+1], in words */
    const uint32_t* addmask;   /* [n_reads * evmask_words] */
} sk_align_batch;

/** Prepared form of a host batch (max_read_len must be set): entries[n_ops + 2*n_cals], evmask[n_reads * words]. */
int32_t sk_align_evmask_words(int32_t max_read_len);
int sk_align_prepare(const sk_align_batch* host_batch, uint32_t* entries, uint32_t* evmask);
/** Column form of a prepared host batch (`entries` set): colmat[sk_align_colmat_words], colmat_off[n_reads+1],
 *  addmask[n_reads * evmask_words]. */
int64_t sk_align_colmat_words(const sk_align_batch* host_batch);
int sk_align_prepare_cols(const sk_align_batch* host_batch, uint32_t* colmat, int64_t* colmat_off, uint32_t* addmask);

/** out_lnp[n_cals]: ln P(read | alignment), double, bit-identical to the reference's sequential accumulation. */
int sk_score_alignments(const sk_align_batch* host_batch, double* out_lnp);
int sk_score_alignments_dev(const sk_align_batch* dev_batch, double* dev_out_lnp, void* hip_stream);

/* Host adapter for hot path A: flattens the reference's own data model -- a read, the reference contig segment and a
 * set of CandidateAlignment objects (CIGAR path + indel keys, L/starling_common/CandidateAlignment.hh:35-78) -- into an
 * sk_align_batch.  The walk over the path is the one scoreCandidateAlignment performs
 * (L/starling_common/starling_read_align_score.cpp:286-493): swap handling (:306-347), edge-insert head position
 * (:334-338,394-398), indel-key lookup (getMatchingIndelKey :177-228) and the non-candidate penalty (:471-487). */

/** ALIGNPATH::align_t (L/blt_util/align_path.hh:36-48) */
enum { SK_SEG_NONE = 0, SK_SEG_MATCH, SK_SEG_INSERT, SK_SEG_DELETE, SK_SEG_SKIP, SK_SEG_SOFT_CLIP, SK_SEG_HARD_CLIP,
       SK_SEG_PAD, SK_SEG_SEQ_MATCH, SK_SEG_SEQ_MISMATCH };
/** INDEL::index_t (L/starling_common/indel_core.hh:57-66) */
enum { SK_INDEL_NONE = 0, SK_INDEL_INDEL, SK_INDEL_MISMATCH, SK_INDEL_BP_LEFT, SK_INDEL_BP_RIGHT };

typedef struct sk_path_seg {
    uint32_t type;
    uint32_t length;
} sk_path_seg;

typedef struct sk_indel_key { /* IndelKey (L/starling_common/IndelKey.hh:39-199) + IndelBuffer::isCandidateIndel */
    int32_t pos;
    int32_t type;
    uint32_t del_len;
    uint32_t ins_len;
    const char* ins_seq; /* ACGTN; for breakpoints the IndelData breakpoint insert sequence */
    int32_t is_candidate;
} sk_indel_key;

typedef struct sk_candidate_alignment {
    int32_t pos; /* al.pos */
    int32_t n_seg;
    const sk_path_seg* path;
    int32_t n_indels;
    const sk_indel_key* indels; /* IndelKey-sorted */
    sk_indel_key leading;       /* type SK_INDEL_NONE when absent */
    sk_indel_key trailing;
} sk_candidate_alignment;

typedef struct sk_align_builder sk_align_builder;
sk_align_builder* sk_align_builder_create(void);
void sk_align_builder_destroy(sk_align_builder* b);
void sk_align_builder_clear(sk_align_builder* b);
/** append the reads of `src` after those of `dst` (joins batches flattened on different host threads) */
int sk_align_builder_append(sk_align_builder* dst, const sk_align_builder* src);
/** Append one read and its candidate alignments.  read_code: BAM 4-bit codes one per byte; ref_seq/ref_offset/ref_len:
 *  the reference_contig_segment (positions outside it read as 'N', L/blt_util/reference_contig_segment.hh:46-51). */
int sk_align_builder_add_read(sk_align_builder* b, const uint8_t* read_code, const uint8_t* read_qual, int32_t read_len,
                              const char* ref_seq, int32_t ref_offset, int32_t ref_len,
                              const sk_candidate_alignment* cals, int32_t n_cals);
/** Host threads sk_align_builder_finish may use to compile the batch's transition entries: 1 = none (default -- the
 *  reference runs one process per core), n = up to n, 0 = up to 16 hardware threads. */
int sk_align_builder_set_host_threads(sk_align_builder* b, int32_t host_threads);
/** Fill `out` with HOST pointers into the builder (valid until the next clear/add/destroy). */
int sk_align_builder_finish(sk_align_builder* b, sk_align_batch* out);
/** error text of the last failing builder call */
const char* sk_align_builder_error(const sk_align_builder* b);

/* ------------------------------------------------------------------------------------------------------------------
 * Hot path A, whole read: realignAndScoreRead (L/starling_common/starling_read_align.cpp:2026-2126) as a batched job
 *
 *   stage 1 (host)  add_read : the gate (check_for_candidate_indel_overlap :219-270), input normalisation (:1998-2057),
 *                              candidate-alignment enumeration (candidate_alignment_search :859-1277 with
 *                              make_start_pos_alignment :394-584 / get_end_pin_start_pos :594-719) and flattening
 *   stage 2 (GPU)            : scoreCandidateAlignment for every (read, candidate alignment) of the job in one launch
 *   stage 3 (host)  finish   : max / smooth-pool selection with the reference's exact tie rules
 *                              (scoreCandidateAlignments :1536-1741), edge soft-clipping of ambiguous pools
 *                              (starling_read_align_clipper.cpp:343-424) and per-indel read support
 *                              (score_indels, starling_read_align_score_indels.cpp:455-1079)
 *   sk_realign_job_run = get_batch -> sk_score_alignments -> finish.
 *
 * Replaces the per-read call at L/starling_common/starling_pos_processor_base.cpp:752: the adapter queues the reads
 * buffered at the READ_BUFFER stage, runs the job, then writes `rseg.realignment/is_realigned` and
 * `IndelSampleData::read_path_lnp[readId]` from the results.
 * DNA reads only: spliced (RNA) read segments with pinned exon edges are not handled (north_star: germline/somatic DNA).
 * ---------------------------------------------------------------------------------------------------------------- */

enum { SK_MAX_SAMPLES = 8 }; /* (round 6: was 4; allele groups of such runs: sk_allele_group_genotype_lhoods_xwide) */
/** MAPLEVEL::index_t (L/blt_common/map_level.hh:32-39) */
enum { SK_MAPLEVEL_UNKNOWN = 0, SK_MAPLEVEL_TIER1 = 1, SK_MAPLEVEL_TIER2 = 2, SK_MAPLEVEL_SUB = 3, SK_MAPLEVEL_UNMAPPED = 4 };

typedef struct sk_realign_options { /* L/starling_common/starling_base_shared.hh */
    int32_t max_read_indel_toggle;        /* 5    (:139) */
    double max_candidate_indel_density;   /* 0.15 (:145) */
    uint32_t max_realignment_candidates;  /* 5000 (:160) */
    uint32_t max_indel_size;              /* 49   (:124) */
    int32_t is_smoothed_alignments;       /* 1    (:170) */
    double smoothed_lnp_range;            /* ln 10 (:171) */
    uint32_t upstream_oligo_size;         /* 0    (:206) */
    int32_t is_haplotyping_enabled;       /* germline 1, somatic 0 (:98, starling_shared.hh:52) */
    int32_t min_read_bp_flank;            /* 5; normal sample of a somatic run 1 */
    int32_t sample_count;                 /* 1..SK_MAX_SAMPLES */
    int32_t host_threads;                 /* host stages of sk_realign_job_add_reads / _finish: 1 = none (default: the reference runs one
                                             process per core), n = up to n threads, 0 = up to 16 hardware threads */
    int32_t enumeration;                  /* where candidate_alignment_search (:859-1277) runs:
                                             0 = host, container-based statement (the one pinned to the reference);
                                             1 = host, the container-free core of csrc/realign_core.h (what the device runs);
                                             2 = device: search, ordering / de-duplication and flattening of every read's
                                                 candidate alignments in kernels (csrc/read_enumerate.hip), scored where they
                                                 are; reads beyond the core's fixed capacities take path 0.
                                             sk_realign_options_default: 2 once sk_init has succeeded, else 0; the
                                             environment variable SK_ENUMERATION overrides the default.
                                             Results are identical in all three. */
} sk_realign_options;
void sk_realign_options_default(sk_realign_options* opt);

/** One entry of the IndelBuffer (IndelKey + the IndelData fields the path consults). */
typedef struct sk_indel_info {
    sk_indel_key key;                 /* key.is_candidate = IndelBuffer::isCandidateIndel */
    double ref_to_indel_log_prob;     /* getErrorRates().refToIndelErrorProb.getLogValue() of the read's sample */
    double indel_to_ref_log_prob;     /* getErrorRates().indelToRefErrorProb.getLogValue() */
    int32_t active_region_id;         /* IndelData::activeRegionId, < 0 = none */
    int8_t haplotype_id[SK_MAX_SAMPLES];            /* IndelSampleData::haplotypeId */
    uint8_t is_haplotyping_bypassed[SK_MAX_SAMPLES];/* IndelSampleData::isHaplotypingBypassed */
    uint8_t is_forced_output;         /* IndelData::isForcedOutput */
    uint8_t not_discovered_from_reads;/* IndelData::status.notDiscoveredFromReads */
} sk_indel_info;

typedef struct sk_read_input {
    const uint8_t* read_code;   /* BAM 4-bit codes, one per byte */
    const uint8_t* read_qual;
    int32_t read_len;
    int32_t pos;                /* input alignment (already left-normalised by the caller, as rseg.getInputAlignment()) */
    int32_t n_seg;
    const sk_path_seg* path;
    int32_t is_fwd_strand;
    int32_t map_level;          /* SK_MAPLEVEL_* (tier1/tier2 reads get indel scores) */
    int32_t sample_index;
    int32_t realign_begin, realign_end; /* realign_buffer_range */
    int32_t n_observed;         /* indels of the table this read was observed to support in its sample */
    const int32_t* observed;    /* (is_usable_indel :289-305): indices into the job's indel table (as given) */
} sk_read_input;

typedef struct sk_read_path_scores { /* ReadPathScores, L/starling_common/IndelData.hh:64-116 */
    int32_t indel;              /* index into the job's indel table (as given to set_indels) */
    float ref_lnp, indel_lnp;
    uint16_t non_ambig, read_length;
    uint8_t is_tier1_read, is_fwd_strand;
    int16_t read_pos, distance_from_closest_read_edge;
    int32_t n_alt;              /* <= 2 */
    int32_t alt_indel[2];
    float alt_lnp[2];
} sk_read_path_scores;

typedef struct sk_read_result {
    int32_t n_candidate_alignments; /* 0 = the read left at the gate (not realignable / no candidate overlap) */
    int32_t is_realigned;
    int32_t realign_pos;
    int32_t realign_n_seg;
    const sk_path_seg* realign_path;
    double max_score;               /* score of the max candidate alignment */
    int32_t n_scores;
    const sk_read_path_scores* scores;
    int32_t n_suboverlap;           /* indels whose breakpoint the read overlaps by 0 < bp < min_read_bp_flank */
    const int32_t* suboverlap;      /* (suboverlap_tier{1,2}_read_ids, score_indels :616-626) */
    int32_t warn_origin_skip, warn_max_toggle_depth;
} sk_read_result;

typedef struct sk_realign_job sk_realign_job;
sk_realign_job* sk_realign_job_create(const sk_realign_options* opt);
void sk_realign_job_destroy(sk_realign_job* job);
const char* sk_realign_job_error(const sk_realign_job* job);
/** reference_contig_segment: positions outside [offset, offset+len) read as 'N' */
int sk_realign_job_set_reference(sk_realign_job* job, const char* ref_seq, int32_t ref_offset, int32_t ref_len);
/** the indels visible to the job's reads (any order; sorted internally in IndelKey order) */
int sk_realign_job_set_indels(sk_realign_job* job, const sk_indel_info* indels, int32_t n_indels);
/** stage 1 for one read; returns the read's index in the job or -1 */
int sk_realign_job_add_read(sk_realign_job* job, const sk_read_input* read);
/** The same for n reads at once: gate, normalisation and enumeration (a1-a4) of the reads run on host threads, their
 *  candidate alignments enter the batch in input order.  Returns the index of the first read added (the rest follow
 *  consecutively), or -1 and adds nothing when any read is rejected (sk_realign_job_error names it). */
int sk_realign_job_add_reads(sk_realign_job* job, const sk_read_input* reads, int32_t n);
/** the flattened (read, candidate alignment) pairs of all reads added so far (host pointers owned by the job) */
int sk_realign_job_get_batch(sk_realign_job* job, sk_align_batch* out);
/** stage 3 from host scores laid out as get_batch's candidate alignments */
int sk_realign_job_finish(sk_realign_job* job, const double* scores);
/** stages 2+3 on the GPU */
int sk_realign_job_run(sk_realign_job* job);
int sk_realign_job_n_reads(const sk_realign_job* job);
int sk_realign_job_read_result(const sk_realign_job* job, int32_t read_index, sk_read_result* out);
/** out[i] = 1 when the candidate status (sk_indel_key.is_candidate) of indel i of the table -- as given to set_indels -- was
 *  consulted by any read of the job so far, at the places where the reference calls IndelBuffer::isCandidateIndel.  The
 *  reference computes and caches an indel's candidate status at its first such call (L/starling_common/IndelBuffer.hh:
 *  153-164); an adapter that had to evaluate the status of every indel of the table up front does that without caching and
 *  commits the cache only for the indels reported here, so that it caches what the reference would have. */
int sk_realign_job_indels_consulted(const sk_realign_job* job, uint8_t* out, int32_t n_indels);
/** How many reads had their candidate alignments enumerated by the container-free core on the host (enumeration == 1), on the
 *  device (== 2), and by the container-based code although 1 or 2 was asked for (a fixed capacity of the core was exceeded). */
int sk_realign_job_enumeration_counts(const sk_realign_job* job, int64_t* n_core, int64_t* n_device, int64_t* n_fallback);
/** Device jobs of this process so far (enumeration == 2; L/starling_common/starling_read_align.cpp:859-1277, :1536-1741 and
 *  starling_read_align_score.cpp:261-499 for every read of a job): run as ONE fixed sequence of launches with one host wait (search levels,
 *  sets, layout, F5 and stage 3 back to back, the prefix sums between them made on the device); of those, run again the staged way
 *  because the device reported that an assumption of the sequence did not hold (deeper search levels than launched, a full leaf pool, a
 *  read outside F5's form); run the staged way (three waits) from the start. */
void sk_realign_device_job_counts(int64_t* n_one_wait, int64_t* n_one_wait_redone, int64_t* n_staged);
/** How many reference bases the realignment jobs of this process have read OUTSIDE the segment given to sk_realign_job_set_reference
 *  (such a position reads as 'N', reference_contig_segment::get_base, L/blt_util/reference_contig_segment.hh:46-51), host stages and
 *  kernels together, since the process started.  A caller whose contig segment is megabases long hands a job the WINDOW of it that the
 *  job's reads can reach (the job copies its reference: 12 MB per job otherwise) and takes the difference of this number around the job: if
 *  it moved and the window was not the whole segment, the window was too narrow and the job is run again with the whole segment. */
int64_t sk_realign_reference_reads_outside(void);
/** How many reads had stage 3 (scoreCandidateAlignments' selection, finishRealignment and score_indels, L/starling_common/
 *  starling_read_align.cpp:1534-1741 and starling_read_align_score_indels.cpp:455-1079) run in its container-free form on the
 *  host (enumeration == 1) and on the device (== 2); the rest went through the container-based code. */
int sk_realign_job_stage3_counts(const sk_realign_job* job, int64_t* n_core, int64_t* n_device);
/** drop reads and results, keep reference/indels/options */
void sk_realign_job_clear_reads(sk_realign_job* job);
/** Measurement: scoreCandidateAlignment (L/starling_common/starling_read_align_score.cpp:261-499) as the device pipeline performs it for
 *  the job that ran last with enumeration == 2 -- from the candidate alignments as the search left them (position, path, indels)
 *  to one double each: the haplotype bytes every alignment faces, its ops, the base comparisons (F1-F3) AND the table sums (A1c),
 *  `reps` times over the resident records.  out_ms: elapsed time of all repetitions (events on the library's stream);
 *  cells = read bases x candidate alignments of one repetition. */
int sk_realign_job_rescore(int32_t reps, float* out_ms, int32_t* out_n_reads, int32_t* out_n_cals, int64_t* out_cells);

/* building blocks, exposed for known-answer tests against the reference's own unit tests
 * (L/starling_common/test/starling_read_align_test.cpp:67-335) */
/** make_start_pos_alignment: out_path capacity >= 2*n_indels+3; returns the number of path segments or -1.
 *  out_leading/out_trailing: index into `indels` or -1. */
int sk_make_start_pos_alignment(int32_t ref_start_pos, int32_t read_start_pos, int32_t is_fwd_strand, uint32_t read_length,
                                const sk_indel_key* indels, int32_t n_indels, int32_t* out_pos, sk_path_seg* out_path,
                                int32_t path_cap, int32_t* out_leading, int32_t* out_trailing);
/** get_end_pin_start_pos */
int sk_get_end_pin_start_pos(const sk_indel_key* indels, int32_t n_indels, uint32_t read_length, int32_t ref_end_pos,
                             int32_t read_end_pos, int32_t* out_ref_start_pos, int32_t* out_read_start_pos);

/* ------------------------------------------------------------------------------------------------------------------
 * Row a8: pileup of aligned reads into per-locus basecall columns
 *
 * Replaces, for a batch of reads, the per-read calls of starling_pos_processor_base::pileup_read_segment
 * (L/starling_common/starling_pos_processor_base.cpp:1127-1421, called from pileup_pos_reads :1107-1123 at stage
 * READ_BUFFER) including the mismatch-density filter (create_mismatch_filter_map, L/starling_common/starling_read_util.cpp:
 * 121-213), the MAPQ-adjusted basecall quality (qphred_to_mapped_qphred, L/blt_util/qscore.hh:104-121) and the ambiguous
 * read-end trim (getReadAmbiguousEndLength, L/htsapi/bam_seq_read_util.cpp:29-54); the columns come out as the CSR
 * sk_pileup_batch the path-B entry points consume -- raw, or already cleaned as PileupCleaner::CleanPileupFilter
 * (L/starling_common/PileupCleaner.cpp:28-66) would.  Calls of a locus appear in read order (the order of `reads`,
 * = the read buffer's order), as insert_pos_basecall appends them.
 * Not produced: the EVS feature accumulators (updateGermlineScoringMetrics / updateSomaticScoringMetrics) and the MAPQ
 * tracker -- scoring-model inputs, outside the likelihood path.
 * ---------------------------------------------------------------------------------------------------------------- */

/** Longest read the pileup kernels take (a read's per-base state lives in LDS); the reference's own limit is
 *  STRELKA_MAX_READ_SIZE = 25000 (L/starling_common/starling_base_shared.hh:37).  sk_pileup_reads and the stream pushes fail on a longer
 *  read; the adapter says so when such a read arrives and points at STRELKA_AMD_PILEUP=0 (limits). */
#define SK_PILEUP_MAX_READ_LEN 1024

typedef struct sk_pileup_options {
    int32_t min_basecall_qscore;              /* blt_options::minBasecallErrorPhredProb: 17 germline (blt_shared.hh:107), 0 somatic */
    int32_t mismatch_density_flank_size;      /* 20 (starling_shared.hh:37); 0 = filter off */
    int32_t mismatch_density_max_count;       /* 2 germline (starling_shared.hh:36), 3 somatic (strelka_shared.hh:70) */
    int32_t use_tier2_evidence;               /* somatic 1 (strelka_shared.hh:78) */
    int32_t tier2_mismatch_density_max_count; /* opt.tier2.mismatchDensityFilterMaxMismatchCount */
    int32_t is_mapq_adjust;                   /* isBasecallQualAdjustedForMapq, 1 (starling_base_shared.hh:225) */
    int32_t min_distance_from_read_edge;      /* 0 (:252) */
    int32_t largest_total_indel_ref_span_per_read; /* the processor's running value, >= maxIndelSize (49) */
    int32_t report_begin, report_end;         /* _reportRange; locus index = ref position - report_begin */
} sk_pileup_options;
void sk_pileup_options_default(sk_pileup_options* opt);

/** reads with their best alignment (the realignment when the read was realigned, else the input alignment) */
typedef struct sk_read_batch {
    int32_t n_reads;
    const int64_t* read_off;    /* [n_reads+1] */
    const uint8_t* read_code;   /* BAM 4-bit codes, one per byte: '=',A,C,G,T,N only */
    const uint8_t* read_qual;
    const int64_t* path_off;    /* [n_reads+1] */
    const sk_path_seg* path;
    const int32_t* pos;         /* best_al.pos */
    const uint8_t* is_fwd;      /* best_al.is_fwd_strand */
    const uint8_t* mapq;        /* rseg.map_qual() */
    const uint8_t* map_level;   /* SK_MAPLEVEL_* */
    const char* ref_seq;        /* reference_contig_segment; positions outside read as 'N' */
    int32_t ref_offset, ref_len;
    const uint8_t* cand_snv_mask; /* optional [ref_len]: bit b set = base id b is a candidate SNV of an active region
                                     at that position (CandidateSnvBuffer::isCandidateSnvAnySample); NULL = none */
} sk_read_batch;

enum { SK_PILEUP_RAW_TIER1 = 0,   /* snp_pos_info::calls */
       SK_PILEUP_RAW_TIER2 = 1,   /* snp_pos_info::tier2_calls */
       SK_PILEUP_CLEAN_TIER1 = 2, /* CleanPileupFilter(pi, false) */
       SK_PILEUP_CLEAN_TIER2 = 3  /* CleanPileupFilter(pi, true): tier1 calls (tier-specific filter waived) then tier2 calls */ };

typedef struct sk_pileup_columns {
    int32_t n_loci;           /* in: report_end - report_begin */
    int64_t capacity;         /* in: entries available in `calls` (total read bases always suffices; twice that for
                                 SK_PILEUP_CLEAN_TIER2) */
    int64_t* call_off;        /* out [n_loci+1] */
    uint16_t* calls;          /* out */
    uint32_t* spandel_count;  /* out [n_loci]: spanningDeletionReadCount; may be NULL */
    uint32_t* submapped_count;/* out [n_loci]: submappedReadCount; may be NULL */
} sk_pileup_columns;

/** host arrays in and out */
int sk_pileup_reads(const sk_read_batch* host_reads, const sk_pileup_options* opt, int mode, sk_pileup_columns* host_out);
/** device arrays in and out (ref_seq/cand_snv_mask device pointers too); dev_scratch >= sk_pileup_scratch_bytes() */
int64_t sk_pileup_scratch_bytes(int32_t n_reads, int64_t n_bases, int32_t n_loci);
int sk_pileup_reads_dev(const sk_read_batch* dev_reads, int64_t n_bases, const sk_pileup_options* opt, int mode,
                        sk_pileup_columns* dev_out, void* dev_scratch, void* hip_stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Next row: GlobalAligner<int>::align, the haplotype-to-reference affine-gap aligner of the
 * active-region code (L/alignment/GlobalAlignerImpl.hh:35-228; used at L/starling_common/ActiveRegionProcessor.cpp:591).
 * Integer DP with match / delete / insert states, off-edge soft clipping, optional edge insertion and required edge
 * deletion; results (score, begin position, CIGAR in '='/'X' form) are identical to the reference's, including the
 * max3 tie order (L/alignment/AlignerBase.hh:75-95) and the order in which traceback start points are considered.
 * ---------------------------------------------------------------------------------------------------------------- */

typedef struct sk_align_scores { /* AlignmentScores<int>, L/alignment/AlignmentScores.hh:27-58 */
    int32_t match, mismatch, open, extend, off_edge, insert_delete;
    int32_t is_allow_edge_insertion, is_require_edge_deletion;
} sk_align_scores;
/** the active-region detector's scores (L/starling_common/ActiveRegionDetector.hh:59-63, .cpp:41) */
void sk_align_scores_default(sk_align_scores* s);

typedef struct sk_global_align_batch {
    int32_t n;               /* problems */
    const int64_t* query_off;/* [n+1] */
    const char* query;       /* symbols compared with ==; 'N' never matches in the '='/'X' expansion */
    const int64_t* ref_off;  /* [n+1] */
    const char* ref;
} sk_global_align_batch;

/** out_score[n], out_begin_pos[n]; problem p's path goes to out_path[path_off[p] .. path_off[p] + out_n_seg[p]) with
 *  path_off[p] = query_off[p] + ref_off[p] + 4*p (capacity query+ref+4 segments: always enough).
 *  Query and reference lengths must be in 1..1024 (the reference asserts non-empty). */
int sk_global_align(const sk_global_align_batch* host_batch, const sk_align_scores* scores, int32_t* out_score,
                    int32_t* out_begin_pos, sk_path_seg* out_path, int32_t* out_n_seg);
/** Device-resident twin: the batch arrays and the four outputs are device pointers (out_path: capacity
 *  total_query_len + total_ref_len + 4n segments, same per-problem offsets).  max_*_len: upper bounds of the problem sizes.
 *  dev_scratch: sk_global_align_scratch_bytes(...) bytes (reversed paths + one back-pointer byte per DP cell). */
size_t sk_global_align_scratch_bytes(int32_t n, int64_t total_query_len, int64_t total_ref_len, int32_t max_query_len,
                                     int32_t max_ref_len);
int sk_global_align_dev(const sk_global_align_batch* dev_batch, int64_t total_query_len, int64_t total_ref_len,
                        int32_t max_query_len, int32_t max_ref_len, const sk_align_scores* scores, int32_t* dev_out_score,
                        int32_t* dev_out_begin_pos, sk_path_seg* dev_out_path, int32_t* dev_out_n_seg, void* dev_scratch,
                        void* hip_stream);

typedef struct sk_discovered_allele { /* an IndelKey found in a haplotype */
    int32_t pos;
    int32_t type;      /* SK_INDEL_INDEL or SK_INDEL_MISMATCH */
    uint32_t del_len;
    uint32_t ins_len;
    int32_t ins_off;   /* the insert sequence is out_ins_seq[ins_off .. ins_off + ins_len) */
} sk_discovered_allele;

/** Replaces the part of ActiveRegionProcessor::discoverIndelsAndMismatches after the aligner call
 *  (L/starling_common/ActiveRegionProcessor.cpp:594-705): walks the path sk_global_align returned for `haplotype` against
 *  the active region [ar_begin, ar_end) of the reference segment (ref_seq covers [ref_offset, ref_offset + ref_len), 'N'
 *  outside), left-shifts insertions and deletions as far as the reference sequence allows, drops indels longer than
 *  max_indel_size, shifted into the previous active region (pos < prev_ar_end), next to an 'N' or past ar_end, and reports
 *  one MISMATCH key per mismatched base at or after prev_ar_end.  *n_indels = the reference's numIndels. */
int sk_discover_indels_and_mismatches(const char* ref_seq, int32_t ref_offset, int32_t ref_len, int32_t ar_begin,
                                      int32_t ar_end, int32_t prev_ar_end, uint32_t max_indel_size, const char* haplotype,
                                      int32_t hap_len, int32_t align_begin_pos, const sk_path_seg* path, int32_t n_seg,
                                      sk_discovered_allele* out, int32_t out_cap, char* out_ins_seq, int32_t ins_cap,
                                      int32_t* n_out, int32_t* n_indels);

/* ------------------------------------------------------------------------------------------------------------------
 * Hot path B (germline SNV): dependent error probabilities + diploid genotype likelihoods
 * ---------------------------------------------------------------------------------------------------------------- */

/** 16-bit packed basecall, identical bit layout to the reference's `base_call` bitfield
 *  (L/blt_common/snp_pos_info.hh:109-117, GCC little-endian allocation):
 *  bits 0-5 qscore, 6-9 base_id (0..3 = ACGT), 10 is_fwd_strand, 11 is_neighbor_mismatch, 12 is_call_filter,
 *  13 is_tier_specific_call_filter. */
#define SK_CALL_Q(c) ((unsigned)((c) & 0x3f))
#define SK_CALL_BASE(c) ((unsigned)(((c) >> 6) & 0xf))
#define SK_CALL_FWD(c) ((unsigned)(((c) >> 10) & 1))
#define SK_CALL_NMM(c) ((unsigned)(((c) >> 11) & 1))
#define SK_CALL_FILTER(c) ((unsigned)(((c) >> 12) & 1))
#define SK_CALL_TSCF(c) ((unsigned)(((c) >> 13) & 1))
#define SK_MAKE_CALL(q, base, fwd, nmm, filt, tscf) \
    ((uint16_t)(((q) & 0x3f) | (((base) & 0xf) << 6) | (((fwd) & 1) << 10) | (((nmm) & 1) << 11) | (((filt) & 1) << 12) | (((tscf) & 1) << 13)))

typedef struct sk_pileup_batch {
    int32_t n_loci;
    const int64_t* call_off; /* [n_loci+1] into calls/de */
    const uint16_t* calls;   /* cleaned pileup (PileupCleaner::CleanPileupFilter output), pileup order */
    const float* de;         /* dependent error prob per call (epi.de); may be NULL only for sk_dependent_eprob */
    const uint8_t* ref_base; /* [n_loci] base id 0..3, 4 = 'N' */
    const uint8_t* ploidy;   /* [n_loci] 0,1,2; NULL = all diploid (dgt.ploidy, position_snp_call_pprob_digt.hh:100) */
} sk_pileup_batch;

/** Options of the germline SNV model (defaults: L/applications/starling/starling_shared.hh:34-39,
 *  L/blt_common/blt_shared.hh:82-84). */
typedef struct sk_germline_options {
    double bsnp_diploid_theta;    /* 0.001 */
    double bsnp_ssd_no_mismatch;  /* 0.35 */
    double bsnp_ssd_one_mismatch; /* 0.6 */
    int32_t is_min_vexp;          /* 1 */
    double min_vexp;              /* 0.25 */
} sk_germline_options;
void sk_germline_options_default(sk_germline_options* opt);

/** a9: replaces adjust_joint_eprob at L/starling_common/PileupCleaner.cpp:73 (L/blt_common/adjust_joint_eprob.cpp:201-243).
 *  out_de[total calls]. */
int sk_dependent_eprob(const sk_pileup_batch* host_batch, const sk_germline_options* opt, float* out_de);
/** dev_scratch: device buffer of >= 4 bytes per call (index array for the per-group sort). */
int sk_dependent_eprob_dev(const sk_pileup_batch* dev_batch, const sk_germline_options* opt, float* dev_out_de,
                           void* dev_scratch, void* hip_stream);

typedef struct sk_digt_result_set { /* diploid_genotype::result_set, position_snp_call_pprob_digt.hh:72-91 */
    double ref_pprob;
    uint32_t max_gt;
    int32_t snp_qphred;
    int32_t max_gt_qphred;
    int32_t _pad;
} sk_digt_result_set;

typedef struct sk_digt_call { /* diploid_genotype + the raw likelihoods */
    float lhood[10];           /* get_diploid_gt_lhood, DIGT order AA,CC,GG,TT,AC,AG,AT,CG,CT,GT */
    uint32_t phredLoghood[10]; /* PLs (:499-511); entries >= 4 are 0 for haploid loci */
    sk_digt_result_set genome; /* genomic prior */
    sk_digt_result_set poly;   /* polymorphic-site prior */
    double strand_bias;        /* :520-538 */
    uint32_t ref_gt;
    uint32_t is_called;        /* 0 when the reference returns early (ref base 'N', :481) */
} sk_digt_call;

/** a10: replaces pprob_digt_caller::position_snp_call_pprob_digt(opt, epi, dgt, is_always_test=true) at
 *  L/applications/starling/starling_pos_processor.cpp:265-266 (L/blt_common/position_snp_call_pprob_digt.cpp:473-539).
 *  out[n_loci]. */
int sk_site_digt_call(const sk_pileup_batch* host_batch, const sk_germline_options* opt, sk_digt_call* out);
int sk_site_digt_call_dev(const sk_pileup_batch* dev_batch, const sk_germline_options* opt, sk_digt_call* dev_out,
                          void* hip_stream);

/** a9+a10 fused: CleanPileupErrorProb followed by position_snp_call_pprob_digt for every locus in one pass
 *  (L/applications/starling/starling_pos_processor.cpp:146-196 calls them back to back per locus).  The pileup is staged
 *  once through LDS; `de` never travels through HBM unless `out_de` is non-NULL.  batch.de is ignored.
 *  dev variant: dev_de_tmp >= 4 bytes per call and dev_scratch >= 4 * (n_calls + n_loci + 4) bytes (n_calls = the
 *  batch's call_off[n_loci], which the host knows) serve the few loci that leave the LDS path -- sort scratch, then a
 *  work-list of their indices; dev_de_tmp doubles as the `de` output when want_de != 0. */
int sk_site_digt_call_fused(const sk_pileup_batch* host_batch, const sk_germline_options* opt, sk_digt_call* out,
                            float* out_de /* may be NULL */);
int sk_site_digt_call_fused_dev(const sk_pileup_batch* dev_batch, const sk_germline_options* opt, sk_digt_call* dev_out,
                                float* dev_de_tmp, int want_de, void* dev_scratch, int64_t n_calls, void* hip_stream);

/* ---- row a8 as a stream over a genome segment, chained into a9+a10 (sk_pileup_reads -> sk_site_digt_call_fused) --------
 * The reference piles the reads of position P up when its READ_BUFFER stage reaches P (pileup_pos_reads,
 * L/starling_common/starling_pos_processor_base.cpp:1107-1123, called at :813) and genotypes a position when the POST_ALIGN
 * stage -- largest_total_indel_ref_span_per_read positions behind (:141-224) -- gets there (process_pos_variants :821-890 ->
 * CleanPileupFilter, PileupCleaner.cpp:28-66; adjust_joint_eprob :73; position_snp_call_pprob_digt,
 * L/applications/starling/starling_pos_processor.cpp:256-267).  A caller that batches by stage windows pushes one window's reads
 * at a time, in read-buffer order, and names the position `final_to` below which no later read can add a basecall; the push
 * returns everything the position processor keeps per position for [begin, end): the raw tier1 / tier2 columns
 * (snp_pos_info::calls / tier2_calls), spanningDeletionReadCount, submappedReadCount, the MapqTracker sums
 * (L/blt_common/MapqTracker.hh:36-42, fed by insert_mapq_count at starling_pos_processor_base.cpp:1346) and -- when the stream
 * was created with germline options -- the diploid genotype of the CleanPileupFilter'ed tier1 column.  The columns never
 * leave the device between the pileup and the genotype kernels; reads that reach past `final_to` stay on the device for the
 * next push.  Not produced: the EVS feature accumulators (updateGermlineScoringMetrics / updateSomaticScoringMetrics).
 * The returned pointers are into a host buffer of the stream (page-locked: the device writes it).  A stream keeps
 * SK_PILEUP_WINDOW_LIFETIME + 1 such buffers in rotation: a window's arrays stay valid through the stream's next
 * SK_PILEUP_WINDOW_LIFETIME pushes and are overwritten by the one after -- long enough for a caller whose POST_ALIGN stage runs a
 * window or two behind its pushes to read the bulk of a window (the EVS words: 8 bytes a basecall, asked for at one position in a
 * few hundred) where the push left it instead of copying it out (adapter/sk_adapter_pileup.cpp, SiteChunk::evsLive). */
#define SK_PILEUP_WINDOW_LIFETIME 2
typedef struct sk_pileup_stream sk_pileup_stream;

/** What the gVCF writer's non-variant block logic reads of a position (gvcf_writer::queue_site_record, L/applications/starling/
 *  gvcf_writer.cpp:278-302; gvcf_block_site_record.cpp:30-184), made on the device from the position's cleaned column and genotype record
 *  (csrc/gvcf_site_core.h): whether process_pos_snp_digt (L/applications/starling/starling_pos_processor.cpp:619-701) would build a
 *  homozygous-reference site locus with no alternate allele for it -- flags bit 0, "plain" -- and, for such a site, its GQX
 *  (LocusSampleInfo::setGqx) and the reference allele's AD counts by strand.  The depth numbers (used / unused basecalls) are
 *  clean_count and the tier1 column's length. */
typedef struct sk_gvcf_site_summary {
    uint32_t flags;            /* bit 0: plain site (see above); valid for the ploidy the window was genotyped with */
    int32_t gqx;               /* min(genome.max_gt_qphred, poly.max_gt_qphred) */
    uint32_t ref_fwd, ref_rev; /* cleaned basecalls equal to the reference base, forward / reverse strand */
} sk_gvcf_site_summary;

/** What decides a plain site's FILTERs and whether it joins the sample's open block: the options ScoringModelManager::applyDepthFilter /
 *  default_classify_site (L/applications/starling/ScoringModelManager.cpp:234-249, :270-311) and gvcf_block_site_record (tolerances,
 *  gvcf_block_site_record.hh:38-44) read.  max_chrom_depth belongs to the chromosome (ScoringModelManager::resetChrom :80-97). */
typedef struct sk_gvcf_block_options {
    uint32_t min_passed_call_depth;  /* gvcf_options::minPassedCallDepth (LowDepth) */
    int32_t is_min_homref_gqx;       /* LowGQX of a homozygous-reference site: gqx < min_homref_gqx */
    double min_homref_gqx;
    int32_t is_max_depth;            /* HighDepth: the chromosome's depth is known and MapqTracker::count > max_chrom_depth */
    int32_t is_max_base_filt;        /* HighBaseFilt: unused / (used + unused) > max_base_filt */
    double max_chrom_depth;
    double max_base_filt;
    uint32_t block_percent_tol, block_abs_tol;
} sk_gvcf_block_options;

/** For a plain site i: the non-variant block that STARTS at it -- gvcf_writer::queue_site_record's greedy joining
 *  (testCanSiteJoinSampleBlock / joinSiteToSampleBlock, gvcf_block_site_record.cpp:77-182) from an empty block over the plain sites
 *  that follow it in the window: how many sites it takes in, and the minimum and maximum of the block's three stream_stat accumulators
 *  after the last of them (the join test reads nothing else of them: check_block_tolerance :41-55; their running means -- a division per
 *  member, read only when a block is written -- are the caller's to make for the blocks it uses).  len = 0: not a plain site.  The block
 *  ends where a site cannot join, at a site that is not plain, and at the window's end (where it may well go on: the caller lets the
 *  next site decide). */
typedef struct sk_gvcf_run {
    int32_t len;
    uint32_t filter_key;           /* which of LowDepth / LowGQX / HighDepth / HighBaseFilt the block's sites carry (bits 0-3) */
    int32_t gqx_min, gqx_max;
    uint32_t dpu_min, dpu_max, dpf_min, dpf_max;
} sk_gvcf_run;

typedef struct sk_pileup_window {
    int32_t begin, end;              /* positions [begin, end); n = end - begin */
    const int64_t* tier1_off;        /* [n+1] */
    const uint16_t* tier1_calls;
    const int64_t* tier2_off;        /* [n+1] */
    const uint16_t* tier2_calls;
    const uint32_t* spandel_count;   /* [n] */
    const uint32_t* submapped_count; /* [n] */
    const uint32_t* mapq_count;      /* [n] MapqTracker::count */
    const uint32_t* mapq_zero_count; /* [n] MapqTracker::zeroCount */
    const uint64_t* mapq_sum_square; /* [n] MapqTracker::sumSquare (a sum of integer squares: exact) */
    const uint32_t* clean_count;     /* [n] calls in the cleaned tier1 column the genotype was computed from */
    const sk_digt_call* genotype;    /* [n], NULL when the stream does not genotype */
    const int64_t* evs_off;          /* [n+1], NULL unless sk_pileup_stream_enable_evs_words: a position has mapq_count words */
    const uint64_t* evs_words;       /* one per live match position of every read, submapped reads included, in pileup order:
                                        base id (bits 0-2) | mapq << 3 | qscore << 11 (MAPQ-adjusted, not capped) | cycle << 18
                                        (align_strand_read_pos) | min(20, distance from the read edge) << 29 | is_submapped << 34
                                        (a submapped position carries base id, mapq and the flag only) */
    const sk_gvcf_site_summary* site_summary; /* [n], NULL when the stream does not genotype */
    const sk_gvcf_run* gvcf_runs;             /* [n], NULL unless sk_pileup_stream_set_gvcf_block_options gave the stream options */
} sk_pileup_window;

/** genotype_opt: NULL = columns only.  opt->report_begin / report_end / largest_total_indel_ref_span_per_read are set per
 *  region and per push. */
sk_pileup_stream* sk_pileup_stream_create(const sk_pileup_options* opt, const sk_germline_options* genotype_opt);
void sk_pileup_stream_destroy(sk_pileup_stream* s);
/** The germline EVS accumulators (updateGermlineScoringMetrics, starling_pos_processor_base.cpp:1346-1357 ->
 *  pos_basecall_buffer.cpp:43-70: mq_ranksum, baseq_ranksum, readPositionRankSum, distanceFromReadEdge) are fed for every basecall
 *  of every read when the germline EVS models are loaded, and read only where a variant record is scored
 *  (L/applications/starling/starling_pos_processor.cpp:235-246).  With this switched on (before the first region) a window also
 *  returns the per-call arguments of that function, evs_off / evs_words, from which the caller rebuilds the four accumulators of the
 *  positions it needs them for. */
int sk_pileup_stream_enable_evs_words(sk_pileup_stream* s, int enable);
/** With options (before or at the start of a region; NULL: off) a genotyping stream also returns, for every plain site of a window, the
 *  non-variant block that would start at it (sk_pileup_window.gvcf_runs): gvcf_block_kernel's walk (csrc/gvcf_block_core.h) made from
 *  every plain site, on the device, behind the site summaries. */
int sk_pileup_stream_set_gvcf_block_options(sk_pileup_stream* s, const sk_gvcf_block_options* opt);
/** resetRegionBase (starling_pos_processor_base.cpp:361-393): the reference segment of the region and its report range */
int sk_pileup_stream_begin_region(sk_pileup_stream* s, const char* ref_seq, int32_t ref_offset, int32_t ref_len,
                                  int32_t report_begin, int32_t report_end, int32_t largest_total_indel_ref_span_per_read);
/** host_reads: the window's reads with their best alignments (ref_seq / ref_offset / ref_len / cand_snv_mask of the batch are
 *  ignored: the stream holds the region's); cand_snv_mask[mask_len] = CandidateSnvBuffer::isCandidateSnvAnySample for
 *  positions [mask_begin, mask_begin + mask_len) as of now (it must cover the new reads; may be NULL with mask_len 0);
 *  final_to: no read of a later push has a basecall or spanning deletion below it (INT32_MAX at the end of a region);
 *  ploidy[ploidy_len]: caller ploidy (1 or 2) of positions from ploidy_begin on, 2 elsewhere (NULL: all 2).
 *  Fails if a read reaches below the previous push's final_to. */
int sk_pileup_stream_push(sk_pileup_stream* s, const sk_read_batch* host_reads, int32_t largest_total_indel_ref_span_per_read,
                          int32_t mask_begin, int32_t mask_len, const uint8_t* cand_snv_mask, int32_t final_to,
                          int32_t ploidy_begin, int32_t ploidy_len, const uint8_t* ploidy, sk_pileup_window* out);
/** The same push in two halves.  _begin checks the reads, packs them (the caller's arrays are free again when it returns) and enqueues
 *  the window's work; _finish waits for the device and hands out the window.  Between the two the stream takes no other call
 *  (_begin_region and a second _begin fail); other entry points of the library may be called -- they run behind the window's work on the
 *  process's stream.  For a caller with host work of its own before it needs the window: the reference's POST_ALIGN stage trails the
 *  READ_BUFFER stage by largest_total_indel_ref_span_per_read positions (starling_pos_processor_base.cpp:141-224) plus what the
 *  adapter defers it by, and every head position in between is read intake the device's ~0.4 ms can hide behind. */
int sk_pileup_stream_push_begin(sk_pileup_stream* s, const sk_read_batch* host_reads, int32_t largest_total_indel_ref_span_per_read,
                                int32_t mask_begin, int32_t mask_len, const uint8_t* cand_snv_mask, int32_t final_to,
                                int32_t ploidy_begin, int32_t ploidy_len, const uint8_t* ploidy);
int sk_pileup_stream_push_finish(sk_pileup_stream* s, sk_pileup_window* out);

/* ------------------------------------------------------------------------------------------------------------------
 * Hot path B (somatic SNV): 30-state frequency-grid likelihoods + 3x2 posterior
 * ---------------------------------------------------------------------------------------------------------------- */

enum { SK_SOM_PRESTRAND = 21, SK_SOM_STATES = 30 };

typedef struct sk_somatic_snv_options { /* L/applications/strelka/strelka_shared.hh; workflow overrides in
                                           src/python/bin/configureStrelkaSomaticWorkflow.py.ini */
    double bsnp_diploid_theta;                      /* 0.001 */
    double somatic_snv_rate;                        /* ssnvPrior 1e-4 */
    double shared_site_error_rate;                  /* ssnvNoise 5e-10 */
    double shared_site_error_strand_bias_fraction;  /* 0 */
    double ssnv_contam_tolerance;                   /* 0.15 */
} sk_somatic_snv_options;
void sk_somatic_snv_options_default(sk_somatic_snv_options* opt);

typedef struct sk_somatic_snv_call { /* snv_result_set, L/applications/strelka/somatic_result_set.hh:32-54 */
    float normal_lhood[SK_SOM_STATES]; /* entries 21..29 unused (0) for the normal sample */
    float tumor_lhood[SK_SOM_STATES];
    uint32_t max_gt;
    int32_t qphred;            /* QSS */
    int32_t from_ntype_qphred; /* QSS_NT */
    uint32_t ntype;            /* SOMATIC_DIGT index of the most likely normal genotype (pre NTYPE remap) */
    float strand_bias;
    uint32_t is_called;        /* 0 = early-out (ref 'N' or both pileups all-ref, :251-254) */
    uint32_t normal_alt_id;
    uint32_t tumor_alt_id;
} sk_somatic_snv_call;

/** a12+a13: replaces somatic_snv_caller_strand_grid::position_somatic_snv_call (single tier) at
 *  L/applications/strelka/strelka_pos_processor.cpp:213-219
 *  (L/applications/strelka/position_somatic_snv_strand_grid.cpp:230-363, qscore_calculator.cpp:47-209).
 *  normal/tumor batches must have the same n_loci and ref_base; `de` is ignored (raw error_prob(q) is used). */
int sk_somatic_snv_call_batch(const sk_pileup_batch* host_normal, const sk_pileup_batch* host_tumor,
                              const sk_somatic_snv_options* opt, int is_forced_output, sk_somatic_snv_call* out);
/* dev_scratch: >= 4 * (n_loci + 4) bytes of device memory (the queue of loci that are not skipped) */
int sk_somatic_snv_call_batch_dev(const sk_pileup_batch* dev_normal, const sk_pileup_batch* dev_tumor,
                                  const sk_somatic_snv_options* opt, int is_forced_output,
                                  sk_somatic_snv_call* dev_out, void* dev_scratch, void* hip_stream);

/** The whole of position_somatic_snv_call: somatic_snv_genotype_grid (L/applications/strelka/somatic_result_set.hh:56-79)
 *  as the reference leaves it -- all-zero result fields on its early returns. */
typedef struct sk_somatic_snv_genotype {
    uint32_t ref_gt;              /* 0 when the reference base is 'N' */
    uint8_t snv_tier;             /* tier of rs.qphred (:324-329) */
    uint8_t snv_from_ntype_tier;  /* tier of every other rs field (:331-337) */
    uint8_t is_forced_output;     /* the input flag, cleared when the reference base is 'N' (:246) */
    uint8_t is_computed;          /* 1 when the locus got past the all-reference early return (:251-254) */
    uint32_t ntype;               /* NTYPE::REF / HOM / HET / CONFLICT = 0..3 (somatic_call_shared.hh:32-40) */
    uint32_t max_gt;              /* DDIGT state of the chosen tier */
    int32_t qphred;               /* QSS */
    int32_t from_ntype_qphred;    /* QSS_NT */
    int32_t nonsomatic_qphred;    /* tier1, only with is_compute_nonsomatic */
    uint32_t normal_alt_id, tumor_alt_id;
    int32_t _pad;
    double strand_bias;           /* snv_result_set::strandBias */
} sk_somatic_snv_genotype;

/** a12+a13 complete: replaces somatic_snv_caller_strand_grid::position_somatic_snv_call at
 *  L/applications/strelka/strelka_pos_processor.cpp:213-219 for a batch of loci, with the reference's tier logic
 *  (L/applications/strelka/position_somatic_snv_strand_grid.cpp:268-362): the all-reference early return is decided on the
 *  tier1 pileups; tier2 is evaluated only where tier1 gave qphred != 0 (otherwise its result is tier1's); snv_tier /
 *  snv_from_ntype_tier / NTYPE conflict as in :321-359; non-somatic quality (:186-214) with is_compute_nonsomatic.
 *  normal_t1/tumor_t1: CleanPileupFilter(pi,false) columns; normal_t2/tumor_t2: CleanPileupFilter(pi,true) columns, both
 *  NULL = no tier2 evidence (opt.useTier2Evidence off).  is_forced_output: [n_loci] or NULL.  All batches share n_loci
 *  and ref_base. */
int sk_somatic_snv_call_tiers(const sk_pileup_batch* host_normal_t1, const sk_pileup_batch* host_tumor_t1,
                              const sk_pileup_batch* host_normal_t2, const sk_pileup_batch* host_tumor_t2,
                              const sk_somatic_snv_options* opt, const uint8_t* is_forced_output,
                              int is_compute_nonsomatic, sk_somatic_snv_genotype* out);
size_t sk_somatic_snv_tiers_scratch_bytes(int32_t n_loci);
int sk_somatic_snv_call_tiers_dev(const sk_pileup_batch* dev_normal_t1, const sk_pileup_batch* dev_tumor_t1,
                                  const sk_pileup_batch* dev_normal_t2, const sk_pileup_batch* dev_tumor_t2,
                                  const sk_somatic_snv_options* opt, const uint8_t* dev_is_forced_output,
                                  int is_compute_nonsomatic, sk_somatic_snv_genotype* dev_out, void* dev_scratch,
                                  void* hip_stream);

/* ---- row a8 for the two samples of a somatic run, chained into a12+a13 (sk_pileup_stream_* x 2 -> sk_somatic_snv_call_tiers) ----
 * strelka_pos_processor::process_pos_snp_somatic (L/applications/strelka/strelka_pos_processor.cpp:166-260) cleans the normal and
 * the tumor sample's pileup of a position twice (CleanPileupFilter(pi, false) and, with tier2 evidence, (pi, true),
 * PileupCleaner.cpp:28-66) and hands the four columns to position_somatic_snv_call (:213-219).  This stream takes one stage
 * window's reads of BOTH samples per push; the two samples share the finalised range [begin, end), each gets its raw tier1 /
 * tier2 columns and counters back as in sk_pileup_stream_push, the four cleaned columns stay on the device and go straight
 * into the somatic kernels (created with options: `genotype`, one record per position of the range).
 * with_read_pos: the tumor window also carries, parallel to its tier1 calls, each call's position in its read and the read's
 * length -- what updateSomaticScoringMetrics (starling_pos_processor_base.cpp:984-1000, called at :1360 for every basecall
 * when the somatic EVS models are loaded) feeds into snp_pos_info::readPositionRankSum and altAlleleReadPositionInfo for the
 * calls of sample != 0 that pass the tier1 filter: the caller rebuilds both accumulators from it for the few positions whose
 * record is written (position_somatic_snv_strand_grid_vcf.cpp:161-208). */
typedef struct sk_somatic_pileup_stream sk_somatic_pileup_stream;

typedef struct sk_somatic_pileup_window {
    sk_pileup_window normal, tumor;             /* same begin / end; clean_count = the CleanPileupFilter(pi,false) column's sizes */
    const uint32_t* normal_clean_tier2_count;   /* [n] sizes of the CleanPileupFilter(pi,true) columns */
    const uint32_t* tumor_clean_tier2_count;
    const uint32_t* tumor_tier1_read_pos;       /* [tumor.tier1_off[n]] read_pos | read_size << 16; NULL without with_read_pos */
    const sk_somatic_snv_genotype* genotype;    /* [n]; NULL when the stream does not genotype or n = 0 */
} sk_somatic_pileup_window;

/** genotype_opt: NULL = columns only.  Tier2 evidence as opt->use_tier2_evidence says. */
sk_somatic_pileup_stream* sk_somatic_pileup_stream_create(const sk_pileup_options* opt, const sk_somatic_snv_options* genotype_opt,
                                                          int with_read_pos);
void sk_somatic_pileup_stream_destroy(sk_somatic_pileup_stream* s);
int sk_somatic_pileup_stream_begin_region(sk_somatic_pileup_stream* s, const char* ref_seq, int32_t ref_offset, int32_t ref_len,
                                          int32_t report_begin, int32_t report_end, int32_t largest_total_indel_ref_span_per_read);
/** as sk_pileup_stream_push for both samples at once; is_forced_output[forced_len]: is_forced_output_pos() of positions from
 *  forced_begin on (0 elsewhere; NULL with forced_len 0); is_compute_nonsomatic: opt.is_somatic_callable(). */
int sk_somatic_pileup_stream_push(sk_somatic_pileup_stream* s, const sk_read_batch* host_normal_reads,
                                  const sk_read_batch* host_tumor_reads, int32_t largest_total_indel_ref_span_per_read,
                                  int32_t mask_begin, int32_t mask_len, const uint8_t* cand_snv_mask, int32_t final_to,
                                  int32_t forced_begin, int32_t forced_len, const uint8_t* is_forced_output,
                                  int is_compute_nonsomatic, sk_somatic_pileup_window* out);
/** the two halves, as sk_pileup_stream_push_begin / _finish */
int sk_somatic_pileup_stream_push_begin(sk_somatic_pileup_stream* s, const sk_read_batch* host_normal_reads,
                                        const sk_read_batch* host_tumor_reads, int32_t largest_total_indel_ref_span_per_read,
                                        int32_t mask_begin, int32_t mask_len, const uint8_t* cand_snv_mask, int32_t final_to,
                                        int32_t forced_begin, int32_t forced_len, const uint8_t* is_forced_output,
                                        int is_compute_nonsomatic);
int sk_somatic_pileup_stream_push_finish(sk_somatic_pileup_stream* s, sk_somatic_pileup_window* out);

/* ------------------------------------------------------------------------------------------------------------------
 * Hot path B (indels): per-read likelihood reductions over IndelSampleData::read_path_lnp
 * ---------------------------------------------------------------------------------------------------------------- */

enum { SK_MAX_ALT = 3, SK_MAX_INDEL_GT = 10 };
/** read_flags bits */
enum { SK_READ_TIER1 = 1, SK_READ_FWD = 2 };

typedef struct sk_indel_options { /* starling_base_options / starling_sample_options / Tier2Options */
    int32_t min_read_bp_flank;        /* 5 germline & tumor; normal sample in somatic mode: 1
                                         (L/starling_common/starling_base_shared.hh:108, strelka_shared.hh:159) */
    double random_base_match_prob;    /* 0.25 germline, 0.5 somatic (starling_base_shared.hh:177, strelka_shared.hh:74) */
    double tier2_random_base_match_prob; /* 0.25 (L/starling_common/Tier2Options.hh:50): used for every read of a tier2 pass */
    double read_confident_support_threshold; /* 0.51 (starling_base_shared.hh:245) */
    int32_t is_use_alt_indel;         /* 1 */
    int32_t fast_form;                /* 1 (default): sk_indel_grid_lhood* evaluate the 21 states' terms with two exp per read shared by
                                         the states and one log per state (3x faster); likelihoods agree with the reference's
                                         operation order to ~1e-13 relative (north_star's bar is 1e-5), the Q-scores, genotypes and
                                         tier decisions derived from them are the reference's in every case tested
                                         (tests/test_gpu_parity.py: 10^6 indels; the somatic end-to-end outputs byte for byte).
                                         0: every term in the reference's operation order, likelihoods bit-identical. */
} sk_indel_options;
void sk_indel_options_default(sk_indel_options* opt, int is_somatic);

/** indels -> reads (CSR): one row per (indel, sample); the per-read fields are ReadPathScores (L/starling_common/IndelData.hh:64-116) */
typedef struct sk_readscore_batch {
    int32_t n_indels;
    const int64_t* read_off;      /* [n_indels+1] */
    const float* ref_lnp;         /* ReadPathScores::ref */
    const float* indel_lnp;       /* ReadPathScores::indel */
    const float* alt_lnp;         /* max over ReadPathScores::alt_indel, NaN when the read has none; NULL = no alts */
    const uint16_t* non_ambig;    /* nonAmbiguousBasesInRead */
    const uint16_t* read_length;
    const uint8_t* read_flags;    /* SK_READ_TIER1 | SK_READ_FWD */
    const uint32_t* del_len;      /* [n_indels] IndelKey::delete_length() */
    const uint32_t* ins_len;      /* [n_indels] IndelKey::insert_length() */
    const uint8_t* is_breakpoint; /* [n_indels] or NULL */
} sk_readscore_batch;

/** a14 (likelihood half): the 21 somatic-grid states of one sample per indel: get_indel_digt_lhood
 *  (L/starling_common/starling_indel_call_pprob_digt.cpp:240-336) + get_indel_het_grid_lhood
 *  (L/applications/strelka/somatic_indel_grid.cpp:66-89).  out_lhood[n_indels][21], double. */
int sk_indel_grid_lhood(const sk_readscore_batch* host_batch, const sk_indel_options* opt, int is_include_tier2,
                        double* out_lhood);
int sk_indel_grid_lhood_dev(const sk_readscore_batch* dev_batch, const sk_indel_options* opt, int is_include_tier2,
                            double* dev_out_lhood, void* hip_stream);

typedef struct sk_somatic_indel_options { /* L/applications/strelka/strelka_shared.hh:126-151, workflow .ini overrides */
    double bindel_diploid_theta;      /* 1e-4 */
    double somatic_indel_rate;        /* sindelPrior 1e-6 */
    double shared_indel_error_factor; /* sindelNoiseFactor 2.2 */
    double indel_contam_tolerance;    /* 0.15 */
} sk_somatic_indel_options;
void sk_somatic_indel_options_default(sk_somatic_indel_options* opt);

typedef struct sk_somatic_indel_call { /* indel_result_set, somatic_result_set.hh:32-54 */
    double normal_lhood[SK_SOM_PRESTRAND];
    double tumor_lhood[SK_SOM_PRESTRAND];
    uint32_t max_gt;
    int32_t qphred;            /* QSI */
    int32_t from_ntype_qphred; /* QSI_NT */
    uint32_t ntype;
} sk_somatic_indel_call;

/** a14: replaces the likelihood + posterior part of somatic_indel_caller_grid::get_somatic_indel for one tier
 *  (L/applications/strelka/somatic_indel_grid.cpp:243-291) at L/applications/strelka/strelka_pos_processor.cpp:343-349.
 *  indel_to_ref_error_prob[n_indels]: tumorIndelSampleData.getErrorRates().indelToRefErrorProb (:273).
 *  The multi-indel-allele filter (:102-177) and the tier combination (:293-361) stay with the host adapter. */
int sk_somatic_indel_call_batch(const sk_readscore_batch* host_normal, const sk_readscore_batch* host_tumor,
                                const sk_indel_options* normal_opt, const sk_indel_options* tumor_opt,
                                const sk_somatic_indel_options* sopt, const double* indel_to_ref_error_prob,
                                int is_include_tier2, sk_somatic_indel_call* out);

/* The whole of get_somatic_indel: multi-indel-allele filter, both tiers, tier combination. */
enum { SK_MAX_ALT_ALLELES = 16 };
typedef struct sk_alt_allele { /* what is_indel_conflict (L/starling_common/indel_util.cpp:27-45) reads of an alternate IndelKey */
    int32_t begin_pos, end_pos; /* IndelKey::pos, IndelKey::right_pos() */
    int32_t is_mismatch;        /* IndelKey::isMismatch() */
} sk_alt_allele;

typedef struct sk_somatic_indel_batch {
    int32_t n_indels;
    sk_readscore_batch normal, tumor;  /* as for sk_indel_grid_lhood: alt_lnp = the read's best alternate score, NaN = none */
    /* the reads' ReadPathScores::alt_indel lists (<= 2 entries, front-packed) for the multi-indel-allele filter */
    const int32_t* normal_alt_key;     /* [normal reads][2] index into this indel's alternate-allele table, -1 = no entry */
    const float* normal_alt_lnp;       /* [normal reads][2] */
    const int32_t* tumor_alt_key;
    const float* tumor_alt_lnp;
    const int64_t* alt_off;            /* [n_indels+1]: alternate-allele table of each indel, <= SK_MAX_ALT_ALLELES entries */
    const sk_alt_allele* alt_alleles;  /* distinct IndelKeys */
    const double* indel_to_ref_error_prob; /* [n_indels] tumorIndelSampleData.getErrorRates().indelToRefErrorProb (:273) */
    const uint8_t* is_forced_output;   /* [n_indels] IndelData::isForcedOutput, or NULL */
} sk_somatic_indel_batch;

typedef struct sk_somatic_indel_genotype { /* somatic_indel_call, L/applications/strelka/somatic_result_set.hh:81-102 */
    uint8_t sindel_tier;             /* tier of rs.qphred (:303-310) */
    uint8_t sindel_from_ntype_tier;  /* tier of the other rs fields (:312-321) */
    uint8_t is_forced_output;
    uint8_t is_overlap;              /* indel_result_set::is_overlap of the chosen tier */
    uint32_t ntype;                  /* NTYPE::REF / HOM / HET / CONFLICT */
    uint32_t max_gt;
    int32_t qphred;                  /* QSI */
    int32_t from_ntype_qphred;       /* QSI_NT */
} sk_somatic_indel_genotype;

/** a14 complete: replaces somatic_indel_caller_grid::get_somatic_indel at
 *  L/applications/strelka/strelka_pos_processor.cpp:343-349 for a batch of candidate indels
 *  (L/applications/strelka/somatic_indel_grid.cpp:181-361): per tier the multi-indel-allele filter (is_multi_indel_allele
 *  :102-177 over get_sum_path_pprob / indel_lnp_to_pprob), the 2 x 21 likelihoods and the posterior; then the skip rules
 *  (:220-232, :239-254, :289-293), tier selection and NTYPE conflict (:295-360).  A record the reference returns without
 *  filling (not forced and a tier with qphred == 0) comes back all-zero.  With use_tier2_evidence == 0 the second tier is
 *  never evaluated (the reference then reads its never-written ntype at :323; it is taken as 0 here). */
int sk_somatic_indel_call_tiers(const sk_somatic_indel_batch* host_batch, const sk_indel_options* normal_opt,
                                const sk_indel_options* tumor_opt, const sk_somatic_indel_options* sopt,
                                int use_tier2_evidence, sk_somatic_indel_genotype* out);

/** allele groups -> reads (CSR): getVariantAlleleGroupGenotypeLhoodsForSample's input after the host adapter resolved
 *  read ids (L/starling_common/AlleleGroupGenotype.cpp:185-258; empty contrast group) */
typedef struct sk_allele_group_batch {
    int32_t n_groups;
    const int64_t* read_off;   /* [n_groups+1]; reads in ascending read-id order (std::set iteration order) */
    const uint8_t* n_alt;      /* [n_groups] non-reference allele count, 1..SK_MAX_ALT */
    const uint8_t* ploidy;     /* [n_groups] 1 or 2 */
    const uint32_t* del_len;   /* [n_groups][SK_MAX_ALT] */
    const uint32_t* ins_len;   /* [n_groups][SK_MAX_ALT] */
    const float* ref_lnp;      /* [reads][SK_MAX_ALT] ReadPathScores::ref of the read in each allele's map */
    const float* allele_lnp;   /* [reads][SK_MAX_ALT] ReadPathScores::indel; NaN = the read is not in that allele's map */
    const uint16_t* non_ambig; /* [reads] */
    const uint16_t* read_length;
    const uint8_t* read_flags;
} sk_allele_group_batch;

typedef struct sk_allele_group_call {
    double lhood[SK_MAX_INDEL_GT];           /* VcfGenotypeUtil::getGenotypeIndex order */
    uint32_t counts[2][SK_MAX_ALT + 2];      /* [fwd,rev][ref, alt.., non-confident] LocusSupportingReadStats */
    uint32_t n_genotypes;
    uint32_t n_reads_used;
} sk_allele_group_call;

/** a11: replaces getVariantAlleleGroupGenotypeLhoodsForSample at L/applications/starling/starling_pos_processor.cpp:1384-1386 */
int sk_allele_group_genotype_lhoods(const sk_allele_group_batch* host_batch, const sk_indel_options* opt,
                                    sk_allele_group_call* out);
int sk_allele_group_genotype_lhoods_dev(const sk_allele_group_batch* dev_batch, const sk_indel_options* opt,
                                        sk_allele_group_call* dev_out, void* hip_stream);

/** The same for allele groups of a multi-sample run: selectTopOrthogonalAllelesInAllSamples
 *  (L/starling_common/OrthogonalVariantAlleleCandidateGroupUtil.cpp:285-340) forms the union of every sample's top alleles, up to
 *  ploidy x sample_count = 8 alternate alleles (9 alleles, 45 diploid genotypes).  Same batch struct with every
 *  `[..][SK_MAX_ALT]` row SK_MAX_ALT_WIDE wide, n_alt 1..SK_MAX_ALT_WIDE; same arithmetic, same kernel (instantiated for the wider
 *  record).  The adapter takes this entry for groups of more than SK_MAX_ALT alternate alleles: no group goes back to the reference. */
enum { SK_MAX_ALT_WIDE = 8, SK_MAX_INDEL_GT_WIDE = 45 };
typedef struct sk_allele_group_call_wide {
    double lhood[SK_MAX_INDEL_GT_WIDE];
    uint32_t counts[2][SK_MAX_ALT_WIDE + 2];
    uint32_t n_genotypes;
    uint32_t n_reads_used;
} sk_allele_group_call_wide;
int sk_allele_group_genotype_lhoods_wide(const sk_allele_group_batch* host_batch, const sk_indel_options* opt,
                                         sk_allele_group_call_wide* out);
int sk_allele_group_genotype_lhoods_wide_dev(const sk_allele_group_batch* dev_batch, const sk_indel_options* opt,
                                             sk_allele_group_call_wide* dev_out, void* hip_stream);

/** ... and for runs of up to eight samples: rows SK_MAX_ALT_XWIDE wide, up to 16 alternate alleles (17 alleles, 153 diploid genotypes).
 *  The same arithmetic and the same kernel once more; a lane of the wave sums three genotypes.  Wider groups (more than eight samples whose
 *  top alleles at one locus are all distinct) are the adapter's one remaining refusal on this site. */
enum { SK_MAX_ALT_XWIDE = 16, SK_MAX_INDEL_GT_XWIDE = 153 };
typedef struct sk_allele_group_call_xwide {
    double lhood[SK_MAX_INDEL_GT_XWIDE];
    uint32_t counts[2][SK_MAX_ALT_XWIDE + 2];
    uint32_t n_genotypes;
    uint32_t n_reads_used;
} sk_allele_group_call_xwide;
int sk_allele_group_genotype_lhoods_xwide(const sk_allele_group_batch* host_batch, const sk_indel_options* opt,
                                          sk_allele_group_call_xwide* out);
int sk_allele_group_genotype_lhoods_xwide_dev(const sk_allele_group_batch* dev_batch, const sk_indel_options* opt,
                                              sk_allele_group_call_xwide* dev_out, void* hip_stream);

/* ------------------------------------------------------------------------------------------------------------------
 * SURVEY.md section 8f rank 4, the feed: BGZF inflation and BAM record decoding.
 *
 * Replaces what htslib (redist/htslib-1.7-6-g6d2bfb7: bgzf.c bgzf_read_block :641 / inflate_block :472, sam.c bam_read1) does
 * behind L/htsapi/bam_streamer.cpp:268 (sam_itr_next) and L/htsapi/bam_record.hh's accessors: a file image goes in, per-record
 * fields, BAM base codes (one per byte, as sk_read_input.read_code takes them), qualities and ALIGNPATH path segments come out.
 * Finding the blocks / records is a chain walk on the host (sk_bgzf_scan, sk_bam_scan_records); the bytes are inflated (DEFLATE,
 * RFC 1951; CRC-32 and ISIZE of every block checked) and decoded by kernels (csrc/bam_feed.hip).
 * normalizeAlignment (L/starling_common/normalizeAlignment.cpp:647-703), which the reference applies to every read as it comes
 * off the stream (starling_run.cpp / strelka_run.cpp via normalizeBamRecordAlignment :707-727), is the third kernel.
 * The index: sk_bai_query is hts_itr_query over a .bai image, sk_bam_region_filter the record test of hts_itr_next -- together with the
 * calls above, what sam_itr_queryi / sam_itr_next (bam_streamer.cpp:228, :268) do for a region.
 * Not built: CRAM.  (The gVCF writer's block logic is below; its formatting is host work and stays in the reference.)
 * ---------------------------------------------------------------------------------------------------------------- */

/** The BGZF blocks of a file image (or of any run of whole blocks): block_off[i] = start of block i, out_off[i] = where its inflated
 *  bytes belong in the concatenated stream; both arrays hold max_blocks + 1 entries, the last one written is the end.  Returns the
 *  number of blocks (may exceed max_blocks: call again with larger arrays), -1 on a malformed block header. */
int64_t sk_bgzf_scan(const uint8_t* data, int64_t n_bytes, int64_t* block_off, int64_t* out_off, int32_t max_blocks);
/** Inflate blocks [0, n_blocks) (offsets as sk_bgzf_scan gives them) into out[out_off[n_blocks]]. */
int sk_bgzf_inflate(const uint8_t* data, const int64_t* block_off, const int64_t* out_off, int32_t n_blocks, uint8_t* out);
/** The same for a caller that reads a region slice by slice: out receives `prefix_len` bytes the caller already holds (the record
 *  the previous slice ended in) followed by the inflated blocks, and the DEVICE copy of exactly those bytes is kept until the next
 *  feed call of this library -- sk_bam_decode_kept decodes from it, so the inflated stream crosses the bus once (down), not three
 *  times (down, up, and the decoded fields down).  n_blocks == 0 with a prefix (a slice that is only the carried record) is valid:
 *  out and the kept stream are the prefix. */
int sk_bgzf_inflate_prefixed(const uint8_t* data, const int64_t* block_off, const int64_t* out_off, int32_t n_blocks,
                             const uint8_t* prefix, int64_t prefix_len, uint8_t* out);
/** block_off relative to dev_data; dev_status[n_blocks]: 0 = ok (else the block is malformed, see csrc/bam_feed.hip) */
int sk_bgzf_inflate_dev(const uint8_t* dev_data, const int64_t* dev_block_off, const int64_t* dev_out_off, int32_t n_blocks,
                        uint8_t* dev_out, int32_t* dev_status, void* hip_stream);

typedef struct sk_bam_record { /* bam1_core_t as L/htsapi/bam_record.hh exposes it */
    int32_t ref_id;        /* target_id() */
    int32_t pos;           /* pos() - 1 (0-based, as stored) */
    int32_t mate_ref_id, mate_pos, template_size;
    int32_t l_seq;         /* read_size() */
    int32_t n_cigar;       /* n_cigar() */
    uint16_t flag;
    uint8_t mapq;          /* map_qual() */
    uint8_t is_fwd_strand; /* is_fwd_strand() */
    uint32_t pad;
} sk_bam_record;

/** Offset of the first record of an inflated BAM stream (after magic, header text and the reference table); -1 = not BAM. */
int64_t sk_bam_header_end(const uint8_t* stream, int64_t stream_len);
/** The records of stream[first, stream_len): rec_off[i] = offset of record i, read_off / path_off = CSR offsets of its bases and
 *  CIGAR operations (arrays of max_records + 1).  A record cut by the end of the stream is not counted.  Returns the number of
 *  records (may exceed max_records), -1 on a malformed record. */
int64_t sk_bam_scan_records(const uint8_t* stream, int64_t stream_len, int64_t first, int64_t* rec_off, int64_t* read_off,
                            int64_t* path_off, int32_t max_records);
/** Decode n_records records: fixed fields, read_code (BAM 4-bit codes, one per byte), read_qual, path (sk_path_seg, type = BAM op + 1). */
int sk_bam_decode(const uint8_t* stream, int64_t stream_len, const int64_t* rec_off, int32_t n_records, const int64_t* read_off,
                  const int64_t* path_off, sk_bam_record* rec, uint8_t* read_code, uint8_t* read_qual, sk_path_seg* path);
/** sk_bam_decode on the stream the last sk_bgzf_inflate_prefixed left on the device (`stream` = the caller's host copy of it, read
 *  only to validate the offsets; stream_len must be that call's prefix_len + inflated size). */
int sk_bam_decode_kept(const uint8_t* stream, int64_t stream_len, const int64_t* rec_off, int32_t n_records, const int64_t* read_off,
                       const int64_t* path_off, sk_bam_record* rec, uint8_t* read_code, uint8_t* read_qual, sk_path_seg* path);
int sk_bam_decode_dev(const uint8_t* dev_stream, const int64_t* dev_rec_off, int32_t n_records, const int64_t* dev_read_off,
                      const int64_t* dev_path_off, sk_bam_record* dev_rec, uint8_t* dev_read_code, uint8_t* dev_read_qual,
                      sk_path_seg* dev_path, void* hip_stream);

typedef struct sk_bai_chunk { /* hts_pair64_t: BGZF virtual offsets, (offset of the block in the file) << 16 | offset in the inflated block */
    uint64_t begin, end;
} sk_bai_chunk;
/** hts_itr_query (htslib hts.c:2066-2169, what sam_itr_queryi of L/htsapi/bam_streamer.cpp:228 runs) over the image of a .bai file
 *  (SAM spec 5.2): the chunks of the BAM that can hold records overlapping [begin, end) of reference ref_id -- the bins of reg2bins
 *  (:1939-1955), bounded below by the linear index (update_loff :1379-1408) and above by the next existing bin, sorted and merged as
 *  htslib leaves them.  Records are read from chunk.begin on, one after the other, while their start is before chunk.end.
 *  Returns the number of chunks (may exceed max_chunks: call again), -1 = malformed index, -2 = ref_id not in the index. */
int32_t sk_bai_query(const uint8_t* bai, int64_t bai_len, int32_t ref_id, int32_t begin, int32_t end, sk_bai_chunk* out, int32_t max_chunks);
/** The record test of hts_itr_next (hts.c:2621-2631) on decoded records in the order the chunks give them: keep[i] = 1 for a record
 *  of ref_id that overlaps [begin, end) (its end: bam_endpos, sam.c:359-365).  Returns how many records the iterator reads before it
 *  finishes -- at the first record of another reference or starting at or after `end`; keep is 0 from there on. */
int32_t sk_bam_region_filter(const sk_bam_record* rec, const int64_t* path_off, const sk_path_seg* path, int32_t n_records, int32_t ref_id,
                             int32_t begin, int32_t end, uint8_t* keep);

/** normalizeAlignment for n_reads alignments against one reference segment, IN PLACE: indels inside the alignment are collapsed
 *  and left-shifted, edge indels normalised, the path cleaned (apath_cleaner).  Read r: bases read_code[read_off[r]..), alignment
 *  pos[r] + path[path_off[r] .. path_off[r] + n_seg[r]) (as sk_bam_decode leaves them; n_seg only ever shrinks), changed[r] = the
 *  function's return value.  ref_seq: the contig segment (positions outside read as 'N', reference_contig_segment::get_base). */
int sk_normalize_alignments(const char* ref_seq, int32_t ref_offset, int32_t ref_len, int32_t n_reads, const int64_t* read_off,
                            const uint8_t* read_code, const int64_t* path_off, int32_t* n_seg, sk_path_seg* path, int32_t* pos,
                            uint8_t* changed);
int sk_normalize_alignments_dev(const char* dev_ref_seq, int32_t ref_offset, int32_t ref_len, int32_t n_reads, const int64_t* dev_read_off,
                                const uint8_t* dev_read_code, const int64_t* dev_path_off, int32_t* dev_n_seg, sk_path_seg* dev_path,
                                int32_t* dev_pos, uint8_t* dev_changed, void* hip_stream);

/* ------------------------------------------------------------------------------------------------------------------
 * SURVEY.md section 8f rank 4, the output side: the non-variant blocks of the gVCF.
 *
 * gvcf_writer::queue_site_record (L/applications/starling/gvcf_writer.cpp:278-302) asks, site after site and per sample, whether the
 * site can join the open block (gvcf_block_site_record::testCanSiteJoinSampleBlock, gvcf_block_site_record.cpp:160-182: same
 * filters, genotype, ploidy and coverage state, depth and GQX within block_percent_tol / block_abs_tol of the block's minimum) and
 * joins it or starts a new block (joinSiteToSampleBlock :126-157).  Here a run of sites of one sample goes in, and which sites start
 * a block, which continue one and which are records of their own comes out, with what write_site_record (:749-806) prints of every
 * block.  Formatting the records stays host work.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct sk_gvcf_site { /* what the block logic reads of a GermlineSiteLocusInfo, for one sample */
    int32_t pos;
    uint8_t is_compressible;   /* gvcf_compressor::is_site_compressible (gvcf_compressor.cpp:41-71): the caller's */
    uint8_t is_gqx;            /* GermlineSiteLocusInfo::is_gqx(sample) */
    uint8_t ploidy;            /* LocusSampleInfo::getPloidy().getPloidy() */
    uint8_t flush_before;      /* the writer flushed its blocks between the previous site and this one (writeAllNonVariantBlockRecords) */
    uint32_t gt;               /* max_gt() as VcfGenotype::operator== compares it: ploidy << 24 | phased << 16 | allele0 << 8 | allele1 */
    uint32_t locus_filters, sample_filters; /* GermlineFilterKeeper bits */
    int32_t gqx;
    uint32_t used_basecalls, unused_basecalls; /* GermlineSiteSampleInfo */
} sk_gvcf_site;
typedef struct sk_gvcf_block { /* gvcf_block_site_record as write_site_record reads it */
    int32_t pos, count;        /* END = pos + count */
    int32_t is_gqx_defined;    /* isBlockGqxDefined */
    int32_t gqx_min;           /* block_gqx.min() */
    int32_t dpu_min;           /* MIN_DP = block_dpu.min() */
    int32_t pad;
    double dpu_mean, dpf_mean; /* DP = round(block_dpu.mean()), DPF = round(block_dpf.mean()): stream_stat's running mean */
} sk_gvcf_block;
/** kind[i]: 0 = site i continues the block of the site before it, 1 = it starts a block (blocks[i] describes the block), 2 = it is
 *  not compressible and is written as a record of its own.  blocks: n_sites entries, written where kind is 1. */
/** sk_gvcf_site_summary of every locus of a device-resident batch (the cleaned columns as sk_site_digt_call_fused_dev takes them) from
 *  the genotype records that call left: what the pileup stream appends to a window (sk_pileup_window.site_summary). */
int sk_gvcf_site_summaries_dev(const sk_pileup_batch* dev_batch, const sk_digt_call* dev_genotypes, sk_gvcf_site_summary* dev_out, void* hip_stream);
/** sk_gvcf_run of every site of a window from its summaries and column sizes (clean_off / raw_off: CSR offsets of the cleaned and the
 *  raw tier1 columns; dev_pod_scratch: 256 + 17 bytes per site): what the pileup stream appends to a window when it has block options. */
int sk_gvcf_plain_runs_dev(const sk_gvcf_site_summary* dev_summary, const int64_t* dev_clean_off, const int64_t* dev_raw_off, const uint32_t* dev_mapq_count,
                           const sk_gvcf_block_options* opt, int32_t n, void* dev_pod_scratch, sk_gvcf_run* dev_runs, void* hip_stream);
/** ... for host arrays (one upload, the launches, one copy back; clean_count / raw_count: the columns' sizes): tests */
int sk_gvcf_plain_runs(const sk_gvcf_site_summary* summary, const uint32_t* clean_count, const uint32_t* raw_count, const uint32_t* mapq_count,
                       const sk_gvcf_block_options* opt, int32_t n, sk_gvcf_run* runs);
/** the same for host arrays (one upload, one launch, one copy back): tests, and callers without a stream */
int sk_gvcf_site_summaries(const sk_pileup_batch* host_batch, const sk_digt_call* genotypes, sk_gvcf_site_summary* out);
int sk_gvcf_block_sites(const sk_gvcf_site* sites, int32_t n_sites, uint32_t block_percent_tol, uint32_t block_abs_tol, uint8_t* kind,
                        sk_gvcf_block* blocks);
int sk_gvcf_block_sites_dev(const sk_gvcf_site* dev_sites, int32_t n_sites, uint32_t block_percent_tol, uint32_t block_abs_tol,
                            uint8_t* dev_kind, sk_gvcf_block* dev_blocks, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* STRELKA_AMD_H */
