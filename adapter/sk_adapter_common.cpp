// sk_adapter_common.cpp -- process set-up and the shadow of the reference's original stage geometry.
#include "sk_adapter_access.hh"

#include "blt_util/log.hh"
#include "starling_common/starling_read.hh"

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>

namespace sk_adapter
{

static unsigned env_unsigned(const char* name, const unsigned def)
{
    const char* v(std::getenv(name));
    if (v == nullptr || *v == 0) return def;
    return static_cast<unsigned>(std::strtoul(v, nullptr, 10));
}

// READ_BUFFER deferral in positions.  (What must NOT move with the stage is the clearing of the active-region read buffer, a ring
// of 1000 positions: clear_active_region_read_buffer_undeferred below.  Before that was kept at its own distance, windows of 500
// and of >= 1500 positions gave different output on the reference's demo data.)
unsigned read_buffer_defer()
{
    // (a caller process shares its GPU with the other segment processes of the node: what a window costs there is the number of
    // times the process waits for the device, not the work in it; what a LARGE window costs is the host's cache -- everything between
    // the head and the deferred stages is kept alive; profiles/r03_v2..v5_gpu_sharing*.txt)
    static const unsigned w(env_unsigned("STRELKA_AMD_READ_WINDOW", 8192));
    return w;
}

static int g_postAlignDefer(-1);

unsigned post_align_defer()
{
    if (g_postAlignDefer < 0) g_postAlignDefer = static_cast<int>(env_unsigned("STRELKA_AMD_SITE_WINDOW", 4096));
    return static_cast<unsigned>(g_postAlignDefer);
}

unsigned post_align_defer(const starling_base_options& opt)
{
    if (g_postAlignDefer < 0)
    {
        const char* v(std::getenv("STRELKA_AMD_SITE_WINDOW"));
        if (v != nullptr && *v != 0) g_postAlignDefer = static_cast<int>(std::strtoul(v, nullptr, 10));
        else if (! pileup_genotypes_with_stream(opt)) g_postAlignDefer = 4096;
        else
        {
            // the genotypes come with the pileup stream's windows: POST_ALIGN is held back only by what hides a window's time on the
            // device -- ~0.45 ms alone, ~400 head positions of read intake at 40x; two to three times that when sixteen callers share
            // the device -- behind the stage machine's own work (the push is begun when READ_BUFFER reaches the window and finished
            // when POST_ALIGN does: sk_adapter_pileup.cpp, pileup_complete_push).  16 callers, interleaved: 256 / 512 / 1024 / 2048
            // positions 9.67 / 9.31 / 9.12 / 9.01 s (profiles/r06_v54).
            const char* a(std::getenv("STRELKA_AMD_PUSH_ASYNC"));
            g_postAlignDefer = (a == nullptr || *a == 0 || std::atoi(a) != 0) ? 1024 : 0;
        }
    }
    return static_cast<unsigned>(g_postAlignDefer);
}

namespace
{

/// sk_init as it has always been made here: the device of this segment process, strict unless told otherwise
struct InitOutcome
{
    int rc = 0;
    bool isInexactLibm = false;
    std::string what, error;
};

InitOutcome run_sk_init()
{
    InitOutcome o;
    // A workflow runs one caller process per core (PY/strelkaSharedOptions.py:153-161) and the device runs eight processes' work side by
    // side: unless told otherwise ($STRELKA_AMD_BROKER=0) a caller process is a client of its device's broker (one GPU context however
    // many callers, started by the first of them: include/strelka_amd.h "the broker").
    {
        const char* v(std::getenv("STRELKA_AMD_BROKER"));
        if (v == nullptr || *v == 0) (void)sk_broker_enable(1);
    }
    // segment process -> device: pyflow starts one process per genome segment; the launcher (or the workflow's task
    // wrapper) exports STRELKA_AMD_DEVICE = segment index mod number of GPUs.  Many processes may share a device.
    // (a launcher may count more devices than this node has: the index is taken modulo the devices present)
    const int deviceCount(std::max(1, sk_device_count()));
    const int device(static_cast<int>(env_unsigned("STRELKA_AMD_DEVICE", 0)) % deviceCount);
    // byte-identical VCFs need the kernels' restated libm routines to be the host's: strict by default
    if (env_unsigned("STRELKA_AMD_ALLOW_INEXACT_LIBM", 0) == 0)
    {
        o.what = "sk_init_strict";
        o.rc = sk_init_strict(device);
    }
    else
    {
        o.what = "sk_init";
        o.rc = sk_init(device);
        if (o.rc == 0) o.isInexactLibm = (sk_libm_restated() != 1);
    }
    if (o.rc != 0) o.error = sk_last_error();
    return o;
}

/// The GPU runtime's start-up (0.2-0.4 s: context, first allocation, tables) on a thread of its own from the moment the program is
/// loaded, beside the caller's own start-up (options, the reference segment, the alignment files' indices, the scoring models):
/// init() -- the first hook to run -- waits for it.  $STRELKA_AMD_EARLY_INIT=0: sk_init where it always was.
struct EarlyInit
{
    std::thread worker;
    InitOutcome outcome;
    bool isStarted = false;
    EarlyInit()
    {
        if (env_unsigned("STRELKA_AMD_EARLY_INIT", 1) == 0) return;
        isStarted = true;
        // the library exports this variable (if unset) before its first runtime call; done HERE, on the main thread and before the worker
        // exists, so that no setenv can run beside another thread's getenv (static constructors, option parsing, htslib)
        (void)::setenv("GPU_MAX_HW_QUEUES", "1", 0);
        worker = std::thread([this]() { outcome = run_sk_init(); });
    }
    ~EarlyInit()
    {
        if (worker.joinable()) worker.join();
    }
};
EarlyInit g_earlyInit;

}

/// The driver runs eight compute processes side by side on a device and time-slices the processes beyond that ("How
/// many caller processes per GPU"): a call into the C-ABI then waits for its process's slice, and the run is slower than with eight.
/// Every caller process of a device holds one of eight advisory file locks for its lifetime (released by the kernel when it exits,
/// however it exits); the ninth finds none free and says so once.  Nothing else depends on the locks.  $STRELKA_AMD_PROCESS_SLOTS: the
/// number of slots (0: no check), $STRELKA_AMD_SLOT_DIR: where the lock files live (default /tmp).
void note_process_slot()
{
    const unsigned slots(env_unsigned("STRELKA_AMD_PROCESS_SLOTS", 8));
    if (slots == 0) return;
    const char* dir(std::getenv("STRELKA_AMD_SLOT_DIR"));
    const int deviceCount(std::max(1, sk_device_count()));
    const int device(static_cast<int>(env_unsigned("STRELKA_AMD_DEVICE", 0)) % deviceCount);
    bool isAnyOpened(false);
    for (unsigned i(0); i < slots; ++i)
    {
        const std::string path(std::string((dir && *dir) ? dir : "/tmp") + "/strelka_amd_device" + std::to_string(device) + "_slot" + std::to_string(i) + ".lock");
        const int fd(::open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0666));
        if (fd < 0) continue;
        isAnyOpened = true;
        if (::flock(fd, LOCK_EX | LOCK_NB) == 0) return; // (held until the process exits: the descriptor is left open on purpose)
        ::close(fd);
    }
    if (! isAnyOpened) return; // (no place for lock files: no check)
    log_os << "WARNING: strelka_amd: more than " << slots << " caller processes on device " << device
           << ": the driver runs eight side by side and time-slices the rest -- expect every process of this device to slow down "
              "(INTEGRATION.md, \"How many caller processes per GPU\")\n";
}

void init()
{
    static bool done(false);
    if (done) return;
    AccumTimer initTimer(state().tInit);
    InitOutcome o;
    if (g_earlyInit.isStarted)
    {
        if (g_earlyInit.worker.joinable()) g_earlyInit.worker.join();
        o = g_earlyInit.outcome;
        if (o.rc == 0) o = run_sk_init(); // (returns at once: the library is up; the current device is set for THIS thread)
    }
    else
    {
        o = run_sk_init();
    }
    if (o.rc != 0) throw blt_exception((std::string("strelka_amd: ") + o.what + ": " + o.error).c_str());
    if (o.isInexactLibm)
    {
        log_os << "WARNING: strelka_amd runs with the device math library; outputs may differ from the reference in the last digit\n";
    }
    if (sk_broker_client() == 0) note_process_slot(); // (clients of the broker share ONE context: the eight-process ceiling is not theirs)
    done = true;
}

State* g_state(nullptr);

State& make_state()
{
    // what went through the C-ABI, on stderr at exit with STRELKA_AMD_VERBOSE=1 (the end-to-end tests read it to make sure
    // the identical VCF was produced by the routed path and not by an idle adapter)
    struct Reporter
    {
        State s;
        ~Reporter()
        {
            if (env_unsigned("STRELKA_AMD_VERBOSE", 0) == 0) return;
            std::cerr << "strelka_amd adapter: realign_jobs=" << s.realignBatches << " realign_reads=" << s.realignReads << " realign_job_reads=" << s.realignJobReads
                      << " site_batches=" << s.siteBatches << " site_loci=" << s.siteLoci << " site_recomputed=" << s.siteRecomputed << " site_recompute_calls=" << s.siteRecomputeCalls
                      << " indel_groups=" << s.indelGroups << " indel_groups_wide=" << s.indelGroupsWide << " indel_groups_xwide=" << s.indelGroupsXWide << " haplotypes=" << s.haplotypes << " haplotype_batches=" << s.haplotypeBatches << " read_window=" << read_buffer_defer()
                      << " site_window=" << post_align_defer() << " enum_device_reads=" << s.realignDeviceEnumerated
                      << " enum_host_instead=" << s.realignHostEnumerated;
            {
                // the device jobs of this process: run as one fixed sequence with one wait / of those run again the staged way / staged
                int64_t oneWait(0), redone(0), staged(0);
                sk_realign_device_job_counts(&oneWait, &redone, &staged);
                std::cerr << " enum_jobs_one_wait=" << oneWait << " enum_jobs_redone=" << redone << " enum_jobs_staged=" << staged
                          << " enum_jobs_host=" << s.realignHostJobs << " realign_ref_window_misses=" << s.realignRefWindowMisses << "\n";
            }
            std::cerr << "strelka_amd adapter pileup: pushes=" << s.pileupBatches << " reads=" << s.pileupReads << " loci=" << s.pileupLoci
                      << " genotyping=" << (s.pileup.isGenotyping ? 1 : 0) << " evs_words_copied=" << s.pileup.evsWordsCopied
                      << " evs_words_left=" << s.pileup.evsWordsLeft << "\n";
            std::cerr << "strelka_amd adapter seconds: realign_hook=" << s.tRealignHook << " realign_abi=" << s.tRealignAbi
                      << " site_hook=" << s.tSiteHook << " site_abi=" << s.tSiteAbi << " pileup_hook=" << s.tPileupHook
                      << " pileup_abi=" << s.tPileupAbi << " pileup_gather=" << s.tPileupGather << " pileup_assign=" << s.tPileupAssign
                      << " pileup_chunk=" << s.tPileupChunk << " init=" << s.tInit << " indel_abi=" << s.tIndelAbi
                      << " haplotype_abi=" << s.tHaplotypeAbi << "\n";
        }
    };
    static Reporter r;
    g_state = &(r.s);
    return r.s;
}

// ---------------------------------------------------------------------------------------------------------------------

void GeometryShadow::reset(const unsigned sampleCount)
{
    isFirstPosSet = false;
    isAnyActiveRegionCleared = false;
    maxPos = 0;
    lastReadBufferPos = 0;
    isAnyReadBufferPos = false;
    segments.clear();
    clearedToPos = 0;
    isAnyCleared = false;
    bufferedReadPos.assign(sampleCount, PosHeap());
}

void GeometryShadow::onSetHeadPos(const pos_t pos, const unsigned readBufferShift, const unsigned indelSpan)
{
    curReadBufferShift = readBufferShift;
    curIndelSpan = indelSpan;

    // stage_manager::handle_new_pos_value (L/blt_util/stage_manager.cpp:160-188): on the first call _max_pos is set
    // before the stages run, afterwards the stages of a head advance run while _max_pos still holds the old value
    pos_t maxPosDuring;
    if (!isFirstPosSet)
    {
        maxPos = pos;
        maxPosDuring = pos;
        isFirstPosSet = true;
    }
    else
    {
        if (pos <= maxPos) return;
        maxPosDuring = maxPos;
        maxPos = pos;
    }

    const pos_t postAlignShift(static_cast<pos_t>(readBufferShift + indelSpan));
    const pos_t rb(pos - static_cast<pos_t>(readBufferShift));
    if ((!isAnyReadBufferPos) || rb > lastReadBufferPos)
    {
        Params p;
        p.upto = rb;
        p.rangeMinOffset = std::max(0, static_cast<pos_t>(indelSpan) - 1);
        p.rangeMaxOffset = std::max(0, static_cast<pos_t>(readBufferShift) - 1);
        p.validThreshold = maxPosDuring - postAlignShift;
        if ((!segments.empty()) && segments.back().rangeMinOffset == p.rangeMinOffset &&
            segments.back().rangeMaxOffset == p.rangeMaxOffset && segments.back().validThreshold == p.validThreshold)
        {
            segments.back().upto = rb;
        }
        else
        {
            segments.push_back(p);
        }
        lastReadBufferPos = rb;
        isAnyReadBufferPos = true;
    }

    // CLEAR_READ_BUFFER sits at the POST_ALIGN distance (starling_pos_processor_base.cpp:198)
    const pos_t cleared(pos - postAlignShift);
    if ((!isAnyCleared) || cleared > clearedToPos)
    {
        clearedToPos = cleared;
        isAnyCleared = true;
        for (auto& held : bufferedReadPos)
        {
            while ((! held.empty()) && held.top() <= clearedToPos) held.pop();
        }
    }
}

GeometryShadow::Params GeometryShadow::query(const pos_t pos) const
{
    // (segments are in ascending order of `upto`; a stage window asks for every read it holds)
    const auto it(std::lower_bound(segments.begin(), segments.end(), pos, [](const Params& p, const pos_t v) { return p.upto < v; }));
    if (it != segments.end()) return *it;
    // not reached by a head advance: the position is handled by the final flush (stage_manager::reset), with the
    // geometry and _max_pos in force at that time
    Params p;
    p.upto = pos;
    p.rangeMinOffset = std::max(0, static_cast<pos_t>(curIndelSpan) - 1);
    p.rangeMaxOffset = std::max(0, static_cast<pos_t>(curReadBufferShift) - 1);
    p.validThreshold = maxPos - static_cast<pos_t>(curReadBufferShift + curIndelSpan);
    return p;
}

void on_reset_region(starling_pos_processor_base& pp)
{
    init();
    // Said before the first read, not by a throw in the middle of a genome: a realignment job and the pileup streams hold per-sample
    // fields for SK_MAX_SAMPLES samples, and an allele group of a multi-sample run can hold ploidy x samples alternate alleles
    // (selectTopOrthogonalAllelesInAllSamples, OrthogonalVariantAlleleCandidateGroupUtil.cpp:285-340) -- the widest record takes
    // SK_MAX_ALT_XWIDE = 2 x SK_MAX_SAMPLES, so with this bound no group can outgrow it.
    static_assert(SK_MAX_ALT_XWIDE >= 2 * SK_MAX_SAMPLES, "the widest allele-group record holds ploidy x samples alternate alleles");
    if (Access::sampleCount(pp) > SK_MAX_SAMPLES)
    {
        std::ostringstream oss;
        oss << "strelka_amd adapter: " << Access::sampleCount(pp) << " samples in one run; this path takes at most SK_MAX_SAMPLES = "
            << SK_MAX_SAMPLES << " (per-sample fields of a realignment job, allele groups of up to " << SK_MAX_ALT_XWIDE
            << " alternate alleles)";
        throw blt_exception(oss.str().c_str());
    }
    State& s(state());
    s.geometry.reset(Access::sampleCount(pp));
    s.isAnyRealigned = false;
    s.realignedTo = 0;
    s.sites.clear();
    s.somaticSites.clear();
    pileup_reset_region(pp);
    gvcf_reset_region();
}

void on_set_head_pos(starling_pos_processor_base& /*pp*/, const pos_t pos, const unsigned readBufferShift, const unsigned indelSpan)
{
    GeometryShadow& g(state().geometry);
    g.onSetHeadPos(pos, readBufferShift, indelSpan);
    // segments behind the windows already realigned are no longer needed (the deferred READ_BUFFER stage has yet to visit
    // everything from realignedTo on -- during this very head advance, too)
    State& s(state());
    if (s.isAnyRealigned)
    {
        while (g.segments.size() > 1 && g.segments.front().upto < s.realignedTo) g.segments.pop_front();
    }
}

void clear_active_region_read_buffer_undeferred(starling_pos_processor_base& pp, const pos_t headStagePos, const unsigned readBufferShift,
                                                const pos_t minPos)
{
    // stage_manager::process_pos (L/blt_util/stage_manager.cpp:283-316): READ_BUFFER runs for headStagePos - shift unless that is
    // before the first position; after a revision of the stage distances a stage resumes where it had got to (never twice)
    GeometryShadow& g(state().geometry);
    const pos_t p(headStagePos - static_cast<pos_t>(readBufferShift));
    if (p < minPos) return;
    if (g.isAnyActiveRegionCleared && p <= g.activeRegionClearedTo) return;
    Access::clearActiveRegionReadBuffer(pp, p);
    g.activeRegionClearedTo = p;
    g.isAnyActiveRegionCleared = true;
}

unsigned buffered_read_count(const starling_pos_processor_base& /*pp*/, const unsigned sampleIndex, const unsigned /*actualCount*/)
{
    return static_cast<unsigned>(state().geometry.bufferedReadPos[sampleIndex].size());
}

void on_read_inserted(starling_pos_processor_base& pp, const unsigned sampleIndex, const starling_read& sread)
{
    if (sread.isSpliced())
    {
        throw blt_exception("strelka_amd adapter: spliced (RNA) reads are not supported on this path");
    }
    state().geometry.bufferedReadPos[sampleIndex].push(sread.get_full_segment().buffer_pos);
    // The device pileup holds a read's per-base state in LDS: 1024 bases (SK_PILEUP_MAX_READ_LEN); the reference takes reads up to
    // STRELKA_MAX_READ_SIZE = 25000.  Said here, when the read arrives, with the way out -- not as a failed push a window later.
    if (sread.get_full_segment().read_size() > SK_PILEUP_MAX_READ_LEN && pileup_enabled(pp))
    {
        std::ostringstream oss;
        oss << "strelka_amd adapter: read of " << sread.get_full_segment().read_size() << " bases (sample " << sampleIndex << ", position "
            << (sread.get_full_segment().buffer_pos + 1) << "): the device pileup (site 9) takes reads of at most " << SK_PILEUP_MAX_READ_LEN
            << " bases; rerun with STRELKA_AMD_PILEUP=0 (the reference's own pileup_read_segment, every other site still routed)";
        throw blt_exception(oss.str().c_str());
    }
    pileup_note_read(sampleIndex, sread.get_full_segment().buffer_pos);
}

}
