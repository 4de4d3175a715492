#!/usr/bin/env python3
"""Write hooked copies of a few reference translation units into oracle/_ref/adapter_src/ (git-ignored, never committed).

The reference has no plug-in seam for this path: the hot-path calls are ordinary C++ calls inside
starling_pos_processor_base / starling_pos_processor / strelka_pos_processor.  A maintainer integrating
libstrelka_amd.so would edit those call sites; this script makes exactly those edits, mechanically, on copies made at
build time from the sources where they lie under $REFERENCE -- each edit is an anchored substitution that must match
exactly once, so a reference version with a different call site fails the build instead of silently changing nothing.
Every inserted line calls a function declared in adapter/sk_adapter.hh; all logic lives in adapter/*.cpp.

usage: apply_hooks.py <reference root> <output dir>
       apply_hooks.py --doc [INTEGRATION.md]     print the table of edits made from HOOKS (markdown), or rewrite the section of the
                                                 named file between the `hooks:begin` / `hooks:end` markers with it -- the document's
                                                 table is GENERATED from this list (tests/test_abi.py compares them), it cannot drift
"""
import os
import re
import sys

L = "src/c++/lib/"

# (file, [(description, anchor regex, replacement)])
HOOKS = [
    (L + "starling_common/starling_pos_processor_base.hh", [
        ("friend + forward declaration",
         r'struct starling_pos_processor_base : public pos_processor_base, private boost::noncopyable\n\{\n',
         '#include "sk_adapter_fwd.hh"\n\\g<0>    friend struct sk_adapter::Access;\n'),
    ]),
    (L + "starling_common/pos_basecall_buffer.hh", [
        ("friend + forward declaration",
         r'struct pos_basecall_buffer\n\{\n',
         '#include "sk_adapter_fwd.hh"\n\\g<0>    friend struct sk_adapter::Access;\n'),
    ]),
    (L + "starling_common/starling_pos_processor_base.cpp", [
        ("include", r'#include "starling_read_align.hh"\n', '\\g<0>#include "sk_adapter.hh"\n'),
        # stage geometry: READ_BUFFER and POST_ALIGN pushed further behind HEAD by the batching windows
        ("READ_BUFFER stage distance", r'\+HAPLOTYPING_PADDING\);', '+HAPLOTYPING_PADDING+sk_adapter::read_buffer_defer());'),
        ("POST_ALIGN stage distance",
         r'sdata\.add_stage\(POST_ALIGN,READ_BUFFER,largest_total_indel_ref_span_per_read\);',
         'sdata.add_stage(POST_ALIGN,READ_BUFFER,largest_total_indel_ref_span_per_read+sk_adapter::post_align_defer(opt));'),
        # empty-site genotypes of the constructor (site 3 with zero-depth loci)
        ("empty-site precompute",
         r'_dopt\.pdcaller\(\)\.position_snp_call_pprob_digt\(_opt,good_epi,\s*\*_empty_dgt\[b\],\s*_opt\.is_all_sites\(\)\);',
         'sk_adapter::empty_site_genotype(*this, b, *_empty_dgt[b]);'),
        ("region reset",
         r'(_stagemanPtr\.reset\(new stage_manager\(STAGE::get_stage_data\(STARLING_INIT_LARGEST_READ_SIZE, get_largest_total_indel_ref_span_per_read\(\), _opt, _dopt\), pr, \*this\)\);\n)',
         '\\1    sk_adapter::on_reset_region(*this);\n'),
        # read-buffer occupancy as the reference would see it (its CLEAR_READ_BUFFER stage runs earlier than ours)
        ("read-buffer capacity test", r'if \(rbuff\.size\(\) >= _opt\.maxBufferedReads\)',
         'if (sk_adapter::buffered_read_count(*this, sampleIndex, rbuff.size()) >= _opt.maxBufferedReads)'),
        ("read inserted",
         r'(        assert\(nullptr!=sread_ptr\);\n)',
         '\\1        sk_adapter::on_read_inserted(*this, sampleIndex, *sread_ptr);\n'),
        # the active-region read buffer is a ring of 1000 positions (ActiveRegionReadBuffer.hh:61) that the reference clears as its
        # READ_BUFFER stage passes; deferred along with that stage the clear would hit slots the head has already refilled
        # (found as wrong output for read windows of 500 and of >= 1500 positions), so it stays at the undeferred distance:
        # right after the HEAD stage of position pos, for position pos - (original READ_BUFFER distance), as stage_manager orders it
        ("active-region read buffer clear, undeferred",
         r'(            _getActiveRegionDetector\(\)\.updateEndPosition\(pos\);\n)',
         '\\1            sk_adapter::clear_active_region_read_buffer_undeferred(*this, pos, get_read_buffer_size(get_largest_read_size(), '
         'get_largest_total_indel_ref_span_per_read())+HAPLOTYPING_PADDING, _stagemanPtr->min_pos());\n'),
        ("active-region read buffer clear, deferred copy removed",
         r'(        write_reads\(pos\);\n\n        if \(is_active_region_detector_enabled\(\)\)\n        \{\n)            _getActiveRegionDetector\(\)\.clearReadBuffer\(pos\);\n',
         '\\1            /* strelka_amd: cleared at the undeferred distance, see the HEAD stage */\n'),
        # site 1
        ("align_pos",
         r'(starling_pos_processor_base::\nalign_pos\(const pos_t pos\)\n\{\n)',
         '\\1    if (sk_adapter::align_pos(*this, pos)) return;\n'),
        # site 9
        ("pileup_pos_reads",
         r'(starling_pos_processor_base::\npileup_pos_reads\(const pos_t pos\)\n\{\n)',
         '\\1    if (sk_adapter::pileup_pos_reads(*this, pos)) return;\n'),
        ("reset: the final flush",
         r'(starling_pos_processor_base::\nreset\(\)\n\{\n    if \(_stagemanPtr\)\n    \{\n)        _stagemanPtr->reset\(\);\n',
         '\\1        sk_adapter::on_flush_begin(*this);\n        _stagemanPtr->reset();\n        sk_adapter::on_flush_end(*this);\n'),
        ("set_head_pos",
         r'(starling_pos_processor_base::\nset_head_pos\(const pos_t pos\)\n\{\n)',
         '\\1    sk_adapter::on_set_head_pos(*this, pos, get_read_buffer_size(get_largest_read_size(), '
         'get_largest_total_indel_ref_span_per_read())+HAPLOTYPING_PADDING, get_largest_total_indel_ref_span_per_read());\n'),
        # the somatic caller's per-sample depth statistics from the stream's column sizes (its cleaned pileups are built on demand)
        ("process_pos_sample_stats",
         r'    _pileupCleaner\.CleanPileupFilter\(pi,is_include_tier2,sif\.cleanedPileup\);\n\n'
         r'(    const unsigned n_spandel\(pi\.spanningDeletionReadCount\);\n    const unsigned n_submapped\(pi\.submappedReadCount\);\n\n'
         r'    if \(pi\.get_ref_base\(\) != \'N\'\)\n    \{\n)'
         r'        sif\.localRegionStatsCollection\.insert\(pos, sif\.cleanedPileup\.usedBasecallCount\(\), sif\.cleanedPileup\.unusedBasecallCount\(\),n_spandel,n_submapped\);\n',
         '    unsigned skUsed(0), skUnused(0);\n'
         '    if (! sk_adapter::sample_stats_counts(*this, pos, sample_no, skUsed, skUnused))\n    {\n'
         '        _pileupCleaner.CleanPileupFilter(pi,is_include_tier2,sif.cleanedPileup);\n'
         '        skUsed = sif.cleanedPileup.usedBasecallCount();\n        skUnused = sif.cleanedPileup.unusedBasecallCount();\n    }\n\n'
         '\\1        sif.localRegionStatsCollection.insert(pos, skUsed, skUnused,n_spandel,n_submapped);\n'),
        # sites 2+3: the window's genotypes are computed when POST_ALIGN reaches its first position
        ("process_pos_variants",
         r'(starling_pos_processor_base::\nprocess_pos_variants\(\n    const pos_t pos,\n    const bool isPosPrecedingReportableRange\)\n\{\n)',
         '\\1    sk_adapter::before_process_pos_variants(*this, pos);\n'),
    ]),
    (L + "applications/starling/starling_pos_processor.hh", [
        ("friend", r'(struct starling_pos_processor : public starling_pos_processor_base\n\{\n)',
         '\\1    friend struct sk_adapter::Access;\n    friend struct sk_adapter::GvcfAccess;\n'),
    ]),
    # site 10: the writer's pipe as the gVCF fast path has to see it (is anything buffered between the caller and the writer?) and
    # the writer's own skip_to_pos / add_site_internal
    (L + "applications/starling/gvcf_aggregator.hh", [
        ("friend + forward declaration", r'class gvcf_aggregator\n\{\n',
         '#include "sk_adapter_fwd.hh"\n\\g<0>    friend struct sk_adapter::GvcfAccess;\n'),
    ]),
    (L + "applications/starling/gvcf_writer.hh", [
        ("friend + forward declaration", r'struct gvcf_writer : public variant_pipe_stage_base\n\{\n',
         '#include "sk_adapter_fwd.hh"\n\\g<0>    friend struct sk_adapter::GvcfAccess;\n'),
    ]),
    (L + "applications/starling/VariantPhaser.hh", [
        ("friend + forward declaration", r'struct VariantPhaser : public variant_pipe_stage_base\n\{\n',
         '#include "sk_adapter_fwd.hh"\n\\g<0>    friend struct sk_adapter::GvcfAccess;\n'),
    ]),
    (L + "applications/starling/VariantOverlapResolver.hh", [
        ("friend + forward declaration", r'struct VariantOverlapResolver : public variant_pipe_stage_base\n\{\n',
         '#include "sk_adapter_fwd.hh"\n\\g<0>    friend struct sk_adapter::GvcfAccess;\n'),
    ]),
    (L + "applications/starling/starling_pos_processor.cpp", [
        ("include", r'#include "starling_pos_processor.hh"\n', '\\g<0>#include "sk_adapter.hh"\n'),
        # site 2: the dependent error probabilities are computed inside the fused site kernel
        ("CleanPileupErrorProb", r'_pileupCleaner\.CleanPileupErrorProb\(sample\(sampleIndex\)\.cleanedPileup\);',
         '/* strelka_amd: adjust_joint_eprob runs fused with the genotype kernel (sk_site_digt_call_fused) */'),
        # site 9 (germline EVS): the position's rank sums, rebuilt from the stream's window just before they are read
        ("germline EVS accumulators",
         r'(            siteSampleInfo\.ReadPosRankSum = pi\.get_read_pos_ranksum\(\);\n)',
         '            sk_adapter::germline_fill_scoring_metrics(sampleIndex, locus.pos, pi);\n\\1'),
        # site 10: a plain homozygous-reference position goes from the stream's window straight into the writer's open block
        ("process_pos_snp",
         r'(starling_pos_processor::\nprocess_pos_snp\(const pos_t pos\)\n\{\n    try\n    \{\n)',
         '\\1        if (sk_adapter::gvcf_plain_site(*this, pos)) return;\n'),
        # site 3
        ("computeSampleDiploidSiteGenotype call",
         r'computeSampleDiploidSiteGenotype\(\n\s*_opt, _dopt, sample\(sampleIndex\), callerPloidy\[sampleIndex\], allDgt\[sampleIndex\]\);',
         'sk_adapter::site_diploid_genotype(*this, pos, sampleIndex, callerPloidy[sampleIndex], allDgt[sampleIndex]);'),
    ]),
    (L + "applications/strelka/strelka_pos_processor.cpp", [
        ("include", r'#include "strelka_pos_processor.hh"\n', '\\g<0>#include "sk_adapter.hh"\n'),
        # site 9 (somatic): the four cleaned pileups of a position are built when something reads them -- the writer of a somatic
        # record, or site 5 for a position the stream's records do not cover -- not for every position
        ("CleanPileup loop",
         r'    for \(unsigned t\(0\); t<n_tier; \+\+t\)\n(    \{\n        const bool is_include_tier2\(t!=0\);\n        if \(is_include_tier2 && \(! _opt\.useTier2Evidence\)\) continue;\n        _pileupCleaner\.CleanPileup\(normal_sif)',
         '    bool skIsCleanDeferred(sk_adapter::somatic_defer_clean(*this, pos));\n    for (unsigned t(0); (! skIsCleanDeferred) && t<n_tier; ++t)\n\\1'),
        # site 5
        ("position_somatic_snv_call",
         r'_dopt\.sscaller_strand_grid\(\)\.position_somatic_snv_call\([^;]*;',
         'sk_adapter::somatic_snv_genotype(*this, pos, normal_cpi_ptr, tumor_cpi_ptr, skIsCleanDeferred, isComputeNonSomatic, sgtg);'),
        # site 9 (somatic): the cleaned pileups and the tumor sample's EVS accumulators of a position, just before its record is written
        ("write_vcf_somatic_snv_genotype_strand_grid",
         r'(\n)(        write_vcf_somatic_snv_genotype_strand_grid\(_opt, _dopt, sgtg,)',
         '\\1        if (skIsCleanDeferred) sk_adapter::somatic_clean_now(*this, pos, normal_cpi_ptr, tumor_cpi_ptr);\n'
         '        sk_adapter::somatic_fill_scoring_metrics(*this, pos);\n\\2'),
        # site 6
        ("get_somatic_indel",
         r'_dopt\.sicaller_grid\(\)\.get_somatic_indel\(_opt,_dopt,',
         'sk_adapter::somatic_indel(_opt,'),
    ]),
    (L + "starling_common/AlleleGroupGenotype.cpp", [
        # site 4: the reference's definition steps aside; adapter/sk_adapter_germline_indel.cpp defines the function
        ("getVariantAlleleGroupGenotypeLhoodsForSample",
         r'\nvoid\ngetVariantAlleleGroupGenotypeLhoodsForSample\(',
         '\nvoid\ngetVariantAlleleGroupGenotypeLhoodsForSample_reference('),
    ]),
    (L + "starling_common/starling_pos_processor_util.cpp", [
        ("include", r'#include "starling_common/starling_pos_processor_util.hh"\n', '\\g<0>#include "sk_adapter.hh"\n'),
        # site 8, second half: the region's alignments normalised in one batch
        ("normalizeAlignment",
         r'        normalizeAlignment\(refBamSeq, readBamSeq, readAlignment\);\n',
         '        if (! sk_adapter::feed_normalize_current(&read_stream, ref, readAlignment)) normalizeAlignment(refBamSeq, readBamSeq, readAlignment);\n'),
    ]),
    (L + "htsapi/bam_streamer.cpp", [
        ("include", r'#include "htsapi/bam_streamer.hh"\n', '\\g<0>#include "sk_adapter.hh"\n'),
        # site 8: the region's reads through the feed
        ("resetRegion",
         r'(    _is_region = true;\n    _region\.clear\(\);\n)',
         '\\1    sk_adapter::feed_reset_region(this, name(), referenceContigId, beginPos, endPos);\n'),
        ("next",
         r'        ret = sam_itr_next\(_hfp, _hitr, _brec\._bp\);\n',
         '        ret = sk_adapter::feed_active(this) ? sk_adapter::feed_next(this, _brec._bp) : sam_itr_next(_hfp, _hitr, _brec._bp);\n'),
        ("destructor",
         r'(bam_streamer::\n~bam_streamer\(\)\n\{\n)',
         '\\1    sk_adapter::feed_drop(this);\n'),
    ]),
    (L + "starling_common/ActiveRegionProcessor.cpp", [
        ("include", r'#include "ActiveRegionProcessor.hh"\n', '\\g<0>#include "sk_adapter.hh"\n'),
        # site 7: haplotype alignment + allele discovery
        ("discoverIndelsAndMismatches",
         r'(    const std::string& haplotypeSeq\(_selectedHaplotypes\[selectedHaplotypeIndex\]\);\n    assert \(haplotypeSeq != _refSegment\);\n)',
         '\\1    if (sk_adapter::discover_indels_and_mismatches(_selectedHaplotypes, selectedHaplotypeIndex, _refSegment, _ref, _posRange.begin_pos(), '
         '_posRange.end_pos(), _prevActiveRegionEnd, _maxIndelSize, discoveredIndelsAndMismatches, numIndels)) return;\n'),
    ]),
]


DOC_BEGIN, DOC_END = "<!-- hooks:begin (generated by adapter/apply_hooks.py --doc; do not edit) -->", "<!-- hooks:end -->"


def doc_table():
    """the edits as a markdown table: reference file, what the edit is for (the description the build prints when an anchor fails), the
    adapter functions the inserted text calls"""
    rows = ["| reference file | edit | calls inserted (`sk_adapter::`) |", "|---|---|---|"]
    for rel, hooks in HOOKS:
        for desc, _anchor, repl in hooks:
            calls = []
            for name in re.findall(r"sk_adapter::([A-Za-z_0-9]+)", repl):
                if name not in calls:
                    calls.append(name)
            if "friend" in desc and set(calls) <= {"Access", "GvcfAccess"}:
                what = " ".join("`friend struct sk_adapter::%s;`" % c for c in calls) + " (no data member: the class layout is unchanged)"
            elif desc == "include":
                what = '`#include "sk_adapter.hh"`'
            elif calls:
                what = ", ".join("`%s`" % c for c in calls)
            elif "_reference(" in repl:
                what = "the reference's definition renamed `..._reference` (never called): `adapter/sk_adapter_germline_indel.cpp` defines the function"
            else:
                what = "(text removed: the work is done by a routed site)"
            rows.append("| `L/%s` | %s | %s |" % (rel[len(L):], desc, what))
    return "\n".join(rows)


def doc_section():
    return DOC_BEGIN + "\n" + doc_table() + "\n" + DOC_END


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "--doc":
        if len(sys.argv) == 2:
            print(doc_table())
            return
        text = open(sys.argv[2]).read()
        a, b = text.index(DOC_BEGIN), text.index(DOC_END) + len(DOC_END)
        with open(sys.argv[2], "w") as f:
            f.write(text[:a] + doc_section() + text[b:])
        return
    ref_root, out_dir = sys.argv[1], sys.argv[2]
    only = set(sys.argv[3:])
    for rel, hooks in HOOKS:
        src = os.path.join(ref_root, rel)
        text = open(src).read()
        for desc, anchor, repl in hooks:
            if only and not any(o in desc for o in only) and "include" not in desc and "friend" not in desc:
                continue
            n = len(re.findall(anchor, text))
            if n != 1:
                sys.exit("apply_hooks: anchor for '%s' matches %d times in %s (expected exactly 1)" % (desc, n, rel))
            text = re.sub(anchor, repl, text, count=1)
        dst = os.path.join(out_dir, rel[len(L):])
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write(text)
        print("hooked", rel, "->", dst)


if __name__ == "__main__":
    main()
