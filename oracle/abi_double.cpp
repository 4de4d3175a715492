// abi_double.cpp -- a CPU TEST DOUBLE of the C-ABI of include/strelka_amd.h.
//
// TEST INFRASTRUCTURE ONLY (built by `make -C oracle double` into oracle/libstrelka_amd_double.so).  It exports the same
// `sk_*` symbols as the product library, with every kernel-backed entry point answered by the CPU oracle
// (oracle/strelka_oracle.c) and the product's own host stages (strelka_amd/host/*.cpp) compiled in unchanged.  Its one
// purpose: let the `-m "not gpu"` suite run the ADAPTER's host logic (adapter/*.cpp: marshalling, stage-window batching,
// geometry shadow, cache validation) end to end on the demo BAMs without a GPU -- binaries `*_dbl` in oracle/_ref/bin.
// Nothing in strelka_amd/ loads it, links it or falls back to it; the product library fails in sk_init without a gfx950
// device.  The `-m gpu` tests run the same adapter against the real library (`*_amd` binaries).
#include "strelka_amd.h"

extern "C" {
#include "strelka_oracle.h"
}

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static std::string g_err;
static bool g_ready = false;

static int fail(const char* m)
{
    g_err = m;
    return 1;
}

// the product's host stages report errors through this (strelka_amd/csrc/sk_common.h)
int sk_fail(const std::string& msg)
{
    g_err = msg;
    return 1;
}

static void to_sko(const sk_germline_options* o, sko_germline_options* g)
{
    g->bsnp_diploid_theta = o->bsnp_diploid_theta;
    g->bsnp_ssd_no_mismatch = o->bsnp_ssd_no_mismatch;
    g->bsnp_ssd_one_mismatch = o->bsnp_ssd_one_mismatch;
    g->is_min_vexp = o->is_min_vexp;
    g->min_vexp = o->min_vexp;
}

extern "C" {

int sk_device_count(void) { return 1; }
void* sk_host_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void sk_host_free(void* p) { std::free(p); }
int sk_broker_enable(int) { return 0; } // (the CPU double has no device side to share)
int sk_broker_client(void) { return 0; }
int sk_init(int) { g_ready = true; return 0; }
int sk_init_strict(int) { g_ready = true; return 0; }
int sk_check_device_errors(void) { return 0; }
int sk_debug_force_device_libm(int) { return 0; }
void sk_shutdown(void) { g_ready = false; }
const char* sk_last_error(void) { return g_err.c_str(); }
int sk_version(void) { return SK_VERSION; }
int sk_is_initialized(void) { return g_ready ? 1 : 0; }
int sk_sync_mode(void) { return g_ready ? 1 : -1; }
int sk_debug_set_g3_variant(int) { return 0; }
int sk_libm_restated(void) { return 1; } // the oracle calls the host libm directly
int sk_abi_double(void) { return 1; }    // marker: lets a test assert which library a binary loaded

int sk_get_qscore_tables(double* q2p, double* q2lncompe, double* q2lne)
{
    sko_get_qscore_tables(q2p, q2lncompe, q2lne);
    return 0;
}

void sk_germline_options_default(sk_germline_options* opt)
{
    opt->bsnp_diploid_theta = 0.001;
    opt->bsnp_ssd_no_mismatch = 0.35;
    opt->bsnp_ssd_one_mismatch = 0.6;
    opt->is_min_vexp = 1;
    opt->min_vexp = 0.25;
}

void sk_somatic_snv_options_default(sk_somatic_snv_options* opt)
{
    opt->bsnp_diploid_theta = 0.001;
    opt->somatic_snv_rate = 1e-4;
    opt->shared_site_error_rate = 5e-10;
    opt->shared_site_error_strand_bias_fraction = 0.0;
    opt->ssnv_contam_tolerance = 0.15;
}

void sk_indel_options_default(sk_indel_options* opt, int is_somatic)
{
    opt->min_read_bp_flank = 5;
    opt->random_base_match_prob = is_somatic ? 0.5 : 0.25;
    opt->tier2_random_base_match_prob = 0.25;
    opt->read_confident_support_threshold = 0.51;
    opt->is_use_alt_indel = 1;
    opt->fast_form = 0;
}

void sk_somatic_indel_options_default(sk_somatic_indel_options* opt)
{
    opt->bindel_diploid_theta = 1e-4;
    opt->somatic_indel_rate = 1e-6;
    opt->shared_indel_error_factor = 2.2;
    opt->indel_contam_tolerance = 0.15;
}

/* the op-level meaning of a flattened batch (include/strelka_amd.h: sk_score_op), sequential double adds */
int sk_score_alignments(const sk_align_batch* b, double* out_lnp)
{
    if (!g_ready) return fail("sk_init() has not succeeded");
    double q2p[71], q2lncompe[71], q2lne[71];
    sko_get_qscore_tables(q2p, q2lncompe, q2lne);
    const double lnthird(-std::log(3.0));
    const double ln_quarter(std::log(0.25));
    const double ln_noncand(std::log(1e-5));
    for (int32_t r = 0; r < b->n_reads; ++r) {
        const uint8_t* code = b->read_code + b->read_off[r];
        const uint8_t* qual = b->read_qual + b->read_off[r];
        const uint8_t* hap = b->hap_code + b->hap_off[r];
        for (int32_t c = b->cal_off[r]; c < b->cal_off[r + 1]; ++c) {
            double lnp = 0.0;
            int64_t rp = 0;
            for (int64_t k = b->op_off[c]; k < b->op_off[c + 1]; ++k) {
                const sk_score_op& op = b->ops[k];
                if (op.kind == SK_OP_BASES) {
                    for (int j = 0; j < op.length; ++j) {
                        const uint8_t rc = code[rp + j];
                        if (rc == SK_BAM_ANY) continue;
                        const uint8_t q = qual[rp + j];
                        if (q > 70) return fail("basecall quality above 70");
                        const bool is_ref = (rc == SK_BAM_REF) || (rc == hap[op.src + j]);
                        lnp += is_ref ? q2lncompe[q] : (q2lne[q] + lnthird);
                    }
                    rp += op.length;
                } else if (op.kind == SK_OP_SOFT_CLIP) {
                    lnp += op.length * ln_quarter;
                    rp += op.length;
                }
                if (op.flags & SK_OPFLAG_NONCANDIDATE_PENALTY) lnp += ln_noncand;
            }
            out_lnp[c] = lnp;
        }
    }
    return 0;
}

int sk_dependent_eprob(const sk_pileup_batch* b, const sk_germline_options* opt, float* out_de)
{
    if (!g_ready) return fail("sk_init() has not succeeded");
    sko_germline_options g;
    to_sko(opt, &g);
    sko_adjust_joint_eprob_batch(b->call_off, b->calls, b->n_loci, &g, out_de);
    return 0;
}

int sk_site_digt_call(const sk_pileup_batch* b, const sk_germline_options* opt, sk_digt_call* out)
{
    if (!g_ready) return fail("sk_init() has not succeeded");
    static_assert(sizeof(sk_digt_call) == sizeof(sko_digt_call), "record layouts must agree");
    sko_germline_options g;
    to_sko(opt, &g);
    sko_site_digt_call_batch(b->call_off, b->calls, b->de, b->ref_base, b->ploidy, b->n_loci, &g,
                             reinterpret_cast<sko_digt_call*>(out));
    return 0;
}

int sk_site_digt_call_fused(const sk_pileup_batch* b, const sk_germline_options* opt, sk_digt_call* out, float* out_de)
{
    if (!g_ready) return fail("sk_init() has not succeeded");
    std::vector<float> de(static_cast<size_t>(b->call_off[b->n_loci]) + 1);
    sko_germline_options g;
    to_sko(opt, &g);
    sko_adjust_joint_eprob_batch(b->call_off, b->calls, b->n_loci, &g, de.data());
    sko_site_digt_call_batch(b->call_off, b->calls, de.data(), b->ref_base, b->ploidy, b->n_loci, &g,
                             reinterpret_cast<sko_digt_call*>(out));
    if (out_de) std::memcpy(out_de, de.data(), sizeof(float) * static_cast<size_t>(b->call_off[b->n_loci]));
    return 0;
}

} // extern "C"

template <int MAXA, typename CallT>
static int allele_groups_t(const sk_allele_group_batch* b, const sk_indel_options* opt, CallT* out)
{
    if (!g_ready) return fail("sk_init() has not succeeded");
    for (int32_t g = 0; g < b->n_groups; ++g) {
        const int64_t r0 = b->read_off[g];
        const int32_t n = static_cast<int32_t>(b->read_off[g + 1] - r0);
        const int32_t n_alt = b->n_alt[g];
        const int ploidy = b->ploidy[g];
        if (n_alt < 1 || n_alt > MAXA) return fail("allele group with an unsupported number of alternate alleles");
        std::vector<float> ref(static_cast<size_t>(n) * n_alt), al(static_cast<size_t>(n) * n_alt);
        std::vector<uint8_t> t1(n), fwd(n);
        for (int32_t r = 0; r < n; ++r) {
            for (int32_t a = 0; a < n_alt; ++a) {
                ref[static_cast<size_t>(r) * n_alt + a] = b->ref_lnp[(r0 + r) * MAXA + a];
                al[static_cast<size_t>(r) * n_alt + a] = b->allele_lnp[(r0 + r) * MAXA + a];
            }
            t1[r] = (b->read_flags[r0 + r] & SK_READ_TIER1) ? 1 : 0;
            fwd[r] = (b->read_flags[r0 + r] & SK_READ_FWD) ? 1 : 0;
        }
        CallT& o = out[g];
        std::memset(&o, 0, sizeof(o));
        uint32_t counts[2 * (MAXA + 2)] = {0};
        sko_allele_group_genotype_lhoods(n, n_alt, ref.data(), al.data(), b->non_ambig + r0, b->read_length + r0, t1.data(),
                                         fwd.data(), b->del_len + static_cast<size_t>(g) * MAXA,
                                         b->ins_len + static_cast<size_t>(g) * MAXA, ploidy, opt->min_read_bp_flank,
                                         opt->random_base_match_prob, opt->read_confident_support_threshold, o.lhood, counts);
        for (int s = 0; s < 2; ++s)
            for (int a = 0; a < n_alt + 2; ++a) o.counts[s][a] = counts[s * (n_alt + 2) + a];
        o.n_genotypes = ploidy == 1 ? uint32_t(n_alt + 1) : uint32_t((n_alt + 1) * (n_alt + 2) / 2);
        uint32_t used = 0;
        for (int32_t r = 0; r < n; ++r) used += t1[r];
        o.n_reads_used = used;
    }
    return 0;
}


extern "C" {

int sk_allele_group_genotype_lhoods(const sk_allele_group_batch* b, const sk_indel_options* opt, sk_allele_group_call* out)
{
    return allele_groups_t<SK_MAX_ALT, sk_allele_group_call>(b, opt, out);
}

int sk_allele_group_genotype_lhoods_wide(const sk_allele_group_batch* b, const sk_indel_options* opt, sk_allele_group_call_wide* out)
{
    return allele_groups_t<SK_MAX_ALT_WIDE, sk_allele_group_call_wide>(b, opt, out);
}

int sk_allele_group_genotype_lhoods_xwide(const sk_allele_group_batch* b, const sk_indel_options* opt, sk_allele_group_call_xwide* out)
{
    return allele_groups_t<SK_MAX_ALT_XWIDE, sk_allele_group_call_xwide>(b, opt, out);
}

int sk_somatic_snv_call_tiers(const sk_pileup_batch* n1, const sk_pileup_batch* t1, const sk_pileup_batch* n2,
                              const sk_pileup_batch* t2, const sk_somatic_snv_options* opt, const uint8_t* is_forced_output,
                              int is_compute_nonsomatic, sk_somatic_snv_genotype* out)
{
    if (!g_ready) return fail("sk_init() has not succeeded");
    static_assert(sizeof(sk_somatic_snv_genotype) == sizeof(sko_somatic_snv_genotype), "record layouts must agree");
    sko_somatic_snv_options so;
    so.bsnp_diploid_theta = opt->bsnp_diploid_theta;
    so.somatic_snv_rate = opt->somatic_snv_rate;
    so.shared_site_error_rate = opt->shared_site_error_rate;
    so.shared_site_error_strand_bias_fraction = opt->shared_site_error_strand_bias_fraction;
    so.ssnv_contam_tolerance = opt->ssnv_contam_tolerance;
    const bool is_tier2 = (n2 != nullptr);
    if (!is_tier2) { n2 = n1; t2 = t1; }
    for (int32_t l = 0; l < n1->n_loci; ++l) {
        sko_position_somatic_snv_call_tiers(n1->calls + n1->call_off[l], int32_t(n1->call_off[l + 1] - n1->call_off[l]),
                                            t1->calls + t1->call_off[l], int32_t(t1->call_off[l + 1] - t1->call_off[l]),
                                            n2->calls + n2->call_off[l], int32_t(n2->call_off[l + 1] - n2->call_off[l]),
                                            t2->calls + t2->call_off[l], int32_t(t2->call_off[l + 1] - t2->call_off[l]),
                                            is_tier2 ? 1 : 0, n1->ref_base[l], &so, is_forced_output ? is_forced_output[l] : 0,
                                            is_compute_nonsomatic, reinterpret_cast<sko_somatic_snv_genotype*>(out + l));
    }
    return 0;
}

int sk_somatic_indel_call_tiers(const sk_somatic_indel_batch* b, const sk_indel_options* nopt, const sk_indel_options* topt,
                                const sk_somatic_indel_options* sopt, int use_tier2_evidence, sk_somatic_indel_genotype* out)
{
    if (!g_ready) return fail("sk_init() has not succeeded");
    static_assert(sizeof(sk_somatic_indel_genotype) == sizeof(sko_somatic_indel_genotype), "record layouts must agree");
    static_assert(sizeof(sk_alt_allele) == sizeof(sko_alt_key), "alternate-allele layouts must agree");
    sko_somatic_indel_params p;
    p.normal_min_read_bp_flank = nopt->min_read_bp_flank;
    p.tumor_min_read_bp_flank = topt->min_read_bp_flank;
    p.random_base_match_prob = topt->random_base_match_prob;
    p.tier2_random_base_match_prob = topt->tier2_random_base_match_prob;
    p.use_tier2_evidence = use_tier2_evidence;
    p.is_use_alt_indel = topt->is_use_alt_indel;
    p.bindel_diploid_theta = sopt->bindel_diploid_theta;
    p.somatic_indel_rate = sopt->somatic_indel_rate;
    p.shared_indel_error_factor = sopt->shared_indel_error_factor;
    p.indel_contam_tolerance = sopt->indel_contam_tolerance;
    for (int32_t i = 0; i < b->n_indels; ++i) {
        if (b->alt_off[i + 1] - b->alt_off[i] > SK_MAX_ALT_ALLELES) return fail("more alternate alleles at one indel than SK_MAX_ALT_ALLELES");
        sko_indel_sample_reads smp[2];
        std::vector<uint8_t> t1[2];
        const sk_readscore_batch* rb[2] = { &b->normal, &b->tumor };
        const int32_t* ak[2] = { b->normal_alt_key, b->tumor_alt_key };
        const float* al[2] = { b->normal_alt_lnp, b->tumor_alt_lnp };
        for (int s = 0; s < 2; ++s) {
            const int64_t r0 = rb[s]->read_off[i];
            const int32_t n = int32_t(rb[s]->read_off[i + 1] - r0);
            t1[s].resize(size_t(n) + 1);
            for (int32_t r = 0; r < n; ++r) t1[s][r] = (rb[s]->read_flags[r0 + r] & SK_READ_TIER1) ? 1 : 0;
            smp[s].n_reads = n;
            smp[s].ref_lnp = rb[s]->ref_lnp + r0;
            smp[s].indel_lnp = rb[s]->indel_lnp + r0;
            smp[s].alt_key = ak[s] + 2 * r0;
            smp[s].alt_lnp = al[s] + 2 * r0;
            smp[s].non_ambig = rb[s]->non_ambig + r0;
            smp[s].read_length = rb[s]->read_length + r0;
            smp[s].is_tier1 = t1[s].data();
        }
        sko_get_somatic_indel(&smp[0], &smp[1], reinterpret_cast<const sko_alt_key*>(b->alt_alleles + b->alt_off[i]),
                              int32_t(b->alt_off[i + 1] - b->alt_off[i]), b->normal.del_len[i], b->normal.ins_len[i],
                              b->normal.is_breakpoint ? b->normal.is_breakpoint[i] : 0, &p, b->indel_to_ref_error_prob[i],
                              b->is_forced_output ? b->is_forced_output[i] : 0, reinterpret_cast<sko_somatic_indel_genotype*>(out + i));
    }
    return 0;
}

void sk_align_scores_default(sk_align_scores* s)
{
    s->match = 1; s->mismatch = -4; s->open = -5; s->extend = -1; s->off_edge = -100; s->insert_delete = -5;
    s->is_allow_edge_insertion = 1; s->is_require_edge_deletion = 1;
}

int sk_global_align(const sk_global_align_batch* b, const sk_align_scores* scores, int32_t* out_score, int32_t* out_begin_pos,
                    sk_path_seg* out_path, int32_t* out_n_seg)
{
    if (!g_ready) return fail("sk_init() has not succeeded");
    sko_align_scores sc;
    sc.match = scores->match; sc.mismatch = scores->mismatch; sc.open = scores->open; sc.extend = scores->extend;
    sc.offEdge = scores->off_edge; sc.insertDelete = scores->insert_delete;
    sc.isAllowEdgeInsertion = scores->is_allow_edge_insertion; sc.isRequireEdgeDeletion = scores->is_require_edge_deletion;
    static_assert(sizeof(sk_path_seg) == sizeof(sko_path_seg), "path segment layouts must agree");
    for (int32_t p = 0; p < b->n; ++p) {
        const int ql = int(b->query_off[p + 1] - b->query_off[p]), rl = int(b->ref_off[p + 1] - b->ref_off[p]);
        const int64_t off = b->query_off[p] + b->ref_off[p] + 4 * int64_t(p);
        const int n = sko_global_align(b->query + b->query_off[p], ql, b->ref + b->ref_off[p], rl, &sc, out_score + p,
                                       out_begin_pos + p, reinterpret_cast<sko_path_seg*>(out_path + off), ql + rl + 4);
        if (n < 0) return fail("sko_global_align failed");
        out_n_seg[p] = n;
    }
    return 0;
}

} // extern "C"

// enumeration == 2 is the device pipeline of the GPU library; the double has no device
struct SkEnumInput;
struct SkEnumOutput;
extern "C" int sk_enum_device_available(void) { return 0; }
namespace skcore { struct PCal; }
extern "C" int sk_enum_device_fetch_cals(uint64_t, int32_t, int32_t, skcore::PCal*) { return 1; }
extern "C" int sk_enum_device_fetch_scores(uint64_t, int32_t, int32_t, double*) { return 1; }
extern "C" void sk_enum_device_job_counts(int64_t* a, int64_t* b, int64_t* c)
{
    if (a) *a = 0;
    if (b) *b = 0;
    if (c) *c = 0;
}
extern "C" int sk_enum_device_rescore(int32_t, float*, int32_t*, int32_t*, int64_t*)
{
    return sk_fail("sk_realign_job_rescore measures the GPU library's kernels");
}
extern "C" int sk_enum_device_run(const SkEnumInput*, SkEnumOutput*)
{
    return sk_fail("candidate enumeration on the device (sk_realign_options.enumeration = 2) needs the GPU library");
}

// ---- the feed: zlib and plain loops stand behind the two kernel-backed entries --------------------------------
#include <zlib.h>

extern "C" int sk_bgzf_inflate(const uint8_t* data, const int64_t* block_off, const int64_t* out_off, int32_t n_blocks, uint8_t* out)
{
    if (n_blocks < 0 || (n_blocks > 0 && (!data || !block_off || !out_off || !out))) return sk_fail("sk_bgzf_inflate: bad argument");
    for (int32_t b = 0; b < n_blocks; ++b) {
        const uint8_t* blk = data + block_off[b];
        const int64_t blen = block_off[b + 1] - block_off[b];
        if (blen < 28 || blk[0] != 31 || blk[1] != 139) return sk_fail("sk_bgzf_inflate: block " + std::to_string(b) + ": not a BGZF block");
        const int xlen = int(blk[10]) | (int(blk[11]) << 8);
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        zs.next_in = const_cast<Bytef*>(blk + 12 + xlen);
        zs.avail_in = uInt(blen - 12 - xlen - 8);
        zs.next_out = out + out_off[b];
        zs.avail_out = uInt(out_off[b + 1] - out_off[b]);
        if (inflateInit2(&zs, -15) != Z_OK) return sk_fail("sk_bgzf_inflate: inflateInit2");
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.avail_out != 0) return sk_fail("sk_bgzf_inflate: block " + std::to_string(b) + ": invalid deflate data");
        const uint8_t* t = blk + blen - 8;
        const uint32_t want = uint32_t(t[0]) | (uint32_t(t[1]) << 8) | (uint32_t(t[2]) << 16) | (uint32_t(t[3]) << 24);
        if (uint32_t(crc32(crc32(0L, nullptr, 0), out + out_off[b], uInt(out_off[b + 1] - out_off[b]))) != want)
            return sk_fail("sk_bgzf_inflate: block " + std::to_string(b) + ": CRC-32 mismatch");
    }
    return 0;
}
static int64_t g_kept_len = -1;
extern "C" int sk_bgzf_inflate_prefixed(const uint8_t* data, const int64_t* block_off, const int64_t* out_off, int32_t n_blocks, const uint8_t* prefix,
                                        int64_t prefix_len, uint8_t* out)
{
    g_kept_len = -1;
    if (prefix_len < 0 || (prefix_len > 0 && !prefix)) return sk_fail("sk_bgzf_inflate: bad prefix");
    if (prefix_len > 0 && out != prefix) std::memmove(out, prefix, size_t(prefix_len));
    if (sk_bgzf_inflate(data, block_off, out_off, n_blocks, out + prefix_len)) return 1;
    g_kept_len = prefix_len + (n_blocks > 0 ? out_off[n_blocks] : 0);
    return 0;
}
extern "C" int sk_bam_decode(const uint8_t* stream, int64_t stream_len, const int64_t* rec_off, int32_t n_records, const int64_t* read_off,
                             const int64_t* path_off, sk_bam_record* rec, uint8_t* read_code, uint8_t* read_qual, sk_path_seg* path);
extern "C" int sk_bam_decode_kept(const uint8_t* stream, int64_t stream_len, const int64_t* rec_off, int32_t n_records, const int64_t* read_off,
                                  const int64_t* path_off, sk_bam_record* rec, uint8_t* read_code, uint8_t* read_qual, sk_path_seg* path)
{
    if (g_kept_len != stream_len) return sk_fail("sk_bam_decode_kept: no stream of this length was kept by sk_bgzf_inflate_prefixed");
    return sk_bam_decode(stream, stream_len, rec_off, n_records, read_off, path_off, rec, read_code, read_qual, path);
}
extern "C" int sk_bgzf_inflate_dev(const uint8_t*, const int64_t*, const int64_t*, int32_t, uint8_t*, int32_t*, void*)
{
    return sk_fail("sk_bgzf_inflate_dev needs the GPU library");
}
extern "C" int sk_bam_decode(const uint8_t* stream, int64_t stream_len, const int64_t* rec_off, int32_t n_records, const int64_t* read_off,
                             const int64_t* path_off, sk_bam_record* rec, uint8_t* read_code, uint8_t* read_qual, sk_path_seg* path)
{
    if (n_records < 0 || stream_len < 0 || (n_records > 0 && (!stream || !rec_off || !read_off || !path_off || !rec || !read_code || !read_qual || !path)))
        return sk_fail("sk_bam_decode: bad argument");
    auto le32 = [](const uint8_t* p) { return int32_t(uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24)); };
    for (int32_t r = 0; r < n_records; ++r) {
        const uint8_t* p = stream + rec_off[r];
        sk_bam_record o;
        std::memset(&o, 0, sizeof(o));
        o.ref_id = le32(p + 4);
        o.pos = le32(p + 8);
        const int l_read_name = p[12];
        o.mapq = p[13];
        o.n_cigar = int(p[16]) | (int(p[17]) << 8);
        o.flag = uint16_t(unsigned(p[18]) | (unsigned(p[19]) << 8));
        o.l_seq = le32(p + 20);
        o.mate_ref_id = le32(p + 24);
        o.mate_pos = le32(p + 28);
        o.template_size = le32(p + 32);
        o.is_fwd_strand = (o.flag & 0x10u) ? 0 : 1;
        rec[r] = o;
        const uint8_t* cig = p + 36 + l_read_name;
        for (int i = 0; i < o.n_cigar; ++i) {
            const uint32_t c = uint32_t(le32(cig + 4 * i));
            path[path_off[r] + i] = sk_path_seg{ (c & 15u) + 1u, c >> 4 };
        }
        const uint8_t* seq = cig + 4 * o.n_cigar;
        const uint8_t* qual = seq + (o.l_seq + 1) / 2;
        for (int32_t i = 0; i < o.l_seq; ++i) {
            read_code[read_off[r] + i] = (i & 1) ? uint8_t(seq[i >> 1] & 15u) : uint8_t(seq[i >> 1] >> 4);
            read_qual[read_off[r] + i] = qual[i];
        }
    }
    return 0;
}
extern "C" int sk_bam_decode_dev(const uint8_t*, const int64_t*, int32_t, const int64_t*, const int64_t*, sk_bam_record*, uint8_t*, uint8_t*, sk_path_seg*, void*)
{
    return sk_fail("sk_bam_decode_dev needs the GPU library");
}

#include "normalize_core.h"
extern "C" int sk_normalize_alignments(const char* ref_seq, int32_t ref_offset, int32_t ref_len, int32_t n_reads, const int64_t* read_off,
                                       const uint8_t* read_code, const int64_t* path_off, int32_t* n_seg, sk_path_seg* path, int32_t* pos, uint8_t* changed)
{
    if (n_reads < 0 || (n_reads > 0 && (!ref_seq || !read_off || !read_code || !path_off || !n_seg || !path || !pos || !changed)))
        return sk_fail("sk_normalize_alignments: bad argument");
    for (int32_t r = 0; r < n_reads; ++r) {
        sknorm::Seqs s{ ref_seq, ref_offset, ref_len, read_code + read_off[r], int32_t(read_off[r + 1] - read_off[r]) };
        sknorm::Aln al{ pos[r], path + path_off[r], n_seg[r] };
        changed[r] = sknorm::normalize_alignment(s, al) ? 1 : 0;
        pos[r] = al.pos;
        n_seg[r] = al.n_seg;
    }
    return 0;
}
extern "C" int sk_normalize_alignments_dev(const char*, int32_t, int32_t, int32_t, const int64_t*, const uint8_t*, const int64_t*, int32_t*, sk_path_seg*,
                                           int32_t*, uint8_t*, void*)
{
    return sk_fail("sk_normalize_alignments_dev needs the GPU library");
}

// ---- the gVCF writer's block logic: the same statement the kernel runs (csrc/gvcf_block_core.h), on this host ----------------------
#include "gvcf_block_core.h"
#include "gvcf_site_core.h"
extern "C" int sk_gvcf_site_summaries(const sk_pileup_batch* hb, const sk_digt_call* genotypes, sk_gvcf_site_summary* out)
{
    if (!g_ready) return sk_fail("sk_init() has not succeeded");
    if (!hb || hb->n_loci < 0 || (hb->n_loci > 0 && (!genotypes || !out))) return sk_fail("sk_gvcf_site_summaries: bad argument");
    for (int32_t i = 0; i < hb->n_loci; ++i)
        out[i] = skgvcf::site_summary(hb->calls + hb->call_off[i], hb->call_off[i + 1] - hb->call_off[i], hb->ref_base[i], hb->ploidy ? hb->ploidy[i] : 2u, genotypes[i]);
    return 0;
}
extern "C" int sk_gvcf_site_summaries_dev(const sk_pileup_batch*, const sk_digt_call*, sk_gvcf_site_summary*, void*)
{
    return sk_fail("sk_gvcf_site_summaries_dev needs the GPU library");
}
extern "C" int sk_gvcf_plain_runs(const sk_gvcf_site_summary* summary, const uint32_t* clean_count, const uint32_t* raw_count, const uint32_t* mapq_count,
                                  const sk_gvcf_block_options* opt, int32_t n, sk_gvcf_run* runs)
{
    if (!g_ready) return sk_fail("sk_init() has not succeeded");
    if (n < 0 || !opt || (n > 0 && (!summary || !clean_count || !raw_count || !mapq_count || !runs))) return sk_fail("sk_gvcf_plain_runs: bad argument");
    std::vector<skgvcf::SitePod> pod(static_cast<size_t>(n) + 1);
    for (int32_t l = 0; l < n; ++l) {
        const sk_gvcf_site_summary& sm = summary[l];
        skgvcf::SitePod p;
        p.gqx = sm.gqx;
        p.used = clean_count[l];
        p.unused = raw_count[l] - p.used;
        p.key_plain = (sm.flags & skgvcf::SITE_PLAIN) ? (uint32_t(skgvcf::POD_PLAIN) | skgvcf::site_filter_key(*opt, sm.gqx, p.used, p.unused, sm.ref_fwd + sm.ref_rev, mapq_count[l])) : 0u;
        pod[static_cast<size_t>(l)] = p;
    }
    std::vector<skgvcf::SiteTile> tiles(static_cast<size_t>(n) / skgvcf::TILE_SITES + 2);
    for (int64_t t = 0; t * skgvcf::TILE_SITES < n; ++t) tiles[static_cast<size_t>(t)] = skgvcf::make_tile(pod.data(), n, t);
    for (int32_t l = 0; l < n; ++l)
        runs[l] = skgvcf::plain_run(pod.data(), tiles.data(), n, l, static_cast<double>(opt->block_percent_tol) / 100., static_cast<int>(opt->block_abs_tol));
    return 0;
}
extern "C" int sk_gvcf_plain_runs_dev(const sk_gvcf_site_summary*, const int64_t*, const int64_t*, const uint32_t*, const sk_gvcf_block_options*, int32_t, void*,
                                      sk_gvcf_run*, void*)
{
    return sk_fail("sk_gvcf_plain_runs_dev needs the GPU library");
}
extern "C" int sk_gvcf_block_sites(const sk_gvcf_site* sites, int32_t n_sites, uint32_t block_percent_tol, uint32_t block_abs_tol, uint8_t* kind,
                                   sk_gvcf_block* blocks)
{
    if (n_sites < 0 || (n_sites > 0 && (!sites || !kind || !blocks))) return sk_fail("sk_gvcf_block_sites: bad argument");
    std::memset(blocks, 0, sizeof(sk_gvcf_block) * size_t(n_sites));
    for (int32_t i = 0; i < n_sites; ++i)
        if (skgvcf::starts_stretch(sites, i)) skgvcf::walk_stretch(sites, n_sites, i, static_cast<double>(block_percent_tol) / 100., int(block_abs_tol), kind, blocks);
    return 0;
}
extern "C" int sk_gvcf_block_sites_dev(const sk_gvcf_site*, int32_t, uint32_t, uint32_t, uint8_t*, sk_gvcf_block*, void*)
{
    return sk_fail("sk_gvcf_block_sites_dev needs the GPU library");

// ---- row a8 as a stream (sk_pileup_stream_*): read after read into per-position columns, exactly as the reference's
// pos_basecall_buffer accumulates them (each read meets the candidate-SNV mask of ITS push), finalised range by range ----
}  // extern "C" (reopened below)

#include <map>

struct sk_pileup_stream
{
    sk_pileup_options opt;
    sk_germline_options gopt;
    bool genotype = false;
    bool somatic = false, want_read_pos = false, want_evs = false;
    bool has_region = false;
    std::string ref;
    int32_t ref_offset = 0, region_begin = 0, region_end = 0;
    std::vector<uint8_t> mask;
    struct Col
    {
        std::vector<uint16_t> t1, t2;
        std::vector<uint32_t> rp; // parallel to t1 (want_read_pos)
        std::vector<uint64_t> ev; // one per live match position (want_evs)
        uint32_t spandel = 0, submapped = 0, mq_n = 0, mq_zero = 0;
        uint64_t mq_sq = 0;
    };
    std::map<int32_t, Col> cols;
    int32_t pending_end = INT32_MIN; // one past the highest position any pushed read covers
    bool has_prev = false;
    int32_t next_begin = 0;
    // output storage
    std::vector<int64_t> o_off1, o_off2;
    std::vector<uint16_t> o_c1, o_c2;
    std::vector<uint32_t> o_sd, o_sm, o_mn, o_mz, o_cn, o_cn4, o_rp;
    std::vector<uint64_t> o_sq;
    // (as the library's output blocks: SK_PILEUP_WINDOW_LIFETIME + 1 in rotation; here for the one array whose longer life the adapter
    // uses -- the window's other arrays live until the next push, the weaker promise)
    std::vector<uint64_t> o_ev_blocks[SK_PILEUP_WINDOW_LIFETIME + 1];
    int o_ev_block = 0;
    std::vector<int64_t> o_evoff;
    std::vector<sk_digt_call> o_g;
    std::vector<sk_gvcf_site_summary> o_sum;
    std::vector<sk_gvcf_run> o_runs;
    bool want_runs = false;
    sk_gvcf_block_options gvcf_opt;
    // the cleaned columns of the last emitted range (CleanPileupFilter(pi,false) / (pi,true))
    std::vector<int64_t> k_off, k_off4;
    std::vector<uint16_t> k_calls, k_calls4;
    std::vector<uint8_t> k_ref;
    bool in_flight = false; // between sk_pileup_stream_push_begin and _finish
    sk_pileup_window pending;
};

struct sk_somatic_pileup_stream
{
    sk_pileup_stream* sample[2] = { nullptr, nullptr };
    sk_somatic_snv_options sopt;
    bool genotype = false, tier2 = false;
    std::vector<sk_somatic_snv_genotype> o_g;
    bool in_flight = false; // between sk_somatic_pileup_stream_push_begin and _finish
    sk_somatic_pileup_window pending;
};

extern "C" {

void sk_pileup_options_default(sk_pileup_options* o)
{
    o->min_basecall_qscore = 17;
    o->mismatch_density_flank_size = 20;
    o->mismatch_density_max_count = 2;
    o->use_tier2_evidence = 0;
    o->tier2_mismatch_density_max_count = 10;
    o->is_mapq_adjust = 1;
    o->min_distance_from_read_edge = 0;
    o->largest_total_indel_ref_span_per_read = 49;
    o->report_begin = 0;
    o->report_end = 0;
}

sk_pileup_stream* sk_pileup_stream_create(const sk_pileup_options* opt, const sk_germline_options* genotype_opt)
{
    if (!g_ready || !opt) {
        fail("sk_pileup_stream_create: not initialised / null options");
        return nullptr;
    }
    sk_pileup_stream* s = new sk_pileup_stream();
    s->opt = *opt;
    if (genotype_opt) {
        s->gopt = *genotype_opt;
        s->genotype = true;
    }
    return s;
}

void sk_pileup_stream_destroy(sk_pileup_stream* s) { delete s; }

int sk_pileup_stream_begin_region(sk_pileup_stream* s, const char* ref_seq, int32_t ref_offset, int32_t ref_len, int32_t report_begin,
                                  int32_t report_end, int32_t span)
{
    if (!g_ready) return fail("sk_init() has not succeeded");
    if (!s || ref_len < 0 || report_end < report_begin) return fail("sk_pileup_stream_begin_region: bad argument");
    if (s->in_flight) return fail("sk_pileup_stream_begin_region: the stream's last push has not been finished");
    s->ref.assign(ref_seq ? ref_seq : "", static_cast<size_t>(ref_len));
    s->ref_offset = ref_offset;
    s->region_begin = report_begin;
    s->region_end = report_end;
    s->opt.report_begin = report_begin;
    s->opt.report_end = report_end;
    s->opt.largest_total_indel_ref_span_per_read = span;
    s->mask.assign(static_cast<size_t>(ref_len) + 1, 0);
    s->cols.clear();
    s->pending_end = INT32_MIN;
    s->has_prev = false;
    s->next_begin = report_begin;
    s->has_region = true;
    return 0;
}

namespace
{

// the new reads into the per-position columns; lo_out = the lowest position they cover (INT32_MAX: none)
int double_ingest(sk_pileup_stream* s, const sk_read_batch* reads, int32_t span, int32_t mask_begin, int32_t mask_len,
                  const uint8_t* cand_snv_mask, int32_t* lo_out)
{
    if (mask_len > 0 && (mask_begin < s->ref_offset || mask_begin + mask_len > s->ref_offset + static_cast<int32_t>(s->ref.size())))
        return fail("sk_pileup_stream_push: candidate-SNV mask window outside the reference segment");
    if (mask_len > 0) std::memcpy(s->mask.data() + (mask_begin - s->ref_offset), cand_snv_mask, static_cast<size_t>(mask_len));
    s->opt.largest_total_indel_ref_span_per_read = span;
    const int n = reads->n_reads;
    // the new reads' own span
    int32_t lo = INT32_MAX, hi = INT32_MIN;
    for (int r = 0; r < n; ++r) {
        const int64_t a = reads->path_off[r], b = reads->path_off[r + 1];
        if (a == b) continue;
        int ref_len = 0;
        for (int64_t i = a; i < b; ++i) {
            const uint32_t t = reads->path[i].type;
            if (t == SK_SEG_MATCH || t == SK_SEG_SEQ_MATCH || t == SK_SEG_SEQ_MISMATCH || t == SK_SEG_DELETE || t == SK_SEG_SKIP) ref_len += static_cast<int>(reads->path[i].length);
        }
        lo = std::min(lo, reads->pos[r]);
        hi = std::max(hi, reads->pos[r] + ref_len);
    }
    if (lo != INT32_MAX) {
        lo = std::max(lo, s->region_begin);
        hi = std::min(hi, s->region_end);
    }
    *lo_out = lo;
    if (lo != INT32_MAX && lo < hi) {
        sko_pileup_options o;
        o.min_basecall_qscore = s->opt.min_basecall_qscore;
        o.mismatch_density_flank_size = s->opt.mismatch_density_flank_size;
        o.mismatch_density_max_count = s->opt.mismatch_density_max_count;
        o.use_tier2_evidence = s->opt.use_tier2_evidence;
        o.tier2_mismatch_density_max_count = s->opt.tier2_mismatch_density_max_count;
        o.is_mapq_adjust = s->opt.is_mapq_adjust;
        o.min_distance_from_read_edge = s->opt.min_distance_from_read_edge;
        o.largest_total_indel_ref_span_per_read = span;
        // the reads see the REGION's report range (is_pos_reportable); columns are collected over [lo, hi)
        sko_read_batch b;
        static_assert(sizeof(sko_path_seg) == sizeof(sk_path_seg), "path segment layouts must agree");
        b.n_reads = n;
        b.read_off = reads->read_off;
        b.read_code = reads->read_code;
        b.read_qual = reads->read_qual;
        b.path_off = reads->path_off;
        b.path = reinterpret_cast<const sko_path_seg*>(reads->path);
        b.pos = reads->pos;
        b.is_fwd = reads->is_fwd;
        b.mapq = reads->mapq;
        b.map_level = reads->map_level;
        b.ref_seq = s->ref.data();
        b.ref_offset = s->ref_offset;
        b.ref_len = static_cast<int32_t>(s->ref.size());
        b.cand_snv_mask = s->mask.data();
        o.report_begin = lo;
        o.report_end = hi;
        // a read that starts inside [lo, hi) but the region's range is narrower: lo/hi were clipped to the region above, and a
        // read is dropped as a whole only by the region's bounds (pos >= report_end / end <= report_begin), which clipping
        // [lo, hi) to the region preserves for every read of this batch
        const size_t nl = static_cast<size_t>(hi - lo);
        const int64_t cap = n ? reads->read_off[n] : 0;
        std::vector<int64_t> off1(nl + 1), off2(nl + 1);
        std::vector<uint16_t> c1(static_cast<size_t>(cap) + 1), c2(static_cast<size_t>(cap) + 1);
        std::vector<uint32_t> sd(nl), sm(nl), mn(nl), mz(nl), rp(static_cast<size_t>(cap) + 1);
        std::vector<uint64_t> sq(nl);
        if (sko_pileup_reads_mapq(&b, &o, 0, off1.data(), c1.data(), cap, sd.data(), sm.data(), mn.data(), mz.data(), sq.data()) < 0)
            return fail("sk_pileup_stream_push: malformed read");
        if (sko_pileup_reads(&b, &o, 1, off2.data(), c2.data(), cap, nullptr, nullptr) < 0) return fail("sk_pileup_stream_push: malformed read");
        if (s->want_read_pos) {
            std::vector<int64_t> offr(nl + 1);
            std::vector<uint16_t> cr(static_cast<size_t>(cap) + 1);
            if (sko_pileup_reads_readpos(&b, &o, offr.data(), cr.data(), cap, rp.data()) < 0) return fail("sk_pileup_stream_push: malformed read");
        }
        std::vector<int64_t> offe(nl + 1);
        std::vector<uint64_t> ew;
        if (s->want_evs) {
            ew.resize(static_cast<size_t>(cap) + 1);
            if (sko_pileup_reads_evs(&b, &o, offe.data(), ew.data(), cap) < 0) return fail("sk_pileup_stream_push: malformed read");
        }
        for (size_t l = 0; l < nl; ++l) {
            const bool any = (off1[l + 1] > off1[l]) || (off2[l + 1] > off2[l]) || sd[l] || sm[l] || mn[l];
            if (!any) continue;
            const int32_t p = lo + static_cast<int32_t>(l);
            if (s->has_prev && p < s->next_begin) return fail("sk_pileup_stream_push: a read reaches positions that an earlier push declared final");
            sk_pileup_stream::Col& c = s->cols[p];
            c.t1.insert(c.t1.end(), c1.begin() + off1[l], c1.begin() + off1[l + 1]);
            c.t2.insert(c.t2.end(), c2.begin() + off2[l], c2.begin() + off2[l + 1]);
            if (s->want_read_pos) c.rp.insert(c.rp.end(), rp.begin() + off1[l], rp.begin() + off1[l + 1]);
            if (s->want_evs) c.ev.insert(c.ev.end(), ew.begin() + offe[l], ew.begin() + offe[l + 1]);
            c.spandel += sd[l];
            c.submapped += sm[l];
            c.mq_n += mn[l];
            c.mq_zero += mz[l];
            c.mq_sq += sq[l];
        }
        s->pending_end = std::max(s->pending_end, hi);
    }
    return 0;
}

// the range a push finalises: as the product computes it (lowest / highest over the reads it still holds).  "lowest" of the
// product = min start over carried + new reads; the carried reads start below next_begin whenever there are any, and the columns
// present in `cols` tell the same story: the first pending position or the new reads' start
void double_extent(const sk_pileup_stream* s, const int32_t lo, int32_t* lowest, int32_t* highest)
{
    *lowest = (lo != INT32_MAX) ? lo : INT32_MAX;
    if (!s->cols.empty()) *lowest = std::min(*lowest, s->cols.begin()->first);
    *highest = s->pending_end;
}

void double_range(const sk_pileup_stream* s, const int32_t lowest, const int32_t highest, const int32_t F, int32_t* begin_out, int32_t* end_out)
{
    int32_t begin = s->next_begin, end = s->next_begin;
    if (lowest != INT32_MAX && highest != INT32_MIN) {
        begin = std::max(s->region_begin, s->has_prev ? std::max(s->next_begin, lowest) : lowest);
        end = std::max(begin, std::min(F, highest));
    }
    if (begin > F) begin = end = std::max(s->next_begin, std::min(begin, F));
    *begin_out = begin;
    *end_out = end;
}

int double_emit(sk_pileup_stream* s, const int32_t begin, const int32_t end, const int32_t F, int32_t ploidy_begin, int32_t ploidy_len,
                const uint8_t* ploidy, sk_pileup_window* out)
{
    const size_t nl = static_cast<size_t>(end - begin);
    s->o_off1.assign(nl + 1, 0); s->o_off2.assign(nl + 1, 0);
    s->o_ev_block = (s->o_ev_block + 1) % (SK_PILEUP_WINDOW_LIFETIME + 1);
    std::vector<uint64_t>& o_ev = s->o_ev_blocks[s->o_ev_block];
    // (a block that comes round again is given up and made anew, so that a reader of a window past its life is a use after free for the
    // sanitizer runs -- tools/diag/sanitize_e2e.sh -- instead of a silent read of newer words)
    std::vector<uint64_t>().swap(o_ev);
    s->o_c1.clear(); s->o_c2.clear(); s->o_rp.clear();
    s->o_evoff.assign(nl + 1, 0);
    s->o_sd.assign(nl, 0); s->o_sm.assign(nl, 0); s->o_mn.assign(nl, 0); s->o_mz.assign(nl, 0); s->o_cn.assign(nl + 1, 0);
    s->o_cn4.assign(nl + 1, 0);
    s->o_sq.assign(nl, 0);
    std::vector<int64_t>& coff = s->k_off;
    std::vector<int64_t>& coff4 = s->k_off4;
    std::vector<uint16_t>& ccalls = s->k_calls;
    std::vector<uint16_t>& ccalls4 = s->k_calls4;
    coff.assign(nl + 1, 0); coff4.assign(nl + 1, 0);
    ccalls.clear(); ccalls4.clear();
    for (size_t l = 0; l < nl; ++l) {
        s->o_off1[l] = static_cast<int64_t>(s->o_c1.size());
        s->o_off2[l] = static_cast<int64_t>(s->o_c2.size());
        s->o_evoff[l] = static_cast<int64_t>(o_ev.size());
        coff[l] = static_cast<int64_t>(ccalls.size());
        coff4[l] = static_cast<int64_t>(ccalls4.size());
        const auto it = s->cols.find(begin + static_cast<int32_t>(l));
        if (it == s->cols.end()) continue;
        const sk_pileup_stream::Col& c = it->second;
        s->o_c1.insert(s->o_c1.end(), c.t1.begin(), c.t1.end());
        s->o_c2.insert(s->o_c2.end(), c.t2.begin(), c.t2.end());
        if (s->want_read_pos) s->o_rp.insert(s->o_rp.end(), c.rp.begin(), c.rp.end());
        if (s->want_evs) o_ev.insert(o_ev.end(), c.ev.begin(), c.ev.end());
        for (const uint16_t bc : c.t1) if (!((bc >> 12) & 1)) ccalls.push_back(bc);
        s->o_cn[l] = static_cast<uint32_t>(ccalls.size() - static_cast<size_t>(coff[l]));
        if (s->somatic) { // CleanPileupFilter(pi, true), PileupCleaner.cpp:43-64
            for (const uint16_t bc : c.t1) if (!((bc >> 12) & 1) || ((bc >> 13) & 1)) ccalls4.push_back(bc);
            for (const uint16_t bc : c.t2) if (!((bc >> 12) & 1)) ccalls4.push_back(bc);
            s->o_cn4[l] = static_cast<uint32_t>(ccalls4.size() - static_cast<size_t>(coff4[l]));
        }
        s->o_sd[l] = c.spandel; s->o_sm[l] = c.submapped; s->o_mn[l] = c.mq_n; s->o_mz[l] = c.mq_zero; s->o_sq[l] = c.mq_sq;
    }
    s->o_off1[nl] = static_cast<int64_t>(s->o_c1.size());
    s->o_off2[nl] = static_cast<int64_t>(s->o_c2.size());
    s->o_evoff[nl] = static_cast<int64_t>(o_ev.size());
    o_ev.push_back(0);
    coff[nl] = static_cast<int64_t>(ccalls.size());
    coff4[nl] = static_cast<int64_t>(ccalls4.size());
    s->o_c1.push_back(0); s->o_c2.push_back(0); ccalls.push_back(0); ccalls4.push_back(0); s->o_rp.push_back(0);
    s->o_g.assign(nl + 1, sk_digt_call());
    s->k_ref.assign(nl + 1, 4);
    for (size_t l = 0; l < nl; ++l) {
        const int64_t k = static_cast<int64_t>(begin) + static_cast<int64_t>(l) - s->ref_offset;
        const char ch = (k >= 0 && k < static_cast<int64_t>(s->ref.size())) ? s->ref[static_cast<size_t>(k)] : 'N';
        s->k_ref[l] = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
    }
    if (s->genotype && nl) {
        std::vector<uint8_t> pl(nl);
        for (size_t l = 0; l < nl; ++l) {
            const int64_t kp = static_cast<int64_t>(begin) + static_cast<int64_t>(l) - ploidy_begin;
            pl[l] = (ploidy && kp >= 0 && kp < ploidy_len) ? ploidy[kp] : 2;
        }
        sk_pileup_batch pb;
        std::memset(&pb, 0, sizeof(pb));
        pb.n_loci = static_cast<int32_t>(nl);
        pb.call_off = coff.data();
        pb.calls = ccalls.data();
        pb.ref_base = s->k_ref.data();
        pb.ploidy = pl.data();
        if (sk_site_digt_call_fused(&pb, &s->gopt, s->o_g.data(), nullptr)) return 1;
    }
    s->cols.erase(s->cols.begin(), s->cols.lower_bound(F));
    if (s->cols.empty()) s->pending_end = INT32_MIN;
    s->has_prev = true;
    s->next_begin = std::max(s->next_begin, F);

    out->begin = begin;
    out->end = end;
    out->tier1_off = s->o_off1.data();
    out->tier1_calls = s->o_c1.data();
    out->tier2_off = s->o_off2.data();
    out->tier2_calls = s->o_c2.data();
    out->spandel_count = s->o_sd.data();
    out->submapped_count = s->o_sm.data();
    out->mapq_count = s->o_mn.data();
    out->mapq_zero_count = s->o_mz.data();
    out->mapq_sum_square = s->o_sq.data();
    out->clean_count = s->o_cn.data();
    out->genotype = s->genotype ? s->o_g.data() : nullptr;
    s->o_sum.assign(nl + 1, sk_gvcf_site_summary());
    if (s->genotype) {
        for (size_t l = 0; l < nl; ++l) {
            const int64_t kp = static_cast<int64_t>(begin) + static_cast<int64_t>(l) - ploidy_begin;
            const unsigned pl = (ploidy && kp >= 0 && kp < ploidy_len) ? ploidy[kp] : 2;
            s->o_sum[l] = skgvcf::site_summary(ccalls.data() + coff[l], coff[l + 1] - coff[l], s->k_ref[l], pl, s->o_g[l]);
        }
    }
    out->site_summary = s->genotype ? s->o_sum.data() : nullptr;
    s->o_runs.assign(nl + 1, sk_gvcf_run());
    if (s->genotype && s->want_runs) {
        std::vector<skgvcf::SitePod> pod(nl + 1);
        for (size_t l = 0; l < nl; ++l) {
            const sk_gvcf_site_summary& sm = s->o_sum[l];
            skgvcf::SitePod p;
            p.gqx = sm.gqx;
            p.used = s->o_cn[l];
            p.unused = static_cast<uint32_t>(s->o_off1[l + 1] - s->o_off1[l]) - p.used;
            p.key_plain = (sm.flags & skgvcf::SITE_PLAIN) ? (uint32_t(skgvcf::POD_PLAIN) | skgvcf::site_filter_key(s->gvcf_opt, sm.gqx, p.used, p.unused, sm.ref_fwd + sm.ref_rev, s->o_mn[l])) : 0u;
            pod[l] = p;
        }
        std::vector<skgvcf::SiteTile> tiles((nl + skgvcf::TILE_SITES - 1) / skgvcf::TILE_SITES + 1);
        for (size_t t = 0; t * skgvcf::TILE_SITES < nl; ++t) tiles[t] = skgvcf::make_tile(pod.data(), static_cast<int64_t>(nl), static_cast<int64_t>(t));
        for (size_t l = 0; l < nl; ++l)
            s->o_runs[l] = skgvcf::plain_run(pod.data(), tiles.data(), static_cast<int64_t>(nl), static_cast<int64_t>(l),
                                             static_cast<double>(s->gvcf_opt.block_percent_tol) / 100., static_cast<int>(s->gvcf_opt.block_abs_tol));
    }
    out->gvcf_runs = (s->genotype && s->want_runs) ? s->o_runs.data() : nullptr;
    out->evs_off = s->want_evs ? s->o_evoff.data() : nullptr;
    out->evs_words = s->want_evs ? s->o_ev_blocks[s->o_ev_block].data() : nullptr;
    return 0;
}

}

int sk_pileup_stream_set_gvcf_block_options(sk_pileup_stream* s, const sk_gvcf_block_options* opt)
{
    if (!s || s->somatic) return fail("sk_pileup_stream_set_gvcf_block_options: bad argument");
    s->want_runs = (opt != nullptr);
    if (opt) s->gvcf_opt = *opt;
    return 0;
}

int sk_pileup_stream_enable_evs_words(sk_pileup_stream* s, int enable)
{
    if (!s || s->somatic) return fail("sk_pileup_stream_enable_evs_words: bad argument");
    s->want_evs = (enable != 0);
    return 0;
}

// (the two halves of a push: the double computes the window in _begin and hands it out in _finish -- the same contract, nothing in flight)
int sk_pileup_stream_push_begin(sk_pileup_stream* s, const sk_read_batch* reads, int32_t span, int32_t mask_begin, int32_t mask_len,
                                const uint8_t* cand_snv_mask, int32_t final_to, int32_t ploidy_begin, int32_t ploidy_len, const uint8_t* ploidy)
{
    if (!s) return fail("sk_pileup_stream_push_begin: null argument");
    if (s->in_flight) return fail("sk_pileup_stream_push_begin: the stream's last push has not been finished");
    if (sk_pileup_stream_push(s, reads, span, mask_begin, mask_len, cand_snv_mask, final_to, ploidy_begin, ploidy_len, ploidy, &s->pending)) return 1;
    s->in_flight = true;
    return 0;
}

int sk_pileup_stream_push_finish(sk_pileup_stream* s, sk_pileup_window* out)
{
    if (!s || !out) return fail("sk_pileup_stream_push_finish: null argument");
    if (!s->in_flight) return fail("sk_pileup_stream_push_finish: no push of this stream has been begun");
    s->in_flight = false;
    *out = s->pending;
    return 0;
}

int sk_pileup_stream_push(sk_pileup_stream* s, const sk_read_batch* reads, int32_t span, int32_t mask_begin, int32_t mask_len,
                          const uint8_t* cand_snv_mask, int32_t final_to, int32_t ploidy_begin, int32_t ploidy_len, const uint8_t* ploidy,
                          sk_pileup_window* out)
{
    if (!g_ready) return fail("sk_init() has not succeeded");
    if (!s || !reads || !out || !s->has_region) return fail("sk_pileup_stream_push: bad argument / no region");
    if (s->in_flight) return fail("sk_pileup_stream_push: the stream's last push has not been finished");
    int32_t lo, lowest, highest, begin, end;
    if (double_ingest(s, reads, span, mask_begin, mask_len, cand_snv_mask, &lo)) return 1;
    const int32_t F = std::min(final_to, s->region_end);
    double_extent(s, lo, &lowest, &highest);
    double_range(s, lowest, highest, F, &begin, &end);
    return double_emit(s, begin, end, F, ploidy_begin, ploidy_len, ploidy, out);
}

sk_somatic_pileup_stream* sk_somatic_pileup_stream_create(const sk_pileup_options* opt, const sk_somatic_snv_options* genotype_opt,
                                                          int with_read_pos)
{
    if (!g_ready || !opt) {
        fail("sk_somatic_pileup_stream_create: not initialised / null options");
        return nullptr;
    }
    sk_somatic_pileup_stream* p = new sk_somatic_pileup_stream();
    for (int i = 0; i < 2; ++i) {
        p->sample[i] = new sk_pileup_stream();
        p->sample[i]->opt = *opt;
        p->sample[i]->somatic = true;
    }
    p->sample[1]->want_read_pos = (with_read_pos != 0);
    p->tier2 = (opt->use_tier2_evidence != 0);
    if (genotype_opt) {
        p->sopt = *genotype_opt;
        p->genotype = true;
    }
    return p;
}

void sk_somatic_pileup_stream_destroy(sk_somatic_pileup_stream* p)
{
    if (!p) return;
    delete p->sample[0];
    delete p->sample[1];
    delete p;
}

int sk_somatic_pileup_stream_begin_region(sk_somatic_pileup_stream* p, const char* ref_seq, int32_t ref_offset, int32_t ref_len,
                                          int32_t report_begin, int32_t report_end, int32_t span)
{
    if (!p) return fail("sk_somatic_pileup_stream_begin_region: null argument");
    for (int i = 0; i < 2; ++i) {
        if (sk_pileup_stream_begin_region(p->sample[i], ref_seq, ref_offset, ref_len, report_begin, report_end, span)) return 1;
    }
    return 0;
}

int sk_somatic_pileup_stream_push_begin(sk_somatic_pileup_stream* p, const sk_read_batch* normal_reads, const sk_read_batch* tumor_reads,
                                        int32_t span, int32_t mask_begin, int32_t mask_len, const uint8_t* cand_snv_mask, int32_t final_to,
                                        int32_t forced_begin, int32_t forced_len, const uint8_t* is_forced_output, int is_compute_nonsomatic)
{
    if (!p) return fail("sk_somatic_pileup_stream_push_begin: null argument");
    if (p->in_flight) return fail("sk_somatic_pileup_stream_push_begin: the stream's last push has not been finished");
    if (sk_somatic_pileup_stream_push(p, normal_reads, tumor_reads, span, mask_begin, mask_len, cand_snv_mask, final_to, forced_begin, forced_len,
                                      is_forced_output, is_compute_nonsomatic, &p->pending))
        return 1;
    p->in_flight = true;
    return 0;
}

int sk_somatic_pileup_stream_push_finish(sk_somatic_pileup_stream* p, sk_somatic_pileup_window* out)
{
    if (!p || !out) return fail("sk_somatic_pileup_stream_push_finish: null argument");
    if (!p->in_flight) return fail("sk_somatic_pileup_stream_push_finish: no push of this stream has been begun");
    p->in_flight = false;
    *out = p->pending;
    return 0;
}

int sk_somatic_pileup_stream_push(sk_somatic_pileup_stream* p, const sk_read_batch* normal_reads, const sk_read_batch* tumor_reads,
                                  int32_t span, int32_t mask_begin, int32_t mask_len, const uint8_t* cand_snv_mask, int32_t final_to,
                                  int32_t forced_begin, int32_t forced_len, const uint8_t* is_forced_output, int is_compute_nonsomatic,
                                  sk_somatic_pileup_window* out)
{
    if (!g_ready) return fail("sk_init() has not succeeded");
    if (p && p->in_flight) return fail("sk_somatic_pileup_stream_push: the stream's last push has not been finished");
    if (!p || !normal_reads || !tumor_reads || !out || !p->sample[0]->has_region) return fail("sk_somatic_pileup_stream_push: bad argument / no region");
    const sk_read_batch* reads[2] = { normal_reads, tumor_reads };
    int32_t lowest = INT32_MAX, highest = INT32_MIN, begin, end;
    for (int i = 0; i < 2; ++i) {
        int32_t lo, l, h;
        if (double_ingest(p->sample[i], reads[i], span, mask_begin, mask_len, cand_snv_mask, &lo)) return 1;
        double_extent(p->sample[i], lo, &l, &h);
        lowest = std::min(lowest, l);
        highest = std::max(highest, h);
    }
    const int32_t F = std::min(final_to, p->sample[0]->region_end);
    double_range(p->sample[0], lowest, highest, F, &begin, &end);
    if (double_emit(p->sample[0], begin, end, F, 0, 0, nullptr, &out->normal)) return 1;
    if (double_emit(p->sample[1], begin, end, F, 0, 0, nullptr, &out->tumor)) return 1;
    const size_t nl = static_cast<size_t>(end - begin);
    p->o_g.assign(nl + 1, sk_somatic_snv_genotype());
    if (p->genotype && nl) {
        std::vector<uint8_t> forced(nl, 0);
        for (size_t l = 0; l < nl; ++l) {
            const int64_t k = static_cast<int64_t>(begin) + static_cast<int64_t>(l) - forced_begin;
            forced[l] = (is_forced_output && k >= 0 && k < forced_len) ? is_forced_output[k] : 0;
        }
        sk_pileup_batch b[4];
        for (int i = 0; i < 2; ++i) {
            sk_pileup_stream* s = p->sample[i];
            std::memset(&b[i], 0, sizeof(sk_pileup_batch));
            b[i].n_loci = static_cast<int32_t>(nl);
            b[i].call_off = s->k_off.data();
            b[i].calls = s->k_calls.data();
            b[i].ref_base = p->sample[0]->k_ref.data();
            b[2 + i] = b[i];
            b[2 + i].call_off = s->k_off4.data();
            b[2 + i].calls = s->k_calls4.data();
        }
        if (sk_somatic_snv_call_tiers(&b[0], &b[1], p->tier2 ? &b[2] : nullptr, p->tier2 ? &b[3] : nullptr, &p->sopt, forced.data(),
                                      is_compute_nonsomatic, p->o_g.data()))
            return 1;
    }
    out->normal_clean_tier2_count = p->sample[0]->o_cn4.data();
    out->tumor_clean_tier2_count = p->sample[1]->o_cn4.data();
    out->tumor_tier1_read_pos = p->sample[1]->want_read_pos ? p->sample[1]->o_rp.data() : nullptr;
    out->genotype = (p->genotype && nl) ? p->o_g.data() : nullptr;
    return 0;
}

}
