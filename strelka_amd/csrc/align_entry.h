// align_entry.h -- the device-ready ("prepared") form of a candidate alignment's scoring ops, shared by the host
// preparation (host/align_flatten.cpp: sk_align_prepare) and kernel A1 (score_alignments.hip).
//
// A candidate alignment is a short list of 32-bit TRANSITION ENTRIES, one per op that spans read bases plus a tail:
//   [9:0]   read position at which the entry takes effect (strictly increasing inside a list; 1023 = end of list)
//   [12:10] number of non-candidate-indel penalties added first (ln 1e-5 each)
//   [13]    soft clip: add (next entry's position - this position) * ln(1/4) once, then read the 0.0 column
//   [17:15] / [20:18] row-column index (0..5) of the haplotype base facing the first / second read position
//   [31:21] hap index base + SK_ENT_HIDX_BIAS: the haplotype base facing read position i is hap[base + i]; soft clips
//           and the tail point into the run of "0.0 column" bytes that follows the read's pool in LDS (index P - pos)
// Candidate c owns the slots [op_off[c] + 2c, op_off[c+1] + 2(c+1)) of the entries array (every op yields at most one
// entry; + tail + end marker); unused slots hold end markers.  A candidate that does not fit the format starts with
// SK_ENT_COMPLEX and is scored by the generic routine.
// The event mask of a read (evmask_words 32-bit words) has bit p set when ANY candidate of the read has an entry at
// read position p, 0 < p <= read length.
#pragma once
#include <cstdint>

constexpr unsigned SK_ENT_POS_MASK = 1023u;
constexpr unsigned SK_ENT_END = 1023u;           // end-of-list marker (position field)
constexpr unsigned SK_ENT_COMPLEX = 0xffffffffu; // first slot: score this candidate with the generic routine
constexpr unsigned SK_ENT_ADD_BITS = (7u << 10) | (1u << 13);
constexpr int SK_ENT_HIDX_BIAS = 1024;
constexpr int SK_ENT_MAX_READ_LEN = 1022;
constexpr int SK_ENT_MAX_POOL = 1023; // hap index bases reach the pool size
constexpr int SK_ENT_ZERO_COL = 5;    // column index of the 0.0 term

#if defined(__HIP__)
__host__ __device__
#endif
static inline unsigned sk_ent_col_index(const unsigned bam_code)
{
    return bam_code == 1u ? 0u : bam_code == 2u ? 1u : bam_code == 4u ? 2u : bam_code == 8u ? 3u : 4u;
}
static inline int sk_ent_evmask_words(const int max_read_len) { return (max_read_len + 1 + 63) / 32 + 1; }

// column form: the term a haplotype base (table column 0..5, above) selects against a read base (BAM code)
constexpr unsigned SK_SEL_MATCH = 0, SK_SEL_MISMATCH = 1, SK_SEL_NONE = 2;
#if defined(__HIP__)
__host__ __device__
#endif
static inline unsigned sk_col_selector(const unsigned col, const unsigned read_code)
{
    if (col == unsigned(SK_ENT_ZERO_COL) || read_code == 15u) return SK_SEL_NONE; // soft clip / past the end; read base N
    if (read_code == 0u) return SK_SEL_MATCH;                                      // '=' always agrees
    const unsigned code_of_col = (col < 4u) ? (1u << col) : 0x100u;
    return code_of_col == read_code ? SK_SEL_MATCH : SK_SEL_MISMATCH;
}
