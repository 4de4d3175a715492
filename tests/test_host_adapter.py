"""CPU suite: the host adapter (CandidateAlignment -> scoring ops) reproduces the reference's indexing.  The flattened
batch is scored by a test-only Python interpreter (tests/flat_interp.py) and compared bit-for-bit with the oracle's walk
over the reference-shaped input."""
import numpy as np
import pytest

from oracle import pyoracle
from strelka_amd import capi, synth
from tests.flat_interp import score_flat

SEG, INDEL = synth.SEG, synth.INDEL


def _tables():
    return capi.qscore_tables()


def test_random_cases_bit_exact(built):
    rng = np.random.default_rng(11)
    cases = synth.align_cases(120, rng)
    batch = synth.build_align_batch(cases)
    _, lnc, lne = _tables()
    got = score_flat(batch, lnc, lne)
    want = pyoracle.score_cases(cases)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    # every candidate's ops span exactly its read
    cov = np.add.reduceat(np.where(batch.ops["kind"] < 2, batch.ops["length"], 0).astype(np.int64), batch.op_off[:-1])
    read_len = np.repeat(np.diff(batch.read_off), np.diff(batch.cal_off))
    assert np.array_equal(cov, read_len)


def _one(read, qual, ref, ref_off, cal):
    case = dict(read_code=np.array(read, np.uint8), read_qual=np.array(qual, np.uint8), ref_seq=ref, ref_offset=ref_off,
                cals=[cal])
    b = synth.build_align_batch([case])
    _, lnc, lne = _tables()
    return score_flat(b, lnc, lne)[0], pyoracle.score_cases([case])[0], b


def test_leading_edge_insert_uses_tail_of_insert_sequence(built):
    # read = last two bases of insertion "ACGT" (G,T) then 4 matching bases
    ref = "AACCGGTT"
    read = [4, 8, 1, 1, 2, 2]  # G T A A C C
    cal = dict(pos=100, path=[(SEG["INSERT"], 2), (SEG["MATCH"], 4)], indels=[],
               leading=dict(pos=100, type=INDEL["INDEL"], del_len=0, ins_seq="ACGT", is_candidate=1), trailing=None)
    got, want, b = _one(read, [30] * 6, ref, 100, cal)
    assert got == want
    _, lnc, _ = _tables()
    assert got == pytest.approx(6 * lnc[30], rel=1e-12)  # all six bases match


def test_swap_and_noncandidate_penalty(built):
    ref = "ACGTACGTACGTACGT"
    # 4M 2D1I 5M : swap at pos 4 (delete 2, insert 'T'), non-candidate -> one ln(1e-5) penalty
    read = [1, 2, 4, 8, 8, 4, 8, 1, 2, 4]
    cal = dict(pos=0, path=[(SEG["MATCH"], 4), (SEG["DELETE"], 2), (SEG["INSERT"], 1), (SEG["MATCH"], 5)],
               indels=[dict(pos=4, type=INDEL["INDEL"], del_len=2, ins_seq="T", is_candidate=0)], leading=None, trailing=None)
    got, want, b = _one(read, [40] * 10, ref, 0, cal)
    assert got == want
    assert int((b.ops["flags"] & 1).sum()) == 1


def test_soft_clip_hard_clip_and_N(built):
    ref = "ACGTNCGTAC"
    read = [15, 1, 2, 4, 8, 0, 2, 4]  # N A C G T = C G   ('=' always matches, N skipped)
    cal = dict(pos=0, path=[(SEG["HARD_CLIP"], 5), (SEG["SOFT_CLIP"], 1), (SEG["MATCH"], 7)], indels=[], leading=None,
               trailing=None)
    got, want, _ = _one(read, [20, 20, 20, 20, 20, 20, 20, 20], ref, 0, cal)
    assert got == want


def test_alignment_off_the_reference_segment_reads_N(built):
    ref = "ACGT"
    read = [1, 2, 4, 8, 1, 2]
    cal = dict(pos=98, path=[(SEG["MATCH"], 6)], indels=[], leading=None, trailing=None)  # segment covers 100..103
    got, want, _ = _one(read, [30] * 6, ref, 100, cal)
    assert got == want


def test_quality_above_70_is_rejected_like_the_reference(built):
    b = capi.AlignBuilder()
    with pytest.raises(capi.StrelkaAmdError) as e:
        b.add_read(np.array([1, 2], np.uint8), np.array([30, 71], np.uint8), "AC", 0,
                   [dict(pos=0, path=[(SEG["MATCH"], 2)], indels=[], leading=None, trailing=None)])
    assert "exceeds the maximum cached" in str(e.value)


def test_gap_without_indel_key_is_an_error(built):
    b = capi.AlignBuilder()
    with pytest.raises(capi.StrelkaAmdError):
        b.add_read(np.array([1, 2, 4], np.uint8), np.array([30, 30, 30], np.uint8), "ACGGT", 0,
                   [dict(pos=0, path=[(SEG["MATCH"], 2), (SEG["DELETE"], 1), (SEG["MATCH"], 1)], indels=[], leading=None,
                         trailing=None)])


def test_empty_read_and_no_candidates(built):
    b = capi.AlignBuilder()
    b.add_read(np.zeros(0, np.uint8), np.zeros(0, np.uint8), "ACGT", 0, [])
    b.add_read(np.array([1], np.uint8), np.array([30], np.uint8), "ACGT", 0,
               [dict(pos=0, path=[(SEG["MATCH"], 1)], indels=[], leading=None, trailing=None)])
    batch = b.finish()
    assert batch.n_reads == 2 and batch.n_cals == 1
    assert list(batch.cal_off) == [0, 0, 1]


def test_column_form_matches_the_ops():
    """sk_align_prepare_cols: every read position of every candidate alignment selects the term its ops say (agree / differ /
    nothing: read base N, soft clip, past the end); the add mask has exactly the positions of entries that add terms"""
    rng = np.random.default_rng(77)
    cases = synth.align_cases(60, rng) + synth.align_cases_h64(8, rng) + synth.align_cases(5, rng, L=203, K=6, max_cals=70)
    hb = synth.build_align_batch(cases).prepare()
    W = hb.evmask_words
    col_of = {1: 0, 2: 1, 4: 2, 8: 3}
    for r in range(hb.n_reads):
        L = int(hb.read_off[r + 1] - hb.read_off[r])
        hap = hb.hap_code[hb.hap_off[r]:hb.hap_off[r + 1]]
        c0, c1 = int(hb.cal_off[r]), int(hb.cal_off[r + 1])
        ncr, nch = c1 - c0, (L + 7) // 8
        cm = hb.colmat[int(hb.colmat_off[r]):int(hb.colmat_off[r + 1])].view(np.uint8).reshape(nch, ncr, 4)
        want_mask = np.zeros(W, np.uint32)
        for j in range(ncr):
            c = c0 + j
            ent = hb.entries[int(hb.op_off[c]) + 2 * c: int(hb.op_off[c + 1]) + 2 * (c + 1)]
            assert ent[0] != 0xffffffff
            for e in ent:
                if (e & 1023) == 1023:
                    break
                if e & ((7 << 10) | (1 << 13)):
                    p = int(e & 1023)
                    want_mask[p >> 5] |= np.uint32(1 << (p & 31))
            want = np.full(8 * nch, 2, np.uint8)
            read = hb.read_code[int(hb.read_off[r]):int(hb.read_off[r + 1])]
            pos = 0
            for op in hb.ops[int(hb.op_off[c]):int(hb.op_off[c + 1])]:
                n = int(op["length"])
                if op["kind"] == capi.OP_BASES:
                    for t in range(n):
                        h, rc = int(hap[int(op["src"]) + t]), int(read[pos + t])
                        want[pos + t] = 2 if rc == 15 else 0 if (rc == 0 or (rc == h and rc in col_of)) else 1
                    pos += n
                elif op["kind"] == capi.OP_SOFT_CLIP:
                    pos += n
            assert pos == L
            by = cm[:, j, :]                                                    # [word][byte]
            nib = np.concatenate([by & 15, by >> 4], axis=1).reshape(-1)        # positions 8k..8k+3 low nibbles, 8k+4..8k+7 high
            assert np.array_equal(nib & 3, want), (r, j)
            # bit 2: exactly one non-candidate penalty (no soft clip) is added before the position's term
            want_flag = np.zeros(8 * nch, np.uint8)
            for e in ent:
                if (e & 1023) == 1023:
                    break
                if e & ((7 << 10) | (1 << 13)):
                    p = int(e & 1023)
                    if ((e >> 10) & 7) == 1 and not (e & (1 << 13)) and p < 8 * nch:
                        want_flag[p] = 1
                    else:
                        want_mask[W - 1] |= np.uint32(1 << 30)
            assert np.array_equal((nib >> 2) & 1, want_flag), (r, j)
            assert not np.any(nib >> 3)
        assert np.array_equal(hb.addmask[r * W:(r + 1) * W], want_mask), r
