"""Device-resident batches for the `*_dev` entry points.  torch is used only as the device allocator / stream provider
(bench.py, tests); nothing here computes."""
import ctypes as C

import numpy as np
import torch

from . import capi


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeviceAlignBatch:
    def __init__(self, hb, device="cuda:0", tile=1, columns=True):
        """Upload a capi.HostAlignBatch; `tile` > 1 replicates it on the device (offsets shifted) to reach bench sizes.
        `columns` False leaves the column form out: the kernel then follows the transition entries."""
        self.device = device
        hb.prepare(columns=columns)
        ops_bytes = hb.ops.view(np.uint8).reshape(-1, 8)
        t = dict(read_off=_t(hb.read_off, device), read_code=_t(hb.read_code, device), read_qual=_t(hb.read_qual, device),
                 hap_off=_t(hb.hap_off, device), hap_code=_t(hb.hap_code, device), cal_off=_t(hb.cal_off, device),
                 op_off=_t(hb.op_off, device), ops=_t(ops_bytes, device))
        if tile > 1:
            def tile_off(x, dtype):
                base = x[:-1]
                total = x[-1]
                k = torch.arange(tile, device=device, dtype=dtype)[:, None] * total
                return torch.cat([(base[None, :] + k).reshape(-1), (total * tile).reshape(1).to(dtype)])
            t["read_off"] = tile_off(t["read_off"], torch.int64)
            t["hap_off"] = tile_off(t["hap_off"], torch.int64)
            t["cal_off"] = tile_off(t["cal_off"].to(torch.int64), torch.int64).to(torch.int32)
            t["op_off"] = tile_off(t["op_off"], torch.int64)
            for k in ("read_code", "read_qual", "hap_code"):
                t[k] = t[k].repeat(tile)
            t["ops"] = t["ops"].repeat(tile, 1)
        # prepared form: candidate c owns entry slots [op_off[c] + 2c, ...), so tiles simply repeat
        ent = hb.entries[:len(hb.ops) + 2 * hb.n_cals]
        msk = hb.evmask[:hb.n_reads * hb.evmask_words]
        t["entries"] = _t(ent.view(np.int32), device).repeat(tile)
        t["evmask"] = _t(msk.view(np.int32), device).repeat(tile)
        self.evmask_words = hb.evmask_words
        self.has_columns = bool(columns)
        if columns:
            words = int(hb.colmat_off[-1])
            t["colmat"] = _t(hb.colmat[:words].view(np.int32), device).repeat(tile)
            off = _t(hb.colmat_off, device)
            if tile > 1:
                k = torch.arange(tile, device=device, dtype=torch.int64)[:, None] * words
                off = torch.cat([(off[None, :-1] + k).reshape(-1), torch.tensor([words * tile], device=device, dtype=torch.int64)])
            t["colmat_off"] = off
            t["addmask"] = _t(hb.addmask[:hb.n_reads * hb.evmask_words].view(np.int32), device).repeat(tile)
        self.t = t
        self.n_reads = hb.n_reads * tile
        self.n_cals = hb.n_cals * tile
        self.n_ops = len(hb.ops) * tile
        self.n_bases = len(hb.read_code) * tile
        self.max_read_len = hb.max_read_len
        self.max_hap_len = hb.max_hap_len
        self.out = torch.empty(self.n_cals, dtype=torch.float64, device=device)

    def struct(self):
        t = self.t
        return capi.AlignBatch(self.n_reads, self.n_cals, self.n_ops, t["read_off"].data_ptr(), t["read_code"].data_ptr(),
                               t["read_qual"].data_ptr(), t["hap_off"].data_ptr(), t["hap_code"].data_ptr(),
                               t["cal_off"].data_ptr(), t["op_off"].data_ptr(), t["ops"].data_ptr(), self.max_read_len,
                               self.max_hap_len, t["entries"].data_ptr(), t["evmask"].data_ptr(), self.evmask_words,
                               t["colmat"].data_ptr() if self.has_columns else None,
                               t["colmat_off"].data_ptr() if self.has_columns else None,
                               t["addmask"].data_ptr() if self.has_columns else None)

    def score(self, generic=False):
        """Enqueue the scoring kernel on torch's current stream; returns the device output tensor."""
        s = self.struct()
        fn = capi.lib().sk_score_alignments_dev_generic if generic else capi.lib().sk_score_alignments_dev
        capi._check(fn(C.byref(s), C.c_void_p(self.out.data_ptr()), _stream_ptr()))
        return self.out


class DevicePileupBatch:
    def __init__(self, hb, device="cuda:0", tile=1, de=None):
        self.device = device
        t = dict(call_off=_t(hb.call_off, device), calls=_t(hb.calls.view(np.int16), device), ref_base=_t(hb.ref_base, device))
        if tile > 1:
            total = t["call_off"][-1]
            k = torch.arange(tile, device=device, dtype=torch.int64)[:, None] * total
            t["call_off"] = torch.cat([(t["call_off"][:-1][None, :] + k).reshape(-1), (total * tile).reshape(1)])
            t["calls"] = t["calls"].repeat(tile)
            t["ref_base"] = t["ref_base"].repeat(tile)
        self.t = t
        self.n_loci = hb.n_loci * tile
        self.n_calls = len(hb.calls) * tile
        self.ploidy = None
        if hb.ploidy is not None:
            self.ploidy = _t(hb.ploidy, device).repeat(tile)
        self.de = torch.empty(self.n_calls, dtype=torch.float32, device=device)
        if de is not None:
            self.de.copy_(_t(de, device).repeat(tile))
        self.scratch = None
        self.digt_out = None
        self.som_out = None

    def struct(self, with_de=True):
        t = self.t
        return capi.PileupBatch(self.n_loci, t["call_off"].data_ptr(), t["calls"].data_ptr(),
                                self.de.data_ptr() if with_de else None, t["ref_base"].data_ptr(),
                                None if self.ploidy is None else self.ploidy.data_ptr())

    def dependent_eprob(self, opt=None):
        opt = opt or capi.germline_options()
        if self.scratch is None:
            self.scratch = torch.empty(self.n_calls + self.n_loci + 4, dtype=torch.int32, device=self.device)
        s = self.struct(with_de=False)
        capi._check(capi.lib().sk_dependent_eprob_dev(C.byref(s), C.byref(opt), C.c_void_p(self.de.data_ptr()),
                                                      C.c_void_p(self.scratch.data_ptr()), _stream_ptr()))
        return self.de

    def site_digt_call(self, opt=None):
        opt = opt or capi.germline_options()
        if self.digt_out is None:
            self.digt_out = torch.empty(self.n_loci * capi.DIGT_CALL_DTYPE.itemsize, dtype=torch.uint8, device=self.device)
        s = self.struct()
        capi._check(capi.lib().sk_site_digt_call_dev(C.byref(s), C.byref(opt), C.c_void_p(self.digt_out.data_ptr()),
                                                     _stream_ptr()))
        return self.digt_out

    def site_digt_call_fused(self, opt=None, want_de=False):
        opt = opt or capi.germline_options()
        if self.digt_out is None:
            self.digt_out = torch.empty(self.n_loci * capi.DIGT_CALL_DTYPE.itemsize, dtype=torch.uint8, device=self.device)
        if self.scratch is None:
            self.scratch = torch.empty(self.n_calls + self.n_loci + 4, dtype=torch.int32, device=self.device)
        s = self.struct(with_de=False)
        capi._check(capi.lib().sk_site_digt_call_fused_dev(C.byref(s), C.byref(opt), C.c_void_p(self.digt_out.data_ptr()),
                                                           C.c_void_p(self.de.data_ptr()), int(want_de),
                                                           C.c_void_p(self.scratch.data_ptr()), self.n_calls, _stream_ptr()))
        return self.digt_out

    def digt_numpy(self):
        return self.digt_out.cpu().numpy().view(capi.DIGT_CALL_DTYPE)


def somatic_snv_call_dev(normal, tumor, opt=None, is_forced_output=False):
    opt = opt or capi.somatic_snv_options()
    if normal.som_out is None:
        normal.som_out = torch.empty(normal.n_loci * capi.SOMATIC_CALL_DTYPE.itemsize, dtype=torch.uint8,
                                     device=normal.device)
    sn, st = normal.struct(with_de=False), tumor.struct(with_de=False)
    if getattr(normal, "som_scratch", None) is None:
        normal.som_scratch = torch.empty(4 * (normal.n_loci + 4), dtype=torch.uint8, device=normal.device)
    capi._check(capi.lib().sk_somatic_snv_call_batch_dev(C.byref(sn), C.byref(st), C.byref(opt), int(is_forced_output),
                                                         C.c_void_p(normal.som_out.data_ptr()),
                                                         C.c_void_p(normal.som_scratch.data_ptr()), _stream_ptr()))
    return normal.som_out


class DeviceReadBatch:
    """synth.ReadBatch resident on the device + output buffers for sk_pileup_reads_dev (row a8)"""

    def __init__(self, rb, n_loci, device="cuda:0", report_begin=0):
        self.rb, self.n_loci, self.device = rb, n_loci, device
        ref = np.frombuffer(rb.ref_seq.encode(), np.uint8).copy()
        self.t = dict(read_off=_t(rb.read_off, device), read_code=_t(rb.read_code, device), read_qual=_t(rb.read_qual, device),
                      path_off=_t(rb.path_off, device), path=_t(rb.path.view(np.int32), device), pos=_t(rb.pos, device),
                      is_fwd=_t(rb.is_fwd, device), mapq=_t(rb.mapq, device), map_level=_t(rb.map_level, device),
                      ref=_t(ref, device))
        t = self.t
        self.s = capi.ReadBatchStruct(rb.n_reads, *[t[k].data_ptr() for k in ("read_off", "read_code", "read_qual", "path_off",
                                                                                "path", "pos", "is_fwd", "mapq", "map_level",
                                                                                "ref")], rb.ref_offset, len(ref), None)
        self.opt = capi.pileup_options(report_begin=report_begin, report_end=report_begin + n_loci)
        self.cap = rb.n_bases + 16
        self.call_off = torch.empty(n_loci + 1, dtype=torch.int64, device=device)
        self.calls = torch.empty(self.cap, dtype=torch.int16, device=device)
        self.spandel = torch.empty(max(n_loci, 1), dtype=torch.int32, device=device)
        self.submapped = torch.empty(max(n_loci, 1), dtype=torch.int32, device=device)
        self.scratch = torch.empty(capi.lib().sk_pileup_scratch_bytes(rb.n_reads, rb.n_bases, n_loci), dtype=torch.uint8,
                                   device=device)
        self.out = capi.PileupColumns(n_loci, self.cap, self.call_off.data_ptr(), self.calls.data_ptr(),
                                      self.spandel.data_ptr(), self.submapped.data_ptr())

    def pileup(self, mode=capi.PILEUP_CLEAN_TIER1):
        capi._check(capi.lib().sk_pileup_reads_dev(C.byref(self.s), self.rb.n_bases, C.byref(self.opt), mode,
                                                   C.byref(self.out), self.scratch.data_ptr(), _stream_ptr()))
        return self.call_off, self.calls


class DeviceReadScoreBatch:
    """sk_readscore_batch resident on the device (a14 likelihood half)."""

    def __init__(self, hb, device="cuda:0"):
        self.device, self.n_indels = device, hb.n_indels
        self.n_reads = int(hb.read_off[-1])
        names = ("read_off", "ref_lnp", "indel_lnp", "alt_lnp", "non_ambig", "read_length", "read_flags", "del_len",
                 "ins_len", "is_breakpoint")
        self.t = {k: (None if getattr(hb, k) is None else _t(getattr(hb, k), device)) for k in names}
        self.out = torch.empty((hb.n_indels, 21), dtype=torch.float64, device=device)

    def struct(self):
        p = {k: (None if v is None else C.c_void_p(v.data_ptr())) for k, v in self.t.items()}
        return capi.ReadScoreBatch(self.n_indels, p["read_off"], p["ref_lnp"], p["indel_lnp"], p["alt_lnp"], p["non_ambig"],
                                   p["read_length"], p["read_flags"], p["del_len"], p["ins_len"], p["is_breakpoint"])

    def grid_lhood(self, opt=None, is_include_tier2=False):
        opt = opt or capi.indel_options(True)
        s = self.struct()
        capi._check(capi.lib().sk_indel_grid_lhood_dev(C.byref(s), C.byref(opt), int(is_include_tier2),
                                                       C.c_void_p(self.out.data_ptr()), _stream_ptr()))
        return self.out


class DeviceAlleleGroupBatch:
    """sk_allele_group_batch resident on the device (a11)."""

    def __init__(self, hb, device="cuda:0"):
        self.device, self.n_groups = device, hb.n_groups
        self.n_reads = int(hb.read_off[-1])
        names = ("read_off", "n_alt", "ploidy", "del_len", "ins_len", "ref_lnp", "allele_lnp", "non_ambig", "read_length",
                 "read_flags")
        self.t = {k: _t(getattr(hb, k), device) for k in names}
        self.out = torch.empty(hb.n_groups * capi.ALLELE_GROUP_CALL_DTYPE.itemsize, dtype=torch.uint8, device=device)

    def struct(self):
        p = {k: C.c_void_p(v.data_ptr()) for k, v in self.t.items()}
        return capi.AlleleGroupBatch(self.n_groups, p["read_off"], p["n_alt"], p["ploidy"], p["del_len"], p["ins_len"],
                                     p["ref_lnp"], p["allele_lnp"], p["non_ambig"], p["read_length"], p["read_flags"])

    def genotype_lhoods(self, opt=None):
        opt = opt or capi.indel_options(False)
        s = self.struct()
        capi._check(capi.lib().sk_allele_group_genotype_lhoods_dev(C.byref(s), C.byref(opt), C.c_void_p(self.out.data_ptr()),
                                                                   _stream_ptr()))
        return self.out


class DeviceGlobalAlignBatch:
    """(haplotype, reference segment) pairs resident on the device for sk_global_align_dev (next row f2)."""

    def __init__(self, pairs, device="cuda:0"):
        self.n = len(pairs)
        qo = np.zeros(self.n + 1, np.int64)
        ro = np.zeros(self.n + 1, np.int64)
        for i, (q, r) in enumerate(pairs):
            qo[i + 1] = qo[i] + len(q)
            ro[i + 1] = ro[i] + len(r)
        self.qo, self.ro = qo, ro
        self.nq, self.nr = int(qo[-1]), int(ro[-1])
        self.max_q = max((len(q) for q, _ in pairs), default=1)
        self.max_r = max((len(r) for _, r in pairs), default=1)
        self.cells = sum(len(q) * len(r) for q, r in pairs)
        q = np.frombuffer("".join(q for q, _ in pairs).encode(), np.uint8).copy()
        r = np.frombuffer("".join(r for _, r in pairs).encode(), np.uint8).copy()
        self.t = dict(qo=_t(qo, device), q=_t(q, device), ro=_t(ro, device), r=_t(r, device))
        self.npath = self.nq + self.nr + 4 * self.n
        self.score = torch.empty(self.n, dtype=torch.int32, device=device)
        self.begin = torch.empty(self.n, dtype=torch.int32, device=device)
        self.nseg = torch.empty(self.n, dtype=torch.int32, device=device)
        self.path = torch.empty((self.npath, 2), dtype=torch.int32, device=device)
        L = capi.lib()
        L.sk_global_align_scratch_bytes.restype = C.c_size_t
        L.sk_global_align_scratch_bytes.argtypes = [C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32]
        nbytes = L.sk_global_align_scratch_bytes(self.n, self.nq, self.nr, self.max_q, self.max_r)
        self.scratch = torch.empty(nbytes, dtype=torch.uint8, device=device)
        L.sk_global_align_dev.argtypes = [C.POINTER(capi.GlobalAlignBatch), C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                          C.POINTER(capi.AlignScores)] + [C.c_void_p] * 6

    def align(self, scores=None):
        scores = scores or capi.align_scores()
        t = self.t
        b = capi.GlobalAlignBatch(self.n, t["qo"].data_ptr(), t["q"].data_ptr(), t["ro"].data_ptr(), t["r"].data_ptr())
        capi._check(capi.lib().sk_global_align_dev(C.byref(b), self.nq, self.nr, self.max_q, self.max_r, C.byref(scores),
                                                   self.score.data_ptr(), self.begin.data_ptr(), self.path.data_ptr(),
                                                   self.nseg.data_ptr(), self.scratch.data_ptr(), _stream_ptr()))

    def results(self):
        """-> [(score, begin_pos, cigar)] like capi.global_align"""
        score, beg, nseg = self.score.cpu().numpy(), self.begin.cpu().numpy(), self.nseg.cpu().numpy()
        path = self.path.cpu().numpy()
        out = []
        for i in range(self.n):
            po = int(self.qo[i] + self.ro[i]) + 4 * i
            out.append((int(score[i]), int(beg[i]),
                        "".join("%d%s" % (path[po + k, 1], capi.CIGAR_CHARS[path[po + k, 0]]) for k in range(nseg[i]))))
        return out


class DeviceBgzfBatch:
    """a BGZF file image on the device, tiled to `tile` copies (the feed leg of bench.py): sk_bgzf_inflate_dev over all blocks"""

    def __init__(self, data, device="cuda:0", tile=1):
        data = np.ascontiguousarray(data, np.uint8)
        block_off, out_off = capi.bgzf_scan(data)
        nb = len(block_off) - 1
        self.n_blocks = nb * tile
        self.in_bytes = len(data) * tile
        self.out_bytes = int(out_off[-1]) * tile
        self.data = torch.from_numpy(data.copy()).to(device).repeat(tile)
        k = torch.arange(tile, dtype=torch.int64)[:, None]
        self.block_off = torch.cat([(torch.from_numpy(block_off[:-1])[None, :] + k * len(data)).reshape(-1),
                                    torch.tensor([len(data) * tile])]).to(device)
        self.out_off = torch.cat([(torch.from_numpy(out_off[:-1])[None, :] + k * int(out_off[-1])).reshape(-1),
                                  torch.tensor([int(out_off[-1]) * tile])]).to(device)
        self.out = torch.empty(self.out_bytes, dtype=torch.uint8, device=device)
        self.status = torch.empty(self.n_blocks, dtype=torch.int32, device=device)

    def inflate(self):
        capi._check(capi.lib().sk_bgzf_inflate_dev(C.c_void_p(self.data.data_ptr()), C.c_void_p(self.block_off.data_ptr()),
                                                   C.c_void_p(self.out_off.data_ptr()), self.n_blocks, C.c_void_p(self.out.data_ptr()),
                                                   C.c_void_p(self.status.data_ptr()), _stream_ptr()))
        return self.out
