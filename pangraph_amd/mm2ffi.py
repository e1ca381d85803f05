"""ctypes binding of the minimap2-sys C-ABI (the drop-in boundary).

The reference's Rust crate `minimap2-sys` binds these symbols with bindgen over
packages/minimap2-sys/minimap2.h; the struct layouts below restate
packages/minimap2-sys/minimap2/minimap.h:64-181 field by field (sizes 24/248/80/80/24 B checked in
tests/test_abi.py).  The class works against any shared object exporting that ABI: the product
(`libpgalign.so`, the HIP backend) and, from tests only, the two checkers under oracle/.

Host-side mirror of packages/minimap2/src/{options_args,options,index,map}.rs:
`Minimap2Args` -> `Mm2Lib.make_options`, `Minimap2Index::new` -> `Mm2Lib.index`,
`Minimap2Mapper::run_map` -> `Mm2Index.map`, `Minimap2PafRow::from_raw` -> `PafRow`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

MM_F_NO_DIAG = 0x001
MM_F_NO_DUAL = 0x002
MM_F_CIGAR = 0x004
MM_F_OUT_CG = 0x020
MM_F_NO_LJOIN = 0x400
MM_F_ALL_CHAINS = 0x800000
MM_F_RMQ = 0x80000000
MM_I_HPC = 0x1
MM_CIGAR_STR = "MIDNSHP=XB"


class mm_idxopt_t(C.Structure):  # minimap.h:123-127
    _fields_ = [("k", C.c_short), ("w", C.c_short), ("flag", C.c_short), ("bucket_bits", C.c_short),
                ("mini_batch_size", C.c_int64), ("batch_size", C.c_uint64)]


class mm_mapopt_t(C.Structure):  # minimap.h:129-181
    _fields_ = [
        ("flag", C.c_int64), ("seed", C.c_int), ("sdust_thres", C.c_int), ("max_qlen", C.c_int),
        ("bw", C.c_int), ("bw_long", C.c_int), ("max_gap", C.c_int), ("max_gap_ref", C.c_int),
        ("max_frag_len", C.c_int), ("max_chain_skip", C.c_int), ("max_chain_iter", C.c_int),
        ("min_cnt", C.c_int), ("min_chain_score", C.c_int),
        ("chain_gap_scale", C.c_float), ("chain_skip_scale", C.c_float),
        ("rmq_size_cap", C.c_int), ("rmq_inner_dist", C.c_int), ("rmq_rescue_size", C.c_int),
        ("rmq_rescue_ratio", C.c_float), ("mask_level", C.c_float), ("mask_len", C.c_int),
        ("pri_ratio", C.c_float), ("best_n", C.c_int), ("alt_drop", C.c_float),
        ("a", C.c_int), ("b", C.c_int), ("q", C.c_int), ("e", C.c_int), ("q2", C.c_int), ("e2", C.c_int),
        ("sc_ambi", C.c_int), ("noncan", C.c_int), ("junc_bonus", C.c_int),
        ("zdrop", C.c_int), ("zdrop_inv", C.c_int), ("end_bonus", C.c_int), ("min_dp_max", C.c_int),
        ("min_ksw_len", C.c_int), ("anchor_ext_len", C.c_int), ("anchor_ext_shift", C.c_int),
        ("max_clip_ratio", C.c_float), ("rank_min_len", C.c_int), ("rank_frac", C.c_float),
        ("pe_ori", C.c_int), ("pe_bonus", C.c_int),
        ("mid_occ_frac", C.c_float), ("q_occ_frac", C.c_float),
        ("min_mid_occ", C.c_int32), ("max_mid_occ", C.c_int32), ("mid_occ", C.c_int32),
        ("max_occ", C.c_int32), ("max_max_occ", C.c_int32), ("occ_dist", C.c_int32),
        ("mini_batch_size", C.c_int64), ("max_sw_mat", C.c_int64), ("cap_kalloc", C.c_int64),
        ("split_prefix", C.c_char_p),
    ]


class mm_extra_t(C.Structure):  # minimap.h:77-83 (flexible cigar[] follows the header)
    _fields_ = [("capacity", C.c_uint32), ("dp_score", C.c_int32), ("dp_max", C.c_int32), ("dp_max2", C.c_int32),
                ("ambi_strand", C.c_uint32),  # n_ambi:30, trans_strand:2
                ("n_cigar", C.c_uint32)]


class mm_reg1_t(C.Structure):  # minimap.h:85-104
    _fields_ = [("id", C.c_int32), ("cnt", C.c_int32), ("rid", C.c_int32), ("score", C.c_int32),
                ("qs", C.c_int32), ("qe", C.c_int32), ("rs", C.c_int32), ("re", C.c_int32),
                ("parent", C.c_int32), ("subsc", C.c_int32), ("as_", C.c_int32),
                ("mlen", C.c_int32), ("blen", C.c_int32), ("n_sub", C.c_int32), ("score0", C.c_int32),
                ("bits", C.c_uint32),  # mapq:8 split:2 rev:1 inv:1 sam_pri:1 proper_frag:1 pe_thru:1 seg_split:1 seg_id:8 split_inv:1 is_alt:1 strand_retained:1 dummy:5
                ("hash", C.c_uint32), ("div", C.c_float), ("p", C.POINTER(mm_extra_t))]


class mm_idx_seq_t(C.Structure):  # minimap.h:58-63
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint64), ("len", C.c_uint32), ("is_alt", C.c_uint32)]


class mm_idx_t(C.Structure):  # minimap.h:65-75
    _fields_ = [("b", C.c_int32), ("w", C.c_int32), ("k", C.c_int32), ("flag", C.c_int32),
                ("n_seq", C.c_uint32), ("index", C.c_int32), ("n_alt", C.c_int32),
                ("seq", C.POINTER(mm_idx_seq_t)), ("S", C.POINTER(C.c_uint32)),
                ("B", C.c_void_p), ("I", C.c_void_p), ("km", C.c_void_p), ("h", C.c_void_p)]


ABI_SYMBOLS = ["mm_set_opt", "mm_idxopt_init", "mm_mapopt_init", "mm_check_opt", "mm_idx_str",
               "mm_mapopt_update", "mm_idx_destroy", "mm_tbuf_init", "mm_tbuf_destroy", "mm_map",
               "mm_event_identity"]


@dataclass
class PafRow:
    """Fields of Minimap2PafRow (packages/minimap2/src/map.rs:263-355) that pangraph consumes
    (align_with_minimap2_lib.rs:88-122), plus the few extra ones useful for parity checks."""
    qname: str
    qlen: int
    qs: int
    qe: int
    strand: str
    tname: str
    tlen: int
    rs: int
    re: int
    mlen: int
    blen: int
    mapq: int
    AS: int
    de: float
    cg: str
    n_ambi: int = 0
    inv: int = 0
    dp_max: int = 0
    cnt: int = 0
    score: int = 0

    def key(self):
        return (self.qname, self.qlen, self.qs, self.qe, self.strand, self.tname, self.tlen, self.rs, self.re,
                self.mlen, self.blen, self.mapq, self.AS, self.de, self.cg, self.n_ambi, self.inv)


class Mm2Index:
    def __init__(self, lib: "Mm2Lib", ptr, io: mm_idxopt_t, mo: mm_mapopt_t, keep):
        self.lib, self.ptr, self.io, self.mo, self._keep = lib, ptr, io, mo, keep
        self._tbuf = lib.dll.mm_tbuf_init()

    def close(self):
        if self.ptr:
            self.lib.dll.mm_tbuf_destroy(self._tbuf)
            self.lib.dll.mm_idx_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def map(self, seq: str, name: str) -> List[PafRow]:
        """Minimap2Result::new (map.rs:45-70): mm_map, convert every reg, free like Drop (map.rs:407-420)."""
        dll = self.lib.dll
        n = C.c_int(0)
        bseq, bname = seq.encode(), name.encode()
        regs = dll.mm_map(self.ptr, len(bseq), bseq, C.byref(n), self._tbuf, C.byref(self.mo), bname)
        out = []
        mi = self.ptr.contents
        for i in range(n.value):
            r = regs[i]
            t = mi.seq[r.rid]
            rev = (r.bits >> 10) & 1
            inv = (r.bits >> 11) & 1
            row = PafRow(qname=name, qlen=len(bseq), qs=r.qs, qe=r.qe, strand="-" if rev else "+",
                         tname=t.name.decode(), tlen=t.len, rs=r.rs, re=r.re, mlen=r.mlen, blen=r.blen,
                         mapq=r.bits & 0xff, AS=0, de=0.0, cg="", inv=inv, cnt=r.cnt, score=r.score)
            if r.p:
                p = r.p.contents
                row.AS, row.dp_max, row.n_ambi = p.dp_score, p.dp_max, p.ambi_strand & 0x3fffffff
                cig = C.cast(C.addressof(p) + C.sizeof(mm_extra_t), C.POINTER(C.c_uint32))
                row.cg = "".join(f"{cig[j] >> 4}{MM_CIGAR_STR[cig[j] & 0xf]}" for j in range(p.n_cigar))
                row.de = 1.0 - dll.mm_event_identity(C.byref(r))
                self.lib.libc.free(r.p)
            out.append(row)
        if regs:
            self.lib.libc.free(regs)
        return out


class Mm2Lib:
    def __init__(self, path: str):
        self.path = path
        self.dll = C.CDLL(path, mode=C.RTLD_LOCAL)
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]
        d = self.dll
        d.mm_set_opt.argtypes = [C.c_char_p, C.POINTER(mm_idxopt_t), C.POINTER(mm_mapopt_t)]
        d.mm_set_opt.restype = C.c_int
        d.mm_check_opt.argtypes = [C.POINTER(mm_idxopt_t), C.POINTER(mm_mapopt_t)]
        d.mm_check_opt.restype = C.c_int
        d.mm_idx_str.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
        d.mm_idx_str.restype = C.POINTER(mm_idx_t)
        d.mm_mapopt_update.argtypes = [C.POINTER(mm_mapopt_t), C.POINTER(mm_idx_t)]
        d.mm_mapopt_update.restype = None
        d.mm_idx_destroy.argtypes = [C.POINTER(mm_idx_t)]
        d.mm_idx_destroy.restype = None
        d.mm_tbuf_init.restype = C.c_void_p
        d.mm_tbuf_destroy.argtypes = [C.c_void_p]
        d.mm_tbuf_destroy.restype = None
        d.mm_map.argtypes = [C.POINTER(mm_idx_t), C.c_int, C.c_char_p, C.POINTER(C.c_int), C.c_void_p,
                             C.POINTER(mm_mapopt_t), C.c_char_p]
        d.mm_map.restype = C.POINTER(mm_reg1_t)
        d.mm_event_identity.argtypes = [C.POINTER(mm_reg1_t)]
        d.mm_event_identity.restype = C.c_double

    def make_options(self, preset: Optional[str], k: Optional[int] = None, w: Optional[int] = None,
                     c: bool = True, X: bool = True, s: Optional[int] = None, bucket_bits: Optional[int] = 14):
        """init_options (options.rs:85-138) + init_opts (options_args.rs:273-551) for the fields pangraph sets."""
        io, mo = mm_idxopt_t(), mm_mapopt_t()
        if self.dll.mm_set_opt(None, C.byref(io), C.byref(mo)) != 0:
            raise RuntimeError("minimap2: mm_set_opt(null, ...): failed to set options: incorrect preset")
        if preset is not None and self.dll.mm_set_opt(preset.encode(), C.byref(io), C.byref(mo)) != 0:
            raise RuntimeError("minimap2: mm_set_opt(preset, ...): failed to set options: incorrect preset")
        if k is not None:
            io.k = k
        if w is not None:
            io.w = w
        if c:
            mo.flag |= MM_F_OUT_CG | MM_F_CIGAR
        if s is not None:
            mo.min_dp_max = s
        if X:
            mo.flag |= MM_F_ALL_CHAINS | MM_F_NO_DIAG | MM_F_NO_DUAL | MM_F_NO_LJOIN
        if bucket_bits is not None:
            io.bucket_bits = bucket_bits
        if self.dll.mm_check_opt(C.byref(io), C.byref(mo)) != 0:
            raise RuntimeError("minimap2: mm_check_opt(): options are invalid")
        return io, mo

    def index(self, seqs: Sequence[str], names: Sequence[str], io: mm_idxopt_t, mo: mm_mapopt_t) -> Mm2Index:
        """Minimap2Index::new (index.rs:17-54)."""
        n = len(seqs)
        bs = [s.encode() for s in seqs]
        bn = [s.encode() for s in names]
        sa = (C.c_char_p * n)(*bs)
        na = (C.c_char_p * n)(*bn)
        ptr = self.dll.mm_idx_str(io.w, io.k, io.flag & MM_I_HPC, io.bucket_bits, n, sa, na)
        if not ptr:
            raise RuntimeError("minimap2: failed to create index")
        mo2 = mm_mapopt_t.from_buffer_copy(mo)
        self.dll.mm_mapopt_update(C.byref(mo2), ptr)
        return Mm2Index(self, ptr, io, mo2, (bs, bn, sa, na))

    def align_all(self, seqs: Sequence[str], names: Sequence[str], sensitivity: int = 10,
                  kmer_length: Optional[int] = None, indel_len_threshold: int = 100) -> List[PafRow]:
        """align_with_minimap2_lib_impl (align_with_minimap2_lib.rs:29-85) with queries in input order."""
        preset = {5: "asm5", 10: "asm10", 20: "asm20"}.get(sensitivity)
        if preset is None:
            raise ValueError(f"Unknown sensitivity preset: {sensitivity}")
        if len(seqs) != len(names):
            raise ValueError("Number of sequences and number of sequence names is expected to be the same")
        io, mo = self.make_options(preset, k=kmer_length, c=True, X=True, s=max(indel_len_threshold - 10, 5), bucket_bits=14)
        idx = self.index(seqs, names, io, mo)
        try:
            rows: List[PafRow] = []
            for s, nm in zip(seqs, names):
                rows.extend(idx.map(s, nm))
            return rows
        finally:
            idx.close()
