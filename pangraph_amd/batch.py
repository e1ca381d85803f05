"""Python binding of the native batch entry (include/pga_align.h) -- host-side mirror of
`align_with_minimap2_lib` (packages/pangraph/src/align/minimap2_lib/align_with_minimap2_lib.rs:15-85) for MANY
block sets at once: one call aligns every group (one group == one `find_matches` call) of a guide-tree level.

The HIP library is the only implementation: if libpgalign.so is missing or sees no device this module raises."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

from .mm2ffi import PafRow, MM_CIGAR_STR

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PGA_LIB") or os.path.join(_HERE, "libpgalign.so")      # (PGA_LIB: a development build of the same library, e.g. dev/pipe_prof.sh)


class pga_params_t(C.Structure):
    _fields_ = [("sensitivity", C.c_int32), ("kmer_length", C.c_int32), ("indel_len_threshold", C.c_int32), ("n_threads", C.c_int32)]


class pga_match_t(C.Structure):
    _fields_ = [("group", C.c_int32), ("qry", C.c_int32), ("ref", C.c_int32), ("qry_len", C.c_int32), ("qry_start", C.c_int32), ("qry_end", C.c_int32),
                ("ref_len", C.c_int32), ("ref_start", C.c_int32), ("ref_end", C.c_int32), ("matches", C.c_int32), ("length", C.c_int32),
                ("quality", C.c_int32), ("reverse", C.c_int32), ("align", C.c_int32), ("n_ambi", C.c_int32), ("inv", C.c_int32),
                ("divergence", C.c_double), ("cigar_off", C.c_uint64), ("n_cigar", C.c_uint32), ("pad", C.c_uint32)]


class pga_stats_t(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("upload", "sketch", "index", "seed", "chain", "align", "total", "n_bases", "n_minimizers", "n_anchors",
                                          "n_dp_jobs", "n_dp_cells", "n_matches", "n_dp_bases")] + \
               [("kern_ms", C.c_double * 16), ("kern_launches", C.c_double * 16), ("kern_alg_bytes", C.c_double * 16), ("kern_cells", C.c_double * 16),
                ("aligned_span", C.c_double)]


KERNELS = ("k_sketch_tiles", "k_chain_fast", "k_bt_list+k_bt_walk", "k_extd2_fast", "k_extd2_wide<256>", "k_ll_i16", "k_rs_init+k_rs_pass+k_rs_small",
           "k_gapfill_band", "k_extd2_wide<512>", "k_extd2_wide<1024>", "index build (sorts + CSR kernels)", "seeding kernels + anchor sort", "k_wstrips+k_bstrips+k_approx_strips", "k_extd2_lanes", "k_ext_pipe", "-")
# what bounds each slot: HBM traffic (scan / sort / hash work) or the integer DP recurrences (VALU + LDS issue; no MFMA)
KERNEL_BOUND = ("hbm", "hbm", "hbm", "dp", "dp", "dp", "hbm", "dp", "dp", "dp", "hbm", "hbm", "dp", "dp", "dp", "-")


class PgaError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PgaError(f"{LIB_PATH} is missing: build it with __graft_entry__.build(); there is no CPU fallback")
        d = C.CDLL(LIB_PATH)
        d.pga_align_groups.restype = C.c_int
        d.pga_align_groups.argtypes = [C.POINTER(pga_params_t), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
        d.pga_batch_create.restype = C.c_int
        d.pga_batch_create.argtypes = [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
        d.pga_batch_align.restype = C.c_int
        d.pga_batch_align.argtypes = [C.c_void_p, C.POINTER(pga_params_t), C.POINTER(C.c_void_p)]
        d.pga_batch_align_shard.restype = C.c_int
        d.pga_batch_align_shard.argtypes = [C.c_void_p, C.POINTER(pga_params_t), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        d.pga_batch_free.argtypes = [C.c_void_p]
        d.pga_busy_begin.restype = C.c_int
        d.pga_busy_begin.argtypes = []
        d.pga_busy_end.restype = C.c_int
        d.pga_busy_end.argtypes = [C.POINTER(C.c_double), C.c_int32]
        d.pga_batch_derive.restype = C.c_int
        d.pga_batch_derive.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
        d.pga_result_n_matches.restype = C.c_int64
        d.pga_result_n_matches.argtypes = [C.c_void_p]
        d.pga_result_matches.restype = C.POINTER(pga_match_t)
        d.pga_result_matches.argtypes = [C.c_void_p]
        d.pga_result_cigars.restype = C.POINTER(C.c_uint32)
        d.pga_result_cigars.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        d.pga_result_stats.restype = C.POINTER(pga_stats_t)
        d.pga_result_stats.argtypes = [C.c_void_p]
        d.pga_result_free.argtypes = [C.c_void_p]
        d.pga_last_error.restype = C.c_char_p
        d.pga_device_count.restype = C.c_int
        d.pga_set_device.restype = C.c_int
        _lib = d
    return _lib


def device_count() -> int:
    return lib().pga_device_count()


def set_device(dev: int) -> None:
    if lib().pga_set_device(dev) != 0:
        raise PgaError(lib().pga_last_error().decode())


class PreparedBatch:
    """Flat C arrays of a list of groups, built once (so a benchmark can time only the library call).
    Sequences may be str, bytes or 1-D uint8 numpy arrays; arrays are passed by address (no copy: they must stay alive and
    contiguous, which slices of a contiguous 1-D array are)."""

    def __init__(self, groups: Sequence[Sequence[str]], names: Optional[Sequence[Sequence[str]]] = None):
        self.n_groups = len(groups)
        if self.n_groups and len(groups[0]) and hasattr(groups[0][0], "ctypes"):
            self._init_arrays(groups, names)
            return
        flat, flat_names, off = [], [], [0]
        for gi, g in enumerate(groups):
            nm = names[gi] if names is not None else [str(i) for i in range(len(g))]
            if len(nm) != len(g):
                raise ValueError("Number of sequences and number of sequence names is expected to be the same")
            for s, n in zip(g, nm):
                flat.append(s if isinstance(s, bytes) else s.encode())
                flat_names.append(n.encode())
            off.append(len(flat))
        n = len(flat)
        self.names = [x.decode() for x in flat_names]
        self._keep = (flat, flat_names)
        self.seqs = (C.c_char_p * n)(*flat)
        self.cnames = (C.c_char_p * n)(*flat_names)
        self.lens = (C.c_uint32 * n)(*[len(b) for b in flat])
        self.off = (C.c_int64 * (self.n_groups + 1))(*off)
        self.total_bases = sum(len(b) for b in flat)


def _init_arrays(self, groups, names):
    flat, flat_names, off = [], [], [0]
    for gi, g in enumerate(groups):
        nm = names[gi] if names is not None else [str(i) for i in range(len(g))]
        if len(nm) != len(g):
            raise ValueError("Number of sequences and number of sequence names is expected to be the same")
        flat.extend(g)
        flat_names.extend(n.encode() for n in nm)
        off.append(len(flat))
    n = len(flat)
    for a in flat:
        if a.dtype.itemsize != 1 or (a.ndim != 1) or (n and a.strides[0] != 1):
            raise ValueError("sequence arrays must be contiguous 1-D uint8")
    self.names = [x.decode() for x in flat_names]
    self._keep = (flat, flat_names)
    self.seqs = C.cast((C.c_void_p * n)(*[a.ctypes.data for a in flat]), C.POINTER(C.c_char_p))
    self.cnames = (C.c_char_p * n)(*flat_names)
    self.lens = (C.c_uint32 * n)(*[len(a) for a in flat])
    self.off = (C.c_int64 * (self.n_groups + 1))(*off)
    self.total_bases = sum(len(a) for a in flat)


PreparedBatch._init_arrays = _init_arrays


@dataclass
class BatchResult:
    groups: List[List[PafRow]]
    stats: dict
    raw_matches: Optional[object] = None  # packed pga_match_t[] as a uint8 numpy VIEW of the result (for the multi-GPU gather)
    raw_cigars: Optional[object] = None   # the CIGAR pool, same
    _handle: Optional[object] = None      # keeps the native result alive while the views are in use

    def close(self):
        if self._handle is not None:
            lib().pga_result_free(self._handle)
            self._handle = None
            self.raw_matches = self.raw_cigars = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stats_dict(st) -> dict:
    out = {}
    for n, t in pga_stats_t._fields_:
        v = getattr(st, n)
        out[n] = list(v) if hasattr(v, "__len__") else v
    return out


class ResidentBatch:
    """Sequences of a PreparedBatch copied to HBM once (pga_batch_create); align() runs the whole hot path on them."""

    def __init__(self, pb: PreparedBatch, derive_from: Optional["ResidentBatch"] = None):
        """derive_from: `pb.src[i]` is the index (in hand-over order) of sequence i in that resident batch and `pb.seqs[i]` is NULL: the bases
        are copied device to device (pga_batch_derive), nothing crosses PCIe."""
        self.pb = pb
        self.h = C.c_void_p()
        if derive_from is not None:
            rc = lib().pga_batch_derive(derive_from.h, pb.n_groups, pb.off, pb.seqs, pb.src, pb.lens, pb.cnames, C.byref(self.h))
        else:
            rc = lib().pga_batch_create(pb.n_groups, pb.off, pb.seqs, pb.lens, pb.cnames, C.byref(self.h))
        if rc != 0:
            raise PgaError(lib().pga_last_error().decode())

    def align(self, sensitivity: int = 10, kmer_length: Optional[int] = None, indel_len_threshold: int = 100, n_threads: int = 0,
              want_rows: bool = False, want_raw: bool = False, shard: Optional[tuple] = None) -> BatchResult:
        """shard = (i, n): map only the i-th of n contiguous query ranges of every group (all sequences are indexed): pga_batch_align_shard"""
        d = lib()
        p = pga_params_t(sensitivity, kmer_length or 0, indel_len_threshold, n_threads)
        out = C.c_void_p()
        rc = d.pga_batch_align(self.h, C.byref(p), C.byref(out)) if shard is None else d.pga_batch_align_shard(self.h, C.byref(p), int(shard[0]), int(shard[1]), C.byref(out))
        if rc != 0:
            raise PgaError(d.pga_last_error().decode())
        return _unpack(self.pb, out, want_rows, want_raw)

    def close(self):
        if self.h:
            lib().pga_batch_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DerivedBatch(ResidentBatch):
    """The batch of the next self-merge round (pga_batch_derive): `groups` holds sequences (str / bytes) or ints -- an int i means "the i-th
    sequence of `old`, as it lies on the device" (numbered in the order `old` was handed over); only the others are uploaded."""

    def __init__(self, old: ResidentBatch, groups, names):
        flat, src, lens, flat_names, off = [], [], [], [], [0]
        old_lens = [old.pb.lens[i] for i in range(sum(1 for _ in old.pb.names))]
        for g, nm in zip(groups, names):
            for s, n in zip(g, nm):
                if isinstance(s, int):
                    flat.append(None); src.append(s); lens.append(old_lens[s])
                else:
                    b = s if isinstance(s, bytes) else s.encode()
                    flat.append(b); src.append(-1); lens.append(len(b))
                flat_names.append(n.encode())
            off.append(len(flat))
        n = len(flat)

        class _PB:
            pass
        pb = _PB()
        pb.n_groups = len(groups); pb.names = [x.decode() for x in flat_names]; pb._keep = (flat, flat_names)
        pb.seqs = (C.c_char_p * n)(*flat); pb.cnames = (C.c_char_p * n)(*flat_names)
        pb.lens = (C.c_uint32 * n)(*lens); pb.off = (C.c_int64 * (len(groups) + 1))(*off); pb.total_bases = sum(lens)
        self.pb = pb
        self.h = C.c_void_p()
        srcs = (C.c_int64 * n)(*src)
        if lib().pga_batch_derive(old.h, pb.n_groups, pb.off, pb.seqs, srcs, pb.lens, pb.cnames, C.byref(self.h)) != 0:
            raise PgaError(lib().pga_last_error().decode())


def align_prepared(pb: PreparedBatch, sensitivity: int = 10, kmer_length: Optional[int] = None, indel_len_threshold: int = 100,
                   n_threads: int = 0, want_rows: bool = True) -> BatchResult:
    d = lib()
    p = pga_params_t(sensitivity, kmer_length or 0, indel_len_threshold, n_threads)
    out = C.c_void_p()
    rc = d.pga_align_groups(C.byref(p), pb.n_groups, pb.off, pb.seqs, pb.lens, pb.cnames, C.byref(out))
    if rc != 0:
        raise PgaError(d.pga_last_error().decode())
    return _unpack(pb, out, want_rows, False)


def _unpack(pb: PreparedBatch, out, want_rows: bool, want_raw: bool) -> BatchResult:
    d = lib()
    try:
        st = d.pga_result_stats(out).contents
        stats = _stats_dict(st)
        groups: List[List[PafRow]] = [[] for _ in range(pb.n_groups)]
        if want_rows:
            n = d.pga_result_n_matches(out)
            m = d.pga_result_matches(out)
            nops = C.c_uint64()
            cg = d.pga_result_cigars(out, C.byref(nops))
            for i in range(n):
                r = m[i]
                b = pb.off[r.group]
                cigar = "".join(f"{cg[r.cigar_off + j] >> 4}{MM_CIGAR_STR[cg[r.cigar_off + j] & 0xf]}" for j in range(r.n_cigar))
                groups[r.group].append(PafRow(qname=pb.names[b + r.qry], qlen=r.qry_len, qs=r.qry_start, qe=r.qry_end, strand="-" if r.reverse else "+",
                                              tname=pb.names[b + r.ref], tlen=r.ref_len, rs=r.ref_start, re=r.ref_end, mlen=r.matches, blen=r.length,
                                              mapq=r.quality, AS=r.align, de=r.divergence, cg=cigar, n_ambi=r.n_ambi, inv=r.inv))
        if want_raw:
            import numpy as np
            n = d.pga_result_n_matches(out)
            nops = C.c_uint64()
            cg = d.pga_result_cigars(out, C.byref(nops))
            raw_m = np.ctypeslib.as_array(C.cast(d.pga_result_matches(out), C.POINTER(C.c_uint8)), shape=(n * C.sizeof(pga_match_t),)) if n else np.zeros(0, np.uint8)
            raw_c = np.ctypeslib.as_array(C.cast(cg, C.POINTER(C.c_uint8)), shape=(nops.value * 4,)) if nops.value else np.zeros(0, np.uint8)
            res = BatchResult(groups, stats, raw_m, raw_c, out)    # the views point into the native result: freed by close()
            out = None
            return res
        return BatchResult(groups, stats)
    finally:
        if out is not None:
            d.pga_result_free(out)


class pga_filter_params_t(C.Structure):
    _fields_ = [("indel_len_threshold", C.c_int32), ("flags", C.c_int32), ("alpha", C.c_double), ("beta", C.c_double)]


def filter_result(res: BatchResult, pb: PreparedBatch, indel_len_threshold: int = 100, alpha: float = 100.0, beta: float = 10.0, split: bool = True,
                  filt: bool = True, want_rows: bool = False) -> BatchResult:
    """SURVEY 8(f)-2 (graph_merging.rs:95-128): drop self matches + split_matches (`split`), filter_matches (`filt`) of a result obtained with
    want_raw=True, on the device; returns a raw result (accepted matches only) that can go into dist.gather_matches."""
    d = lib()
    if res._handle is None:
        raise PgaError("filter_result needs a result obtained with want_raw=True")
    d.pga_result_filter.restype = C.c_int
    d.pga_result_filter.argtypes = [C.c_void_p, C.POINTER(pga_filter_params_t), C.POINTER(C.c_void_p)]
    fp = pga_filter_params_t(indel_len_threshold, (1 if split else 0) | (2 if filt else 0), alpha, beta)
    out = C.c_void_p()
    if d.pga_result_filter(res._handle, C.byref(fp), C.byref(out)) != 0:
        raise PgaError(d.pga_last_error().decode())
    return _unpack(pb, out, want_rows, True)


def align_groups(groups, names=None, **kw) -> BatchResult:
    return align_prepared(PreparedBatch(groups, names), **kw)


def busy_begin() -> None:
    """Opens the busy-interval log of the library (pga_busy_begin): see busy_end."""
    if lib().pga_busy_begin() != 0:
        raise PgaError(lib().pga_last_error().decode())


def busy_end() -> dict:
    """Closes the log: {kernel family: ms of the union of its launch intervals} + "any": the union over all families + "intervals"."""
    n = len(KERNELS)
    out = (C.c_double * (n + 1))()
    rc = lib().pga_busy_end(out, n + 1)
    if rc < 0:
        raise PgaError(lib().pga_last_error().decode())
    d = {KERNELS[i]: out[i] for i in range(n) if KERNELS[i] != "-" and out[i] > 0}
    d["any"] = out[n]
    d["intervals"] = rc
    return d
