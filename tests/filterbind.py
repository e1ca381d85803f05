"""ctypes bindings of the split / filter step: oracle (oracle/pgo_filter.c) and product (pga_filter_matches)"""
import ctypes as C
import re

import numpy as np

from pangraph_amd.batch import pga_match_t

OPS = "MIDNSHP=X"
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def parse_cigar(s):
    return [(int(n) << 4) | OPS.index(o) for n, o in re.findall(r"(\d+)([MIDNSHP=X])", s.replace(" ", ""))]


def cigar_str(ops):
    return "".join(f"{o >> 4}{OPS[o & 15]}" for o in ops)


def aln(group, qry, qlen, qiv, ref, rlen, riv, cigar, matches=0, length=0, quality=0, reverse=0, divergence=0.0):
    """a record in the vocabulary of the reference's tests: Hit::new(BlockId(qry), qlen, qiv) ..."""
    return dict(group=group, qry=qry, qry_len=qlen, qry_start=qiv[0], qry_end=qiv[1], ref=ref, ref_len=rlen, ref_start=riv[0], ref_end=riv[1],
                matches=matches, length=length, quality=quality, reverse=reverse, divergence=divergence, cigar=cigar.replace(" ", ""))


def _pack(alns):
    n = len(alns)
    m = (pga_match_t * max(n, 1))()
    pool = []
    for i, a in enumerate(alns):
        ops = parse_cigar(a["cigar"])
        for k in ("group", "qry", "ref", "qry_len", "qry_start", "qry_end", "ref_len", "ref_start", "ref_end", "matches", "length", "quality", "reverse"):
            setattr(m[i], k, a[k])
        m[i].divergence = a["divergence"]
        m[i].cigar_off = len(pool); m[i].n_cigar = len(ops)
        pool += ops
    cg = (C.c_uint32 * max(len(pool), 1))(*pool)
    return m, cg, len(pool)


def _unpack(n, mp, cp):
    out = []
    for i in range(n):
        r = mp[i]
        out.append(dict(group=r.group, qry=r.qry, qry_len=r.qry_len, qry_start=r.qry_start, qry_end=r.qry_end, ref=r.ref, ref_len=r.ref_len, ref_start=r.ref_start,
                        ref_end=r.ref_end, matches=r.matches, length=r.length, quality=r.quality, reverse=r.reverse, divergence=r.divergence,
                        cigar=cigar_str([cp[r.cigar_off + j] for j in range(r.n_cigar)])))
    return out


def oracle_split_filter(dll, alns, thr=100, alpha=100.0, beta=10.0, flags=3):
    m, cg, _ = _pack(alns)
    om = C.POINTER(pga_match_t)(); oc = C.POINTER(C.c_uint32)(); nops = C.c_uint64(0)
    dll.pgo_split_filter.restype = C.c_int64
    dll.pgo_split_filter.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.POINTER(pga_match_t)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint64)]
    n = dll.pgo_split_filter(len(alns), m, cg, thr, alpha, beta, flags, C.byref(om), C.byref(oc), C.byref(nops))
    if n < 0:
        raise RuntimeError("Unexpected CIGAR operation")
    out = _unpack(n, om, oc)
    if om:
        _libc.free(C.cast(om, C.c_void_p))
    if oc:
        _libc.free(C.cast(oc, C.c_void_p))
    return out


def oracle_keep_groups(dll, cigar, thr):
    ops = parse_cigar(cigar)
    cg = (C.c_uint32 * max(len(ops), 1))(*ops)
    g = (C.c_int32 * (2 * len(ops) + 2))()
    dll.pgo_keep_groups.restype = C.c_int
    dll.pgo_keep_groups.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    n = dll.pgo_keep_groups(cg, len(ops), thr, g)
    return [(g[2 * i], g[2 * i + 1]) for i in range(n)]


def oracle_energy2(dll, a, alpha, beta):
    m, _, _ = _pack([a])
    dll.pgo_energy2.restype = C.c_double
    dll.pgo_energy2.argtypes = [C.c_void_p, C.c_double, C.c_double]
    return dll.pgo_energy2(m, alpha, beta)


class FilterParams(C.Structure):
    _fields_ = [("indel_len_threshold", C.c_int32), ("flags", C.c_int32), ("alpha", C.c_double), ("beta", C.c_double)]


def product_split_filter(dll, alns, thr=100, alpha=100.0, beta=10.0, flags=3):
    m, cg, n_ops = _pack(alns)
    fp = FilterParams(thr, flags, alpha, beta)
    out = C.c_void_p()
    dll.pga_filter_matches.restype = C.c_int
    dll.pga_filter_matches.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(FilterParams), C.POINTER(C.c_void_p)]
    if dll.pga_filter_matches(len(alns), m, cg, n_ops, C.byref(fp), C.byref(out)) != 0:
        dll.pga_last_error.restype = C.c_char_p
        raise RuntimeError(dll.pga_last_error().decode())
    dll.pga_result_n_matches.restype = C.c_int64; dll.pga_result_n_matches.argtypes = [C.c_void_p]
    dll.pga_result_matches.restype = C.POINTER(pga_match_t); dll.pga_result_matches.argtypes = [C.c_void_p]
    dll.pga_result_cigars.restype = C.POINTER(C.c_uint32); dll.pga_result_cigars.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    n = dll.pga_result_n_matches(out)
    nops = C.c_uint64(0)
    res = _unpack(n, dll.pga_result_matches(out), dll.pga_result_cigars(out, C.byref(nops)))
    dll.pga_result_free.argtypes = [C.c_void_p]
    dll.pga_result_free(out)
    return res
