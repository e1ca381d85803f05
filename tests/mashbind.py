"""ctypes bindings of the guide-tree path: oracle (oracle/pgo_mash.c) and product (pga_mash_* of libpgalign.so)"""
import ctypes as C

import numpy as np


class Mz(C.Structure):
    _fields_ = [("value", C.c_uint64), ("position", C.c_uint64)]


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def oracle_hash(dll, x, mask):
    dll.pgo_mash_hash.restype = C.c_uint64
    dll.pgo_mash_hash.argtypes = [C.c_uint64, C.c_uint64]
    return dll.pgo_mash_hash(x, mask)


def oracle_sketch(dll, seq, sid, k=15, w=100):
    dll.pgo_mash_sketch.restype = C.c_size_t
    dll.pgo_mash_sketch.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.POINTER(Mz)), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    out = C.POINTER(Mz)(); n = C.c_size_t(0); cap = C.c_size_t(0)
    b = seq.encode() if isinstance(seq, str) else bytes(seq)
    dll.pgo_mash_sketch(b, len(b), sid, k, w, C.byref(out), C.byref(n), C.byref(cap))
    res = [(out[i].value, out[i].position) for i in range(n.value)]
    if out:
        _libc.free(C.cast(out, C.c_void_p))
    return res


def _seq_arrays(seqs):
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    n = len(bs)
    return bs, (C.c_char_p * n)(*bs), (C.c_size_t * n)(*[len(b) for b in bs])


def oracle_distance(dll, seqs, k=15, w=100):
    bs, sp, lp = _seq_arrays(seqs)
    n = len(bs)
    d = np.zeros((n, n), dtype=np.float64)
    dll.pgo_mash_distance.restype = C.c_int
    dll.pgo_mash_distance.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_void_p]
    rc = dll.pgo_mash_distance(n, sp, lp, k, w, d.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"no minimizer found for sequence {-rc - 1}")
    return d


def oracle_q_matrix(dll, d):
    d = np.ascontiguousarray(d, dtype=np.float64); m = d.shape[0]
    q = np.zeros_like(d)
    dll.pgo_nj_q_matrix.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    dll.pgo_nj_q_matrix(m, d.ctypes.data, q.ctypes.data)
    return q


def oracle_nj_dist(dll, d, i, j):
    d = np.ascontiguousarray(d, dtype=np.float64); m = d.shape[0]
    out = np.zeros(m, dtype=np.float64)
    dll.pgo_nj_dist.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    dll.pgo_nj_dist(m, d.ctypes.data, i, j, out.ctypes.data)
    return out


def oracle_nj(dll, d):
    d = np.ascontiguousarray(d, dtype=np.float64); n = d.shape[0]
    merges = np.zeros((max(n - 1, 0), 2), dtype=np.int32)
    dll.pgo_nj_tree.restype = C.c_int
    dll.pgo_nj_tree.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    rc = dll.pgo_nj_tree(n, d.ctypes.data, merges.ctypes.data)
    if rc != 0:
        raise RuntimeError("neighbor joining failed")
    return merges


def newick(merges, names):
    n = len(merges) + 1
    txt = {i: names[i] for i in range(n)}
    for t, (a, b) in enumerate(merges):
        txt[n + t] = f"({txt[int(a)]},{txt[int(b)]})"
    return txt[n + len(merges) - 1]


# ---------------------------------------------------------------- product
def product_sketch(dll, seqs, k=15, w=100):
    """per sequence: [(value, position)] in the reference's order"""
    bs, sp, lp32 = _seq_arrays(seqs)
    n = len(bs)
    lens = (C.c_uint32 * n)(*[len(b) for b in bs])
    val = C.POINTER(C.c_uint64)(); pos = C.POINTER(C.c_uint64)(); off = (C.c_uint64 * (n + 1))()
    dll.pga_stage_mash_sketch.restype = C.c_int
    dll.pga_stage_mash_sketch.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
    rc = dll.pga_stage_mash_sketch(n, sp, lens, k, w, C.byref(val), C.byref(pos), off)
    if rc != 0:
        dll.pga_last_error.restype = C.c_char_p
        raise RuntimeError(dll.pga_last_error().decode())
    out = [[(val[i], pos[i]) for i in range(off[s], off[s + 1])] for s in range(n)]
    dll.pga_free.argtypes = [C.c_void_p]
    dll.pga_free(val); dll.pga_free(pos)
    return out


def product_distance(dll, seqs, k=15, w=100):
    bs, sp, _ = _seq_arrays(seqs)
    n = len(bs)
    lens = (C.c_uint32 * n)(*[len(b) for b in bs])
    d = np.zeros((n, n), dtype=np.float64)
    dll.pga_mash_distance.restype = C.c_int
    dll.pga_mash_distance.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_void_p]
    rc = dll.pga_mash_distance(n, sp, lens, k, w, d.ctypes.data)
    if rc != 0:
        dll.pga_last_error.restype = C.c_char_p
        raise RuntimeError(dll.pga_last_error().decode())
    return d


def product_nj(dll, d):
    d = np.ascontiguousarray(d, dtype=np.float64); n = d.shape[0]
    merges = np.zeros((max(n - 1, 0), 2), dtype=np.int32)
    dll.pga_guide_tree_nj.restype = C.c_int
    dll.pga_guide_tree_nj.argtypes = [C.c_int32, C.c_void_p, C.c_void_p]
    rc = dll.pga_guide_tree_nj(n, d.ctypes.data, merges.ctypes.data)
    if rc != 0:
        dll.pga_last_error.restype = C.c_char_p
        raise RuntimeError(dll.pga_last_error().decode())
    return merges


def product_nj_near_ties(dll):
    """(count, first join) of the calling thread's last joining call: pga_nj_near_ties"""
    dll.pga_nj_near_ties.restype = C.c_int32
    dll.pga_nj_near_ties.argtypes = [C.POINTER(C.c_int32)]
    first = C.c_int32(-2)
    return int(dll.pga_nj_near_ties(C.byref(first))), int(first.value)
