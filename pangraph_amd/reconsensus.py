"""Python host mirror of the reconsensus entry: `pga_reconsensus` (include/pga_align.h) replaces analyze_blocks_for_reconsensus
and the per-block work of reconsensus_graph (packages/pangraph/src/reconsensus/reconsensus.rs:32-126) for all updated blocks of a merge at
once.  ctypes only; the HIP library does the work."""
import ctypes as C

from . import batch
from .mapvar import del_t, ins_t, params, params_t, res_t, sub_t  # noqa: F401


class rc_block_t(C.Structure):
    _fields_ = [("consensus", C.c_char_p), ("cons_len", C.c_uint32), ("n_members", C.c_uint32)]


class rc_member_t(C.Structure):
    _fields_ = [("n_subs", C.c_uint32), ("n_dels", C.c_uint32), ("n_inss", C.c_uint32)]


class rc_block_res_t(C.Structure):
    _fields_ = [("kind", C.c_int32), ("cons_len", C.c_uint32), ("cons_off", C.c_uint64), ("n_subs", C.c_uint32), ("n_dels", C.c_uint32), ("n_inss", C.c_uint32),
                ("sub_off", C.c_uint64), ("del_off", C.c_uint64), ("ins_off", C.c_uint64)]


class rc_out_t(C.Structure):
    _fields_ = [("blocks", C.POINTER(rc_block_res_t)), ("members", C.POINTER(res_t)),
                ("subs", C.POINTER(sub_t)), ("dels", C.POINTER(del_t)), ("inss", C.POINTER(ins_t)), ("ins_seq", C.POINTER(C.c_char)),
                ("m_subs", C.POINTER(sub_t)), ("m_dels", C.POINTER(del_t)), ("m_inss", C.POINTER(ins_t)), ("m_ins_seq", C.POINTER(C.c_char)),
                ("cons", C.POINTER(C.c_char))]


def _edit(subs, dels, inss, iseq, r_subs, r_dels, r_inss):
    so, ns = r_subs
    do, nd = r_dels
    io, ni = r_inss
    base = C.addressof(iseq.contents) if iseq else 0
    return {"inss": [(inss[io + k].pos, C.string_at(base + inss[io + k].seq_off, inss[io + k].len).decode()) for k in range(ni)],
            "dels": [(dels[do + k].pos, dels[do + k].len) for k in range(nd)],
            "subs": [(subs[so + k].pos, chr(subs[so + k].alt)) for k in range(ns)]}


def reconsensus(blocks, p=None, dll=None):
    """blocks: [(consensus, [edit, ...])] with edit = {"subs": [(pos, letter)], "dels": [(pos, len)], "inss": [(pos, seq)]}, members in the
    reference's BTreeMap order.  -> [(kind, new consensus, [member edit, ...], majority edit, [member status, ...])] per block"""
    p = p or params()
    dll = dll or batch.lib()
    nb = len(blocks)
    keep = [b[0].encode() for b in blocks]
    B = (rc_block_t * max(nb, 1))()
    n_mem = sum(len(b[1]) for b in blocks)
    M = (rc_member_t * max(n_mem, 1))()
    subs, dels, inss, letters = [], [], [], bytearray()
    m = 0
    for i, (cons, members) in enumerate(blocks):
        B[i].consensus = keep[i]; B[i].cons_len = len(keep[i]); B[i].n_members = len(members)
        for e in members:
            M[m].n_subs = len(e["subs"]); M[m].n_dels = len(e["dels"]); M[m].n_inss = len(e["inss"])
            subs += [(pos, ord(a)) for pos, a in e["subs"]]
            dels += list(e["dels"])
            for pos, seq in e["inss"]:
                inss.append((pos, len(seq), len(letters)))
                letters += seq.encode()
            m += 1
    S = (sub_t * max(len(subs), 1))(*[sub_t(*x) for x in subs])
    D = (del_t * max(len(dels), 1))(*[del_t(*x) for x in dels])
    I = (ins_t * max(len(inss), 1))(*[ins_t(*x) for x in inss])
    L = C.create_string_buffer(bytes(letters), max(len(letters), 1))
    out = rc_out_t()
    dll.pga_reconsensus.restype = C.c_int
    dll.pga_reconsensus.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    dll.pga_rc_free.argtypes = [C.c_void_p]
    dll.pga_last_error.restype = C.c_char_p
    if dll.pga_reconsensus(nb, B, M, S, D, I, L, C.byref(p), C.byref(out)) != 0:
        raise batch.PgaError(dll.pga_last_error().decode())
    res = []
    m = 0
    try:
        cbase = C.addressof(out.cons.contents)
        for i, (cons, members) in enumerate(blocks):
            r = out.blocks[i]
            maj = _edit(out.m_subs, out.m_dels, out.m_inss, out.m_ins_seq, (r.sub_off, r.n_subs), (r.del_off, r.n_dels), (r.ins_off, r.n_inss))
            mem, status = [], []
            for _ in members:
                v = out.members[m]
                mem.append(_edit(out.subs, out.dels, out.inss, out.ins_seq, (v.sub_off, v.n_subs), (v.del_off, v.n_dels), (v.ins_off, v.n_inss)))
                status.append(v.status)
                m += 1
            res.append((r.kind, C.string_at(cbase + r.cons_off, r.cons_len).decode(), mem, maj, status))
    finally:
        dll.pga_rc_free(C.byref(out))
    return res
