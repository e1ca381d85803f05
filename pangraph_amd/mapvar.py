"""Python host mirror of the re-alignment entry: `pga_map_variations` (include/pga_align.h) replaces the loop of
MergePromise::solve_promise (packages/pangraph/src/pangraph/reweave.rs:40-94) over map_variations (align/map_variations.rs:39-77), and the
same call inside PangraphBlock::edit_consensus_and_realign (pangraph/pangraph_block.rs:295-332).  ctypes only; the HIP library does the work."""
import ctypes as C

from . import batch


class params_t(C.Structure):
    _fields_ = [("score_match", C.c_int32), ("penalty_mismatch", C.c_int32), ("penalty_gap_open", C.c_int32), ("penalty_gap_extend", C.c_int32),
                ("left_terminal_gaps_free", C.c_int32), ("right_terminal_gaps_free", C.c_int32), ("gap_align_left", C.c_int32),
                ("min_length", C.c_int32), ("max_alignment_attempts", C.c_int32), ("extra_band_width", C.c_int32)]


class sub_t(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("alt", C.c_uint32)]


class del_t(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("len", C.c_uint32)]


class ins_t(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("len", C.c_uint32), ("seq_off", C.c_uint64)]


class job_t(C.Structure):
    _fields_ = [("ref", C.c_char_p), ("qry", C.c_char_p), ("ref_len", C.c_uint32), ("qry_len", C.c_uint32), ("mean_shift", C.c_int32), ("band_width", C.c_uint32)]


class res_t(C.Structure):
    _fields_ = [("status", C.c_int32), ("score", C.c_int32), ("attempts", C.c_int32), ("hit_boundary", C.c_int32),
                ("n_subs", C.c_uint32), ("n_dels", C.c_uint32), ("n_inss", C.c_uint32), ("n_ins_bases", C.c_uint32),
                ("sub_off", C.c_uint64), ("del_off", C.c_uint64), ("ins_off", C.c_uint64)]


def params(min_length=1, max_alignment_attempts=4, extra_band_width=5, **kw):
    """map_variations' parameters (map_variations.rs:45-52 over NextalignParams::default(), params.rs:142-170; PangraphBuildArgs defaults
    build_args.rs:76-85)"""
    d = dict(score_match=3, penalty_mismatch=1, penalty_gap_open=6, penalty_gap_extend=0, left_terminal_gaps_free=1, right_terminal_gaps_free=1, gap_align_left=1,
             min_length=min_length, max_alignment_attempts=max_alignment_attempts, extra_band_width=extra_band_width)
    d.update(kw)
    return params_t(**d)


def map_variations(jobs, p=None, dll=None):
    """jobs: [(ref, qry, mean_shift, band_width)] -> list of dicts as above (the product: pga_map_variations, include/pga_align.h)"""
    p = p or params()
    dll = dll or batch.lib()
    n = len(jobs)
    keep = [(j[0].encode(), j[1].encode()) for j in jobs]
    cache = {}
    J = (job_t * max(n, 1))()
    for i, (j, (rb, qb)) in enumerate(zip(jobs, keep)):
        rb = cache.setdefault(rb, rb)                       # jobs onto the same consensus share one buffer, as the caller's would
        J[i].ref = rb; J[i].qry = qb; J[i].ref_len = len(rb); J[i].qry_len = len(qb); J[i].mean_shift = j[2]; J[i].band_width = j[3]
    R = (res_t * max(n, 1))()
    subs = C.POINTER(sub_t)(); dels = C.POINTER(del_t)(); inss = C.POINTER(ins_t)(); iseq = C.POINTER(C.c_char)()
    dll.pga_map_variations.restype = C.c_int
    dll.pga_map_variations.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.POINTER(sub_t)), C.POINTER(C.POINTER(del_t)), C.POINTER(C.POINTER(ins_t)), C.POINTER(C.POINTER(C.c_char))]
    dll.pga_last_error.restype = C.c_char_p
    dll.pga_free.argtypes = [C.c_void_p]
    rc = dll.pga_map_variations(n, J, C.byref(p), R, C.byref(subs), C.byref(dels), C.byref(inss), C.byref(iseq))
    if rc != 0:
        raise batch.PgaError(dll.pga_last_error().decode())
    out = []
    for i in range(n):
        r = R[i]
        out.append(dict(status=r.status, score=r.score, attempts=r.attempts, hit_boundary=r.hit_boundary,
                        subs=[(subs[r.sub_off + k].pos, chr(subs[r.sub_off + k].alt)) for k in range(r.n_subs)],
                        dels=[(dels[r.del_off + k].pos, dels[r.del_off + k].len) for k in range(r.n_dels)],
                        inss=[(inss[r.ins_off + k].pos, C.string_at(C.addressof(iseq.contents) + inss[r.ins_off + k].seq_off, inss[r.ins_off + k].len).decode()) for k in range(r.n_inss)]))
    for ptr in (subs, dels, inss, iseq):
        if ptr:
            dll.pga_free(C.cast(ptr, C.c_void_p))
    return out


def shard_jobs(jobs, world):
    """jobs of one call split over `world` ranks, balanced by band cells (rows x band width): [[job index, ...] per rank].  Jobs are
    independent (one per member sequence), so a multi-GPU host needs no collective: every rank calls map_variations on its share and the
    edits are gathered like match lists.  Deterministic (largest first, ties by index), identical on every rank."""
    cost = [(max(1, len(j[0])) * (2 * (j[3] + 5) + 1), i) for i, j in enumerate(jobs)]
    order = sorted(cost, key=lambda t: (-t[0], t[1]))
    load = [0] * world
    out = [[] for _ in range(world)]
    for c, i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i); load[r] += c
    for r in range(world):
        out[r].sort()
    return out
