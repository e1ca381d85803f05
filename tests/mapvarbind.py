"""ctypes bindings of the re-alignment step: oracle (oracle/pgo_mapvar.c); the product's binding is pangraph_amd/mapvar.py"""
import ctypes as C

import numpy as np

from pangraph_amd.mapvar import del_t, ins_t, job_t, params, params_t, res_t, sub_t  # noqa: F401
from pangraph_amd import mapvar as _mv


class ores_t(C.Structure):
    _fields_ = [("status", C.c_int32), ("score", C.c_int32), ("attempts", C.c_int32), ("hit_boundary", C.c_int32),
                ("n_subs", C.c_uint32), ("n_dels", C.c_uint32), ("n_inss", C.c_uint32), ("n_ins_bases", C.c_uint32)]


def oracle_map_variations(dll, ref, qry, mean_shift, band_width, p=None, want_aln=False):
    """-> dict(status, score, attempts, hit_boundary, subs [(pos, letter)], dels [(pos, len)], inss [(pos, seq)]) (+ qry_aln, ref_aln)"""
    p = p or params()
    rb, qb = ref.encode(), qry.encode()
    n = len(rb) + len(qb) + 2
    subs = (sub_t * n)(); dels = (del_t * n)(); inss = (ins_t * n)(); iseq = C.create_string_buffer(len(qb) + 1)
    aln = C.create_string_buffer(2 * (len(rb) + len(qb)) + 2); alen = C.c_int64(0)
    r = ores_t()
    dll.pgo_map_variations.restype = C.c_int
    dll.pgo_map_variations.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    dll.pgo_map_variations(rb, len(rb), qb, len(qb), mean_shift, band_width, C.byref(p), C.byref(r), subs, dels, inss, iseq, aln, C.byref(alen))
    out = dict(status=r.status, score=r.score, attempts=r.attempts, hit_boundary=r.hit_boundary,
               subs=[(subs[i].pos, chr(subs[i].alt)) for i in range(r.n_subs)], dels=[(dels[i].pos, dels[i].len) for i in range(r.n_dels)],
               inss=[(inss[i].pos, iseq.raw[inss[i].seq_off:inss[i].seq_off + inss[i].len].decode()) for i in range(r.n_inss)])
    if want_aln and r.status == 0:
        L = len(rb) + len(qb)
        out["qry_aln"] = aln.raw[:alen.value].decode(); out["ref_aln"] = aln.raw[L:L + alen.value].decode()
    return out


def product_map_variations(dll, jobs, p=None):
    """jobs: [(ref, qry, mean_shift, band_width)] -> list of dicts as above, through the product (pga_map_variations)"""
    return _mv.map_variations(jobs, p, dll)


def apply_edit(ref, e):
    """Edit::apply (packages/pangraph/src/pangraph/edits.rs:307-329): the query the edits describe"""
    q = list(ref)
    for pos, alt in e["subs"]:
        q[pos] = alt
    for pos, ln in e["dels"]:
        for k in range(pos, pos + ln):
            q[k] = "-"
    for pos, seq in sorted(e["inss"], reverse=True):
        q[pos:pos] = list(seq)
    return "".join(c for c in q if c != "-")


def mutate(rng, ref, snp=0.02, indel=0.004, max_indel=30, n_frac=0.0):
    """a member sequence: the consensus with substitutions, short indels and unknown letters; returns (qry, edit-derived shift bound)"""
    out = []
    i = 0
    L = len(ref)
    while i < L:
        r = rng.random()
        if r < indel / 2:
            i += int(rng.integers(1, max_indel + 1))
            continue
        if r < indel:
            out.append("".join("ACGT"[k] for k in rng.integers(0, 4, int(rng.integers(1, max_indel + 1)))))
        c = ref[i]
        if rng.random() < snp:
            c = "ACGT"[int(rng.integers(0, 4))]
        if n_frac and rng.random() < n_frac:
            c = "N"
        out.append(c)
        i += 1
    return "".join(out)


def random_seq(rng, n):
    return "".join(np.array(list("ACGT"))[rng.integers(0, 4, n)])


# ---- host-side helpers of the callers (packages/pangraph/src/pangraph/edits.rs:418-531), restated for the tests ----
def aligned_count_after(e, p, cons_len):
    total = max(cons_len - p, 0)
    overlap = sum((d[0] + d[1]) - max(p, d[0]) for d in e["dels"] if d[0] + d[1] > p)
    return max(total - overlap, 0)


def _round_half_away(x):
    return int(x + 0.5) if x >= 0 else -int(-x + 0.5)          # f64::round


def band_from_edits(e, cons_len):
    """BandParameters::from_edits (align/map_variations.rs:29-37) = (Edit::aln_mean_shift, Edit::aln_bandwidth)"""
    ac = aligned_count_after(e, 0, cons_len)
    if ac == 0:
        return None
    total = 0
    for pos, seq in e["inss"]:
        total -= len(seq) * aligned_count_after(e, pos, cons_len)
    for pos, ln in e["dels"]:
        total += ln * aligned_count_after(e, pos, cons_len)
    ms = _round_half_away(total / ac)
    tuples = sorted([(pos, -len(seq)) for pos, seq in e["inss"]] + [(pos, ln) for pos, ln in e["dels"]], key=lambda t: t[0])   # sorted_by_key: stable
    bw, cur = 0, 0
    for i, (pos, shift) in enumerate(tuples):
        if i == 0 and pos > 0:
            bw = max(bw, abs(cur - ms))
        cur += shift
        if i == len(tuples) - 1 and (pos == cons_len or (shift > 0 and pos + shift == cons_len)):
            continue
        bw = max(bw, abs(cur - ms))
    return ms, bw


def realign_jobs(consensus, members, majority):
    """PangraphBlock::edit_consensus_and_realign (pangraph/pangraph_block.rs:295-332): the new consensus and one map_variations job per member"""
    new_cons = apply_edit(consensus, majority)
    bms, bbw = band_from_edits(majority, len(consensus))
    jobs = []
    for e in members:
        oms, obw = band_from_edits(e, len(consensus))
        jobs.append((new_cons, apply_edit(consensus, e), oms - bms, obw + bbw))
    return new_cons, jobs
