"""Checks the oracle restatement against the compiled reference (oracle/_ref) on fresh seeded inputs, stage by stage.
CPU-only; skipped when the reference build is unavailable."""
import ctypes as C

import numpy as np
import pytest

import stagebind as sb
from pangraph_amd.synth import evolve_population, random_seq, mutate
from util import rows_to_lists


def test_radix_sort_tie_order(oracle_lib, ref_lib):
    rng = np.random.default_rng(5)
    for n, nkeys in [(10, 3), (64, 5), (65, 4), (300, 7), (5000, 40), (5000, 5000000), (70000, 300)]:
        x = rng.integers(0, nkeys, size=n).astype(np.uint64)
        if n == 5000 and nkeys == 40:
            x = x << np.uint64(33) | (x * np.uint64(977)) % np.uint64(7)      # spread over several byte levels
        y = np.arange(n, dtype=np.uint64)
        a = np.stack([x, y], axis=1).copy()
        b = a.copy()
        ref_lib.dll.radix_sort_128x(C.c_void_p(a.ctypes.data), C.c_void_p(a.ctypes.data + a.nbytes))
        oracle_lib.dll.pgo_radix_sort_128x(C.c_void_p(b.ctypes.data), C.c_void_p(b.ctypes.data + b.nbytes))
        assert (a == b).all(), (n, nkeys)


def test_sketch_random(oracle_lib, ref_lib):
    rng = np.random.default_rng(6)
    for trial in range(60):
        L = int(rng.integers(1, 3000))
        s = random_seq(rng, L)
        if trial % 4 == 0:
            for _ in range(int(rng.integers(1, 5))):
                p = int(rng.integers(0, L)); s[p:p + int(rng.integers(1, 30))] = ord("N")
        if trial % 5 == 0:
            s[: L // 2] = np.tile(np.frombuffer(b"AT", np.uint8), L)[: L // 2]
        w, k = int(rng.integers(1, 40)), int(rng.integers(1, 29))
        q = s.tobytes().decode()
        assert sb.oracle_sketch(oracle_lib.dll, q, w, k, 3) == sb.ref_sketch(ref_lib.dll, q, w, k, 3), (L, w, k)


def test_chain_stage(oracle_lib, ref_lib):
    seqs = evolve_population(11, 3, 20000, snp=0.02, indel=0.002, n_inv=1, n_ins=1, n_del=1, max_event=3000)
    seqs[2] = seqs[2] + seqs[2][3000:6000] + seqs[0][1000:4000]          # repeats -> equal anchors, tied priorities
    names = ["30", "4", "5"]
    io, mo = oracle_lib.make_options("asm10", s=90)
    idx = oracle_lib.index(seqs, names, io, mo)
    mo = idx.mo
    per_q = sb.oracle_anchors(oracle_lib.dll, seqs, names, mo, io.w, io.k)
    assert sum(len(a) for a, _ in per_q) > 1000
    for anchors, _ in per_q:
        assert sb.oracle_chain(oracle_lib.dll, anchors, mo, io.k) == sb.ref_chain(ref_lib.dll, anchors, mo, io.k)
    idx.close()


def test_ksw_banded_random(oracle_lib, ref_lib):
    """narrow bands: the cells outside the band that SSE keeps computing leak into the result"""
    rng = np.random.default_rng(7)
    mat = sb.simple_mat(1, 9, 1)
    for trial in range(40):
        L = int(rng.integers(30, 300))
        t = random_seq(rng, L)
        q = mutate(rng, t, snp=0.05, indel=0.02)
        w = int(rng.integers(1, 40))
        flag = [0x40, 0x40 | 0x02 | 0x80, 0, 0x08][trial % 4]
        zdrop = [200, 50, -1][trial % 3]
        a = sb.oracle_extd2(oracle_lib.dll, sb.nt4(q.tobytes().decode()), sb.nt4(t.tobytes().decode()), mat, 16, 2, 41, 1, w, zdrop, -1, flag)
        b = sb.ref_extd2(ref_lib.dll, sb.nt4(q.tobytes().decode()), sb.nt4(t.tobytes().decode()), mat, 16, 2, 41, 1, w, zdrop, -1, flag)
        for k in ("zdropped", "reach_end", "cigar", "score") + (() if flag & 8 else ("max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q")):
            assert a[k] == b[k], (k, trial, L, w, flag)


def test_e2e_synthetic_population(oracle_lib, ref_lib):
    seqs = evolve_population(12, 4, 30000, snp=0.01, indel=0.001, n_inv=1, n_ins=1, n_del=1, max_event=5000)
    names = ["101", "22", "3", "44"]
    a = rows_to_lists(ref_lib.align_all(seqs, names))
    b = rows_to_lists(oracle_lib.align_all(seqs, names))
    assert a == b and len(a) >= 6


def _lb_problem(rng, trial, w):
    """an extension towards a block end: 1 ... 64 target bases, a query that runs on; tails built to put as many matches as possible far from the diagonal"""
    tl = int(rng.integers(1, 65)) if trial % 8 else (1, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64)[trial // 8 % 12]
    t = random_seq(rng, tl)
    ql = w + 2 * tl + int(rng.integers(0, 1200))
    kind = trial % 8
    head = mutate(rng, t, snp=0.08 * (trial % 3), indel=0.03 * (trial % 2)) if kind != 6 else random_seq(rng, tl)
    body = random_seq(rng, max(1, ql - len(head)))
    if kind == 1:
        for p in range(int(rng.integers(0, 60)), len(body) - tl - 1, max(tl + int(rng.integers(0, 5)), 5)):
            body[p:p + tl] = t
    elif kind == 2:
        t = np.full(tl, ord("ACGT"[trial % 4]), dtype=np.uint8); head = t.copy(); body[: int(len(body) * rng.random())] = t[0]
    elif kind == 3 and tl > 6:
        t = t.copy(); t[tl // 3: tl // 3 + 2] = ord("N"); body[100:130] = ord("N")
    elif kind == 4:
        reps = np.tile(t, 1 + len(body) // max(1, tl))[: len(body)]; body[:] = reps
    elif kind == 5:                                                   # copies of the target's LAST bases: what the last column meets again and again
        k = max(1, tl // 2)
        for p in range(0, len(body) - k, k + 1):
            body[p:p + k] = t[-k:]
    elif kind == 7:                                                   # dinucleotide repeats in both
        t = np.tile(np.frombuffer(b"AC", np.uint8), tl)[:tl].copy(); head = t.copy(); body[:] = np.tile(np.frombuffer(b"AC", np.uint8), len(body))[: len(body)]
    q = np.concatenate([head, body])[: max(ql, 1)]
    return sb.nt4(q.tobytes().decode()), sb.nt4(t.tobytes().decode())


def test_length_bound_stop_rule_against_the_full_sweep(oracle_lib, ref_lib):
    """pangraph_amd/csrc/pga_dp.h (length-bound stop): the product ends an extension whose target window is the <= 64 bases before a block end once
    alignment length alone bounds every cell the reference would still visit below ez.max / ez.mte.  The restatement evaluates the same rule beside
    its full sweep (pgo_ksw.c, the observer) and counts the problems whose record changed after the rule had closed: none may, over windows built
    against the bound (copies of the target all along the query, homopolymers, tandem and dinucleotide repeats, N runs), three presets, bands 64 ...
    1501, both gap alignments.  A sample of the same problems is held against the compiled reference (ksw_extd2_sse) so that the observer sits on
    the reference's own sweep."""
    rng = np.random.default_rng(99)
    cnt = (C.c_longlong * 3)()
    oracle_lib.dll.pgo_lb_counters(cnt, 1)
    EXTZ, RIGHT, REV = 0x40, 0x02, 0x80
    n = 0
    for pname, (ma, mb, q1, e1, q2, e2) in {"asm10": (1, 9, 16, 2, 41, 1), "asm5": (1, 19, 39, 3, 81, 1), "asm20": (1, 4, 6, 2, 26, 1)}.items():
        mat = sb.simple_mat(ma, mb, 1)
        for trial in range(8000):
            w = (1501, 751, 300, 64, 100, 1501, 200, 1000)[trial % 8]
            qn, tn = _lb_problem(rng, trial, w)
            flag = (EXTZ, EXTZ | RIGHT | REV, EXTZ | REV, EXTZ | RIGHT, 0, RIGHT)[trial % 6] | 0x01          # score only: the sweep, not the traceback, is what is checked
            zd = (200, 400, 100, -1)[trial % 4]
            o = sb.oracle_extd2(oracle_lib.dll, qn, tn, mat, q1, e1, q2, e2, w, zd, -1, flag)
            if trial % 16 == 0:
                r = sb.ref_extd2(ref_lib.dll, qn, tn, mat, q1, e1, q2, e2, w, zd, -1, flag)
                for k in ("zdropped", "max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score"):
                    assert o[k] == r[k], (pname, trial, k)
            n += 1
    oracle_lib.dll.pgo_lb_counters(cnt, 0)
    assert cnt[1] == 0, f"the rule closed on {cnt[1]} records that the full sweep still changed"
    assert cnt[0] > 0.5 * n, (cnt[0], n)          # and it does close (unless the band is too narrow to leave it room: w = 64, 100)
    assert cnt[2] > 200 * cnt[0]                   # ... long before the sweep ends
