"""SURVEY 8(f)-2 on the GPU: pga_filter_matches (pangraph_amd/csrc/pga_filter.hip) against the CPU restatement oracle/pgo_filter.c
(itself pinned by the reference's unit-test vectors in tests/test_filter_cpu.py)."""
import numpy as np
import pytest

import filterbind as fb
from filterbind import aln
from test_filter_cpu import CG, _core, _exp

pytestmark = pytest.mark.gpu


def test_reference_known_answers_through_the_product(gpu_lib):
    d = gpu_lib.dll
    a = aln(0, 0, 500, (200, 255), 1, 500, (100, 140), CG, quality=10, reverse=0, divergence=0.1)
    assert _core(fb.product_split_filter(d, [a], thr=10, flags=1)) == [
        _exp(0, 500, (203, 220), 1, 500, (100, 118), 14, 21, "6M 3I 3M 4D 5M", 0), _exp(0, 500, (234, 253), 1, 500, (118, 141), 15, 27, "7M 3D 4I 5M 5D 3M", 0)]
    a = aln(0, 0, 500, (200, 256), 1, 500, (100, 141), CG, quality=10, reverse=1, divergence=0.1)
    assert _core(fb.product_split_filter(d, [a], thr=10, flags=1)) == [
        _exp(0, 500, (236, 253), 1, 500, (100, 118), 14, 21, "6M 3I 3M 4D 5M", 1), _exp(0, 500, (203, 222), 1, 500, (118, 141), 15, 27, "7M 3D 4I 5M 5D 3M", 1)]
    a = aln(0, 0, 257, (200, 257), 1, 56, (0, 56), "3I 3D 6M 3I 3M 4D 5M 14I 7M 3D 4I 5M 5D 3M 4I 12D", matches=29, length=84, quality=10, reverse=0, divergence=0.1)
    assert _core(fb.product_split_filter(d, [a], thr=10, flags=1)) == [
        _exp(0, 257, (203, 220), 1, 56, (0, 21), 14, 24, "3D 6M 3I 3M 4D 5M", 0), _exp(0, 257, (234, 257), 1, 56, (21, 44), 15, 31, "7M 3D 4I 5M 5D 3M 4I", 0)]
    cg2 = "3I 3D 6M 3I 3M 4D 5M 14I 7M 3D 4I 5M 5D 3M 4I 5D"
    a = aln(0, 0, 257, (200, 257), 1, 49, (0, 49), cg2, matches=29, length=77, quality=10, reverse=1, divergence=0.1)
    assert _core(fb.product_split_filter(d, [a], thr=10, flags=1)) == [
        _exp(0, 257, (237, 257), 1, 49, (0, 21), 14, 27, "3I 3D 6M 3I 3M 4D 5M", 1), _exp(0, 257, (204, 223), 1, 49, (21, 49), 15, 32, "7M 3D 4I 5M 5D 3M 5D", 1)]
    a = aln(0, 0, 257, (0, 57), 1, 49, (0, 49), cg2, matches=29, length=77, quality=10, reverse=1, divergence=0.1)
    assert _core(fb.product_split_filter(d, [a], thr=10, flags=1)) == [
        _exp(0, 257, (37, 54), 1, 49, (0, 21), 14, 24, "3D 6M 3I 3M 4D 5M", 1), _exp(0, 257, (0, 23), 1, 49, (21, 49), 15, 36, "7M 3D 4I 5M 5D 3M 5D 4I", 1)]
    with pytest.raises(RuntimeError):
        fb.product_split_filter(d, [aln(0, 0, 500, (0, 50), 1, 500, (0, 50), "40M10S")], thr=10, flags=1)
    a0 = aln(0, 0, 500, (100, 200), 1, 500, (200, 300), "100M", matches=100, divergence=0.05)
    a1 = aln(0, 2, 500, (100, 200), 3, 500, (200, 300), "100M", matches=100, divergence=0.02)
    a2 = aln(0, 2, 500, (150, 250), 4, 500, (200, 300), "100M", matches=100, divergence=0.05)
    a3 = aln(0, 5, 500, (100, 200), 6, 500, (200, 300), "100M", matches=100, divergence=0.1)
    assert _core(fb.product_split_filter(d, [a0, a1, a2, a3], alpha=10.0, beta=10.0, flags=2)) == _core([a1, a0])
    assert fb.product_split_filter(d, [], flags=3) == []


def _random_alignments(rng, n, n_groups, n_blocks):
    out = []
    for _ in range(n):
        ops = []
        for _ in range(int(rng.integers(1, 40))):
            k = "MMMM=XIDID"[int(rng.integers(0, 10))]
            L = int(rng.integers(1, 400)) if k in "M=X" else int(rng.choice([1, 2, 5, 30, 99, 100, 101, 250]))
            if ops and ops[-1][1] == k:
                ops[-1] = (ops[-1][0] + L, k)
            else:
                ops.append((L, k))
        cg = "".join(f"{L}{k}" for L, k in ops)
        qspan = sum(L for L, k in ops if k in "M=XI"); rspan = sum(L for L, k in ops if k in "M=XD")
        qpad = [int(x) for x in rng.choice([0, 0, 1, 50, 99, 100, 101, 3000], 2)]
        rpad = [int(x) for x in rng.choice([0, 0, 1, 50, 99, 100, 101, 3000], 2)]
        q, r = [int(x) for x in rng.choice(n_blocks, 2, replace=bool(rng.integers(0, 20) == 0))]
        out.append(aln(int(rng.integers(0, n_groups)), q, qpad[0] + qspan + qpad[1], (qpad[0], qpad[0] + qspan), r, rpad[0] + rspan + rpad[1], (rpad[0], rpad[0] + rspan), cg,
                       matches=sum(L for L, k in ops if k in "M=X"), length=sum(L for L, k in ops), quality=int(rng.integers(0, 61)), reverse=int(rng.integers(0, 2)),
                       divergence=float(rng.choice([0.0, 0.001, 0.02, 0.1, 0.3]))))
    return out


def test_random_alignments_vs_oracle(gpu_lib, oracle_lib):
    rng = np.random.default_rng(71)
    alns = _random_alignments(rng, 3000, 7, 12)
    for thr, alpha, beta in ((100, 100.0, 10.0), (10, 10.0, 10.0), (1, 0.0, 0.0), (250, 1000.0, 50.0)):
        for flags in (1, 2, 3):
            want = fb.oracle_split_filter(oracle_lib.dll, alns, thr=thr, alpha=alpha, beta=beta, flags=flags)
            got = fb.product_split_filter(gpu_lib.dll, alns, thr=thr, alpha=alpha, beta=beta, flags=flags)
            assert got == want, (thr, alpha, beta, flags, len(got), len(want))
            if flags == 3:
                assert 0 < len(got) < len(alns) * 3


def test_match_lists_of_a_build_vs_oracle(gpu_lib, oracle_lib):
    """the real thing: the match lists of every wave of a small simulated build (leaf pairs, joined graphs, merged graphs), split and
    filtered with the reference's defaults"""
    from pangraph_amd import levels, batch
    pop = levels.Population(11, 8, 60000)
    n_acc = 0
    for label, groups, names in pop.build_waves():
        res = batch.align_groups(groups, names, want_rows=True)
        alns = []
        for g, rows in enumerate(res.groups):
            idx = {nm: i for i, nm in enumerate(names[g])}
            for r in rows:
                alns.append(aln(g, idx[r.qname], r.qlen, (r.qs, r.qe), idx[r.tname], r.tlen, (r.rs, r.re), r.cg, matches=r.mlen, length=r.blen, quality=r.mapq,
                                reverse=1 if r.strand == "-" else 0, divergence=r.de))
        want = fb.oracle_split_filter(oracle_lib.dll, alns)
        got = fb.product_split_filter(gpu_lib.dll, alns)
        assert got == want, label
        n_acc += len(got)
    assert n_acc > 20
