"""SURVEY 8(f)-3 on the GPU: pga_stage_mash_sketch / pga_mash_distance / pga_guide_tree_nj (pangraph_amd/csrc/pga_mash.hip) against the
CPU restatement oracle/pgo_mash.c (itself pinned by the reference's unit-test vectors in tests/test_mash_cpu.py)."""
import numpy as np
import pytest

import mashbind as mb
from test_mash_cpu import GENERAL, WIKI

pytestmark = pytest.mark.gpu


def _rand(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def test_reference_known_answers_through_the_product(gpu_lib):
    seq = "CGATCCTTCGGGAACGTGTGACGCGAAGGTGCATGGGAGATCTCGCATTGCTGTTCTGGACGACGCGAAGAGTACTGCTACTTTCATGTCGCCTACGCCT"
    want = [(9685, 4294967328), (7669, 4294967355), (5583, 4294967359), (3600, 4294967386), (2383, 4294967415), (4791, 4294967427),
            (5338, 4294967451), (2190, 4294967461), (378, 4294967466)]
    got = mb.product_sketch(gpu_lib.dll, ["ACGTACGTACGTACGTACGTAGCTAGCTAGCTTTGACA", seq], k=8, w=16)   # (the vector's sequence has id 1)
    assert got[1] == want
    d = mb.product_distance(gpu_lib.dll, GENERAL, k=8, w=16)
    want_d = np.array([[0.0, 1. - 6. / 9., 0.75, 1.0, 1.0, 1.0], [1. - 6. / 9., 0.0, 0.5, 1.0, 1.0, 1.0], [0.75, 0.5, 0.0, 1.0, 1.0, 1.0],
                       [1.0, 1.0, 1.0, 0.0, 0.625, 0.875], [1.0, 1.0, 1.0, 0.625, 0.0, 5. / 7.], [1.0, 1.0, 1.0, 0.875, 5. / 7., 0.0]])
    assert d.tobytes() == want_d.tobytes()
    assert mb.product_distance(gpu_lib.dll, [GENERAL[0], GENERAL[0]]).tolist() == [[0., 0.], [0., 0.]]
    assert mb.product_distance(gpu_lib.dll, [GENERAL[0][:50]]).tolist() == [[0.0]]
    with pytest.raises(RuntimeError):
        mb.product_distance(gpu_lib.dll, [GENERAL[0], "ACGT"])
    assert mb.newick(mb.product_nj(gpu_lib.dll, WIKI), "ABCDE") == "((((A,B),C),D),E)"


def test_sketch_vs_oracle(gpu_lib, oracle_lib):
    """chunk boundaries (a lane per 4096 bases with a w+k warm-up), N runs, sequences shorter than k / w+k, homopolymers and tandem
    arrays (equal hashes inside a window), the extreme parameters"""
    rng = np.random.default_rng(11)
    seqs = []
    for L in (0, 5, 14, 15, 16, 113, 114, 115, 116, 4095, 4096, 4097, 4096 + 114, 4096 + 115, 8192, 12289, 30011):
        seqs.append(_rand(rng, L))
    s = list(_rand(rng, 20000))
    for p in (100, 4090, 4096, 4100, 8191, 12000):
        for q in range(p, p + int(rng.integers(1, 40))):
            s[q] = "N"
    seqs.append("".join(s))
    seqs.append("A" * 9000)
    seqs.append(("ACGTTGCA" * 3 + "T") * 700)
    seqs.append(_rand(rng, 300) * 40)
    seqs.append("acgtnACGURYKM" * 900)
    for (k, w) in ((15, 100), (8, 16), (31, 255), (1, 1), (3, 7), (19, 10)):
        got = mb.product_sketch(gpu_lib.dll, seqs, k=k, w=w)
        for i, sq in enumerate(seqs):
            want = mb.oracle_sketch(oracle_lib.dll, sq, i, k=k, w=w)
            assert got[i] == want, (k, w, i, len(sq), len(got[i]), len(want))


def test_distance_and_tree_vs_oracle(gpu_lib, oracle_lib):
    from pangraph_amd import levels
    pop = levels.Population(7, 24, 40000)
    seqs = [pop.genomes[v].tobytes().decode() for v in pop.leaves]
    for (k, w) in ((15, 100), (11, 20)):
        want = mb.oracle_distance(oracle_lib.dll, seqs, k=k, w=w)
        got = mb.product_distance(gpu_lib.dll, seqs, k=k, w=w)
        assert got.tobytes() == want.tobytes()
        assert (mb.product_nj(gpu_lib.dll, got) == mb.oracle_nj(oracle_lib.dll, want)).all()


def test_neighbor_joining_vs_oracle_random_matrices(gpu_lib, oracle_lib):
    rng = np.random.default_rng(19)
    for n in (2, 3, 4, 8, 9, 57, 300, 1000):
        d = rng.random((n, n)); d = (d + d.T) / 2; np.fill_diagonal(d, 0.0)
        if n == 57:
            d = np.round(d * 4) / 4            # ties: the first minimum in row-major order decides
            d = (d + d.T) / 2; np.fill_diagonal(d, 0.0)
        got = mb.product_nj(gpu_lib.dll, d); want = mb.oracle_nj(oracle_lib.dll, d)
        assert (got == want).all(), n


def test_near_tie_flag_of_the_joining_step(gpu_lib, oracle_lib):
    """neighbor_joining.rs:77-96 takes the smallest Q; the sums behind Q are f64 sums in ndarray's order as restated (not pinned against ndarray).
    pga_nj_near_ties counts the joins in which ANOTHER pair's Q lies within the reordering error of the chosen one: none for a matrix of
    well separated distances, some as soon as distances repeat (quantised matrix: exact ties, the first minimum decides)."""
    rng = np.random.default_rng(41)
    n = 200
    d = rng.random((n, n)); d = (d + d.T) / 2; np.fill_diagonal(d, 0.0)
    got = mb.product_nj(gpu_lib.dll, d)
    assert (got == mb.oracle_nj(oracle_lib.dll, d)).all()
    assert mb.product_nj_near_ties(gpu_lib.dll) == (0, -1)
    q = np.round(d * 4) / 4; q = (q + q.T) / 2; np.fill_diagonal(q, 0.0)
    got = mb.product_nj(gpu_lib.dll, q)
    assert (got == mb.oracle_nj(oracle_lib.dll, q)).all()
    cnt, first = mb.product_nj_near_ties(gpu_lib.dll)
    assert cnt > 0 and 0 <= first < n - 1
    # two pairs whose Q differ by a few parts in 10^15: flagged although the values are not equal (six nodes: the joins at m = 4 and
    # m = 3, tied in exact arithmetic in every tree, are not counted)
    e = np.full((6, 6), 4.0); np.fill_diagonal(e, 0.0)
    e[0, 1] = e[1, 0] = 1.0; e[2, 3] = e[3, 2] = 1.0 + 4e-15; e[4, 5] = e[5, 4] = 3.0
    mb.product_nj(gpu_lib.dll, e)
    assert mb.product_nj_near_ties(gpu_lib.dll) == (1, 0)
    # the big-matrix variant of the kernel (state in device memory) carries the flag too
    n = 2100
    d = rng.random((n, n)); d = (d + d.T) / 2; np.fill_diagonal(d, 0.0)
    mb.product_nj(gpu_lib.dll, d)
    assert mb.product_nj_near_ties(gpu_lib.dll)[0] == 0


def test_bit_matrix_slabs_do_not_change_the_counts(gpu_lib, oracle_lib, monkeypatch):
    rng = np.random.default_rng(23)
    seqs = [_rand(rng, 20000) for _ in range(5)]
    seqs += [s[:10000] + _rand(rng, 10000) for s in seqs[:3]]
    want = mb.oracle_distance(oracle_lib.dll, seqs, k=11, w=8)
    monkeypatch.setenv("PGA_MASH_SLAB_MB", "0.001")     # many slabs of the value axis
    assert mb.product_distance(gpu_lib.dll, seqs, k=11, w=8).tobytes() == want.tobytes()
