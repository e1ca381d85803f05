"""SURVEY 8(f)-3, guide tree: the CPU restatement (oracle/pgo_mash.c) against the known-answer vectors the reference's own unit tests
hold (hash.rs:19-27, minimizer.rs:188-210, mash_distance.rs:84-152, neighbor_joining.rs:113-151).  CPU only."""
import ctypes as C

import numpy as np
import pytest

import mashbind as mb


def test_hash_known_answers(oracle_lib):
    # hash.rs:21-24
    for (x, mask), want in (((0, 0), 0), ((123, 0), 0), ((0, 456), 136), ((123, 456), 384)):
        assert mb.oracle_hash(oracle_lib.dll, x, mask) == want


def test_sketch_known_answer(oracle_lib):
    # minimizer.rs:188-210
    seq = "CGATCCTTCGGGAACGTGTGACGCGAAGGTGCATGGGAGATCTCGCATTGCTGTTCTGGACGACGCGAAGAGTACTGCTACTTTCATGTCGCCTACGCCT"
    want = [(9685, 4294967328), (7669, 4294967355), (5583, 4294967359), (3600, 4294967386), (2383, 4294967415), (4791, 4294967427),
            (5338, 4294967451), (2190, 4294967461), (378, 4294967466)]
    assert mb.oracle_sketch(oracle_lib.dll, seq, 1, k=8, w=16) == want
    assert mb.oracle_sketch(oracle_lib.dll, "", 0, k=15, w=100) == []       # minimizer.rs:213-220: the reference reports an error


GENERAL = [
    "CATAGAAGCAGTCCCTGAGCACGACGCGTGTAACAATCGTTTTCAGACCTAGGACGTTAGAATATCGATCGCACGCTACGACCGACGATTAGCCGCACGAGCAAGTCGAAAACCCGAGTTAAGAGGCTGGACGTGATCCTAGACTTCGTC",
    "CATAGAAGCAGTCCCTGAGCACGAGGCGCGCAACAATCGTTTTCAGCCCTAGGACGTTAGAATATTGATCACAAGCTACGACCGACGATTAGCCGCACGAGCAAGTCGACAACCCGAGTTAAGAGGCTGGACGTGATGCTAGACTTCGTC",
    "CATAGAAGCAGTCCCTGAGCATGACGCGCGCAACGATCGTTTTCAGCCCTAGCACGTGAGAATATTGATCACAAGCTACGACCGACGATTAGCCGCACGAGCTAGTCGCCAACCCGAGTAAGGAGGCTGGACGTGATGCTAGACTACGTC",
    "ACATCAAAACTTAAAGTCGGTTACCATCTACAAATGTAGTAAGGGGGATTCTAATGAGAGAAGTGGACTGTGTAGATGGACCCGCTCACCTGCCCAGTATCTTAGTGGCGTATTCAGGATCTGGGAGGATTTGTTATTGCCTATTAGAGA",
    "ACATCAAAACTTAAAGTCGGTTCCCATCTACAAAAGTAGAAAGGGGGATTCTAATGAGAGATGTGGACTGTGTAGATGGACCCGCTAACCTGGCCAGTTTCTTAGTGGCTTAATCAGGATCTGGGAGGATTCGTTACTGCCTATTAGAGA",
    "ACATCAGAACTTAAAGTCGGTTCCTATCTCCAAAAGTATAAAGTGGGATTCTAATGAGAGATGTGGACTGTGTCGATAAACCCGCTAACCTGGCCTGTTTCTTGTTGGCTTAATCAGGATCTGAGAGGATTCGTTACTGCCTAGTAGTGA",
]


def test_mash_distance_known_answers(oracle_lib):
    # mash_distance.rs:84-123 (w = 16, k = 8); assert_eq! on f64: exact
    d = mb.oracle_distance(oracle_lib.dll, GENERAL, k=8, w=16)
    want = np.array([
        [0.0, 1. - 6. / 9., 0.75, 1.0, 1.0, 1.0],
        [1. - 6. / 9., 0.0, 0.5, 1.0, 1.0, 1.0],
        [0.75, 0.5, 0.0, 1.0, 1.0, 1.0],
        [1.0, 1.0, 1.0, 0.0, 0.625, 0.875],
        [1.0, 1.0, 1.0, 0.625, 0.0, 5. / 7.],
        [1.0, 1.0, 1.0, 0.875, 5. / 7., 0.0]])
    assert d.tobytes() == want.tobytes()
    # :133-141 two equal sequences, default parameters; :144-152 a single sequence
    assert mb.oracle_distance(oracle_lib.dll, [GENERAL[0], GENERAL[0]], k=15, w=100).tolist() == [[0., 0.], [0., 0.]]
    assert mb.oracle_distance(oracle_lib.dll, [GENERAL[0][:50]], k=15, w=100).tolist() == [[0.0]]
    with pytest.raises(RuntimeError):
        mb.oracle_distance(oracle_lib.dll, [GENERAL[0], "ACGT"], k=15, w=100)   # no minimizer: the reference panics


WIKI = np.array([[0.0, 5.0, 9.0, 9.0, 8.0], [5.0, 0.0, 10.0, 10.0, 9.0], [9.0, 10.0, 0.0, 8.0, 7.0], [9.0, 10.0, 8.0, 0.0, 3.0], [8.0, 9.0, 7.0, 3.0, 0.0]])


def test_neighbor_joining_known_answers(oracle_lib):
    # neighbor_joining.rs:113-136 (Q matrix) and :138-151 (dist)
    inf = float("inf")
    q = mb.oracle_q_matrix(oracle_lib.dll, WIKI)
    assert q.tolist() == [[inf, -50.0, -38.0, -34.0, -34.0], [-50.0, inf, -38.0, -34.0, -34.0], [-38.0, -38.0, inf, -40.0, -40.0],
                          [-34.0, -34.0, -40.0, inf, -48.0], [-34.0, -34.0, -40.0, -48.0, inf]]
    assert mb.oracle_nj_dist(oracle_lib.dll, WIKI, 0, 1).tolist() == [0., 0., 7., 7., 6.]
    # the joins of the two trees the reference's (disabled) tests spell out, :185-199 and :208-246: ((((A,B),C),D),E) and (((A,H),(((B,E),D),(C,G))),F)
    assert mb.newick(mb.oracle_nj(oracle_lib.dll, WIKI), "ABCDE") == "((((A,B),C),D),E)"
    d8 = np.array([[0, 46, 37, 46, 46, 14, 37, 1], [46, 0, 46, 7, 1, 46, 46, 46], [37, 46, 0, 46, 46, 37, 1, 37], [46, 7, 46, 0, 7, 46, 46, 46],
                   [46, 1, 46, 7, 0, 46, 46, 46], [14, 46, 37, 46, 46, 0, 37, 14], [37, 46, 1, 46, 46, 37, 0, 37], [1, 46, 37, 46, 46, 14, 37, 0]], dtype=np.float64)
    assert mb.newick(mb.oracle_nj(oracle_lib.dll, d8), "ABCDEFGH") == "(((A,H),(((B,E),D),(C,G))),F)"


def test_unrolled_sum_order_differs_from_sequential(oracle_lib):
    """the restated ndarray summation orders are really two different ones (documented as "parity unpinned" in oracle/pgo_mash.c)"""
    rng = np.random.default_rng(3)
    d = rng.random((37, 37)); d = (d + d.T) / 2; np.fill_diagonal(d, 0.0)
    q = mb.oracle_q_matrix(oracle_lib.dll, d)
    assert np.isinf(np.diag(q)).all() and np.isfinite(q[~np.eye(37, dtype=bool)]).all()
    assert not np.array_equal(q, q.T)          # sum_0 and sum_1 round differently
    assert np.allclose(q[~np.eye(37, dtype=bool)], q.T[~np.eye(37, dtype=bool)], rtol=1e-12)
