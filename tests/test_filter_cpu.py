"""SURVEY 8(f)-2, split_matches + filter_matches + alignment_energy2: the CPU restatement (oracle/pgo_filter.c) against the known-answer
vectors of the reference's own unit tests (split_matches.rs:243-593, energy.rs:86-110, graph_merging.rs:253-375).  CPU only."""
import pytest

import filterbind as fb
from filterbind import aln

CG = "3I 6M 3I 3M 4D 5M 14I 7M 3D 4I 5M 5D 3M 3I"


def _core(rows):
    return [{k: r[k] for k in ("qry", "qry_len", "qry_start", "qry_end", "ref", "ref_len", "ref_start", "ref_end", "matches", "length", "quality", "reverse", "divergence", "cigar")} for r in rows]


def _exp(qry, qlen, qiv, ref, rlen, riv, matches, length, cigar, reverse):
    return dict(qry=qry, qry_len=qlen, qry_start=qiv[0], qry_end=qiv[1], ref=ref, ref_len=rlen, ref_start=riv[0], ref_end=riv[1], matches=matches, length=length,
                quality=10, reverse=reverse, divergence=0.1, cigar=cigar.replace(" ", ""))


def test_keep_groups_known_answer(oracle_lib):
    # split_matches.rs:243-260
    cig = "10I 20D 10M 20I 190D   40M 1D 1I 40M 1I 40M   1D 100I   200M 60I 60D 140M   200D   40M 2I 70M"
    assert fb.oracle_keep_groups(oracle_lib.dll, cig, 100) == [(5, 10), (13, 16), (18, 20)]


def test_split_matches_known_answers(oracle_lib):
    d = oracle_lib.dll
    # :262-330 forward
    a = aln(0, 0, 500, (200, 255), 1, 500, (100, 140), CG, quality=10, reverse=0, divergence=0.1)
    assert _core(fb.oracle_split_filter(d, [a], thr=10, flags=1)) == [
        _exp(0, 500, (203, 220), 1, 500, (100, 118), 14, 21, "6M 3I 3M 4D 5M", 0), _exp(0, 500, (234, 253), 1, 500, (118, 141), 15, 27, "7M 3D 4I 5M 5D 3M", 0)]
    # :332-400 reverse
    a = aln(0, 0, 500, (200, 256), 1, 500, (100, 141), CG, quality=10, reverse=1, divergence=0.1)
    assert _core(fb.oracle_split_filter(d, [a], thr=10, flags=1)) == [
        _exp(0, 500, (236, 253), 1, 500, (100, 118), 14, 21, "6M 3I 3M 4D 5M", 1), _exp(0, 500, (203, 222), 1, 500, (118, 141), 15, 27, "7M 3D 4I 5M 5D 3M", 1)]
    # :402-462 side patches, forward
    a = aln(0, 0, 257, (200, 257), 1, 56, (0, 56), "3I 3D 6M 3I 3M 4D 5M 14I 7M 3D 4I 5M 5D 3M 4I 12D", matches=29, length=84, quality=10, reverse=0, divergence=0.1)
    assert _core(fb.oracle_split_filter(d, [a], thr=10, flags=1)) == [
        _exp(0, 257, (203, 220), 1, 56, (0, 21), 14, 24, "3D 6M 3I 3M 4D 5M", 0), _exp(0, 257, (234, 257), 1, 56, (21, 44), 15, 31, "7M 3D 4I 5M 5D 3M 4I", 0)]
    # :464-526 side patches, reverse, query patch leading
    cg2 = "3I 3D 6M 3I 3M 4D 5M 14I 7M 3D 4I 5M 5D 3M 4I 5D"
    a = aln(0, 0, 257, (200, 257), 1, 49, (0, 49), cg2, matches=29, length=77, quality=10, reverse=1, divergence=0.1)
    assert _core(fb.oracle_split_filter(d, [a], thr=10, flags=1)) == [
        _exp(0, 257, (237, 257), 1, 49, (0, 21), 14, 27, "3I 3D 6M 3I 3M 4D 5M", 1), _exp(0, 257, (204, 223), 1, 49, (21, 49), 15, 32, "7M 3D 4I 5M 5D 3M 5D", 1)]
    # :528-592 side patches, reverse, query patch trailing
    a = aln(0, 0, 257, (0, 57), 1, 49, (0, 49), cg2, matches=29, length=77, quality=10, reverse=1, divergence=0.1)
    assert _core(fb.oracle_split_filter(d, [a], thr=10, flags=1)) == [
        _exp(0, 257, (37, 54), 1, 49, (0, 21), 14, 24, "3D 6M 3I 3M 4D 5M", 1), _exp(0, 257, (0, 23), 1, 49, (21, 49), 15, 36, "7M 3D 4I 5M 5D 3M 5D 4I", 1)]
    with pytest.raises(RuntimeError):
        fb.oracle_split_filter(d, [aln(0, 0, 500, (0, 50), 1, 500, (0, 50), "40M10S")], thr=10, flags=1)      # :62-65 (a leading clip is skipped like a leading indel, :41-46)


def test_energy2_known_answer(oracle_lib):
    # energy.rs:86-110: assert_ulps_eq!(alignment_energy2(..), -12.0) with alpha = beta = 10
    a = aln(0, 3, 100, (0, 50), 4, 200, (120, 200), "10I40M10D", matches=40, length=60, quality=100, reverse=0, divergence=0.02)
    assert fb.oracle_energy2(oracle_lib.dll, a, 10.0, 10.0) == -12.0


def test_filter_matches_known_answer(oracle_lib):
    # graph_merging.rs:306-375: [aln_0, aln_1, aln_2, aln_3] -> [aln_1, aln_0] (aln_2 overlaps aln_1 on block 2, aln_3 has E >= 0)
    a0 = aln(0, 0, 500, (100, 200), 1, 500, (200, 300), "100M", matches=100, divergence=0.05)
    a1 = aln(0, 2, 500, (100, 200), 3, 500, (200, 300), "100M", matches=100, divergence=0.02)
    a2 = aln(0, 2, 500, (150, 250), 4, 500, (200, 300), "100M", matches=100, divergence=0.05)
    a3 = aln(0, 5, 500, (100, 200), 6, 500, (200, 300), "100M", matches=100, divergence=0.1)
    got = fb.oracle_split_filter(oracle_lib.dll, [a0, a1, a2, a3], alpha=10.0, beta=10.0, flags=2)
    assert _core(got) == _core([a1, a0])
    # is_match_compatible, :253-303: [100,200)+[300,400) on block 0, [200,300)+[400,500) on block 1
    acc = [aln(0, 0, 1000, (100, 200), 1, 1000, (200, 300), "100M", matches=500, divergence=0.0), aln(0, 0, 1000, (300, 400), 1, 1000, (400, 500), "100M", matches=499, divergence=0.0)]
    ok = aln(0, 0, 1000, (210, 290), 1, 1000, (310, 390), "90M", matches=80, length=80, quality=10, reverse=1, divergence=0.05)
    no = aln(0, 0, 1000, (310, 390), 1, 1000, (310, 390), "90M", matches=80, length=80, quality=10, reverse=1, divergence=0.05)
    assert len(fb.oracle_split_filter(oracle_lib.dll, acc + [ok], alpha=0.0, beta=0.0, flags=2)) == 3
    assert len(fb.oracle_split_filter(oracle_lib.dll, acc + [no], alpha=0.0, beta=0.0, flags=2)) == 2


def test_groups_are_filtered_apart_and_ties_keep_input_order(oracle_lib):
    a = aln(3, 0, 500, (100, 200), 1, 500, (200, 300), "100M", matches=100, divergence=0.0)
    b = aln(7, 0, 500, (150, 250), 1, 500, (200, 300), "100M", matches=100, divergence=0.0)     # same blocks, other group: no conflict
    c = aln(3, 0, 500, (150, 250), 2, 500, (100, 200), "100M", matches=100, divergence=0.0)       # same energy as a, overlaps it on block 0: a came first
    got = fb.oracle_split_filter(oracle_lib.dll, [a, b, c], alpha=10.0, beta=10.0, flags=2)
    assert [(r["group"], r["qry_start"], r["ref"]) for r in got] == [(3, 100, 1), (7, 150, 1)]
