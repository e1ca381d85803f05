"""SURVEY 8(f)-1, map_variations / align_with_nextclade: the CPU restatement (oracle/pgo_mapvar.c) against the known-answer vectors of
the reference's own unit tests (align_with_nextclade.rs:92-311, map_variations.rs:190-365, align.rs:191-250).  CPU only."""
import json
import os

import numpy as np

import mapvarbind as mb

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _strip(qry_aln, ref_aln):
    return "".join(q for q, r in zip(qry_aln, ref_aln) if r != "-")


def test_nuc_alphabet_and_match_matrix(oracle_lib):
    # the 16 x 16 table of score_matrix_nuc.rs:6-26 and the letter order of nuc.rs:10-30 (fixture made by tests/golden/make_golden_mapvar.py)
    g = json.load(open(os.path.join(GOLDEN, "nuc_matrix.json")))
    d = oracle_lib.dll
    for i, c in enumerate(g["letters"]):
        assert d.pgo_to_nuc(ord(c)) == i
    for c in "acgtnXZ*. ":
        assert d.pgo_to_nuc(ord(c)) == -1
    assert [[d.pgo_nuc_match(x, y) for y in range(16)] for x in range(16)] == g["matrix"]


def test_align_with_nextclade_known_answers(oracle_lib):
    d = oracle_lib.dll
    p = mb.params(min_length=3, max_alignment_attempts=3)
    # align_with_nextclade.rs:92-152 (general case; BandParameters::new(0, 4 + EXTRA_BANDWIDTH))
    ref = "CTTGGAGGTTCCGTGGCTAGATAACAGAACATTCTTGGAATGCTGATCTTTATAAGCTCATGCGACACTTCGCATGGTGAGCCTTTGT"
    qry = "CTTGGAGGTTCCGTGGCTATAAAGATAACAGAACATTCTTGGAATGCTGATCAAGCTCATGGGACANNTCGCATGGTGGACAGCCTTTGT"
    r = mb.oracle_map_variations(d, ref, qry, 0, 4, p, want_aln=True)
    assert r["status"] == 0 and r["hit_boundary"] == 0
    assert r["ref_aln"] == "CTTGGAGGTTCCGTGGCTA----GATAACAGAACATTCTTGGAATGCTGATCTTTATAAGCTCATGCGACACTTCGCATGGTG---AGCCTTTGT"
    assert _strip(r["qry_aln"], r["ref_aln"]) == "CTTGGAGGTTCCGTGGCTAGATAACAGAACATTCTTGGAATGCTGATC-----AAGCTCATGGGACANNTCGCATGGTGAGCCTTTGT"
    assert r["subs"] == [(62, "G"), (67, "N"), (68, "N")]
    assert r["dels"] == [(48, 5)]
    assert r["inss"] == [(18 + 1, "TAAA"), (78 + 1, "GAC")]
    # :154-206 (N in the reference: substitutions, never matches)
    ref = "TGGTGCTGCAGCTTATTATGTGGNNNNNTTTTCTATTAAAATATAATGAAA"
    qry = "TGGTGCTGCAGCTTATTATGTGGAGGACTTTTCTATTAAAATATAATGAAA"
    r = mb.oracle_map_variations(d, ref, qry, 0, 0, p, want_aln=True)
    assert (r["qry_aln"], r["ref_aln"]) == (qry, ref) and r["hit_boundary"] == 0
    assert r["subs"] == [(23, "A"), (24, "G"), (25, "G"), (26, "A"), (27, "C")] and r["dels"] == [] and r["inss"] == []
    # :208-272 (edge case)
    ref = "TGGTGCTGCNNNNNATTATGTGGGTTATCTTCAACCTTTTTTTAAAATATAATGAAAATGGAACCATTACAGATGCTNNNNNNNNTGCACTTGACCCTCTC"
    qry = "TGGTGCTGCAGCTTATTATGTGGGTTATCTTCAACCTTTTTTTAAAATATAATGAAAATGGAACCATTACAGATGCTGTAGACTGTGCACTTGACCCTCTC"
    r = mb.oracle_map_variations(d, ref, qry, 0, 0, p, want_aln=True)
    assert (r["qry_aln"], r["ref_aln"]) == (qry, ref) and r["hit_boundary"] == 0
    assert r["subs"] == [(9, "A"), (10, "G"), (11, "C"), (12, "T"), (13, "T"), (77, "G"), (78, "T"), (79, "A"), (80, "G"), (81, "A"), (82, "C"), (83, "T"), (84, "G")]
    assert r["dels"] == [] and r["inss"] == []
    # :274-311 (nothing alignable inside the band: the whole reference deleted, the whole query inserted behind it)
    r = mb.oracle_map_variations(d, "A" * 37, "G" * 18, 70, 0, p, want_aln=True)
    assert r["ref_aln"] == "A" * 37 + "-" * 18 and _strip(r["qry_aln"], r["ref_aln"]) == "-" * 37
    assert r["subs"] == [] and r["dels"] == [(0, 37)] and r["inss"] == [(36 + 1, "G" * 18)] and r["hit_boundary"] == 0


MAPVAR_KATS = [
    # map_variations.rs:190-231
    ("ACTTTGCGTCTGATAGCTTAGCGGATATTTACTGTA", "ACTAGATTGAGTCTGATAGCTTAGCGGATATTGTA", -2, 3, [(6, "A")], [(29, 4)], [(3, "AGA")]),
    # :233-275 (leading and trailing deletions come behind the internal ones: align_with_nextclade.rs:46-64)
    ("ACACTGATTTCGTCCCTTAGGTACTCTACACTGTAGCCTA", "CTGATTTAGTCCCTTAGGGGTTACTCTACACTGTAG", 2, 2, [(10, "A")], [(0, 3), (36, 4)], [(21, "GGT")]),
    # :277-319
    ("ACACTGATTTCGTCCCTTAGGTACTCTACACTGTAGCCTA", "CCTGACACTGATTTAGTCCTAGGGGTTACTCTACACCGTAGCCTAGCCGCCG", -4, 2, [(10, "A"), (31, "C")], [(15, 2)],
     [(0, "CCTG"), (21, "GGT"), (40, "GCCGCCG")]),
    # :321-365
    ("CGCCCTACTACAAGAGGGAACTTTTTTTTTAAGTATAGCCACAATAGCTGG", "CGCCCTACTACAAGAGGGAACGGGGGGGGGGGGGAAGTATAGCCACAATAGCTGG", -2, 11, [], [(21, 9)], [(21, "G" * 13)]),
]


def test_map_variations_known_answers(oracle_lib):
    for ref, qry, ms, bw, subs, dels, inss in MAPVAR_KATS:
        r = mb.oracle_map_variations(oracle_lib.dll, ref, qry, ms, bw)
        assert r["status"] == 0
        assert (r["subs"], r["dels"], r["inss"]) == (subs, dels, inss)
        assert mb.apply_edit(ref, r) == qry


def E(inss=(), dels=(), subs=()):
    return dict(inss=list(inss), dels=list(dels), subs=list(subs))


# reconsensus.rs:185-329: blocks 1 and 3 of the reference's reconsensus tests -- consensus, member edits, majority edits (:338-358), and what
# edit_consensus_and_realign must return (:400-428): new consensus and the members' edits after map_variations
RECONSENSUS_KATS = [
    ("AGGACTTCGATCTATTCGGAGAA",
     [E([(17, "TTTT")], [(5, 2)], [(1, "T"), (17, "A")]), E([], [(5, 2)], [(1, "T"), (10, "C")]), E([], [(5, 2), (16, 2)], [(1, "T"), (10, "C")]),
      E([], [(9, 3)], [(1, "C"), (17, "A")]), E([(5, "AA")], [(5, 2)], [(17, "A")])],
     E([], [(5, 2)], [(1, "T"), (17, "A")]),
     "ATGACCGATCTATTCAGAGAA",
     [E([(15, "TTTT")]), E([], [], [(8, "C"), (15, "G")]), E([], [(14, 2)], [(8, "C")]), E([(5, "TT")], [(7, 3)], [(1, "C")]), E([(5, "AA")], [], [(1, "G")])]),
    ("GCCTCTTCCCGACCACGCGTTACAACATGGGACAGGCCTGCGCTTGAGGC",
     [E([], [(19, 4)], [(5, "A")]), E([(35, "AA"), (50, "TT")], [(20, 3)], [(5, "A")]), E([], [], [(14, "G"), (27, "G")]), E([(50, "TT")], [(20, 3)], [(5, "A")]), E([(50, "TT")])],
     E([(50, "TT")], [(20, 3)], [(5, "A")]),
     "GCCTCATCCCGACCACGCGTAACATGGGACAGGCCTGCGCTTGAGGCTT",
     [E([], [(19, 1), (47, 2)]), E([(32, "AA")]), E([(20, "TAC")], [(47, 2)], [(5, "T"), (14, "G"), (24, "G")]), E(), E([(20, "TAC")], [], [(5, "T")])]),
]


def test_band_parameters_known_answers():
    # map_variations.rs:88-188 (host-side helper of the callers, restated in tests/mapvarbind.py)
    assert mb.band_from_edits(E(), 10) == (0, 0)
    assert mb.band_from_edits(E([(0, "AAA")]), 10) == (-3, 0)
    assert mb.band_from_edits(E([], [(0, 2)]), 10) == (2, 0)
    assert mb.band_from_edits(E([(9, "C")]), 10) == (0, 1)
    assert mb.band_from_edits(E([(2, "CCC")], [(2, 3)]), 25) == (0, 3)
    assert mb.band_from_edits(E([(8, "CCC"), (20, "GG")], [(2, 3), (15, 2)], [(5, "A"), (10, "T")]), 25) == (1, 2)
    for ref, qry, ms, bw, subs, dels, inss in MAPVAR_KATS:             # :214-218 and the like: the examples state their own band
        assert mb.band_from_edits(E(inss, dels, subs), len(ref)) == (ms, bw)


def test_reconsensus_realign_known_answers(oracle_lib):
    # the re-alignment inside reconsensus is the same map_variations: reconsensus.rs:400-428
    for cons, members, majority, new_cons, expected in RECONSENSUS_KATS:
        got_cons, jobs = mb.realign_jobs(cons, members, majority)
        assert got_cons == new_cons
        for (r, q, ms, bw), exp in zip(jobs, expected):
            g = mb.oracle_map_variations(oracle_lib.dll, r, q, ms, bw)
            assert g["status"] == 0 and E(g["inss"], g["dels"], g["subs"]) == exp


def test_simplestripe_band_hit_known_answers(oracle_lib):
    d = oracle_lib.dll
    # align.rs:191-222: NextalignParams::default() with one attempt, the band handed over as it is (no extra width)
    core = "TTGGCCCCGGTGCTGTCCGTCAACACGTCGTCGTCCGGCGACCTACCTGGTCTCAAAGGAGGTTTTGTTAAATGAATTAGATGGGTAAGGTTACCACGTCA"
    ref, qry = core + "A" * 30, "G" * 30 + core
    p = mb.params(min_length=100, max_alignment_attempts=1, extra_band_width=0)
    assert mb.oracle_map_variations(d, ref, qry, -30, 1, p)["hit_boundary"] == 0
    assert mb.oracle_map_variations(d, ref, qry, 0, 31, p)["hit_boundary"] == 0
    assert mb.oracle_map_variations(d, ref, qry, 0, 30, p)["hit_boundary"] == 1
    # :224-250
    p = mb.params(min_length=3, max_alignment_attempts=1, extra_band_width=0)
    r = mb.oracle_map_variations(d, "A" * 37, "G" * 18, 70, 0, p, want_aln=True)
    assert (r["ref_aln"], r["qry_aln"], r["score"], r["hit_boundary"]) == ("A" * 37 + "-" * 18, "-" * 37 + "G" * 18, 0, 0)
    # align.rs:42-46: a query shorter than min_length is an error; nuc.rs:99-121: so is a letter outside the alphabet
    assert mb.oracle_map_variations(d, "ACGT", "AC", 0, 0, p)["status"] == 1
    assert mb.oracle_map_variations(d, "ACGT", "ACgT", 0, 0, p)["status"] == 2


def test_random_members_round_trip(oracle_lib):
    # size-independent property: the edits map the consensus back onto the member sequence, whatever the band did
    rng = np.random.default_rng(7)
    n_retry = 0
    for it in range(60):
        L = int(rng.integers(1, 900))
        ref = mb.random_seq(rng, L)
        qry = mb.mutate(rng, ref, snp=0.03, indel=0.01, max_indel=int(rng.integers(1, 40)), n_frac=0.01 if it % 3 == 0 else 0.0) or "A"
        ms = int(rng.integers(-3, 4)); bw = int(rng.integers(0, 12))
        p = mb.params(gap_align_left=it % 2, penalty_gap_extend=it % 4 == 3, left_terminal_gaps_free=it % 5 != 4, right_terminal_gaps_free=it % 7 != 6)
        r = mb.oracle_map_variations(oracle_lib.dll, ref, qry, ms, bw, p)
        assert r["status"] == 0 and mb.apply_edit(ref, r) == qry
        n_retry += r["attempts"] > 1
    assert n_retry > 0


def test_job_sharding_is_balanced_and_complete():
    # multi-GPU: jobs are independent, every rank takes a share (pangraph_amd/mapvar.py:shard_jobs); no collective on this path
    from pangraph_amd.mapvar import shard_jobs
    rng = np.random.default_rng(5)
    jobs = [("A" * int(rng.integers(1, 20000)), "A", 0, int(rng.integers(0, 60))) for _ in range(500)]
    for world in (1, 2, 8):
        sh = shard_jobs(jobs, world)
        assert sorted(i for s in sh for i in s) == list(range(len(jobs)))
        cost = [sum(len(jobs[i][0]) * (2 * (jobs[i][3] + 5) + 1) for i in s) for s in sh]
        assert max(cost) <= 1.05 * (sum(cost) / world) + max(len(j[0]) * (2 * (j[3] + 5) + 1) for j in jobs)
        assert sh == shard_jobs(jobs, world)


def _overlap_score(ref, qry, match=3, mismatch=1, gap_open=6):
    """An independent statement of what the aligner maximises with its default switches (gap extension 0, terminal gaps free on both
    sides of both sequences): textbook three-state dynamic programming over the FULL matrix, best cell of the last row or column."""
    NEG = -10 ** 9
    n, m = len(ref), len(qry)
    H = [[0] * (m + 1) for _ in range(n + 1)]
    E = [[NEG] * (m + 1) for _ in range(n + 1)]          # gap in the reference (query letters inserted)
    F = [[NEG] * (m + 1) for _ in range(n + 1)]          # gap in the query (reference letters deleted)
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            a, b = ref[i - 1], qry[j - 1]
            s = match - 1 if "N" in (a, b) else (match if a == b else -mismatch)
            E[i][j] = max(E[i][j - 1], H[i][j - 1] - gap_open)
            F[i][j] = max(F[i - 1][j], H[i - 1][j] - gap_open)
            H[i][j] = max(H[i - 1][j - 1] + s, E[i][j], F[i][j])
    return max(max(H[n]), max(row[m] for row in H))


def test_full_band_score_is_the_optimum_of_an_independent_dp(oracle_lib):
    # with a band that covers the whole matrix the corner score must be the optimum, whatever the tie rules did to the path
    rng = np.random.default_rng(31)
    for it in range(150):
        ref = mb.random_seq(rng, int(rng.integers(1, 45)))
        qry = (mb.mutate(rng, ref, snp=0.1, indel=0.08, max_indel=5, n_frac=0.03) if it % 3 else mb.random_seq(rng, int(rng.integers(1, 45)))) or "A"
        r = mb.oracle_map_variations(oracle_lib.dll, ref, qry, 0, 200, mb.params(gap_align_left=it % 2))
        assert r["status"] == 0 and r["hit_boundary"] == 0 and r["attempts"] == 1
        assert r["score"] == _overlap_score(ref, qry), (ref, qry)
        assert mb.apply_edit(ref, r) == qry
