"""SURVEY 8(f)-1 on the GPU: pga_map_variations (pangraph_amd/csrc/pga_mapvar.hip) against the CPU restatement oracle/pgo_mapvar.c
(itself pinned by the reference's unit-test vectors in tests/test_mapvar_cpu.py)."""
import numpy as np
import pytest

import mapvarbind as mb
from test_mapvar_cpu import MAPVAR_KATS, RECONSENSUS_KATS, E

pytestmark = pytest.mark.gpu

KEYS = ("status", "score", "attempts", "hit_boundary", "subs", "dels", "inss")


def _same(got, exp):
    for g, e in zip(got, exp):
        if e["status"] != 0:
            assert g["status"] == e["status"]
        else:
            assert {k: g[k] for k in KEYS} == {k: e[k] for k in KEYS}


def _oracle(d, jobs, p):
    return [mb.oracle_map_variations(d, r, q, ms, bw, p) for r, q, ms, bw in jobs]


def test_reference_known_answers_through_the_product(gpu_lib):
    d = gpu_lib.dll
    jobs = [(r, q, ms, bw) for r, q, ms, bw, *_ in MAPVAR_KATS]
    got = mb.product_map_variations(d, jobs)
    for g, (r, q, ms, bw, subs, dels, inss) in zip(got, MAPVAR_KATS):
        assert g["status"] == 0 and (g["subs"], g["dels"], g["inss"]) == (subs, dels, inss)
        assert mb.apply_edit(r, g) == q
    # align_with_nextclade.rs:92-152, :274-311; align.rs:191-250
    p = mb.params(min_length=3, max_alignment_attempts=3)
    ref = "CTTGGAGGTTCCGTGGCTAGATAACAGAACATTCTTGGAATGCTGATCTTTATAAGCTCATGCGACACTTCGCATGGTGAGCCTTTGT"
    qry = "CTTGGAGGTTCCGTGGCTATAAAGATAACAGAACATTCTTGGAATGCTGATCAAGCTCATGGGACANNTCGCATGGTGGACAGCCTTTGT"
    g = mb.product_map_variations(d, [(ref, qry, 0, 4), ("A" * 37, "G" * 18, 70, 0)], p)
    assert (g[0]["subs"], g[0]["dels"], g[0]["inss"], g[0]["hit_boundary"]) == ([(62, "G"), (67, "N"), (68, "N")], [(48, 5)], [(19, "TAAA"), (79, "GAC")], 0)
    assert (g[1]["subs"], g[1]["dels"], g[1]["inss"], g[1]["score"], g[1]["hit_boundary"]) == ([], [(0, 37)], [(37, "G" * 18)], 0, 0)
    core = "TTGGCCCCGGTGCTGTCCGTCAACACGTCGTCGTCCGGCGACCTACCTGGTCTCAAAGGAGGTTTTGTTAAATGAATTAGATGGGTAAGGTTACCACGTCA"
    ref, qry = core + "A" * 30, "G" * 30 + core
    p = mb.params(min_length=100, max_alignment_attempts=1, extra_band_width=0)
    g = mb.product_map_variations(d, [(ref, qry, -30, 1), (ref, qry, 0, 31), (ref, qry, 0, 30)], p)
    assert [x["hit_boundary"] for x in g] == [0, 0, 1]
    # the reference's errors, per job
    g = mb.product_map_variations(d, [("ACGT", "AC", 0, 0), ("ACGT", "ACgT", 0, 0), ("ACGT", "ACGT", 0, 0)], mb.params(min_length=3))
    assert [x["status"] for x in g] == [1, 2, 0]
    assert mb.product_map_variations(d, []) == []


def test_reconsensus_realign_known_answers_through_the_product(gpu_lib):
    # reconsensus.rs:400-428: edit_consensus_and_realign of the reference's test blocks 1 and 3, all members of both blocks in one call
    jobs, exp = [], []
    for cons, members, majority, new_cons, expected in RECONSENSUS_KATS:
        got_cons, j = mb.realign_jobs(cons, members, majority)
        assert got_cons == new_cons
        jobs += j; exp += expected
    got = mb.product_map_variations(gpu_lib.dll, jobs)
    assert [E(g["inss"], g["dels"], g["subs"]) for g in got] == exp


def test_random_members_vs_oracle(gpu_lib, oracle_lib):
    # members of blocks of 1 .. 3000 bp under every combination of the scoring switches; narrow bands force the retry rounds
    rng = np.random.default_rng(11)
    for variant in range(6):
        p = mb.params(gap_align_left=variant % 2, penalty_gap_extend=(0, 1, 2)[variant % 3], left_terminal_gaps_free=variant != 3, right_terminal_gaps_free=variant != 4,
                      max_alignment_attempts=(4, 1, 3)[variant % 3], extra_band_width=(5, 0, 2)[variant % 3])
        jobs = []
        for it in range(120):
            L = int(rng.integers(1, 3000 if it % 10 == 0 else 400))
            ref = mb.random_seq(rng, L)
            for _ in range(int(rng.integers(1, 4))):                       # several members per consensus
                qry = mb.mutate(rng, ref, snp=0.03, indel=0.008, max_indel=int(rng.integers(1, 60)), n_frac=0.01 if it % 4 == 0 else 0.0) or "C"
                if it % 17 == 0:
                    qry = qry[int(rng.integers(0, 30)):max(1, len(qry) - int(rng.integers(0, 30)))] or "G"   # terminal deletions
                jobs.append((ref, qry, int(rng.integers(-4, 5)), int(rng.integers(0, 15))))
        exp = _oracle(oracle_lib.dll, jobs, p)
        got = mb.product_map_variations(gpu_lib.dll, jobs, p)
        _same(got, exp)
        assert any(e["attempts"] > 1 for e in exp) or p.max_alignment_attempts == 1
        for (ref, qry, _, _), g in zip(jobs, got):
            assert mb.apply_edit(ref, g) == qry


def test_wide_bands_and_ambiguity_letters_vs_oracle(gpu_lib, oracle_lib):
    # bands of 100 .. 3000 columns (LDS rings of 512 and 2048 columns, the device-memory ring), shifted bands, IUPAC letters, unrelated pairs
    rng = np.random.default_rng(13)
    jobs = []
    iupac = np.array(list("TAWCYMHGKRDSBVN"))
    for bw in (70, 200, 260, 900, 1100, 2600):
        L = int(rng.integers(bw, 3 * bw + 300))
        ref = mb.random_seq(rng, L)
        qry = mb.mutate(rng, ref, snp=0.02, indel=0.004, max_indel=bw // 2)
        jobs.append((ref, qry, int(rng.integers(-20, 21)), bw))
        big = mb.random_seq(rng, bw - 10)
        jobs.append((ref, ref[:L // 3] + big + ref[L // 3:], -(bw // 2), bw // 2 + 3))          # one long insertion
        jobs.append((ref, ref[:L // 4] + ref[L // 4 + bw - 20:], bw // 2, bw // 2))              # one long deletion
    for _ in range(40):
        ref = "".join(iupac[rng.integers(0, 15, int(rng.integers(5, 300)))])
        qry = "".join(iupac[rng.integers(0, 15, int(rng.integers(5, 300)))])
        jobs.append((ref, qry, int(rng.integers(-50, 50)), int(rng.integers(0, 40))))           # unrelated: every attempt hits the boundary
    jobs.append(("A", "C", 0, 0)); jobs.append(("ACGT" * 10, "A", -100, 0)); jobs.append(("A", "ACGT" * 10, 100, 0))
    jobs.append(("", "ACGT", 0, 0)); jobs.append(("ACGT", "", 0, 0)); jobs.append(("", "", 0, 0))    # empty sides (min_length 0 below): one insertion / one deletion / nothing
    jobs.append(("ACGT" * 50, "ACGT" * 50, 0, 2 ** 31 - 1)); jobs.append(("ACGT" * 50, "TTTT" + "ACGT" * 50, -(2 ** 31 - 1), 0))   # band and shift at the ends of their types
    p = mb.params(min_length=0)
    exp = _oracle(oracle_lib.dll, jobs, p)
    got = mb.product_map_variations(gpu_lib.dll, jobs, p)
    _same(got, exp)
    for (ref, qry, _, _), g in zip(jobs, got):
        assert mb.apply_edit(ref, g) == qry


def test_output_pools_overflow_round(gpu_lib, oracle_lib, monkeypatch):
    # the edit pools are sized for typical divergence; jobs that do not fit run again in a round with worst-case pools
    monkeypatch.setenv("PGA_MAPVAR_TIGHT_POOLS", "1")
    rng = np.random.default_rng(19)
    jobs = []
    for it in range(150):
        ref = mb.random_seq(rng, int(rng.integers(20, 500)))
        qry = mb.random_seq(rng, int(rng.integers(20, 500))) if it % 3 == 0 else mb.mutate(rng, ref, snp=0.2, indel=0.05, max_indel=6)
        jobs.append((ref, qry or "A", int(rng.integers(-3, 4)), int(rng.integers(0, 30))))
    got = mb.product_map_variations(gpu_lib.dll, jobs)
    _same(got, _oracle(oracle_lib.dll, jobs, mb.params()))
    for (ref, qry, _, _), g in zip(jobs, got):
        assert mb.apply_edit(ref, g) == qry


def test_block_of_many_members_round_trip(gpu_lib, oracle_lib):
    # a merge at bench scale: 600 members of a 12 kb block (sampled against the oracle, all through the round trip)
    rng = np.random.default_rng(17)
    ref = mb.random_seq(rng, 12000)
    jobs = [(ref, mb.mutate(rng, ref, snp=0.01, indel=0.001, max_indel=80), 0, 40) for _ in range(600)]
    got = mb.product_map_variations(gpu_lib.dll, jobs)
    for (r, q, _, _), g in zip(jobs, got):
        assert g["status"] == 0 and mb.apply_edit(r, g) == q
    idx = list(range(0, 600, 60))
    _same([got[i] for i in idx], _oracle(oracle_lib.dll, [jobs[i] for i in idx], mb.params()))


def test_sharded_jobs_give_the_same_edits(gpu_lib):
    # multi-GPU shape of this path: every rank takes its share of the jobs (mapvar.shard_jobs), no collective; two shares on one GPU here
    from pangraph_amd.mapvar import shard_jobs
    rng = np.random.default_rng(23)
    jobs = []
    for _ in range(40):
        ref = mb.random_seq(rng, int(rng.integers(50, 3000)))
        jobs += [(ref, mb.mutate(rng, ref, snp=0.02, indel=0.005, max_indel=25) or "A", int(rng.integers(-2, 3)), int(rng.integers(0, 20))) for _ in range(5)]
    whole = mb.product_map_variations(gpu_lib.dll, jobs)
    merged = [None] * len(jobs)
    for share in shard_jobs(jobs, 2):
        for i, g in zip(share, mb.product_map_variations(gpu_lib.dll, [jobs[i] for i in share])):
            merged[i] = g
    assert merged == whole


def test_several_jobs_per_wave_vs_oracle(gpu_lib, oracle_lib):
    # narrow bands share a wave (k_mapvar_packed: segments of 16 or 32 lanes): ragged groups, references of very different length in one
    # wave, jobs with an error between good ones, last rows at the width limit, and the same jobs one per wave (PGA_MAPVAR_NO_PACK)
    rng = np.random.default_rng(29)
    for extra, bws in ((0, (0, 1, 3, 7)), (5, (0, 2)), (5, (3, 6, 10)), (2, (13,))):
        p = mb.params(extra_band_width=extra, min_length=3)
        jobs = []
        for it in range(61):                                                   # not a multiple of 2 or 4
            ref = mb.random_seq(rng, int(rng.integers(1, 40)) if it % 9 == 0 else int(rng.integers(40, 2500)))
            qry = mb.mutate(rng, ref, snp=0.03, indel=0.004, max_indel=int(rng.integers(1, 6))) or "A"
            if it % 13 == 5:
                qry = qry[:max(1, len(qry) // 2)] + "x" + qry[len(qry) // 2:]  # to_nuc rejects it
            if it % 17 == 3:
                qry = "AC"                                                      # shorter than min_length
            if it % 11 == 7:
                qry = qry + mb.random_seq(rng, int(rng.integers(1, 12)))       # a tail that widens the last row
            jobs.append((ref, qry, int(rng.integers(-2, 3)), int(bws[it % len(bws)])))
        exp = _oracle(oracle_lib.dll, jobs, p)
        got = mb.product_map_variations(gpu_lib.dll, jobs, p)
        _same(got, exp)
