"""SURVEY 8(f)-4, reconsensus on the device (pga_reconsensus): the reference's own unit-test blocks (tests/reconsensus_vectors.py) through the
product, and random blocks (majority substitutions / deletions / insertions, ties, overlapping and adjacent deletions, insertions at the
ends, members that disagree) against the CPU restatement oracle/pgo_reconsensus.py over oracle/pgo_mapvar.c."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pgo_reconsensus as rc  # noqa: E402
import mapvarbind as mb  # noqa: E402
import reconsensus_vectors as V  # noqa: E402
from reconsensus_vectors import E  # noqa: E402

pytestmark = pytest.mark.gpu


def _norm(e):
    return E(e["inss"], e["dels"], e["subs"])


def _oracle_mv(oracle_lib):
    def f(ref, qry, ms, bw):
        g = mb.oracle_map_variations(oracle_lib.dll, ref, qry, ms, bw)
        assert g["status"] == 0
        return E(g["inss"], g["dels"], g["subs"])
    return f


def test_reference_unit_test_blocks(gpu_lib):
    from pangraph_amd.reconsensus import reconsensus
    blocks = [V.BLOCK_0, V.BLOCK_1, V.BLOCK_2, V.BLOCK_3, V.EDGE_BLOCK, (V.REALIGN_KAT[0], V.REALIGN_KAT[1]), ("ATCG", [E(), E(), E()])]
    got = reconsensus(blocks, dll=gpu_lib.dll)
    # reconsensus.rs:255-307: the analysis
    for b in range(4):
        assert got[b][0] == V.KINDS[b] and _norm(got[b][3]) == V.MAJORITY[b]
    # :319-428: the blocks afterwards
    assert (got[0][1], [_norm(e) for e in got[0][2]]) == V.BLOCK_0_RECONSENSUS
    assert (got[1][1], [_norm(e) for e in got[1][2]]) == V.BLOCK_1_RECONSENSUS
    assert (got[3][1], [_norm(e) for e in got[3][2]]) == V.BLOCK_3_RECONSENSUS
    assert got[2][0] == 2 and got[2][1] == rc.apply_edit(V.BLOCK_2[0], V.MAJORITY[2])
    for e, old in zip(got[2][2], V.BLOCK_2[1]):
        assert rc.apply_edit(got[2][1], _norm(e)) == rc.apply_edit(V.BLOCK_2[0], old)
    # the edge case of :470-500 (block part) and pangraph_block.rs:617-630 (no variation: untouched)
    assert got[4][0] == 2 and got[4][1] == V.EDGE_EXPECTED_CONS
    for i, exp in V.EDGE_EXPECTED_MEMBERS.items():
        assert _norm(got[4][2][i]) == exp
    assert got[6][0] == 0 and got[6][1] == "ATCG" and all(_norm(e) == E() for e in got[6][2])
    assert all(s == 0 for g in got for s in g[4])
    # pangraph_block.rs: find_majority_* one by one (kind and majority edit of single-purpose blocks)
    singles = [(m, E(subs=x)) for m, x in V.MAJ_SUBS] + [(m, E(dels=x)) for m, x in V.MAJ_DELS] + [(m, E(inss=x)) for m, x in V.MAJ_INSS] + [V.MAJ_ALL]
    got = reconsensus([("ATCGAATTCC", m) for m, _ in singles], dll=gpu_lib.dll)
    for (m, exp), g in zip(singles, got):
        assert _norm(g[3]) == exp
    # change_consensus_nucleotide_at_pos through majority substitutions (pangraph_block.rs:669-725 have no majority: use block 1's substitution part)
    sub_only = (V.BLOCK_1[0], [E(subs=e["subs"]) for e in V.BLOCK_1[1]])
    g = reconsensus([sub_only], dll=gpu_lib.dll)[0]
    exp_cons, exp_mem = rc.apply_substitutions_to_block(sub_only[0], sub_only[1], rc.find_majority_substitutions(sub_only[1]))
    assert g[0] == 1 and g[1] == exp_cons and [_norm(e) for e in g[2]] == exp_mem


def _random_block(rng, L, depth, p_shared, subs_only=False):
    cons = mb.random_seq(rng, L)
    # a few shared (majority or near-majority) events plus private noise
    shared = []
    for _ in range(int(rng.integers(0, 5))):
        kind = 0 if subs_only else int(rng.integers(0, 3))
        pos = int(rng.integers(0, L))
        if kind == 0:
            shared.append(("s", pos, "ACGT"[int(rng.integers(0, 4))]))
        elif kind == 1:
            shared.append(("d", pos, int(rng.integers(1, min(12, L - pos) + 1))))
        else:
            shared.append(("i", int(rng.integers(0, L + 1)), mb.random_seq(rng, int(rng.integers(1, 8)))))
    members = []
    for _ in range(depth):
        subs, dels, inss = {}, [], {}
        for ev in shared:
            if rng.random() < p_shared:
                if ev[0] == "s" and cons[ev[1]] != ev[2]:
                    subs[ev[1]] = ev[2]
                elif ev[0] == "d":
                    dels.append((ev[1], ev[2]))
                elif ev[0] == "i":
                    inss[ev[1]] = ev[2]
        for _ in range(int(rng.integers(0, 4))):
            pos = int(rng.integers(0, L))
            a = "ACGT"[int(rng.integers(0, 4))]
            if a != cons[pos] and pos not in subs:
                subs[pos] = a
        if rng.random() < (0.1 if subs_only else 0.4):
            pos = int(rng.integers(0, L))
            dels.append((pos, int(rng.integers(1, min(6, L - pos) + 1))))
        if rng.random() < 0.3:
            pos = int(rng.integers(0, L + 1))
            if pos not in inss:
                inss[pos] = mb.random_seq(rng, int(rng.integers(1, 5)))
        # a substitution inside a deletion is an inconsistent edit (the reference errors on it when it reconciles): keep them apart
        subs = {p: a for p, a in subs.items() if not any(d[0] <= p < d[0] + d[1] for d in dels)}
        e = E(sorted(inss.items()), sorted(dels), sorted(subs.items()))
        if rc.aligned_count_after(e, 0, L) == 0:
            e = E()
        members.append(e)
    return cons, members


def test_random_blocks_vs_restatement(gpu_lib, oracle_lib):
    from pangraph_amd.reconsensus import reconsensus
    rng = np.random.default_rng(2024)
    mv = _oracle_mv(oracle_lib)
    blocks = []
    for _ in range(120):
        blocks.append(_random_block(rng, int(rng.integers(30, 400)), int(rng.integers(1, 12)), float(rng.choice([0.3, 0.55, 0.8, 1.0]))))
    for _ in range(40):                                                                 # majority substitutions only: consensus letters change, members are reconciled
        blocks.append(_random_block(rng, int(rng.integers(30, 300)), int(rng.integers(3, 12)), 0.8, subs_only=True))
    blocks.append(_random_block(rng, 5000, 40, 0.7))
    got = reconsensus(blocks, dll=gpu_lib.dll)
    kinds = [0, 0, 0]
    for (cons, members), g in zip(blocks, got):
        maj = rc.find_majority_edits(members)
        assert _norm(g[3]) == maj
        if (maj["inss"] or maj["dels"]) and (rc.apply_edit(cons, maj) == "" or rc.band_from_edits(maj, len(cons)) is None):
            assert g[0] < 0
            continue
        kind, new_cons, new_members, _ = rc.reconsensus_block(cons, members, mv)
        assert g[0] == kind and g[1] == new_cons
        assert [_norm(e) for e in g[2]] == new_members
        assert all(s == 0 for s in g[4])
        kinds[kind] += 1
        for e, old in zip(g[2], members):                                          # the members are still the same sequences
            assert rc.apply_edit(new_cons, _norm(e)) == rc.apply_edit(cons, old)
    assert min(kinds) > 5, kinds


def test_a_block_whose_reconciliation_fails_comes_back_untouched(gpu_lib):
    """apply_substitutions_to_block (edits.rs:196-238) fails for a member that holds two substitutions at one position: the reference returns Err
    for the call; here the block reports kind -5 with its ORIGINAL consensus and every member's original edits (decided before anything of the
    block is packed), the member carries its own status, and the other blocks of the call are unaffected."""
    from pangraph_amd.reconsensus import reconsensus
    bad = ("ATCGAATTCC", [E(subs=[(2, "G")]), E(subs=[(2, "G")]), E(subs=[(2, "T"), (2, "A")]), E(subs=[(2, "G")])])
    ok = ("ATCGAATTCC", [E(subs=[(2, "G")]), E(subs=[(2, "G")]), E(subs=[(5, "C")])])
    got = reconsensus([ok, bad, ok], dll=gpu_lib.dll)
    assert got[1][0] == -5 and got[1][1] == bad[0]
    assert [_norm(e) for e in got[1][2]] == bad[1]
    assert got[1][4][2] != 0 and got[1][4][0] == 0
    for g in (got[0], got[2]):
        assert g[0] == 1 and g[1] == "ATGGAATTCC" and all(s == 0 for s in g[4])
        assert [_norm(e) for e in g[2]] == [E(), E(), E(subs=[(2, "C"), (5, "C")])]
