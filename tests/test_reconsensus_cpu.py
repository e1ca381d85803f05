"""SURVEY 8(f)-4, reconsensus: the CPU restatement (oracle/pgo_reconsensus.py over oracle/pgo_mapvar.c) against every known-answer vector the
reference's own unit tests hold for it (tests/reconsensus_vectors.py).  CPU only."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pgo_reconsensus as rc  # noqa: E402
import mapvarbind as mb  # noqa: E402
import reconsensus_vectors as V  # noqa: E402
from reconsensus_vectors import E  # noqa: E402


def _mapvar(oracle_lib):
    def f(ref, qry, ms, bw):
        g = mb.oracle_map_variations(oracle_lib.dll, ref, qry, ms, bw)
        assert g["status"] == 0
        return E(g["inss"], g["dels"], g["subs"])
    return f


def test_find_majority_known_answers():
    for members, exp in V.MAJ_SUBS:
        assert rc.find_majority_substitutions(members) == exp
    for members, exp in V.MAJ_DELS:
        assert rc.find_majority_deletions(members) == exp
    for members, exp in V.MAJ_INSS:
        assert rc.find_majority_insertions(members) == exp
    assert rc.find_majority_edits([E(), E(), E()]) == E()
    assert rc.find_majority_edits(V.MAJ_ALL[0]) == V.MAJ_ALL[1]
    for b, blk in ((0, V.BLOCK_0), (1, V.BLOCK_1), (2, V.BLOCK_2), (3, V.BLOCK_3)):
        assert rc.find_majority_edits(blk[1]) == V.MAJORITY[b]


def test_apply_and_change_consensus_known_answers():
    cons, e, exp = V.APPLY_KAT
    assert rc.apply_edit(cons, e) == exp
    for cons, members, sub, new_cons, new_members in V.CHANGE_KATS:
        assert rc.apply_substitutions_to_block(cons, members, [sub]) == (new_cons, new_members)
    with pytest.raises(ValueError):
        rc.apply_substitutions_to_block("ATCG", [E()], [(4, "A")])            # pangraph_block.rs:727-741
    with pytest.raises(ValueError, match="already"):
        rc.apply_substitutions_to_block("ATCG", [E()], [(1, "T")])            # :743-757


def test_reconsensus_blocks_known_answers(oracle_lib):
    mv = _mapvar(oracle_lib)
    # mutations only (reconsensus.rs:319-345)
    assert rc.apply_substitutions_to_block(*V.BLOCK_0, V.MAJORITY[0]["subs"]) == V.BLOCK_0_RECONSENSUS
    assert rc.apply_substitutions_to_block(*V.BLOCK_1, V.MAJORITY[1]["subs"]) == V.BLOCK_1_MUT_RECONSENSUS
    # the whole per-block step (reconsensus.rs:255-274, 347-428)
    for b, blk, exp in ((0, V.BLOCK_0, V.BLOCK_0_RECONSENSUS), (1, V.BLOCK_1, V.BLOCK_1_RECONSENSUS), (3, V.BLOCK_3, V.BLOCK_3_RECONSENSUS)):
        kind, cons, members, maj = rc.reconsensus_block(blk[0], blk[1], mv)
        assert kind == V.KINDS[b] and maj == V.MAJORITY[b]
        assert (cons, members) == exp, b
    kind, cons, members, _ = rc.reconsensus_block(*V.BLOCK_2, mv)
    assert kind == 2 and cons == rc.apply_edit(V.BLOCK_2[0], V.MAJORITY[2])
    for e, old in zip(members, V.BLOCK_2[1]):                                   # (no expected block in the reference: the members must round-trip)
        assert rc.apply_edit(cons, e) == rc.apply_edit(V.BLOCK_2[0], old)
    # pangraph_block.rs:786-830
    cons, members, edits, new_cons, exp = V.REALIGN_KAT
    got_cons, jobs = rc.realign_jobs(cons, members, edits)
    assert got_cons == new_cons and [mv(*j) for j in jobs] == exp
    # the edge case of reconsensus.rs:470-500, block part
    kind, cons, members, _ = rc.reconsensus_block(*V.EDGE_BLOCK, mv)
    assert kind == 2 and cons == V.EDGE_EXPECTED_CONS
    for i, exp in V.EDGE_EXPECTED_MEMBERS.items():
        assert members[i] == exp, i
