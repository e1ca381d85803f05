"""Known-answer vectors of the reference's reconsensus unit tests, as data (edit = dict inss / dels / subs, members in NodeId order).
Sources: packages/pangraph/src/reconsensus/reconsensus.rs:141-330 (blocks 0-3 and their expected results, the edge-case block of :470-500)
and packages/pangraph/src/pangraph/pangraph_block.rs:376-860 (find_majority_*, change_consensus_nucleotide_at_pos, edit_consensus_and_realign)."""


def E(inss=(), dels=(), subs=()):
    return {"inss": [tuple(x) for x in inss], "dels": [tuple(x) for x in dels], "subs": [tuple(x) for x in subs]}


BLOCK_0 = ("ATGCGATCGATCGA", [E(subs=[(1, "C")]), E(subs=[(1, "C")]), E(subs=[(1, "C")]), E(subs=[(10, "G")]), E(subs=[(10, "G")])])
BLOCK_0_RECONSENSUS = ("ACGCGATCGATCGA", [E(), E(), E(), E(subs=[(1, "T"), (10, "G")]), E(subs=[(1, "T"), (10, "G")])])
BLOCK_1 = ("AGGACTTCGATCTATTCGGAGAA", [
    E([(17, "TTTT")], [(5, 2)], [(1, "T"), (17, "A")]),
    E([], [(5, 2)], [(1, "T"), (10, "C")]),
    E([], [(5, 2), (16, 2)], [(1, "T"), (10, "C")]),
    E([], [(9, 3)], [(1, "C"), (17, "A")]),
    E([(5, "AA")], [(5, 2)], [(17, "A")])])
BLOCK_1_MUT_RECONSENSUS = ("ATGACTTCGATCTATTCAGAGAA", [
    E([(17, "TTTT")], [(5, 2)], []),
    E([], [(5, 2)], [(10, "C"), (17, "G")]),
    E([], [(5, 2), (16, 2)], [(10, "C")]),
    E([], [(9, 3)], [(1, "C")]),
    E([(5, "AA")], [(5, 2)], [(1, "G")])])
BLOCK_1_RECONSENSUS = ("ATGACCGATCTATTCAGAGAA", [
    E([(15, "TTTT")], [], []),
    E([], [], [(8, "C"), (15, "G")]),
    E([], [(14, 2)], [(8, "C")]),
    E([(5, "TT")], [(7, 3)], [(1, "C")]),
    E([(5, "AA")], [], [(1, "G")])])
BLOCK_2 = ("AGGACTTCGATCTATTCGGAGAA", [
    E([(0, "G"), (3, "AA"), (13, "AA")], [(5, 2), (20, 1)], [(1, "T"), (17, "A")]),
    E([(0, "G"), (13, "AA"), (23, "TT")], [(5, 2), (20, 2)], [(1, "T"), (10, "C")]),
    E([(23, "TT")], [(4, 4)], [(1, "T"), (10, "C")]),
    E([(3, "C"), (23, "TT")], [(9, 3)], [(1, "C"), (17, "A")]),
    E([(0, "G"), (3, "C"), (13, "AA")], [(19, 2)], [(17, "A")])])
BLOCK_3 = ("GCCTCTTCCCGACCACGCGTTACAACATGGGACAGGCCTGCGCTTGAGGC", [
    E([], [(19, 4)], [(5, "A")]),
    E([(35, "AA"), (50, "TT")], [(20, 3)], [(5, "A")]),
    E([], [], [(14, "G"), (27, "G")]),
    E([(50, "TT")], [(20, 3)], [(5, "A")]),
    E([(50, "TT")], [], [])])
BLOCK_3_RECONSENSUS = ("GCCTCATCCCGACCACGCGTAACATGGGACAGGCCTGCGCTTGAGGCTT", [
    E([], [(19, 1), (47, 2)], []),
    E([(32, "AA")], [], []),
    E([(20, "TAC")], [(47, 2)], [(5, "T"), (14, "G"), (24, "G")]),
    E([], [], []),
    E([(20, "TAC")], [], [(5, "T")])])
MAJORITY = {                                                                # reconsensus.rs:276-307
    0: E([], [], [(1, "C")]),
    1: E([], [(5, 2)], [(1, "T"), (17, "A")]),
    2: E([(0, "G"), (13, "AA"), (23, "TT")], [(5, 2), (20, 1)], [(1, "T"), (17, "A")]),
    3: E([(50, "TT")], [(20, 3)], [(5, "A")]),
}
KINDS = {0: 1, 1: 2, 2: 2, 3: 2}                                            # reconsensus.rs:255-274: mutations_only [0], need_realignment [1, 2, 3]
APPLY_KAT = ("AGGACTTCGATCTATTCGGAGAA", E([(0, "G"), (13, "AA"), (23, "TT")], [(5, 2), (20, 1)]), "GAGGACCGATCTAAATTCGGAAATT")   # reconsensus.rs:309-317
# the edge-case block (reconsensus.rs:470-500) before detach_unaligned_nodes drops the member that no longer aligns: the block part of its expectation
EDGE_BLOCK = ("GCCTCTTCCCGACCACGCGTTACAACATGGGACAGGCCTGCGCTTGAGGC", [E([], [(0, 40)]), E([], [(35, 15)]), E([], [(35, 15)]), E([], [(35, 15)]), E()])
EDGE_EXPECTED_CONS = "GCCTCTTCCCGACCACGCGTTACAACATGGGACAG"
EDGE_EXPECTED_MEMBERS = {1: E(), 2: E(), 3: E(), 4: E([(35, "GCCTGCGCTTGAGGC")])}   # members 2..5 of the reference (member 1 is detached by the host afterwards)

# pangraph_block.rs: (members, expected) per function
MAJ_SUBS = [
    ([E(subs=[(0, "G"), (2, "A")])], [(0, "G"), (2, "A")]),                                                         # :376-388
    ([E(subs=[(0, "G")]), E(subs=[(0, "C")]), E(subs=[(0, "T")])], []),                                            # :390-404
    ([E(subs=[(0, "G"), (2, "A")]), E(subs=[(0, "G"), (3, "A")]), E(subs=[(0, "C"), (2, "A")])], [(0, "G"), (2, "A")]),   # :406-419
    ([E(), E(), E(subs=[(0, "C")]), E(subs=[(0, "C")])], []),                                                       # :421-434
]
MAJ_DELS = [
    ([E(dels=[(1, 2), (4, 1)]), E(dels=[(1, 2), (5, 1)]), E(dels=[(0, 1), (4, 1)])], [(1, 2), (4, 1)]),             # :466-479
    ([E(dels=[(1, 3)]), E(dels=[(2, 3)]), E(dels=[(3, 2)]), E(dels=[(6, 1)]), E(dels=[(6, 2)])], [(3, 1)]),         # :481-496
    ([E(dels=[(1, 1), (2, 1), (3, 1)]), E(dels=[(1, 3)]), E(dels=[(1, 1), (2, 2)]), E(dels=[(5, 1)]), E(dels=[(5, 1), (6, 1)])], [(1, 3)]),   # :498-513
]
MAJ_INSS = [
    ([E([(1, "GGG"), (3, "A")]), E([(1, "GGG"), (2, "TT")]), E([(1, "CC"), (3, "A")])], [(1, "GGG"), (3, "A")]),     # :552-565
    ([E([(1, "ATG")]), E([(1, "ATG")]), E([(1, "ATG")]), E([(1, "GTA")]), E([(1, "GTA")])], [(1, "ATG")]),          # :567-582
    ([E([(0, "G"), (2, "T"), (4, "C")]), E([(0, "G"), (3, "A"), (5, "T")]), E([(1, "A"), (2, "T"), (4, "C")]), E([(0, "C"), (2, "T"), (6, "G")]),
      E([(0, "G"), (3, "A"), (4, "C")])], [(0, "G"), (2, "T"), (4, "C")]),                                          # :584-599
    ([E(), E(), E([(1, "AA")]), E([(1, "AA")])], []),                                                               # :601-614
]
MAJ_ALL = ([E([(1, "GG"), (4, "C")], [(2, 1), (6, 1)], [(0, "G"), (5, "C")]), E([(1, "GG"), (3, "A")], [(2, 1), (7, 1)], [(0, "G"), (5, "T")]),
            E([(1, "AA"), (4, "C")], [(2, 1), (6, 1)], [(0, "C"), (5, "C")]), E([(1, "GG"), (4, "C")], [(1, 1), (6, 1)], [(0, "G"), (4, "A")]),
            E([(1, "GG"), (4, "C")], [(2, 1), (5, 1)], [(0, "G"), (5, "C")])],
           E([(1, "GG"), (4, "C")], [(2, 1), (6, 1)], [(0, "G"), (5, "C")]))                                         # :632-666
CHANGE_KATS = [                                                                                                     # :669-725
    ("ATCG", [E(), E(subs=[(1, "G"), (2, "C")]), E(subs=[(1, "A")])], (1, "G"), "AGCG", [E(subs=[(1, "T")]), E(subs=[(2, "C")]), E(subs=[(1, "A")])]),
    ("ATCG", [E(dels=[(1, 2)]), E(), E(subs=[(1, "A")]), E(subs=[(1, "G")])], (1, "G"), "AGCG", [E(dels=[(1, 2)]), E(subs=[(1, "T")]), E(subs=[(1, "A")]), E()]),
]
REALIGN_KAT = ("ATCGGCGATG", [E(), E([], [(6, 2)], [(2, "G")])], E([(10, "AAA")], [(6, 2)], [(0, "G")]), "GTCGGCTGAAA",
               [E([(6, "GA")], [(8, 3)], [(0, "A")]), E([], [(8, 3)], [(0, "A"), (2, "G")])])                       # :786-830 (the map holds two members: NodeId(2) is given twice)
