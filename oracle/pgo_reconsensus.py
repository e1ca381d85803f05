"""oracle/pgo_reconsensus.py -- TEST INFRASTRUCTURE ONLY (never imported by the product).

CPU restatement of the reference's reconsensus step, plain Python over small blocks:

    find_majority_substitutions / _deletions / _insertions   packages/pangraph/src/pangraph/pangraph_block.rs:191-256
    positions_to_intervals                                    packages/pangraph/src/utils/interval.rs:60-86
    Edit::apply                                               packages/pangraph/src/pangraph/edits.rs:307-329
    Edit::reconcile_substitution_with_consensus               packages/pangraph/src/pangraph/edits.rs:157-238
    PangraphBlock::change_consensus_nucleotide_at_pos         packages/pangraph/src/pangraph/pangraph_block.rs:258-291
    apply_substitutions_to_block                              packages/pangraph/src/reconsensus/reconsensus.rs:128-137
    PangraphBlock::edit_consensus_and_realign                 packages/pangraph/src/pangraph/pangraph_block.rs:295-332
    BandParameters::from_edits (Edit::aln_mean_shift / aln_bandwidth)   packages/pangraph/src/align/map_variations.rs:29-37, edits.rs:418-531
    analyze_blocks_for_reconsensus / reconsensus_graph (the block part)  packages/pangraph/src/reconsensus/reconsensus.rs:32-126

The re-alignment itself is oracle/pgo_mapvar.c (`pgo_map_variations` of libpgoracle.so).  Pinned on the reference's own unit-test vectors
(tests/test_reconsensus_cpu.py): blocks 0-3 and the edge-case block of reconsensus.rs:141-520, the find_majority_* cases of
pangraph_block.rs:337-560.  The node / path bookkeeping around it (detach_unaligned_nodes, graph maps) is host work and not restated.

An edit is a dict {"subs": [(pos, letter)], "dels": [(pos, len)], "inss": [(pos, seq)]}; a block is (consensus, [edit of member 0, ...])
with the members in the reference's BTreeMap order (ascending NodeId).
"""
from collections import Counter


def is_majority(count, depth):                                   # pangraph_block.rs:200-204
    return count > depth // 2


def find_majority_substitutions(members):                        # pangraph_block.rs:207-223
    depth = len(members)
    by_pos = {}
    for e in members:
        for pos, alt in e["subs"]:
            by_pos.setdefault(pos, []).append(alt)
    out = []
    for pos, alts in by_pos.items():
        # max_by_key over a HashMap of counts: only a count > depth / 2 survives the filter, and at most one letter can have it
        for alt, c in Counter(alts).items():
            if is_majority(c, depth):
                out.append((pos, alt))
    out.sort(key=lambda s: s[0])
    return out


def positions_to_intervals(positions):                           # interval.rs:60-86
    out = []
    for p in sorted(set(positions)):
        if out and out[-1][1] == p:
            out[-1][1] = p + 1
        else:
            out.append([p, p + 1])
    return [(a, b) for a, b in out]


def find_majority_deletions(members):                            # pangraph_block.rs:226-240
    depth = len(members)
    cnt = Counter()
    for e in members:
        for pos, ln in e["dels"]:
            for p in range(pos, pos + ln):
                cnt[p] += 1
    return [(a, b - a) for a, b in positions_to_intervals([p for p, c in cnt.items() if is_majority(c, depth)])]


def find_majority_insertions(members):                           # pangraph_block.rs:243-256
    depth = len(members)
    cnt = Counter()
    for e in members:
        for pos, seq in e["inss"]:
            cnt[(pos, seq)] += 1
    out = [(pos, seq) for (pos, seq), c in cnt.items() if is_majority(c, depth)]
    out.sort(key=lambda i: i[0])
    return out


def find_majority_edits(members):                                # pangraph_block.rs:192-198
    return {"inss": find_majority_insertions(members), "dels": find_majority_deletions(members), "subs": find_majority_substitutions(members)}


def apply_edit(ref, e):                                          # edits.rs:307-329
    q = list(ref)
    for pos, alt in e["subs"]:
        q[pos] = alt
    for pos, ln in e["dels"]:
        for k in range(pos, pos + ln):
            q[k] = "-"
    for pos, seq in sorted(e["inss"], reverse=True):
        q[pos:pos] = list(seq)
    return "".join(c for c in q if c != "-")


def is_position_deleted(e, pos):                                 # edits.rs:157-159
    return any(p <= pos < p + ln for p, ln in e["dels"])


def reconcile_substitution_with_consensus(e, sub, original):     # edits.rs:196-238 (raises where the reference returns an error)
    pos, alt = sub
    at = [s for s in e["subs"] if s[0] == pos]
    if len(at) == 0:
        if not is_position_deleted(e, pos):                      # add_substitution_if_not_deleted, edits.rs:162-167
            e["subs"].append((pos, original))
            e["subs"].sort(key=lambda s: s[0])                   # sort_by_key: stable
    elif len(at) == 1:
        if is_position_deleted(e, pos):
            raise ValueError(f"At position {pos}: sequence has both a substitution and a deletion")
        if at[0][1] == alt:                                      # remove_substitution_if_matching, edits.rs:170-178
            e["subs"] = [s for s in e["subs"] if not (s[0] == pos and s[1] == alt)]
    else:
        raise ValueError(f"At position {pos}: sequence states disagree")


def apply_substitutions_to_block(consensus, members, subs):      # reconsensus.rs:128-137 + pangraph_block.rs:260-291
    cons = list(consensus)
    members = [{"subs": list(e["subs"]), "dels": list(e["dels"]), "inss": list(e["inss"])} for e in members]
    for pos, alt in subs:
        if pos >= len(cons):
            raise ValueError("Position out of bounds")
        original = cons[pos]
        if original == alt:
            raise ValueError("Cannot change consensus character: it is already that letter")
        cons[pos] = alt
        for e in members:
            reconcile_substitution_with_consensus(e, (pos, alt), original)
    return "".join(cons), members


def aligned_count_after(e, p, cons_len):                         # edits.rs:418-440
    total = max(cons_len - p, 0)
    overlap = sum((d[0] + d[1]) - max(p, d[0]) for d in e["dels"] if d[0] + d[1] > p)
    return max(total - overlap, 0)


def _round_half_away(x):
    return int(x + 0.5) if x >= 0 else -int(-x + 0.5)            # f64::round


def band_from_edits(e, cons_len):                                # map_variations.rs:29-37, edits.rs:442-531
    ac = aligned_count_after(e, 0, cons_len)
    if ac == 0:
        return None
    total = 0
    for pos, seq in e["inss"]:
        total -= len(seq) * aligned_count_after(e, pos, cons_len)
    for pos, ln in e["dels"]:
        total += ln * aligned_count_after(e, pos, cons_len)
    ms = _round_half_away(total / ac)
    tuples = sorted([(pos, -len(seq)) for pos, seq in e["inss"]] + [(pos, ln) for pos, ln in e["dels"]], key=lambda t: t[0])
    bw, cur = 0, 0
    for i, (pos, shift) in enumerate(tuples):
        if i == 0 and pos > 0:
            bw = max(bw, abs(cur - ms))
        cur += shift
        if i == len(tuples) - 1 and (pos == cons_len or (shift > 0 and pos + shift == cons_len)):
            continue
        bw = max(bw, abs(cur - ms))
    return ms, bw


def realign_jobs(consensus, members, majority):                  # pangraph_block.rs:295-332: the new consensus and one map_variations job per member
    new_cons = apply_edit(consensus, majority)
    bms, bbw = band_from_edits(majority, len(consensus))
    jobs = []
    for e in members:
        oms, obw = band_from_edits(e, len(consensus))
        jobs.append((new_cons, apply_edit(consensus, e), oms - bms, obw + bbw))
    return new_cons, jobs


def reconsensus_block(consensus, members, map_variations):
    """analyze_blocks_for_reconsensus + the per-block work of reconsensus_graph (reconsensus.rs:32-126).
    map_variations(ref, qry, mean_shift, band_width) -> edit dict (the oracle's pgo_map_variations).
    Returns (kind, new consensus, new member edits, majority edits); kind 0 untouched, 1 substitutions only, 2 realigned."""
    maj = find_majority_edits(members)
    if maj["inss"] or maj["dels"]:                               # has_indels
        new_cons, jobs = realign_jobs(consensus, members, maj)
        return 2, new_cons, [map_variations(*j) for j in jobs], maj
    if maj["subs"]:
        cons, mem = apply_substitutions_to_block(consensus, members, maj["subs"])
        return 1, cons, mem, maj
    return 0, consensus, members, maj
