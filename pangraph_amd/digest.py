"""Record lists and their digests: the form in which whole builds are compared with the reference.

A `find_matches` call (packages/pangraph/src/pangraph/graph_merging.rs:176-185) returns a list of alignment records; the bit-exact
contract is the 17 observable fields of every record, in the aligner's order.  `tests/golden/builds_expected.json.gz`
holds, for every call of a simulated build, (number of records, sha256 of the JSON of that list) as the compiled reference produced them
(tests/golden/make_golden_builds.py).  Here: packed `pga_match_t` records -> those lists -> digests, and the expected digests per
(guide-tree node, self-merge round).  Used by the `-m gpu` tests and by bench.py's check of the step it timed.
"""
from __future__ import annotations

import gzip
import hashlib
import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILDS_EXPECTED = os.path.join(_ROOT, "tests", "golden", "builds_expected.json.gz")
_OPS = "MIDNSHP=XB"


def digest(rows) -> str:
    return hashlib.sha256(json.dumps(rows, separators=(",", ":")).encode()).hexdigest()


def records_to_lists(rec, pool, names: Sequence[Sequence[str]]):
    """packed pga_match_t records (pangraph_amd.dist.MATCH_DTYPE) + CIGAR pool -> per group, plain lists of the 17 observable fields in the
    reference's order (the same lists tests/util.py:rows_to_lists makes of PafRows).  names[g] = the names of group g's sequences."""
    out = [[] for _ in names]
    if len(rec) == 0:
        return out
    cols = {k: rec[k].tolist() for k in ("group", "qry", "ref", "qry_len", "qry_start", "qry_end", "ref_len", "ref_start", "ref_end", "matches",
                                           "length", "quality", "reverse", "align", "divergence", "cigar_off", "n_cigar", "n_ambi", "inv")}
    pool = np.asarray(pool)
    for i in range(len(rec)):
        g = cols["group"][i]
        o, n = cols["cigar_off"][i], cols["n_cigar"][i]
        c = pool[o:o + n].tolist()
        cg = "".join([f"{x >> 4}{_OPS[x & 0xf]}" for x in c])
        nm = names[g]
        out[g].append([nm[cols["qry"][i]], cols["qry_len"][i], cols["qry_start"][i], cols["qry_end"][i], "-" if cols["reverse"][i] else "+",
                       nm[cols["ref"][i]], cols["ref_len"][i], cols["ref_start"][i], cols["ref_end"][i], cols["matches"][i], cols["length"][i],
                       cols["quality"][i], cols["align"][i], repr(float(cols["divergence"][i])), cg, cols["n_ambi"][i], cols["inv"][i]])
    return out


def expected_build(seed: int, n: int, length: int) -> Optional[dict]:
    """the golden build with these parameters (c5 / c4), or None"""
    if not os.path.exists(BUILDS_EXPECTED):
        return None
    with gzip.open(BUILDS_EXPECTED, "rt") as f:
        e = json.load(f)
    for v in e.values():
        p = v["params"]
        if (p["seed"], p["n"], p["length"]) == (seed, n, length):
            return v
    return None


def expected_by_call(pop, build: dict) -> Dict[Tuple[int, int], Tuple[int, str]]:
    """(guide-tree node, self-merge round) -> (records, sha256) of that find_matches call.  The golden file is laid out by WAVES
    (pangraph_amd/levels.py:build_waves: wave 2(h-1)+r = the merges of tree height h in node order, round r)."""
    out = {}
    hmax = pop.nodes[0].height
    for h in range(1, hmax + 1):
        merges = [nd for nd in pop.nodes if nd.children and nd.height == h]
        for r in (0, 1):
            w = build["waves"][2 * (h - 1) + r]
            if len(w["groups"]) != len(merges):
                raise ValueError(f"golden build does not match the population at height {h}")
            for nd, x in zip(merges, w["groups"]):
                out[(nd.id, r)] = (int(x["n"]), x["sha256"])
    return out


def check_calls(results, want: Dict[Tuple[int, int], Tuple[int, str]]):
    """results: per batch (tasks, records with group = index into tasks, CIGAR pool, covered) -- covered = the indices into `tasks` this
    result speaks for (None: all of them).  Returns (calls checked, list of (node, round) that differ).  A call may appear only once."""
    seen, bad = set(), []
    for ts, rec, pool, covered in results:
        lists = records_to_lists(rec, pool, [t.names for t in ts])
        for i in (range(len(ts)) if covered is None else covered):
            t, rows = ts[i], lists[i]
            key = (t.node, t.round)
            if key in seen:
                raise ValueError(f"call {key} aligned twice")
            seen.add(key)
            if (len(rows), digest(rows)) != want[key]:
                bad.append(key)
    return len(seen), bad
