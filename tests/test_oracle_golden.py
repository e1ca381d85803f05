"""Pins the oracle (oracle/libpgoracle.so, the plain-C restatement) against the committed golden vectors, which were
produced by the reference itself (tests/golden/make_golden.py).  CPU-only."""
import hashlib
import json
import os

import pytest

import stagebind as sb
from util import load_golden, read_fasta, rows_to_lists, GOLDEN


def test_reference_known_answer(oracle_lib):
    """packages/pangraph/src/align/minimap2_lib/align_with_minimap2_lib.rs:135-204"""
    kat = load_golden("kat_unit_pair.json")
    rows = oracle_lib.align_all(kat["seqs"], kat["names"], **kat["params"])
    assert rows_to_lists(rows) == kat["expected"]
    assert len(rows) == 1
    r, e = rows[0], kat["rust_expected"]
    assert [r.qname, r.qlen, r.qs, r.qe] == e["qry"] and [r.tname, r.tlen, r.rs, r.re] == e["reff"]
    assert (r.mlen, r.blen, r.mapq, r.strand, r.cg, float(r.AS)) == (e["matches"], e["length"], e["quality"], e["orientation"], e["cigar"], e["align"])
    assert r.de == e["divergence"]


@pytest.mark.parametrize("idx", range(14))
def test_e2e_cases(oracle_lib, idx):
    cases = load_golden("e2e_cases.json.gz")
    c = cases[idx]
    rows = oracle_lib.align_all(c["seqs"], c["names"], **c["params"])
    assert rows_to_lists(rows) == c["expected"], c["name"]


def test_e2e_case_count():
    assert len(load_golden("e2e_cases.json.gz")) == 14


def test_plasmids_four(oracle_lib):
    g = load_golden("plasmids_expected.json.gz")
    _, seqs = read_fasta(os.path.join(GOLDEN, "plasmids.fa.gz"))
    rows = rows_to_lists(oracle_lib.align_all(seqs[:4], g["names"][:4], sensitivity=10))
    assert rows == g["four_asm10"]
    assert any(r[4] == "-" for r in rows) and len(rows) > 50


def test_stage_sketch(oracle_lib):
    st = load_golden("stage_vectors.json.gz")
    assert len(st["sketch"]) >= 50
    for v in st["sketch"]:
        got = sb.oracle_sketch(oracle_lib.dll, v["seq"], v["w"], v["k"], v["rid"])
        assert [[str(x), str(y)] for x, y in got] == v["mz"], (v["w"], v["k"], v["rid"])


def test_stage_ksw(oracle_lib):
    st = load_golden("stage_vectors.json.gz")
    scorings = {"asm5": (1, 19, 39, 3, 81, 1), "asm10": (1, 9, 16, 2, 41, 1), "asm20": (1, 4, 6, 2, 26, 1)}
    assert len(st["ksw"]) >= 200
    for v in st["ksw"]:
        ma, mb, q1, e1, q2, e2 = scorings[v["preset"]]
        got = sb.oracle_extd2(oracle_lib.dll, sb.nt4(v["q"]), sb.nt4(v["t"]), sb.simple_mat(ma, mb, 1), q1, e1, q2, e2, v["w"], v["zdrop"], v["end_bonus"], v["flag"])
        exp = v["ez"]
        keys = ["zdropped", "reach_end", "cigar", "score"]
        if not (v["flag"] & 0x08):
            keys += ["max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q"]
        for k in keys:
            assert got[k] == exp[k], (k, v["preset"], v["w"], v["flag"], len(v["q"]), len(v["t"]))


# ---- block sets of upper tree levels (tests/golden/make_golden_levels.py) ----
def test_real_block_sets(oracle_lib):
    """the reference's own finished builds: 137 plasmid blocks with their real decimal BlockIds, the russian-doll plasmids"""
    exp = load_golden("levels_expected.json.gz")
    names, seqs = read_fasta(os.path.join(GOLDEN, "plasmids_blocks.fa.gz"))
    assert len(names) == 137 and all(n.isdigit() for n in names)
    assert rows_to_lists(oracle_lib.align_all(seqs, names, sensitivity=10)) == exp["plasmids_blocks"]["asm10"]
    names, seqs = read_fasta(os.path.join(GOLDEN, "russian_doll_plasmids.fa.gz"))
    for sens in (10, 20):
        assert rows_to_lists(oracle_lib.align_all(seqs, names, sensitivity=sens)) == exp["russian_doll"][f"asm{sens}"]


def test_c3_small_upper_waves(oracle_lib):
    """every wave above the leaf merges of the simulated 10-genome build (the leaf wave is left to the GPU suite: 15 s here)"""
    from levels_util import digest
    from pangraph_amd.levels import Population, Rates
    exp = load_golden("levels_expected.json.gz")["c3_small"]
    p = exp["params"]
    waves = Population(p["seed"], p["n"], p["length"], Rates(**p["rates"])).build_waves()
    assert len(waves) == len(exp["waves"])
    for (label, groups, names), e in list(zip(waves, exp["waves"]))[1:]:
        assert [len(g) for g in groups] == e["n_blocks"], label
        got = [rows_to_lists(oracle_lib.align_all([a.tobytes().decode() for a in g], n, sensitivity=10)) for g, n in zip(groups, names)]
        assert [dict(n=len(r), sha256=digest(r)) for r in got] == e["groups"], label
