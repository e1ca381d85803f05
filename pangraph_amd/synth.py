"""Deterministic synthetic genomes for tests and bench.py.

`evolve_population` draws a random ancestor and evolves it along a random bifurcating tree with SNPs, short
indels, inversions, insertions of novel sequence (HGT-like) and deletions, so that pairwise alignments contain
everything the hot path has to handle: both strands, long gaps, z-drop splits, unrelated flanks.
"""
from __future__ import annotations

import numpy as np

_ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b


def random_seq(rng: np.random.Generator, n: int) -> np.ndarray:
    return _ALPHA[rng.integers(0, 4, size=n)]


def revcomp(s: np.ndarray) -> np.ndarray:
    return _COMP[s[::-1]]


def mutate(rng: np.random.Generator, s: np.ndarray, snp: float = 0.01, indel: float = 0.001, n_inv: int = 0, n_ins: int = 0,
           n_del: int = 0, max_event: int = 20000) -> np.ndarray:
    s = s.copy()
    n = len(s)
    # SNPs
    m = rng.random(n) < snp
    s[m] = _ALPHA[(np.searchsorted(_ALPHA, s[m]) + rng.integers(1, 4, size=int(m.sum()))) % 4]
    # short indels (geometric length, mean 3)
    k = rng.binomial(n, indel)
    if k:
        pos = np.sort(rng.choice(n, size=k, replace=False))
        parts, last = [], 0
        for p in pos:
            parts.append(s[last:p])
            ln = int(rng.geometric(1 / 3.0))
            if rng.random() < 0.5:
                parts.append(random_seq(rng, ln))
                last = p
            else:
                last = min(n, p + ln)
        parts.append(s[last:])
        s = np.concatenate(parts)
    for _ in range(n_inv):
        n = len(s); ln = int(rng.integers(max(2, max_event // 20), max_event)); ln = min(ln, n // 2)
        p = int(rng.integers(0, n - ln))
        s[p:p + ln] = revcomp(s[p:p + ln])
    for _ in range(n_ins):
        n = len(s); ln = int(rng.integers(max(2, max_event // 20), max_event)); p = int(rng.integers(0, n))
        s = np.concatenate([s[:p], random_seq(rng, ln), s[p:]])
    for _ in range(n_del):
        n = len(s); ln = int(rng.integers(max(2, max_event // 20), max_event)); ln = min(ln, n // 4)
        p = int(rng.integers(0, n - ln))
        s = np.concatenate([s[:p], s[p + ln:]])
    return s


def evolve_population(seed: int, n: int, length: int, snp: float = 0.005, indel: float = 0.0005, n_inv: int = 1, n_ins: int = 2,
                      n_del: int = 1, max_event: int = 20000, rotate: bool = False) -> list[str]:
    """n genomes related by a random binary tree (each internal branch applies `mutate`)."""
    rng = np.random.default_rng(seed)
    pool = [random_seq(rng, length)]
    while len(pool) < n:
        i = int(rng.integers(0, len(pool)))
        parent = pool.pop(i)
        for _ in range(2):
            pool.append(mutate(rng, parent, snp, indel, n_inv, n_ins, n_del, max_event))
    out = []
    for g in pool[:n]:
        if rotate:
            p = int(rng.integers(0, len(g)))
            g = np.concatenate([g[p:], g[:p]])
        out.append(g.tobytes().decode())
    return out


def sibling_pairs(seed: int, n_genomes: int, length: int, div: float):
    """n_genomes/2 independent sibling pairs at `div` pairwise divergence with 2 inversions, 6 insertions and 4 deletions each
    (the round-1 leaf-level workload: far more divergent than sibling leaves of a 1000-genome tree, hence DP-heavy).
    Returns (groups of two byte strings, decimal names)."""
    groups, names = [], []
    for g in range(n_genomes // 2):
        rng = np.random.default_rng(seed * 1000003 + g)
        anc = random_seq(rng, length)
        ev = max(2000, min(50000, length // 40))
        kids = [mutate(rng, anc, snp=div / 2, indel=div / 20, n_inv=2, n_ins=6, n_del=4, max_event=ev) for _ in range(2)]
        groups.append([k.tobytes() for k in kids])
        names.append([str(2 * g), str(2 * g + 1)])
    return groups, names
