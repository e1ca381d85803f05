"""Level-by-level block sets of a `pangraph build` on a synthetic population.

The reference walks a guide tree bottom-up; at every internal node it joins the two child graphs and calls
`find_matches` on ALL blocks of the joined graph, merges, and repeats until nothing matches
(packages/pangraph/src/pangraph/graph_merging.rs:26-69,95-98; packages/pangraph/src/commands/build/build_run.rs:111-128).
The Rust merge loop does not run here, so this module produces the block sets it WOULD hand to the aligner:

  * a population evolves top-down along a random bifurcating tree (Yule topology and branch times) from one ancestor:
    SNPs, short indels, inversions, insertions of novel sequence (HGT), deletions and duplications (IS-like copies),
    all Poisson with rates proportional to branch time; every event is logged with its breakpoints;
  * graph(leaf) = one block, the genome (rotated: circular chromosomes start anywhere);
  * graph(v), v internal = the pangenome of the clade under v: the ancestral sequence at v cut at the breakpoints of every
    event that happened below v (lifted into v's coordinates through the per-branch coordinate maps), fragments
    shorter than `min_block` joined to their neighbour, plus one block per novel insertion that arose below v;
  * merge(v) with children c1, c2 issues two `find_matches` rounds: round 0 on blocks(c1) + blocks(c2) (the joined graph),
    round 1 on blocks(v) (the merged graph, re-indexed and re-mapped in full; it finds only paralogs, and the loop ends).
    Block names are decimal u64 BlockIds (hash-like, so strcmp order != numeric order != input order).

A WAVE is everything that can be aligned at once in a level-synchronous host: wave 2h holds round 0 of every merge of
height h, wave 2h+1 their round 1.  `build_waves` returns the waves as zero-copy views into the node sequences.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

_ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b
_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


@dataclass
class Rates:
    """Expected number of events on a root-to-leaf path (C3 = 1-2 % pairwise divergence, 5 inversions,
    20 HGT insertions, 10 deletions per path)."""
    snp: float = 0.0075          # substitutions per base per path (pairwise divergence = 2x)
    indel: float = 0.00075       # short indels per base per path (geometric length, mean 3)
    inv: float = 5.0
    hgt: float = 20.0
    dele: float = 10.0
    dup: float = 3.0             # IS-like copies of 0.8-2.5 kb (paralogs: what the second self-merge round finds)
    ev_min: int = 1000           # length range of inversions / insertions / deletions
    ev_max: int = 50000


@dataclass
class Node:
    id: int
    parent: int = -1
    children: Tuple[int, int] = ()
    t: float = 0.0               # time of the node (root 0, leaves 1)
    height: int = 0              # leaves 0; internal 1 + max(children)
    seq: Optional[np.ndarray] = None      # ASCII uint8
    # coordinate map child -> parent (built by _evolve_branch): runs of the child that are copies of parent ranges
    run_c: Optional[np.ndarray] = None    # child start of each run (sorted)
    run_p: Optional[np.ndarray] = None    # parent coordinate of the run's first child base
    run_l: Optional[np.ndarray] = None    # run length
    run_s: Optional[np.ndarray] = None    # +1 / -1
    cuts_in_parent: List[int] = field(default_factory=list)    # breakpoints of this branch's own events, parent coordinates
    novel: List[np.ndarray] = field(default_factory=list)      # sequences inserted on the branch into this node


class Population:
    """A simulated population with its true tree (used as the guide tree)."""

    def __init__(self, seed: int, n: int, length: int, rates: Rates = Rates(), rotate: bool = True):
        self.seed, self.n, self.length, self.rates = seed, n, length, rates
        rng = np.random.default_rng(seed)
        self.nodes: List[Node] = [Node(0)]
        # Yule tree, top-down: pick a random tip, split it; waiting time with k tips ~ Exp(k)
        tips, t = [0], 0.0
        while len(tips) < n:
            t += rng.exponential(1.0 / len(tips))
            i = int(rng.integers(0, len(tips)))
            v = tips.pop(i)
            self.nodes[v].t = t
            a, b = Node(len(self.nodes), parent=v), Node(len(self.nodes) + 1, parent=v)
            self.nodes += [a, b]
            self.nodes[v].children = (a.id, b.id)
            tips += [a.id, b.id]
        t += rng.exponential(1.0 / len(tips))
        for v in tips:
            self.nodes[v].t = t
        for nd in self.nodes:
            nd.t /= t
        self.leaves = [nd.id for nd in self.nodes if not nd.children]
        for nd in reversed(self.nodes):      # children have larger ids than their parent
            if nd.children:
                nd.height = 1 + max(self.nodes[c].height for c in nd.children)
        # sequences, top-down
        self.nodes[0].seq = _ALPHA[rng.integers(0, 4, size=length)]
        for nd in self.nodes[1:]:
            self._evolve_branch(rng, self.nodes[nd.parent], nd)
        self.genomes = {}
        for v in self.leaves:
            g = self.nodes[v].seq
            if rotate:
                p = int(rng.integers(0, len(g)))
                g = np.concatenate([g[p:], g[:p]])
            self.genomes[v] = g

    # ---- one branch -------------------------------------------------------------------------------------------
    def _evolve_branch(self, rng, par: Node, ch: Node) -> None:
        R, dt = self.rates, ch.t - par.t
        src = par.seq
        # piece list over the parent: [kind (0 parent / 1 novel), start, length, strand, novel id]; big events first
        pieces = [[0, 0, len(src), 1, -1]]
        novel: List[np.ndarray] = []
        cuts: List[int] = []

        def total():
            return sum(p[2] for p in pieces)

        def split_at(x):
            """make x a piece boundary; returns the index of the piece that starts at x (len(pieces) if x == total)"""
            o = 0
            for i, p in enumerate(pieces):
                if x == o:
                    return i
                if x < o + p[2]:
                    k, s, l, d, nid = p
                    a = x - o
                    if k == 0:
                        cuts.append(s + a if d > 0 else s + l - a)
                    if d > 0:
                        pieces[i:i + 1] = [[k, s, a, d, nid], [k, s + a, l - a, d, nid]]
                    else:
                        pieces[i:i + 1] = [[k, s + l - a, a, d, nid], [k, s, l - a, d, nid]]
                    return i + 1
                o += p[2]
            return len(pieces)

        def boundary_cut(i):
            # a cut that falls exactly on an existing boundary: log the parent coordinate of the piece starting there
            if i < len(pieces) and pieces[i][0] == 0:
                p = pieces[i]
                cuts.append(p[1] if p[3] > 0 else p[1] + p[2])

        def ev_len():
            return int(rng.integers(R.ev_min, R.ev_max + 1))

        events = [("inv", rng.poisson(R.inv * dt)), ("hgt", rng.poisson(R.hgt * dt)), ("del", rng.poisson(R.dele * dt)), ("dup", rng.poisson(R.dup * dt))]
        order = [k for k, c in events for _ in range(int(c))]
        order = [order[i] for i in rng.permutation(len(order))] if order else []
        for kind in order:
            n = total()
            if kind == "inv":
                ln = min(ev_len(), n // 2)
                p = int(rng.integers(0, n - ln))
                i = split_at(p); j = split_at(p + ln)
                boundary_cut(i); boundary_cut(j)
                seg = pieces[i:j][::-1]
                for q in seg:
                    q[3] = -q[3]
                pieces[i:j] = seg
            elif kind == "hgt":
                ln = ev_len()
                p = int(rng.integers(0, n + 1))
                i = split_at(p)
                boundary_cut(i)
                novel.append(_ALPHA[rng.integers(0, 4, size=ln)])
                pieces.insert(i, [1, 0, ln, 1, len(novel) - 1])
            elif kind == "del":
                ln = min(ev_len(), n // 4)
                p = int(rng.integers(0, n - ln))
                i = split_at(p); j = split_at(p + ln)
                boundary_cut(i); boundary_cut(j)
                del pieces[i:j]
            else:  # dup: copy of [p, p+ln) inserted at q
                ln = int(rng.integers(800, 2501))
                p = int(rng.integers(0, n - ln))
                i = split_at(p); j = split_at(p + ln)
                boundary_cut(i); boundary_cut(j)
                copy = [list(q) for q in pieces[i:j]]
                q = int(rng.integers(0, n + 1))
                k = split_at(q)
                boundary_cut(k)
                pieces[k:k] = copy
        # assemble
        parts = []
        for k, s, l, d, nid in pieces:
            a = src[s:s + l] if k == 0 else novel[nid][s:s + l]
            parts.append(a if d > 0 else _COMP[a[::-1]])
        asm = np.concatenate(parts) if len(parts) > 1 else parts[0].copy()
        # piece map: assembled coordinate -> parent coordinate
        pc = np.cumsum([0] + [p[2] for p in pieces])[:-1]
        # SNPs (positions may repeat; a repeat is a double hit)
        n = len(asm)
        k = int(rng.poisson(R.snp * dt * n))
        if k:
            pos = rng.integers(0, n, size=k)
            asm[pos] = _ALPHA[(_CODE[asm[pos]] + rng.integers(1, 4, size=k)) % 4]
        # short indels: runs of the assembled sequence that survive, inserted bases between them
        k = int(rng.poisson(R.indel * dt * n))
        if k:
            pos = np.unique(rng.integers(1, n, size=k))
            lens = rng.geometric(1 / 3.0, size=len(pos))
            is_ins = rng.random(len(pos)) < 0.5
            out, keep_a, last = [], [], 0     # keep_a: (asm start, length) of surviving runs
            for p, ln, ins in zip(pos.tolist(), lens.tolist(), is_ins.tolist()):
                if p < last:
                    continue
                out.append(asm[last:p]); keep_a.append((last, p - last))
                if ins:
                    out.append(_ALPHA[rng.integers(0, 4, size=ln)]); keep_a.append((-1, ln))
                    last = p
                else:
                    last = min(n, p + ln)
            out.append(asm[last:]); keep_a.append((last, n - last))
            child = np.concatenate(out)
        else:
            child, keep_a = asm, [(0, n)]
        ch.seq = child
        ch.novel = novel
        ch.cuts_in_parent = cuts
        # child -> parent runs: intersect the surviving runs with the pieces
        rc, rp, rl, rs = [], [], [], []
        co = 0
        pl = [p[2] for p in pieces]
        for a0, ln in keep_a:
            if a0 >= 0 and ln > 0:
                x, end = a0, a0 + ln
                i = int(np.searchsorted(pc, x, side="right")) - 1
                while x < end:
                    k_, s, l, d, _nid = pieces[i]
                    off = x - int(pc[i])
                    m = min(end - x, pl[i] - off)
                    if k_ == 0 and m > 0:
                        rc.append(co + (x - a0)); rl.append(m); rs.append(d)
                        rp.append(s + off if d > 0 else s + l - 1 - off)
                    x += m
                    i += 1
            co += ln
        ch.run_c, ch.run_p, ch.run_l, ch.run_s = (np.asarray(v, dtype=np.int64) for v in (rc, rp, rl, rs))

    # ---- clade pangenomes -------------------------------------------------------------------------------------------
    def lift(self, ch: Node, x: np.ndarray) -> np.ndarray:
        """child coordinates -> parent coordinates (-1 where the base has no parent: novel or inserted sequence)"""
        if len(x) == 0 or len(ch.run_c) == 0:
            return np.full(len(x), -1, dtype=np.int64)
        i = np.searchsorted(ch.run_c, x, side="right") - 1
        i = np.clip(i, 0, len(ch.run_c) - 1)
        off = x - ch.run_c[i]
        ok = (off >= 0) & (off < ch.run_l[i])
        return np.where(ok, ch.run_p[i] + ch.run_s[i] * off, -1)

    def clade_blocks(self, min_block: int = 100):
        """per node: (list of block arrays (views), list of decimal names).  Computed bottom-up; see the module docstring."""
        cuts = {}     # node -> np.ndarray of cut positions in the node's own coordinates
        novel = {}    # node -> list of novel arrays arisen below it
        blocks = {}
        for nd in reversed(self.nodes):
            if not nd.children:
                cuts[nd.id], novel[nd.id] = np.zeros(0, dtype=np.int64), []
                blocks[nd.id] = ([self.genomes[nd.id]], [str(splitmix64(self.seed * 1000003 + nd.id * 4099))])
                continue
            cs, nv = [], []
            for c in nd.children:
                chn = self.nodes[c]
                cs.append(np.asarray(chn.cuts_in_parent, dtype=np.int64))
                l = self.lift(chn, cuts[c])
                cs.append(l[l >= 0])
                nv += chn.novel + novel[c]
            allc = np.unique(np.concatenate(cs)) if cs else np.zeros(0, dtype=np.int64)
            cuts[nd.id], novel[nd.id] = allc, nv
            n = len(nd.seq)
            keep, last = [0], 0
            for x in allc.tolist():
                if x - last >= min_block and n - x >= min_block:
                    keep.append(x); last = x
            keep.append(n)
            arr = [nd.seq[a:b] for a, b in zip(keep[:-1], keep[1:])] + [a for a in nv if len(a) >= min_block]
            names = [str(splitmix64(self.seed * 1000003 + nd.id * 4099 + 7 * i + 1)) for i in range(len(arr))]
            blocks[nd.id] = (arr, names)
        return blocks

    def build_waves(self, min_block: int = 100, rounds: int = 2):
        """list of waves; a wave = (label, groups, names) with groups[g] = list of uint8 arrays, names[g] = list of decimal ids"""
        blocks = self.clade_blocks(min_block)
        hmax = self.nodes[0].height
        waves = []
        for h in range(1, hmax + 1):
            merges = [nd for nd in self.nodes if nd.children and nd.height == h]
            g0 = [blocks[nd.children[0]][0] + blocks[nd.children[1]][0] for nd in merges]
            n0 = [blocks[nd.children[0]][1] + blocks[nd.children[1]][1] for nd in merges]
            waves.append((f"height {h} round 0 ({len(merges)} merges)", g0, n0))
            if rounds > 1:
                waves.append((f"height {h} round 1 ({len(merges)} merges)", [blocks[nd.id][0] for nd in merges], [blocks[nd.id][1] for nd in merges]))
        return waves


def waves_bases(waves) -> int:
    return sum(int(len(a)) for _, groups, _ in waves for g in groups for a in g)


def c2_population(seed: int = 1, n: int = 168, length: int = 29900) -> Tuple[List[bytes], List[str]]:
    """Config C2 ("sc2-like"): n genomes of ~29.9 kb with pairwise divergence <= 0.1 %, ONE all-vs-all group.
    Every minimizer occurs ~n times: above min_mid_occ = 50, i.e. the high-occurrence path of seeding (seed.c:56-96, options.c:70-76)."""
    pop = Population(seed, n, length, Rates(snp=0.0005, indel=0.00003, inv=0, hgt=0, dele=0, dup=0), rotate=False)
    seqs = [pop.genomes[v].tobytes() for v in pop.leaves]
    names = [str(splitmix64(seed * 7919 + i)) for i in range(n)]
    return seqs, names
