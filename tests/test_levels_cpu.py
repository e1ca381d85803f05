"""The level harness (pangraph_amd/levels.py): deterministic, structurally sound block sets.  CPU-only."""
import numpy as np

from pangraph_amd.levels import Population, Rates, _COMP, c2_population, waves_bases


def test_population_is_deterministic_and_maps_are_consistent():
    a = Population(7, 12, 120_000, Rates(ev_min=200, ev_max=5000))
    b = Population(7, 12, 120_000, Rates(ev_min=200, ev_max=5000))
    assert all((a.genomes[v] == b.genomes[v]).all() for v in a.leaves)
    assert len(a.leaves) == 12 and a.nodes[0].height >= 4
    # the child -> parent runs really are copies of the parent (up to substitutions)
    for nd in a.nodes[1:]:
        par = a.nodes[nd.parent]
        tot = same = 0
        for c, p, l, s in zip(nd.run_c, nd.run_p, nd.run_l, nd.run_s):
            x = nd.seq[c:c + l]
            y = par.seq[p:p + l] if s > 0 else _COMP[par.seq[p - l + 1:p + 1][::-1]]
            tot += l
            same += int((x == y).sum())
        assert tot > 0 and same / tot > 0.98


def test_waves_cover_every_merge_twice():
    pop = Population(3, 9, 80_000, Rates(ev_min=200, ev_max=4000))
    waves = pop.build_waves()
    n_merges = sum(1 for nd in pop.nodes if nd.children)
    assert n_merges == 8 and len(waves) == 2 * pop.nodes[0].height
    assert sum(len(g) for _, g, _ in waves) == 2 * n_merges
    for label, groups, names in waves:
        for g, n in zip(groups, names):
            assert len(g) == len(n) == len(set(n)) and all(x.isdigit() for x in n)
            assert all(len(a) >= 100 for a in g)
    # round 0 of a leaf pair = the two genomes; round 1 of the root = its whole pangenome
    label, groups, _ = waves[0]
    assert all(len(g) == 2 for g in groups)
    root = waves[-1][1][0]
    assert sum(len(a) for a in root) >= len(pop.nodes[0].seq)
    assert waves_bases(waves) == sum(len(a) for _, gs, _ in waves for g in gs for a in g)


def test_c2_population_shape():
    seqs, names = c2_population(n=20, length=5000)
    assert len(seqs) == 20 and len(set(names)) == 20
    d = np.mean([a != b for a, b in zip(seqs[0][:4000], seqs[1][:4000])])
    assert d < 0.01
