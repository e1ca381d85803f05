"""Helpers for the level / block-set parity tests: the compiled reference (oracle/_ref) run group by group on the host
cores (one process per group, like bench.py's cpu_baseline), digests of record lists, synthetic high-occurrence groups."""
import hashlib
import json
import multiprocessing as mp
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmm2ref.so")


from pangraph_amd.digest import digest  # noqa: E402,F401


def _as_str(s):
    if isinstance(s, str):
        return s
    if isinstance(s, bytes):
        return s.decode()
    return s.tobytes().decode()


def _ref_worker(args):
    so, seqs, names, kw = args
    from pangraph_amd.mm2ffi import Mm2Lib
    from util import rows_to_lists
    return rows_to_lists(Mm2Lib(so).align_all(seqs, names, **kw))


def usable_cpus() -> int:
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def ref_align_groups(groups, names, so=REF_SO, procs=None, **kw):
    """record lists (util.rows_to_lists) per group, from the reference build"""
    jobs = [(so, [_as_str(s) for s in g], list(n), kw) for g, n in zip(groups, names)]
    procs = procs or min(usable_cpus(), len(jobs))
    if procs <= 1 or len(jobs) == 1:
        return [_ref_worker(j) for j in jobs]
    order = sorted(range(len(jobs)), key=lambda i: -sum(len(s) for s in jobs[i][1]))
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_ref_worker, [jobs[i] for i in order], chunksize=1)
    out = [None] * len(jobs)
    for i, r in zip(order, res):
        out[i] = r
    return out


def product_align_groups(groups, names, **kw):
    """the same through the native batch entry of the HIP library (one call for all groups)"""
    from pangraph_amd import batch
    from util import rows_to_lists
    res = batch.align_groups([[s if isinstance(s, (str, bytes)) else s.tobytes() for s in g] for g in groups], names, want_rows=True, **kw)
    return [rows_to_lists(r) for r in res.groups]


def high_occ_groups(seed=5):
    """Three groups that drive seeding through its occurrence rules (seed.c:56-96, options.c:70-76):
    (a) 520 sequences sharing one 2 kb element: every minimizer of it occurs 520 x > mid_occ (clamped to 500): high-occurrence
        streaks, of which mm_seed_select keeps the (streak length / 500 + .499) lowest-occurrence seeds;
    (b) 4200 short sequences sharing a 60 bp element (occurrence > max_max_occ = 4095: always dropped) among which 24 also share a
        1.2 kb element (ordinary hits);
    (c) 60 sequences sharing a 3 kb element at ~1 % divergence (occurrence 50..60 around min_mid_occ = 50)."""
    from pangraph_amd.synth import random_seq, mutate
    from pangraph_amd.levels import splitmix64
    rng = np.random.default_rng(seed)
    groups, names = [], []
    el = random_seq(rng, 2000)
    g = []
    for i in range(520):
        e = mutate(rng, el, snp=0.002, indel=0.0)
        g.append(np.concatenate([random_seq(rng, int(rng.integers(300, 900))), e, random_seq(rng, int(rng.integers(300, 900)))]))
    groups.append(g)
    el60, el12 = random_seq(rng, 60), random_seq(rng, 1200)
    g = []
    for i in range(4200):
        parts = [random_seq(rng, int(rng.integers(100, 250))), el60, random_seq(rng, int(rng.integers(100, 250)))]
        if i % 175 == 0:
            parts.append(mutate(rng, el12, snp=0.01, indel=0.0))
            parts.append(random_seq(rng, 150))
        g.append(np.concatenate(parts))
    groups.append(g)
    el3 = random_seq(rng, 3000)
    groups.append([np.concatenate([random_seq(rng, 500), mutate(rng, el3, snp=0.005, indel=0.0005), random_seq(rng, 500)]) for _ in range(60)])
    for gi, g in enumerate(groups):
        names.append([str(splitmix64(seed * 31 + gi * 100003 + i)) for i in range(len(g))])
    return [[a.tobytes().decode() for a in g] for g in groups], names


from pangraph_amd.digest import records_to_lists  # noqa: E402,F401  (moved: bench.py digests the step it timed with the same code)
