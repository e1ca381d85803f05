import sys, os, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
faulthandler.dump_traceback_later(200, exit=True)
import numpy as np
from pangraph_amd import batch
from pangraph_amd.mm2ffi import Mm2Lib
from pangraph_amd.synth import random_seq, mutate, evolve_population
from util import rows_to_lists
ref = Mm2Lib('oracle/_ref/libmm2ref.so')
rng = np.random.default_rng(5)
pop = evolve_population(7, 4, 30000, snp=0.01, indel=0.001, n_inv=1, n_ins=1, n_del=1, max_event=3000)
a = pop[0]; b = pop[1]
groups = [[], [a], [a, b], ["ACGT", "ACG"], [a.lower(), b], ["N" * 500, a[:500]], [a[:5000] + "N" * 300 + a[5300:12000], b[:12000]], [a, a], []]
names = [[str(i) for i in range(len(g))] for g in groups]
res = batch.align_groups(groups, names, sensitivity=10)
for gi, (g, nm, rows) in enumerate(zip(groups, names, res.groups)):
    exp = rows_to_lists(ref.align_all(g, nm, sensitivity=10)) if g else []
    print(gi, len(g), len(rows), "OK" if rows_to_lists(rows) == exp else "DIFF", flush=True)
try:
    r = batch.align_groups([], [], sensitivity=10); print("no groups:", r.stats["n_bases"])
except Exception as e:
    print("no groups: error", e)
