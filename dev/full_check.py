"""One-off validation: N full-size bench pairs through the native batch entry vs the reference build, group by group."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from pangraph_amd.synth import sibling_pairs
from pangraph_amd import batch
from pangraph_amd.mm2ffi import Mm2Lib
from util import rows_to_lists
n_genomes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ref = Mm2Lib('oracle/_ref/libmm2ref.so')
groups, names = sibling_pairs(20260928, n_genomes, 5_000_000, 0.01)
t0 = time.time()
res = batch.align_groups([[s.decode() for s in g] for g in groups], names, sensitivity=10)
print(f"gpu: {time.time()-t0:.1f} s", flush=True)
bad = 0; n = 0
t0 = time.time()
for gi, (g, nm, rows) in enumerate(zip(groups, names, res.groups)):
    exp = rows_to_lists(ref.align_all([s.decode() for s in g], nm, sensitivity=10))
    got = rows_to_lists(rows)
    n += len(exp)
    if got != exp:
        bad += 1
        print("group", gi, "DIFFERS:", len(got), "vs", len(exp), flush=True)
print(f"ref: {time.time()-t0:.1f} s; {n} records in {len(groups)} groups; {bad} groups differ")
