import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangraph_amd import batch, schedule as sched
from pangraph_amd.levels import Population
pop = Population(20260928, 1000, 5_000_000)
tasks = sched.build_tasks(pop)
top = sorted(tasks, key=lambda t: -pop.nodes[t.node].height)[:2]
first, n = {}, 0
for t in top:
    t.prepare(); first[t.tid] = n; n += len(t.seqs)
lib = batch.ResidentBatch(sched.TaskBatch(top))
for rep in range(3):
    for t in top:
        if rep == 2: sys.stderr.write(f"==== h{pop.nodes[t.node].height} r{t.round}\n"); sys.stderr.flush()
        rb = batch.ResidentBatch(sched.TaskBatch([t], first), derive_from=lib)
        res = rb.align(sensitivity=10, want_raw=False, n_threads=8)
        res.close(); rb.close()
