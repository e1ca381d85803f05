import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangraph_amd import batch, schedule as sched
from pangraph_amd.levels import Population
H = int(os.environ.get("H", "8"))
pop = Population(20260928, 1000, 5_000_000)
tasks = sched.build_tasks(pop)
ts = [t for t in tasks if pop.nodes[t.node].height == H and t.round == 0][:int(os.environ.get("N", "100000"))]
first, n = {}, 0
for t in ts:
    t.prepare(); first[t.tid] = n; n += len(t.seqs)
lib = batch.ResidentBatch(sched.TaskBatch(ts))
for rep in range(3):
    if rep == 2: sys.stderr.write("==== last\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    tb = sched.TaskBatch(ts, first)
    t1 = time.perf_counter()
    rb = batch.ResidentBatch(tb, derive_from=lib)
    t2 = time.perf_counter()
    res = rb.align(sensitivity=10, want_raw=False, n_threads=8)
    t3 = time.perf_counter()
    st = res.stats
    res.close(); rb.close()
    t4 = time.perf_counter()
    print(f"h{H} r0 calls={len(ts)} n_seq={n} Mbp={sum(t.bases for t in ts)/1e6:.1f} matches={int(st['n_matches'])} | python batch {1e3*(t1-t0):.1f} derive {1e3*(t2-t1):.1f} align {1e3*(t3-t2):.1f} close {1e3*(t4-t3):.1f} ms | stages " +
          " ".join(f"{k} {1e3*st[k]:.1f}" for k in ("upload", "sketch", "index", "seed", "chain", "align", "total")))
