"""one find_matches call of tree height H (round ROUNDSEL) alone on the device, REPS times (see dev/call_timeline.sh)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangraph_amd import batch, schedule as sched
from pangraph_amd.levels import Population
H = int(os.environ.get("H", "10")); N = int(os.environ.get("N", "1")); RS = int(os.environ.get("ROUNDSEL", "0")); REPS = int(os.environ.get("REPS", "4"))
pop = Population(20260928, 1000, 5_000_000)
tasks = sched.build_tasks(pop)
ts = [t for t in tasks if pop.nodes[t.node].height == H and t.round == RS]
ts.sort(key=lambda t: -t.bases)
ts = ts[:N]
first, n = {}, 0
for t in ts:
    t.prepare(); first[t.tid] = n; n += len(t.seqs)
lib = batch.ResidentBatch(sched.TaskBatch(ts))
for rep in range(REPS):
    tb = sched.TaskBatch(ts, first)
    t1 = time.perf_counter()
    rb = batch.ResidentBatch(tb, derive_from=lib)
    t2 = time.perf_counter()
    res = rb.align(sensitivity=10, want_raw=False, n_threads=8)
    t3 = time.perf_counter()
    st = res.stats
    res.close(); rb.close()
    print(f"reps={REPS} h{H} r{RS} calls={len(ts)} n_seq={n} Mbp={sum(t.bases for t in ts)/1e6:.1f} matches={int(st['n_matches'])} anchors={int(st['n_anchors'])} dp_jobs={int(st['n_dp_jobs'])} | derive {1e3*(t2-t1):.1f} align {1e3*(t3-t2):.1f} ms | stages " +
          " ".join(f"{k} {1e3*st[k]:.1f}" for k in ("upload", "sketch", "index", "seed", "chain", "align", "total")))
