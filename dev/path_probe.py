"""What is the CRITICAL PATH of the build?  Every find_matches call of tree height >= HMIN alone on the device (one batch per call, inputs
resident), its wall time and stage split; then the longest dependency chain through those calls with the measured times, and the slowest
calls (set VERBOSE_TOP=k to re-run the k slowest under PGA_VERBOSE=1 on stderr).
    HMIN=5 python dev/path_probe.py          (VERBOSE_TOP=k: the k slowest calls of the chain again under PGA_VERBOSE=1; VERBOSE_CALL=h17r1: that one)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangraph_amd import batch, schedule as sched
from pangraph_amd.levels import Population

HMIN = int(os.environ.get("HMIN", "5"))
pop = Population(20260928, 1000, 5_000_000)
tasks = sched.build_tasks(pop)
sel = [t for t in tasks if pop.nodes[t.node].height >= HMIN]
first, n = {}, 0
for t in sel:
    t.prepare(); first[t.tid] = n; n += len(t.seqs)
lib = batch.ResidentBatch(sched.TaskBatch(sel))
solo = {}


def run(t):
    tb = sched.TaskBatch([t], first)
    t0 = time.perf_counter()
    rb = batch.ResidentBatch(tb, derive_from=lib)
    res = rb.align(sensitivity=10, want_raw=False, n_threads=8)
    dt = time.perf_counter() - t0
    st = res.stats
    res.close(); rb.close()
    return dt, st


for rep in range(2):
    for t in sel:
        solo[t.tid] = run(t)
# longest chain (by measured solo time) from any selected task to the root
best = {}
for t in sorted(sel, key=lambda t: -t.tid):          # children have larger tids? not guaranteed: do it by recursion
    pass


def chain(tid):
    if tid in best:
        return best[tid]
    t = tasks[tid]
    up = max((chain(u) for u in t.users if u in solo), default=(0.0, []))
    best[tid] = (solo[tid][0] + up[0], [tid] + up[1])
    return best[tid]


sys.setrecursionlimit(10000)
tot, path = max((chain(t.tid) for t in sel), key=lambda x: x[0])
print(f"calls of height >= {HMIN}: {len(sel)}, sum of solo times {sum(v[0] for v in solo.values()):.3f} s, longest dependency chain {tot:.3f} s over {len(path)} calls")
keys = ("upload", "sketch", "index", "seed", "chain", "align")
agg = {k: sum(solo[tid][1][k] for tid in path) for k in keys}
print("stage sums along the chain: " + " ".join(f"{k} {agg[k]:.3f}" for k in keys))
for tid in path:
    t = tasks[tid]; dt, st = solo[tid]
    print(f"  h{pop.nodes[t.node].height:2d} r{t.round} n_seq={len(t.seqs):6d} Mbp={t.bases / 1e6:6.1f} matches={int(st['n_matches']):5d} anchors={int(st['n_anchors']):8d} dp_jobs={int(st['n_dp_jobs']):6d} | {1e3 * dt:7.1f} ms | " +
          " ".join(f"{k} {1e3 * st[k]:.1f}" for k in keys))
k = int(os.environ.get("VERBOSE_TOP", "0"))
want = os.environ.get("VERBOSE_CALL")                      # e.g. "h17r1": that call of the chain again under PGA_VERBOSE=1
if want:
    os.environ["PGA_VERBOSE"] = "1"
    for tid in path:
        t = tasks[tid]
        if f"h{pop.nodes[t.node].height}r{t.round}" == want:
            sys.stderr.write(f"==== call {want} n_seq={len(t.seqs)} Mbp={t.bases / 1e6:.1f}: {1e3 * solo[tid][0]:.1f} ms alone\n"); sys.stderr.flush()
            run(t)
    os.environ.pop("PGA_VERBOSE")
if k:
    os.environ["PGA_VERBOSE"] = "1"
    for tid in sorted(path, key=lambda i: -solo[i][0])[:k]:
        t = tasks[tid]
        sys.stderr.write(f"==== call h{pop.nodes[t.node].height} r{t.round} n_seq={len(t.seqs)} Mbp={t.bases / 1e6:.1f}: {1e3 * solo[tid][0]:.1f} ms alone\n"); sys.stderr.flush()
        run(t)
