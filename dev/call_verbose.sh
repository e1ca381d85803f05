#!/bin/bash
# one find_matches call alone on the device, last repetition under PGA_VERBOSE=1 (host-side phase times in the library's own words)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
H=${H:-10} N=${N:-1} ROUNDSEL=${ROUNDSEL:-0} REPS=3 PGA_VERBOSE_LAST=1 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from pangraph_amd import batch, schedule as sched
from pangraph_amd.levels import Population
H = int(os.environ["H"]); N = int(os.environ["N"]); RS = int(os.environ["ROUNDSEL"])
pop = Population(20260928, 1000, 5_000_000)
tasks = sched.build_tasks(pop)
ts = sorted([t for t in tasks if pop.nodes[t.node].height == H and t.round == RS], key=lambda t: -t.bases)[:N]
first, n = {}, 0
for t in ts:
    t.prepare(); first[t.tid] = n; n += len(t.seqs)
lib = batch.ResidentBatch(sched.TaskBatch(ts))
for rep in range(4):
    if rep == 3: os.environ["PGA_VERBOSE"] = "1"; sys.stderr.write("==== verbose repetition\n"); sys.stderr.flush()
    tb = sched.TaskBatch(ts, first)
    t1 = time.perf_counter(); rb = batch.ResidentBatch(tb, derive_from=lib); t2 = time.perf_counter()
    res = rb.align(sensitivity=10, want_raw=False, n_threads=8); t3 = time.perf_counter()
    st = res.stats; res.close(); rb.close()
    print(f"h{H} r{RS} n_seq={n} Mbp={sum(t.bases for t in ts)/1e6:.1f} derive {1e3*(t2-t1):.2f} align {1e3*(t3-t2):.2f} ms | " + " ".join(f"{k} {1e3*st[k]:.2f}" for k in ("upload", "sketch", "index", "seed", "chain", "align", "total")), flush=True)
PY
