"""How many kernels run at the same time?  Reads a rocprofv3 kernel trace (CSV) and writes the time-weighted distribution of the number of
kernels in flight, of the number of distinct hardware queues with a kernel in flight, the busy time per queue, and the sum of kernel time
over the wall time (the average concurrency).   python dev/concurrency.py <kernel_trace.csv> <out.json>
"""
import csv, json, sys, collections

rows = []
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        try:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]))
        except (KeyError, ValueError):
            continue
rows.sort()
if not rows:
    raise SystemExit("no kernel rows")
# the timed part: from the middle of the trace on would need markers; take everything and say so
t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
ev = []
for i, (a, b, q, k) in enumerate(rows):
    ev.append((a, 1, q)); ev.append((b, -1, q))
ev.sort(key=lambda e: (e[0], e[1]))
n = 0; per_q = collections.Counter(); hist = collections.Counter(); qhist = collections.Counter(); last = ev[0][0]
for t, d, q in ev:
    if t > last:
        hist[n] += t - last; qhist[sum(1 for v in per_q.values() if v > 0)] += t - last
        last = t
    n += d; per_q[q] += d
wall = t_hi - t_lo
ksum = sum(b - a for a, b, _, _ in rows)
qbusy = collections.Counter()
for q in set(r[2] for r in rows):
    iv = sorted((a, b) for a, b, qq, _ in rows if qq == q)
    lo, hi, tot = iv[0][0], iv[0][1], 0
    for a, b in iv[1:]:
        if a > hi: tot += hi - lo; lo, hi = a, b
        elif b > hi: hi = b
    qbusy[q] = tot + hi - lo
out = {"trace": sys.argv[1].split("/")[-1], "note": "whole trace (warm-up step and timed step of bench.py under rocprofv3 --kernel-trace), runtime blit kernels included",
       "kernels": len(rows), "wall_s": wall * 1e-9, "sum_of_kernel_time_s": ksum * 1e-9, "average_kernels_in_flight": ksum / wall,
       "average_kernels_in_flight_while_any_runs": ksum / max(1, wall - hist.get(0, 0)),
       "share_of_wall_time_by_kernels_in_flight": {str(k): round(v / wall, 4) for k, v in sorted(hist.items())},
       "share_of_wall_time_by_queues_with_a_kernel_in_flight": {str(k): round(v / wall, 4) for k, v in sorted(qhist.items())},
       "busy_share_per_queue": {str(q): round(v / wall, 4) for q, v in sorted(qbusy.items(), key=lambda kv: -kv[1])}}
# dispatches by family (the trace holds STEPS steps: the warm-up step and the timed one)
STEPS = 2
fam = collections.Counter(); fam_ns = collections.Counter()
for a, b, q, k in rows:
    f = "runtime copyBuffer" if "copyBuffer" in k else "runtime fillBuffer" if "fillBuffer" in k else "rocPRIM" if "rocprim" in k else "own kernels (k_*)"
    fam[f] += 1; fam_ns[f] += b - a
out["steps_in_trace"] = STEPS
out["dispatches_per_step"] = len(rows) // STEPS
out["dispatches_per_step_by_family"] = {f: {"dispatches": fam[f] // STEPS, "kernel_time_s": round(fam_ns[f] * 1e-9 / STEPS, 4)} for f in fam}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out)[:3000])
