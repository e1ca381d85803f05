#!/bin/bash
# GPU timeline of ONE find_matches call alone on the device: every kernel of the last repetition with its start, duration and the gap since
# the previous kernel ended (any stream).   gpurun -- 'H=10 dev/call_timeline.sh <tag>'  ->  gpurun_out/<tag>_timeline.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
R=$PWD; TAG=${1:-tl}
( cd /tmp && H=${H:-10} N=${N:-1} ROUNDSEL=${ROUNDSEL:-0} timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$TAG -o t -- python $R/dev/call_timeline.py > $R/gpurun_out/${TAG}_probe.out 2> $R/gpurun_out/${TAG}_probe.err ); echo "rc=$?"
kt=$(find gpurun_out/tl_$TAG -name "*kernel_trace.csv" | head -1)
python - "$kt" gpurun_out/${TAG}_probe.out > gpurun_out/${TAG}_timeline.txt <<'PY'
import csv, sys, re
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pga::", "")[:60], r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))))
rows.sort()
# the last repetition = the kernels after the largest idle gap in the last third of the trace ... simpler: the marker kernels: take the last N where N = count / reps
out = open(sys.argv[2]).read()
m = re.search(r"reps=(\d+)", out); reps = int(m.group(1)) if m else 3
n = len(rows) // reps
last = rows[-n:]
t0 = last[0][0]
print(out.strip())
print(f"kernels in the trace {len(rows)}, per repetition ~{n}; last repetition spans {(last[-1][1] - t0) / 1e6:.3f} ms, sum of kernel time {sum(e - s for s, e, *_ in last) / 1e6:.3f} ms")
end_prev = t0
busy_until = t0; union = 0
for s, e, name, q, g, w in last:
    if s > busy_until: union += 0; 
    print(f"{(s - t0) / 1e3:10.1f} us  +{(s - end_prev) / 1e3:8.1f} gap  {(e - s) / 1e3:9.1f} us  q{q:>3} grid {g:>9} wg {w:>5}  {name}")
    end_prev = max(end_prev, e)
iv = sorted((s, e) for s, e, *_ in last); cur_s, cur_e = iv[0]; tot = 0
for s, e in iv[1:]:
    if s > cur_e: tot += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
tot += cur_e - cur_s
print(f"device busy (union of kernel intervals) {tot / 1e6:.3f} ms of {(last[-1][1] - t0) / 1e6:.3f} ms")
PY
find gpurun_out/tl_$TAG -name "*.csv" -size +5M -delete
head -5 gpurun_out/${TAG}_timeline.txt; tail -2 gpurun_out/${TAG}_timeline.txt
