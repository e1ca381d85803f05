#!/bin/bash
# round-2 profiles of the BASELINE workload (bench.py defaults: 1000 x 5 Mbp, all waves): rocprofv3 kernel stats of one step, then
# the HBM traffic counters, one counter per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: dev/r02_profile.sh <tag>   (writes profiles/r02_<tag>_*)
TAG=${1:-x}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out profiles gpurun_out/profiles_out
( cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -o r02 -- python $R/bench.py --cpu-budget 0 --steps 1 --warmup 1 > $R/gpurun_out/r02_bench_prof.json 2> $R/gpurun_out/r02_bench_prof.err ); echo "prof rc=$?"
f=$(find gpurun_out/prof_r02 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" profiles/r02_${TAG}_c5_kernel_stats.csv
cp gpurun_out/r02_bench_prof.json profiles/r02_${TAG}_bench_c5_under_rocprof.json
find gpurun_out/prof_r02 -name "*.db" -size +20M -delete; find gpurun_out/prof_r02 -name "*kernel_trace.csv" -size +20M -delete
if [ "${PMC:-1}" = "1" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 1500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/bench.py --cpu-budget 0 --steps 1 --warmup 0 > $R/gpurun_out/pmc_$c.json 2> $R/gpurun_out/pmc_$c.err ); echo "$c rc=$?"
done
python - <<PY
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print("no counter file for", c); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0]
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    out[c] = {k: {"sum": v[0], "dispatches": v[1]} for k, v in agg.items()}
    top = sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]
    print(c); [print("  %-70s sum %.4g KB over %d dispatches" % (k[:70], v[0], v[1])) for k, v in top]
json.dump(out, open("profiles/r02_${TAG}_pmc_hbm_traffic_c5.json", "w"), indent=1)
PY
find gpurun_out/pmc_* -name "*.csv" -size +20M -delete
fi
# profiles/ on the GPU box is not merged back: hand the artifacts over through gpurun_out/
cp profiles/r02_${TAG}_* gpurun_out/profiles_out/ 2>/dev/null
head -25 profiles/r02_${TAG}_c5_kernel_stats.csv | cut -c1-200
