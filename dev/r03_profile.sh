#!/bin/bash
# round-3 profiles of the BASELINE workload (bench.py defaults: 1000 x 5 Mbp, all waves): rocprofv3 kernel stats of one step, then
# the HBM traffic counters, one counter per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: dev/r03_profile.sh <tag>   (writes profiles/r03_<tag>_*)
TAG=${1:-x}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out profiles gpurun_out/profiles_out
( cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03 -o r03 -- python $R/bench.py --cpu-budget 0 --steps 1 --warmup 1 > $R/gpurun_out/r03_bench_prof.json 2> $R/gpurun_out/r03_bench_prof.err ); echo "prof rc=$?"
f=$(find gpurun_out/prof_r03 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" profiles/r03_${TAG}_c5_kernel_stats.csv
cp gpurun_out/r03_bench_prof.json profiles/r03_${TAG}_bench_c5_under_rocprof.json
find gpurun_out/prof_r03 -name "*.db" -size +20M -delete; find gpurun_out/prof_r03 -name "*kernel_trace.csv" -size +20M -delete
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
out["_meta"] = {"steps": 1, "warmup": 0, "note": "one step of the BASELINE build per pass; sums are KB over all dispatches of the step"}
json.dump(out, open("profiles/r03_${TAG}_pmc_hbm_traffic_c5.json", "w"), indent=1)
PY
find gpurun_out/pmc_* -name "*.csv" -size +20M -delete
fi
# profiles/ on the GPU box is not merged back: hand the artifacts over through gpurun_out/
cp profiles/r03_${TAG}_* gpurun_out/profiles_out/ 2>/dev/null
( timeout 900 python bench.py > gpurun_out/profiles_out/r03_${TAG}_bench_c5.json 2> gpurun_out/r03_bench.err ); echo "bench rc=$?"
head -25 profiles/r03_${TAG}_c5_kernel_stats.csv | cut -c1-200
python -c "import json; d=json.load(open('gpurun_out/profiles_out/r03_${TAG}_bench_c5.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'])"
