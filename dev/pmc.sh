#!/bin/bash
# HBM traffic counters of the default workload, one counter per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
export TMPDIR=/tmp
R=$PWD
G=${GENOMES:-512}
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 1200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/bench.py --cpu-budget 0 --steps 1 --warmup 1 --genomes $G > $R/gpurun_out/pmc_$c.json 2> $R/gpurun_out/pmc_$c.err ); echo "$c rc=$?"
  ls -la gpurun_out/pmc_$c | head
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/*counter_collection.csv")
    if not f: print("no counter file for", c); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0]
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    out[c] = {k: {"sum": v[0], "dispatches": v[1]} for k, v in agg.items()}
    top = sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]
    print(c); [print("  %-60s sum %.4g over %d dispatches" % (k[:60], v[0], v[1])) for k, v in top]
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1)
PY
