#!/bin/bash
# SURVEY 8(f)-1 kernel (k_mapvar): rocprofv3 kernel stats and counters of dev/f1_bench.py (200 blocks x 100 members x ~10 kb, band 20), one counter per pass
# usage: [BAND=20] dev/r02_f1_pmc.sh <tag>  -> gpurun_out/profiles_out/r02_<tag>_f1_mapvar_{kernel_stats.csv,pmc.json}
TAG=${1:-x}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/profiles_out
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f1prof -o f1 -- python $R/dev/f1_bench.py 200 100 10000 ${BAND:-20} > $R/gpurun_out/f1_prof.json 2>/dev/null )
f=$(find gpurun_out/f1prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/profiles_out/r02_${TAG}_f1_mapvar_kernel_stats.csv
for c in SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/f1pmc_$c -o pmc -- python $R/dev/f1_bench.py 200 100 10000 ${BAND:-20} > /dev/null 2> $R/gpurun_out/f1pmc_$c.err ); echo "$c rc=$?"
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(dict)
for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/f1pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print("no counter file for", c); continue
    tot = collections.defaultdict(float); nd = collections.defaultdict(int)
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k] += float(row["Counter_Value"]); nd[k] += 1
    for k, v in tot.items(): agg[k][c] = v; agg[k]["dispatches"] = nd[k]
out = {}
for k, v in agg.items():
    if "k_mapvar" not in k and "k_mv_encode" not in k: continue
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    v["valu_issue_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc if wc else None
    v["lds_issue_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_LDS", 0.0) / wc if wc else None
    # MI355X_MICROARCH.md, HBM section: the counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced read: doubled here, WRITE_SIZE as it is
    v["hbm_bytes_fetch_x2_plus_write"] = (v.get("FETCH_SIZE", 0.0) * 2 + v.get("WRITE_SIZE", 0.0)) * 1024 if ("FETCH_SIZE" in v or "WRITE_SIZE" in v) else None
    out[k] = v
    print(k[:50], json.dumps({a: b for a, b in v.items()}))
json.dump(out, open("gpurun_out/profiles_out/r02_${TAG}_f1_mapvar_pmc.json", "w"), indent=1)
PY
rm -rf gpurun_out/f1prof gpurun_out/f1pmc_*
