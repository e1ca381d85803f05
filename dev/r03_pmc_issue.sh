#!/bin/bash
# instruction-issue counters of the path's kernels on the BASELINE workload, one counter per pass (SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count
# quad-cycles per wave, MI355X_MICROARCH.md).  usage: dev/r03_pmc_issue.sh <tag>  -> gpurun_out/profiles_out/r03_<tag>_pmc_issue_dp_kernels.json
TAG=${1:-x}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/profiles_out
for c in SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmci_$c -o pmc -- python $R/bench.py --cpu-budget 0 --steps 1 --warmup 0 --no-next-rows > /dev/null 2> $R/gpurun_out/pmci_$c.err ); echo "$c rc=$?"
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(dict)
for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU"):
    f = glob.glob(f"gpurun_out/pmci_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print("no counter file for", c); continue
    tot = collections.defaultdict(float)
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k] += float(row["Counter_Value"])
    for k, v in tot.items(): agg[k][c] = v
out = {}
for k, v in agg.items():
    if not any(s in k for s in ("k_extd2", "k_gapfill", "k_ll_i16", "k_approx_strips", "k_rs_pass", "k_chain_fast", "k_sketch_tiles", "k_bt_walk", "k_cigar_finish")): continue
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    v["valu_issue_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc if wc else None
    v["lds_issue_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_LDS", 0.0) / wc if wc else None
    out[k] = v
    print("%-40s VALU issue %.3f  LDS issue %.3f  of wave cycles;  insts VALU %.3g SALU %.3g" % (k[:40], v["valu_issue_frac_of_wave_cycles"] or 0, v["lds_issue_frac_of_wave_cycles"] or 0, v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0)))
json.dump({"_source": "dev/r03_pmc_issue.sh: one step of the BASELINE build per counter pass (six batches in flight)", "kernels": out}, open("gpurun_out/profiles_out/r03_${TAG}_pmc_issue_dp_kernels.json", "w"), indent=1)
PY
find gpurun_out/pmci_* -name "*.csv" -size +20M -delete
