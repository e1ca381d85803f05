#!/bin/bash
export TMPDIR=/tmp
R=$PWD
PGA_VERBOSE=1 python dev/wide_one.py 8000 0 2>&1 | grep "dp class\|L="
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  ( cd /tmp && rm -rf /tmp/pw && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pw -o w -- python $R/dev/wide_one.py 8000 0 > /dev/null 2>&1 )
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pw/*counter_collection.csv")
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for row in csv.DictReader(open(f[0])):
    if "k_extd2_wide" in row["Kernel_Name"]:
        agg[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
print({k: (v / n[k]) for k, v in agg.items()}, "dispatches", dict(n))
PY
done
