#!/bin/bash
# verbose log of the C5 build, one batch per wave (clean, sequential log)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PGA_VERBOSE=1 timeout 900 python bench.py --steps 1 --warmup 0 --cpu-budget 0 --no-next-rows --schedule waves > gpurun_out/r03_u.json 2> gpurun_out/r03_u.err
wc -l gpurun_out/r03_u.err
grep -v "workgroup kernel\|run-length walk\|pass [0-9]: \|after pass" gpurun_out/r03_u.err | gzip > gpurun_out/r03_u.err.gz
rm gpurun_out/r03_u.err
python -c "import json; d=json.load(open('gpurun_out/r03_u.json')); print(round(d['value'],3), round(d['ms_per_step'])); [print(w) for w in d['waves_rank0']]"
