#!/bin/bash
# verbose timeline of the calls of a 16-genome build, one wave at a time (the top waves are whole-genome calls like those of the C5 critical path)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PGA_VERBOSE=1 PGA_CHAIN_PROF=1 timeout 900 python bench.py --genomes 16 --length 5000000 --steps 1 --warmup 1 --cpu-budget 0 --no-next-rows --schedule waves > gpurun_out/r03_s.json 2> gpurun_out/r03_s.err
wc -l gpurun_out/r03_s.err
python -c "import json; d=json.load(open('gpurun_out/r03_s.json')); print(round(d['value'],3), round(d['ms_per_step']))"
