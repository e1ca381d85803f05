#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PGA_VERBOSE=1 timeout 900 python bench.py --steps 1 --warmup 0 --cpu-budget 0 --no-next-rows --schedule waves 2>&1 >/dev/null | grep "dp class 1[01]\|class 1[01]:" > gpurun_out/r03_w.txt
wc -l gpurun_out/r03_w.txt
