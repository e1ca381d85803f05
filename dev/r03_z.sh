#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PGA_VERBOSE=1 timeout 900 python bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-next-rows --schedule waves --leaf-only 2>&1 >/dev/null | grep "chain stage\|sort replay:\|backtrack:\|chain:\|n_seq=" > gpurun_out/r03_z.txt
wc -l gpurun_out/r03_z.txt
