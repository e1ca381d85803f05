#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
PGA_VERBOSE=1 timeout 600 python bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-next-rows --schedule waves --leaf-only > gpurun_out/r03_h_verbose.json 2> gpurun_out/r03_h_verbose.err
grep -n "497156 records in 2 arrays" -B 24 gpurun_out/r03_h_verbose.err | grep -E "pass 0|run-length walk|workgroup kernel" | head -8
