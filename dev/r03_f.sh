#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_levels.py -x -q -k "sort_replay or c3_small or high_occ or plasmid or c5 or staph or c2" ) > gpurun_out/r03_j_tests1.log 2>&1
tail -4 gpurun_out/r03_j_tests1.log
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_j_bench.json 2> gpurun_out/r03_j_bench.err
python -c "import json; d=json.load(open('gpurun_out/r03_j_bench.json')); print(round(d['value'],3), round(d['ms_per_step']), {k: round(v,2) for k,v in d['stages_s'].items()}); print({k:(round(v['device_ms_per_step']),v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})"
PGA_VERBOSE=1 timeout 600 python bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-next-rows --schedule waves --leaf-only > gpurun_out/r03_j_verbose.json 2> gpurun_out/r03_j_verbose.err
grep -n "497156 records in 2 arrays" -B 24 gpurun_out/r03_j_verbose.err | grep -E "pass 0|run-length walk|workgroup kernel" | head -8
grep "sort replay:" gpurun_out/r03_j_verbose.err | tail -8
