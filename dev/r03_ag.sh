#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_mash.py -x -q ) 2>&1 | tail -8
timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_ag.json 2> gpurun_out/r03_ag.err
python -c "import json; d=json.load(open('gpurun_out/r03_ag.json')); r=d['roofline']; print(round(d['value'],3), round(d['ms_per_step']), r['any_kernel_busy_ms_per_step'], r['timed_intervals_per_step']); print({n:(round(v['device_ms_per_step']),round(v['busy_ms_per_step'])) for n,v in r['kernels'].items()})" || tail -5 gpurun_out/r03_ag.err
