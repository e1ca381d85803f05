#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_av_$name.json 2> gpurun_out/r03_av_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_av_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_av_$name.err
}
run q7 GPU_MAX_HW_QUEUES=7
run q7s7 GPU_MAX_HW_QUEUES=7 PGA_BENCH_SLOTS=7 PGA_SLAB_KEEP_GB=200
run q6 GPU_MAX_HW_QUEUES=6
run q7b GPU_MAX_HW_QUEUES=7
