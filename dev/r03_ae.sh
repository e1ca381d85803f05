#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_ae_$name.json 2> gpurun_out/r03_ae_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_ae_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_ae_$name.err
}
run s6q8 PGA_BENCH_SLOTS=6 GPU_MAX_HW_QUEUES=8
run s6q12 PGA_BENCH_SLOTS=6 GPU_MAX_HW_QUEUES=12
run s6q4 PGA_BENCH_SLOTS=6 GPU_MAX_HW_QUEUES=4
run s7 PGA_BENCH_SLOTS=7
run s5 PGA_BENCH_SLOTS=5
run s6 PGA_BENCH_SLOTS=6
