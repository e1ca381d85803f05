#!/bin/bash
# round 3: hardware queues x ready-set schedule
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_b_$name.json 2> gpurun_out/r03_b_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_b_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {k: round(v,2) for k,v in d['stages_s'].items()})"
}
run q4_s3 GPU_MAX_HW_QUEUES=4 PGA_BENCH_SLOTS=3
run q8_s3 GPU_MAX_HW_QUEUES=8 PGA_BENCH_SLOTS=3
run q12_s3 GPU_MAX_HW_QUEUES=12 PGA_BENCH_SLOTS=3
run q16_s3 GPU_MAX_HW_QUEUES=16 PGA_BENCH_SLOTS=3
run q24_s4 GPU_MAX_HW_QUEUES=24 PGA_BENCH_SLOTS=4 PGA_BENCH_CAP_GBP=0.8
run q6_s3_sets1 GPU_MAX_HW_QUEUES=6 PGA_BENCH_SLOTS=3 PGA_ALIGN_SETS=1
run q12_s3_sets1 GPU_MAX_HW_QUEUES=12 PGA_BENCH_SLOTS=3 PGA_ALIGN_SETS=1
run q12_s4_sets1 GPU_MAX_HW_QUEUES=12 PGA_BENCH_SLOTS=4 PGA_ALIGN_SETS=1 PGA_BENCH_CAP_GBP=0.8
