#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_ab_$name.json 2> gpurun_out/r03_ab_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_ab_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_ab_$name.err
}
run s3 PGA_BENCH_SLOTS=3
run s4 PGA_BENCH_SLOTS=4
run s5 PGA_BENCH_SLOTS=5
run s4q8 PGA_BENCH_SLOTS=4 GPU_MAX_HW_QUEUES=8
run s3cap06 PGA_BENCH_SLOTS=3 PGA_BENCH_CAP_GBP=0.6
run s4cap06 PGA_BENCH_SLOTS=4 PGA_BENCH_CAP_GBP=0.6
