#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_p_$name.json 2> gpurun_out/r03_p_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_p_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})"
}
run base X=1
run serial PGA_DP_SERIAL=1
run serial_q4 PGA_DP_SERIAL=1 GPU_MAX_HW_QUEUES=4
run base_q5 GPU_MAX_HW_QUEUES=5
run base_q7 GPU_MAX_HW_QUEUES=7
run slots2_cap17 PGA_BENCH_SLOTS=2 PGA_BENCH_CAP_GBP=1.8
