#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_ah_$name.json 2> gpurun_out/r03_ah_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_ah_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_ah_$name.err
}
run base X=1
run onmain PGA_DP_ON_MAIN=1
run onmain_s8 PGA_DP_ON_MAIN=1 PGA_BENCH_SLOTS=8
run onmain_s8q8 PGA_DP_ON_MAIN=1 PGA_BENCH_SLOTS=8 GPU_MAX_HW_QUEUES=8
run onmain_s12q12 PGA_DP_ON_MAIN=1 PGA_BENCH_SLOTS=12 GPU_MAX_HW_QUEUES=12 PGA_BENCH_CAP_GBP=0.6
run base2 X=1
