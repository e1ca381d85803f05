#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_n_$name.json 2> gpurun_out/r03_n_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_n_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})"
}
run base X=1
run tail4 PGA_TAIL_QUERIES=4 PGA_TAIL_THREADS=2
run tail8 PGA_TAIL_QUERIES=8 PGA_TAIL_THREADS=3
run tail16_q8 PGA_TAIL_QUERIES=16 PGA_TAIL_THREADS=4 GPU_MAX_HW_QUEUES=8
