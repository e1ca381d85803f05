#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_au_$name.json 2> gpurun_out/r03_au_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_au_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_au_$name.err
}
run q8 GPU_MAX_HW_QUEUES=8
run q12 GPU_MAX_HW_QUEUES=12
run q5 GPU_MAX_HW_QUEUES=5
run q6 GPU_MAX_HW_QUEUES=6
