#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_aw_$name.json 2> gpurun_out/r03_aw_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_aw_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']))" || tail -5 gpurun_out/r03_aw_$name.err
}
for i in 1 2 3; do
run q6_$i GPU_MAX_HW_QUEUES=6
run q7_$i GPU_MAX_HW_QUEUES=7
done
