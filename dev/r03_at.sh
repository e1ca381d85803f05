#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_levels.py tests/test_gpu_dist.py -x -q -k "not c3_full_size" ) 2>&1 | tail -3
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_at_$name.json 2> gpurun_out/r03_at_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_at_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_at_$name.err
}
run a X=1
run b X=1
run hlhl PGA_LANE_PRIO=hlhl
run lhlh PGA_LANE_PRIO=lhlh
