#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lanes or one_wave or e2e" ) 2>&1 | tail -2
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_az_$name.json 2> gpurun_out/r03_az_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_az_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_az_$name.err
}
run a X=1
run b X=1
run c X=1
