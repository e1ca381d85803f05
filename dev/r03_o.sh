#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_levels.py -x -q -k "not c3_full_size and not c5" ) 2>&1 | tail -3
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_o_$name.json 2> gpurun_out/r03_o_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_o_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})"
}
run lanes6 X=1
run lanes4 PGA_DP_LANES4=1
run lanes6_b X=1
run lanes4_b PGA_DP_LANES4=1
