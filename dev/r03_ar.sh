#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_ar_$name.json 2> gpurun_out/r03_ar_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_ar_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_ar_$name.err
}
run base1 X=1
run three1 PGA_DP_THREE_LANES=1
run lo1 PGA_LANE_FLAT_PRIO=3
run base2 X=1
run three2 PGA_DP_THREE_LANES=1
run lo2 PGA_LANE_FLAT_PRIO=3
