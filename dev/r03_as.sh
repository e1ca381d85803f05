#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_as_$name.json 2> gpurun_out/r03_as_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_as_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_as_$name.err
}
run v3a PGA_LANE_FLAT_PRIO=3
run v5a PGA_LANE_FLAT_PRIO=5
run v3b PGA_LANE_FLAT_PRIO=3
run v5b PGA_LANE_FLAT_PRIO=5
run v3s7 PGA_LANE_FLAT_PRIO=3 PGA_BENCH_SLOTS=7 PGA_SLAB_KEEP_GB=200
run v5s7 PGA_LANE_FLAT_PRIO=5 PGA_BENCH_SLOTS=7 PGA_SLAB_KEEP_GB=200
