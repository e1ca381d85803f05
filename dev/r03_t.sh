#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_levels.py -x -q -k "resident or c3_small or c2" ) 2>&1 | tail -5
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_t_$name.json 2> gpurun_out/r03_t_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_t_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), d.get('inputs_made_resident_s'), {a: round(b,2) for a,b in d['stages_s'].items()}, d['rank0_seconds_per_step']['hand_over'])" || tail -5 gpurun_out/r03_t_$name.err
}
run resident PGA_BENCH_INPUTS=resident
run host PGA_BENCH_INPUTS=host
run resident2 PGA_BENCH_INPUTS=resident
