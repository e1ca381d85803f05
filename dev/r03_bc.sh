#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_bc_$name.json 2> gpurun_out/r03_bc_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_bc_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_bc_$name.err
}
run base X=1
run mainhi PGA_MAIN_PRIO=h PGA_LANE_PRIO=lnnl
run mainhi2 PGA_MAIN_PRIO=h PGA_LANE_PRIO=nlln
run base2 X=1
run mainhi3 PGA_MAIN_PRIO=h PGA_LANE_PRIO=lnnl
