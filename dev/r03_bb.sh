#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_bb_$name.json 2> gpurun_out/r03_bb_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_bb_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_bb_$name.err
}
run t16a PGA_BENCH_SLOT_THREADS=16
run t8a X=1
run t16b PGA_BENCH_SLOT_THREADS=16
run t8b X=1
