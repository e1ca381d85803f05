#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_ac_$name.json 2> gpurun_out/r03_ac_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_ac_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_ac_$name.err
}
run s4a PGA_BENCH_SLOTS=4
run s4b PGA_BENCH_SLOTS=4
run s4cap09 PGA_BENCH_SLOTS=4 PGA_BENCH_CAP_GBP=0.9
run s4cap16 PGA_BENCH_SLOTS=4 PGA_BENCH_CAP_GBP=1.6
run s4t4 PGA_BENCH_SLOTS=4 PGA_BENCH_SLOT_THREADS=4
run s4t12 PGA_BENCH_SLOTS=4 PGA_BENCH_SLOT_THREADS=12
run s6 PGA_BENCH_SLOTS=6
