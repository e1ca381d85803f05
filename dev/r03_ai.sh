#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_ai_$name.json 2> gpurun_out/r03_ai_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_ai_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_ai_$name.err
}
run s8t2 PGA_BENCH_SLOTS=8 PGA_BENCH_SLOT_THREADS=2
run s8t4 PGA_BENCH_SLOTS=8 PGA_BENCH_SLOT_THREADS=4
run s6a X=1
run s6b X=1
run s6c X=1
