#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
nproc; cat /sys/fs/cgroup/cpu.max
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_m_$name.json 2> gpurun_out/r03_m_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_m_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})"
}
run t8 PGA_BENCH_SLOT_THREADS=8
run t16 PGA_BENCH_SLOT_THREADS=16
run t32 PGA_BENCH_SLOT_THREADS=32
run t5 PGA_BENCH_SLOT_THREADS=5
