#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_aj_$name.json 2> gpurun_out/r03_aj_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_aj_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_aj_$name.err
  rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
}
run s8k200 PGA_BENCH_SLOTS=8 PGA_SLAB_KEEP_GB=200
run s7k200 PGA_BENCH_SLOTS=7 PGA_SLAB_KEEP_GB=200
run s10k200 PGA_BENCH_SLOTS=10 PGA_SLAB_KEEP_GB=200
run s6k200 PGA_BENCH_SLOTS=6 PGA_SLAB_KEEP_GB=200
run s8k200q8 PGA_BENCH_SLOTS=8 PGA_SLAB_KEEP_GB=200 GPU_MAX_HW_QUEUES=8
