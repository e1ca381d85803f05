#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_l_$name.json 2> gpurun_out/r03_l_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_l_$name.json')); k=d['roofline']['kernels']; print('$name', round(d['value'],3), round(d['ms_per_step']), {n:round(v['device_ms_per_step']) for n,v in k.items() if 'rs_' in n}, {a: round(b,2) for a,b in d['stages_s'].items()})"
}
run big16k PGA_RS_BIG_MIN=16384
run big4k PGA_RS_BIG_MIN=4096
run big2k PGA_RS_BIG_MIN=2048
run big2k_g1024 PGA_RS_BIG_MIN=2048 PGA_RS_BIG_GRID=1024
run nobig PGA_RS_NO_BIG=1
run big16k_waves PGA_RS_BIG_MIN=16384 PGA_BENCH_SCHEDULE=waves
run nobig_waves PGA_RS_NO_BIG=1 PGA_BENCH_SCHEDULE=waves
