#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_levels.py -x -q -k "not c3_full_size" ) 2>&1 | tail -4
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_al_$name.json 2> gpurun_out/r03_al_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_al_$name.json')); k=d['roofline']['kernels']; print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()}, {n:(round(v['device_ms_per_step']),round(v['busy_ms_per_step'])) for n,v in k.items() if 'bt' in n})" || tail -5 gpurun_out/r03_al_$name.err
}
run a X=1
run b X=1
