#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_levels.py -x -q -k "not c3_full_size" ) 2>&1 | tail -3
PGA_VERBOSE=1 H=1 N=56 timeout 300 python dev/mid_probe.py 2>&1 >/dev/null | awk "/==== last/{f=1} f" | grep "dp class 10: 2477\|n_seq=112" | cut -c1-200
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_bd_$name.json 2> gpurun_out/r03_bd_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_bd_$name.json')); k=d['roofline']['kernels']; print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()}, {n:(round(v['device_ms_per_step']),round(v['busy_ms_per_step'])) for n,v in k.items() if 'lanes' in n})" || tail -5 gpurun_out/r03_bd_$name.err
}
run a X=1
run b X=1
run c X=1
