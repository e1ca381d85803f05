#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q ) 2>&1 | tail -3
H=8 timeout 300 python dev/mid_probe.py 2>/dev/null | tail -1
PGA_VERBOSE=1 H=8 timeout 300 python dev/mid_probe.py 2>&1 >/dev/null | awk "/==== last/{f=1} f" | grep "dp class 1[01]" | cut -c1-120
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_an_$name.json 2> gpurun_out/r03_an_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_an_$name.json')); k=d['roofline']['kernels']; print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()}, {n:(round(v['device_ms_per_step']),round(v['busy_ms_per_step'])) for n,v in k.items() if 'lanes' in n})" || tail -5 gpurun_out/r03_an_$name.err
}
run a X=1
run b X=1
run c X=1
