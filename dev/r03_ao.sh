#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-next-rows --leaf-only > gpurun_out/r03_ao_$name.json 2> gpurun_out/r03_ao_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_ao_$name.json')); k=d['roofline']['kernels']; print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()}, {n:(round(v['device_ms_per_step']),round(v['busy_ms_per_step'])) for n,v in k.items() if 'lanes' in n or 'strips' in n})" || tail -5 gpurun_out/r03_ao_$name.err
}
run base X=1
run c10w3 PGA_C10_WAVES=3
run c10w4 PGA_C10_WAVES=4
run base2 X=1
run c10w4b PGA_C10_WAVES=4
