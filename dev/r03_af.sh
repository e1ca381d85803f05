#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_af_$name.json 2> gpurun_out/r03_af_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_af_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_af_$name.err
}
run base X=1
run dpserial PGA_DP_SERIAL=1
run nobig PGA_RS_NO_BIG=1
run c10w1 PGA_C10_WAVES=1
run c11w4 PGA_C11_WAVES=4
run nonarrow PGA_NO_LANES_NARROW=1
run base2 X=1
