#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_levels.py tests/test_gpu_parity.py -x -q -k "not c3_full_size" ) 2>&1 | tail -3
H=8 timeout 300 python dev/mid_probe.py 2>/dev/null | tail -1
PGA_RS_OLD_DIGIT_WALK=1 H=8 timeout 300 python dev/mid_probe.py 2>/dev/null | tail -1
H=1 N=56 timeout 300 python dev/mid_probe.py 2>/dev/null | tail -1
PGA_RS_OLD_DIGIT_WALK=1 H=1 N=56 timeout 300 python dev/mid_probe.py 2>/dev/null | tail -1
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_ba_$name.json 2> gpurun_out/r03_ba_$name.err
  python -c "import json; d=json.load(open('gpurun_out/r03_ba_$name.json')); print('$name', round(d['value'],3), round(d['ms_per_step']), {a: round(b,2) for a,b in d['stages_s'].items()})" || tail -5 gpurun_out/r03_ba_$name.err
}
run new1 X=1
run old1 PGA_RS_OLD_DIGIT_WALK=1
run new2 X=1
run old2 PGA_RS_OLD_DIGIT_WALK=1
