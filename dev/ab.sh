#!/bin/bash
# A/B of environment settings on the default bench: ab.sh "VAR=1" "VAR=2 OTHER=x" ...
mkdir -p gpurun_out
for cfg in "$@"; do
  env $cfg PGA_VERBOSE=1 timeout 600 python bench.py --cpu-budget 0 --steps ${STEPS:-3} > gpurun_out/ab.json 2> gpurun_out/ab.err
  echo "== $cfg: $(python -c "import json;d=json.load(open('gpurun_out/ab.json'));print(d['value'],d['ms_per_step'],d['stages_s'])")"
done
