#!/bin/bash
# A/B of environment knobs on the full bench: dev/r02_ab.sh "VAR=1 VAR2=x" "..." (one bench run per argument; "" = defaults)
mkdir -p gpurun_out
for cfg in "$@"; do
  env $cfg timeout 900 python bench.py --cpu-budget 0 --steps ${STEPS:-1} --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$cfg" <<'PY'
import json, sys
try:
    d = json.load(open('gpurun_out/ab.json'))
    print("%-60s %.3f Gbp/s  %.0f ms/step  stages %s" % (sys.argv[1] or "(defaults)", d['value'], d['ms_per_step'], {k: round(v, 2) for k, v in d['stages_s'].items()}))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open('gpurun_out/ab.err').read()[-400:])
PY
done
