#!/bin/bash
# quick GPU check: parity suite, then the default bench with stage timings
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -5
PGA_VERBOSE=1 timeout 600 python bench.py --cpu-budget 0 --steps ${STEPS:-2} > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err; echo "bench rc=$?"
grep -v "launching\|launched" gpurun_out/quick_bench.err | tail -${TAILN:-28}
python - <<'PY'
import json
d=json.load(open("gpurun_out/quick_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","stages_s","kernels_ms")})
PY
