#!/bin/bash
# throughput vs batch size
for g in ${GENOMES:-256 512}; do
  s=$(date +%s.%N)
  timeout 900 python bench.py --cpu-budget 0 --steps 2 --warmup 1 --genomes $g > gpurun_out/big_$g.json 2> gpurun_out/big_$g.err
  e=$(date +%s.%N)
  python - <<PY
import json
d=json.load(open("gpurun_out/big_$g.json"))
print($g, "wall %.1f s" % ($e-$s), d["value"], d["ms_per_step"], d["stages_s"], d["kernels_ms"])
PY
done
