#!/bin/bash
# dev/solo_stats.sh [slots]: kernel stats of one step with that many batches in flight (default 1), beside the usual six: which kernels stretch when batches share the device.
# Output: gpurun_out/profiles_out/${ROUND}_slots${S}_c5_kernel_stats.csv
R=$(cd "$(dirname "$0")/.." && pwd); ROUND=${ROUND:-r05}; S=${1:-1}
mkdir -p $R/gpurun_out/profiles_out
( cd /tmp && export TMPDIR=/tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_slots$S -o p -- python $R/bench.py --cpu-budget 0 --no-next-rows --no-resident-rate --no-parity-check --slots $S --steps 1 --warmup 1 > $R/gpurun_out/prof_slots$S.json 2> $R/gpurun_out/prof_slots$S.err ); echo "rc=$?"
f=$(find $R/gpurun_out/prof_slots$S -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/profiles_out/${ROUND}_slots${S}_c5_kernel_stats.csv
find $R/gpurun_out/prof_slots$S \( -name "*.db" -o -name "*kernel_trace.csv" \) -size +20M -delete
tail -1 $R/gpurun_out/prof_slots$S.json | cut -c1-300
