#!/bin/bash
# HIP runtime API statistics of one timed step of the BASELINE workload (host-side cost of the path: launches, copies, synchronisations).
#   gpurun -- 'dev/hip_api_stats.sh <tag>'  ->  gpurun_out/profiles_out/<ROUND>_<tag>_hip_api_stats.csv
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
R=$PWD; ROUND=${ROUND:-r05}; TAG=${1:-x}
mkdir -p gpurun_out/profiles_out
( cd /tmp && timeout 1500 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $R/gpurun_out/hipapi_$TAG -o h -- python $R/bench.py --cpu-budget 0 --no-next-rows --no-resident-rate --no-parity-check --steps 1 --warmup 1 --detail $R/gpurun_out/hipapi_${TAG}_detail.json > $R/gpurun_out/hipapi_$TAG.json 2> $R/gpurun_out/hipapi_$TAG.err ); echo "rc=$?"
f=$(find gpurun_out/hipapi_$TAG -name "*hip_api_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/profiles_out/${ROUND}_${TAG}_hip_api_stats.csv && head -30 "$f" | cut -c1-150
find gpurun_out/hipapi_$TAG \( -name "*.db" -o -name "*trace.csv" \) -size +20M -delete
tail -2 gpurun_out/hipapi_$TAG.json | cut -c1-300
