#!/bin/bash
# round 3, first GPU call: the new full-size parity tests, the ready-set schedule against the level-synchronous one, a verbose log of one step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_levels.py::test_c5_every_wave tests/test_gpu_dist.py::test_c4_every_wave_two_ranks_gather_vs_reference_digests -x -q ) > gpurun_out/r03_a_tests.log 2>&1
for mode in "waves 3 1.2" "ready 3 1.2" "ready 2 1.2" "ready 4 0.8" "ready 6 0.6"; do
  set -- $mode
  timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows --schedule $1 --slots $2 --cap-gbp $3 > gpurun_out/r03_a_bench_$1_$2_$3.json 2> gpurun_out/r03_a_bench_$1_$2_$3.err
done
PGA_VERBOSE=1 timeout 600 python bench.py --steps 1 --warmup 1 --cpu-budget 0 --no-next-rows --schedule waves > gpurun_out/r03_a_verbose.json 2> gpurun_out/r03_a_verbose.err
tail -5 gpurun_out/r03_a_tests.log
for f in gpurun_out/r03_a_bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'])"; done
