#!/bin/bash
# round-1 deliverables on the GPU box: parity suite, smoke, default bench, the same bench under rocprofv3
mkdir -p gpurun_out
echo "nproc=$(nproc) cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" > gpurun_out/r01_env.txt
python -c "import os;print('cpu_count',os.cpu_count(),'affinity',len(os.sched_getaffinity(0)))" >> gpurun_out/r01_env.txt
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 > gpurun_out/r01_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r01_env.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r01_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r01_env.txt
PGA_VERBOSE=1 timeout 900 python bench.py > gpurun_out/r01_bench.json 2> gpurun_out/r01_bench.err; echo "bench rc=$?" >> gpurun_out/r01_env.txt
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01 -o r01 -- python $R/bench.py --cpu-budget 0 > $R/gpurun_out/r01_bench_prof.json 2> $R/gpurun_out/r01_bench_prof.err ); echo "prof rc=$?" >> gpurun_out/r01_env.txt
find gpurun_out/prof_r01 -name "*.db" -size +20M -delete
ls -laR gpurun_out/prof_r01 >> gpurun_out/r01_env.txt
tail -5 gpurun_out/r01_pytest_gpu.log; cat gpurun_out/r01_env.txt; cat gpurun_out/r01_bench.json
