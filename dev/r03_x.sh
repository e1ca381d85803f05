#!/bin/bash
# how much does a second (third) PROCESS on the same GPU add?  (hardware queues are per process)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for n in 2 3; do
  PGA_BENCH_SINGLE_DEVICE=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $n --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows > gpurun_out/r03_x_$n.json 2> gpurun_out/r03_x_$n.err
  python -c "import json; d=json.load(open('gpurun_out/r03_x_$n.json')); print($n, round(d['value'],3), round(d['ms_per_step']), d['n_matches_gathered'], d['rank0_seconds_per_step'])" || tail -5 gpurun_out/r03_x_$n.err
done
