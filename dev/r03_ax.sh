#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" timeout 900 python bench.py --steps 2 --warmup 1 --cpu-budget 0 --no-next-rows --leaf-only --cap-gbp 0.6 > gpurun_out/r03_ax_$name.json 2> gpurun_out/r03_ax_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/r03_ax_$name.json')); b=[x for x in d['batches_rank0'] if x['Mbp']>400]
st=d['stages_s']; n=len(b)
print('$name', round(d['value'],3), round(d['ms_per_step']), 'big batches', n, 'avg dur', round(sum(x['t1']-x['t0'] for x in b)/max(n,1),3), {a: round(v/max(n,1),3) for a,v in st.items()})" || tail -5 gpurun_out/r03_ax_$name.err
}
run s1 PGA_BENCH_SLOTS=1
run s2 PGA_BENCH_SLOTS=2
run s3 PGA_BENCH_SLOTS=3
run s6 PGA_BENCH_SLOTS=6
