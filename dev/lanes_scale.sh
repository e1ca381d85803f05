#!/bin/bash
# the lane kernel alone: junk end extensions (700 homologous bases, then random; band 1500, exact maximum) -- kernel time against problems per launch and workgroups per CU
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for wv in 1 2 3 4; do for n in 1024 4096; do
  echo "== PGA_C10_WAVES=$wv n=$n"; PGA_C10_WAVES=$wv PGA_VERBOSE=1 python dev/dp_probe.py 9900 $n 1500 0x40 0.0 2>&1 | grep "dp class" | tail -2
done; done
