#!/bin/bash
# junk end extensions (700 homologous bases, then random; band 1500, exact maximum) and full-length ones: the workgroup pipeline against the lane kernel
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for mode in force off; do for n in 1 64 512 4096; do
  echo "== PGA_PIPE=$mode n=$n junk"; PGA_BSTRIPS=off PGA_PIPE=$mode PGA_VERBOSE=1 python dev/dp_probe.py 9900 $n 1500 0x40 0.0 2>&1 | grep "dp class" | tail -1
done; done
for mode in force off; do
  echo "== PGA_PIPE=$mode n=4 full length (19 799 diagonals)"; PGA_BSTRIPS=off PGA_PIPE=$mode PGA_VERBOSE=1 python dev/dp_probe.py 9900 4 1500 0x40 1.0 2>&1 | grep "dp class" | tail -1
done
